import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import genie_smem_amd as g
from genie_smem_amd import synth
lib = g._native.lib()
P = lambda t: C.c_void_p(t.data_ptr())
for n, N in ((100_000, 1_000_000), (1_000_000, 4_000_000)):
    ref = synth.synth_ref(n, n)
    ix = g.GenieIndex.build(ref, 15); ix.train_rmi([1000]); ix = ix.to("cuda")
    L = 150
    reads = synth.reads_from_ref_device(torch.as_tensor(ref).cuda(), N, L, 1002)
    status = torch.zeros(N, dtype=torch.int32, device="cuda"); offsets = torch.empty(N + 1, dtype=torch.int64, device="cuda")
    out = torch.empty((N * 20, 4), dtype=torch.int32, device="cuda")
    wsb = int(lib.genie_find_smems_workspace_bytes(N, L)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ix.set_option(5, 1)
    info = ix.launch_info("lut", L)
    print(info)
    for dbg in (128 + 32, 128 + 32 + 256, 128 + 32 + 512, 128 + 32 + 768, 128 + 32 + 64 + 256):
        ix.set_option(7, dbg)
        for _ in range(3):
            rc = lib.genie_find_smems_csr(ix._h, 1, P(reads), None, N, L, L, 1, P(offsets), P(out), out.shape[0], P(status), P(ws), wsb, sp)
        torch.cuda.synchronize()
        W = info["grid"] * 8
        t = ws[:8 * W].view(torch.int32).cpu().numpy().astype(np.int64)
        end, start = t[:W], t[W:]
        t0 = start.min()
        e = (end - t0); s0 = start - t0
        dur = e.max()
        print(n, "dbg", dbg, "waves", W, "start spread", s0.max(), "end: min %d mean %.0f p50 %d p90 %d p99 %d max %d  -> mean/max %.3f" % (e.min(), e.mean(), np.percentile(e, 50), np.percentile(e, 90), np.percentile(e, 99), e.max(), e.mean() / e.max()))
        eb = e.reshape(-1, 8).max(1)
        print("   per block end: mean %.0f max %d; per-wave busy std %.0f" % (eb.mean(), eb.max(), (e - s0).std()))
        eb2 = e.reshape(-1, 8).max(1)
        for x in range(3):
            print("     blocks %d .. %d: mean end %.0f  min %d max %d" % (256 * x, 256 * x + 255, eb2[256 * x:256 * x + 256].mean(), eb2[256 * x:256 * x + 256].min(), eb2[256 * x:256 * x + 256].max()))

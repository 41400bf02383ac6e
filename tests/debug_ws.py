import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import golden_util as G
import genie_smem_amd as g
from genie_smem_amd import synth as B
N_ = g._native
d, _ = G.load("syn100k_K15")
ix = g.GenieIndex.build(d["ref_codes"], 15).to("cuda")
n = int(os.environ.get("NREADS", "1000000")); L = 150
rd_np = B.reads_from_ref_fast(d["ref_codes"], n, L, 1002)
if os.environ.get("REVERSE"): rd_np = rd_np[::-1].copy()
rd = torch.as_tensor(rd_np).cuda()
lib = N_.lib()
P = lambda t: C.c_void_p(t.data_ptr())
dev = torch.device("cuda", 0)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for trial, mode in enumerate(os.environ.get("MODES", "bwa,bwa,lut").split(",")):
    counts = torch.empty(n, dtype=torch.int32, device=dev); slots = torch.empty((n, L, 4), dtype=torch.int32, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    wsb = int(lib.genie_find_smems_workspace_bytes(n, L))
    ws = torch.full((wsb,), 0xAB, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    N_.check(lib.genie_find_smems(ix._h, N_.MODES[mode], P(rd), None, n, L, L, 1, P(counts), P(slots), L, P(status), P(ws), wsb, sp), "find")
    torch.cuda.synchronize()
    sl = slots.cpu().numpy(); c = counts.cpu().numpy()
    neg = np.nonzero((sl[:, :, 2] < 0) & (np.arange(L)[None, :] < c[:, None]))
    u = np.unique(neg[0])
    print(os.environ.get("GENIE_DEBUG"), n, trial, mode, "| SMEMs with lo<0:", len(neg[0]), "reads", len(u), "first", u[:4], "last", u[-2:] if len(u) else "")
    if mode == "bwa" and len(u):
        lo = sl[neg[0], neg[1], 2]
        print("  sample lo/hi:", lo[:10].tolist(), sl[neg[0], neg[1], 3][:10].tolist())
        print("  retry outcomes: second attempt ok:", int((lo < -1000000).sum()), " second attempt also -1:", int((lo == -1000000 + 1).sum()), "other", int(((lo > -1000000)).sum()))
        print("  r%64 hist (8 bins):", np.histogram(u % 64, bins=8, range=(0,64))[0].tolist())
        print("  wave-in-block hist:", np.bincount((u % 256) // 64, minlength=4).tolist())
        print("  block%8 hist:", np.bincount((u // 256) % 8, minlength=8).tolist())
        print("  block%32 hist:", np.bincount((u // 256) % 32, minlength=32).tolist())
        b = u // 256
        print("  failing blocks:", len(np.unique(b)), "of", (n + 255) // 256, "| fails per failing block: mean", len(u) / len(np.unique(b)))
        t = neg[1]
        print("  slot index hist:", np.bincount(t, minlength=40)[:40].tolist())
        ks = sl[neg[0], neg[1], 0]; js = sl[neg[0], neg[1], 1]
        print("  len hist:", np.bincount(js - ks, minlength=40)[:45].tolist())
        print("  start>>5 hist:", np.bincount(ks >> 5, minlength=5).tolist(), " (start&31) hist/4:", np.histogram(ks & 31, bins=8, range=(0,32))[0].tolist())

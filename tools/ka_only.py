"""Run the hot path a few times without checking results (timing experiments under rocprofv3)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import genie_smem_amd as g
from genie_smem_amd import synth
n = int(os.environ.get("REF_N", 100_000)); N = int(os.environ.get("READS", 1_000_000)); L = int(os.environ.get("READ_LEN", 150))
mode = os.environ.get("MODE", "lut")
ref = synth.synth_ref(n, n)
ix = g.GenieIndex.build(ref, 15, table_bits=int(os.environ.get("TABLE_BITS", "0")), table_format=os.environ.get("FMT", "auto"))
ix.train_rmi([1000])
ix = ix.to("cuda")
kind = os.environ.get("KIND", "fromref")
if kind == "random":
    reads = torch.as_tensor(np.random.default_rng(7).integers(0, 4, (N, L)).astype(np.uint8)).cuda()
else:
    reads = torch.as_tensor(synth.reads_from_ref_fast(ref, N, L, 1002)).cuda()
lib = g._native.lib()
for k, v in ((4, "GROUP_POS"), (5, "SEARCH_ONLY"), (2, "SEARCH_ALL"), (6, "BPC"), (7, "DBG"), (8, "SCHED")):
    if os.environ.get(v):
        ix.set_option(k, int(os.environ[v]))
status = torch.empty(N, dtype=torch.int32, device="cuda"); offsets = torch.empty(N + 1, dtype=torch.int64, device="cuda")
out = torch.empty((N * max(40, L // 3), 4), dtype=torch.int32, device="cuda")
wsb = int(lib.genie_find_smems_workspace_bytes(N, L)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
if os.environ.get("PACKED"):
    from genie_smem_amd import packing
    pk = torch.as_tensor(packing.pack_reads(reads.cpu().numpy())).cuda()
    c8 = torch.empty(N, dtype=torch.uint8, device="cuda"); s8 = torch.empty(N, dtype=torch.uint8, device="cuda")
    r8 = torch.empty((out.shape[0], 8), dtype=torch.uint8, device="cuda"); tot = torch.zeros(2, dtype=torch.int64, device="cuda")
    esc = torch.empty((1024, 2), dtype=torch.int64, device="cuda")
    for _ in range(int(os.environ.get("ITERS", 8))):
        rc = lib.genie_find_smems_packed(ix._h, g._native.MODES[mode], P(pk), None, N, pk.shape[1], L, 1, P(c8), P(s8), P(r8), r8.shape[0], P(tot), P(esc), 1024, P(ws), wsb, sp)
        assert rc in (0, 1), rc
    torch.cuda.synchronize()
    print("ok packed")
    sys.exit(0)
for _ in range(int(os.environ.get("ITERS", 8))):
    rc = lib.genie_find_smems_csr(ix._h, g._native.MODES[mode], P(reads), None, N, L, L, 1, P(offsets), P(out), out.shape[0], P(status), P(ws), wsb, sp)
    assert rc in (0, 1), rc            # 1 = GENIE_W_SEARCH_ONLY
torch.cuda.synchronize()
print("ok")

"""Device time of genie_find_smems_packed vs genie_find_smems_csr on resident inputs, and the copy stages of the
from-host pipeline one by one (BASELINE config 1 batch)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import genie_smem_amd as g
from genie_smem_amd import synth, packing
n, N, L = 100_000, 1_000_000, 150
ref = synth.synth_ref(n, n)
ix = g.GenieIndex.build(ref, 15); ix.train_rmi([1000]); ix = ix.to("cuda", seed_table=False)
rd = synth.reads_from_ref_device(ref, N, L, 1002, device="cuda")
pk_h = torch.as_tensor(packing.pack_reads(rd.cpu().numpy())).pin_memory()
pk = pk_h.cuda()
lib = g._native.lib(); P = lambda t: C.c_void_p(t.data_ptr())
cap = N * 16
ws_b = int(lib.genie_find_smems_workspace_bytes(N, L)); ws = torch.empty(ws_b, dtype=torch.uint8, device="cuda")
c8 = torch.empty(N, dtype=torch.uint8, device="cuda"); s8 = torch.empty(N, dtype=torch.uint8, device="cuda")
r8 = torch.empty((cap, 8), dtype=torch.uint8, device="cuda"); tot = torch.zeros(2, dtype=torch.int64, device="cuda"); esc = torch.empty((1024, 2), dtype=torch.int64, device="cuda")
st = torch.empty(N, dtype=torch.int32, device="cuda"); off = torch.empty(N + 1, dtype=torch.int64, device="cuda"); rows = torch.empty((cap, 4), dtype=torch.int32, device="cuda")
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def packed(): g._native.check(lib.genie_find_smems_packed(ix._h, 1, P(pk), None, N, pk.shape[1], L, 1, P(c8), P(s8), P(r8), cap, P(tot), P(esc), 1024, P(ws), ws_b, sp), "p")
def csr(): g._native.check(lib.genie_find_smems_csr(ix._h, 1, P(rd), None, N, L, L, 1, P(off), P(rows), cap, P(st), P(ws), ws_b, sp), "c")
def timeit(f, k=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
print("packed call ms", timeit(packed), " csr call ms", timeit(csr))
total = int(tot[0].item()); print("rows", total)
h_r8 = torch.empty((cap, 8), dtype=torch.uint8).pin_memory(); h_rows = torch.empty((cap, 4), dtype=torch.int32).pin_memory()
h_c8 = torch.empty(N, dtype=torch.uint8).pin_memory()
print("H2D packed 40MB ms", timeit(lambda: pk.copy_(pk_h, non_blocking=True)))
print("D2H rows8 %d MB ms" % (cap * 8 >> 20), timeit(lambda: h_r8.copy_(r8, non_blocking=True)))
print("D2H rows8 used part ms", timeit(lambda: h_r8[:total].copy_(r8[:total], non_blocking=True)))
print("D2H rows16 %d MB ms" % (cap * 16 >> 20), timeit(lambda: h_rows.copy_(rows, non_blocking=True)))
print("D2H counts 1MB ms", timeit(lambda: h_c8.copy_(c8, non_blocking=True)))

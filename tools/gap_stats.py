#!/usr/bin/env python3
"""How many positions the sampled search has to search: distribution of the second work list's size
(gaps whose ends disagree) over groups of five 150-base from-ref reads, read off the fwd[] rows that
the search kernel leaves in the workspace."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genie_smem_amd as g
from genie_smem_amd import synth
n, L, N, K = 100_000, 150, 100_000, 15
ref = synth.synth_ref(n, n)
ix = g.GenieIndex.build(ref, K).to("cuda")
lib = g._native.lib()
reads = torch.as_tensor(synth.reads_from_ref_fast(ref, N, L, 1002)).cuda()
ws_b = int(lib.genie_find_smems_workspace_bytes(N, L))
status = torch.empty(N, dtype=torch.int32, device="cuda"); offsets = torch.empty(N + 1, dtype=torch.int64, device="cuda")
out = torch.empty((N * 60, 4), dtype=torch.int32, device="cuda"); ws = torch.zeros(ws_b, dtype=torch.uint8, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
g._native.check(lib.genie_find_smems_csr(ix._h, 1, P(reads), None, N, L, L, 1, P(offsets), P(out), out.shape[0], P(status), P(ws), ws_b, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "csr")
torch.cuda.synchronize()
fs = 156
fw = ws[:N * fs].cpu().numpy().reshape(N, fs)[:, :L].astype(np.int64)
assert (np.diff(fw, axis=1) >= 0).all()
S = 4
left = np.arange(0, L, S)
interior = []
for r0 in range(0, N, 5):
    tot = 0
    for r in range(r0, min(N, r0 + 5)):
        row = fw[r]
        for a in left:
            ic = min(S - 1, L - 1 - a)
            if ic <= 0: continue
            if a + S < L and row[a] == row[a + S]: continue
            tot += ic
    interior.append(tot)
interior = np.asarray(interior)
print("phase-1 entries per group of 5: mean %.1f  p50 %d  p90 %d  max %d ; fraction > 192: %.3f ; distinct fwd values per read %.1f" % (
    interior.mean(), np.median(interior), np.percentile(interior, 90), interior.max(), (interior > 192).mean(), np.mean([len(np.unique(x)) for x in fw[:2000]])))

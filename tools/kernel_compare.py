#!/usr/bin/env python3
"""Search-kernel variants on two read distributions: ms per batch (whole CSR call; KC_L, KC_N select read length and count)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genie_smem_amd as g
from genie_smem_amd import synth

def main():
    n, K = 100_000, 15
    L, N = int(os.environ.get("KC_L", "150")), int(os.environ.get("KC_N", "1000000"))
    ref = synth.synth_ref(n, n)
    ix = g.GenieIndex.build(ref, K)
    ix.train_rmi([1000])
    ix = ix.to("cuda")
    lib = g._native.lib()
    sets = {"from-ref": synth.reads_from_ref_fast(ref, N, L, 1002), "random": synth.reads_random(N, L, 7),
            "exact": np.stack([ref[s:s + L] for s in np.random.default_rng(3).integers(0, n - L, N)]).astype(np.uint8)}
    ws_b = int(lib.genie_find_smems_workspace_bytes(N, L))
    status = torch.empty(N, dtype=torch.int32, device="cuda"); offsets = torch.empty(N + 1, dtype=torch.int64, device="cuda")
    out = torch.empty((N * max(60, L // 2), 4), dtype=torch.int32, device="cuda"); ws = torch.empty(ws_b, dtype=torch.uint8, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    s0 = torch.cuda.current_stream()
    for name, rd in sets.items():
        reads = torch.as_tensor(rd).cuda()
        for mode in (0, 1, 2):
            res = []
            for sampled in (False, True):
                ix.set_option(g._native.OPT_SEARCH_ALL, 0 if sampled else 1)
                def call():
                    g._native.check(lib.genie_find_smems_csr(ix._h, mode, P(reads), None, N, L, L, 1, P(offsets), P(out), out.shape[0],
                                                             P(status), P(ws), ws_b, C.c_void_p(s0.cuda_stream)), "csr")
                for _ in range(2): call()
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(10): call()
                torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 10 * 1e3)
                res.append(int(offsets[-1]))
            print("%-9s mode %d: search-all %.3f ms, sampled (default) %.3f ms  (rows %d / %d)" % (name, mode, res[0], res[2], res[1], res[3]), flush=True)
main()

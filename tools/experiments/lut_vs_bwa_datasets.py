import sys, numpy as np
import os; R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_util as G
from oracle import oracle as orc
from genie_smem_amd import synth as B
orc.build()
tot = diff = 0
for ds in ("syn100k_K15", "big100k_K15", "syn10k_K8", "medium_K6"):
    d, _ = G.load(ds)
    ref, K = d["ref_codes"], int(d["K"])
    o = orc.Oracle(ref, K)
    rng = np.random.default_rng(1)
    for kind in ("fromref", "random", "exact", "mut"):
        for L in (max(K, 20), 60, 150, 250):
            n = 3000
            if kind == "fromref": rd = B.reads_from_ref(ref, n, L, 10 + L)
            elif kind == "random": rd = B.reads_random(n, L, 11 + L)
            else:
                if L > len(ref): continue
                st = rng.integers(0, len(ref) - L + 1, n)
                rd = np.stack([ref[s:s + L] for s in st]).astype(np.uint8)
                if kind == "mut":
                    for r in range(n):
                        for _ in range(int(rng.integers(1, 4))):
                            rd[r, int(rng.integers(0, L))] = rng.integers(0, 4)
            ca, ra = o.find_smems_batch("bwa", rd, nthreads=8)
            cl, rl = o.find_smems_batch("lut", rd, nthreads=8)
            nd = 0
            for r in range(n):
                if ca[r] < 0 or cl[r] < 0: continue
                tot += 1
                if ca[r] != cl[r] or not (ra[r, :ca[r]] == rl[r, :cl[r]]).all(): nd += 1
            diff += nd
            if nd: print(ds, kind, L, "differ:", nd, "of", n)
print("total", tot, "differ", diff)

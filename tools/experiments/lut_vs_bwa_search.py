import sys, numpy as np, time
import os; R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import oracle as orc
orc.build()
rng = np.random.default_rng(12345)
tot = diff = 0
t0 = time.time()
examples = []
it = 0
while time.time() - t0 < 420:
    it += 1
    n = int(rng.integers(20, 400))
    sigma = int(rng.choice([2, 3, 4]))                 # small alphabets: many repeats and ties
    ref = rng.integers(0, sigma, n).astype(np.uint8)
    if rng.random() < 0.3:                             # tandem repeats
        unit = rng.integers(0, sigma, int(rng.integers(1, 7))).astype(np.uint8)
        ref = np.concatenate([ref[:n // 3], np.tile(unit, int(rng.integers(3, 30))), ref[n // 3:]]).astype(np.uint8)
    if len(set(ref.tolist())) < 2: continue
    K = int(rng.integers(2, 9))
    if len(ref) < K + 2: continue
    try:
        o = orc.Oracle(ref, K)
    except Exception as ex:
        continue
    L = int(rng.integers(K, 80))
    N = 400
    rd = rng.integers(0, sigma, (N, L)).astype(np.uint8)
    # half of the reads: stitched from reference pieces
    for r in range(0, N, 2):
        buf = []
        while sum(len(b) for b in buf) < L:
            p = int(rng.integers(0, len(ref))); buf.append(ref[p:p + int(rng.integers(1, 25))])
        rd[r] = np.concatenate(buf)[:L]
    ca, ra = o.find_smems_batch("bwa", rd, nthreads=8)
    cl, rl = o.find_smems_batch("lut", rd, nthreads=8)
    for r in range(N):
        if ca[r] < 0 or cl[r] < 0: continue
        tot += 1
        if ca[r] != cl[r] or not (ra[r, :ca[r]] == rl[r, :cl[r]]).all():
            diff += 1
            if len(examples) < 3:
                examples.append((ref.tolist(), K, rd[r].tolist(), ra[r, :ca[r]].tolist(), rl[r, :cl[r]].tolist()))
print("iterations", it, "reads", tot, "differ", diff)
for e in examples: print(e)

#!/usr/bin/env python3
"""One-off wider sweep of tests/test_gpu_parity.py::test_randomized_configurations: SEEDS x (reference size and content,
K, read length incl. long reads, read kind) against the CPU oracle.  usage: SEEDS="1 2 3" python tools/experiments/fuzz_gpu.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import genie_smem_amd as pkg
import oracle as oracle_mod

def rows_per_read(offsets, smems):
    o, s = offsets.cpu().numpy(), smems.cpu().numpy()
    return [s[o[r]:o[r + 1]] for r in range(len(o) - 1)]

bad = cases = 0
for seed in [int(x) for x in os.environ.get("SEEDS", "1").split()]:
    rng = np.random.default_rng(seed)
    for n in (5, 37, 400, 3000, 70000):
        for rep in range(4 if n < 70000 else 2):
            alphabet = 4 if rng.random() < 0.7 else (3 if rng.random() < 0.7 else 2)
            ref = rng.integers(0, alphabet, n).astype(np.uint8)
            if n >= 400 and rep % 2:
                unit = rng.integers(0, alphabet, int(rng.integers(1, 7))).astype(np.uint8)
                span = int(min(n // 2, 600))
                ref[:span] = np.tile(unit, span // len(unit) + 1)[:span]
            K = int(rng.choice([k for k in (3, 6, 8, 12, 15) if k <= n]))
            L = int(rng.integers(max(K, 1), 256)) if rng.random() < 0.75 else int(rng.integers(256, 1200))
            N = 61
            reads = np.empty((N, L), np.uint8)
            for r in range(N):
                kind = r % 3
                if kind == 0 or n < 3:
                    reads[r] = rng.integers(0, alphabet, L)
                else:
                    buf = []
                    while sum(len(b) for b in buf) < L:
                        p0 = int(rng.integers(0, n)); s0 = int(rng.integers(1, 31 if kind == 1 else 4 * L))
                        buf.append(ref[p0:p0 + s0])
                    reads[r] = np.concatenate(buf)[:L]
            lens = None
            if rng.random() < 0.4:
                lens = rng.integers(0, L + 1, N).astype(np.int32); lens[0] = L
            fmt = "compact" if rng.random() < 0.7 else "wide"            # both forms of the match table
            n_extra = 0
            if rng.random() < 0.35 and n >= 3000 and alphabet == 4 and K >= 8:   # (the literal oracle's check_sequential is quadratic in a K-mer's
                # occurrences: small K or a small alphabet on a big reference takes it minutes per case)
                # a bigger reference now and then: the table outgrows
                n_extra = int(rng.integers(200_000, 600_000))            # an XCD's L2, overflow blocks appear
                ref = np.concatenate([ref, rng.integers(0, alphabet, n_extra).astype(np.uint8)])
            ix = pkg.GenieIndex.build(ref, K, table_format=fmt)
            coefs, icpts, _, _, _ = ix.train_rmi([10])
            ix = ix.to("cuda", seed_table=bool(rng.random() < 0.5))
            o = oracle_mod.Oracle(ref, K)
            o.set_rmi([10], coefs, icpts)
            for algo in ("bwa", "lut", "rmi"):
                ml = int(rng.integers(1, 20)) if algo == "bwa" else 1
                offsets, smems, st = ix.find_smems(algo, reads, lens=lens, min_len=ml)
                rows = rows_per_read(offsets, smems)
                counts, out = o.find_smems_batch(algo, reads, lens=lens, min_len=ml, nthreads=8)
                st = st.cpu().numpy()
                for r in range(N):
                    ok = (st[r] != 0) if counts[r] < 0 else (st[r] == 0 and rows[r].tolist() == out[r, :counts[r]].tolist())
                    if not ok:
                        bad += 1
                        print("MISMATCH seed", seed, "n", len(ref), fmt, "K", K, "L", L, algo, "read", r, "status", st[r], "oracle count", counts[r])
                if L <= 255:                                             # the packed entry point gives the same rows
                    import torch
                    from genie_smem_amd import packing
                    c8, s8, r8, esc = ix.find_smems_packed(algo, torch.as_tensor(packing.pack_reads(reads)).cuda(), L,
                                                           lens=None if lens is None else torch.as_tensor(lens).cuda(), min_len=ml)
                    off2, rows2 = packing.unpack_rows(c8.cpu().numpy(), r8.cpu().numpy(), esc.cpu().numpy())
                    c6, s6, r6, esc6 = ix.find_smems_packed(algo, torch.as_tensor(packing.pack_reads(reads)).cuda(), L,
                                                            lens=None if lens is None else torch.as_tensor(lens).cuda(), min_len=ml, row_bytes=6)
                    off6, rows6 = packing.unpack_rows(c6.cpu().numpy(), r6.cpu().numpy(), esc6.cpu().numpy(), row_bytes=6)
                    if not (np.array_equal(off2, offsets.cpu().numpy()) and np.array_equal(rows2, smems.cpu().numpy())
                            and np.array_equal(off6, off2) and np.array_equal(rows6, rows2)
                            and np.array_equal(s8.cpu().numpy().astype(np.int32), st)):
                        bad += 1
                        print("PACKED MISMATCH seed", seed, "n", len(ref), fmt, "K", K, "L", L, algo)
                cases += 1
    print("seed", seed, "done:", cases, "cases so far,", bad, "mismatches", flush=True)
print("TOTAL", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)

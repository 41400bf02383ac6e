"""One traversal for the three modes: SMEM.get_smems_lut / get_smems_rmi
(SMEM/SMEM.py:20-384) emit the ordered SMEMs of SMEM.get_SMEMS(q, 1) (SMEM/SMEM.py:456-484).  The device
therefore runs the get_SMEMS loop for all three and has no K-frame state machine (SMEM.py:49-186), no
check_sequential (SMEM.py:196-202) and no stale-frame rule (SMEM.py:75, 94-95).  That equivalence is
load-bearing, so it is checked here on the LITERAL oracle (which keeps all three, pinned to the reference's
per-step traces by test_oracle_golden.py):

  * exhaustively on a small model -- EVERY reference of 1 .. 9 bases over {A, C} and of 1 .. 7 bases over
    {A, C, G}, EVERY read of K .. 9 bases over the same letters, K in {2, 3, 4}: lut == bwa row for row, and
    rmi == bwa with a model trained on that reference (numpy restatement of RMI.fit, experts [4]);
  * on a fixed, seeded set of adversarial random cases (tandem repeats, K up to 8, reads up to 80 bases).

CPU only; no GPU code is involved.  The enumeration is spread over worker processes (the oracle is fastest
single-threaded on reads this short); GENIE_EQUIV_WORKERS overrides the pool size.
"""
import itertools
import multiprocessing as mp
import os

import numpy as np

K_SET = (2, 3, 4)
MAX_READ = 9
SMALL_MODEL = ((2, 9), (3, 7))            # (letters, longest reference)


def _all_strings(sigma, max_len):
    """Every string of 1 .. max_len letters, zero padded to max_len, with its length."""
    rows, lens = [], []
    for L in range(1, max_len + 1):
        a = np.array(list(itertools.product(range(sigma), repeat=L)), np.uint8)
        pad = np.zeros((len(a), max_len), np.uint8)
        pad[:, :L] = a
        rows.append(pad)
        lens.append(np.full(len(a), L, np.int32))
    return np.concatenate(rows), np.concatenate(lens)


def _train(o, ref, K, experts=(4,)):
    """RMI_LUT.train_RMI (SMEM/RMI_LUT.py:36-50) on the oracle's own suffix array, fitted with the package's numpy
    restatement of RMI.fit; installs the model in the oracle.  False when the reference holds no K-mer."""
    from genie_smem_amd.rmi import RMI
    sa = o.suffix_array.astype(np.int64) - 1
    rows = np.nonzero(sa + K <= len(ref))[0]
    if len(rows) == 0:
        return False
    key = np.zeros(len(rows), np.int64)
    for j in range(K):
        key = (key << 2) | ref[sa[rows] + j].astype(np.int64)
    m = RMI(list(experts)).fit(key.astype(np.float64), rows.astype(np.float64))
    coefs, icpts = m.coefficients()
    o.set_rmi(list(experts), coefs, icpts)
    return True


def _same(ca, ra, cb, rb):
    """Row-for-row equality of two batch results (rows beyond a read's count are unspecified)."""
    if not (ca == cb).all():
        return False
    live = np.arange(ra.shape[1])[None, :] < np.maximum(ca, 0)[:, None]
    return bool((ra[live] == rb[live]).all())


def _check_refs(args):
    """Worker: every read against each of `refs` for every K; returns (reads compared, refused reads, first failure)."""
    sigma, refs = args
    from oracle import oracle as orc
    reads, lens = _all_strings(sigma, MAX_READ)
    by_k = {}
    for K in K_SET:
        sel = lens >= K
        by_k[K] = (np.ascontiguousarray(reads[sel]), np.ascontiguousarray(lens[sel]))
    compared = refused = 0
    for ref in refs:
        ref = np.asarray(ref, np.uint8)
        for K in K_SET:
            if len(ref) < K:
                continue                    # no K-mer table: the reference cannot run LUT/RMI mode at all
            rd, ln = by_k[K]
            o = orc.Oracle(ref, K)
            ca, ra = o.find_smems_batch("bwa", rd, lens=ln)
            cl, rl = o.find_smems_batch("lut", rd, lens=ln)
            if not _same(ca, ra, cl, rl):
                return compared, refused, ("lut", ref.tolist(), K)
            if _train(o, ref, K):
                cr, rr = o.find_smems_batch("rmi", rd, lens=ln)
                if not _same(ca, ra, cr, rr):
                    return compared, refused, ("rmi", ref.tolist(), K)
            compared += int((ca >= 0).sum())
            refused += int((ca < 0).sum())          # a base that never occurs in this reference: every mode refuses
    return compared, refused, None


def _workers():
    return max(1, int(os.environ.get("GENIE_EQUIV_WORKERS", min(8, os.cpu_count() or 1))))


def test_exhaustive_small_model(oracle_mod, capsys):
    oracle_mod.lib()                                     # built before the workers fork
    jobs = []
    n_refs = 0
    for sigma, max_ref in SMALL_MODEL:
        refs, lens = _all_strings(sigma, max_ref)
        refs = [refs[i, :lens[i]].tolist() for i in range(len(refs))]
        n_refs += len(refs)
        chunk = 16 if sigma == 3 else 64
        jobs += [(sigma, refs[i:i + chunk]) for i in range(0, len(refs), chunk)]
    compared = refused = 0
    with mp.get_context("fork").Pool(_workers()) as pool:
        for c, r, bad in pool.imap_unordered(_check_refs, jobs):
            assert bad is None, f"{bad[0]} traversal differs from get_SMEMS on reference {bad[1]}, K = {bad[2]}"
            compared += c
            refused += r
    with capsys.disabled():
        print(f"\n[mode equivalence] {n_refs} references x every read of K..{MAX_READ} bases x K in {K_SET}: "
              f"{compared} reads with lut == bwa == rmi row for row, {refused} refused alike (absent base)")
    assert compared > 50_000_000 and n_refs == 1022 + 3279


def test_seeded_adversarial_search(oracle_mod):
    """Fixed iteration count and seed (the same inputs on every host): tiny references over 2 .. 4 letters with tandem
    repeats, K = 2 .. 8, stitched and random reads of up to 80 bases (tools/experiments/lut_vs_bwa_search.py ran this
    search over 26 M reads)."""
    rng = np.random.default_rng(2024)
    total = cases = 0
    while cases < 160:
        n = int(rng.integers(20, 300))
        sigma = int(rng.choice([2, 3, 4]))
        ref = rng.integers(0, sigma, n).astype(np.uint8)
        if rng.random() < 0.3:
            unit = rng.integers(0, sigma, int(rng.integers(1, 7))).astype(np.uint8)
            ref = np.concatenate([ref[:n // 3], np.tile(unit, int(rng.integers(3, 30))), ref[n // 3:]]).astype(np.uint8)
        K = int(rng.integers(2, 9))
        if len(set(ref.tolist())) < 2 or len(ref) < K + 2:
            continue
        cases += 1
        o = oracle_mod.Oracle(ref, K)
        L = int(rng.integers(K, 80))
        rd = rng.integers(0, sigma, (200, L)).astype(np.uint8)
        for r in range(0, 200, 2):
            buf = []
            while sum(len(b) for b in buf) < L:
                p = int(rng.integers(0, len(ref)))
                buf.append(ref[p:p + int(rng.integers(1, 25))])
            rd[r] = np.concatenate(buf)[:L]
        ca, ra = o.find_smems_batch("bwa", rd)
        cl, rl = o.find_smems_batch("lut", rd)
        assert _same(ca, ra, cl, rl), (ref.tolist(), K)
        if _train(o, ref, K, experts=(8,)):
            cr, rr = o.find_smems_batch("rmi", rd)
            assert _same(ca, ra, cr, rr), ("rmi", ref.tolist(), K)
        total += int((ca >= 0).sum())
    assert total > 25_000

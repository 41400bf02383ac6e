"""score_LUT -- the reference's benchmark driver (reference SMEM/SMEM.py:508-539) on the drop-in
classes: generate `num_iter` queries from the reference sequence, run the three SMEM algorithms on
each, check that they return the same number of SMEMs (the reference's own consistency check,
:535-536) and return the average seconds per query of (LUT, BWA, RMI).

    python -m genie_smem_amd.score big_data.fa --iters 10 --size 2000 [--data-dir data] [--batched]

`--batched` additionally times the batched entry points (one call for all queries), which is how
the GPU path is meant to be used.
"""
import argparse
import datetime
import time

from .exact_match import ExactMatch
from .rmi_lut import RMI_LUT
from .smem import SMEM, create_query_from_ref


def score_LUT(num_iter, ref_file="big_data.fa", query_size=2000, data_dir="data", lut_size=None, experts=(10, 100),
              batched=False):
    match = ExactMatch(ref_file, data_dir=data_dir)
    try:
        match.load_fm_index()
    except FileNotFoundError:
        match.load_ref_sequence()                       # no FM json: build the index natively
    match.load_ref_sequence()
    smem = SMEM(match, lut_size=lut_size)
    rmi = RMI_LUT(list(experts), smem.lut.lut_size, ref_file, matcher=match)
    rmi.train_RMI()
    smem.rmi_lut = rmi
    ref = match.ref_sequence[:-1]                       # the reference draws from ref + "$" (SMEM.py:517)
    queries = [create_query_from_ref(ref, query_size) for _ in range(num_iter)]
    smem.get_smems_lut(queries[0])                      # first call uploads the index
    smem.get_smems_rmi(queries[0])
    lut_total = og_total = rmi_total = 0.0
    for q in queries:
        start = datetime.datetime.now()
        a = smem.get_smems_lut(q)
        mid = datetime.datetime.now()
        b = smem.get_SMEMS(q, 1)
        end = datetime.datetime.now()
        c = smem.get_smems_rmi(q)
        after = datetime.datetime.now()
        lut_total += (mid - start).total_seconds()
        og_total += (end - mid).total_seconds()
        rmi_total += (after - end).total_seconds()
        if len(a) != len(b) or len(b) != len(c):
            raise Exception("the three SMEM algorithms disagree")
    out = (lut_total / num_iter, og_total / num_iter, rmi_total / num_iter)
    if batched:
        import torch
        t = []
        for fn in (smem.find_smems_lut, lambda r: smem.find_smems_bwa(r, 1), smem.find_smems_rmi):
            fn(queries)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(queries)
            torch.cuda.synchronize()
            t.append((time.perf_counter() - t0) / num_iter)
        out = out + tuple(t)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ref_file")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--size", type=int, default=2000)
    ap.add_argument("--data-dir", default="data")
    ap.add_argument("--lut-size", type=int, default=None, help="build the K-mer table instead of loading <stem>-LUT.json")
    ap.add_argument("--batched", action="store_true")
    args = ap.parse_args()
    res = score_LUT(args.iters, args.ref_file, args.size, args.data_dir, args.lut_size, batched=args.batched)
    print("Lut time:")
    print(res[0])
    print("OG time")
    print(res[1])
    print("RMI time")
    print(res[2])
    if args.batched:
        print("batched seconds per query (LUT, BWA, RMI):", res[3:])


if __name__ == "__main__":
    main()

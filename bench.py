#!/usr/bin/env python3
"""bench.py -- query-bases/s of SMEM discovery on the BASELINE.json workload.

  python bench.py [--gpus N --steps K --warmup W] [--mode lut|rmi|bwa] [--config 1|2|3]

N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
one rank per GPU.  Rank 0 builds the index and broadcasts its image ONCE over RCCL (xGMI); after
that ranks never communicate inside the timed region (reads are independent units, weak scaling:
every rank processes its own batch of the same shape).

A "step" = one pass of the hot path over one batch: ONE genie_find_smems_csr call (match statistics,
traversal, offsets scan, interval search writing the CSR rows) with the reads already resident in HBM.
Rank 0 prints ONE JSON line (schema in the task contract) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import genie_smem_amd as g                      # noqa: E402
from genie_smem_amd import parallel, synth       # noqa: E402

HBM_PEAK_GBS = 8000.0                            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# BASELINE.json `configs`: index 1 is the one the headline metric is quoted on.
CONFIGS = {
    1: dict(name="100 kb reference, 1M x 150 bp reads, LUT-SMEM", n=100_000, ref_seed=100_000, reads=1_000_000,
            L=150, mode="lut", read_seed=1002),
    2: dict(name="100 kb reference, 1M x 150 bp reads, RMI-SMEM (experts [1000])", n=100_000, ref_seed=100_000,
            reads=1_000_000, L=150, mode="rmi", read_seed=1003),
    3: dict(name="1 Mb reference, 10M x 150 bp reads, RMI-SMEM", n=1_000_000, ref_seed=1_000_000, reads=10_000_000,
            L=150, mode="rmi", read_seed=1004),
}
K = 15
EXPERTS = [1000]


def search_kernel_name(mode_id, L, search_all):
    """Instantiation of the dominant (search) kernel the library picks for fixed-length reads of L bases."""
    ns = min(4, (L + 63) // 64)
    if search_all:
        pair = "true" if L <= 255 and L - 64 * (ns - 1) <= 32 else "false"
        return f"match_stats_kernel<{mode_id}, {ns}, {'true' if L > 255 else 'false'}, {pair}, false>"
    if L > 255:
        return f"match_stats_sampled_long_kernel<{mode_id}, false>"
    grp = min(8, max(1, 192 // ((L - 1) // 4 + 1)))
    return f"match_stats_sampled_kernel<{mode_id}, false, {5 if grp == 5 else 0}>"


def build_index(cfg, device):
    ref = synth.synth_ref(cfg["n"], cfg["ref_seed"])
    m = g.ExactMatch(f"REF_{cfg['n']}.fa", device=str(device))
    m.set_reference("".join("ACGT"[c] for c in ref))
    rl = g.RMI_LUT(EXPERTS, K, m.ref_seq_file, matcher=m)
    rl.train_RMI()
    return ref, rl._index(), rl


def cpu_baseline(ref, cfg, rl, mode, sample_reads):
    """The CPU oracle (a C restatement of the reference's algorithm, kind 'port') timed on this
    host's cores on a bounded sample of the SAME workload.  Reported, not a target."""
    from oracle import oracle as orc
    orc.build()
    threads = max(1, min(os.cpu_count() or 1, 64))
    o = orc.Oracle(ref, K)
    coefs, icpts = rl.rmi.coefficients()
    o.set_rmi(EXPERTS, coefs, icpts)
    rd = synth.reads_from_ref_fast(ref, sample_reads, cfg["L"], cfg["read_seed"])
    o.find_smems_batch(mode, rd[:2000], nthreads=threads)               # warm-up
    t0 = time.perf_counter()
    counts, _ = o.find_smems_batch(mode, rd, nthreads=threads)
    dt = time.perf_counter() - t0
    assert (counts >= 0).all()
    return {"value": sample_reads * cfg["L"] / dt, "unit": "query-bases/s", "cores": threads, "kind": "port",
            "sample": f"{sample_reads} x {cfg['L']} bp reads of the same from-ref distribution, mode {mode}, "
                      f"OpenMP over reads, {dt:.2f} s wall"}, rd, counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--mode", default=None, choices=["bwa", "lut", "rmi"])
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step (default: the config's)")
    ap.add_argument("--read-len", type=int, default=None, help="off-config read length (sweeps only; named in config.workload)")
    ap.add_argument("--cpu-sample", type=int, default=300_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--search-all", action="store_true", help="GENIE_OPT_SEARCH_ALL: search every read position (A/B runs)")
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    if args.read_len and args.read_len != cfg["L"]:
        cfg["L"] = args.read_len
        cfg["name"] += f" [read_len={args.read_len}: off-config sweep point]"
    mode = args.mode or cfg["mode"]
    n_reads = args.reads or cfg["reads"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)

    # ---- index: rank 0 builds, ONE broadcast of the image (RCCL over xGMI), no later traffic
    t_build = time.perf_counter()
    if rank == 0:
        ref, ix, rl = build_index(cfg, device)
    else:
        ref, ix, rl = None, None, None
    if world > 1:
        ix = parallel.broadcast_index(ix, src=0, device=device)
    t_build = time.perf_counter() - t_build
    if args.search_all:
        ix.set_option(g._native.OPT_SEARCH_ALL, 1)

    # ---- this rank's batch (weak scaling: same shape on every rank, seed + rank)
    ref_codes = synth.synth_ref(cfg["n"], cfg["ref_seed"])
    reads = torch.as_tensor(synth.reads_from_ref_fast(ref_codes, n_reads, cfg["L"], cfg["read_seed"] + rank)).to(device)
    L = cfg["L"]
    cap = L
    status = torch.empty(n_reads, dtype=torch.int32, device=device)
    offsets = torch.empty(n_reads + 1, dtype=torch.int64, device=device)
    out = torch.empty((n_reads * max(40, L // 3), 4), dtype=torch.int32, device=device)      # CSR rows (>= 3x the mean count)
    ws_bytes = int(g._native.lib().genie_find_smems_workspace_bytes(n_reads, L))
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=device)

    import ctypes as C
    lib = g._native.lib()
    P = lambda t: C.c_void_p(t.data_ptr())                                       # noqa: E731
    stream = torch.cuda.current_stream(device)
    sp = C.c_void_p(stream.cuda_stream)
    mode_id = g._native.MODES[mode]

    # Events: (a) around the whole hot-path call, (b) -- through the C ABI's profiling hook -- around
    # its dominant kernel (the suffix-array search), recorded by the library on the launch stream.
    def mk():
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)                       # materialise the hipEvent_t handle
        return e
    ev = [(mk(), mk()) for _ in range(args.steps)]
    ev_k = [(mk(), mk()) for _ in range(args.steps)]
    torch.cuda.synchronize(device)
    EV = lambda e: C.c_void_p(e.cuda_event)                                      # noqa: E731

    def step(i=None):
        if i is not None:
            lib.genie_index_set_stage_events(ix._h, EV(ev_k[i][0]), EV(ev_k[i][1]))
            ev[i][0].record(stream)
        g._native.check(lib.genie_find_smems_csr(ix._h, mode_id, P(reads), None, n_reads, L, L, 1, P(offsets), P(out),
                                                 out.shape[0], P(status), P(ws), ws_bytes, sp), "genie_find_smems_csr")
        if i is not None:
            ev[i][1].record(stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # ---- sanity on the last step's output (outside the timed region)
    assert int(status.abs().sum().item()) == 0, "a read was flagged"
    total = int(offsets[-1].item())
    assert total <= out.shape[0]
    lib.genie_index_set_stage_events(ix._h, None, None)
    path_ms = [a.elapsed_time(b) for a, b in ev]
    kern_ms = [a.elapsed_time(b) for a, b in ev_k]
    kern_ms_avg = float(np.mean(kern_ms))
    path_ms_avg = float(np.mean(path_ms))

    if rank == 0:
        smems_per_read = total / n_reads
        probes = math.ceil(math.log2(cfg["n"] + 1))
        # B_alg per read = L (read) + 16*S (output) + L * ceil(log2(n+1)) * 12 (SA + packed-ref probes).
        # The dominant kernel (match statistics) carries the read and probe terms; the 16*S output term
        # belongs to the traversal/interval kernels and is counted in `path`.
        bytes_search = L + L * probes * 12
        bytes_path = bytes_search + 16.0 * smems_per_read
        achieved = bytes_search * n_reads / (kern_ms_avg * 1e-3) / 1e9
        launch = ix.launch_info(mode, L)
        ns = min(4, (L + 63) // 64)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                traffic = json.load(fh).get(f"config{args.config}:{mode}:{n_reads}")
        except OSError:
            pass
        line = {
            "metric": "query-bases/sec SMEM discovery, 100kb ref x 150bp reads; bit-exact SMEM set",
            "value": world * n_reads * L * args.steps / dt,
            "unit": "query-bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int32 (2-bit packed bases, int32 SA rows; f64 only in the RMI predict)",
            "data": "synthetic",
            "config": {"workload": (cfg["name"] if mode == cfg["mode"] else cfg["name"] + f" [mode={mode}]") + (" [search-all]" if args.search_all else ""),
                       "reference_bases": cfg["n"], "reads_per_gpu_per_step": n_reads, "read_len": L, "K": K,
                       "mode": mode, "rmi_experts": EXPERTS, "read_distribution": "from-ref segments U{1..30}",
                       "parallelism": f"query-sharded x{world}, index replicated (one RCCL broadcast)",
                       "smems_per_read": round(smems_per_read, 3), "launch": launch,
                       "index_build_plus_broadcast_s": round(t_build, 3)},
            "roofline": {"bound": "hbm", "kernel": search_kernel_name(mode_id, L, args.search_all), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_read": bytes_search, "kernel_ms_avg": kern_ms_avg,
                         "kernel_ms_min": float(np.min(kern_ms)),
                         "note": "algorithmic bytes follow SURVEY 8(d) (a ceil(log2(n+1))-probe bound search per query "
                                 "position, 12 B per probe); the kernel serves most of them from LDS (the P-mer directory "
                                 "replaces the first ~14 of 17 probe steps) and L2 (the index is cache-resident), so "
                                 "achieved can exceed the HBM peak; `traffic` is the measured HBM bytes per launch",
                         "path": {"kernels": "match_stats + traverse + offsets scan + interval->CSR (one genie_find_smems_csr call)",
                                  "ms_avg": path_ms_avg, "alg_bytes_per_read": bytes_path,
                                  "achieved": bytes_path * n_reads / (path_ms_avg * 1e-3) / 1e9,
                                  "share_of_step": path_ms_avg / (dt / args.steps * 1e3)}},
        }
        if not args.no_cpu_baseline and world == 1:
            base, rd_s, cnt_s = cpu_baseline(ref_codes, cfg, rl, mode, args.cpu_sample)
            line["cpu_baseline"] = base
            # bonus parity check on the CPU sample: same reads through the GPU path
            o2, s2, st2 = ix.find_smems(mode, rd_s[:20000])
            assert (np.diff(o2.cpu().numpy()) == cnt_s[:20000]).all(), "GPU/oracle SMEM counts differ"
            line["cpu_baseline"]["parity_check"] = "20000 sample reads: GPU SMEM counts == oracle"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- query-bases/s of SMEM discovery on the BASELINE.json workloads.

  python bench.py [--gpus N --steps K --warmup W] [--config 1|2|3|4|5] [--mode lut|rmi|bwa]

N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
one rank per GPU.  Rank 0 builds the index and broadcasts its image ONCE over RCCL (xGMI); after that ranks
never communicate inside the timed region (reads are independent units).  Configs 1-3: weak scaling (every
rank processes its own batch of the config's shape).  Config 4 (BASELINE configs[4]): strong scaling -- ONE
80 M-read batch cut into contiguous shards with parallel.shard_bounds, per-rank outputs stay rank-local.  For N > 1 the
line's `config` carries what the process group reports (backend, world_size) and broadcast_ms / broadcast_bytes.

A "step" = one pass of the hot path over one batch: ONE genie_find_smems_csr call (match statistics, traversal,
scan of the block sums, interval search writing the offsets and the CSR rows) with the reads already resident in HBM
(one code per base; generated on the device).  Rank 0 prints ONE JSON line (schema in the task contract) with `roofline`
and `cpu_baseline`; the default run (config 1) also carries the other BASELINE configs measured in the same process
(`other_configs`: configs 2, 3 and 4 on one GPU; on N > 1 GPUs config 3, weak, and config 4, the 8 x 10^7-read batch cut
into N shards, strong -- each record with its own n_gpus / scaling / broadcast_ms) and `value_from_host` (pinned host buffers -> H2D -> call -> D2H through genie_find_smems_packed6 and,
beside it, through the CSR entry point: the host link's rate, never `value`).

What the roofline object says: the path moves few bytes and is bound by the L1 miss queue's
concurrency x latency (and, on tables beyond an XCD's L2, by the L2's misses), so
  achieved / frac   = COMPULSORY HBM bytes of the dominant kernel (what it must read and write: the reads, its
                      hand-off rows) / its measured time, against the 8 TB/s HBM peak -- always <= 1;
  traffic           = its measured HBM bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, separate passes;
                      from profiles/pmc_counters.json, written by tools/profile_round.sh on the same workload);
  step              = the same two for the whole call (all kernels), and their ratio;
  binding           = the counters that do bind (L1->L2 requests, their latency and number in flight, L2 misses, VALU issue);
  survey_8d         = the SURVEY 8(d) byte model (a 12-byte probe x ceil(log2 n) per position) for reference: the
                      kernel does not do that work (one 16-byte table entry per looked-up position instead), so it is NOT
                      reported as a fraction of anything.
"""
import argparse
import ctypes as C
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
L2_PEAK_GBS = 34500.0                            # aggregate L2 rate (MI355X_MICROARCH.md, section L2)

# BASELINE.json `configs`: index 1 is the one the headline metric is quoted on.
CONFIGS = {
    1: dict(name="100 kb reference, 1M x 150 bp reads, LUT-SMEM", ref="100kb", n=100_000, ref_seed=100_000, reads=1_000_000,
            L=150, mode="lut", read_seed=1002, scaling="weak"),
    2: dict(name="100 kb reference, 1M x 150 bp reads, RMI-SMEM (experts [1000])", ref="100kb", n=100_000, ref_seed=100_000,
            reads=1_000_000, L=150, mode="rmi", read_seed=1003, scaling="weak"),
    3: dict(name="1 Mb reference, 10M x 150 bp reads, RMI-SMEM", ref="1Mb", n=1_000_000, ref_seed=1_000_000, reads=10_000_000,
            L=150, mode="rmi", read_seed=1004, scaling="weak"),
    4: dict(name="1 Mb reference, 80M x 150 bp reads query-sharded over the GPUs, RMI-SMEM", ref="1Mb", n=1_000_000,
            ref_seed=1_000_000, reads=80_000_000, L=150, mode="rmi", read_seed=1005, scaling="strong"),
    # not a BASELINE config: the size of the reference's own data/full_data.fa (SURVEY: the reference cannot index it)
    5: dict(name="[off-BASELINE] 2.5 Mb reference (size of the reference's full_data.fa), 4M x 150 bp reads, RMI-SMEM", ref="2.5Mb",
            n=2_499_750, ref_seed=2_499_750, reads=4_000_000, L=150, mode="rmi", read_seed=1006, scaling="weak"),
}
K = 15
EXPERTS = [1000]
GEN_CHUNK = 10_000_000                           # reads are generated (on the device) in pieces


def shard_seed(cfg, world, rank):
    """Seed of shard `rank` of `world` of a strong-scaling batch (BASELINE configs[4])."""
    return cfg["read_seed"] + 1000 * world + rank


def gen_reads(ref_dev, n, L, seed, piece, read_kind="fromref"):
    """Piece `piece` (GEN_CHUNK reads at most) of a rank's batch, generated on the device that holds `ref_dev`
    (uint8 tensor of the reference's codes): seconds instead of minutes of host time for 10^7 .. 10^8 reads."""
    if read_kind == "random":
        gen = torch.Generator(device=ref_dev.device)
        gen.manual_seed(seed + 7919 * piece)
        return torch.randint(0, 4, (n, L), generator=gen, device=ref_dev.device, dtype=torch.uint8)
    return synth.reads_from_ref_device(ref_dev, n, L, seed + 7919 * piece)


def build_index(cfg, device):
    """Native index build + RMI training (genie_index_create / genie_index_train_rmi); the device gets the image
    WITHOUT the K-mer hash table (GENIE_IMAGE_NO_SEED_TABLE): nothing in find_smems reads it, and it is what a
    multi-GPU run broadcasts."""
    ref = synth.synth_ref(cfg["n"], cfg["ref_seed"])
    ix = g.GenieIndex.build(ref, K)
    coefs, icpts, _, _, _ = ix.train_rmi(EXPERTS)
    full_bytes = int(g._native.lib().genie_index_blob_bytes(ix._h))
    ix.to(device, seed_table=False)
    return ref, ix, {"coefs": coefs, "icpts": icpts, "full_image_bytes": full_bytes}


def cpu_baseline(ref, cfg, rl, mode, sample_reads):
    """The CPU oracle (a C restatement of the reference's algorithm, kind 'port') timed on this host's cores on a
    bounded sample of the SAME workload: once with OpenMP over reads on all cores, once on ONE thread.
    Reported, not a target."""
    from oracle import oracle as orc
    orc.build()
    nproc = os.cpu_count() or 1
    threads = nproc                                  # all host cores (and one thread, below)
    o = orc.Oracle(ref, K)
    o.set_rmi(EXPERTS, rl["coefs"], rl["icpts"])
    rd = synth.reads_from_ref_fast(ref, sample_reads, cfg["L"], cfg["read_seed"])
    o.find_smems_batch(mode, rd[:2000], nthreads=threads)               # warm-up
    t0 = time.perf_counter()
    counts, rows = o.find_smems_batch(mode, rd, nthreads=threads)
    dt = time.perf_counter() - t0
    assert (counts >= 0).all()
    n1 = max(2000, sample_reads // 4)
    t1 = time.perf_counter()
    o.find_smems_batch(mode, rd[:n1], nthreads=1)
    dt1 = time.perf_counter() - t1
    out = {"value": sample_reads * cfg["L"] / dt, "unit": "query-bases/s", "cores": threads, "kind": "port", "nproc": nproc,
           "single_thread": {"value": n1 * cfg["L"] / dt1, "cores": 1, "sample": f"{n1} reads, {dt1:.2f} s wall"},
           "sample": f"{sample_reads} x {cfg['L']} bp reads of the same from-ref distribution, mode {mode}, "
                     f"OpenMP over reads, {dt:.2f} s wall"}
    if nproc > 64:                                # hyper-threads and remote sockets cost this port more than they bring
        t2 = time.perf_counter()
        o.find_smems_batch(mode, rd, nthreads=64)
        dt2 = time.perf_counter() - t2
        out["threads_64"] = {"value": sample_reads * cfg["L"] / dt2, "cores": 64, "sample": f"the same sample, {dt2:.2f} s wall"}
    return out, rd, counts, rows


def bind_to_gpu_numa(device):
    """Best effort: run this process (and so place its pinned buffers) on the NUMA node the GPU hangs off; the from-host
    rate is a host-link measurement and varies several-fold with where the pinned pages live.  Returns the node or None."""
    try:
        bdf = None
        props = torch.cuda.get_device_properties(device)
        if hasattr(props, "pci_bus_id") and hasattr(props, "pci_device_id"):
            bdf = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        if bdf is None or not os.path.exists(f"/sys/bus/pci/devices/{bdf}/numa_node"):
            return None
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if allowed:
            os.sched_setaffinity(0, allowed)
            return node
    except Exception:  # noqa: BLE001 - measurement hygiene only
        pass
    return None


def from_host_rate(lib, ix, mode_id, host_reads, L, rows_per_read, steps=9, packed=6):
    """Section 8(d)'s metric as SURVEY words it: pinned host reads -> H2D -> the call -> D2H of the results, a
    three-stage pipeline (copy-in, kernels, copy-out streams) over three buffers.  The host link sets this rate; it is never `value`.
    packed = 6 / 8: genie_find_smems_packed6 / genie_find_smems_packed -- 2-bit packed reads in (40 B per 150-base read), a count
    and a status byte per read and 6- / 8-byte rows out; 0: genie_find_smems_csr -- one byte per base in, int64 offsets and
    16-byte rows out."""
    from genie_smem_amd import packing
    N = host_reads.shape[0]
    cap = int(N * rows_per_read * 1.05) + 1024
    cap_esc = 4096                             # (the 6-byte rows escape at spans of 255: rare on a random reference, not absent;
                                               #  the list travels back with the rows)
    P = lambda t: C.c_void_p(t.data_ptr())                                       # noqa: E731
    if packed:
        host_in = torch.as_tensor(packing.pack_reads(host_reads.numpy())).pin_memory()      # host-side layout, outside the timing
    else:
        host_in = host_reads

    class Buf:
        def __init__(self):
            self.ev_in, self.ev_k, self.ev_out = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            self.reads = torch.empty(tuple(host_in.shape), dtype=torch.uint8, device="cuda")
            self.ws_b = int(lib.genie_find_smems_workspace_bytes(N, L))
            self.ws = torch.empty(self.ws_b, dtype=torch.uint8, device="cuda")
            if packed:
                # the five outputs are pieces of ONE device buffer (the ABI takes plain pointers), so that they leave in one copy
                r16 = lambda x: (x + 15) & ~15                                               # noqa: E731
                o_tot = 0; o_esc = 16; o_cnt = o_esc + 16 * cap_esc; o_st = o_cnt + r16(N); o_rows = o_st + r16(N)
                self.out = torch.zeros(o_rows + r16(cap * packed), dtype=torch.uint8, device="cuda")
                self.h_out = torch.empty(self.out.shape[0], dtype=torch.uint8).pin_memory()
                pieces = lambda t: (t[o_tot:o_tot + 16].view(torch.int64), t[o_esc:o_esc + 16 * cap_esc].view(torch.int64).view(cap_esc, 2),   # noqa: E731
                                    t[o_cnt:o_cnt + N], t[o_st:o_st + N], t[o_rows:o_rows + cap * packed].view(cap, packed))
                self.totals, self.esc, self.counts8, self.status8, self.rows = pieces(self.out)
                self.h_totals, self.h_esc, self.h_counts, self.h_status, self.h_rows = pieces(self.h_out)
            else:
                self.status = torch.empty(N, dtype=torch.int32, device="cuda")
                self.offsets = torch.empty(N + 1, dtype=torch.int64, device="cuda")
                self.rows = torch.empty((cap, 4), dtype=torch.int32, device="cuda")
                self.h_off = torch.empty(N + 1, dtype=torch.int64).pin_memory()
                self.h_rows = torch.empty((cap, 4), dtype=torch.int32).pin_memory()

        def run(self):
            # three stages on three streams, chained by events: the next buffer's H2D and the previous one's D2H run
            # beside this one's kernels
            s_in.wait_event(self.ev_out)                               # the buffer's previous results have left
            with torch.cuda.stream(s_in):
                self.reads.copy_(host_in, non_blocking=True)
                self.ev_in.record(s_in)
            s_k.wait_event(self.ev_in)
            with torch.cuda.stream(s_k):
                sp = C.c_void_p(s_k.cuda_stream)
                if packed:
                    entry = lib.genie_find_smems_packed6 if packed == 6 else lib.genie_find_smems_packed
                    g._native.check(entry(ix._h, mode_id, P(self.reads), None, N, host_in.shape[1], L, 1,
                                          P(self.counts8), P(self.status8), P(self.rows), cap, P(self.totals),
                                          P(self.esc), cap_esc, P(self.ws), self.ws_b, sp), "genie_find_smems_packed")
                else:
                    g._native.check(lib.genie_find_smems_csr(ix._h, mode_id, P(self.reads), None, N, L, L, 1, P(self.offsets),
                                                             P(self.rows), cap, P(self.status), P(self.ws), self.ws_b, sp),
                                    "genie_find_smems_csr")
                self.ev_k.record(s_k)
            s_out.wait_event(self.ev_k)
            with torch.cuda.stream(s_out):
                if packed:
                    self.h_out.copy_(self.out, non_blocking=True)
                else:
                    self.h_off.copy_(self.offsets, non_blocking=True)
                    self.h_rows.copy_(self.rows, non_blocking=True)
                self.ev_out.record(s_out)

    nbuf = int(os.environ.get("GENIE_BENCH_NBUF", "3"))
    bufs = [Buf() for _ in range(nbuf)]
    # Four rounds, each on its own three streams, the fastest one reported (all recorded): the first stream set a process
    # creates ran the same pipeline at half the rate of every later one (tools/experiments/pipeline_probe.py: 3.93 ms per
    # step, then 2.16 ms, fresh buffers or not) -- how the runtime maps streams onto hardware queues, not the path.
    rounds = []
    for rnd in range(4):
        s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        for b in bufs:
            b.run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(steps):
            bufs[i % nbuf].run()
        torch.cuda.synchronize()
        rounds.append((time.perf_counter() - t) / steps)
    dt = min(rounds)
    if packed:
        assert int(bufs[0].h_totals[0]) <= cap and int(bufs[0].h_totals[1]) <= (cap_esc if packed == 6 else 0) and int(bufs[0].h_status.sum()) == 0
        assert int(bufs[0].h_counts.to(torch.int64).sum()) == int(bufs[0].h_totals[0])
        bytes_per_read = host_in.shape[1] + 2 + float(packed) * cap / N + 16.0 * cap_esc / N
    else:
        assert int(bufs[0].h_off[-1]) <= cap
        bytes_per_read = L + 8 + 16.0 * cap / N
    return N * L / dt, dt * 1e3, bytes_per_read, [round(x * 1e3, 3) for x in rounds]


def load_counters(key):
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as fh:
            return json.load(fh).get(key)
    except OSError:
        return None


def run_workload(args, cfg_id, mode, steps, warmup, world, rank, device, total_reads=None, read_kind="fromref", read_len=None):
    """Build (rank 0) and broadcast the index of BASELINE config `cfg_id`, generate this rank's batch, time `steps` calls of
    the hot path after `warmup` untimed ones (barrier + synchronize on both sides, MAX over ranks).  Returns everything
    the report needs."""
    cfg = dict(CONFIGS[cfg_id])
    if read_len:
        cfg["L"] = read_len
    # ---- index: rank 0 builds, ONE broadcast of the image (RCCL over xGMI), no later traffic
    t_build = time.perf_counter()
    if rank == 0:
        ref, ix, rl = build_index(cfg, device)
    else:
        ref, ix, rl = None, None, None
    t_build = time.perf_counter() - t_build
    broadcast_ms = None
    if world > 1:
        torch.cuda.synchronize(device)
        dist.barrier()
        t_bc = time.perf_counter()
        ix = parallel.broadcast_index(ix, src=0, device=device)
        torch.cuda.synchronize(device)
        broadcast_ms = (time.perf_counter() - t_bc) * 1e3
    if args.search_all:
        ix.set_option(g._native.OPT_SEARCH_ALL, 1)

    # ---- this rank's batch.  weak: the config's shape on every rank (seed + rank); strong: this rank's
    # contiguous shard of ONE batch (the shard is generated with the shard's own seed: shard s of W is the
    # same reads whichever rank holds it)
    ref_codes = synth.synth_ref(cfg["n"], cfg["ref_seed"])
    total_reads = total_reads or cfg["reads"]
    if cfg["scaling"] == "strong":
        lo, hi = parallel.shard_bounds(total_reads, rank, world)
        n_reads, seed = hi - lo, shard_seed(cfg, world, rank)
    else:
        n_reads, seed = total_reads, cfg["read_seed"] + rank
    L = cfg["L"]
    reads = torch.empty((n_reads, L), dtype=torch.uint8, device=device)
    ref_dev = torch.as_tensor(ref_codes).to(device)
    for c0 in range(0, n_reads, GEN_CHUNK):
        c1 = min(n_reads, c0 + GEN_CHUNK)
        reads[c0:c1] = gen_reads(ref_dev, c1 - c0, L, seed, c0 // GEN_CHUNK, read_kind)
    del ref_dev
    status = torch.empty(n_reads, dtype=torch.int32, device=device)
    offsets = torch.empty(n_reads + 1, dtype=torch.int64, device=device)
    rows_cap = n_reads * (max(40, L // 3) if read_kind == "random" else max(16, L // 9))    # >= 1.3x the mean count
    out = torch.empty((rows_cap, 4), dtype=torch.int32, device=device)
    lib = g._native.lib()
    ws_bytes = int(lib.genie_find_smems_workspace_bytes(n_reads, L))
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=device)

    P = lambda t: C.c_void_p(t.data_ptr())                                       # noqa: E731
    stream = torch.cuda.current_stream(device)
    sp = C.c_void_p(stream.cuda_stream)
    mode_id = g._native.MODES[mode]

    # Events: (a) around the whole hot-path call, (b) -- through the C ABI's profiling hook -- around its
    # dominant kernel (the match-statistics search), recorded by the library on the launch stream.
    def mk():
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)                       # materialise the hipEvent_t handle
        return e
    ev = [(mk(), mk()) for _ in range(steps)]
    ev_k = [(mk(), mk()) for _ in range(steps)]
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

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
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
    assert total <= out.shape[0], "CSR rows overflowed the output buffer"
    lib.genie_index_set_stage_events(ix._h, None, None)
    tot = torch.tensor([n_reads, total], dtype=torch.float64, device=device)      # units all ranks processed
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    kern_ms = [a.elapsed_time(b) for a, b in ev_k]
    path_ms = [a.elapsed_time(b) for a, b in ev]
    return dict(cfg=cfg, cfg_id=cfg_id, mode=mode, steps=steps, warmup=warmup, ix=ix, rl=rl, ref_codes=ref_codes, reads=reads,
                n_reads=n_reads, L=L, dt=dt, total=total, all_reads=int(tot[0].item()), all_rows=int(tot[1].item()),
                kern_ms=kern_ms, path_ms=path_ms, t_build=t_build, broadcast_ms=broadcast_ms, read_kind=read_kind,
                image_bytes=int(ix.blob.numel()))


def roofline_of(w, offcfg=()):
    """The `roofline` object of one measured workload (module docstring)."""
    cfg, ix, mode, L, n_reads = w["cfg"], w["ix"], w["mode"], w["L"], w["n_reads"]
    kern_ms_avg, path_ms_avg = float(np.mean(w["kern_ms"])), float(np.mean(w["path_ms"]))
    S = w["total"] / n_reads
    ms_step = w["dt"] / w["steps"] * 1e3
    shape = ix.workspace_shape(L)           # bytes per read of the hand-off rows
    # COMPULSORY HBM bytes per read: what a kernel must read and write (inputs + hand-offs + outputs)
    search_bytes = L + 4 + shape["fwd_stride"] + 16 * shape["qp_recs"]
    step_bytes = L + 16.0 * S + 12            # reads in, (start, end, lo, hi) rows + offset + status out
    key = f"config{w['cfg_id']}:{mode}:{n_reads}" + ("" if not offcfg else ":" + ",".join(offcfg))
    ctr = load_counters(key)
    kname = ix.search_kernel_name(mode, L)
    achieved = search_bytes * n_reads / (kern_ms_avg * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_read": search_bytes,
            "kernel_ms_avg": kern_ms_avg, "kernel_ms_min": float(np.min(w["kern_ms"])),
            "definition": "achieved = compulsory HBM bytes of this kernel (reads in; fwd rows, the packed read as plain words, "
                          "status out) / its time; the kernel is bound by random L1->L2 requests and VALU issue, "
                          "not by HBM (see `binding`)",
            "step": {"kernels": "match statistics + traversal + scan of the block sums + interval search -> offsets and CSR rows",
                     "ms_avg": path_ms_avg, "compulsory_bytes_per_read": step_bytes,
                     "achieved": step_bytes * n_reads / (path_ms_avg * 1e-3) / 1e9,
                     "frac": step_bytes * n_reads / (path_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "share_of_step": path_ms_avg / ms_step},
            "survey_8d": {"alg_bytes_per_read": L + 16.0 * S + L * math.ceil(math.log2(cfg["n"] + 1)) * 12,
                          "equiv_GBps": (L + 16.0 * S + L * math.ceil(math.log2(cfg["n"] + 1)) * 12) * n_reads / (path_ms_avg * 1e-3) / 1e9,
                          "note": "SURVEY 8(d) prices a ceil(log2(n+1))-probe suffix-array search per position; the kernel "
                                  "reads one 16-byte table entry per looked-up position instead, so this is an equivalent rate, not "
                                  "a fraction of a roof"}}
    if ctr:
        ks = ctr["kernels"]
        dom = ks.get(kname) or {}
        roof["traffic"] = dom.get("hbm_bytes")
        hbm_step = sum(k.get("hbm_bytes", 0.0) for k in ks.values())
        roof["traffic_raw"] = dom.get("hbm_bytes_raw")
        roof["traffic_note"] = ("traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, the guide's gfx950 correction; this kernel asks for its "
                                "input through the scalar cache (64-byte s_load fetches, which FETCH_SIZE counts in full, where it counted "
                                "the streamed vector reads at half), so the correction overstates it: traffic_raw = FETCH_SIZE + WRITE_SIZE "
                                "is the closer figure")
        roof["counters_tag"] = ctr.get("tag")
        roof["step"].update({"hbm_measured_bytes": hbm_step, "hbm_measured_bytes_raw": sum(k.get("hbm_bytes_raw", 0.0) for k in ks.values()),
                             "hbm_note": "FETCH_SIZE x 2 + WRITE_SIZE (MI355X_MICROARCH HBM correction; an upper bound for narrow accesses); _raw = "
                                         "FETCH_SIZE + WRITE_SIZE; both count table lines that miss L2 and hit the Infinity Cache", "hbm_measured_frac": hbm_step / (path_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic_ratio": hbm_step / (step_bytes * n_reads)})
        if "TCP_TCC_READ_REQ_sum" in dom:
            req = dom["TCP_TCC_READ_REQ_sum"]
            cycles = dom.get("GRBM_GUI_ACTIVE", 0) / ctr.get("xcds", 8)            # kernel duration in shader cycles
            lat = dom.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / req if req else None
            roof["binding"] = {
                "resource": "vector-instruction issue and L1 miss handling: on a table that fits the L2 (100 kb) VALU issue is the first limit "
                            "(valu_issue_share) with ~60 L1->L2 read requests in flight per CU at the L2's ~180 cycles (one 64-byte line per "
                            "table entry); on one that does not (1 Mb) the mean latency -- and with it the time -- follows the L2 MISSES, each "
                            "a trip through the fabric (~120 requests in flight per CU, the L1 reporting pending stalls 4/5 of the time)",
                "l2_read_requests_per_read": req / n_reads,
                "mean_request_latency_cycles": lat,
                "requests_in_flight_per_cu": (dom.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / (cycles * 256)) if cycles else None,
                "l1_miss_queue_full_share": (dom.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / cycles) if cycles else None,
                "l2_misses_per_read": dom["TCC_MISS_sum"] / n_reads if "TCC_MISS_sum" in dom else None,
                "l2_hit_rate": dom["TCC_HIT_sum"] / (dom["TCC_HIT_sum"] + dom["TCC_MISS_sum"]) if "TCC_MISS_sum" in dom else None,
                "fabric_read_requests_per_read": dom["TCC_EA0_RDREQ_sum"] / n_reads if "TCC_EA0_RDREQ_sum" in dom else None,
                "l2_read_GBps": req * 64 / (kern_ms_avg * 1e-3) / 1e9, "l2_peak_GBps": L2_PEAK_GBS,
                "l2_frac": req * 64 / (kern_ms_avg * 1e-3) / 1e9 / L2_PEAK_GBS,
                "valu_insts_per_read": dom.get("SQ_INSTS_VALU", 0) / n_reads,
                "valu_issue_share": (dom.get("SQ_ACTIVE_INST_VALU", 0) * 4 / ctr.get("simds", 1024)) / cycles if cycles else None,
                "wave_wait_share": dom.get("SQ_WAIT_ANY", 0) / dom["SQ_WAVE_CYCLES"] if dom.get("SQ_WAVE_CYCLES") else None,
                "source": "profiles/pmc_counters.json (rocprofv3 --pmc passes of the same workload, tools/profile_round.sh)"}
    return roof, key, S, ms_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--mode", default=None, choices=["bwa", "lut", "rmi"])
    ap.add_argument("--reads", type=int, default=None, help="reads per step: per GPU (configs 1-3) or in all (config 4)")
    ap.add_argument("--read-len", type=int, default=None, help="off-config read length (sweeps only; named in config.workload)")
    ap.add_argument("--read-kind", default="fromref", choices=["fromref", "random"], help="off-config read distribution (sweeps only)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-from-host", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the records of the other BASELINE configs appended to the default line")
    ap.add_argument("--search-all", action="store_true", help="GENIE_OPT_SEARCH_ALL: look up every position, no sampling (A/B runs)")
    args = ap.parse_args()

    cfg0 = CONFIGS[args.config]
    offcfg = []
    if args.read_len and args.read_len != cfg0["L"]:
        offcfg.append(f"read_len={args.read_len}")
    if args.read_kind != "fromref":
        offcfg.append(f"reads={args.read_kind}")
    mode = args.mode or cfg0["mode"]
    if mode != cfg0["mode"]:
        offcfg.append(f"mode={mode}")
    if args.search_all:
        offcfg.append("search-all")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # one rank per GPU; (a rehearsal of the N > 1 path on a box with fewer GPUs than ranks shares them)
    device = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    backend = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL over xGMI.  GENIE_BENCH_BACKEND=gloo is a REHEARSAL switch (two ranks on one GPU cannot form an RCCL
        # communicator); the driver's runs use the default.
        backend = os.environ.get("GENIE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    w = run_workload(args, args.config, mode, args.steps, args.warmup, world, rank, device, total_reads=args.reads,
                     read_kind=args.read_kind, read_len=args.read_len)
    cfg, ix, L, n_reads = w["cfg"], w["ix"], w["L"], w["n_reads"]

    if rank == 0:
        roof, key, S, ms_step = roofline_of(w, offcfg)
        par = {"parallelism": f"query-sharded x{world}, index replicated (one RCCL broadcast)"}
        if world > 1:
            # what the process group itself reports, so that the line proves N ranks took part
            par.update({"backend": dist.get_backend(), "backend_requested": backend, "world_size": dist.get_world_size(),
                        "broadcast_ms": round(w["broadcast_ms"], 3), "broadcast_bytes": w["image_bytes"],
                        "devices_visible": torch.cuda.device_count()})
        line = {
            "metric": f"query-bases/sec SMEM discovery, {cfg['ref']} ref x {L}bp reads; bit-exact SMEM set",
            "value": w["all_reads"] * L * args.steps / w["dt"],
            "unit": "query-bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": cfg["scaling"],
            "vs_baseline": None,
            "dtype": "u8/int32 (2-bit packed bases, int32 SA rows; f64 only in the RMI predict)",
            "data": "synthetic",
            "config": {"workload": cfg["name"] + ("" if not offcfg else " [off-config: " + ", ".join(offcfg) + "]"),
                       "reference_bases": cfg["n"], "reads_per_step_all_gpus": w["all_reads"], "reads_per_gpu_per_step": n_reads,
                       "read_len": L, "K": K, "mode": mode, "rmi_experts": EXPERTS,
                       "read_distribution": "from-ref segments U{1..30}" if args.read_kind == "fromref" else "uniform random",
                       **par,
                       "smems_per_read": round(w["all_rows"] / w["all_reads"], 3), "launch": ix.launch_info(mode, L), "counters_key": key,
                       "index_build_s": round(w["t_build"], 3), "index_image_bytes": w["image_bytes"],
                       "index_image_note": "image without the K-mer hash table (find_smems does not read it)",
                       "index_full_image_bytes": (w["rl"] or {}).get("full_image_bytes")},
            "roofline": roof,
        }
        lib = g._native.lib()
        if world == 1 and not args.no_from_host and n_reads <= 2_000_000 and L <= 255:
            saved_affinity = os.sched_getaffinity(0)
            node = bind_to_gpu_numa(device)
            host_reads = w["reads"].cpu().pin_memory()
            v, ms, bpr, rr = from_host_rate(lib, ix, g._native.MODES[mode], host_reads, L, S, packed=6)
            v8, ms8, bpr8, rr8 = from_host_rate(lib, ix, g._native.MODES[mode], host_reads, L, S, packed=8)
            v0, ms0, bpr0, rr0 = from_host_rate(lib, ix, g._native.MODES[mode], host_reads, L, S, packed=0)
            line["value_from_host"] = {"value": v, "unit": "query-bases/s", "ms_per_step": ms, "ms_per_step_rounds": rr, "link_bytes_per_read": round(bpr, 1),
                                       "what": "pinned host reads, 2-bit packed -> H2D -> genie_find_smems_packed6 -> D2H of count / status "
                                               "bytes + 6-byte rows, copy-in / kernels / copy-out pipelined on three streams (SURVEY 8d's wording of the metric; "
                                               "host-link bound; never `value`)",
                                       "rows8": {"value": v8, "ms_per_step": ms8, "ms_per_step_rounds": rr8, "link_bytes_per_read": round(bpr8, 1),
                                                 "what": "the same through genie_find_smems_packed (8-byte rows: any reference size)"},
                                       "unpacked": {"value": v0, "ms_per_step": ms0, "ms_per_step_rounds": rr0, "link_bytes_per_read": round(bpr0, 1),
                                                    "what": "the same through genie_find_smems_csr: a byte per base in, int64 offsets + 16-byte rows out"}}
            line["value_from_host"]["numa_node_bound"] = node
            os.sched_setaffinity(0, saved_affinity)
            del host_reads
            # the same entry point on RESIDENT packed reads (device time only; no pack stage, a quarter of the input bytes)
            from genie_smem_amd import packing
            pk = torch.as_tensor(packing.pack_reads(w["reads"].cpu().numpy())).to(device)
            cap = int(n_reads * S * 1.05) + 1024
            bufs = [torch.empty(n_reads, dtype=torch.uint8, device=device), torch.empty(n_reads, dtype=torch.uint8, device=device),
                    torch.empty((cap, 8), dtype=torch.uint8, device=device), torch.zeros(2, dtype=torch.int64, device=device),
                    torch.empty((1024, 2), dtype=torch.int64, device=device)]
            ws_b = int(lib.genie_find_smems_workspace_bytes(n_reads, L))
            ws2 = torch.empty(ws_b, dtype=torch.uint8, device=device)
            PP = lambda t: C.c_void_p(t.data_ptr())                                      # noqa: E731
            sp2 = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

            def packed_step():
                g._native.check(lib.genie_find_smems_packed(ix._h, g._native.MODES[mode], PP(pk), None, n_reads, pk.shape[1], L, 1, PP(bufs[0]),
                                                            PP(bufs[1]), PP(bufs[2]), cap, PP(bufs[3]), PP(bufs[4]), 1024, PP(ws2), ws_b, sp2),
                                "genie_find_smems_packed")
            for _ in range(args.warmup):
                packed_step()
            torch.cuda.synchronize(device)
            t_p = time.perf_counter()
            for _ in range(args.steps):
                packed_step()
            torch.cuda.synchronize(device)
            dt_p = (time.perf_counter() - t_p) / args.steps
            assert int(bufs[3][0].item()) == w["total"] and int(bufs[1].sum().item()) == 0
            line["packed_resident"] = {"value": n_reads * L / dt_p, "unit": "query-bases/s", "ms_per_step": dt_p * 1e3,
                                       "what": "genie_find_smems_packed on 2-bit packed reads already in HBM (not `value`: the headline keeps one "
                                               "code per base resident, as in rounds 1 and 2)"}
            del pk, bufs, ws2
        if not args.no_cpu_baseline and world == 1:
            base, rd_s, cnt_s, rows_s = cpu_baseline(w["ref_codes"], cfg, w["rl"], mode, args.cpu_sample)
            line["cpu_baseline"] = base
            # parity check on the CPU sample: the same reads through the GPU path, rows compared one by one
            nchk = min(20000, len(rd_s))
            o2, s2, st2 = ix.find_smems(mode, rd_s[:nchk])
            o2, s2 = o2.cpu().numpy(), s2.cpu().numpy()
            assert (np.diff(o2) == cnt_s[:nchk]).all(), "GPU/oracle SMEM counts differ"
            for r in range(nchk):
                assert (s2[o2[r]:o2[r + 1]] == rows_s[r, :cnt_s[r]]).all(), f"GPU/oracle rows differ at sample read {r}"
            line["cpu_baseline"]["parity_check"] = f"{nchk} sample reads: GPU (start, end, lo, hi) rows == oracle rows"
    # ---- the default (driver-run) line also carries the other BASELINE configs, measured in the same process with fewer
    # steps (config 1 stays the headline `value`): on one GPU configs[2], [3] and [4] (the 8 x 10^7-read batch whole); on
    # N > 1 GPUs configs[3] (weak: 10^7 reads per rank) and configs[4] (strong: the one batch cut into N contiguous shards)
    if args.config == 1 and not offcfg and not args.reads and not args.no_other_configs:
        del w, ix
        torch.cuda.empty_cache()
        others = []
        for cid, st in (((2, 5), (3, 3), (4, 2)) if world == 1 else ((3, 3), (4, 2))):
            w2 = run_workload(args, cid, CONFIGS[cid]["mode"], st, 1, world, rank, device)
            if rank == 0:
                r2, key2, S2, ms2 = roofline_of(w2)
                rec = {"config": CONFIGS[cid]["name"], "baseline_configs_index": cid, "mode": w2["mode"],
                       "value": w2["all_reads"] * w2["L"] * st / w2["dt"], "unit": "query-bases/s", "n_gpus": world,
                       "scaling": w2["cfg"]["scaling"], "steps": st, "warmup": 1,
                       "ms_per_step": ms2, "reads_per_step_all_gpus": w2["all_reads"], "reads_per_gpu_per_step": w2["n_reads"],
                       "smems_per_read": round(w2["all_rows"] / w2["all_reads"], 3),
                       "kernel": r2["kernel"], "kernel_ms_avg": r2["kernel_ms_avg"],
                       "roofline": {k: r2.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_raw")},
                       "counters_key": key2}
                if world > 1:
                    rec.update({"broadcast_ms": round(w2["broadcast_ms"], 3), "broadcast_bytes": w2["image_bytes"]})
                others.append(rec)
            del w2
            torch.cuda.empty_cache()
        if rank == 0:
            line["other_configs"] = others
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

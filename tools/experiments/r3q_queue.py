"""GENIE_OPT_SCHEDULING A/B (bit 1: fixed share of read groups per wave in K_A; 2: no priority rotation in K_A; 4: none in K_C):
K_A timed with the ABI's stage events, the whole call with events around it.  usage (GPU box): python tools/experiments/r3q_queue.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import genie_smem_amd as g
from genie_smem_amd import synth
lib = g._native.lib()
P = lambda t: C.c_void_p(t.data_ptr())
VARS = [int(x) for x in os.environ.get("VARS", "0 4 2 6 3 7").split()]
CASES = ((100_000, 1_000_000, "fromref", 150), (1_000_000, 4_000_000, "fromref", 150), (100_000, 1_000_000, "random", 150),
         (100_000, 300_000, "fromref", 500), (100_000, 75_000, "fromref", 2000), (100_000, 18_750, "fromref", 8000))
if os.environ.get("CASE"):
    c = os.environ["CASE"].split(","); CASES = ((int(c[0]), int(c[1]), c[2], int(c[3])),)
for n, N, kind, L in CASES[int(os.environ.get("FIRST_CASE", "0")):]:
    ref = synth.synth_ref(n, n)
    ix = g.GenieIndex.build(ref, 15); ix.train_rmi([1000]); ix = ix.to("cuda")
    if os.environ.get("BPC"): ix.set_option(g._native.OPT_SEARCH_BLOCKS_PER_CU, int(os.environ["BPC"]))
    if kind == "random":
        reads = torch.as_tensor(np.random.default_rng(7).integers(0, 4, (N, L)).astype(np.uint8)).cuda()
    else:
        reads = synth.reads_from_ref_device(torch.as_tensor(ref).cuda(), N, L, 1002)
    status = torch.empty(N, dtype=torch.int32, device="cuda"); offsets = torch.empty(N + 1, dtype=torch.int64, device="cuda")
    out = torch.empty((N * 40, 4), dtype=torch.int32, device="cuda")
    wsb = int(lib.genie_find_smems_workspace_bytes(N, L)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev: e.record(s)
    torch.cuda.synchronize()
    ka = {v: [] for v in VARS}; call = {v: [] for v in VARS}
    ref_rows = None
    for rep in range(6):
        for v in VARS:
            ix.set_option(g._native.OPT_SCHEDULING, v)
            lib.genie_index_set_stage_events(ix._h, C.c_void_p(ev[0].cuda_event), C.c_void_p(ev[1].cuda_event))
            for i in range(3):
                if i == 2: ev[2].record(s)
                rc = lib.genie_find_smems_csr(ix._h, 1, P(reads), None, N, L, L, 1, P(offsets), P(out), out.shape[0], P(status), P(ws), wsb, sp)
                assert rc == 0
            ev[3].record(s)
            torch.cuda.synchronize()
            ka[v].append(ev[0].elapsed_time(ev[1])); call[v].append(ev[2].elapsed_time(ev[3]))
            if rep == 0:                                         # the same rows whatever the schedule
                tot = int(offsets[-1].item()); chk = (int(out[:tot].to(torch.int64).sum().item()), tot)
                assert ref_rows in (None, chk), (v, chk, ref_rows); ref_rows = chk
    print(n, N, kind, L, "  ".join("[%d] K_A %.4f call %.4f" % (v, np.median(ka[v]), np.median(call[v])) for v in VARS), flush=True)

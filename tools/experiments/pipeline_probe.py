"""Why does the from-host pipeline of the packed entry point not overlap?  Variants of bench.from_host_rate."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import genie_smem_amd as g
from genie_smem_amd import synth, packing
n, N, L = 100_000, 1_000_000, 150
ref = synth.synth_ref(n, n)
ix = g.GenieIndex.build(ref, 15); ix.train_rmi([1000]); ix = ix.to("cuda", seed_table=False)
rd_h = synth.reads_from_ref_device(ref, N, L, 1002, device="cuda").cpu().pin_memory()
pk_h = torch.as_tensor(packing.pack_reads(rd_h.numpy())).pin_memory()
lib = g._native.lib(); P = lambda t: C.c_void_p(t.data_ptr())
cap = int(N * 12.6)
ws_b = int(lib.genie_find_smems_workspace_bytes(N, L))

def run(variant, nbuf=3, steps=9):
    s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    packed = variant != "csr"
    host_in = pk_h if packed else rd_h
    class Buf:
        def __init__(self):
            self.ev_in, self.ev_k, self.ev_out = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            self.reads = torch.empty(tuple(host_in.shape), dtype=torch.uint8, device="cuda")
            self.ws = torch.empty(ws_b, dtype=torch.uint8, device="cuda")
            self.c8 = torch.empty(N, dtype=torch.uint8, device="cuda"); self.s8 = torch.empty(N, dtype=torch.uint8, device="cuda")
            self.tot = torch.zeros(2, dtype=torch.int64, device="cuda"); self.esc = torch.empty((1024, 2), dtype=torch.int64, device="cuda")
            self.st = torch.empty(N, dtype=torch.int32, device="cuda"); self.off = torch.empty(N + 1, dtype=torch.int64, device="cuda")
            self.rows = torch.empty(cap * (8 if packed else 16), dtype=torch.uint8, device="cuda")
            self.h_rows = torch.empty(cap * (8 if packed else 16), dtype=torch.uint8).pin_memory()
            self.h_c8 = torch.empty(N, dtype=torch.uint8).pin_memory(); self.h_s8 = torch.empty(N, dtype=torch.uint8).pin_memory()
            self.h_tot = torch.empty(2, dtype=torch.int64).pin_memory(); self.h_off = torch.empty(N + 1, dtype=torch.int64).pin_memory()
        def run(self):
            s_in.wait_event(self.ev_out)
            with torch.cuda.stream(s_in):
                self.reads.copy_(host_in, non_blocking=True); self.ev_in.record(s_in)
            s_k.wait_event(self.ev_in)
            with torch.cuda.stream(s_k):
                sp = C.c_void_p(s_k.cuda_stream)
                if packed:
                    g._native.check(lib.genie_find_smems_packed(ix._h, 1, P(self.reads), None, N, host_in.shape[1], L, 1, P(self.c8), P(self.s8), P(self.rows), cap, P(self.tot), P(self.esc), 1024, P(self.ws), ws_b, sp), "p")
                else:
                    g._native.check(lib.genie_find_smems_csr(ix._h, 1, P(self.reads), None, N, L, L, 1, P(self.off), P(self.rows), cap, P(self.st), P(self.ws), ws_b, sp), "c")
                self.ev_k.record(s_k)
            s_out.wait_event(self.ev_k)
            with torch.cuda.stream(s_out):
                if packed and variant != "packed_rows_only":
                    self.h_c8.copy_(self.c8, non_blocking=True); self.h_s8.copy_(self.s8, non_blocking=True); self.h_tot.copy_(self.tot, non_blocking=True)
                if not packed:
                    self.h_off.copy_(self.off, non_blocking=True)
                if variant == "packed_rows_last_small":
                    pass
                self.h_rows.copy_(self.rows, non_blocking=True)
                self.ev_out.record(s_out)
    bufs = [Buf() for _ in range(nbuf)]
    for b in bufs: b.run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps): bufs[i % nbuf].run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3

import subprocess
print(subprocess.run("cat /sys/class/drm/card*/device/numa_node; nproc; cat /sys/devices/system/node/online", shell=True, capture_output=True, text=True).stdout)
print("affinity", len(os.sched_getaffinity(0)))
for v in ("packed", "packed", "packed", "csr", "packed", "packed"):
    print(v, "%.3f ms" % run(v), flush=True)
# fresh pinned blocks again (bigger cap so the cached blocks do not fit)
cap = int(N * 14.1)
for v in ("packed", "packed"):
    print("bigger cap", v, "%.3f ms" % run(v), flush=True)

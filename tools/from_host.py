#!/usr/bin/env python3
"""PCIe-inclusive rate of the LUT workload: pinned host reads -> H2D -> one genie_find_smems_csr call -> D2H of
offsets + rows, (a) serial on one stream, (b) double-buffered on two streams.  Never the bench `value`."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genie_smem_amd as g
from genie_smem_amd import synth

def main():
    n, L, N, K = 100_000, 150, 1_000_000, 15
    ref = synth.synth_ref(n, n)
    ix = g.GenieIndex.build(ref, K).to("cuda")
    lib = g._native.lib()
    host_reads = torch.as_tensor(synth.reads_from_ref_fast(ref, N, L, 1002)).pin_memory()
    P = lambda t: C.c_void_p(t.data_ptr())
    cap = N * 16
    class Buf:
        def __init__(self):
            self.stream = torch.cuda.Stream()
            self.reads = torch.empty((N, L), dtype=torch.uint8, device="cuda")
            self.status = torch.empty(N, dtype=torch.int32, device="cuda")
            self.offsets = torch.empty(N + 1, dtype=torch.int64, device="cuda")
            self.rows = torch.empty((cap, 4), dtype=torch.int32, device="cuda")
            self.ws_b = int(lib.genie_find_smems_workspace_bytes(N, L))
            self.ws = torch.empty(self.ws_b, dtype=torch.uint8, device="cuda")
            self.h_off = torch.empty(N + 1, dtype=torch.int64).pin_memory()
            self.h_rows = torch.empty((cap, 4), dtype=torch.int32).pin_memory()
        def run(self):
            with torch.cuda.stream(self.stream):
                self.reads.copy_(host_reads, non_blocking=True)
                g._native.check(lib.genie_find_smems_csr(ix._h, 1, P(self.reads), None, N, L, L, 1, P(self.offsets), P(self.rows), cap,
                                                         P(self.status), P(self.ws), self.ws_b, C.c_void_p(self.stream.cuda_stream)), "csr")
                self.h_off.copy_(self.offsets, non_blocking=True)
                self.h_rows[:12_100_000].copy_(self.rows[:12_100_000], non_blocking=True)      # ~12 rows per read on this workload
    bufs = [Buf(), Buf()]
    for b in bufs: b.run()
    torch.cuda.synchronize()
    steps = 10
    t = time.perf_counter()
    for _ in range(steps):
        bufs[0].run(); bufs[0].stream.synchronize()
    dt = (time.perf_counter() - t) / steps
    print("serial, one stream:      %.2f ms per 10^6 reads = %.1f G bases/s" % (dt * 1e3, N * L / dt / 1e9))
    t = time.perf_counter()
    for i in range(steps):
        bufs[i & 1].run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print("double-buffered, 2 streams: %.2f ms per 10^6 reads = %.1f G bases/s" % (dt * 1e3, N * L / dt / 1e9))
main()

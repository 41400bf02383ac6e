#!/usr/bin/env python3
"""Experiment: do the three kernels of the pipeline overlap usefully when two half-batches run on two
streams?  Prints ms per full batch for (a) one call, (b) two concurrent half-batch calls."""
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
    if os.environ.get("BPC"):
        ix.set_option(6, int(os.environ["BPC"]))
    reads = torch.as_tensor(synth.reads_from_ref_fast(ref, N, L, 1002)).cuda()
    def bufs(nr):
        ws_b = int(lib.genie_find_smems_workspace_bytes(nr, L))
        return dict(n=nr, status=torch.empty(nr, dtype=torch.int32, device="cuda"), offsets=torch.empty(nr + 1, dtype=torch.int64, device="cuda"),
                    out=torch.empty((nr * 40, 4), dtype=torch.int32, device="cuda"), ws=torch.empty(ws_b, dtype=torch.uint8, device="cuda"), ws_b=ws_b)
    P = lambda t: C.c_void_p(t.data_ptr())
    def call(b, rd, stream):
        g._native.check(lib.genie_find_smems_csr(ix._h, 1, P(rd), None, b["n"], L, L, 1, P(b["offsets"]), P(b["out"]), b["out"].shape[0],
                                                 P(b["status"]), P(b["ws"]), b["ws_b"], C.c_void_p(stream.cuda_stream)), "csr")
    full = bufs(N)
    s0 = torch.cuda.current_stream()
    for _ in range(3): call(full, reads, s0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): call(full, reads, s0)
    torch.cuda.synchronize(); print("one call, full batch: %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
    parts = int(os.environ.get("PARTS", "2"))
    h = N // parts
    bs = [bufs(h) for _ in range(parts)]
    rds = [reads[i * h:(i + 1) * h] for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def both():
        for i in range(parts): call(bs[i], rds[i], streams[i])
    for _ in range(3): both()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): both()
    torch.cuda.synchronize(); print("%d concurrent calls of 1/%d batch: %.3f ms per full batch" % (parts, parts, (time.perf_counter() - t) / 20 * 1e3))
main()

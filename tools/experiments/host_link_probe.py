#!/usr/bin/env python3
"""Experiment: what the host link gives -- H2D alone, D2H alone, both at once on two streams -- next to the
bench's `value_from_host` pipeline (150 MB of reads in, ~200 MB of rows out per 10^6 reads)."""
import time
import torch

def rate(fn, n=6):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

MB = 1 << 20
h_in = torch.empty(150 * MB, dtype=torch.uint8).pin_memory()
d_in = torch.empty(150 * MB, dtype=torch.uint8, device="cuda")
d_out = torch.empty(200 * MB, dtype=torch.uint8, device="cuda")
h_out = torch.empty(200 * MB, dtype=torch.uint8).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def h2d():
    with torch.cuda.stream(s1): d_in.copy_(h_in, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h_out.copy_(d_out, non_blocking=True)
def both():
    h2d(); d2h()
t = rate(h2d); print("H2D 150 MB: %.2f ms = %.1f GB/s" % (t * 1e3, 150 * MB / t / 1e9))
t = rate(d2h); print("D2H 200 MB: %.2f ms = %.1f GB/s" % (t * 1e3, 200 * MB / t / 1e9))
t = rate(both); print("both at once: %.2f ms = %.1f GB/s in all" % (t * 1e3, 350 * MB / t / 1e9))

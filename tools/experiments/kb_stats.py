import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import genie_smem_amd as g
from genie_smem_amd import synth
ref = synth.synth_ref(100000, 100000)
m = g.ExactMatch("x.fa", device="cuda:0"); m.set_reference("".join("ACGT"[c] for c in ref))
rl = g.RMI_LUT([1000], 15, "x.fa", matcher=m); rl.train_RMI(); ix = rl._index()
lib = g._native.lib()
lib.genie_debug_kb_stats.argtypes = [C.c_void_p, C.c_int]
N = 200000
for kind in ("fromref", "random"):
    rd = synth.reads_from_ref_fast(ref, N, 150, 1002) if kind == "fromref" else np.random.default_rng(3).integers(0, 4, (N, 150)).astype(np.uint8)
    for mode in ("lut", "bwa"):
        lib.genie_debug_kb_stats(None, 1)
        o, s, st = ix.find_smems(mode, rd)
        torch.cuda.synchronize()
        out = (C.c_ulonglong * 8)()
        lib.genie_debug_kb_stats(out, 0)
        v = [x / N for x in out]
        print(f"{kind} {mode}: per read: outer {v[0]:.1f} hits {v[1]:.1f} back_ext {v[2]:.1f} kmin-steps {v[3]:.1f} scan-iters {v[4]:.1f} scans {v[5]:.1f} smems {int(o[-1])/N:.1f}")

"""Scratch-branch experiment: the traversal kernel with the interval search fused in at every emit (the structure
that returned wrong intervals in round 1) against the three-kernel pipeline, same reads, 1M x 150 bp."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import genie_smem_amd as g
from genie_smem_amd import synth
n, N, L = 100_000, 1_000_000, 150
ref = synth.synth_ref(n, n)
m = g.ExactMatch("x.fa", device="cuda:0"); m.set_reference("".join("ACGT"[c] for c in ref))
rl = g.RMI_LUT([1000], 15, "x.fa", matcher=m); rl.train_RMI(); ix = rl._index()
reads = torch.as_tensor(synth.reads_from_ref_fast(ref, N, L, 1002)).cuda()
cap = 48
for rep in range(3):
    for mode in ("bwa", "lut", "rmi"):
        ix.set_option(5, 0)
        c0, s0, st0 = ix.find_smems_slots(mode, reads, cap=cap)
        ix.set_option(5, int(os.environ.get("FUSED", "2")))
        c1, s1, st1 = ix.find_smems_slots(mode, reads, cap=cap)
        ix.set_option(5, 0)
        torch.cuda.synchronize()
        assert torch.equal(c0, c1), mode
        valid = torch.arange(cap, device="cuda")[None, :] < c0[:, None]
        diff = ((s0 != s1).any(dim=2) & valid)
        nbad = int(diff.sum().item())
        first = torch.nonzero(diff.any(dim=1))[:5, 0].tolist()
        print(f"rep {rep} {mode}: rows differing {nbad} of {int(valid.sum().item())}; first reads {first}", flush=True)
        if nbad:
            r = first[0]; t = int(torch.nonzero(diff[r])[0, 0].item())
            print("   e.g. read", r, "row", t, "pipeline", s0[r, t].tolist(), "fused", s1[r, t].tolist())

import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
import genie_smem_amd as g
import golden_util as G
d, _ = G.load("medium_K6")
ref = d["ref_codes"]
ix = g.GenieIndex.build(ref, 6).to("cuda")
rd = G.reads("medium_K6", "edge60")
print("reads", rd.shape if hasattr(rd, "shape") else len(rd))
r0 = np.asarray(rd[:4])
off, rows, st = ix.find_smems("bwa", torch.as_tensor(r0).cuda(), min_len=1)
print(off.cpu().tolist(), st.cpu().tolist()); print(rows.cpu().numpy()[:12])
items = G.ref_items("medium_K6", "edge60", "bwa")
for r in range(4): print([(len(s), lo, hi) for s, lo, hi in items[r]])
tr = G.ref_trace("medium_K6", "edge60", "bwa")
for r in range(4): print(tr[r].tolist())

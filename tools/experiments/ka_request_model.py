"""CPU model of the match-statistics kernel's memory requests (no GPU): which positions it looks up (round 1 samples,
round 2 interiors), which entries need their second half, which go to the slow list and what the slow path then loads
(chained entries, suffix-array rows, range-table entry + bisection probes).  Reads the index image the product builds.
usage: python tools/experiments/ka_request_model.py [n_ref] [n_reads] [table_bits]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import genie_smem_amd as g                      # noqa: E402
from genie_smem_amd import synth                # noqa: E402
from oracle import oracle as orc                # noqa: E402

n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tbits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = 150
ref = synth.synth_ref(n_ref, n_ref)
ix = g.GenieIndex.build(ref, 15, table_bits=tbits)
img = ix.serialize().numpy()
hdr = img[:512].tobytes()
# BlobHeader (genie_internal.h)
(magic, version, header_bytes, total_bytes, n, K, P, off_sa, off_ref, off_dir, off_lut, off_rmi, ref_recs, dir_entries,
 lut_slots, lut_keys, rmi_models, nlev) = struct.unpack_from("<QIIqqiiqqqqqqqqqqi", hdr, 0)
o = struct.calcsize("<QIIqqiiqqqqqqqqqqi")
o += 4 * (4 + 4 + 5) + 4 * 8
o = (o + 7) & ~7
off_dir2, dir2_entries, P2, flags, off_rmi_err, rmi_err_entries, off_mtab, mtab_entries = struct.unpack_from("<qqiiqqqq", hdr, o)
print("n", n, "P", P, "P2", P2, "mtab entries", mtab_entries, "table MB", mtab_entries * 32 / 2 ** 20)
mt = img[off_mtab:off_mtab + mtab_entries * 32].view(np.uint32).reshape(-1, 8)
meta_all = mt[:1 << (2 * P2), 0]
rows_all = meta_all >> 24
print("rows/entry mean %.2f  slow entries %.3f  more(chain) %.3f" % (rows_all.mean(), ((meta_all >> 16) & 1).mean(), ((meta_all >> 17) & 1).mean()))

orac = orc.Oracle(ref, 15)
reads = synth.reads_from_ref(ref, n_reads, L, 1004)


def true_fwd(rd):
    f = np.zeros(L, np.int32)
    b = 0
    for a in range(L):
        b = max(b, a)
        while b < L and orac.back_prop(rd[a:b + 1])[0] >= 0:
            b += 1
        f[a] = b
    return f


stats = dict(r1=0, r2=0, second_half=0, slow_full=0, slow_chain=0, slow_general=0, chain_entries=0, rows_cmp=0, gen_probes=0, reads=0)


def lookup(rd, a, f):
    """One table lookup at position a (m = L - a bases left); returns nothing, updates stats."""
    m = L - a
    c = 0
    for j in range(P2):
        c = (c << 2) | (int(rd[a + j]) if a + j < L else 0)
    xk = 0
    for j in range(16):
        xk = (xk << 2) | (int(rd[a + P2 + j]) if a + P2 + j < L else 0)
    e = mt[c]
    meta = int(e[0])
    rows = meta >> 24
    k0, k1 = int(e[2]), int(e[3])
    second = rows > 2 and xk > k1
    stats["second_half"] += second
    keys = [int(e[2 + i]) for i in range(6)] if second else [k0, k1] + [k0] * 4
    x = min(xk ^ k for k in keys)
    lcp = 16 if x == 0 else (32 - x.bit_length()) >> 1
    lm = (meta >> 8) & 0x1F
    slow = bool(meta & (1 << 16)) or (lcp + lm == 16 + 0x1F)
    best0 = (meta & 0xFF) + (lcp & lm)
    slow = slow and m > best0
    if not slow:
        return
    if (meta & (3 << 16)) == (1 << 16):                     # general: range-table entry + bisection over its rows
        stats["slow_general"] += 1
        cnt = rows if rows < 255 else 300
        stats["gen_probes"] += 1 + max(1, int(np.ceil(np.log2(cnt + 1))))
        return
    if meta & (1 << 17):
        stats["slow_chain"] += 1
        ce = (rows - 5 + 7) >> 3
        stats["chain_entries"] += ce
        allk = [int(e[2 + i]) for i in range(5)]
        base = int(e[7])
        for t in range(ce):
            allk += [int(v) for v in mt[base + t]]
        allk = allk[:rows]
        xm = min(xk ^ k for k in allk)
        lc = 16 if xm == 0 else (32 - xm.bit_length()) >> 1
        if lc == 16 and m > P2 + 16:
            stats["rows_cmp"] += sum(1 for k in allk if k == xk)
    else:
        stats["slow_full"] += 1
        allk = [int(e[2 + i]) for i in range(6)][:rows]
        stats["rows_cmp"] += sum(1 for k in allk if k == xk)


for rd in reads:
    f = true_fwd(rd)
    stats["reads"] += 1
    for a in range(0, L, 4):
        stats["r1"] += 1
        lookup(rd, a, f)
    for a in range(0, L, 4):
        m = L - a
        v0 = f[a]
        has_right = m > 4
        v4 = f[a + 4] if has_right else 0
        need = m > 1 and v0 != L and not (has_right and v4 == v0)
        if not need:
            continue
        km = 2 if m > 2 else 1
        stats["r2"] += 1
        lookup(rd, a + km, f)
        if km == 1:
            continue
        vm = f[a + 2]
        need1 = vm != v0
        need3 = m > 3 and not ((has_right and vm == v4) or vm == L)
        for need_, k_ in ((need1, 1), (need3, 3)):
            if need_:
                stats["r2"] += 1
                lookup(rd, a + k_, f)
R = stats.pop("reads")
print({k: round(v / R, 2) for k, v in stats.items()})

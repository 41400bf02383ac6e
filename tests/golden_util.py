"""Loaders for the golden fixtures under tests/golden/ (written by make_golden.py)."""
import functools
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ACGT = "ACGT"
DATASETS = ["medium_K6", "syn10k_K8", "syn100k_K15", "big100k_K15"]


def codes_to_str(codes, alphabet=ACGT):
    return "".join(alphabet[int(c)] for c in codes)


def str_to_codes(s, alphabet=ACGT):
    return np.asarray([alphabet.index(c) for c in s], np.uint8)


def have(name):
    return os.path.exists(os.path.join(GOLDEN, name + ".npz"))


@functools.lru_cache(maxsize=None)
def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    with open(os.path.join(GOLDEN, name + ".json")) as fh:
        meta = json.load(fh)
    return {k: z[k] for k in z.files}, meta


@functools.lru_cache(maxsize=None)
def known():
    with open(os.path.join(GOLDEN, "known_answers.json")) as fh:
        return json.load(fh)


def groups(name):
    _, meta = load(name)
    return [(tag, g["algos"]) for tag, g in meta["groups"].items()]


def group_cases():
    out = []
    for ds in DATASETS:
        if have(ds):
            for tag, algos in groups(ds):
                for a in algos:
                    out.append((ds, tag, a))
    return out


def ref_items(ds, tag, algo):
    """Per read: the reference's returned dict as an ordered list [(string codes, lo, hi)...]."""
    d, _ = load(ds)
    p = f"{tag}.{algo}."
    off, soff, s, iv = d[p + "items_off"], d[p + "items_soff"], d[p + "items_str"], d[p + "items_iv"]
    out = []
    for r in range(len(off) - 1):
        items = []
        for t in range(off[r], off[r + 1]):
            items.append((bytes(s[soff[t]:soff[t + 1]]), int(iv[t, 0]), int(iv[t, 1])))
        out.append(items)
    return out


def ref_trace(ds, tag, algo):
    """Per read: the ordered emission trace. bwa: rows (start,end,lo,hi); lut/rmi: (start,end)."""
    d, _ = load(ds)
    p = f"{tag}.{algo}."
    off, tr = d[p + "trace_off"], d[p + "trace"]
    return [tr[off[r]:off[r + 1]] for r in range(len(off) - 1)]


def ref_status(ds, tag, algo):
    d, _ = load(ds)
    return d[f"{tag}.{algo}.status"]


def reads(ds, tag):
    d, _ = load(ds)
    return d[f"{tag}.reads"]


def dict_view(read, rows):
    """What the reference's dict would hold for an emission list rows=(start,end,lo,hi):
    key = substring, value = interval, insertion-ordered, duplicates collapse (SMEM.py:185)."""
    out = {}
    for s, e, lo, hi in rows:
        out[bytes(read[s:e])] = (int(lo), int(hi))
    return [(k, v[0], v[1]) for k, v in out.items()]


def sa_sha256(sa1):
    return hashlib.sha256(np.asarray(sa1).astype("<i4").tobytes()).hexdigest()


def lut_sha256(codes, lo, hi, sa1):
    """Same stream as make_golden.lut_summary: per key ascending [code, lo, hi, positions...]."""
    h = hashlib.sha256()
    npos = 0
    maxocc = 0
    for c, a, b in zip(codes.tolist(), lo.tolist(), hi.tolist()):
        pos = sa1[a:b + 1].tolist()
        h.update(np.asarray([c, a, b] + pos, "<i8").tobytes())
        npos += len(pos)
        maxocc = max(maxocc, len(pos))
    return h.hexdigest(), npos, maxocc

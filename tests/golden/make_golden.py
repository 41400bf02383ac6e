#!/usr/bin/env python3
"""Golden-vector generator: runs the UNMODIFIED reference (GENIE-SMEM) in this container.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the build container);
the GPU box never sees the reference, it only sees the small fixtures this script wrote
into tests/golden/.  Nothing from the reference's source is copied: the reference modules
are imported from where they lie (sys.path) and only *data* (inputs + the outputs the
reference computed) is written out.

How the reference is made importable:
  * it uses bare script-style imports and cwd-relative "data/" paths, so we chdir into a
    scratch directory whose path contains "SMEM" (RMI_LUT.py:115-117 splits cwd on "SMEM");
  * `ExactMatch`, `LUT`, `RMI` import as-is.  `RMI_LUT` (and `SMEM`, which imports it at
    SMEM.py:3) do `from Bio import SeqIO` (RMI_LUT.py:2) and biopython is not installed.
    Bio.SeqIO is used for exactly one thing -- reading the single FASTA record's `.seq`
    (RMI_LUT.py:24-25) -- i.e. file I/O, no arithmetic of the hot path.  A 12-line
    in-memory module supplying `SeqIO.parse(path, "fasta")` is registered in sys.modules
    so that the reference's own algorithm code runs unmodified.  Fixtures are tagged with
    `needs_bio_shim` so a reader can see which ones depend on it (the FM / LUT / RMI-predict
    ones do not).

Usage:  python3 -B tests/golden/make_golden.py [--only NAME] [--scratch /tmp/oracle]
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import time
import types

import numpy as np

REF_ROOT = "/root/reference"
REF_SMEM = os.path.join(REF_ROOT, "SMEM")
HERE = os.path.dirname(os.path.abspath(__file__))
ACGT = "ACGT"


# ----------------------------------------------------------------------------- set-up
def install_bio_shim():
    """In-memory stand-in for `Bio.SeqIO.parse(path, 'fasta')` (FASTA reading only)."""
    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")

    class _Rec:  # noqa: D401 - tiny record with a str `.seq`
        def __init__(self, seq):
            self.seq = seq

    def parse(path, fmt):
        assert fmt == "fasta"
        seq, out = [], []
        with open(path) as fh:
            for line in fh:
                if line.startswith(">"):
                    if seq:
                        out.append(_Rec("".join(seq)))
                        seq = []
                else:
                    seq.append(line.strip())
        if seq:
            out.append(_Rec("".join(seq)))
        return iter(out)

    seqio.parse = parse
    bio.SeqIO = seqio
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqIO"] = seqio


def setup_scratch(scratch):
    work = os.path.join(scratch, "SMEM")
    data = os.path.join(work, "data")
    os.makedirs(data, exist_ok=True)
    for f in os.listdir(os.path.join(REF_SMEM, "data")):
        dst = os.path.join(data, f)
        if not os.path.exists(dst):
            shutil.copy(os.path.join(REF_SMEM, "data", f), dst)
            os.chmod(dst, 0o644)
    os.chdir(work)
    sys.path.insert(0, REF_SMEM)
    sys.setrecursionlimit(2500)
    sys.dont_write_bytecode = True
    return work, data


def write_fasta(data_dir, name, seq, header="synthetic"):
    with open(os.path.join(data_dir, name), "w") as fh:
        fh.write(">" + header + "\n")
        for i in range(0, len(seq), 50):
            fh.write(seq[i:i + 50] + "\n")


def read_fasta(path):
    with open(path) as fh:
        fh.readline()
        return "".join(line.strip() for line in fh)


# ----------------------------------------------------------------------------- inputs
def synth_ref(n, seed):
    """i.i.d. uniform ACGT, numpy default_rng(seed).integers(0,4,n)."""
    codes = np.random.default_rng(seed).integers(0, 4, n).astype(np.uint8)
    return codes


def codes_to_str(codes):
    return "".join(ACGT[c] for c in codes)


def str_to_codes(s):
    lut = np.full(256, 255, np.uint8)
    for i, c in enumerate(ACGT):
        lut[ord(c)] = i
    return lut[np.frombuffer(s.encode(), np.uint8)]


def reads_from_ref(ref_codes, N, L, seed):
    """Mirror of the reference's create_query_from_ref distribution (SMEM.py:496-505):
    concatenated segments ref[p:p+s], p~U{0..n-1}, s~U{1..30}, rejected when p+s>n,
    truncated to L -- but drawn from the reference WITHOUT the trailing '$'."""
    rng = np.random.default_rng(seed)
    n = len(ref_codes)
    out = np.empty((N, L), np.uint8)
    for r in range(N):
        buf, have = [], 0
        while have < L:
            p = int(rng.integers(0, n))
            s = int(rng.integers(1, 31))
            if p + s > n:
                continue
            buf.append(ref_codes[p:p + s])
            have += s
        out[r] = np.concatenate(buf)[:L]
    return out


def reads_random(N, L, seed):
    """Mirror of create_random_query (SMEM.py:489-493): i.i.d. uniform bases."""
    return np.random.default_rng(seed).integers(0, 4, (N, L)).astype(np.uint8)


def reads_edge(ref_codes, L, K, seed):
    """Hand-picked edge reads of length L: a pure reference window, the reference tail
    (exercises the '$' row), homopolymers, a read with one SNP per K bases, tandem repeat."""
    rng = np.random.default_rng(seed)
    n = len(ref_codes)
    rows = []
    for p in (0, 1, n // 2, n - L - 1, n - L):           # windows incl. the exact tail
        if 0 <= p and p + L <= n:
            rows.append(ref_codes[p:p + L].copy())
    for b in range(4):                                     # homopolymers
        rows.append(np.full(L, b, np.uint8))
    w = ref_codes[n // 3:n // 3 + L].copy()               # SNP every K bases
    w[K - 1::K] = (w[K - 1::K] + 1) % 4
    rows.append(w)
    w = ref_codes[n // 5:n // 5 + L].copy()               # SNP every 2K+1 bases
    w[2 * K::2 * K + 1] = (w[2 * K::2 * K + 1] + 2) % 4
    rows.append(w)
    unit = ref_codes[7:7 + 5]
    rows.append(np.resize(unit, L).astype(np.uint8))       # tandem repeat of a ref 5-mer
    tail = ref_codes[n - (L // 2):]                        # half tail + half random
    rows.append(np.concatenate([tail, rng.integers(0, 4, L - len(tail)).astype(np.uint8)]))
    rows.append(np.concatenate([rng.integers(0, 4, L - len(tail)).astype(np.uint8), tail]))
    return np.stack(rows)


# ----------------------------------------------------------------------------- reference drivers
class Ref:
    """Thin driver around the reference's own classes for one (fasta, K) pair."""

    def __init__(self, fasta, K, data_dir, experts=None, build_fm="auto"):
        from ExactMatch import ExactMatch
        from LUT import LUT
        self.fasta, self.K, self.data_dir = fasta, K, data_dir
        stem = fasta.split(".")[0]
        self.matcher = ExactMatch(fasta)
        fm_path = os.path.join(data_dir, stem + "-FM.json")
        if not os.path.exists(fm_path):
            t = time.time()
            self.matcher.create_fm_index()                  # ExactMatch.py:22 (O(n^2) builder)
            print(f"  [ref] create_fm_index({fasta}) {time.time() - t:.1f}s", flush=True)
        self.matcher = ExactMatch(fasta)
        self.matcher.load_fm_index()
        self.matcher.load_ref_sequence()
        lut_path = os.path.join(data_dir, stem + "-LUT.json")
        need = True
        if os.path.exists(lut_path):
            with open(lut_path) as fh:
                need = json.load(fh)["lut_size"] != K
        if need:
            t = time.time()
            lut = LUT(self.matcher)
            lut.generate_lut(K)                              # LUT.py:15
            lut.save_lut()
            print(f"  [ref] generate_lut({K}) {time.time() - t:.1f}s", flush=True)
        from SMEM import SMEM                                # needs the Bio shim (SMEM.py:3)
        self.smem = SMEM(self.matcher)
        assert self.smem.lut.lut_size == K
        self.rmi = None
        if experts is not None:
            self.train_rmi(experts)

    def train_rmi(self, experts):
        from RMI_LUT import RMI_LUT
        import io
        import contextlib
        r = RMI_LUT(list(experts), self.K, self.fasta)
        with contextlib.redirect_stdout(io.StringIO()):
            r.train_RMI()                                    # RMI_LUT.py:36
        r.save("rmi_file.pkl")                               # name hard-coded at SMEM.py:207
        self.rmi = r
        return r

    # -- traversals -----------------------------------------------------------------
    def run_lut_or_rmi(self, q, which):
        """Returns (status, dict items, [(start,end)...]) -- the (start,end) trace is read
        off the reference's own loop variables at the `while end_smem_index < len(query)`
        line (SMEM.py:49 / :235) via sys.settrace, without touching its code."""
        fn = self.smem.get_smems_lut if which == "lut" else self.smem.get_smems_rmi
        line = 49 if which == "lut" else 235
        code = fn.__func__.__code__
        trace = []

        def local(frame, event, arg):
            if event == "line" and frame.f_lineno == line:
                trace.append((frame.f_locals["end_smem_index"] - frame.f_locals["prev_smem_length"],
                              frame.f_locals["end_smem_index"]))
            return local

        def glob(frame, event, arg):
            return local if frame.f_code is code else None

        import io
        import contextlib
        sys.settrace(glob)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                res = fn(q)
            status = 0
        except RecursionError:
            res, status = {}, 2
        except Exception as e:  # noqa: BLE001 - the reference raises bare exceptions
            res, status = {}, 1
            self.last_error = repr(e)
        finally:
            sys.settrace(None)
        items = [(k, int(v[0]), int(v[1])) for k, v in res.items()]
        return status, items, trace

    def run_bwa(self, q, min_len=1):
        """get_SMEMS (SMEM.py:456); the per-step [string, interval, end] comes from wrapping
        the public get_SMEM_at_index on the instance."""
        steps = []
        orig = self.smem.get_SMEM_at_index

        def wrapped(query, start_index):
            r = orig(query, start_index)
            steps.append((r[2] - len(r[0]), r[2], int(r[1][0]), int(r[1][1])))
            return r

        self.smem.get_SMEM_at_index = wrapped
        try:
            res = self.smem.get_SMEMS(q, min_len)
            status = 0
        except Exception as e:  # noqa: BLE001
            res, status = {}, 1
            self.last_error = repr(e)
        finally:
            del self.smem.get_SMEM_at_index
        items = [(k, int(v[0]), int(v[1])) for k, v in res.items()]
        return status, items, steps


def flatten_items(per_read_items):
    off = [0]
    strs, ivs = [], []
    soff = [0]
    for items in per_read_items:
        for s, lo, hi in items:
            strs.append(str_to_codes(s))
            soff.append(soff[-1] + len(s))
            ivs.append((lo, hi))
        off.append(off[-1] + len(items))
    return dict(
        items_off=np.asarray(off, np.int64),
        items_soff=np.asarray(soff, np.int64),
        items_str=(np.concatenate(strs) if strs else np.zeros(0, np.uint8)),
        items_iv=np.asarray(ivs, np.int32).reshape(-1, 2),
    )


def flatten_trace(per_read):
    off = [0]
    rows = []
    for t in per_read:
        rows.extend(t)
        off.append(off[-1] + len(t))
    w = len(rows[0]) if rows else 2
    return np.asarray(off, np.int64), np.asarray(rows, np.int32).reshape(-1, w)


def run_group(ref, reads, algos, tag, rmi_limit=None):
    """Run the reference on every read of `reads` for each algorithm; return npz dict."""
    out = {f"{tag}.reads": reads}
    qs = [codes_to_str(r) for r in reads]
    for algo in algos:
        t0 = time.time()
        sub = qs if not (algo == "rmi" and rmi_limit) else qs[:rmi_limit]
        stat, items, traces = [], [], []
        for q in sub:
            if algo == "bwa":
                s, it, tr = ref.run_bwa(q, 1)
            else:
                s, it, tr = ref.run_lut_or_rmi(q, algo)
            stat.append(s)
            items.append(it)
            traces.append(tr)
        fl = flatten_items(items)
        for k, v in fl.items():
            out[f"{tag}.{algo}.{k}"] = v
        toff, trows = flatten_trace(traces)
        out[f"{tag}.{algo}.trace_off"] = toff
        out[f"{tag}.{algo}.trace"] = trows          # bwa: (start,end,lo,hi); lut/rmi: (start,end)
        out[f"{tag}.{algo}.status"] = np.asarray(stat, np.int8)
        nb = sum(len(q) for q in sub)
        print(f"  [{tag}] {algo}: {len(sub)} reads, {nb / (time.time() - t0):.0f} bases/s, "
              f"{fl['items_iv'].shape[0]} items, raised={int(np.sum(np.asarray(stat) != 0))}", flush=True)
    return out


# ----------------------------------------------------------------------------- fixture builders
def fm_summary(matcher):
    fm = matcher.fm_index
    sa = np.asarray(fm["suffix_array"], np.int64)
    return dict(
        n_rows=int(len(sa)),
        sa_sha256=hashlib.sha256(sa.astype("<i4").tobytes()).hexdigest(),
        sa_head=[int(x) for x in sa[:64]],
        sa_tail=[int(x) for x in sa[-64:]],
        count_dic={k: int(v) for k, v in fm["count_dic"].items()},
    )


def lut_summary(lut_obj):
    """Digest of the reference LUT: sorted (code, lo, hi, positions...) stream."""
    h = hashlib.sha256()
    keys = sorted(lut_obj.lut.keys(), key=int)
    npos = 0
    maxocc = 0
    for k in keys:
        iv, pos = lut_obj.lut[k]
        h.update(np.asarray([int(k), iv[0], iv[1]] + list(pos), "<i8").tobytes())
        npos += len(pos)
        maxocc = max(maxocc, len(pos))
    return dict(K=int(lut_obj.lut_size), n_keys=len(keys), n_pos=npos, max_occ=maxocc,
                sha256=h.hexdigest(),
                head=[[int(k)] + [int(x) for x in lut_obj.lut[k][0]] for k in keys[:16]])


def exact_match_patterns(ref_codes, count, seed, maxlen=150):
    rng = np.random.default_rng(seed)
    n = len(ref_codes)
    pats = []
    for i in range(count):
        kind = i % 5
        ln = int(rng.integers(1, maxlen + 1))
        if kind == 0:                                   # present
            p = int(rng.integers(0, n - ln + 1))
            pats.append(ref_codes[p:p + ln].copy())
        elif kind == 1:                                 # random (absent when long)
            pats.append(rng.integers(0, 4, ln).astype(np.uint8))
        elif kind == 2:                                 # near miss: last base flipped
            p = int(rng.integers(0, n - ln + 1))
            w = ref_codes[p:p + ln].copy()
            w[-1] = (w[-1] + 1 + rng.integers(0, 3)) % 4
            pats.append(w)
        elif kind == 3:                                 # short (wide intervals)
            ln = int(rng.integers(1, 9))
            pats.append(rng.integers(0, 4, ln).astype(np.uint8))
        else:                                           # reference tail / tail + extra base
            ln = min(ln, n)
            w = ref_codes[n - ln:].copy()
            if rng.integers(0, 2):
                w = np.concatenate([w, rng.integers(0, 4, 1).astype(np.uint8)])[-maxlen:]
            pats.append(w)
    return pats


def g5_exact_match(ref, ref_codes, count, seed):
    pats = exact_match_patterns(ref_codes, count, seed)
    off = [0]
    res = []
    for p in pats:
        r = ref.matcher.exact_match_back_prop(codes_to_str(p))      # ExactMatch.py:132
        res.append((-1, -1) if r == -1 else (int(r[0]), int(r[1])))
        off.append(off[-1] + len(p))
    return {
        "g5.pat": np.concatenate(pats),
        "g5.pat_off": np.asarray(off, np.int64),
        "g5.lohi": np.asarray(res, np.int32),
    }


def g4_rmi(ref, ref_codes, experts, n_kmers, seed, tag):
    """RMI fixtures: per-level float64 coefficients (plain arrays, not pickles), the float64
    prediction of rmi_predict and (lower, upper) of get_suffix_rmi for a set of K-mers."""
    rmi_lut = ref.train_rmi(experts)
    K = ref.K
    out = {}
    for lvl, models in enumerate(rmi_lut.rmi.models):
        out[f"{tag}.coef{lvl}"] = np.asarray([float(np.ravel(m.coef_)[0]) for m in models], np.float64)
        out[f"{tag}.icpt{lvl}"] = np.asarray([float(m.intercept_) for m in models], np.float64)
    out[f"{tag}.experts"] = np.asarray(list(experts), np.int64)
    rng = np.random.default_rng(seed)
    n = len(ref_codes)
    sa = ref.matcher.fm_index["suffix_array"]
    kmers = []
    # (a) K-mers adjacent to the "None rows" (suffix shorter than K) -- the defect zone
    none_rows = [i for i in range(len(sa)) if sa[i] - 1 + K > n]
    for r in none_rows:
        for d in (-2, -1, 1, 2):
            rr = r + d
            if 0 <= rr < len(sa) and sa[rr] - 1 + K <= n:
                kmers.append(ref_codes[sa[rr] - 1:sa[rr] - 1 + K].copy())
    # (b) present K-mers, (c) random (mostly absent), (d) near misses
    while len(kmers) < n_kmers:
        kind = len(kmers) % 3
        if kind == 0:
            p = int(rng.integers(0, n - K + 1))
            kmers.append(ref_codes[p:p + K].copy())
        elif kind == 1:
            kmers.append(rng.integers(0, 4, K).astype(np.uint8))
        else:
            p = int(rng.integers(0, n - K + 1))
            w = ref_codes[p:p + K].copy()
            j = int(rng.integers(0, K))
            w[j] = (w[j] + 1) % 4
            kmers.append(w)
    kmers = np.stack(kmers[:n_kmers])
    pred = np.zeros(len(kmers), np.float64)
    lohi = np.zeros((len(kmers), 2), np.int64)
    status = np.zeros(len(kmers), np.int8)
    truth = np.zeros((len(kmers), 2), np.int32)
    for i, km in enumerate(kmers):
        s = codes_to_str(km)
        pred[i] = float(rmi_lut.rmi_predict(s, False)[0])           # RMI_LUT.py:53
        t = ref.matcher.exact_match_back_prop(s)
        truth[i] = (-1, -1) if t == -1 else t
        try:
            lo, hi = rmi_lut.get_suffix_rmi(s, False)               # RMI_LUT.py:67
            lohi[i] = (lo, hi)
        except RecursionError:
            status[i] = 2
        except Exception:  # noqa: BLE001
            status[i] = 1
    out[f"{tag}.kmers"] = kmers
    out[f"{tag}.pred"] = pred
    out[f"{tag}.lohi"] = lohi
    out[f"{tag}.status"] = status
    out[f"{tag}.truth"] = truth
    absent = truth[:, 0] < 0
    ok = status == 0
    rmi_absent = lohi[:, 0] > lohi[:, 1]
    defect = ok & ((absent != rmi_absent) | (~absent & ((lohi[:, 0] != truth[:, 0]) | (lohi[:, 1] != truth[:, 1]))))
    out[f"{tag}.defect"] = (defect | ~ok).astype(np.int8)
    print(f"  [{tag}] experts={experts}: {len(kmers)} kmers, raised={int(np.sum(~ok))}, "
          f"defects={int(np.sum(defect))}", flush=True)
    return out


# ----------------------------------------------------------------------------- datasets
def ds_known(data_dir):
    """G1: the reference's own documented examples, re-measured."""
    from ExactMatch import ExactMatch
    from LUT import LUT
    from SMEM import SMEM
    out = {}
    m = ExactMatch("mississippi.fa")
    m.load_fm_index()
    m.load_ref_sequence()
    out["mississippi"] = dict(
        ref="mississippi",
        fm=fm_summary(m),
        back_prop={q: (lambda r: list(r) if r != -1 else -1)(m.exact_match_back_prop(q))
                   for q in ["iss", "ssi", "i", "m", "p", "s", "mississippi", "ississippi", "pp",
                             "sip", "issip", "ssissi", "mm", "pi", "ip", "ppi", "", "ims"]},
        exact_match={q: m.exact_match(q) for q in ["iss", "i", "ssi", "p"]},
    )
    # LUT object needs a LUT json in the *current* (ACGT) format: the checked-in
    # mississippi-LUT.json is stale (raw-string keys), so only the BWA path is asked here.
    s = SMEM(m)            # loads the (stale-format) mississippi-LUT.json; only get_SMEMS is used
    bwa = {}
    for q in ["pissssi", "missi", "mmissi", "missippi", "mmiss", "mmissippss", "ssissippim", "ipssm"]:
        r = s.get_SMEMS(q, 1)
        bwa[q] = [[k, int(v[0]), int(v[1])] for k, v in r.items()]
    out["mississippi"]["get_SMEMS"] = bwa
    bwa2 = {}
    for q in ["mmissippss", "pissssi"]:
        r = s.get_SMEMS(q, 3)
        bwa2[q] = [[k, int(v[0]), int(v[1])] for k, v in r.items()]
    out["mississippi"]["get_SMEMS_min3"] = bwa2

    # paper section 2 example: ref CTCAATGC, query ACTGC
    write_fasta(data_dir, "paperex.fa", "CTCAATGC", "paper example")
    m2 = ExactMatch("paperex.fa")
    m2.create_fm_index()
    m2 = ExactMatch("paperex.fa")
    m2.load_fm_index()
    lut = LUT(m2)
    lut.generate_lut(2)
    lut.save_lut()
    s2 = SMEM(m2)
    out["paperex"] = dict(
        ref="CTCAATGC", K=2, fm=fm_summary(m2),
        lut={k: v for k, v in s2.lut.lut.items()},
        get_SMEMS={q: [[k, int(v[0]), int(v[1])] for k, v in s2.get_SMEMS(q, 1).items()]
                   for q in ["ACTGC", "CTCAATGC", "GGGG", "TTCAATT"]},
        get_smems_lut={q: [[k, int(v[0]), int(v[1])] for k, v in s2.get_smems_lut(q).items()]
                       for q in ["ACTGC", "CTCAATGC", "TTCAATT", "CAATGCA"]},
    )
    # small_data.fa (50 bases) with the checked-in FM fixture
    m3 = ExactMatch("small_data.fa")
    m3.load_fm_index()
    m3.load_ref_sequence()
    out["small_data"] = dict(ref=m3.ref_sequence[:-1], fm=fm_summary(m3))
    return out


def ds_generic(name, data_dir, ref_codes, fasta, K, groups, experts, n_g5, n_g4, rmi_limit):
    ref = Ref(fasta, K, data_dir, experts=experts)
    out = {"ref_codes": ref_codes, "K": np.asarray(K), "experts": np.asarray(experts, np.int64)}
    meta = dict(name=name, n=int(len(ref_codes)), K=K, experts=list(experts),
                fm=fm_summary(ref.matcher), lut=lut_summary(ref.smem.lut),
                needs_bio_shim=True, groups={})
    for tag, reads, algos in groups:
        out.update(run_group(ref, reads, algos, tag, rmi_limit=rmi_limit))
        meta["groups"][tag] = dict(n=int(reads.shape[0]), L=int(reads.shape[1]), algos=algos)
    if n_g5:
        out.update(g5_exact_match(ref, ref_codes, n_g5, seed=55))
    if n_g4:
        for ex in n_g4:
            out.update(g4_rmi(ref, ref_codes, ex, 3000, seed=44, tag="g4_" + "_".join(map(str, ex))))
        ref.train_rmi(experts)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    with open(os.path.join(HERE, name + ".json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print(f"wrote {name}.npz ({os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--scratch", default="/tmp/oracle")
    args = ap.parse_args()
    if not os.path.isdir(REF_SMEM):
        sys.exit("reference not present: goldens can only be (re)generated in the build container")
    install_bio_shim()
    work, data_dir = setup_scratch(args.scratch)

    def want(n):
        return args.only is None or args.only == n

    if want("known"):
        with open(os.path.join(HERE, "known_answers.json"), "w") as fh:
            json.dump(ds_known(data_dir), fh, indent=1, sort_keys=True)
        print("wrote known_answers.json", flush=True)

    if want("medium_K6"):
        # reference's own 10 kb fixture files (FM + K=6 LUT are the checked-in JSONs, not rebuilt)
        codes = str_to_codes(read_fasta(os.path.join(data_dir, "medium_data.fa")))
        groups = [
            ("fromref100", reads_from_ref(codes, 200, 100, 601), ["bwa", "lut", "rmi"]),
            ("random100", reads_random(100, 100, 602), ["bwa", "lut", "rmi"]),
            ("edge60", reads_edge(codes, 60, 6, 603), ["bwa", "lut", "rmi"]),
        ]
        ds_generic("medium_K6", data_dir, codes, "medium_data.fa", 6, groups, [10, 100], 2000, None, 60)

    if want("syn10k_K8"):
        codes = synth_ref(10_000, 10_000)
        write_fasta(data_dir, "syn10k.fa", codes_to_str(codes))
        groups = [
            ("fromref150", reads_from_ref(codes, 300, 150, 801), ["bwa", "lut", "rmi"]),
            ("random150", reads_random(150, 150, 802), ["bwa", "lut", "rmi"]),
            ("edge40", reads_edge(codes, 40, 8, 803), ["bwa", "lut", "rmi"]),
            ("fromref8", reads_from_ref(codes, 40, 8, 804), ["bwa", "lut", "rmi"]),      # L == K
            ("fromref9", reads_from_ref(codes, 40, 9, 805), ["bwa", "lut", "rmi"]),      # L == K+1
            ("fromref600", reads_from_ref(codes, 20, 600, 806), ["bwa", "lut"]),
        ]
        ds_generic("syn10k_K8", data_dir, codes, "syn10k.fa", 8, groups, [10, 100], 2000, None, 80)

    if want("syn100k_K15"):
        codes = synth_ref(100_000, 100_000)
        write_fasta(data_dir, "syn100k.fa", codes_to_str(codes))
        groups = [
            ("cfg1_fromref100", reads_from_ref(codes, 1000, 100, 1001), ["bwa", "lut", "rmi"]),
            ("cfg2_fromref150", reads_from_ref(codes, 1000, 150, 1002), ["bwa", "lut", "rmi"]),
            ("random150", reads_random(200, 150, 1502), ["bwa", "lut", "rmi"]),
            ("edge150", reads_edge(codes, 150, 15, 1503), ["bwa", "lut", "rmi"]),
            ("fromref15", reads_from_ref(codes, 30, 15, 1504), ["bwa", "lut", "rmi"]),
            ("fromref2000", reads_from_ref(codes, 10, 2000, 1505), ["bwa", "lut"]),
        ]
        ds_generic("syn100k_K15", data_dir, codes, "syn100k.fa", 15, groups, [1000], 6000,
                   [[1000], [10, 100]], 100)

    if want("big100k_K15"):
        # the reference's own real-genome file (telomere repeats => wide SA intervals)
        codes = str_to_codes(read_fasta(os.path.join(data_dir, "big_data.fa")))
        groups = [
            ("fromref150", reads_from_ref(codes, 500, 150, 2001), ["bwa", "lut", "rmi"]),
            ("random150", reads_random(100, 150, 2002), ["bwa", "lut", "rmi"]),
            ("edge150", reads_edge(codes, 150, 15, 2003), ["bwa", "lut", "rmi"]),
            ("head150", np.stack([codes[i:i + 150] for i in range(0, 9000, 300)]), ["bwa", "lut", "rmi"]),
        ]
        ds_generic("big100k_K15", data_dir, codes, "big_data.fa", 15, groups, [1000], 4000,
                   [[1000]], 60)


if __name__ == "__main__":
    main()

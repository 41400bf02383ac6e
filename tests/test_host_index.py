"""CPU-only tests of the product's host side: the C-ABI library loads and exports every
declared symbol, and the native index builder (suffix array, K-mer table, prefix directory,
image) agrees with the reference's outputs held in tests/golden/.  No compute kernels run here.
"""
import os
import re
import struct

import numpy as np
import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import genie_smem_amd as g
    g._native.build()
    g._native.lib()
    return g


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "genie_smem.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(genie_[a-z_0-9]+)\s*\(", hdr)))
    assert sorted(pkg._native.SYMBOLS) == declared
    lib = pkg._native.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.genie_abi_version() == pkg._native.ABI_VERSION == 2
    assert lib.genie_strerror(-7) == b"no RMI model installed"


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._native, "_lib", None)
    monkeypatch.setattr(pkg._native, "LIB_PATH", "/nonexistent/libgenie_smem.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg._native.lib()


def test_compute_without_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ix = pkg.GenieIndex.build(np.asarray([0, 1, 2, 3, 0, 1], np.uint8), 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ix.sa_interval(np.zeros((1, 2), np.uint8))
    with pytest.raises(RuntimeError):
        ix.to("cpu")


# ------------------------------------------------------------------ SA + K-mer table vs reference
def test_known_suffix_arrays(pkg):
    k = G.known()
    for name, alpha in (("mississippi", "imps"), ("paperex", "ACGT"), ("small_data", "ACGT")):
        ix = pkg.GenieIndex.build(G.str_to_codes(k[name]["ref"], alpha), 0)
        assert G.sa_sha256(ix.suffix_array()) == k[name]["fm"]["sa_sha256"], name
    ix = pkg.GenieIndex.build(G.str_to_codes("CTCAATGC"), 2)
    assert ix.suffix_array().tolist() == [9, 4, 5, 8, 3, 1, 7, 2, 6]
    codes, lo, hi = ix.lut_arrays()
    sa = ix.suffix_array()
    got = {str(int(c)): [[int(a), int(b)], sa[a:b + 1].tolist()] for c, a, b in zip(codes, lo, hi)}
    assert got == k["paperex"]["lut"]


@pytest.mark.parametrize("ds", [d for d in G.DATASETS if G.have(d)])
def test_index_matches_reference_digests(pkg, ds):
    d, meta = G.load(ds)
    ix = pkg.GenieIndex.build(d["ref_codes"], int(d["K"]))
    sa = ix.suffix_array()
    assert G.sa_sha256(sa) == meta["fm"]["sa_sha256"]
    codes, lo, hi = ix.lut_arrays()
    sha, npos, maxocc = G.lut_sha256(codes, lo, hi, sa)
    assert (len(codes), sha, npos, maxocc) == (meta["lut"]["n_keys"], meta["lut"]["sha256"], meta["lut"]["n_pos"],
                                               meta["lut"]["max_occ"])
    # adopting the reference's own suffix array gives the same index image
    ix2 = pkg.GenieIndex.build(d["ref_codes"], int(d["K"]), sa_one_based=sa.copy())
    assert bytes(ix2.serialize().numpy()) == bytes(ix.serialize().numpy())


def test_create_from_sa_rejects_garbage(pkg):
    codes = np.asarray([0, 1, 2, 3, 0, 1, 2], np.uint8)
    good = pkg.GenieIndex.build(codes, 0).suffix_array().copy()
    bad = good.copy()
    bad[1], bad[2] = bad[2], bad[1]
    with pytest.raises(pkg._native.GenieError):
        pkg.GenieIndex.build(codes, 0, sa_one_based=bad)
    bad = good.copy()
    bad[3] = bad[4]
    with pytest.raises(pkg._native.GenieError):
        pkg.GenieIndex.build(codes, 0, sa_one_based=bad)
    with pytest.raises(pkg._native.GenieError):
        pkg.GenieIndex.build(np.asarray([0, 4], np.uint8), 0)


@pytest.mark.skipif(not os.path.isdir("/root/reference/SMEM/data"), reason="reference data files not present")
def test_reference_checked_in_fixtures(pkg):
    """The reference's own FM / LUT json fixtures (read as data, only where the reference exists)."""
    import json
    data = "/root/reference/SMEM/data"
    for stem in ("small_data", "medium_data"):
        with open(os.path.join(data, stem + ".fa")) as fh:
            fh.readline()
            seq = "".join(line.strip() for line in fh)
        fm = json.load(open(os.path.join(data, stem + "-FM.json")))
        ix = pkg.GenieIndex.build(G.str_to_codes(seq), 6)
        assert ix.suffix_array().tolist() == fm["suffix_array"]
    lut = json.load(open(os.path.join(data, "medium_data-LUT.json")))
    assert lut["lut_size"] == 6
    codes, lo, hi = ix.lut_arrays()
    sa = ix.suffix_array()
    got = {str(int(c)): [[int(a), int(b)], sa[a:b + 1].tolist()] for c, a, b in zip(codes, lo, hi)}
    assert got == lut["lut"]


# ------------------------------------------------------------------ image + prefix directory
HDR = struct.Struct("<QIIqqiiqqqqqqqqqqi4i4i5i8Iqqiiqqqqqq")


def _parse(img):
    f = HDR.unpack(bytes(img[:HDR.size]))
    keys = ["magic", "version", "header_bytes", "total_bytes", "n", "K", "P", "off_sa", "off_ref", "off_dir",
            "off_lut", "off_rmi", "ref_recs", "dir_entries", "lut_slots", "lut_keys", "rmi_models", "nlev"]
    h = dict(zip(keys, f[:18]))
    h["padtail"] = list(f[-18:-10])
    h["off_dir2"], h["dir2_entries"], h["P2"] = f[-10], f[-9], f[-8]
    h["flags"] = f[-7]
    h["off_rmi_err"], h["rmi_err_entries"] = f[-6], f[-5]
    h["off_mtab"], h["mtab_entries"] = f[-4], f[-3]
    h["off_ov"], h["ov_entries"] = f[-2], f[-1]
    return h


def _dir_interval(h, dirv, code, m):
    """Python mirror of dir_lb / dir_ub in kernels.hip."""
    P = h["P"]
    x = code << (2 * (P - m))
    lb = int(dirv[x]) - sum(1 for l in range(m, P) if h["padtail"][l] == x)
    y = (code + 1) << (2 * (P - m))
    ub = int(dirv[y]) - sum(1 for l in range(1, P) if h["padtail"][l] == y)
    return lb, ub


@pytest.mark.parametrize("tail", ["", "A", "AAAAAAAA", "TTTTTTTT", "CAAAAAA", "GTTTTTT", "ACGTTTA", "TTTTTTA", "C"])
@pytest.mark.parametrize("P", [3, 7])
def test_prefix_directory_brute_force(pkg, tail, P):
    rng = np.random.default_rng(len(tail) * 31 + P)
    body = "".join("ACGT"[c] for c in rng.integers(0, 4, 300))
    ref = body + tail
    ix = pkg.GenieIndex.build(G.str_to_codes(ref), 4, dir_bits=P)
    img = ix.serialize().numpy()
    h = _parse(img)
    assert h["magic"] == 0x58444947454E4547 and h["n"] == len(ref) and h["P"] == P and h["total_bytes"] == len(img)
    dirv = np.frombuffer(bytes(img[h["off_dir"]:h["off_dir"] + 4 * h["dir_entries"]]), np.uint32)
    recs = np.frombuffer(bytes(img[h["off_sa"]:h["off_sa"] + 16 * (len(ref) + 1)]), np.int32).reshape(-1, 4)
    sa0 = recs[:, 0]
    keys = np.frombuffer(bytes(img[h["off_sa"]:h["off_sa"] + 16 * (len(ref) + 1)]), np.uint64).reshape(-1, 2)[:, 1]
    for r in (0, 1, len(ref) // 2, len(ref)):                     # inline key = 32 bases after the P-base prefix
        want = 0
        for j in range(32):
            p_ = int(sa0[r]) + P + j
            want |= ("ACGT".index(ref[p_]) if p_ < len(ref) else 0) << (62 - 2 * j)
        assert int(keys[r]) == want
    assert (sa0 + 1).tolist() == ix.suffix_array().tolist()
    suffixes = [ref[s:] + "$" for s in sa0]
    assert suffixes == sorted(suffixes)
    assert dirv[-1] == len(ref) + 1
    for m in range(1, P + 1):
        for code in range(4 ** m):
            pat = "".join("ACGT"[(code >> (2 * (m - 1 - j))) & 3] for j in range(m))
            rows = [r for r, s in enumerate(suffixes) if s.startswith(pat)]
            lb, ub = _dir_interval(h, dirv, code, m)
            if rows:
                assert (lb, ub) == (rows[0], rows[-1] + 1), (pat, tail)
            else:
                assert lb >= ub, (pat, tail)


def test_packed_reference_and_lut_slots(pkg):
    d, meta = G.load("medium_K6")
    ix = pkg.GenieIndex.build(d["ref_codes"], 6)
    img = ix.serialize().numpy()
    h = _parse(img)
    n = h["n"]
    recs = np.frombuffer(bytes(img[h["off_ref"]:h["off_ref"] + 16 * h["ref_recs"]]), np.uint64).reshape(-1, 2)
    assert (recs[:-1, 1] == recs[1:, 0]).all()
    words = recs[:, 0]
    bases = np.zeros(len(words) * 32, np.uint8)
    for j in range(32):
        bases[j::32] = (words >> np.uint64(62 - 2 * j)) & np.uint64(3)
    assert (bases[:n] == d["ref_codes"]).all() and not bases[n:].any()
    slots = np.frombuffer(bytes(img[h["off_lut"]:h["off_lut"] + 16 * h["lut_slots"]]), np.int32).reshape(-1, 4)
    used = slots[slots[:, 1] >= 0]
    codes, lo, hi = ix.lut_arrays()
    assert len(used) == len(codes) == h["lut_keys"]
    order = np.argsort(used[:, 0].astype(np.uint32))
    assert (used[order, 0].astype(np.uint32) == codes).all() and (used[order, 1] == lo).all() and (used[order, 2] == hi).all()


# ------------------------------------------------------------------ RMI host side
@pytest.mark.parametrize("ds", [d for d in ("syn100k_K15",) if G.have(d)])
def test_rmi_predict_bit_exact_with_reference_coefficients(pkg, ds):
    d, _ = G.load(ds)
    K = int(d["K"])
    for tag in sorted({k.split(".")[0] for k in d if k.startswith("g4_")}):
        ex = [int(x) for x in d[f"{tag}.experts"]]
        rmi = pkg.RMI.from_coefficients(ex, [d[f"{tag}.coef{l}"] for l in range(len(ex) + 1)],
                                        [d[f"{tag}.icpt{l}"] for l in range(len(ex) + 1)])
        w = 4 ** np.arange(K - 1, -1, -1, dtype=np.int64)
        codes = (d[f"{tag}.kmers"].astype(np.int64) * w).sum(1)
        assert (rmi.predict(codes.reshape(-1, 1)) == d[f"{tag}.pred"]).all()


def test_rmi_fit_is_a_usable_model(pkg):
    d, _ = G.load("syn10k_K8")
    m = pkg.ExactMatch("syn10k.fa")
    m.set_reference(G.codes_to_str(d["ref_codes"]))
    r = pkg.RMI_LUT([10, 100], 8, "syn10k.fa", matcher=m)
    r.train_RMI()
    assert [len(l) for l in r.rmi.models] == [1, 10, 100]
    sa = np.asarray(m.host_index(8).suffix_array(), np.int64)
    rows = np.nonzero(sa - 1 + 8 <= len(d["ref_codes"]))[0]
    codes = d["ref_codes"].astype(np.int64)
    key = np.zeros(len(rows), np.int64)
    for j in range(8):
        key = (key << 2) | codes[sa[rows] - 1 + j]
    err = np.abs(r.rmi.predict(key.reshape(-1, 1)) - rows)
    assert err.mean() < 30 and err.max() < 400
    # scalar API shape: array of one float64, like the reference's rmi_predict
    q = G.codes_to_str(d["ref_codes"][100:108])
    assert r.rmi_predict(q).shape == (1,)


@pytest.mark.parametrize("ds", ["syn100k_K15", "big100k_K15"])
def test_native_rmi_fit_against_reference_sklearn_coefficients(pkg, ds):
    """genie_index_train_rmi against the models the REFERENCE fitted with scikit-learn (RMI.fit, SMEM/RMI.py:10-50;
    coefficients exported into the fixtures by make_golden.py): on the fixture K-mers the two models' predictions
    differ by less than one suffix-array row -- the fits differ in least-squares arithmetic only.  (Results cannot
    depend on it: a prediction only picks where a bounded search starts.)"""
    d, _ = G.load(ds)
    ref, K = d["ref_codes"], int(d["K"])
    for tag in sorted({k.split(".")[0] for k in d if k.startswith("g4_")}):
        experts = [int(x) for x in d[tag + ".experts"]]
        coefs = [d[f"{tag}.coef{l}"] for l in range(len(experts) + 1)]
        icpts = [d[f"{tag}.icpt{l}"] for l in range(len(experts) + 1)]
        ix = pkg.GenieIndex.build(ref, K)
        ncoefs, nicpts, _, _, _ = ix.train_rmi(experts)
        assert [len(c) for c in ncoefs] == [len(c) for c in coefs]
        kmers = d[tag + ".kmers"].astype(np.int64)
        key = np.zeros(len(kmers), np.int64)
        for j in range(K):
            key = (key << 2) | kmers[:, j]
        p_ref = pkg.RMI.from_coefficients(experts, coefs, icpts).predict(key)
        assert np.array_equal(p_ref, d[tag + ".pred"])                      # the fixture's own float64 predictions
        p_nat = pkg.RMI.from_coefficients(experts, ncoefs, nicpts).predict(key)
        # K-mers routed to the same leaf by both models (a boundary K-mer may go to the neighbouring leaf)
        delta = np.abs(p_nat - p_ref)
        assert np.median(delta) < 1e-2 and (delta < 1.0).mean() > 0.995, (tag, np.median(delta), (delta < 1.0).mean())


def test_native_rmi_training(pkg):
    """genie_index_train_rmi (C++) against the numpy restatement of RMI.fit on the same pairs: same
    structure, near-identical predictions, and leaf error bounds that do bound every training pair."""
    d, _ = G.load("syn100k_K15")
    ref, K = d["ref_codes"], 15
    for experts in ([1000], [10, 100], []):
        ix = pkg.GenieIndex.build(ref, K)
        coefs, icpts, err, mean, worst = ix.train_rmi(experts)
        assert [len(c) for c in coefs] == [1] + list(experts) and len(err) == ([1] + list(experts))[-1]
        sa0 = ix.suffix_array().astype(np.int64) - 1
        rows = np.nonzero(sa0 + K <= len(ref))[0]
        key = np.zeros(len(rows), np.int64)
        for j in range(K):
            key = (key << 2) | ref[sa0[rows] + j]
        native = pkg.RMI.from_coefficients(experts, coefs, icpts)
        ref_fit = pkg.RMI(list(experts)).fit(key.reshape(-1, 1), rows)
        p_nat, p_np = native.predict(key), ref_fit.predict(key)
        # the two fits differ in summation order only: predictions agree to a fraction of a row almost everywhere
        assert np.median(np.abs(p_nat - p_np)) < 1e-3
        e = np.abs(np.clip(np.trunc(p_nat), 0, len(ref)).astype(np.int64) - rows)
        assert abs(e.mean() - mean) < 1e-9 and e.max() == worst
        # leaf of every pair, by replaying the routing
        idx = np.zeros(len(key), np.int64)
        x = key.astype(np.float64)
        for lvl, scale in enumerate(list(experts)):
            p = x * coefs[lvl][idx] + icpts[lvl][idx]
            idx = np.minimum(scale - 1, np.maximum(0, np.trunc(p))).astype(np.int64)
        assert (e <= err[idx]).all()
        h = _parse(ix.serialize().numpy())
        assert h["rmi_err_entries"] == len(err) and h["nlev"] == len(experts) + 1
    # coefficients installed from outside carry no error table
    ix = pkg.GenieIndex.build(ref[:20000], K)
    ix.set_rmi([], [np.asarray([1e-5])], [np.asarray([0.0])])
    assert _parse(ix.serialize().numpy())["rmi_err_entries"] == 0


def test_second_level_range_table(pkg):
    """dir2[b] = {lb, cnt | short flag, key of row lb} for every P2-mer b (cnt == 0: absent), checked
    against the suffix array and the suffix-array records."""
    d, _ = G.load("syn100k_K15")
    ref = d["ref_codes"]
    ix = pkg.GenieIndex.build(ref, 15)
    img = ix.serialize().numpy()
    h = _parse(img)
    P2 = h["P2"]
    assert P2 == 8 and h["dir2_entries"] == 4 ** P2
    head = np.frombuffer(bytes(img[h["off_dir2"]:h["off_dir2"] + 16 * h["dir2_entries"]]),
                         np.dtype([("lb", "<u4"), ("meta", "<u4"), ("key", "<u8")]))
    cnt = (head["meta"] & 0x7FFFFFFF).astype(np.int64)
    pairs = np.stack([head["lb"].astype(np.int64), head["lb"].astype(np.int64) + cnt], axis=1)
    sarec = np.frombuffer(bytes(img[h["off_sa"]:h["off_sa"] + 16 * (len(ref) + 1)]),
                          np.dtype([("s", "<i4"), ("pad", "<i4"), ("key", "<u8")]))
    occ = cnt > 0
    assert (head["key"][occ] == sarec["key"][head["lb"][occ]]).all()                  # copy of the first row's key
    short = (head["meta"][occ] >> 31).astype(bool)
    assert (short == (len(ref) - sarec["s"][head["lb"][occ]] < h["P"] + 32)).all()
    sa0 = ix.suffix_array().astype(np.int64) - 1
    n = len(ref)
    full = sa0 + P2 <= n
    codes = np.zeros(len(sa0), np.int64)
    for j in range(P2):
        codes = (codes << 2) | np.where(full, ref[np.minimum(sa0 + j, n - 1)], 0)
    rows = np.nonzero(full)[0]
    c = codes[rows]
    assert (np.diff(c) >= 0).all()                                  # rows with a full P2-mer are sorted by it
    first = np.full(4 ** P2, -1, np.int64)
    last = np.full(4 ** P2, -1, np.int64)
    first[c[::-1]] = rows[::-1]
    last[c] = rows
    present = first >= 0
    assert (pairs[present, 0] == first[present]).all() and (pairs[present, 1] == last[present] + 1).all()
    assert (pairs[~present, 0] >= pairs[~present, 1]).all()
    # contiguity: the rows between first and last all carry the same P2-mer
    assert (last[present] - first[present] + 1 == np.bincount(c, minlength=4 ** P2)[present]).all()
    # small references get the smallest table (P2 = P + 1); the table size can be chosen at build time
    assert _parse(pkg.GenieIndex.build(ref[:5000], 8).serialize().numpy())["P2"] == 8
    assert _parse(pkg.GenieIndex.build(ref[:5000], 8, table_bits=9).serialize().numpy())["P2"] == 9


def _match_table(img, h):
    return np.frombuffer(bytes(img[h["off_mtab"]:h["off_mtab"] + 32 * h["mtab_entries"]]),
                         np.dtype([("meta", "<u4"), ("lb", "<u4"), ("key", "<u4", (6,))]))


@pytest.mark.parametrize("case", ["syn10k", "tail_A", "tiny", "repeat"])
def test_match_table_against_brute_force(pkg, case):
    """MatchRec per P2-mer (genie_internal.h): base / lmask / flags / row count / first row and the 16-base
    continuations of its suffixes -- in the entry, or chained through overflow entries -- against a direct
    enumeration of the reference's substrings."""
    rng = np.random.default_rng(77)
    if case == "syn10k":
        ref = rng.integers(0, 4, 10_000).astype(np.uint8)
    elif case == "tail_A":                       # the reference ends in A's: cut-short suffixes look like padding
        ref = np.concatenate([rng.integers(0, 4, 3000), np.zeros(40, np.int64)]).astype(np.uint8)
    elif case == "tiny":
        ref = rng.integers(0, 4, 13).astype(np.uint8)
    else:                                        # tandem repeats: P2-mers with 10 and with 60 suffixes
        ref = np.concatenate([np.tile([0, 1, 2, 3, 3, 1], 60), rng.integers(0, 4, 1000), np.tile([2, 2, 0, 1, 3, 0, 1, 1, 2, 3], 11),
                              rng.integers(0, 4, 1000)]).astype(np.uint8)
    n = len(ref)
    ix = pkg.GenieIndex.build(ref, 6, table_format="wide")
    img = ix.serialize().numpy()
    h = _parse(img)
    P2 = h["P2"]
    assert h["dir2_entries"] == 4 ** P2 <= h["mtab_entries"] and h["flags"] & 2 == 0 and h["ov_entries"] == 0
    mt = _match_table(img, h)
    raw = np.frombuffer(bytes(img[h["off_mtab"]:h["off_mtab"] + 32 * h["mtab_entries"]]), "<u4").reshape(-1, 8)
    base, lmask = mt["meta"] & 0xFF, (mt["meta"] >> 8) & 0xFF
    slow, more, rows = (mt["meta"] >> 16) & 1, (mt["meta"] >> 17) & 1, mt["meta"] >> 24
    s = "".join("ACGT"[c] for c in ref)
    occ = [set()] + [{s[i:i + t] for i in range(n - t + 1)} for t in range(1, P2 + 1)]
    sa0 = ix.suffix_array().astype(np.int64) - 1
    by_code, first_row = {}, {}
    for row, st in enumerate(sa0):               # suffix-array order
        if n - st >= P2:
            by_code.setdefault(s[st:st + P2], []).append(int(st))
            first_row.setdefault(s[st:st + P2], row)
    code_of = lambda t: int("".join(str("ACGT".index(c)) for c in t), 4)        # noqa: E731

    def key_of(st):
        k = 0
        for j in range(16):
            k |= (int(ref[st + P2 + j]) if st + P2 + j < n else 0) << (30 - 2 * j)
        return k

    seen, used, chained = 0, 4 ** P2, 0
    for mer, starts in by_code.items():
        c = code_of(mer)
        seen += 1
        cut = any(n - st < P2 + 16 for st in starts)
        chain = not cut and 6 < len(starts) <= 29
        assert base[c] == P2 and rows[c] == min(len(starts), 255) and mt["lb"][c] == first_row[mer]
        assert slow[c] == int(len(starts) > 6 or cut) and more[c] == int(chain)
        assert lmask[c] == (0 if slow[c] else 0x1F)
        keys = [key_of(st) for st in starts]
        if chain:
            chained += 1
            assert mt["key"][c][:5].tolist() == keys[:5]
            o = int(mt["key"][c][5])
            extra = (len(starts) - 5 + 7) // 8
            assert used <= o and o + extra <= h["mtab_entries"]
            got = raw[o:o + extra].reshape(-1).tolist()
            assert got[:len(keys) - 5] == keys[5:] and all(g == keys[0] for g in got[len(keys) - 5:]), mer
            assert keys == sorted(keys)
        else:
            k6 = keys[:6] + [keys[0]] * max(0, 6 - len(keys))
            assert mt["key"][c].tolist() == k6, mer
            assert slow[c] or keys == sorted(keys)      # ascending: suffix-array order
    if case == "repeat":
        assert chained >= 5 and (slow & (1 - more) & (rows > 29)).any()
    absent = np.nonzero(base[:4 ** P2] < P2)[0]
    assert len(absent) == 4 ** P2 - seen
    for c in (absent if len(absent) < 3000 else rng.choice(absent, 3000, replace=False)):
        mer = "".join("ACGT"[(int(c) >> (2 * (P2 - 1 - j))) & 3] for j in range(P2))
        t = max([t for t in range(1, P2) if mer[:t] in occ[t]], default=0)
        assert base[c] == t and slow[c] == 0 and rows[c] == 0, mer
        if t:                                    # the rows of the longest occurring prefix (the interval kernel's answer)
            want = [row for row, st in enumerate(sa0) if s[st:st + t] == mer[:t]]
            assert (int(mt["lb"][c]), int(mt["key"][c][0])) == (want[0], want[-1]) and len(want) == want[-1] - want[0] + 1, mer


@pytest.mark.parametrize("case", ["syn10k", "tail_A", "tiny", "repeat"])
def test_compact_match_table_against_brute_force(pkg, case):
    """MatchRec16 per P2-mer + its overflow blocks (genie_internal.h, the form for tables that do not fit an XCD's L2):
    first row, suffix count, the 8-base continuations (inline up to six; 7 .. 13: five inline + a block of eight), the
    rows-decide flag (a cut-short suffix, or more than 13) and, for an absent P2-mer, its longest occurring prefix and that
    prefix's rows -- against a direct enumeration."""
    rng = np.random.default_rng(77)
    if case == "syn10k":
        ref = rng.integers(0, 4, 10_000).astype(np.uint8)
    elif case == "tail_A":
        ref = np.concatenate([rng.integers(0, 4, 3000), np.zeros(40, np.int64)]).astype(np.uint8)
    elif case == "tiny":
        ref = rng.integers(0, 4, 13).astype(np.uint8)
    else:
        ref = np.concatenate([np.tile([0, 1, 2, 3, 3, 1], 60), rng.integers(0, 4, 1000), np.tile([2, 2, 0, 1, 3, 0, 1, 1, 2, 3], 11),
                              rng.integers(0, 4, 1000)]).astype(np.uint8)
    n = len(ref)
    ix = pkg.GenieIndex.build(ref, 6, table_format="compact")
    img = ix.serialize().numpy()
    h = _parse(img)
    P2 = h["P2"]
    assert h["flags"] & 2 and h["mtab_entries"] == h["dir2_entries"] == 4 ** P2 and 1 <= h["ov_entries"] <= 65536
    raw = np.frombuffer(bytes(img[h["off_mtab"]:h["off_mtab"] + 16 * h["mtab_entries"]]), np.dtype([("w0", "<u4"), ("key", "<u2", (6,))]))
    ov = np.frombuffer(bytes(img[h["off_ov"]:h["off_ov"] + 16 * h["ov_entries"]]), "<u2").reshape(-1, 8)
    lb, cnt4, nib = raw["w0"] & 0xFFFFFF, (raw["w0"] >> 24) & 15, raw["w0"] >> 28
    s = "".join("ACGT"[c] for c in ref)
    occ = [set()] + [{s[i:i + t] for i in range(n - t + 1)} for t in range(1, P2 + 1)]
    sa0 = ix.suffix_array().astype(np.int64) - 1

    def key_of(st, nb=8):
        k = 0
        for j in range(nb):
            k |= (int(ref[st + P2 + j]) if st + P2 + j < n else 0) << (2 * (nb - 1 - j))
        return k

    by_code, first_row = {}, {}
    for row, st in enumerate(sa0):
        if n - st >= P2:
            by_code.setdefault(s[st:st + P2], []).append(int(st))
            first_row.setdefault(s[st:st + P2], row)
    code_of = lambda t: int("".join(str("ACGT".index(c)) for c in t), 4)        # noqa: E731
    blocks, many, wide = set(), 0, 0
    for mer, starts in by_code.items():
        c = code_of(mer)
        keys = [key_of(st) for st in starts]
        k = len(starts)
        cut = any(n - st < P2 + 8 for st in starts)
        assert lb[c] == first_row[mer]
        assert cut or keys == sorted(keys)
        if k <= 3 and not any(n - st < P2 + 16 for st in starts):            # wide keys: three 16-base keys in the twelve bytes
            wide += 1
            k32 = [key_of(st, 16) for st in starts]
            assert cnt4[c] == k and nib[c] == 4 and raw["key"][c].view("<u4").tolist() == k32 + [k32[0]] * (3 - k), mer
        elif 7 <= k <= 13 and not cut:
            assert cnt4[c] == 7 and nib[c] == k - 7 and raw["key"][c][:5].tolist() == keys[:5], mer
            b = int(raw["key"][c][5])
            assert 1 <= b < h["ov_entries"] and b not in blocks
            blocks.add(b)
            assert ov[b].tolist() == keys[5:] + [keys[5]] * (13 - k), mer
        else:
            many += k > 13
            assert cnt4[c] == min(k, 6) and nib[c] == (8 if cut or k > 6 else 0), mer
            assert raw["key"][c].tolist() == ((keys + [keys[0]] * 6)[:6] if k <= 6 else keys[:6]), mer
    assert wide > 0 or case == "tiny"                           # (13 bases: every suffix is cut short)
    assert len(blocks) == h["ov_entries"] - 1
    if case == "repeat":
        assert len(blocks) >= 5 and many >= 1
    absent = np.nonzero(cnt4 == 0)[0]
    assert len(absent) == 4 ** P2 - len(by_code)
    for c in (absent if len(absent) < 3000 else rng.choice(absent, 3000, replace=False)):
        mer = "".join("ACGT"[(int(c) >> (2 * (P2 - 1 - j))) & 3] for j in range(P2))
        t = max([t for t in range(1, P2) if mer[:t] in occ[t]], default=0)
        assert nib[c] == t, mer
        if t:
            want = [row for row, st in enumerate(sa0) if s[st:st + t] == mer[:t]]
            last = int(raw["key"][c][0]) | int(raw["key"][c][1]) << 16
            assert (int(lb[c]), last) == (want[0], want[-1]) and len(want) == want[-1] - want[0] + 1, mer
    # compact is the default form
    assert _parse(pkg.GenieIndex.build(ref, 6).serialize().numpy())["flags"] & 2


def test_corrupt_image_is_rejected(pkg):
    """genie_index_open validates every section of the image header: a truncated or corrupt image must
    come back as GENIE_E_BAD_BLOB, not as device pointers outside the allocation.  (Host-side check:
    the device pointer is never dereferenced by open.)"""
    import ctypes as C
    lib = pkg._native.lib()
    ref = np.random.default_rng(5).integers(0, 4, 3000).astype(np.uint8)
    ix = pkg.GenieIndex.build(ref, 6)
    ix.train_rmi([10])
    img = ix.serialize().numpy().copy()
    h = _parse(img)
    fake_dev = C.c_void_p(0x7f0000000000)                         # 16-byte aligned, never touched

    def try_open(buf, nbytes=None):
        out = C.c_void_p(None)
        rc = lib.genie_index_open(buf.ctypes.data_as(C.c_void_p), fake_dev, len(buf) if nbytes is None else nbytes, 0,
                                  C.byref(out))
        if rc == 0:
            lib.genie_index_destroy(out)
        return rc

    assert try_open(img) == 0
    assert try_open(img, h["total_bytes"] - 256) == -8            # truncated
    fields = {name: HDR.unpack(bytes(img[:HDR.size])) for name in ["ok"]}["ok"]
    names = ["magic", "version", "header_bytes", "total_bytes", "n", "K", "P", "off_sa", "off_ref", "off_dir",
             "off_lut", "off_rmi", "ref_recs", "dir_entries", "lut_slots", "lut_keys", "rmi_models", "nlev"]
    pos = {k: i for i, k in enumerate(names)}
    tail = {"off_dir2": -10, "dir2_entries": -9, "P2": -8, "flags": -7, "off_rmi_err": -6, "rmi_err_entries": -5, "off_mtab": -4,
            "mtab_entries": -3, "off_ov": -2, "ov_entries": -1}

    def corrupt(**kw):
        f = list(fields)
        for k, v in kw.items():
            f[pos[k] if k in pos else len(f) + tail[k]] = v
        buf = img.copy()
        buf[:HDR.size] = np.frombuffer(HDR.pack(*f), np.uint8)
        return try_open(buf)

    big = h["total_bytes"]
    for bad in (dict(off_sa=big), dict(off_ref=big - 16), dict(off_dir=big + 4096), dict(off_lut=-256),
                dict(off_rmi=big), dict(off_dir2=big - 64), dict(off_mtab=big - 64), dict(off_rmi_err=big),
                dict(off_sa=h["off_sa"] + 4),                     # misaligned
                dict(lut_slots=0), dict(lut_slots=h["lut_keys"]), dict(lut_slots=1 << 40), dict(n=h["n"] + 10 ** 7),
                dict(K=17), dict(K=-1), dict(nlev=5), dict(nlev=-1), dict(rmi_models=h["rmi_models"] + 1),
                dict(rmi_err_entries=h["rmi_err_entries"] + 1), dict(P2=h["P"]), dict(P2=13),
                dict(mtab_entries=h["dir2_entries"] - 1), dict(mtab_entries=1 << 27), dict(dir2_entries=4), dict(ref_recs=1), dict(version=6)):
        assert corrupt(**bad) == -8, bad


def test_image_without_seed_table(pkg):
    """GENIE_IMAGE_NO_SEED_TABLE: the same image minus the K-mer hash table; it opens, says so in its flags, and every
    other section is byte-identical."""
    import ctypes as C
    ref = np.random.default_rng(3).integers(0, 4, 20_000).astype(np.uint8)
    ix = pkg.GenieIndex.build(ref, 12)
    ix.train_rmi([10])
    full, slim = ix.serialize().numpy(), ix.serialize(seed_table=False).numpy()
    hf, hs = _parse(full), _parse(slim)
    assert hs["lut_slots"] == 8 and hf["lut_slots"] > 2 * hf["lut_keys"] > 16 and hs["lut_keys"] == hf["lut_keys"]
    assert hs["flags"] == hf["flags"] | 4 and len(slim) < len(full) - 16 * (hf["lut_slots"] - 8) + 512
    for name, size in (("off_sa", 16 * (len(ref) + 1)), ("off_dir", 4 * hf["dir_entries"]), ("off_dir2", 16 * hf["dir2_entries"]),
                       ("off_mtab", 16 * hf["mtab_entries"]), ("off_ov", 16 * hf["ov_entries"]), ("off_rmi", 16 * hf["rmi_models"])):
        assert bytes(full[hf[name]:hf[name] + size]) == bytes(slim[hs[name]:hs[name] + size]), name
    out = C.c_void_p(None)
    lib = pkg._native.lib()
    assert lib.genie_index_open(slim.ctypes.data_as(C.c_void_p), C.c_void_p(0x7f0000000000), len(slim), 0, C.byref(out)) == 0
    lib.genie_index_destroy(out)
    assert lib.genie_index_image_bytes(ix._h, 2) == -1 and lib.genie_index_image_bytes(ix._h, 1) == len(slim)

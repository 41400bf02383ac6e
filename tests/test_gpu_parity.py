"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against (a) the golden vectors produced by the unmodified reference and (b) the CPU oracle on
fresh seeded inputs.  Integer work: every comparison is bit-exact.
"""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import genie_smem_amd as g
    g._native.lib()
    return g


_IX = {}


def _index(pkg, ds, with_rmi=None):
    """GenieIndex for a golden dataset (optionally with the fixture's RMI coefficients)."""
    key = (ds, with_rmi)
    if key not in _IX:
        d, _ = G.load(ds)
        ix = pkg.GenieIndex.build(d["ref_codes"], int(d["K"]))
        if with_rmi is not None:
            ex = [int(x) for x in d[f"{with_rmi}.experts"]]
            ix.set_rmi(ex, [d[f"{with_rmi}.coef{l}"] for l in range(len(ex) + 1)],
                       [d[f"{with_rmi}.icpt{l}"] for l in range(len(ex) + 1)])
        _IX[key] = ix.to("cuda")
    return _IX[key]


def _rmi_tag(ds):
    d, _ = G.load(ds)
    tag = "g4_" + "_".join(str(int(x)) for x in d["experts"])
    return tag if f"{tag}.coef0" in d else None


def _index_for(pkg, ds, algo):
    if algo != "rmi":
        return _index(pkg, ds)
    tag = _rmi_tag(ds)
    if tag:
        return _index(pkg, ds, tag)
    key = (ds, "trained")
    if key not in _IX:                       # no reference coefficients in the fixture: fit our own
        d, _ = G.load(ds)
        m = pkg.ExactMatch(ds + ".fa")
        m.set_reference(G.codes_to_str(d["ref_codes"]))
        r = pkg.RMI_LUT([int(x) for x in d["experts"]], int(d["K"]), ds + ".fa", matcher=m)
        r.train_RMI()
        _IX[key] = r._index()
    return _IX[key]


def _rows_per_read(offsets, smems):
    off = offsets.cpu().numpy()
    sm = smems.cpu().numpy()
    return [sm[off[i]:off[i + 1]] for i in range(len(off) - 1)]


# ------------------------------------------------------------------ A1: exact_match_back_prop
@pytest.mark.parametrize("ds", [d for d in G.DATASETS if G.have(d)])
def test_sa_interval_golden(pkg, ds):
    d, _ = G.load(ds)
    ix = _index(pkg, ds)
    off, pat, want = d["g5.pat_off"], d["g5.pat"], d["g5.lohi"]
    lens = np.diff(off).astype(np.int32)
    mat = np.zeros((len(lens), int(lens.max())), np.uint8)
    for i in range(len(lens)):
        mat[i, :lens[i]] = pat[off[i]:off[i + 1]]
    got = ix.sa_interval(mat, lens).cpu().numpy()
    assert (got == want).all(), np.nonzero((got != want).any(1))[0][:10]


def test_sa_interval_edges(pkg):
    ix = _index(pkg, "medium_K6")
    n = ix.n
    mat = np.zeros((3, 4), np.uint8)
    mat[1] = [0, 1, 7, 0]
    got = ix.sa_interval(mat, np.asarray([0, 4, 1], np.int32)).cpu().numpy()
    assert got[0].tolist() == [0, n]                    # "" -> (0, n) like the reference
    assert got[1].tolist() == [-2, -2]                  # code > 3 (reference: KeyError)
    assert got[2, 0] >= 1


def test_dropin_exact_match_mississippi(pkg):
    k = G.known()["mississippi"]
    m = pkg.ExactMatch("mississippi.fa")
    m.set_reference(k["ref"])
    for q, want in k["back_prop"].items():
        got = m.exact_match_back_prop(q)
        assert got == (-1 if want == -1 else tuple(want)), q
    for q, pos in k["exact_match"].items():
        assert m.exact_match(q) == pos
    with pytest.raises(KeyError):
        m.exact_match_back_prop("mixs")                 # 'x' not in count_dic (ExactMatch.py:139)
    with pytest.raises(TypeError):
        m.exact_match("mm")                             # reference unpacks -1 (ExactMatch.py:181)
    # one backward-search step == searching the longer string
    iv = m.exact_match_back_prop("ssi")
    assert m.exact_match_back_prop_add_one("i", iv) == m.exact_match_back_prop("issi")
    assert m.exact_match_back_prop_add_one("p", iv) == -1
    assert m.get_positions(*m.exact_match_back_prop("ssi")) == [6, 3] and m.get_position(0) == 12


# ------------------------------------------------------------------ A2 / A5 / A9: traversals vs goldens
@pytest.mark.parametrize("ds,tag,algo", G.group_cases())
def test_traversal_golden(pkg, ds, tag, algo):
    ix = _index_for(pkg, ds, algo)
    rd = G.reads(ds, tag)
    items = G.ref_items(ds, tag, algo)
    trace = G.ref_trace(ds, tag, algo)
    status = G.ref_status(ds, tag, algo)
    lut_items = G.ref_items(ds, tag, "lut") if algo == "rmi" else None
    offsets, smems, st = ix.find_smems(algo, rd[:len(status)], min_len=1)
    assert (st.cpu().numpy() == 0).all()
    rows = _rows_per_read(offsets, smems)
    checked = 0
    for r in range(len(status)):
        if algo == "rmi":
            assert G.dict_view(rd[r], rows[r]) == lut_items[r], r      # always-on gate
            if status[r] != 0 or items[r] != lut_items[r]:
                continue                                               # tagged reference defect
        assert G.dict_view(rd[r], rows[r]) == items[r], r
        if algo == "bwa":
            assert rows[r].tolist() == trace[r].tolist(), r
        else:
            assert rows[r][:, :2].tolist() == trace[r][:len(rows[r])].tolist(), r
        checked += 1
    assert checked >= (len(status) * 3) // 4


def test_dropin_smem_api(pkg):
    k = G.known()
    m = pkg.ExactMatch("mississippi.fa")
    m.set_reference(k["mississippi"]["ref"])
    s = pkg.SMEM(m, lut_size=3)
    for q, want in k["mississippi"]["get_SMEMS"].items():
        assert [(a, b[0], b[1]) for a, b in s.get_SMEMS(q, 1).items()] == [tuple(w) for w in want], q
    for q, want in k["mississippi"]["get_SMEMS_min3"].items():
        assert [(a, b[0], b[1]) for a, b in s.get_SMEMS(q, 3).items()] == [tuple(w) for w in want], q
    with pytest.raises(KeyError):
        s.get_SMEMS("aaaaa", 1)                           # aaaaa -> KeyError('a')
    # single-step helpers keep the reference's shapes
    assert s.get_SMEM_at_index("pissssi", 1) == ["iss", (3, 4), 4]
    fm, longest = s.forward_extension("missippi", 0)
    assert longest == "missi" and list(fm) == ["m", "mi", "mis", "miss", "missi"]
    assert s.check_sequential([5, 2], [6, 3]) and not s.check_sequential([5], [7])

    m2 = pkg.ExactMatch("paperex.fa")
    m2.set_reference(k["paperex"]["ref"])
    s2 = pkg.SMEM(m2, lut_size=2)
    assert {kk: [list(v[0]), v[1]] for kk, v in s2.lut.lut.items()} == k["paperex"]["lut"]
    for q, want in k["paperex"]["get_SMEMS"].items():
        assert [(a, b[0], b[1]) for a, b in s2.get_SMEMS(q, 1).items()] == [tuple(w) for w in want], q
    for q, want in k["paperex"]["get_smems_lut"].items():
        assert [(a, b[0], b[1]) for a, b in s2.get_smems_lut(q).items()] == [tuple(w) for w in want], q
    with pytest.raises(ValueError):
        s2.get_smems_lut("A")                             # shorter than K


def test_score_lut_driver(pkg, tmp_path):
    """The reference's benchmark driver (SMEM.py:508-539) on the drop-in classes, paper-sized queries."""
    import random
    from genie_smem_amd.score import score_LUT
    d, _ = G.load("syn10k_K8")
    data = tmp_path / "data"
    data.mkdir()
    with open(data / "syn10k.fa", "w") as fh:
        fh.write(">syn\n" + G.codes_to_str(d["ref_codes"]) + "\n")
    random.seed(3)
    res = score_LUT(4, "syn10k.fa", query_size=600, data_dir=str(data), lut_size=8, batched=True)
    assert len(res) == 6 and all(t > 0 for t in res)


# ------------------------------------------------------------------ A6 / A8: seeds
def test_lut_seed_lookup(pkg):
    ds = "syn100k_K15"
    d, _ = G.load(ds)
    ix = _index(pkg, ds)
    K = int(d["K"])
    codes, lo, hi = ix.lut_arrays()
    sel = np.random.default_rng(5).choice(len(codes), 20000, replace=False)
    kmers = ((codes[sel, None].astype(np.int64) >> (2 * (K - 1 - np.arange(K)))) & 3).astype(np.uint8)
    got = ix.seed_lookup("lut", kmers).cpu().numpy()
    assert (got[:, 0] == lo[sel]).all() and (got[:, 1] == hi[sel]).all()
    absent = np.random.default_rng(6).integers(0, 4, (20000, K)).astype(np.uint8)
    got = ix.seed_lookup("lut", absent).cpu().numpy()
    w = 4 ** np.arange(K - 1, -1, -1, dtype=np.int64)
    present = np.isin((absent.astype(np.int64) * w).sum(1), codes.astype(np.int64))
    assert ((got[:, 0] >= 0) == present).all()


@pytest.mark.parametrize("ds", [d for d in ("syn100k_K15", "big100k_K15") if G.have(d)])
def test_rmi_predict_bit_exact_and_true_interval(pkg, ds):
    d, _ = G.load(ds)
    for tag in sorted({k.split(".")[0] for k in d if k.startswith("g4_")}):
        ix = _index(pkg, ds, tag)
        lohi, pred = ix.seed_lookup("rmi", d[f"{tag}.kmers"], want_pred=True)
        lohi, pred = lohi.cpu().numpy(), pred.cpu().numpy()
        assert (pred == d[f"{tag}.pred"]).all(), tag                 # float64, bit for bit
        truth = d[f"{tag}.truth"]
        absent = truth[:, 0] < 0
        assert (lohi[~absent] == truth[~absent]).all(), tag
        assert (lohi[absent, 0] > lohi[absent, 1]).all(), tag        # reference convention lower > upper
        # the reference itself agrees wherever it is not tagged as defective
        ok = d[f"{tag}.defect"] == 0
        ref = d[f"{tag}.lohi"]
        assert (lohi[ok & ~absent] == ref[ok & ~absent]).all()


def test_native_rmi_fenced_lookup(pkg):
    """A natively trained RMI carries per-leaf error bounds and genie_seed_lookup fences its last-mile
    search with them: every K-mer of the reference gets its true rows, absent K-mers keep lower > upper
    and the same values as the unfenced search."""
    d, _ = G.load("big100k_K15")
    ref = d["ref_codes"]
    ix = pkg.GenieIndex.build(ref, 15)
    ix.train_rmi([1000])
    codes, lo, hi = ix.lut_arrays()
    ix = ix.to("cuda")
    kmers = ((codes[:, None].astype(np.int64) >> (2 * np.arange(14, -1, -1))) & 3).astype(np.uint8)
    got = ix.seed_lookup("rmi", kmers).cpu().numpy()
    assert (got[:, 0] == lo).all() and (got[:, 1] == hi).all()
    rng = np.random.default_rng(9)
    rnd = rng.integers(0, 4, (20000, 15)).astype(np.uint8)
    plain = _index(pkg, "big100k_K15", _rmi_tag("big100k_K15"))          # fixture coefficients: no error table
    a, b = ix.seed_lookup("rmi", rnd).cpu().numpy(), plain.seed_lookup("rmi", rnd).cpu().numpy()
    assert (a == b).all() and (a[:, 0] > a[:, 1]).sum() > 19000


def test_dropin_rmi_lut(pkg):
    d, _ = G.load("syn10k_K8")
    m = pkg.ExactMatch("syn10k.fa")
    m.set_reference(G.codes_to_str(d["ref_codes"]))
    r = pkg.RMI_LUT([10, 100], 8, "syn10k.fa", matcher=m)
    r.train_RMI()
    q = G.codes_to_str(d["ref_codes"][500:508])
    assert r.get_suffix_rmi(q) == m.exact_match_back_prop(q)
    lo, hi = r.get_suffix_rmi("ACGTACGT" if m.exact_match_back_prop("ACGTACGT") == -1 else "TTTTTTTT")
    s = pkg.SMEM(m, lut_size=8)
    s.rmi_lut = r
    rd = G.reads("syn10k_K8", "fromref150")[:20]
    for row in rd:
        q = G.codes_to_str(row)
        assert s.get_smems_rmi(q) == s.get_smems_lut(q) == s.get_SMEMS(q, 1)


# ------------------------------------------------------------------ vs the CPU oracle on fresh inputs
@pytest.mark.parametrize("algo", ["bwa", "lut", "rmi"])
@pytest.mark.parametrize("ds,L,kind", [("syn100k_K15", 150, "fromref"), ("syn100k_K15", 150, "random"),
                                       ("big100k_K15", 150, "fromref"), ("syn10k_K8", 97, "fromref"),
                                       ("medium_K6", 64, "random"), ("syn100k_K15", 777, "fromref")])
def test_vs_oracle_fresh_reads(pkg, oracle_mod, ds, L, kind, algo):
    from genie_smem_amd import synth as B
    d, _ = G.load(ds)
    ix = _index_for(pkg, ds, algo)
    o = oracle_mod.Oracle(d["ref_codes"], int(d["K"]))
    if algo == "rmi":
        o.set_rmi([], [np.asarray([o.n / 4.0 ** o.K])], [np.asarray([0.0])])   # true-interval contract: any model
    n_reads = 4000 if L <= 150 else 300
    rd = B.reads_from_ref(d["ref_codes"], n_reads, L, 4242) if kind == "fromref" else B.reads_random(n_reads, L, 4243)
    offsets, smems, st = ix.find_smems(algo, rd)
    assert (st.cpu().numpy() == 0).all()
    rows = _rows_per_read(offsets, smems)
    counts, out = o.find_smems_batch(algo, rd, nthreads=8)
    assert (counts >= 0).all()
    for r in range(n_reads):
        assert rows[r].tolist() == out[r, :counts[r]].tolist(), r


@pytest.mark.parametrize("search_all", [0, 1])
@pytest.mark.parametrize("algo", ["bwa", "lut", "rmi"])
def test_fixed_length_slot_layouts(pkg, oracle_mod, algo, search_all):
    """Fixed-length batches of every length around the group / quad / 64-position boundaries of the search
    kernel, odd batch sizes (a last, partly filled group) and flagged reads inside groups; with the sampled
    lookup (default) and with every position looked up (GENIE_OPT_SEARCH_ALL)."""
    from genie_smem_amd import synth as B
    d, _ = G.load("syn100k_K15")
    ix = _index_for(pkg, "syn100k_K15", algo)
    ix.set_option(pkg._native.OPT_SEARCH_ALL, search_all)
    o = oracle_mod.Oracle(d["ref_codes"], 15)
    if algo == "rmi":
        o.set_rmi([], [np.asarray([o.n / 4.0 ** o.K])], [np.asarray([0.0])])
    try:
        _slot_layout_cases(pkg, ix, o, d, algo)
    finally:
        ix.set_option(pkg._native.OPT_SEARCH_ALL, 0)


def _slot_layout_cases(pkg, ix, o, d, algo):
    from genie_smem_amd import synth as B
    # (.. 24/25, 56/57, .. 248/249: where the packed read in the workspace gains a 16-byte piece)
    for L in (15, 20, 24, 25, 31, 32, 33, 56, 57, 64, 65, 70, 88, 89, 96, 97, 120, 121, 128, 129, 131, 134, 150, 152, 153, 160, 161,
              184, 185, 192, 193, 200, 216, 217, 224, 225, 248, 249, 255):
        n_reads = 301
        rd = B.reads_from_ref(d["ref_codes"], n_reads, L, 1000 + L)
        rd[7, L // 2] = 5                       # flagged read as the second of a pair
        rd[10, 0] = 4                           # ... and as the first
        rd[300, L - 1] = 7 if L % 2 else rd[300, L - 1]          # sometimes the partnerless last read too
        offsets, smems, st = ix.find_smems(algo, rd)
        st = st.cpu().numpy()
        rows = _rows_per_read(offsets, smems)
        bad = {7, 10} | ({300} if L % 2 else set())
        good = np.asarray([r for r in range(n_reads) if r not in bad])
        counts, out = o.find_smems_batch(algo, np.ascontiguousarray(rd[good]), nthreads=8)
        for r in bad:
            assert st[r] == pkg._native.READ_BAD_BASE and len(rows[r]) == 0, (L, r)
        for i, r in enumerate(good):
            assert st[r] == 0 and rows[r].tolist() == out[i, :counts[i]].tolist(), (L, int(r))


def test_stride_wider_than_fixed_length(pkg):
    """C ABI: fixed_len < stride with no lens array -- the columns past fixed_len are never read."""
    import ctypes as C
    import torch
    from genie_smem_amd import synth as B
    d, _ = G.load("syn100k_K15")
    ix = _index(pkg, "syn100k_K15")
    N, L, stride = 1001, 150, 176
    rd = B.reads_from_ref(d["ref_codes"], N, L, 321)
    wide = np.full((N, stride), 9, np.uint8)                   # 9 would flag the read if it were looked at
    wide[:, :L] = rd
    want = ix.find_smems("lut", rd)
    lib = pkg._native.lib()
    dev = torch.as_tensor(wide).cuda()
    status = torch.empty(N, dtype=torch.int32, device="cuda")
    offsets = torch.empty(N + 1, dtype=torch.int64, device="cuda")
    rows = torch.empty((int(want[0][-1]) + 8, 4), dtype=torch.int32, device="cuda")
    wsb = int(lib.genie_find_smems_workspace_bytes(N, L))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())                                       # noqa: E731
    pkg._native.check(lib.genie_find_smems_csr(ix._h, pkg._native.MODES["lut"], P(dev), None, N, stride, L, 1, P(offsets),
                                               P(rows), rows.shape[0], P(status), P(ws), wsb,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "genie_find_smems_csr")
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0 and torch.equal(offsets, want[0])
    assert torch.equal(rows[:int(offsets[-1])], want[1][:int(offsets[-1])])


@pytest.mark.parametrize("ds", ["syn100k_K15", "big100k_K15"])
def test_long_exact_matches_and_low_complexity(pkg, oracle_mod, ds):
    """Reads that ARE reference substrings (matches far longer than the 32-base inline keys, so the
    comparison continues in the packed reference), the same with one substitution, reads running off the
    reference end, and low-complexity reads -- against the CPU oracle, all three modes."""
    d, _ = G.load(ds)
    ref = d["ref_codes"]
    n = len(ref)
    rng = np.random.default_rng(17)
    o = oracle_mod.Oracle(ref, 15)
    o.set_rmi([], [np.asarray([o.n / 4.0 ** o.K])], [np.asarray([0.0])])
    for L in (150, 255, 600):
        starts = np.concatenate([rng.integers(0, n - L, 120), [0, n - L, n - L - 1], rng.integers(0, 6000, 40)])
        rd = np.stack([ref[s0:s0 + L] for s0 in starts]).astype(np.uint8)
        mut = rd[:60].copy()
        pos = rng.integers(0, L, 60)
        mut[np.arange(60), pos] = (mut[np.arange(60), pos] + 1 + rng.integers(0, 3, 60)) % 4
        tail = np.stack([np.concatenate([ref[n - t:], rng.integers(0, 4, L - t).astype(np.uint8)]) for t in (1, 5, 6, 7, 8, 40)])
        low = np.stack([np.full(L, b, np.uint8) for b in range(4)] +
                       [np.tile(np.asarray(p_, np.uint8), L // len(p_) + 1)[:L] for p_ in ([0, 1], [3, 0, 0, 1, 1, 1], [2, 3, 2])])
        batch = np.ascontiguousarray(np.concatenate([rd, mut, tail, low]))
        for algo in ("bwa", "lut", "rmi"):
            ix = _index_for(pkg, ds, algo)
            offsets, smems, st = ix.find_smems(algo, batch)
            rows = _rows_per_read(offsets, smems)
            counts, out = o.find_smems_batch(algo, batch, nthreads=8)
            st = st.cpu().numpy()
            for r in range(len(batch)):
                if counts[r] < 0:                       # e.g. a base that never occurs: flagged on both sides
                    assert st[r] != 0, (L, algo, r)
                else:
                    assert st[r] == 0 and rows[r].tolist() == out[r, :counts[r]].tolist(), (L, algo, r)


def test_first_run_start_whatever_the_bytes(pkg, oracle_mod):
    """The traversal kernel marks run starts of fwd[] by comparing bytes of its row with their left neighbours;
    position 0 has none.  Reads built so that fwd[0] = a and fwd[3] = 255 - a (a byte and its complement -- the
    pair a careless sentinel for 'the byte before position 0' would take for equal) and other first-dword patterns,
    against the CPU oracle."""
    rng = np.random.default_rng(99)
    n = 30_000
    ref = rng.integers(0, 4, n).astype(np.uint8)
    reads = []
    q = 15_000
    for a in (20, 40, 64, 100, 127):
        p0 = 1000 + 300 * len(reads)
        x = ref[p0:p0 + a].copy()
        y = rng.integers(0, 4, 255 - 2 * a).astype(np.uint8)
        y[0] = (ref[p0 + a] + 1) % 4                         # locus A stops matching after a bases
        ref[q - 1] = (ref[p0 + 2] + 1) % 4                   # locus B cannot be entered from position 2
        ref[q:q + a - 3] = x[3:]
        ref[q + a - 3:q + a - 3 + len(y)] = y
        rd = np.concatenate([x, y, rng.integers(0, 4, 255).astype(np.uint8)])[:255]
        rd[255 - a] = (ref[q + a - 3 + len(y)] + 1) % 4      # ... and stops after position 255 - a
        reads.append(rd)
        q += 400
    reads += [rng.integers(0, 4, 255).astype(np.uint8) for _ in range(59)]
    batch = np.ascontiguousarray(np.stack(reads))
    ix = pkg.GenieIndex.build(ref, 12).to("cuda")
    o = oracle_mod.Oracle(ref, 12)
    fwd0 = []
    for algo in ("bwa", "lut"):
        offsets, smems, st = ix.find_smems(algo, batch)
        rows = _rows_per_read(offsets, smems)
        counts, out = o.find_smems_batch(algo, batch, nthreads=8)
        assert (st.cpu().numpy() == 0).all() and (counts > 0).all()
        for r in range(len(batch)):
            assert rows[r].tolist() == out[r, :counts[r]].tolist(), (algo, r)
        fwd0 = [int(out[r, 0, 1]) for r in range(5)]
    assert fwd0 == [20, 40, 64, 100, 127]                     # the construction did what it says: fwd[0] = a


@pytest.mark.parametrize("ds", ["syn100k_K15", "big100k_K15"])
def test_long_reads_vs_oracle(pkg, oracle_mod, ds):
    """Reads of 256 .. 8192 bases (the 16-lanes-per-read traversal, windows of 704 positions in the match
    statistics): from-ref, random, reference substrings (one match spanning many 32-position passes), the same
    with substitutions, low complexity, ragged lengths and a BWA length filter -- against the CPU oracle."""
    d, _ = G.load(ds)
    ref = d["ref_codes"]
    n = len(ref)
    rng = np.random.default_rng(31)
    from genie_smem_amd import synth as B
    o = oracle_mod.Oracle(ref, 15)
    o.set_rmi([], [np.asarray([o.n / 4.0 ** o.K])], [np.asarray([0.0])])
    for L, nr in ((256, 40), (1000, 30), (4097, 12), (8192, 8)):
        exact = np.stack([ref[s0:s0 + L] for s0 in rng.integers(0, n - L, nr // 2)]).astype(np.uint8)
        mut = exact.copy()
        for row in mut:
            pos = rng.integers(0, L, max(1, L // 300))
            row[pos] = (row[pos] + 1 + rng.integers(0, 3, len(pos))) % 4
        low = np.stack([np.full(L, 1, np.uint8), np.tile(np.asarray([3, 0, 0, 1, 1, 1], np.uint8), L // 6 + 1)[:L]])
        batch = np.ascontiguousarray(np.concatenate([B.reads_from_ref(ref, nr, L, 40 + L), B.reads_random(nr // 2, L, 41 + L),
                                                     exact, mut, low]))
        lens = np.full(len(batch), L, np.int32)
        lens[::5] = rng.integers(0, L + 1, len(lens[::5]))
        for algo, min_len, use_lens in (("bwa", 1, False), ("bwa", 19, True), ("lut", 1, True), ("rmi", 1, False)):
            ix = _index_for(pkg, ds, algo)
            ll = lens if use_lens else None
            offsets, smems, st = ix.find_smems(algo, batch, lens=ll, min_len=min_len)
            rows = _rows_per_read(offsets, smems)
            counts, out = o.find_smems_batch(algo, batch, lens=ll, min_len=min_len, nthreads=8)
            st = st.cpu().numpy()
            for r in range(len(batch)):
                if counts[r] < 0:
                    assert st[r] != 0, (L, algo, r)
                else:
                    assert st[r] == 0 and rows[r].tolist() == out[r, :counts[r]].tolist(), (L, algo, min_len, r)
    # a batch large enough for the 8-lanes-per-read form of the traversal (the batches above use 16)
    big = np.ascontiguousarray(np.concatenate([B.reads_from_ref(ref, 50000, 270, 77), B.reads_random(16000, 270, 78)]))
    big[:40, :260] = np.stack([ref[s0:s0 + 260] for s0 in rng.integers(0, n - 260, 40)])
    lens = np.full(len(big), 270, np.int32)
    lens[::7] = rng.integers(0, 271, len(lens[::7]))
    for algo, min_len in (("bwa", 12), ("lut", 1)):
        offsets, smems, st = _index_for(pkg, ds, algo).find_smems(algo, big, lens=lens, min_len=min_len)
        counts, out = o.find_smems_batch(algo, big, lens=lens, min_len=min_len, nthreads=8)
        offs, flat, st = offsets.cpu().numpy(), smems.cpu().numpy(), st.cpu().numpy()
        ok = counts >= 0
        assert ((st == 0) == ok).all() and (np.diff(offs)[ok] == counts[ok]).all()
        want = np.concatenate([out[r, :counts[r]] for r in np.nonzero(ok)[0]])
        assert np.array_equal(flat[:offs[-1]], want)


@pytest.mark.parametrize("ds", ["syn100k_K15", "big100k_K15"])
def test_sampled_search_equals_full_search(pkg, oracle_mod, ds):
    """Default search (every 4th position looked up, the three between two of them only where their values
    differ) against GENIE_OPT_SEARCH_ALL (every position looked up) on from-ref, random, exact and ragged
    batches, all modes; one batch also against the CPU oracle so that both are pinned."""
    import torch
    from genie_smem_amd import synth as B
    d, _ = G.load(ds)
    ref = d["ref_codes"]
    n = len(ref)
    rng = np.random.default_rng(23)
    batches = []
    for L in (15, 33, 64, 100, 150, 187):
        rd = B.reads_from_ref(ref, 403, L, 500 + L)
        rd[5, L // 3] = 7                                            # one flagged read per batch
        batches.append((rd, None))
    batches.append((B.reads_random(1001, 150, 9), None))
    batches.append((np.stack([ref[s0:s0 + 150] for s0 in rng.integers(0, n - 150, 600)]).astype(np.uint8), None))
    rag = B.reads_from_ref(ref, 300, 180, 77)
    batches.append((rag, rng.integers(0, 181, 300).astype(np.int32)))
    for L in (256, 705, 1409, 3000):                                  # long reads: windows of 704 positions
        lr = B.reads_from_ref(ref, 41, L, 900 + L)
        lr[3, L - 2] = 6
        batches.append((lr, None))
    batches.append((np.stack([ref[s0:s0 + 2500] for s0 in rng.integers(0, n - 2500, 30)]).astype(np.uint8), None))
    lrag = B.reads_from_ref(ref, 60, 1500, 78)
    batches.append((lrag, rng.integers(0, 1501, 60).astype(np.int32)))
    for algo in ("bwa", "lut", "rmi"):
        ix = _index_for(pkg, ds, algo)
        for rd, lens in batches:
            a = ix.find_smems(algo, rd, lens=lens)
            ix.set_option(pkg._native.OPT_SEARCH_ALL, 1)
            try:
                b = ix.find_smems(algo, rd, lens=lens)
            finally:
                ix.set_option(pkg._native.OPT_SEARCH_ALL, 0)
            assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (algo, rd.shape)
    o = oracle_mod.Oracle(ref, 15)
    rd = batches[7][0][:200]
    counts, out = o.find_smems_batch("lut", rd, nthreads=8)
    rows = _rows_per_read(*_index_for(pkg, ds, "lut").find_smems("lut", rd)[:2])
    for r in range(len(rd)):
        assert rows[r].tolist() == out[r, :counts[r]].tolist()


def test_randomized_configurations(pkg, oracle_mod):
    """Seeded sweep over reference size and content (tiny, repeat-rich, a base missing), K, read length
    and read kind; BWA, LUT and natively trained RMI against the CPU oracle, default (sampled) search."""
    rng = np.random.default_rng(20260)
    cases = 0
    for n in (5, 37, 400, 3000, 70000):
        for rep in range(4 if n < 70000 else 2):
            alphabet = 4 if rng.random() < 0.7 else 3                     # sometimes T never occurs
            ref = rng.integers(0, alphabet, n).astype(np.uint8)
            if n >= 400 and rep % 2:                                        # splice in a tandem repeat
                unit = rng.integers(0, alphabet, int(rng.integers(1, 7))).astype(np.uint8)
                span = int(min(n // 2, 600))
                ref[:span] = np.tile(unit, span // len(unit) + 1)[:span]
            K = int(rng.choice([k for k in (3, 6, 8, 12, 15) if k <= n]))
            L = int(rng.integers(max(K, 1), 256))
            N = 61
            reads = np.empty((N, L), np.uint8)
            for r in range(N):
                kind = r % 3
                if kind == 0 or n < 3:
                    reads[r] = rng.integers(0, alphabet, L)
                else:
                    buf = []
                    while sum(len(b) for b in buf) < L:
                        p0 = int(rng.integers(0, n))
                        s0 = int(rng.integers(1, 31 if kind == 1 else 4 * L))
                        buf.append(ref[p0:p0 + s0])
                    reads[r] = np.concatenate(buf)[:L]
            o = oracle_mod.Oracle(ref, K)
            want = {}
            for fmt in ("wide", "compact"):                                 # both forms of the match table
                ix = pkg.GenieIndex.build(ref, K, table_format=fmt)
                coefs, icpts, _, _, _ = ix.train_rmi([10])
                ix = ix.to("cuda")
                o.set_rmi([10], coefs, icpts)
                for algo in ("bwa", "lut", "rmi"):
                    offsets, smems, st = ix.find_smems(algo, reads)
                    rows = _rows_per_read(offsets, smems)
                    if algo not in want:
                        want[algo] = o.find_smems_batch(algo, reads, nthreads=8)
                    counts, out = want[algo]
                    st = st.cpu().numpy()
                    for r in range(N):
                        if counts[r] < 0:
                            assert st[r] != 0, (n, K, L, algo, fmt, r)
                        else:
                            assert st[r] == 0 and rows[r].tolist() == out[r, :counts[r]].tolist(), (n, K, L, algo, fmt, r)
                    cases += 1
    assert cases == 108


def test_sampled_search_fuzz_lengths(pkg):
    """Differential sweep: sampled search against the search of every position over random read lengths
    (1..3000, so both the short-read groups and the long-read windows with every alignment of their
    boundaries), batch sizes and read kinds."""
    import torch
    d, _ = G.load("big100k_K15")
    ref = d["ref_codes"]
    n = len(ref)
    rng = np.random.default_rng(4711)
    ix = {a: _index_for(pkg, "big100k_K15", a) for a in ("bwa", "lut", "rmi")}
    for it in range(48):
        L = int(rng.choice([rng.integers(1, 256), rng.integers(256, 3001), 704 + rng.integers(-2, 3), 1408 + rng.integers(-2, 3)]))
        N = int(rng.integers(1, 120))
        rd = np.empty((N, L), np.uint8)
        for r in range(N):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                rd[r] = rng.integers(0, 4, L)
            elif kind == 1 and L <= n:
                s0 = int(rng.integers(0, n - L + 1))
                rd[r] = ref[s0:s0 + L]
                if L > 3 and rng.random() < 0.5:
                    rd[r, int(rng.integers(0, L))] ^= 1
            else:
                buf = []
                while sum(len(b) for b in buf) < L:
                    p0 = int(rng.integers(0, n))
                    buf.append(ref[p0:p0 + int(rng.integers(1, 200))])
                rd[r] = np.concatenate(buf)[:L]
        algo = ("bwa", "lut", "rmi")[it % 3]
        if algo != "bwa" and L < 15:
            algo = "bwa"
        h = ix[algo]
        a = h.find_smems(algo, rd)
        h.set_option(pkg._native.OPT_SEARCH_ALL, 1)
        try:
            b = h.find_smems(algo, rd)
        finally:
            h.set_option(pkg._native.OPT_SEARCH_ALL, 0)
        assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (it, algo, L, N)


def test_ragged_and_status(pkg, oracle_mod):
    d, _ = G.load("syn10k_K8")
    ix = _index(pkg, "syn10k_K8")
    o = oracle_mod.Oracle(d["ref_codes"], 8)
    from genie_smem_amd import synth as B
    rd = B.reads_from_ref(d["ref_codes"], 64, 120, 99)
    lens = np.random.default_rng(3).integers(0, 121, 64).astype(np.int32)
    lens[:4] = [0, 1, 7, 8]
    rd[5, 3] = 9                                                    # bad base inside the read
    lens[5] = 50
    for algo in ("bwa", "lut"):
        offsets, smems, st = ix.find_smems(algo, rd, lens=lens)
        st = st.cpu().numpy()
        rows = _rows_per_read(offsets, smems)
        for r in range(64):
            if r == 5:
                assert st[r] == pkg._native.READ_BAD_BASE and len(rows[r]) == 0
                continue
            n, want = o.find_smems(algo, rd[r, :lens[r]])
            if algo == "lut" and lens[r] < 8:
                assert st[r] == pkg._native.READ_TOO_SHORT and n == -3 and len(rows[r]) == 0
            else:
                assert st[r] == 0 and rows[r].tolist() == want.tolist(), (algo, r)
    # overflow: cap smaller than the SMEM count keeps the true count and flags the read
    counts, slots, st = ix.find_smems_slots("bwa", rd[8:9], cap=2)
    n, want = o.find_smems("bwa", rd[8])
    assert counts.item() == n and st.item() == pkg._native.READ_OVERFLOW
    assert slots.cpu().numpy()[0].tolist() == want[:2].tolist()


def test_absent_base_is_flagged(pkg):
    m = pkg.ExactMatch("x.fa")
    m.set_reference("ACACACCACAACCA")                              # no G, no T
    ix = m.index(2)
    rd = np.asarray([[0, 1, 0, 2, 0, 1]], np.uint8)                 # contains G
    for algo in ("bwa", "lut"):
        _, _, st = ix.find_smems(algo, rd)
        assert st.item() == pkg._native.READ_ABSENT_BASE
    with pytest.raises(KeyError):
        pkg.SMEM(m, lut_size=2).get_SMEMS("ACAGAC", 1)


# ------------------------------------------------------------------ multi-GPU plumbing on one GPU
def test_report_helpers(pkg):
    """genie_search_kernel_name / genie_launch_info: what bench.py and the profile summaries key on."""
    ix = _index(pkg, "syn10k_K8")
    assert ix.search_kernel_name("lut", 150) == "match_table_kernel<6, true, false>"      # <waves per SIMD, compact table, packed reads>
    assert ix.search_kernel_name("bwa", 2000) == "match_table_long_kernel<true>"
    info = ix.launch_info("lut", 150)
    assert info["block"] == 512 and 0 < info["lds_bytes"] <= 160 * 1024 and info["grid"] > 0
    # the stage switches are ignored unless the search kernel runs alone
    ix.set_option(pkg._native.OPT_SEARCH_STAGES_OFF, 63)
    from genie_smem_amd import synth as B
    d, _ = G.load("syn10k_K8")
    rd = B.reads_from_ref(d["ref_codes"], 200, 100, 5)
    a = ix.find_smems("lut", rd)
    ix.set_option(pkg._native.OPT_SEARCH_STAGES_OFF, 0)
    b = ix.find_smems("lut", rd)
    import torch
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_handle_opened_from_image(pkg):
    """What ranks != 0 do in the sharded run: open a device-resident copy of the serialized image
    (genie_index_open) and search with it -- same rows as the handle that built the index."""
    import torch
    from genie_smem_amd import synth as B
    d, _ = G.load("syn100k_K15")
    src = pkg.GenieIndex.build(d["ref_codes"], 15)
    src.train_rmi([1000])
    image = src.serialize()
    a = src.to("cuda")
    b = pkg.GenieIndex.from_image(image.clone().cuda())
    assert b.info()["has_host"] == 0 and b.info()["n"] == a.info()["n"]
    rd = B.reads_from_ref(d["ref_codes"], 3001, 150, 31)
    for algo in ("bwa", "lut", "rmi"):
        x, y = a.find_smems(algo, rd), b.find_smems(algo, rd)
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
    pats = rd[:500, :40]
    assert torch.equal(a.sa_interval(pats), b.sa_interval(pats))
    assert torch.equal(a.seed_lookup("rmi", rd[:500, :15]), b.seed_lookup("rmi", rd[:500, :15]))
    pa, pb = a.locate(x[1][:1000]), b.locate(y[1][:1000])
    assert torch.equal(pa[0], pb[0]) and torch.equal(pa[1], pb[1])


def test_rccl_broadcast_world1(pkg):
    """parallel.broadcast_index over the nccl (= RCCL) backend, world size 1 on this GPU."""
    import socket
    import torch
    import torch.distributed as dist
    from genie_smem_amd import parallel, synth as B
    if dist.is_initialized():
        pytest.skip("process group already initialised")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        d, _ = G.load("syn10k_K8")
        ix = pkg.GenieIndex.build(d["ref_codes"], 8)
        got = parallel.broadcast_index(ix, src=0, device="cuda:0")
        assert got is ix and ix.blob is not None and ix.blob.is_cuda
        buf = parallel.broadcast_image(ix.blob, src=0, device="cuda:0")
        assert torch.equal(buf, ix.blob)
        rd = B.reads_from_ref(d["ref_codes"], 64, 100, 5)
        assert int(ix.find_smems("lut", rd)[2].abs().sum()) == 0
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ rows -> coordinates
def test_locate_positions(pkg):
    """genie_locate == ExactMatch.get_positions (suffix-array entries of rows lo..hi, 1-based, row order)
    for SMEM rows, absent intervals, and intervals of thousands of rows (short patterns); the sorted
    form == ExactMatch.exact_match, pinned by the reference's mississippi answers."""
    import torch
    from genie_smem_amd import synth as B
    d, _ = G.load("big100k_K15")
    ix = _index(pkg, "big100k_K15")
    sa = ix.suffix_array().astype(np.int64)
    rd = B.reads_from_ref(d["ref_codes"], 500, 150, 11)
    offsets, smems, st = ix.find_smems("lut", rd)
    po, pos = ix.locate(smems)
    po, pos, sm = po.cpu().numpy(), pos.cpu().numpy(), smems.cpu().numpy()
    assert po[0] == 0 and po[-1] == len(pos) == int((sm[:, 3] - sm[:, 2] + 1).sum())
    for t in range(0, len(sm), 7):
        assert pos[po[t]:po[t + 1]].tolist() == sa[sm[t, 2]:sm[t, 3] + 1].tolist()
    # short patterns: wide intervals (whole-wave copies), the empty pattern, absent ones
    pats = np.zeros((40, 12), np.uint8)
    lens = np.asarray([0, 1, 1, 1, 1, 2, 2, 3, 3, 4] + [12] * 30, np.int32)
    rng = np.random.default_rng(5)
    pats[:] = rng.integers(0, 4, pats.shape)
    iv = ix.sa_interval(pats, lens)
    po, pos = ix.locate(iv)
    po, pos, ivh = po.cpu().numpy(), pos.cpu().numpy(), iv.cpu().numpy()
    assert ivh[0].tolist() == [0, len(d["ref_codes"])] and (ivh[10:, 0] == -1).any()
    for t in range(40):
        lo, hi = ivh[t]
        want = sa[lo:hi + 1].tolist() if lo >= 0 else []
        assert pos[po[t]:po[t + 1]].tolist() == want, t
    _, spos = ix.locate(iv, sort=True)
    spos = spos.cpu().numpy()
    for t in range(40):
        assert spos[po[t]:po[t + 1]].tolist() == sorted(pos[po[t]:po[t + 1]].tolist())
    assert ix.locate(torch.empty((0, 2), dtype=torch.int32))[0].tolist() == [0]
    # the reference's own answers
    k = G.known()["mississippi"]
    m = pkg.ExactMatch("mississippi.fa")
    m.set_reference(k["ref"])
    pats = sorted(k["exact_match"])
    got = m.exact_match_positions_batch(pats + ["ppp", "mississippi"])
    assert got[:len(pats)] == [k["exact_match"][p_] for p_ in pats] and got[-2] == [] and got[-1] == [1]
    assert all(m.exact_match(p_) == k["exact_match"][p_] for p_ in pats)


# ------------------------------------------------------------------ 1 Mb reference (BASELINE configs 3-4 shape)
def test_one_megabase_reference(pkg, oracle_mod):
    """REF_1M (seed 1 000 000), K = 15, natively trained RMI [1000]: the three modes against the CPU
    oracle on from-ref and random reads, plus exact-match patterns."""
    from genie_smem_amd import synth as B
    ref = B.synth_ref(1_000_000, 1_000_000)
    m = pkg.ExactMatch("REF_1M.fa")
    m.set_reference(G.codes_to_str(ref))
    r = pkg.RMI_LUT([1000], 15, "REF_1M.fa", matcher=m)
    r.train_RMI()
    ix = r._index()
    assert ix.search_kernel_name("rmi", 150) == "match_table_kernel<6, true, false>"      # 1 Mb: the compact table (4 MB of 16-byte entries)
    o = oracle_mod.Oracle(ref, 15)
    coefs, icpts = r.rmi.coefficients()
    o.set_rmi([1000], coefs, icpts)
    for kind, rd in (("fromref", B.reads_from_ref(ref, 3000, 150, 1004)), ("random", B.reads_random(1000, 150, 1005))):
        for algo in ("bwa", "lut", "rmi"):
            offsets, smems, st = ix.find_smems(algo, rd)
            assert int(st.abs().sum().item()) == 0
            rows = _rows_per_read(offsets, smems)
            counts, want = o.find_smems_batch(algo, rd, nthreads=16)
            for i in range(len(rd)):
                assert rows[i].tolist() == want[i, :counts[i]].tolist(), (kind, algo, i)
    # the 32-byte form at this size (8 MB table: the two-blocks-per-CU instance of the kernel)
    ixw = pkg.GenieIndex.build(ref, 15, table_format="wide")
    ixw.set_rmi([1000], coefs, icpts)
    ixw = ixw.to("cuda")
    assert ixw.search_kernel_name("rmi", 150) == "match_table_kernel<4, false, false>"
    rd = B.reads_from_ref(ref, 1500, 150, 1007)
    for algo in ("bwa", "rmi"):
        offsets, smems, st = ixw.find_smems(algo, rd)
        assert int(st.abs().sum().item()) == 0
        rows = _rows_per_read(offsets, smems)
        counts, want = o.find_smems_batch(algo, rd, nthreads=16)
        for i in range(len(rd)):
            assert rows[i].tolist() == want[i, :counts[i]].tolist(), ("wide", algo, i)
    del ixw
    rng = np.random.default_rng(9)
    pats = np.zeros((2000, 60), np.uint8)
    lens = rng.integers(1, 61, 2000).astype(np.int32)
    for i in range(2000):
        p0 = int(rng.integers(0, len(ref) - 60))
        pats[i, :lens[i]] = ref[p0:p0 + lens[i]]
        if i % 3 == 0:
            pats[i, lens[i] - 1] = (pats[i, lens[i] - 1] + 1) % 4
    got = ix.sa_interval(pats, lens).cpu().numpy()
    for i in range(2000):
        assert tuple(got[i]) == o.back_prop(pats[i, :lens[i]]), i


# ------------------------------------------------------------------ full-size properties
def _full_size_properties(ix, o, rd_np, L, n_ref):
    """Size-independent properties of a full-size batch: the three traversals agree bit for bit, repeated runs are
    identical, covers are monotone, every emitted substring re-searches to its own interval with the batched
    exact-match kernel (a different kernel) -- plus three 20 000-read slices (start, middle, end of the batch)
    compared row by row with the CPU oracle."""
    import torch
    n_reads = len(rd_np)
    rd = torch.as_tensor(rd_np).cuda()
    res = {}
    for rep in range(2):
        for algo in ("bwa", "lut", "rmi"):
            offsets, smems, st = ix.find_smems(algo, rd)
            assert int(st.abs().sum().item()) == 0
            if rep == 0:
                res[algo] = (offsets, smems)
            else:
                assert torch.equal(res[algo][0], offsets) and torch.equal(res[algo][1], smems), algo   # deterministic
                res[algo] = (offsets, smems) if algo == "lut" else None
            del offsets, smems
        if rep == 0:
            for algo in ("bwa", "rmi"):
                assert torch.equal(res[algo][0], res["lut"][0]) and torch.equal(res[algo][1], res["lut"][1]), algo
    offsets, smems = res["lut"]
    del res
    start, end, lo, hi = smems[:, 0], smems[:, 1], smems[:, 2], smems[:, 3]
    assert bool(((start >= 0) & (start < end) & (end <= L) & (lo >= 0) & (lo <= hi) & (hi <= n_ref)).all())
    # last SMEM of each read ends at the read end; ends strictly increase inside a read
    last = offsets[1:] - 1
    assert bool((end[last] == L).all())
    same_read = torch.ones(len(end) - 1, dtype=torch.bool, device=end.device)
    same_read[(offsets[1:-1] - 1)] = False
    assert bool((end[1:][same_read] > end[:-1][same_read]).all())
    del same_read, last
    # re-search emitted substrings with the batched exact-match kernel
    sel = torch.randint(0, len(start), (400_000,), device=start.device)
    read_of = torch.searchsorted(offsets, sel, right=True) - 1
    lens = (end[sel] - start[sel]).to(torch.int32)
    idx = start[sel].long()[:, None] + torch.arange(L, device=rd.device)[None, :]
    pats = torch.gather(rd[read_of], 1, idx.clamp(max=L - 1))
    got = ix.sa_interval(pats, lens)
    assert torch.equal(got[:, 0], lo[sel]) and torch.equal(got[:, 1], hi[sel])
    # oracle, row by row, on slices taken from the start, the middle and the end of the batch
    off = offsets.cpu().numpy()
    for a in (0, (n_reads // 2) // 20_000 * 20_000, n_reads - 20_000):
        b = a + 20_000
        sm = smems[off[a]:off[b]].cpu().numpy()
        counts, want = o.find_smems_batch("lut", rd_np[a:b], nthreads=16)
        assert (np.diff(off[a:b + 1]) == counts).all()
        rows = np.repeat(np.arange(b - a), counts)
        tpos = np.arange(len(rows)) - np.repeat(off[a:b] - off[a], counts)
        assert (want[rows, tpos] == sm).all()


def test_full_size_properties(pkg, oracle_mod):
    """1M x 150 bp reads on the 100 kb reference (BASELINE config 1/2 shape): see _full_size_properties.
    (This test is the guard that caught wrong intervals from an earlier kernel structure that only misbehaved
    at full occupancy; keep it at full size.)"""
    import os
    from genie_smem_amd import synth as B
    d, _ = G.load("syn100k_K15")
    ix = _index_for(pkg, "syn100k_K15", "rmi")
    n_reads = 1_000_000
    seed = int(os.environ.get("GENIE_TEST_SEED", "1002"))            # vary to soak-test the full-occupancy guard
    rd_np = B.reads_from_ref_fast(d["ref_codes"], n_reads, 150, seed)
    if seed % 2:                                                     # odd seeds: a third of the batch uniform-random
        rd_np[::3] = B.reads_random((n_reads + 2) // 3, 150, seed + 1)
    _full_size_properties(ix, oracle_mod.Oracle(d["ref_codes"], int(d["K"])), rd_np, 150, ix.n)


def _index_1mb(pkg, oracle_mod):
    from genie_smem_amd import synth as B
    ref = B.synth_ref(1_000_000, 1_000_000)
    m = pkg.ExactMatch("REF_1M.fa")
    m.set_reference(G.codes_to_str(ref))
    r = pkg.RMI_LUT([1000], 15, "REF_1M.fa", matcher=m)
    r.train_RMI()
    o = oracle_mod.Oracle(ref, 15)
    coefs, icpts = r.rmi.coefficients()
    o.set_rmi([1000], coefs, icpts)
    return ref, r._index(), o


def test_full_size_properties_one_megabase(pkg, oracle_mod):
    """BASELINE configs[3] at full size: REF_1M, natively trained RMI [1000], 10 M x 150 bp from-ref reads (every
    100th read uniform-random): the same properties as on the 100 kb reference -- the larger, cache-missing index
    at full occupancy."""
    from genie_smem_amd import synth as B
    ref, ix, o = _index_1mb(pkg, oracle_mod)
    n_reads = 10_000_000
    rd_np = np.concatenate([B.reads_from_ref_fast(ref, 2_500_000, 150, 1004 + 31 * c) for c in range(4)])
    rd_np[::100] = B.reads_random(n_reads // 100, 150, 1006)
    _full_size_properties(ix, o, rd_np, 150, ix.n)


def test_sharded_run_equals_unsharded(pkg, oracle_mod):
    """BASELINE configs[4] cuts ONE batch into contiguous shards (parallel.shard_bounds), one per GPU, with rank-local
    outputs.  On one GPU: run W in {2, 4, 8} shards one after the other and check that the concatenated offsets /
    rows / status equal the unsharded call bit for bit (reads are independent units: SMEM/SMEM.py:20,206,456)."""
    import torch
    from genie_smem_amd import parallel, synth as B
    ref, ix, _ = _index_1mb(pkg, oracle_mod)
    n_reads = 200_003                                                # not a multiple of any W
    rd = torch.as_tensor(B.reads_from_ref_fast(ref, n_reads, 150, 1005)).cuda()
    rd[7, 3] = 9                                                     # a flagged read must stay flagged in its shard
    for algo in ("rmi", "lut"):
        off0, rows0, st0 = ix.find_smems(algo, rd)
        assert int(st0[7].item()) != 0 and int(st0.abs().sum().item()) == int(st0[7].abs().item())
        for W in (2, 4, 8):
            offs, rows, sts, base = [], [], [], 0
            for rank in range(W):
                lo, hi = parallel.shard_bounds(n_reads, rank, W)
                o_, r_, s_ = ix.find_smems(algo, rd[lo:hi])
                offs.append(o_[:-1] + base)
                base += int(o_[-1].item())
                rows.append(r_)
                sts.append(s_)
            offs.append(torch.tensor([base], dtype=off0.dtype, device=off0.device))
            assert torch.equal(torch.cat(offs), off0) and torch.equal(torch.cat(rows), rows0) and torch.equal(torch.cat(sts), st0), (algo, W)


def test_config4_eighty_million_reads_in_eight_shards(pkg, oracle_mod):
    """BASELINE configs[4] at FULL size on one GPU: the 80 M x 150 bp batch on the 1 Mb reference (natively trained RMI
    [1000]) cut into the W = 8 contiguous shards of parallel.shard_bounds and run one after the other, each shard generated
    exactly as rank r of `bench.py --config 4 --gpus 8` generates it (bench.shard_seed / bench.gen_reads).  Per shard:
    RMI and LUT runs bit-equal, no read flagged, every cover monotone and ending at the read end, intervals inside the
    suffix array, and a 20 000-read slice (its position moving through the shard) row by row against the CPU oracle.
    Reads are independent units (SMEM/SMEM.py:20,206,456), so the eight rank-local outputs ARE the result."""
    import torch
    import bench
    from genie_smem_amd import parallel
    cfg = bench.CONFIGS[4]
    assert cfg["reads"] == 80_000_000 and cfg["n"] == 1_000_000 and cfg["L"] == 150 and cfg["scaling"] == "strong"
    ref, ix, o = _index_1mb(pkg, oracle_mod)
    ref_dev = torch.as_tensor(ref).cuda()
    W, L, total_reads, total_rows = 8, cfg["L"], 0, 0
    for rank in range(W):
        lo, hi = parallel.shard_bounds(cfg["reads"], rank, W)
        n = hi - lo
        rd = torch.empty((n, L), dtype=torch.uint8, device="cuda")
        for c0 in range(0, n, bench.GEN_CHUNK):
            c1 = min(n, c0 + bench.GEN_CHUNK)
            rd[c0:c1] = bench.gen_reads(ref_dev, c1 - c0, L, bench.shard_seed(cfg, W, rank), c0 // bench.GEN_CHUNK)
        off, rows, st = ix.find_smems("rmi", rd, rows_hint=n * 16)
        assert int(st.abs().sum().item()) == 0
        off2, rows2, st2 = ix.find_smems("lut", rd, rows_hint=n * 16)
        assert torch.equal(off, off2) and torch.equal(rows, rows2) and torch.equal(st, st2), rank
        del off2, rows2, st2
        start, end, rlo, rhi = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
        assert bool(((start >= 0) & (start < end) & (end <= L) & (rlo >= 0) & (rlo <= rhi) & (rhi <= ix.n)).all())
        assert bool((end[off[1:] - 1] == L).all())                   # the last SMEM of every read ends at the read end
        inside = torch.ones(len(end) - 1, dtype=torch.bool, device=end.device)
        inside[off[1:-1] - 1] = False
        assert bool((end[1:][inside] > end[:-1][inside]).all())      # ends strictly increase inside a read
        del inside
        a = (n - 20_000) * rank // (W - 1)                           # oracle slice: start .. end of the shard as rank grows
        offc = off[a:a + 20_001].cpu().numpy()
        sm = rows[offc[0]:offc[-1]].cpu().numpy()
        counts, want = o.find_smems_batch("rmi", rd[a:a + 20_000].cpu().numpy(), nthreads=16)
        assert (np.diff(offc) == counts).all(), rank
        rr = np.repeat(np.arange(20_000), counts)
        tt = np.arange(len(rr)) - np.repeat(offc[:-1] - offc[0], counts)
        assert (want[rr, tt] == sm).all(), rank
        total_reads += n
        total_rows += int(off[-1].item())
        del rd, off, rows, st
        torch.cuda.empty_cache()
    assert total_reads == 80_000_000 and 8 * total_reads < total_rows < 16 * total_reads


# ------------------------------------------------------------------ both forms of the match table on the small fixtures
@pytest.mark.parametrize("fmt", ["wide", "compact"])
@pytest.mark.parametrize("ds", [d for d in G.DATASETS if G.have(d)])
def test_table_forms_golden_and_oracle(pkg, oracle_mod, ds, fmt):
    """The match table has two forms (genie_internal.h): compact 16-byte entries with 8-base keys (the default below 2^24
    bases) and 32-byte entries with 16-base keys.  Each is forced on the golden datasets: every golden group's ordered
    rows against the reference's own outputs, and fresh reads (from-ref, random, long exact copies of the reference,
    low complexity, ragged lengths, 1500 bases) against the CPU oracle, with the sampled and the every-position lookup."""
    import torch
    d, _ = G.load(ds)
    ref, K = d["ref_codes"], int(d["K"])
    ix = pkg.GenieIndex.build(ref, K, table_format=fmt)
    coefs, icpts, _, _, _ = ix.train_rmi([10, 100] if ds.startswith("medium") else [1000])
    ix = ix.to("cuda")
    assert ix.search_kernel_name("lut", 150) == ("match_table_kernel<6, true, false>" if fmt == "compact" else "match_table_kernel<8, false, false>")
    for dsg, tag, algo in G.group_cases():
        if dsg != ds or algo == "rmi":
            continue
        rd = G.reads(ds, tag)
        items = G.ref_items(ds, tag, algo)
        offsets, smems, st = ix.find_smems(algo, rd[:len(items)], min_len=1)
        assert (st.cpu().numpy() == 0).all()
        rows = _rows_per_read(offsets, smems)
        for r in range(len(items)):
            assert G.dict_view(rd[r], rows[r]) == items[r], (tag, algo, r)
    o = oracle_mod.Oracle(ref, K)
    o.set_rmi([10, 100] if ds.startswith("medium") else [1000], coefs, icpts)
    from genie_smem_amd import synth as B
    rng = np.random.default_rng(31)
    n = len(ref)
    batches = [B.reads_from_ref(ref, 700, 150, 41), B.reads_random(300, 150, 42), B.reads_from_ref(ref, 60, 1500, 43)]
    exact = np.stack([ref[p:p + 250] for p in rng.integers(0, n - 250, 200)])            # long exact matches
    exact[::2, 125] = (exact[::2, 125] + 1) % 4                                          # ... half of them with one substitution
    batches.append(exact)
    low = np.tile(rng.integers(0, 4, (100, 5)).astype(np.uint8), (1, 30))               # period-5 reads
    batches.append(low)
    batches.append(np.concatenate([ref[n - 300:], ref[:300]])[None, :].repeat(3, 0).astype(np.uint8))   # the reference's end, then its start
    for rd in batches:
        for sa in (0, 1):
            ix.set_option(pkg._native.OPT_SEARCH_ALL, sa)
            for algo in ("bwa", "lut", "rmi"):
                offsets, smems, st = ix.find_smems(algo, rd)
                assert int(st.abs().sum().item()) == 0
                rows = _rows_per_read(offsets, smems)
                counts, want = o.find_smems_batch(algo, rd, nthreads=16)
                for i in range(len(rd)):
                    assert rows[i].tolist() == want[i, :counts[i]].tolist(), (algo, sa, rd.shape, i)
    ix.set_option(pkg._native.OPT_SEARCH_ALL, 0)
    lens = rng.integers(K, 151, 500).astype(np.int32)                                    # ragged
    rd = B.reads_from_ref(ref, 500, 150, 44)
    offsets, smems, st = ix.find_smems("lut", torch.as_tensor(rd).cuda(), lens=torch.as_tensor(lens).cuda())
    rows = _rows_per_read(offsets, smems)
    counts, want = o.find_smems_batch("lut", rd, nthreads=16, lens=lens)
    for i in range(500):
        assert rows[i].tolist() == want[i, :counts[i]].tolist(), i


# ------------------------------------------------------------------ packed reads in, 8-byte rows out
def test_packed_entry_point_equals_csr(pkg, oracle_mod):
    """genie_find_smems_packed (2-bit packed reads, count / status bytes, 8-byte rows + escape list) gives, after
    packing.unpack_rows, exactly the offsets / int32 rows / status of genie_find_smems_csr: golden datasets with both table
    forms, ragged lengths, a flagged read, every length 1 .. 40 and 239 .. 255, the 10^6-read batch; and row spans of
    65535 or more (a reference in which every A is followed by C: the reads 'AG...' end an SMEM on a single base)."""
    import torch
    from genie_smem_amd import packing, synth as B
    rng = np.random.default_rng(5)

    def check(ix, mode, codes, lens=None, min_len=1):
        want = ix.find_smems(mode, codes, lens=None if lens is None else torch.as_tensor(lens).cuda(), min_len=min_len)
        c8, s8, r8, esc = ix.find_smems_packed(mode, torch.as_tensor(packing.pack_reads(codes)).cuda(), codes.shape[1],
                                               lens=None if lens is None else torch.as_tensor(lens).cuda(), min_len=min_len)
        off, rows = packing.unpack_rows(c8.cpu().numpy(), r8.cpu().numpy(), esc.cpu().numpy())
        assert np.array_equal(off, want[0].cpu().numpy()) and np.array_equal(rows, want[1].cpu().numpy())
        assert np.array_equal(s8.cpu().numpy().astype(np.int32), want[2].cpu().numpy())
        # the 6-byte rows of genie_find_smems_packed6 (lo in 24 bits, span in 8 with its own escapes): the same rows again
        c6, s6, r6, esc6 = ix.find_smems_packed(mode, torch.as_tensor(packing.pack_reads(codes)).cuda(), codes.shape[1],
                                                lens=None if lens is None else torch.as_tensor(lens).cuda(), min_len=min_len, row_bytes=6)
        assert r6.shape[1] == 6 and len(esc6) >= len(esc)
        off6, rows6 = packing.unpack_rows(c6.cpu().numpy(), r6.cpu().numpy(), esc6.cpu().numpy(), row_bytes=6)
        assert np.array_equal(off6, off) and np.array_equal(rows6, rows) and np.array_equal(s6.cpu().numpy(), s8.cpu().numpy())
        assert len(esc6) == int((rows[:, 3] - rows[:, 2] >= 255).sum())
        return len(esc)

    for ds in ("syn10k_K8", "big100k_K15"):
        d, _ = G.load(ds)
        ref, K = d["ref_codes"], int(d["K"])
        for fmt in ("wide", "compact"):
            ix = pkg.GenieIndex.build(ref, K, table_format=fmt)
            ix.train_rmi([10])
            ix = ix.to("cuda")
            rd = np.concatenate([B.reads_from_ref(ref, 400, 150, 3), B.reads_random(200, 150, 4)])
            for mode in ("bwa", "lut", "rmi"):
                check(ix, mode, rd)
            check(ix, "bwa", rd, min_len=12)
            lens = rng.integers(K, 151, len(rd)).astype(np.int32)
            check(ix, "lut", rd, lens)
            for L in list(range(1, 41)) + list(range(239, 256)):
                check(ix, "bwa", B.reads_from_ref(ref, 37, L, 100 + L))
    # the full-size batch of BASELINE config 1 through the host-side entry: counts and every row
    d, _ = G.load("syn100k_K15")
    ix = _index_for(pkg, "syn100k_K15", "rmi")
    rd = B.reads_from_ref_fast(d["ref_codes"], 1_000_000, 150, 1002)
    assert check(ix, "lut", rd) == 0
    off, rows, st = ix.find_smems_host("rmi", rd[:5000])
    counts, want = oracle_mod.Oracle(d["ref_codes"], 15).find_smems_batch("lut", rd[:5000], nthreads=16)
    assert (np.diff(off) == counts).all() and st.sum() == 0
    for i in range(5000):
        assert rows[off[i]:off[i + 1]].tolist() == want[i, :counts[i]].tolist(), i
    # spans of 65535 rows or more go through the escape list
    toks = [np.array(t, np.uint8) for t in ([0, 1], [1], [2], [3])]
    ref = np.concatenate([toks[i] for i in rng.choice(4, 400_000, p=[0.3, 0.2, 0.25, 0.25])])
    assert (ref == 0).sum() > 70_000 and not ((ref[:-1] == 0) & (ref[1:] != 1)).any()
    ix = pkg.GenieIndex.build(ref, 2).to("cuda")
    assert ix.search_kernel_name("bwa", 40) == "match_table_kernel<6, true, false>"
    rd = rng.integers(0, 4, (300, 40)).astype(np.uint8)
    rd[:, 0::7] = 0
    rd[:, 1::7] = 2                                                   # 'AG': the A is an SMEM of one base
    o = oracle_mod.Oracle(ref, 2)
    assert check(ix, "bwa", rd) > 300
    off, rows, st = ix.find_smems_host("bwa", rd)
    counts, want = o.find_smems_batch("bwa", rd, nthreads=8)
    assert (rows[:, 3] - rows[:, 2] >= 65535).sum() > 300
    for i in range(300):
        assert rows[off[i]:off[i + 1]].tolist() == want[i, :counts[i]].tolist(), i
    # argument checks of the entry point itself
    import ctypes as C
    lib = pkg._native.lib()
    pk = torch.as_tensor(packing.pack_reads(rd)).cuda()
    c8 = torch.empty(300, dtype=torch.uint8, device="cuda")
    r8 = torch.empty((4096, 8), dtype=torch.uint8, device="cuda")
    tot = torch.zeros(2, dtype=torch.int64, device="cuda")
    wsb = int(lib.genie_find_smems_workspace_bytes(300, 300))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())                                                  # noqa: E731

    def call(stride=pk.shape[1], L=40, reads=pk, rows=r8, totals=tot):
        return lib.genie_find_smems_packed(ix._h, 0, P(reads), None, 300, stride, L, 1, P(c8), P(c8), P(rows), 4096, P(totals), None, 0,
                                           P(ws), wsb, None)
    assert call(L=256) == -6                                           # start / end are bytes
    assert call(stride=pk.shape[1] + 2) == -1 and call(stride=8) == -1          # a multiple of 4, at least 4 * ceil(L / 16)
    assert lib.genie_find_smems_packed(ix._h, 0, C.c_void_p(pk.data_ptr() + 1), None, 300, pk.shape[1], 40, 1, P(c8), P(c8), P(r8), 4096,
                                       P(tot), None, 0, P(ws), wsb, None) == -1            # misaligned reads
    assert lib.genie_find_smems_packed(ix._h, 7, P(pk), None, 300, pk.shape[1], 40, 1, P(c8), P(c8), P(r8), 4096, P(tot), None, 0,
                                       P(ws), wsb, None) == -1                              # no such mode
    assert call() == 0 and int(tot[1].item()) > 300                    # no escape list given: the count still says how many there are
    torch.cuda.synchronize()
    # a bad base is caught while packing on the host, as the reference raises KeyError
    bad = rd[:3].copy()
    bad[1, 5] = 7
    with pytest.raises(KeyError):
        packing.pack_reads(bad)


def test_image_validation_and_seedless_image(pkg):
    """genie_index_validate: an image whose CONTENTS point outside its sections (a suffix start beyond n, a match-table row
    beyond the suffix array, an overflow link beyond the blocks) is refused when it is bound, for both table forms; the
    image without the K-mer hash table finds SMEMs like the full one and refuses LUT seed lookups."""
    import struct
    import torch
    from test_host_index import _parse
    from genie_smem_amd import synth as B
    ref = np.random.default_rng(8).integers(0, 4, 50_000).astype(np.uint8)
    for fmt in ("compact", "wide"):
        ix = pkg.GenieIndex.build(ref, 12, table_format=fmt)
        img = ix.serialize().clone()
        h = _parse(img.numpy())
        assert pkg.GenieIndex.from_image(img.cuda()).n == len(ref)
        esz = 16 if fmt == "compact" else 32
        first_present = 11
        edits = [(h["off_sa"] + 16 * 777, struct.pack("<i", len(ref) + 5)),                         # a suffix start beyond n
                 (h["off_dir"] + 4 * 100, struct.pack("<I", len(ref) + 9)),                         # a directory entry beyond the rows
                 (h["off_dir2"] + 16 * 5, struct.pack("<II", len(ref) - 1, 7))]                    # a range beyond the rows
        if fmt == "compact":
            edits.append((h["off_mtab"] + esz * first_present, struct.pack("<I", (len(ref) - 2) | (6 << 24))))     # 6 rows from n - 2
            edits.append((h["off_mtab"] + esz * first_present, struct.pack("<I", 5 | (7 << 24)) + bytes(10) + struct.pack("<H", 60000)))   # overflow link
        else:
            edits.append((h["off_mtab"] + esz * first_present, struct.pack("<II", h["P2"] | (0x1F << 8) | (6 << 24), len(ref) - 2)))
        for off, blob in edits:
            bad = img.clone()
            bad[off:off + len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            with pytest.raises(pkg._native.GenieError) as e:
                pkg.GenieIndex.from_image(bad.cuda())
            assert e.value.status == -8, (fmt, off)
    ix = pkg.GenieIndex.build(ref, 12)
    ix.train_rmi([10])
    full = pkg.GenieIndex.from_image(ix.serialize().cuda())
    slim = pkg.GenieIndex.from_image(ix.serialize(seed_table=False).cuda())
    assert slim.blob.numel() < full.blob.numel()
    rd = B.reads_from_ref(ref, 300, 120, 2)
    for mode in ("bwa", "lut", "rmi"):
        a, b = full.find_smems(mode, rd), slim.find_smems(mode, rd)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    km = rd[:50, :12]
    assert torch.equal(full.seed_lookup("rmi", km), slim.seed_lookup("rmi", km))
    full.seed_lookup("lut", km)
    with pytest.raises(pkg._native.GenieError) as e:
        slim.seed_lookup("lut", km)
    assert e.value.status == -9


# ------------------------------------------------------------------ beyond the BASELINE sizes
def test_two_and_a_half_megabase_reference(pkg, oracle_mod):
    """A reference of the size of the reference's own data/full_data.fa (2 499 750 bases -- the one its README says it
    cannot index): synthetic bases, K = 15, natively trained RMI [1000], a 16 MB compact table that no XCD's L2 holds.
    3 000 from-ref and 1 000 random reads, three modes, row by row against the CPU oracle; exact-match patterns too."""
    from genie_smem_amd import synth as B
    n = 2_499_750
    ref = B.synth_ref(n, n)
    ix = pkg.GenieIndex.build(ref, 15)
    coefs, icpts, _, _, _ = ix.train_rmi([1000])
    ix = ix.to("cuda", seed_table=False)
    assert ix.search_kernel_name("rmi", 150) == "match_table_kernel<4, true, false>"      # the table exceeds an XCD's L2
    o = oracle_mod.Oracle(ref, 15)
    o.set_rmi([1000], coefs, icpts)
    for kind, rd in (("fromref", B.reads_from_ref(ref, 3000, 150, 2501)), ("random", B.reads_random(1000, 150, 2502))):
        for algo in ("bwa", "lut", "rmi"):
            offsets, smems, st = ix.find_smems(algo, rd)
            assert int(st.abs().sum().item()) == 0
            rows = _rows_per_read(offsets, smems)
            counts, want = o.find_smems_batch(algo, rd, nthreads=16)
            for i in range(len(rd)):
                assert rows[i].tolist() == want[i, :counts[i]].tolist(), (kind, algo, i)
    rng = np.random.default_rng(10)
    pats = np.zeros((1500, 60), np.uint8)
    lens = rng.integers(1, 61, 1500).astype(np.int32)
    for i in range(1500):
        p0 = int(rng.integers(0, n - 60))
        pats[i, :lens[i]] = ref[p0:p0 + lens[i]]
        if i % 3 == 0:
            pats[i, lens[i] - 1] = (pats[i, lens[i] - 1] + 1) % 4
    got = ix.sa_interval(pats, lens).cpu().numpy()
    for i in range(1500):
        assert tuple(got[i]) == o.back_prop(pats[i, :lens[i]]), i


def test_seventeen_megabase_reference_takes_the_wide_table(pkg, oracle_mod):
    """Past 2^24 bases a suffix-array row no longer fits the compact entries' 24 bits: the index falls back to the
    32-byte form by itself.  17 000 000 synthetic bases (P2 = 12: a 537 MB match table, a 1 GB image), K = 15, natively
    trained RMI: from-ref and random reads in the three modes and exact-match patterns against the CPU oracle."""
    from genie_smem_amd import synth as B
    n = 17_000_000
    ref = B.synth_ref(n, n)
    ix = pkg.GenieIndex.build(ref, 15)
    coefs, icpts, _, _, _ = ix.train_rmi([1000])
    ix = ix.to("cuda", seed_table=False)
    assert ix.search_kernel_name("rmi", 150) == "match_table_kernel<4, false, false>"
    o = oracle_mod.Oracle(ref, 15)
    o.set_rmi([1000], coefs, icpts)
    for kind, rd in (("fromref", B.reads_from_ref(ref, 2000, 150, 1701)), ("random", B.reads_random(500, 150, 1702))):
        for algo in ("bwa", "lut", "rmi"):
            offsets, smems, st = ix.find_smems(algo, rd)
            assert int(st.abs().sum().item()) == 0
            rows = _rows_per_read(offsets, smems)
            counts, want = o.find_smems_batch(algo, rd, nthreads=16)
            for i in range(len(rd)):
                assert rows[i].tolist() == want[i, :counts[i]].tolist(), (kind, algo, i)
    rng = np.random.default_rng(17)
    pats = np.zeros((1000, 60), np.uint8)
    lens = rng.integers(1, 61, 1000).astype(np.int32)
    for i in range(1000):
        p0 = int(rng.integers(0, n - 60))
        pats[i, :lens[i]] = ref[p0:p0 + lens[i]]
        if i % 3 == 0:
            pats[i, lens[i] - 1] = (pats[i, lens[i] - 1] + 1) % 4
    got = ix.sa_interval(pats, lens).cpu().numpy()
    for i in range(1000):
        assert tuple(got[i]) == o.back_prop(pats[i, :lens[i]]), i


@pytest.mark.gpu
def test_scheduling_option_changes_no_result(pkg):
    """GENIE_OPT_SCHEDULING (fixed shares / no priority rotation in the search and interval kernels: A/B timing only): every value
    gives the rows of the default, on a batch large enough that every block of the persistent grids has several groups, on long
    reads and on a ragged tail (N not a multiple of anything)."""
    import torch
    from genie_smem_amd import synth as B
    d, _ = G.load("syn100k_K15")
    ix = _index_for(pkg, "syn100k_K15", "lut")
    cases = [("lut", B.reads_from_ref_fast(d["ref_codes"], 200_003, 150, 77)), ("bwa", B.reads_from_ref(d["ref_codes"], 1501, 700, 78)),
             ("lut", B.reads_random(4097, 100, 79))]
    try:
        for mode, rd in cases:
            rd = torch.as_tensor(rd).cuda()
            ix.set_option(pkg._native.OPT_SCHEDULING, 0)
            want = [t.clone() for t in ix.find_smems(mode, rd)]
            for v in range(1, 16):
                ix.set_option(pkg._native.OPT_SCHEDULING, v)
                got = ix.find_smems(mode, rd)
                assert all(torch.equal(a, b) for a, b in zip(got, want)), (mode, v)
    finally:
        ix.set_option(pkg._native.OPT_SCHEDULING, 0)


@pytest.mark.gpu
def test_long_reads_large_batch_against_oracle(pkg, oracle_mod):
    """Long reads in a batch of 32 768 or more take the two-lanes-per-read traversal (eight positions per lane and pass, the row
    staged in LDS windows): every row of 40 000 x 300-base and 33 000 x 1000-base reads (from-ref, random, exact copies of the
    reference -- one match longer than a window -- and ragged lengths) against the CPU oracle, in two modes."""
    import torch
    from genie_smem_amd import synth as B
    d, _ = G.load("syn100k_K15")
    ref = d["ref_codes"]
    ix = _index_for(pkg, "syn100k_K15", "lut")
    o = oracle_mod.Oracle(ref, 15)
    rng = np.random.default_rng(9)
    for n_reads, L in ((40_000, 300), (33_000, 1000)):
        rd = B.reads_from_ref_fast(ref, n_reads, L, 500 + L)
        rd[:200] = B.reads_random(200, L, 7)
        for t, p in enumerate(rng.integers(0, len(ref) - L, 100)):           # whole-read matches, some with one substitution
            rd[200 + t] = ref[p:p + L]
            if t & 1:
                rd[200 + t, L // 2] = (rd[200 + t, L // 2] + 1) % 4
        lens = np.full(n_reads, L, np.int32)
        lens[300:2300] = rng.integers(15, L + 1, 2000)
        for mode in ("lut", "bwa"):
            off, rows, st = ix.find_smems(mode, torch.as_tensor(rd).cuda(), lens=torch.as_tensor(lens).cuda(), min_len=1)
            assert int(st.abs().sum().item()) == 0
            off, rows = off.cpu().numpy(), rows.cpu().numpy()
            sel = np.concatenate([np.arange(0, 2400), rng.integers(2400, n_reads, 1600)])
            counts, want = o.find_smems_batch(mode, rd[sel], nthreads=16, lens=lens[sel])
            for t, r in enumerate(sel):
                assert rows[off[r]:off[r + 1]].tolist() == want[t, :counts[t]].tolist(), (L, mode, int(r))

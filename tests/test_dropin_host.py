"""CPU-only tests of the drop-in classes' host logic: file formats (FASTA, the reference's FM / LUT
JSON schemas), encoders, mapping views, error behaviour.  No kernels run (the search methods need
a GPU and are covered by tests/test_gpu_parity.py)."""
import json
import os

import numpy as np
import pytest

import golden_util as G


@pytest.fixture(scope="module")
def pkg():
    import genie_smem_amd as g
    g._native.build()
    return g


@pytest.fixture()
def data_dir(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    return d


def _write_fasta(d, name, seq):
    with open(d / name, "w") as fh:
        fh.write(">refseq\n")
        for i in range(0, len(seq), 50):
            fh.write(seq[i:i + 50] + "\n")


def test_fasta_and_fm_json_schema(pkg, data_dir):
    """create_fm_index writes the reference's schema (ExactMatch.py:29-33); values checked against
    the reference's own outputs for mississippi (tests/golden/known_answers.json)."""
    k = G.known()["mississippi"]
    _write_fasta(data_dir, "mississippi.fa", k["ref"])
    m = pkg.ExactMatch("mississippi.fa", data_dir=str(data_dir))
    m.load_ref_sequence()
    assert m.ref_sequence == "mississippi$" and m.ref_size == 12 and m.alphabet == "imps"
    m.create_fm_index()
    fm = json.load(open(data_dir / "mississippi-FM.json"))
    assert sorted(fm) == ["bwt_array", "count_dic", "occurance_matrix", "ref_size", "suffix_array"]
    assert fm["suffix_array"] == k["fm"]["sa_head"] and fm["count_dic"] == k["fm"]["count_dic"] and fm["ref_size"] == 12
    assert "".join(fm["bwt_array"]) == "ipssm$pissii"
    assert fm["occurance_matrix"]["s"] == [0, 0, 1, 2, 2, 2, 2, 2, 3, 4, 4, 4]
    # round trip: a fresh object adopts the file's suffix array
    m2 = pkg.ExactMatch("mississippi.fa", data_dir=str(data_dir))
    m2.load_fm_index()
    assert m2.ref_size == 12 and m2.get_positions(3, 4) == [5, 2] and m2.get_position(0) == 12
    with pytest.raises(FileNotFoundError, match="No FM index file found"):
        pkg.ExactMatch("nothing.fa", data_dir=str(data_dir)).load_fm_index()


@pytest.mark.skipif(not os.path.isdir("/root/reference/SMEM/data"), reason="reference data files not present")
def test_fm_json_identical_to_reference_fixture(pkg, data_dir):
    ref_dir = "/root/reference/SMEM/data"
    with open(os.path.join(ref_dir, "small_data.fa")) as fh:
        fh.readline()
        seq = "".join(line.strip() for line in fh)
    _write_fasta(data_dir, "small_data.fa", seq)
    m = pkg.ExactMatch("small_data.fa", data_dir=str(data_dir))
    m.create_fm_index()
    assert json.load(open(data_dir / "small_data-FM.json")) == json.load(open(os.path.join(ref_dir, "small_data-FM.json")))


def test_lut_class_and_json(pkg, data_dir):
    k = G.known()["paperex"]
    _write_fasta(data_dir, "paperex.fa", k["ref"])
    m = pkg.ExactMatch("paperex.fa", data_dir=str(data_dir))
    lut = pkg.LUT(m)
    with pytest.raises(RuntimeError, match="LUT has not been created yet"):
        lut.save_lut()
    lut.generate_lut(2)
    assert lut.lut_size == 2 and len(lut.lut) == len(k["lut"])
    assert {kk: [list(v[0]), v[1]] for kk, v in lut.lut.items()} == k["lut"]
    assert "3" in lut.lut and "999" not in lut.lut and 3 not in lut.lut          # keys are strings (LUT.py:35)
    with pytest.raises(KeyError):
        lut.lut["999"]
    lut.save_lut()
    saved = json.load(open(data_dir / "paperex-LUT.json"))
    assert saved == {"lut": k["lut"], "lut_size": 2}
    lut2 = pkg.LUT(m)
    lut2.load_lut()
    assert lut2.lut_size == 2 and lut2.lut.keys() == lut.lut.keys()
    # encoder (LUT.py:37-48)
    assert pkg.LUT.convert_seq_to_num("ACGT") == 0b00011011 and pkg.LUT.convert_seq_to_num("") == 0
    with pytest.raises(KeyError):
        pkg.LUT.convert_seq_to_num("ACNT")


def test_encode_and_keyerrors(pkg):
    m = pkg.ExactMatch("x.fa")
    m.set_reference("ACACACCACAACCA")
    assert m.alphabet == "ACGT" and m.encode("CAC").tolist() == [1, 0, 1]
    with pytest.raises(KeyError):
        m.encode("ACGA")                        # G never occurs in this reference: count_dic has no 'G'
    with pytest.raises(KeyError):
        m.encode("AC$")
    m2 = pkg.ExactMatch("y.fa")
    with pytest.raises(ValueError, match="at most 4 distinct symbols"):
        m2.set_reference("ACGTN")
    q = m.create_query(5)
    assert len(q) == 5 and q in m.ref_sequence


def test_add_one_matches_full_search_on_host_arrays(pkg):
    """exact_match_back_prop_add_one is a host-side LF step; check it against brute force."""
    d, _ = G.load("medium_K6")
    ref = G.codes_to_str(d["ref_codes"][:600])
    m = pkg.ExactMatch("m.fa")
    m.set_reference(ref)
    sa = m.host_index(0).suffix_array()
    suffixes = [ref[s - 1:] + "$" for s in sa]

    def brute(p):
        rows = [r for r, s in enumerate(suffixes) if s.startswith(p)]
        return (rows[0], rows[-1]) if rows else -1

    rng = np.random.default_rng(0)
    for _ in range(200):
        a = int(rng.integers(0, len(ref) - 8))
        pat = ref[a + 1:a + int(rng.integers(2, 8))]
        ch = "ACGT"[int(rng.integers(0, 4))]
        assert m.exact_match_back_prop_add_one(ch, brute(pat)) == brute(ch + pat)


def test_rmi_lut_save_load_roundtrip(pkg, tmp_path):
    d, _ = G.load("syn10k_K8")
    m = pkg.ExactMatch("syn10k.fa")
    m.set_reference(G.codes_to_str(d["ref_codes"]))
    r = pkg.RMI_LUT([10, 100], 8, "syn10k.fa", matcher=m)
    r.train_RMI()
    f = str(tmp_path / "rmi_file.npz")
    r.save(f)
    r2 = pkg.RMI_LUT.load(f, matcher=m)
    assert r2.structure == [10, 100] and r2.prediction_size == 8
    q = G.codes_to_str(d["ref_codes"][40:48])
    assert r2.rmi_predict(q)[0] == r.rmi_predict(q)[0]
    assert r.get_ref_seq(0) is None and r.get_ref_seq(5) == m.ref_sequence[r.suffix_array[5] - 1:][:8]
    with pytest.raises(KeyError):
        r.rmi_predict("ACGTNACG")


def test_query_generators(pkg):
    import random
    random.seed(1)
    q = pkg.create_random_query(40)
    assert len(q) == 40 and set(q) <= set("ACGT")
    q2 = pkg.create_query_from_ref("ACGTTGCATGCAGTCAGTCGATCGATGCATGCATGCAAGTC", 25)
    assert len(q2) == 25


def test_workspace_rows_and_index_create_ex_arguments():
    """Host-only ABI entry points: per-read workspace rows (traffic accounting) and argument checks of
    genie_index_create_ex (no GPU involved)."""
    import ctypes as C
    import genie_smem_amd as pkg
    w = pkg.GenieIndex.workspace_shape(150)
    assert w == {"fwd_stride": 160, "qp_recs": 3, "kj_row_bytes": 2 * 152}
    w = pkg.GenieIndex.workspace_shape(2000)
    assert w["fwd_stride"] == 4000 and w["qp_recs"] == 63
    lib = pkg._native.lib()
    assert lib.genie_find_smems_workspace_rows(-1, (C.c_int32 * 4)()) == -1
    codes = np.zeros(64, np.uint8)
    h = C.c_void_p()
    u8p = C.POINTER(C.c_uint8)
    assert lib.genie_index_create_ex(codes.ctypes.data_as(u8p), 64, None, 6, 7, 13, C.byref(h)) == -1      # table_bits out of range
    assert lib.genie_index_create_ex(codes.ctypes.data_as(u8p), 64, None, 6, 7, 7, C.byref(h)) == -1       # ... not above dir_bits
    assert lib.genie_index_create_ex(codes.ctypes.data_as(u8p), 64, None, 6, 0, 5, C.byref(h)) == -1       # (dir_bits 0 = 7)
    assert lib.genie_index_create_ex(codes.ctypes.data_as(u8p), 64, None, 6, 7, 9, C.byref(h)) == 0
    lib.genie_index_destroy(h)


def test_packing_layouts():
    """packing.pack_reads / unpack_rows: the host side of genie_find_smems_packed (layout conversions only)."""
    from genie_smem_amd import packing
    rng = np.random.default_rng(1)
    for L in (1, 3, 4, 15, 16, 17, 100, 150, 255):
        c = rng.integers(0, 4, (9, L)).astype(np.uint8)
        p = packing.pack_reads(c)
        assert p.shape == (9, packing.packed_stride(L)) and p.shape[1] % 4 == 0 and p.shape[1] >= 4 * ((L + 15) // 16)
        assert (packing.unpack_reads(p, L) == c).all()
        assert p[0, 0] >> 6 == c[0, 0]                                   # the first base in the top bits of byte 0
        assert (p[:, (L + 3) // 4:] == 0).all()
    with pytest.raises(KeyError):
        packing.pack_reads(np.array([[0, 4, 1]], np.uint8))
    counts = np.array([2, 0, 1], np.uint8)
    r8 = np.zeros(3, packing.ROW8)
    r8["start"], r8["end"], r8["span"], r8["lo"] = [0, 5, 0], [5, 9, 4], [0, 0xFFFF, 3], [7, 100, 9]
    off, rows = packing.unpack_rows(counts, r8.view(np.uint8), np.array([[1, 70100], [9, 1]]))       # (an escape of a dropped row is ignored)
    assert off.tolist() == [0, 2, 2, 3] and rows.tolist() == [[0, 5, 7, 7], [5, 9, 100, 70100], [0, 4, 9, 12]]
    with pytest.raises(ValueError):
        packing.unpack_rows(counts, r8.view(np.uint8), np.zeros((0, 2), np.int64))
    # the 6-byte rows of genie_find_smems_packed6: start, end, lo (24 bits, little endian), span (0xFF: on the escape list)
    r6 = np.array([[0, 5, 7, 0, 0, 0], [5, 9, 0x10, 0x27, 0x01, 0xFF], [0, 4, 9, 0, 0, 3]], np.uint8)
    off, rows = packing.unpack_rows(counts, r6, np.array([[1, 0x12710 + 300]]), row_bytes=6)
    assert off.tolist() == [0, 2, 2, 3] and rows.tolist() == [[0, 5, 7, 7], [5, 9, 0x12710, 0x12710 + 300], [0, 4, 9, 12]]
    with pytest.raises(ValueError):
        packing.unpack_rows(counts, r6, row_bytes=6)
    with pytest.raises(ValueError):
        packing.unpack_rows(counts, r6, row_bytes=7)
    with pytest.raises(ValueError):
        packing.unpack_rows(counts, r8.view(np.uint8))

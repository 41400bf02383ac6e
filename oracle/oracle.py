"""ctypes binding of the CPU oracle (oracle/libsmem_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See oracle/smem_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsmem_oracle.so")

ERR = {-1: "ABSENT", -2: "EKEY", -3: "ESHORT", -4: "ERUNAWAY", -5: "ECAP", -6: "ENOMODEL", -7: "ERECURSE"}
MODES = {"bwa": 0, "lut": 1, "rmi": 2}


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("smem_oracle.c", "smem_oracle.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsmem_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, u8p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_uint8)
        i32p, i64p, dp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        L.orc_index_build.restype = vp
        L.orc_index_build.argtypes = [u8p, i64, C.c_int]
        L.orc_index_free.argtypes = [vp]
        L.orc_n.restype = i64
        L.orc_n.argtypes = [vp]
        L.orc_suffix_array.restype = i32p
        L.orc_suffix_array.argtypes = [vp]
        L.orc_bwt.restype = u8p
        L.orc_bwt.argtypes = [vp]
        L.orc_count.restype = i32
        L.orc_count.argtypes = [vp, C.c_int]
        L.orc_occ.restype = i32p
        L.orc_occ.argtypes = [vp, C.c_int]
        L.orc_lut_size.restype = i64
        L.orc_lut_size.argtypes = [vp]
        L.orc_lut_codes.restype = C.POINTER(C.c_uint32)
        L.orc_lut_codes.argtypes = [vp]
        L.orc_lut_lo.restype = i32p
        L.orc_lut_lo.argtypes = [vp]
        L.orc_lut_hi.restype = i32p
        L.orc_lut_hi.argtypes = [vp]
        L.orc_back_prop.restype = C.c_int
        L.orc_back_prop.argtypes = [vp, u8p, C.c_int, i32p, i32p]
        L.orc_set_rmi.restype = C.c_int
        L.orc_set_rmi.argtypes = [vp, C.c_int, i32p, i32p, dp, dp]
        L.orc_rmi_predict.restype = C.c_double
        L.orc_rmi_predict.argtypes = [vp, C.c_uint64]
        L.orc_rmi_suffix.restype = C.c_int
        L.orc_rmi_suffix.argtypes = [vp, u8p, C.c_int, i64p, i64p]
        L.orc_find_smems.restype = C.c_int
        L.orc_find_smems.argtypes = [vp, C.c_int, u8p, C.c_int, C.c_int, i32p, C.c_int]
        L.orc_find_smems_batch.restype = None
        L.orc_find_smems_batch.argtypes = [vp, C.c_int, u8p, i64, i32, i32p, i32, C.c_int, i32p, i32p,
                                           C.c_int, C.c_int]
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Oracle:
    """One reference + K: SA, FM index, LUT (and optionally an RMI model)."""

    def __init__(self, codes, K):
        codes = np.ascontiguousarray(codes, np.uint8)
        assert codes.ndim == 1 and (codes.size == 0 or codes.max() <= 3)
        self.codes, self.K, self.n = codes, int(K), int(codes.size)
        self._h = lib().orc_index_build(_u8(codes), self.n, self.K)

    def __del__(self):
        if getattr(self, "_h", None) and callable(lib):          # (module globals are gone at interpreter shutdown)
            lib().orc_index_free(self._h)
            self._h = None

    # -- index views --------------------------------------------------------------
    @property
    def suffix_array(self):
        """1-based, row 0 = n+1 (the reference's `fm_index["suffix_array"]`)."""
        return np.ctypeslib.as_array(lib().orc_suffix_array(self._h), (self.n + 1,)).copy()

    def lut_arrays(self):
        m = lib().orc_lut_size(self._h)
        if m == 0:
            z = np.zeros(0, np.int32)
            return z.astype(np.uint32), z, z
        return (np.ctypeslib.as_array(lib().orc_lut_codes(self._h), (m,)).copy(),
                np.ctypeslib.as_array(lib().orc_lut_lo(self._h), (m,)).copy(),
                np.ctypeslib.as_array(lib().orc_lut_hi(self._h), (m,)).copy())

    def count(self, code):
        return lib().orc_count(self._h, code)

    def occ(self, code):
        return np.ctypeslib.as_array(lib().orc_occ(self._h, code), (self.n + 1,)).copy()

    def bwt(self):
        return np.ctypeslib.as_array(lib().orc_bwt(self._h), (self.n + 1,)).copy()

    # -- A1 -----------------------------------------------------------------------
    def back_prop(self, pat):
        pat = np.ascontiguousarray(pat, np.uint8)
        lo, hi = C.c_int32(), C.c_int32()
        rc = lib().orc_back_prop(self._h, _u8(pat), int(pat.size), C.byref(lo), C.byref(hi))
        if rc == -2:
            raise KeyError("base outside the reference alphabet")
        return (-1, -1) if rc else (lo.value, hi.value)

    # -- A8 -----------------------------------------------------------------------
    def set_rmi(self, experts, coefs, icpts):
        """experts: the reference's list (e.g. [1000]); coefs/icpts: per-level float64 arrays
        (level 0 has one model, level l has experts[l-1])."""
        sizes = np.asarray([len(c) for c in coefs], np.int32)
        scales = np.asarray(list(experts) + [1], np.int32)
        assert len(sizes) == len(scales) and sizes[0] == 1
        coef = np.ascontiguousarray(np.concatenate(coefs), np.float64)
        icpt = np.ascontiguousarray(np.concatenate(icpts), np.float64)
        i32p, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        lib().orc_set_rmi(self._h, len(sizes), sizes.ctypes.data_as(i32p), scales.ctypes.data_as(i32p),
                          coef.ctypes.data_as(dp), icpt.ctypes.data_as(dp))

    def rmi_predict(self, code):
        return lib().orc_rmi_predict(self._h, int(code))

    def rmi_suffix(self, kmer, compat=False):
        kmer = np.ascontiguousarray(kmer, np.uint8)
        lo, hi = C.c_int64(), C.c_int64()
        rc = lib().orc_rmi_suffix(self._h, _u8(kmer), int(bool(compat)), C.byref(lo), C.byref(hi))
        return rc, lo.value, hi.value

    # -- A2/A5/A9 -----------------------------------------------------------------
    def find_smems(self, mode, read, min_len=1):
        read = np.ascontiguousarray(read, np.uint8)
        cap = int(read.size) + 1
        out = np.empty((cap, 4), np.int32)
        rc = lib().orc_find_smems(self._h, MODES[mode], _u8(read), int(read.size), int(min_len),
                                  out.ctypes.data_as(C.POINTER(C.c_int32)), cap)
        if rc < 0:
            return rc, None
        return rc, out[:rc].copy()

    def find_smems_batch(self, mode, reads, min_len=1, cap=None, nthreads=1, lens=None):
        reads = np.ascontiguousarray(reads, np.uint8)
        N, L = reads.shape
        cap = cap or (L + 1)
        counts = np.empty(N, np.int32)
        out = np.empty((N, cap, 4), np.int32)
        i32p = C.POINTER(C.c_int32)
        lp = None if lens is None else np.ascontiguousarray(lens, np.int32).ctypes.data_as(i32p)
        lib().orc_find_smems_batch(self._h, MODES[mode], _u8(reads), N, L, lp, L, int(min_len),
                                   counts.ctypes.data_as(i32p), out.ctypes.data_as(i32p), cap, int(nthreads))
        return counts, out

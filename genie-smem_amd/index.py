"""GenieIndex: the device-resident index (suffix array, 2-bit packed reference, P-mer prefix
directory, K-mer table, optional RMI) and the batched entry points of the hot path.

Everything that computes goes through the C ABI (include/genie_smem.h); torch only owns the
device memory, the stream and -- multi-GPU -- the one-time broadcast of the index image.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device):
    if device.type != "cuda":
        raise RuntimeError("the SMEM hot path runs on an MI355X only (no CPU fallback); tensor is on " + str(device))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class GenieIndex:
    def __init__(self):
        self._h = C.c_void_p(0)
        self.blob = None            # torch uint8 tensor holding the device image (keeps it alive)
        self._host_blob = None
        self.device = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.lib().genie_index_destroy(h)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass

    # ------------------------------------------------------------------ construction (host)
    @classmethod
    def build(cls, codes, K, dir_bits=7, sa_one_based=None, table_bits=0, table_format="auto"):
        """codes: uint8 array of base codes 0..3; K: LUT / RMI key size (0 = none);
        sa_one_based: adopt this suffix array (reference JSON convention) instead of building;
        table_bits: P2 of the per-P2-mer tables (0 = automatic); table_format: "auto" (compact 16-byte entries below 2^24
        bases), "wide" or "compact" -- tuning knobs of the index image only."""
        codes = np.ascontiguousarray(codes, np.uint8)
        self = cls()
        u8p = C.POINTER(C.c_uint8)
        sa_p = None
        if sa_one_based is not None:
            sa = np.ascontiguousarray(sa_one_based, np.int32)
            if sa.size != codes.size + 1:
                raise ValueError("suffix array must have n+1 rows")
            sa_p = sa.ctypes.data_as(C.POINTER(C.c_int32))
        fmt = {"auto": 0, "wide": N.TABLE_WIDE, "compact": N.TABLE_COMPACT}[table_format]
        rc = N.lib().genie_index_create_ex(codes.ctypes.data_as(u8p), codes.size, sa_p, int(K), int(dir_bits),
                                           int(table_bits) | fmt, C.byref(self._h))
        N.check(rc, "genie_index_create")
        return self

    def set_rmi(self, experts, coefs, icpts):
        """experts: the reference's list (RMI.experts); coefs/icpts: per-level float64 arrays."""
        sizes = np.asarray([len(c) for c in coefs], np.int32)
        scales = np.asarray(list(experts) + [1], np.int32)
        if len(sizes) != len(scales):
            raise ValueError("need len(experts)+1 levels of coefficients")
        coef = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float64) for c in coefs]))
        icpt = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float64) for c in icpts]))
        i32p, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        N.check(N.lib().genie_index_set_rmi(self._h, len(sizes), sizes.ctypes.data_as(i32p),
                                            scales.ctypes.data_as(i32p), coef.ctypes.data_as(dp),
                                            icpt.ctypes.data_as(dp)), "genie_index_set_rmi")
        self._host_blob = None

    def train_rmi(self, experts):
        """Native RMI training (genie_index_train_rmi; RMI_LUT.train_RMI + RMI.fit without scikit-learn).
        Returns (coefs, icpts, leaf_err, mean_abs_err, max_abs_err): per-level float64 arrays, the
        per-leaf error bounds, and the error statistics over the training pairs."""
        ex = np.asarray(list(experts), np.int32)
        mean = C.c_double(0.0)
        worst = C.c_int32(0)
        N.check(N.lib().genie_index_train_rmi(self._h, len(ex), ex.ctypes.data_as(C.c_void_p), C.byref(mean),
                                              C.byref(worst)), "genie_index_train_rmi")
        self._host_blob = None
        sizes = [1] + [int(e) for e in ex]
        coef = np.empty(sum(sizes), np.float64)
        icpt = np.empty(sum(sizes), np.float64)
        err = np.empty(sizes[-1], np.int32)
        N.check(N.lib().genie_index_rmi_models(self._h, coef.ctypes.data_as(C.c_void_p), icpt.ctypes.data_as(C.c_void_p),
                                               err.ctypes.data_as(C.c_void_p)), "genie_index_rmi_models")
        cuts = np.cumsum([0] + sizes)
        coefs = [coef[cuts[i]:cuts[i + 1]].copy() for i in range(len(sizes))]
        icpts = [icpt[cuts[i]:cuts[i + 1]].copy() for i in range(len(sizes))]
        return coefs, icpts, err, float(mean.value), int(worst.value)

    def info(self):
        inf = N.GenieInfo()
        N.check(N.lib().genie_index_info(self._h, C.byref(inf)), "genie_index_info")
        return {f: getattr(inf, f) for f, _ in N.GenieInfo._fields_}

    @property
    def n(self):
        return self.info()["n"]

    @property
    def K(self):
        return self.info()["K"]

    def suffix_array(self):
        """1-based suffix array as the reference stores it (row 0 = n+1); host copy."""
        p = N.lib().genie_index_suffix_array(self._h)
        if not p:
            raise RuntimeError("this handle has no host arrays (opened from a broadcast image)")
        return np.ctypeslib.as_array(p, (self.n + 1,))

    def lut_arrays(self):
        """Sorted distinct K-mer codes and their inclusive SA intervals (host views)."""
        cp, lp, hp = C.POINTER(C.c_uint32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        N.check(N.lib().genie_index_lut_arrays(self._h, C.byref(cp), C.byref(lp), C.byref(hp)), "genie_index_lut_arrays")
        m = self.info()["lut_keys"]
        if m == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.int32), np.zeros(0, np.int32)
        return (np.ctypeslib.as_array(cp, (m,)), np.ctypeslib.as_array(lp, (m,)), np.ctypeslib.as_array(hp, (m,)))

    # ------------------------------------------------------------------ image / device
    def serialize(self, seed_table=True):
        """The flat index image as a CPU uint8 tensor (header + sections).  seed_table=False leaves out the K-mer hash
        table (GENIE_IMAGE_NO_SEED_TABLE): the image of a rank that only finds SMEMs."""
        if not seed_table:
            nbytes = N.lib().genie_index_image_bytes(self._h, N.IMAGE_NO_SEED_TABLE)
            if nbytes <= 0:
                raise N.GenieError(int(nbytes), "genie_index_image_bytes")
            buf = torch.empty(nbytes, dtype=torch.uint8)
            N.check(N.lib().genie_index_serialize_image(self._h, N.IMAGE_NO_SEED_TABLE, C.c_void_p(buf.data_ptr()), nbytes),
                    "genie_index_serialize_image")
            return buf
        if self._host_blob is None:
            nbytes = N.lib().genie_index_blob_bytes(self._h)
            if nbytes <= 0:
                raise N.GenieError(int(nbytes), "genie_index_blob_bytes")
            buf = torch.empty(nbytes, dtype=torch.uint8)
            N.check(N.lib().genie_index_serialize(self._h, C.c_void_p(buf.data_ptr()), nbytes), "genie_index_serialize")
            self._host_blob = buf
        return self._host_blob

    def to(self, device, seed_table=True):
        """Upload the image with torch and bind it (torch owns the device memory)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("GenieIndex.to: an MI355X device is required (no CPU fallback)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        host = self.serialize(seed_table)
        self._bind(host.to(device), host[:N.HEADER_BYTES])
        return self

    def _bind(self, dev_blob, host_header):
        host_header = host_header.contiguous()
        dev_index = dev_blob.device.index if dev_blob.device.type == "cuda" else -1
        N.check(N.lib().genie_index_open(C.c_void_p(host_header.data_ptr()), C.c_void_p(dev_blob.data_ptr()),
                                         dev_blob.numel(), dev_index, C.byref(self._h)), "genie_index_open")
        self.blob = dev_blob
        self.device = dev_blob.device
        if dev_blob.device.type == "cuda":
            # the header was checked by open; the contents (row numbers, entry links) on the device, once
            what = C.c_uint32(0)
            with torch.cuda.device(self.device):
                rc = N.lib().genie_index_validate(self._h, C.byref(what), _stream(self.device))
            if rc:
                self.blob = None
                raise N.GenieError(rc, f"genie_index_validate (sections {what.value:#x})")

    @classmethod
    def from_image(cls, dev_blob):
        """Open an image that already sits in device memory (e.g. received by broadcast)."""
        self = cls()
        self._bind(dev_blob, dev_blob[:N.HEADER_BYTES].cpu())
        return self

    # ------------------------------------------------------------------ hot path
    def _need_device(self):
        if self.blob is None or self.device is None or self.device.type != "cuda":
            raise RuntimeError("index is not on a GPU: call GenieIndex.to('cuda') first (no CPU fallback)")

    def _as_dev(self, t, dtype):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.ascontiguousarray(t))
        t = t.to(device=self.device, dtype=dtype)
        return t.contiguous()

    def sa_interval(self, pats, lens=None):
        """Batched exact_match_back_prop.  pats: [N, stride] uint8 codes; lens: [N] or None.
        Returns int32 [N, 2] (lo, hi); (-1,-1) absent; (-2,-2) code > 3."""
        self._need_device()
        pats = self._as_dev(pats, torch.uint8)
        if pats.dim() != 2:
            raise ValueError("pats must be [N, stride]")
        n_pat, stride = pats.shape
        fixed = stride
        if lens is not None:
            lens = self._as_dev(lens, torch.int32)
            fixed = int(lens.max().item()) if n_pat else 0
            if fixed > stride or (n_pat and int(lens.min().item()) < 0):
                raise ValueError("pattern length outside [0, stride]")
        if stride == 0:
            pats = torch.zeros((n_pat, 1), dtype=torch.uint8, device=self.device)
            stride = 1
        out = torch.empty((n_pat, 2), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().genie_sa_interval(self._h, _ptr(pats), _ptr(lens), n_pat, stride, fixed, _ptr(out),
                                              _stream(self.device)), "genie_sa_interval")
        return out

    def seed_lookup(self, mode, kmers, want_pred=False):
        """Batched K-mer seed lookup (LUT membership / RMI get_suffix_rmi). kmers: [N, K] codes."""
        self._need_device()
        kmers = self._as_dev(kmers, torch.uint8)
        if kmers.dim() != 2 or kmers.shape[1] != self.K:
            raise ValueError("kmers must be [N, K]")
        out = torch.empty((kmers.shape[0], 2), dtype=torch.int32, device=self.device)
        pred = torch.empty(kmers.shape[0], dtype=torch.float64, device=self.device) if want_pred else None
        with torch.cuda.device(self.device):
            N.check(N.lib().genie_seed_lookup(self._h, N.MODES[mode], _ptr(kmers), kmers.shape[0], _ptr(out),
                                              _ptr(pred), _stream(self.device)), "genie_seed_lookup")
        return (out, pred) if want_pred else out

    def find_smems_slots(self, mode, reads, lens=None, min_len=1, cap=None):
        """One launch of the SMEM kernel.  reads: [N, stride] uint8 codes on the device.
        Returns (counts int32[N], slots int32[N, cap, 4], status int32[N])."""
        self._need_device()
        reads = self._as_dev(reads, torch.uint8)
        if reads.dim() != 2:
            raise ValueError("reads must be [N, stride]")
        n_reads, stride = reads.shape
        fixed = stride
        if lens is not None:
            lens = self._as_dev(lens, torch.int32)
            fixed = int(lens.max().item()) if n_reads else 0
            if fixed > stride or (n_reads and int(lens.min().item()) < 0):
                raise ValueError("read length outside [0, stride]")
        if cap is None:
            cap = max(1, fixed)
        counts = torch.empty(n_reads, dtype=torch.int32, device=self.device)
        slots = torch.empty((n_reads, cap, 4), dtype=torch.int32, device=self.device)
        status = torch.empty(n_reads, dtype=torch.int32, device=self.device)
        if n_reads == 0 or stride == 0:
            counts.zero_()
            status.zero_()
            return counts, slots, status
        ws_bytes = int(N.lib().genie_find_smems_workspace_bytes(n_reads, fixed))
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().genie_find_smems(self._h, N.MODES[mode], _ptr(reads), _ptr(lens), n_reads, stride, fixed,
                                             int(min_len), _ptr(counts), _ptr(slots), cap, _ptr(status), _ptr(ws),
                                             ws_bytes, _stream(self.device)), "genie_find_smems")
        return counts, slots, status

    def compact(self, counts, slots, out=None):
        """Slotted output -> CSR: (offsets int64[N+1], smems int32[S, 4])."""
        self._need_device()
        n_reads, cap = slots.shape[0], slots.shape[1]
        offsets = torch.empty(n_reads + 1, dtype=torch.int64, device=self.device)
        tmp = torch.empty(max(int(N.lib().genie_compact_tmp_bytes(n_reads)), 16), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            st = _stream(self.device)
            if out is None:
                N.check(N.lib().genie_compact_smems(_ptr(counts), _ptr(slots), n_reads, cap, _ptr(offsets),
                                                    C.c_void_p(0), 0, _ptr(tmp), st), "genie_compact_smems")
                total = int(offsets[-1].item())
                out = torch.empty((max(total, 1), 4), dtype=torch.int32, device=self.device)
            N.check(N.lib().genie_compact_smems(_ptr(counts), _ptr(slots), n_reads, cap, _ptr(offsets), _ptr(out),
                                                out.shape[0], _ptr(tmp), st), "genie_compact_smems")
        return offsets, out

    def locate(self, intervals, sort=False):
        """Rows -> reference coordinates for a batch of SA intervals (ExactMatch.get_positions,
        ExactMatch.py:195-199): `intervals` is int32 [S, 2] (lo, hi) -- a sa_interval result -- or
        int32 [S, 4] (start, end, lo, hi) -- find_smems rows.  Returns (pos_offsets int64[S+1],
        positions int32[T]): 1-based positions in row order, or ascending per interval with sort=True
        (ExactMatch.exact_match's order, :174-192)."""
        self._need_device()
        iv = self._as_dev(intervals, torch.int32)
        if iv.dim() != 2 or iv.shape[1] not in (2, 4):
            raise ValueError("intervals must be [S, 2] (lo, hi) or [S, 4] (start, end, lo, hi)")
        iv = iv.contiguous()
        S, width = iv.shape
        offsets = torch.empty(S + 1, dtype=torch.int64, device=self.device)
        tmp_bytes = int(N.lib().genie_locate_tmp_bytes(S))
        tmp = torch.empty(max(tmp_bytes, 256), dtype=torch.uint8, device=self.device)
        base = iv.data_ptr() + (8 if width == 4 else 0)
        with torch.cuda.device(self.device):
            st = _stream(self.device)
            N.check(N.lib().genie_locate(self._h, C.c_void_p(base), width, S, _ptr(offsets), C.c_void_p(0), 0, _ptr(tmp),
                                         tmp_bytes, st), "genie_locate")
            total = int(offsets[-1].item())
            pos = torch.empty(max(total, 1), dtype=torch.int32, device=self.device)
            N.check(N.lib().genie_locate(self._h, C.c_void_p(base), width, S, _ptr(offsets), _ptr(pos), total, _ptr(tmp),
                                         tmp_bytes, st), "genie_locate")
        pos = pos[:total]
        if sort and total:
            # ascending inside every interval: one sort on (interval, position) keys -- torch plumbing
            seg = torch.repeat_interleave(torch.arange(S, device=self.device), offsets[1:] - offsets[:-1])
            key = seg * (int(self.info()["n"]) + 2) + pos.to(torch.int64)
            pos = pos[torch.argsort(key)]
        return offsets, pos

    def find_smems(self, mode, reads, lens=None, min_len=1, cap=None, rows_hint=None):
        """Batched SMEM discovery -> (offsets int64[N+1], smems int32[S,4] = (start,end,lo,hi), status).
        Uses the fused CSR entry point (genie_find_smems_csr) unless an explicit slot capacity is asked for."""
        self._need_device()
        reads = self._as_dev(reads, torch.uint8)
        if reads.dim() != 2:
            raise ValueError("reads must be [N, stride]")
        n_reads, stride = reads.shape
        fixed = stride
        if lens is not None:
            lens = self._as_dev(lens, torch.int32)
            fixed = int(lens.max().item()) if n_reads else 0
            if fixed > stride or (n_reads and int(lens.min().item()) < 0):
                raise ValueError("read length outside [0, stride]")
        if cap is not None or n_reads == 0 or stride == 0:
            counts, slots, status = self.find_smems_slots(mode, reads, lens, min_len, cap)
            offsets, out = self.compact(counts, slots)
            total = int(offsets[-1].item())
            return offsets, out[:total], status
        offsets = torch.empty(n_reads + 1, dtype=torch.int64, device=self.device)
        status = torch.empty(n_reads, dtype=torch.int32, device=self.device)
        ws_bytes = int(N.lib().genie_find_smems_workspace_bytes(n_reads, fixed))
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=self.device)
        cap_rows = int(rows_hint) if rows_hint else n_reads * max(8, fixed // 6)
        while True:
            rows = torch.empty((max(cap_rows, 1), 4), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                N.check(N.lib().genie_find_smems_csr(self._h, N.MODES[mode], _ptr(reads), _ptr(lens), n_reads, stride,
                                                     fixed, int(min_len), _ptr(offsets), _ptr(rows), rows.shape[0],
                                                     _ptr(status), _ptr(ws), ws_bytes, _stream(self.device)),
                        "genie_find_smems_csr")
            total = int(offsets[-1].item())
            if total <= rows.shape[0]:
                return offsets, rows[:total], status
            cap_rows = total                      # capacity guess too small: rerun with the exact size

    def find_smems_packed(self, mode, packed, max_len, lens=None, min_len=1, rows_hint=None, row_bytes=8):
        """genie_find_smems_packed (or, `row_bytes` = 6, genie_find_smems_packed6): 2-bit packed reads (packing.pack_reads;
        uint8 [N, stride] on the device) -> (counts8 uint8[N], status8 uint8[N], rows uint8[S, row_bytes], escapes int64[E, 2]);
        packing.unpack_rows turns them into the offsets / int32 rows of find_smems.  Reads of at most 255 bases."""
        if row_bytes not in (6, 8):
            raise ValueError("row_bytes is 6 or 8")
        entry = N.lib().genie_find_smems_packed if row_bytes == 8 else N.lib().genie_find_smems_packed6
        self._need_device()
        packed = self._as_dev(packed, torch.uint8)
        if packed.dim() != 2:
            raise ValueError("packed reads must be [N, stride]")
        n_reads, stride = packed.shape
        if lens is not None:
            lens = self._as_dev(lens, torch.int32)
        counts8 = torch.zeros(n_reads, dtype=torch.uint8, device=self.device)
        status8 = torch.zeros(n_reads, dtype=torch.uint8, device=self.device)
        totals = torch.zeros(2, dtype=torch.int64, device=self.device)
        if n_reads == 0:
            return counts8, status8, torch.zeros((0, row_bytes), dtype=torch.uint8, device=self.device), torch.zeros((0, 2), dtype=torch.int64, device=self.device)
        ws_bytes = int(N.lib().genie_find_smems_workspace_bytes(n_reads, int(max_len)))
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=self.device)
        cap_rows = int(rows_hint) if rows_hint else n_reads * max(8, int(max_len) // 6)
        cap_esc = 1024
        while True:
            rows8 = torch.empty((max(cap_rows, 1), row_bytes), dtype=torch.uint8, device=self.device)
            esc = torch.empty((max(cap_esc, 1), 2), dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                N.check(entry(self._h, N.MODES[mode], _ptr(packed), _ptr(lens), n_reads, stride,
                                                        int(max_len), int(min_len), _ptr(counts8), _ptr(status8), _ptr(rows8),
                                                        rows8.shape[0], _ptr(totals), _ptr(esc), esc.shape[0], _ptr(ws), ws_bytes,
                                                        _stream(self.device)), "genie_find_smems_packed")
            total, n_esc = (int(x) for x in totals.cpu())
            if total <= rows8.shape[0] and n_esc <= esc.shape[0]:
                return counts8, status8, rows8[:total], esc[:n_esc]
            cap_rows, cap_esc = max(cap_rows, total), max(cap_esc, n_esc)      # too small: rerun with the exact sizes

    def find_smems_host(self, mode, codes, lens=None, min_len=1):
        """Host arrays in, host arrays out, through the packed entry point (what SMEM.find_smems_* does with host inputs):
        codes uint8 [N, L] (L <= 255) -> (offsets int64[N+1], rows int32[S, 4], status int32[N]) as numpy arrays."""
        from . import packing
        codes = np.ascontiguousarray(codes, np.uint8)
        packed = torch.as_tensor(packing.pack_reads(codes))
        rb = 6 if self.n + 1 < (1 << 24) else 8                     # the 6-byte rows hold lo in 24 bits
        c8, s8, r8, esc = self.find_smems_packed(mode, packed, codes.shape[1], lens=lens, min_len=min_len, row_bytes=rb)
        offsets, rows = packing.unpack_rows(c8.cpu().numpy(), r8.cpu().numpy(), esc.cpu().numpy(), row_bytes=rb)
        return offsets, rows, s8.cpu().numpy().astype(np.int32)

    def set_option(self, option, value):
        N.check(N.lib().genie_index_set_option(self._h, int(option), int(value)), "genie_index_set_option")

    @staticmethod
    def workspace_shape(max_len):
        """Per-read rows of the find_smems workspace (bytes / counts), for traffic accounting."""
        out = (C.c_int32 * 4)()
        N.check(N.lib().genie_find_smems_workspace_rows(int(max_len), out), "genie_find_smems_workspace_rows")
        return {"fwd_stride": out[0], "qp_recs": out[1], "kj_row_bytes": out[3]}

    def search_kernel_name(self, mode, max_len):
        buf = C.create_string_buffer(160)
        N.check(N.lib().genie_search_kernel_name(self._h, N.MODES[mode], int(max_len), buf, 160), "genie_search_kernel_name")
        return buf.value.decode()

    def launch_info(self, mode, max_len):
        g, b, l = C.c_int32(), C.c_int32(), C.c_int32()
        N.check(N.lib().genie_launch_info(self._h, N.MODES[mode], int(max_len), C.byref(g), C.byref(b), C.byref(l)),
                "genie_launch_info")
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}

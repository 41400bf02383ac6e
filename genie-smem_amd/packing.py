"""Host side of genie_find_smems_packed (include/genie_smem.h): what crosses the host link.

Reads travel 2-bit packed (a quarter of the bytes of one code per base) and results come back as 8-byte rows plus a
count byte and a status byte per read (half the bytes of the int32 rows and int64 offsets).  pack_reads() and
unpack_rows() are pure layout conversions; unpack_rows() returns exactly the (offsets, int32 (start, end, lo, hi) rows)
that genie_find_smems_csr produces.
"""
import numpy as np

ROW8 = np.dtype([("start", "u1"), ("end", "u1"), ("span", "<u2"), ("lo", "<u4")])


def packed_stride(max_len):
    """Bytes per packed read row: a multiple of 4, at least 4 * ceil(max_len / 16)."""
    return 4 * max(1, (int(max_len) + 15) // 16)


def pack_reads(codes, out=None):
    """codes: uint8 [N, L] base codes 0..3 -> uint8 [N, packed_stride(L)]: byte i = bases 4i .. 4i+3, base 4i in bits 7..6.
    A code > 3 raises KeyError (the reference's error for a base outside its alphabet, ExactMatch.py:139)."""
    codes = np.ascontiguousarray(codes, np.uint8)
    if codes.ndim != 2:
        raise ValueError("reads must be [N, L]")
    n, L = codes.shape
    if codes.size and int(codes.max()) > 3:
        raise KeyError("base outside the reference alphabet")
    stride = packed_stride(L)
    if out is None:
        out = np.zeros((n, stride), np.uint8)
    else:
        out[...] = 0
    full = L // 4
    if full:
        q = codes[:, :4 * full].reshape(n, full, 4)
        out[:, :full] = (q[:, :, 0] << 6) | (q[:, :, 1] << 4) | (q[:, :, 2] << 2) | q[:, :, 3]
    for j in range(L - 4 * full):                       # the last, partial byte
        out[:, full] |= codes[:, 4 * full + j] << (6 - 2 * j)
    return out


def unpack_reads(packed, length):
    """Inverse of pack_reads (tests)."""
    packed = np.asarray(packed, np.uint8)
    n = packed.shape[0]
    b = np.stack([(packed >> 6) & 3, (packed >> 4) & 3, (packed >> 2) & 3, packed & 3], axis=2).reshape(n, -1)
    return np.ascontiguousarray(b[:, :length])


def unpack_rows(counts8, rows8, escapes=None):
    """counts8: uint8 [N]; rows8: the dense 8-byte rows (any array of >= sum(counts8) * 8 bytes); escapes: int64 [E, 2]
    (row index, hi) for rows whose span field is 0xFFFF.  Returns (offsets int64 [N + 1], rows int32 [S, 4])."""
    counts8 = np.asarray(counts8, np.uint8)
    offsets = np.zeros(len(counts8) + 1, np.int64)
    np.cumsum(counts8, out=offsets[1:])
    total = int(offsets[-1])
    r = np.frombuffer(np.ascontiguousarray(rows8).view(np.uint8).reshape(-1)[:8 * total].tobytes(), ROW8)
    rows = np.empty((total, 4), np.int32)
    rows[:, 0] = r["start"]
    rows[:, 1] = r["end"]
    rows[:, 2] = r["lo"].astype(np.int64)
    rows[:, 3] = r["lo"].astype(np.int64) + r["span"]
    wide = np.nonzero(r["span"] == 0xFFFF)[0]
    if len(wide):
        if escapes is None:
            raise ValueError("rows with span 0xFFFF need the escape list")
        esc = np.asarray(escapes, np.int64).reshape(-1, 2)
        esc = esc[esc[:, 0] < total]
        order = np.argsort(esc[:, 0], kind="stable")
        esc = esc[order]
        if not np.array_equal(esc[:, 0], wide):
            raise ValueError("escape list does not cover the rows marked 0xFFFF")
        rows[wide, 3] = esc[:, 1]
    return offsets, rows

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


def unpack_rows(counts8, rows8, escapes=None, row_bytes=8):
    """counts8: uint8 [N]; rows8: the dense 8-byte rows of genie_find_smems_packed, or the 6-byte rows of
    genie_find_smems_packed6 (`row_bytes` = 6) -- any array of >= sum(counts8) * row_bytes bytes; escapes: int64 [E, 2]
    (row index, hi) for rows whose span field is saturated (0xFFFF / 0xFF).  Returns (offsets int64 [N + 1], rows int32 [S, 4])."""
    if row_bytes not in (6, 8):
        raise ValueError("row_bytes is 6 or 8")
    counts8 = np.asarray(counts8, np.uint8)
    offsets = np.zeros(len(counts8) + 1, np.int64)
    np.cumsum(counts8, out=offsets[1:])
    total = int(offsets[-1])
    raw = np.ascontiguousarray(rows8).view(np.uint8).reshape(-1)[:row_bytes * total]
    rows = np.empty((total, 4), np.int32)
    if row_bytes == 8:
        r = np.frombuffer(raw.tobytes(), ROW8)
        lo, span, top = r["lo"].astype(np.int64), r["span"].astype(np.int64), 0xFFFF
        rows[:, 0] = r["start"]
        rows[:, 1] = r["end"]
    else:
        b = raw.reshape(total, 6).astype(np.int64)
        lo, span, top = b[:, 2] | b[:, 3] << 8 | b[:, 4] << 16, b[:, 5], 0xFF
        rows[:, 0] = b[:, 0]
        rows[:, 1] = b[:, 1]
    rows[:, 2] = lo
    rows[:, 3] = lo + span
    wide = np.nonzero(span == top)[0]
    if len(wide):
        if escapes is None:
            raise ValueError("rows with a saturated span need the escape list")
        esc = np.asarray(escapes, np.int64).reshape(-1, 2)
        esc = esc[esc[:, 0] < total]
        order = np.argsort(esc[:, 0], kind="stable")
        esc = esc[order]
        if not np.array_equal(esc[:, 0], wide):
            raise ValueError("escape list does not cover the rows with a saturated span")
        rows[wide, 3] = esc[:, 1]
    return offsets, rows

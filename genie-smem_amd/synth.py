"""Seeded synthetic inputs of the shapes BASELINE.json names: i.i.d. uniform
references and the two read distributions of the reference's own generators
(SMEM/SMEM.py:489-505).  Used by bench.py and by the tests; no reference code involved.
"""
import numpy as np


def synth_ref(n, seed):
    """REF_100K = synth_ref(100_000, 100_000); REF_1M = synth_ref(1_000_000, 1_000_000)."""
    return np.random.default_rng(seed).integers(0, 4, n).astype(np.uint8)


def reads_random(n_reads, length, seed):
    """create_random_query distribution: i.i.d. uniform bases."""
    return np.random.default_rng(seed).integers(0, 4, (n_reads, length)).astype(np.uint8)


def reads_from_ref(ref_codes, n_reads, length, seed):
    """create_query_from_ref distribution, one read at a time (same stream as the golden
    generator): segments ref[p:p+s], p~U{0..n-1}, s~U{1..30}, rejected if p+s>n, truncated."""
    rng = np.random.default_rng(seed)
    n = len(ref_codes)
    out = np.empty((n_reads, length), np.uint8)
    for r in range(n_reads):
        buf, have = [], 0
        while have < length:
            p = int(rng.integers(0, n))
            s = int(rng.integers(1, 31))
            if p + s > n:
                continue
            buf.append(ref_codes[p:p + s])
            have += s
        out[r] = np.concatenate(buf)[:length]
    return out


def reads_from_ref_fast(ref_codes, n_reads, length, seed, chunk=65536):
    """Same distribution, vectorised (a different but equally seeded random stream): for the
    10^6..10^8-read configurations."""
    rng = np.random.default_rng(seed)
    n = len(ref_codes)
    out = np.empty((n_reads, length), np.uint8)
    nseg = length                      # every segment has >= 1 base, so `length` segments always suffice
    nseg = min(nseg, max(16, length // 4))
    for c0 in range(0, n_reads, chunk):
        m = min(chunk, n_reads - c0)
        p = rng.integers(0, n, (m, nseg))
        s = rng.integers(1, 31, (m, nseg))
        bad = p + s > n
        while bad.any():               # rejection, as the reference does
            k = int(bad.sum())
            p[bad] = rng.integers(0, n, k)
            s[bad] = rng.integers(1, 31, k)
            bad = p + s > n
        cum = np.cumsum(s, axis=1)
        short = cum[:, -1] < length
        j = np.arange(length)
        big = np.int64(1) << 20                                            # row separator for one flat search
        rows = np.arange(m, dtype=np.int64)[:, None] * big
        seg = np.searchsorted((cum + rows).ravel(), (j[None, :] + rows).ravel(), side="right").reshape(m, length)
        seg = np.minimum(seg - np.arange(m)[:, None] * nseg, nseg - 1)      # segment of each base
        start = np.take_along_axis(cum - s, seg, 1)
        pos = np.take_along_axis(p, seg, 1) + (j[None, :] - start)
        out[c0:c0 + m] = ref_codes[np.minimum(pos, n - 1)]
        for r in np.nonzero(short)[0]:                                      # (practically never)
            out[c0 + r] = reads_from_ref(ref_codes, 1, length, int(rng.integers(0, 2 ** 31)))[0]
    return out


def reads_from_ref_device(ref_codes, n_reads, length, seed, device=None, chunk=1 << 20):
    """The create_query_from_ref distribution generated with torch ON `device` (a third, equally seeded stream):
    10^7 .. 10^8-read batches in seconds instead of minutes of host time.  `ref_codes`: uint8 tensor (any device) or
    numpy array.  Returns a uint8 tensor [n_reads, length] on `device`.  Plumbing only: inputs, no arithmetic of
    the path."""
    import torch
    if not isinstance(ref_codes, torch.Tensor):
        ref_codes = torch.as_tensor(np.ascontiguousarray(ref_codes))
    device = torch.device(device) if device is not None else ref_codes.device
    ref = ref_codes.to(device)
    n = int(ref.numel())
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    nseg = min(length, max(16, length // 4))      # every segment has >= 1 base; a shortfall is redrawn below
    out = torch.empty((n_reads, length), dtype=torch.uint8, device=device)
    j = torch.arange(length, device=device)

    def draw(m):
        p = torch.randint(0, n, (m, nseg), generator=gen, device=device)
        s = torch.randint(1, 31, (m, nseg), generator=gen, device=device)
        bad = p + s > n
        while bool(bad.any()):                    # rejection, as the reference does (SMEM.py:496-505)
            k = int(bad.sum())
            p[bad] = torch.randint(0, n, (k,), generator=gen, device=device)
            s[bad] = torch.randint(1, 31, (k,), generator=gen, device=device)
            bad = p + s > n
        return p, s

    for c0 in range(0, n_reads, chunk):
        m = min(chunk, n_reads - c0)
        p, s = draw(m)
        cum = s.cumsum(1)
        short = cum[:, -1] < length
        while bool(short.any()):                  # (practically never: the segments of a row sum to ~15 x nseg)
            p2, s2 = draw(int(short.sum()))
            p[short], s[short] = p2, s2
            cum = s.cumsum(1)
            short = cum[:, -1] < length
        seg = torch.searchsorted(cum, j[None, :].expand(m, length).contiguous(), right=True).clamp_(max=nseg - 1)
        start = torch.gather(cum - s, 1, seg)
        pos = torch.gather(p, 1, seg) + (j[None, :] - start)
        out[c0:c0 + m] = ref[pos.clamp_(max=n - 1)]
    return out

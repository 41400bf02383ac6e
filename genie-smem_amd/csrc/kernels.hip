// kernels.hip -- gfx950 (MI355X, wave64) kernels of the batched SMEM finder and their launch code.
//
// Device building blocks (this file): packed 32-base windows, the P-mer prefix directory with its
// exact tail corrections, suffix comparison on inline-key suffix-array records, interval search,
// K-mer hash table probe, RMI prediction + last-mile search.
// The read pipeline K_A (match statistics, one wave per read) -> K_B (traversal, one lane per read)
// -> K_C (intervals, one lane per SMEM) lives in short_read_kernel.inc; the single-step kernels
// (batched exact match, seed lookup) and the offsets scan / compaction are below.
// Pure integer / indexing work (one fp64 multiply-add per RMI level); no MFMA.
//
// The traversal is the *reduced form* of the reference's code; the
// CPU oracle under oracle/ keeps the reference's original shape, so the two check each other.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

#include "genie_internal.h"

namespace genie {

namespace {

constexpr int kWave = 64;

// ------------------------------------------------------------------ small wave helpers
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS traffic between lanes of ONE wave: the LDS executes a wave's instructions in order, so
// only the compiler has to be stopped from reordering.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// wave max in 6 DPP steps (row_shr 1,2,4,8, row_bcast 15/31), result read from lane 63: no LDS crossbar
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ------------------------------------------------------------------ packed windows
// 32 bases starting at base offset `pos`: base j of the window sits in bits [62-2j, 63-2j].
__device__ __forceinline__ uint64_t funnel(uint64_t w0, uint64_t w1, int sh /* 0..62, even */)
{
    return sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0;
}

__device__ __forceinline__ uint64_t qwin(const uint64_t *qp, int pos)
{
    int i = pos >> 5;
    return funnel(qp[i], qp[i + 1], (pos & 31) * 2);
}

__device__ __forceinline__ uint64_t rwin(const RefRec *ref, int pos)
{
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(ref + (pos >> 5));   // one 16-byte load
    return funnel(v.x, v.y, (pos & 31) * 2);
}

// Where a packed query lives.  QWords: 8-byte words (LDS scratch of the wave that packed it).
// QRecs: the 16-byte overlapping records {w[i], w[i+1]} that K_A writes to global memory for
// K_B -- every 32-base window is ONE 16-byte ALIGNED load.  (Two adjacent 8-byte global loads
// get merged by the compiler into a dwordx4 at an 8-byte-aligned address; on MI355X that form
// returned wrong window words under load in the lane-per-read kernel -- found by the 1M-read
// parity test -- so global windows are always fetched as aligned records.)
struct QWords {
    const uint64_t *p;
    __device__ __forceinline__ uint64_t win(int pos) const { return qwin(p, pos); }
};
struct QRecs {
    const RefRec *p;
    __device__ __forceinline__ uint64_t win(int pos) const { return rwin(p, pos); }
};

// ------------------------------------------------------------------ prefix directory (LDS)
// dir[x] = number of SA rows whose suffix is smaller than the P-mer string x.  Rows whose
// suffix has fewer than P bases ("tails", P-1 of them plus the '$' row) need the two
// corrections below; padtail[l] is the A-padded code of the tail of length l.
//
// The directory as 16-bit deltas against every 16th entry (36 KB instead of 64 KB of LDS):
// dir[x] = coarse[x >> 4] + delta[x].  Usable when the image carries kFlagDir16.
struct Dir16 {
    const uint32_t *coarse;
    const uint16_t *delta;
    __device__ __forceinline__ uint32_t operator[](uint32_t x) const { return coarse[x >> 4] + delta[x]; }
};

// Rows with prefix `code` (m bases, 1 <= m <= P) are exactly [dir_lb, dir_ub).
template <class D>
__device__ __forceinline__ uint32_t dir_lb(const DevIndex &ix, const D dir, uint32_t code, int m)
{
    const uint32_t x = code << (2 * (ix.P - m));
    uint32_t v = dir[x];
    // tails  pat + A^t + '$'  (length l >= m) are >= pat but were counted as < x
#pragma unroll
    for (int l = 0; l < GENIE_MAX_DIR_BITS; l++)
        if (l >= m && l < ix.P && ix.padtail[l] == x) v--;
    return v;
}

template <class D>
__device__ __forceinline__ uint32_t dir_ub(const DevIndex &ix, const D dir, uint32_t code, int m)
{
    const uint32_t x = (code + 1) << (2 * (ix.P - m));
    uint32_t v = dir[x];
    // every tail sitting just below the next prefix is outside this prefix's rows
#pragma unroll
    for (int l = 1; l < GENIE_MAX_DIR_BITS; l++)
        if (l < ix.P && ix.padtail[l] == x) v--;
    return v;
}

// ------------------------------------------------------------------ suffix comparison
struct Cmp {
    int l;        // common prefix length of pattern q[a : a+m) and the suffix, capped at m
    bool less;    // suffix < pattern  ('$' smallest; a suffix that has the pattern as prefix is not less)
};

// Compare q[a : a+m) with the reference suffix starting at 0-based s, the first `skip` bases
// being known equal (skip <= min(m, n - s)).
template <class Q>
__device__ __forceinline__ Cmp cmp_suffix(const DevIndex &ix, const Q qp, int a, int m, int s, int skip)
{
    const int avail = ix.n - s;
    const int lim = m < avail ? m : avail;
    int l = skip;
    bool less = false;
    bool diff = false;
    while (l < lim) {
        const uint64_t xq = qp.win(a + l), xr = rwin(ix.ref, s + l);
        const uint64_t x = xq ^ xr;
        if (x) {
            l += __clzll((long long)x) >> 1;
            less = xr < xq;
            diff = true;
            break;
        }
        l += 32;
    }
    if (!diff || l >= lim) {      // ran off the pattern (prefix match) or off the reference ('$')
        l = lim;
        less = lim < m;
    }
    return {l, less};
}

// The same comparison against a suffix-array record, the first P bases known equal (every row
// of a directory bucket): the inline key decides the next 32 bases without touching the reference.
// `xq` = qp.win(a + P), hoisted by the caller.
template <class Q>
__device__ __forceinline__ Cmp cmp_rec(const DevIndex &ix, const Q qp, int a, int m, const SaRec rec, uint64_t xq)
{
    const int avail = ix.n - rec.s;
    const int lim = m < avail ? m : avail;
    int l = ix.P;
    bool less;
    const uint64_t x = xq ^ rec.key;
    if (x) {
        l += __clzll((long long)x) >> 1;
        less = rec.key < xq;
        if (l >= lim) { l = lim; less = lim < m; }
    } else if (l + 32 >= lim) {
        l = lim;
        less = lim < m;
    } else {                                   // more than P + 32 equal bases: continue in the reference
        return cmp_suffix(ix, qp, a, m, rec.s, l + 32);
    }
    return {l, less};
}

// Probe position inside the open row range [lo, hi).  Rows of a directory bucket share their first P
// bases, and their inline keys (the next 32 bases) are close to uniformly distributed, so the position
// of the pattern's key `xq32` between the key bounds already seen (`vlo`, `vhi`: top 32 bits) predicts
// the row -- a learned-index step at bucket scale.  Every other step is a plain bisection, which keeps
// the worst case logarithmic on repeat-rich references (many equal keys).  Only the CHOICE of the
// probe is heuristic: correctness rests on the lo/hi updates of the caller.
__device__ __forceinline__ int pick_probe(int lo, int hi, uint32_t xq32, uint32_t vlo, uint32_t vhi, int step)
{
    const int n = hi - lo;
    if ((step & 1) || n <= 2 || vhi <= vlo) return (lo + hi) >> 1;
    if (xq32 <= vlo) return lo;
    if (xq32 >= vhi) return hi - 1;
    const float f = (float)(xq32 - vlo) / (float)(vhi - vlo);
    int mid = lo + (int)(f * (float)n);
    mid = mid < lo ? lo : (mid > hi - 1 ? hi - 1 : mid);
    return mid;
}

__device__ __forceinline__ SaRec load_rec(const SaRec *sa, int row)
{
    int4 v = *reinterpret_cast<const int4 *>(sa + row);                // one 16-byte load
    // all four dwords are declared live: otherwise hipcc narrows the access to a dword + a dwordx2
    // load (the pad word is unused), i.e. TWO vector-memory instructions per probe
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    SaRec r;
    r.s = v.x;
    r.pad = 0;
    r.key = ((uint64_t)(uint32_t)v.w << 32) | (uint32_t)v.z;
    return r;
}

// ------------------------------------------------------------------ seeds
// LUT mode: `encoded_sub in self.lut.lut` + `self.lut.lut[encoded_sub][0]` (SMEM.py:65-70) as an
// open-addressing probe of the K-mer hash table (one 16-byte slot per probe).
__device__ __forceinline__ bool lut_probe(const DevIndex &ix, uint32_t code, int &lo, int &hi)
{
    uint32_t p = (uint32_t)(((uint64_t)(code * 0x9E3779B1u) * (uint64_t)ix.lut_slots) >> 32);
    for (;;) {
        const int4 v = *reinterpret_cast<const int4 *>(ix.lut + p);
        if (v.y < 0) return false;
        if ((uint32_t)v.x == code) { lo = v.y; hi = v.z; return true; }
        p = p + 1 == ix.lut_slots ? 0 : p + 1;
    }
}

// RMI.predict for one key (SMEM/RMI.py:52-69): per level p = coef*x + intercept, rounded after
// the multiply and after the add (sklearn computes X @ coef_ + intercept_), next expert =
// min(scale-1, max(0, int(p))).  `leaf` = the first `leaf_cnt` models of the last level staged in LDS (or nullptr).
__device__ __forceinline__ double rmi_predict(const DevIndex &ix, const RmiModel *leaf, int leaf_cnt, uint32_t code,
                                              int *leaf_idx = nullptr)
{
    const double x = (double)code;
    double p = 0.0;
    int idx = 0;
    for (int l = 0; l < ix.nlev; l++) {
        if (leaf_idx && l == ix.nlev - 1) *leaf_idx = idx;
        double2 m;
        if (leaf && l == ix.nlev - 1 && idx < leaf_cnt) m = *reinterpret_cast<const double2 *>(leaf + idx);   // LDS
        else m = *reinterpret_cast<const double2 *>(ix.rmi + ix.rmi_off[l] + idx);               // global
        p = __dadd_rn(__dmul_rn(m.x, x), m.y);
        const int scale = ix.rmi_scale[l];
        idx = !(p > 0.0) ? 0 : (p >= (double)scale ? scale - 1 : (int)p);
    }
    return p;
}

// K-mer `code` against the suffix of SA row r, over K symbols with '$' smallest:
// -1 suffix < kmer, 0 kmer is a prefix of the suffix, +1 suffix > kmer.
__device__ __forceinline__ int kmer_cmp_row(const DevIndex &ix, int r, uint32_t code)
{
    const int s = ix.sa[r].s;
    const int avail = ix.n - s;
    if (avail == 0) return -1;
    const uint64_t w = rwin(ix.ref, s);
    if (avail >= ix.K) {
        const uint32_t rc = (uint32_t)(w >> (64 - 2 * ix.K));
        return rc < code ? -1 : (rc > code ? 1 : 0);
    }
    const uint32_t rc = (uint32_t)(w >> (64 - 2 * avail));
    const uint32_t qc = code >> (2 * (ix.K - avail));
    return rc <= qc ? -1 : 1;          // equal so far: the suffix ends ('$') first
}

// RMI_LUT.get_suffix_rmi (SMEM/RMI_LUT.py:67-78): predict a row, then the last-mile search --
// gallop out from the predicted row to bracket the K-mer, then bound-search inside the bracket.
// Contract behaviour: always the TRUE interval; hit <=> lo <= hi.
__device__ __forceinline__ bool rmi_lookup(const DevIndex &ix, const RmiModel *leaf, uint32_t code, int &lo, int &hi,
                                           double *pred_out)
{
    int leaf_idx = 0;
    const double p = rmi_predict(ix, leaf, 0, code, &leaf_idx);
    if (pred_out) *pred_out = p;
    const int rows = ix.n + 1;
    const int r0 = !(p > 0.0) ? 0 : (p >= (double)rows ? rows - 1 : (int)p);     // int(start_sa), clamped
    // bracket the first row that is not smaller than the K-mer: rows < L are smaller, row R is not
    int L, R;
    bool fenced = false;
    if (ix.rmi_err) {
        // natively trained model: every row of a K-mer that occurs lies within the leaf's error bound of
        // the prediction, so one window replaces the doubling probes.  A K-mer that does not occur may
        // fall outside it; that case is recognised below and redone with the galloping bracket.
        const int e = ix.rmi_err[leaf_idx];
        L = r0 - e < 0 ? 0 : r0 - e;
        R = r0 + e + 1 > rows ? rows : r0 + e + 1;
        int l2 = L, r2 = R;
        while (l2 < r2) {
            const int mid = (l2 + r2) >> 1;
            if (kmer_cmp_row(ix, mid, code) < 0) l2 = mid + 1; else r2 = mid;
        }
        if (l2 < rows && kmer_cmp_row(ix, l2, code) == 0 && (l2 > L || L == 0 || kmer_cmp_row(ix, l2 - 1, code) < 0)) {
            L = R = l2;
            fenced = true;
        }
    }
    if (!fenced) {
    if (kmer_cmp_row(ix, r0, code) < 0) {
        L = r0 + 1;
        R = rows;
        for (int step = 1;; step <<= 1) {
            const int pr = r0 + step;
            if (pr >= rows) break;
            if (kmer_cmp_row(ix, pr, code) < 0) L = pr + 1;
            else { R = pr; break; }
        }
    } else {
        R = r0;
        L = 0;
        for (int step = 1;; step <<= 1) {
            const int pr = r0 - step;
            if (pr < 0) break;
            if (kmer_cmp_row(ix, pr, code) < 0) { L = pr + 1; break; }
            R = pr;
        }
    }
    }
    while (L < R) {
        const int mid = (L + R) >> 1;
        if (kmer_cmp_row(ix, mid, code) < 0) L = mid + 1; else R = mid;
    }
    const int first = L;
    if (first >= rows || kmer_cmp_row(ix, first, code) != 0) { lo = first; hi = first - 1; return false; }
    // last row that still has the K-mer as prefix: gallop right, then bisect
    int a = first, b = rows;                       // row a matches, row b does not (or is past the end)
    for (int step = 1;; step <<= 1) {
        const int pr = first + step;
        if (pr >= rows) break;
        if (kmer_cmp_row(ix, pr, code) == 0) a = pr;
        else { b = pr; break; }
    }
    while (b - a > 1) {
        const int mid = (a + b) >> 1;
        if (kmer_cmp_row(ix, mid, code) == 0) a = mid; else b = mid;
    }
    lo = first;
    hi = a;
    return true;
}

// ------------------------------------------------------------------ interval search
// Inclusive SA interval of q[a : a+m) (== ExactMatch.exact_match_back_prop of that substring);
// (-1,-1) if absent, (0,n) for the empty pattern.
template <class Q>
__device__ __forceinline__ int2 sa_interval(const DevIndex &ix, const uint32_t *dir, const Q qp, int a, int m)
{
    if (m == 0) return make_int2(0, ix.n);
    const int P = ix.P;
    const uint64_t w = qp.win(a);
    if (m <= P) {
        const uint32_t code = (uint32_t)(w >> (64 - 2 * m));
        const int lb = (int)dir_lb(ix, dir, code, m), ub = (int)dir_ub(ix, dir, code, m);
        return lb < ub ? make_int2(lb, ub - 1) : make_int2(-1, -1);
    }
    const uint64_t xq = qp.win(a + P);
    int lo, hi, h;
    int known = -1;                                     // a row known to carry the pattern as prefix
    if (ix.P2 && m >= ix.P2) {
        // second-level table: exact rows of the first P2 bases + the first row's inline key (first probe)
        int4 hd = *reinterpret_cast<const int4 *>(ix.dir2 + (uint32_t)(w >> (64 - 2 * ix.P2)));
        asm volatile("" : "+v"(hd.x), "+v"(hd.y), "+v"(hd.z), "+v"(hd.w));
        const uint32_t cnt = (uint32_t)hd.y & ~kHeadShort;
        if (cnt == 0) return make_int2(-1, -1);
        lo = hd.x;
        hi = h = hd.x + (int)cnt;
        const uint64_t key = ((uint64_t)(uint32_t)hd.w << 32) | (uint32_t)hd.z;
        const uint64_t x = xq ^ key;
        if (!((uint32_t)hd.y & kHeadShort) && (x != 0 || m <= P + 32)) {      // decided inside the key
            const int l = x ? P + (__clzll((long long)x) >> 1) : m;
            if (l >= m) { h = lo; known = lo; }                              // row lb matches: it is the lower bound
            else if (key < xq) lo = lo + 1;
            else h = lo;                                                      // row lb > pattern, no match: absent
        }
    } else {
        const uint32_t b = (uint32_t)(w >> (64 - 2 * P));
        lo = (int)dir[b];
        hi = h = (int)dir_ub(ix, dir, b, P);
    }
    uint32_t vlo = 0, vhi = 0xFFFFFFFFu;
    int step = 0;
    while (lo < h) {                                    // first row whose suffix is not < pattern
        const int mid = pick_probe(lo, h, (uint32_t)(xq >> 32), vlo, vhi, step++);
        const SaRec rec = load_rec(ix.sa, mid);
        const Cmp c = cmp_rec(ix, qp, a, m, rec, xq);
        if (c.less) { lo = mid + 1; vlo = (uint32_t)(rec.key >> 32); }
        else { h = mid; vhi = (uint32_t)(rec.key >> 32); known = c.l >= m ? mid : -1; }
    }
    const int first = lo;
    if (known == first) lo = first + 1;                 // already seen to match: start above it
    h = hi;
    // first row that no longer has the pattern as prefix: matching rows are adjacent to `first`
    // (usually one or two), so gallop up from it and bisect the last gap
    int stepw = 1;                                      // 0 once a non-matching row bounds the range: bisect
    while (lo < h) {
        int mid = stepw ? lo + stepw - 1 : (lo + h) >> 1;
        if (mid >= h) mid = (lo + h) >> 1;
        if (cmp_rec(ix, qp, a, m, load_rec(ix.sa, mid), xq).l >= m) { lo = mid + 1; stepw = stepw && stepw < (1 << 20) ? stepw << 1 : stepw; }
        else { h = mid; stepw = 0; }
    }
    return first < lo ? make_int2(first, lo - 1) : make_int2(-1, -1);
}

// ------------------------------------------------------------------ per-wave scratch in LDS (K1)
struct WaveScratch {
    uint64_t *qp;      // packed pattern, (Lmax+31)/32 + 2 words, zero padded
    uint8_t *raw;      // raw codes while packing
};

__host__ __device__ inline int scratch_qp_bytes(int Lmax) { return (((Lmax + 31) / 32 + 2) * 8 + 15) & ~15; }
__host__ __device__ inline int scratch_raw_bytes(int Lmax) { return (Lmax + 15) & ~15; }
__host__ __device__ inline int scratch_bytes(int Lmax) { return scratch_qp_bytes(Lmax) + scratch_raw_bytes(Lmax); }

__device__ __forceinline__ WaveScratch carve(uint8_t *base, int Lmax)
{
    WaveScratch s;
    s.qp = reinterpret_cast<uint64_t *>(base);
    s.raw = base + scratch_qp_bytes(Lmax);
    return s;
}

// coalesced byte loads, validation, 2-bit packing.  Returns false on a code > 3.
__device__ __forceinline__ bool load_and_pack(const uint8_t *src, int L, int Lmax, const WaveScratch &ws, int lane)
{
    uint8_t *raw = ws.raw;
    bool bad = false;
    for (int i = lane; i < L; i += kWave) {
        const uint8_t c = src[i];
        bad |= c > 3;
        raw[i] = c;
    }
    wave_lds_fence();
    const int nw = (L + 31) / 32 + 2;
    for (int wi = lane; wi < nw; wi += kWave) {
        uint64_t w = 0;
        const int b0 = wi * 32;
        const int cnt = L - b0 < 32 ? L - b0 : 32;
        for (int j = 0; j < cnt; j++) w |= (uint64_t)(raw[b0 + j] & 3) << (62 - 2 * j);
        ws.qp[wi] = w;
    }
    wave_lds_fence();
    return !__any(bad);
}

#include "short_read_kernel.inc"
#include "match_table_kernel.inc"

// ------------------------------------------------------------------ K1: batched exact_match_back_prop
__global__ void __launch_bounds__(256) sa_interval_kernel(DevIndex ix, const uint8_t *__restrict__ pats,
                                                          const int32_t *__restrict__ lens, long long N, int stride,
                                                          int fixed_len, int Lmax, int2 *__restrict__ out)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = rfl((int)(threadIdx.x >> 6));       // wave-uniform by construction
    const int waves_per_block = blockDim.x >> 6;
    uint32_t *dir = reinterpret_cast<uint32_t *>(smem);
    const int dir_bytes = (ix.dir_entries * 4 + 15) & ~15;
    for (int i = threadIdx.x; i < ix.dir_entries; i += blockDim.x) dir[i] = ix.dir[i];
    __syncthreads();
    const WaveScratch ws = carve(smem + dir_bytes + wave * scratch_bytes(Lmax), Lmax);
    for (long long r = (long long)blockIdx.x * waves_per_block + wave; r < N;
         r += (long long)gridDim.x * waves_per_block) {
        const int L = lens ? lens[r] : fixed_len;
        int2 iv = make_int2(-2, -2);
        if (L >= 0 && L <= Lmax && load_and_pack(pats + r * (long long)stride, L, Lmax, ws, lane))
            iv = sa_interval(ix, dir, QWords{ws.qp}, 0, L);  // every lane computes the same interval
        if (lane == 0) out[r] = iv;
        wave_lds_fence();
    }
}

// ------------------------------------------------------------------ batched seed lookup (A6 / A8)
template <int MODE>
__global__ void __launch_bounds__(256) seed_lookup_kernel(DevIndex ix, const uint8_t *__restrict__ kmers, long long N,
                                                          int2 *__restrict__ out, double *__restrict__ pred)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int K = ix.K;
    uint32_t code = 0;
    bool bad = false;
    for (int j = 0; j < K; j++) {
        const uint8_t c = kmers[i * K + j];
        bad |= c > 3;
        code = (code << 2) | (c & 3);
    }
    int lo = -1, hi = -1;
    double p = 0.0;
    if (bad) {
        lo = hi = -2;
    } else if (MODE == GENIE_MODE_LUT) {
        if (!lut_probe(ix, code, lo, hi)) lo = hi = -1;
    } else {
        rmi_lookup(ix, nullptr, code, lo, hi, &p);      // absent keeps the reference's lower > upper
    }
    out[i] = make_int2(lo, hi);
    if (pred) pred[i] = p;
}

// ------------------------------------------------------------------ compaction to CSR
constexpr int kScanBlock = 1024;

__global__ void __launch_bounds__(kScanBlock) compact_block_sums(const int32_t *__restrict__ counts, long long N, int cap,
                                                                 unsigned long long *__restrict__ block_sums)
{
    __shared__ unsigned long long wsum[kScanBlock / kWave];
    const long long i = (long long)blockIdx.x * kScanBlock + threadIdx.x;
    unsigned long long v = 0;                                  // 64-bit sums: interval sizes (genie_locate) can be large
    if (i < N) { const int c = counts[i]; v = (unsigned long long)(c < 0 ? 0 : (c < cap ? c : cap)); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int lo = (unsigned)__shfl_xor((int)(unsigned)v, off, kWave);
        const unsigned int hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), off, kWave);
        v += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < kScanBlock / kWave; w++) s += wsum[w];
        block_sums[blockIdx.x] = s;
    }
}

// single block: exclusive scan of the block sums in place (+ grand total at [nblocks])
__global__ void __launch_bounds__(kScanBlock) compact_scan_sums(unsigned long long *__restrict__ block_sums, long long nblocks)
{
    __shared__ unsigned long long part[kScanBlock];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long base = 0; base < nblocks; base += kScanBlock) {
        const long long i = base + threadIdx.x;
        const unsigned long long v = i < nblocks ? block_sums[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kScanBlock; off <<= 1) {           // Hillis-Steele inclusive scan
            unsigned long long add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nblocks) block_sums[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += part[kScanBlock - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nblocks] = carry;
}

__global__ void __launch_bounds__(kScanBlock) compact_scatter(const int32_t *__restrict__ counts,
                                                              const int4 *__restrict__ slots, long long N, int cap,
                                                              const unsigned long long *__restrict__ block_sums,
                                                              long long *__restrict__ offsets, int4 *__restrict__ out,
                                                              long long out_cap_rows, int *__restrict__ overflow)
{
    __shared__ unsigned long long part[kScanBlock];
    const long long i = (long long)blockIdx.x * kScanBlock + threadIdx.x;
    unsigned long long v = 0;
    if (i < N) { const int c = counts[i]; v = (unsigned long long)(c < 0 ? 0 : (c < cap ? c : cap)); }
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        unsigned long long add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    const unsigned long long base = block_sums[blockIdx.x] + part[threadIdx.x] - v;
    if (i < N) {
        offsets[i] = (long long)base;
        if (i == N - 1) offsets[N] = (long long)(base + v);
        if (!out) {
            /* offsets only */
        } else if ((long long)(base + v) > out_cap_rows) {
            *overflow = 1;
        } else {
            for (unsigned int t = 0; t < (unsigned int)v; t++) out[base + t] = slots[i * (long long)cap + t];
        }
    }
}

// ------------------------------------------------------------------ host side: launch plumbing
std::mutex g_err_mu;
std::string g_err;

#define HIP_TRY(expr)                                                         \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) { set_hip_error(#expr, (int)e_); return GENIE_E_HIP; } \
    } while (0)

struct Geometry {
    int grid, block, lds, leaf_in_lds;
    int blocks_a;    // paired search: blocks [0, blocks_a) take role A, the rest role B
    int sampled;     // sampled search (match_stats_sampled_kernel) with this work-list capacity, 0 = not used
    int grp, sshift; // its reads per wave iteration and log2 of the sampling stride
    int win;         // long reads: window of the sampled search (match_stats_sampled_long_kernel), 0 = not used
    int mt;          // match-table kernel (reads of at most 255 bases): slow-list capacity, 0 = not used
    uint32_t recip;  // its ceil(2^32 / max_len)
    int pair;        // narrow fixed-length batch whose last slot holds <= 32 positions: two reads per wave iteration
    int ns;          // position slots per chunk in K_A (1..4)
    int wide;        // reads longer than 255 bases: uint16 fwd[], K_B reads it from global memory
    int max_len;
    int fwd_stride;  // bytes per fwd[] row in the workspace
    int qp_words;    // packed-read words per wave in LDS (incl. 2 zero words)
    int qp_recs;     // 16-byte packed-read records per read in the workspace
    int hm_words;    // hit-mask words per read + 1 (longest match)
    int kj_row;      // emitted-pair entries per read
};

inline int fwd_row_bytes(int max_len, bool wide)
{
    if (wide) return ((max_len * 2) + 15) & ~15;
    int dw = (max_len + 3) / 4;
    if (dw < 1) dw = 1;
    if ((dw & 1) == 0) dw++;              // odd dword stride: conflict-free LDS rows in K_B
    return dw * 4;
}

inline void shape_for(int max_len, Geometry *g)
{
    g->max_len = max_len;
    g->wide = max_len > 255;
    g->ns = g->wide ? 4 : std::max(1, (max_len + 63) / 64);
    g->qp_words = g->wide ? (max_len + 31) / 32 + 2 : 2 * g->ns + 2;
    g->qp_recs = g->qp_words - 2;
    g->hm_words = (g->wide ? (max_len + 63) / 64 : g->ns) + 1;
    g->fwd_stride = fwd_row_bytes(max_len, g->wide);
    g->kj_row = (std::max(max_len, 1) + 7) & ~7;
}

inline int sample_shift()
{
    const char *e = std::getenv("GENIE_SAMPLE_SHIFT");
    const int v = e ? std::atoi(e) : 2;
    return v >= 1 && v <= 4 ? v : 2;
}

int plan_find_smems(const genie_index *ix, int mode, int max_len, bool fixed_len, long long N, Geometry *g)
{
    const DevIndex &d = ix->dev;
    const int dir_bytes = (d.dir_entries * 4 + 15) & ~15;
    const int lds_cap = 160 * 1024;
    shape_for(max_len, g);
    g->pair = !g->wide && fixed_len && max_len > 0 && max_len - 64 * (g->ns - 1) <= 32;
    const int per_wave = g->qp_words * 8 * (g->pair ? 2 : 1);     // a wave packs one or two reads per iteration
    // RMI leaf models: as many as fit beside the directory while TWO 16-wave blocks still share a CU's
    // LDS (experts [1000] = 16 000 B fit but for ~30 models); the rest are read from global memory.
    int leaf_bytes = 0;
    if (mode == GENIE_MODE_RMI) {
        const int cnt = d.rmi_off[d.nlev] - d.rmi_off[d.nlev - 1];
        int room = lds_cap / 2 - dir_bytes - 16 * per_wave;
        if (room < 0) room = 0;
        leaf_bytes = std::min(cnt, room / 16) * 16;
    }
    int waves = (lds_cap - dir_bytes - leaf_bytes) / per_wave;
    if (waves < 1) return GENIE_E_TOO_LONG;
    if (waves > 16) waves = 16;
    int w2 = 1;
    while (w2 * 2 <= waves) w2 *= 2;
    waves = w2;
    const int lds = dir_bytes + leaf_bytes + waves * per_wave;
    int blocks_per_cu = lds_cap / lds;
    if (blocks_per_cu * waves > 32) blocks_per_cu = 32 / waves;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    const int cus = ix->num_cus > 0 ? ix->num_cus : 256;
    long long grid = (long long)cus * blocks_per_cu;
    const long long per_block = (long long)waves * (g->pair ? 2 : 1);
    const long long need = (N + per_block - 1) / per_block;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    g->blocks_a = (int)grid;
    if (g->pair && g->ns > 1) {
        // two block roles (see match_stats_kernel); 48 % of the blocks in role A balances the two
        // loops for 150-base reads in all three modes (measured optimum 45..50 %)
        if (grid < 2) grid = 2;
        int pct = 48;
        if (const char *e = std::getenv("GENIE_PAIR_SPLIT")) { const int v = std::atoi(e); if (v > 0 && v < 100) pct = v; }
        long long a = (grid * pct + 50) / 100;
        a = a < 1 ? 1 : (a > grid - 1 ? grid - 1 : a);
        g->blocks_a = (int)a;
    }
    g->sampled = 0;
    g->win = 0;
    g->mt = 0;
    g->recip = 0;
    if (!g->wide && max_len > 0 && !ix->opt_legacy_search) {
        // match-table kernel: `grp` reads per wave iteration (about kMtTarget positions: a few passes of
        // 3 x 64), eight waves per block, nothing block-wide in LDS
        const int grp = std::min(std::min(kMtMaxG, kWave / mt_pack_dwords(max_len) * 4), std::max(1, (ix->opt_group_positions > 0 ? ix->opt_group_positions : kMtTarget) / max_len));
        const int wpb = 8;
        g->mt = grp * max_len;
        g->grp = grp;
        g->recip = max_len >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)max_len - 1) / (uint64_t)max_len) : 0u;
        g->lds = wpb * mt_wave_bytes(grp, max_len, g->qp_recs, g->fwd_stride);
        g->leaf_in_lds = 0;
        g->block = wpb * kWave;
        int bpc = std::min(lds_cap / g->lds, 32 / wpb);
        if (ix->opt_search_blocks_per_cu > 0) bpc = std::min(bpc, ix->opt_search_blocks_per_cu);
        if (bpc < 1) bpc = 1;
        long long gr = (long long)cus * bpc;
        const long long need2 = (N + (long long)wpb * grp - 1) / ((long long)wpb * grp);
        if (gr > need2) gr = need2;
        if (gr < 1) gr = 1;
        g->grid = (int)gr;
        return GENIE_OK;
    }
    if (g->wide && !ix->opt_legacy_search) {
        // long reads: one wave per read, eight waves per block
        const int wpb = 8;
        g->mt = 1;
        g->grp = 1;
        g->lds = wpb * mt_long_wave_bytes(max_len, g->qp_recs);
        g->leaf_in_lds = 0;
        g->block = wpb * kWave;
        int bpc = std::min(lds_cap / g->lds, 32 / wpb);
        if (ix->opt_search_blocks_per_cu > 0) bpc = std::min(bpc, ix->opt_search_blocks_per_cu);
        if (bpc < 1) bpc = 1;
        long long gr = (long long)cus * bpc;
        const long long need2 = (N + wpb - 1) / wpb;
        if (gr > need2) gr = need2;
        if (gr < 1) gr = 1;
        g->grid = (int)gr;
        return GENIE_OK;
    }
    if (g->wide && (d.flags & kFlagDir16) && !ix->opt_search_all) {
        // sampled search for long reads: one wave per read, windows of 704 positions
        const int win = 704;
        const int ncoarse = ((d.dir_entries - 1) >> 4) + 1;
        const int dir16 = ((ncoarse * 4 + 15) & ~15) + ((d.dir_entries * 2 + 15) & ~15);
        const int pw = g->qp_words * 8 + (((win + 8) * 2 + 15) & ~15) + ((win * 2 + 15) & ~15);
        int w = (lds_cap / 2 - dir16) / pw;
        int w2 = 1;
        while (w2 * 2 <= w && w2 < 16) w2 *= 2;
        if (w >= 4) {
            int leaf2 = 0;
            if (mode == GENIE_MODE_RMI) {
                const int cnt = d.rmi_off[d.nlev] - d.rmi_off[d.nlev - 1];
                int room = lds_cap / 2 - dir16 - w2 * pw;
                if (room < 0) room = 0;
                leaf2 = std::min(cnt, room / 16) * 16;
            }
            g->win = win;
            g->sshift = sample_shift();
            g->lds = dir16 + leaf2 + w2 * pw;
            g->leaf_in_lds = leaf2 / 16;
            g->block = w2 * kWave;
            long long gr = (long long)cus * 2;
            const long long need2 = (N + w2 - 1) / w2;
            if (gr > need2) gr = need2;
            if (gr < 1) gr = 1;
            g->grid = (int)gr;
            return GENIE_OK;
        }
    }
    if (!g->wide && (d.flags & kFlagDir16) && max_len > 0 && !ix->opt_search_all) {
        // sampled search: directory as 16-bit deltas; `grp` reads per wave with their fwd rows and a work
        // list: as many as keep two blocks per CU in LDS and fill, not overflow, one 192-entry chunk of
        // first-round searches
        const int sshift = sample_shift();
        const int ncoarse = ((d.dir_entries - 1) >> 4) + 1;
        const int dir16 = ((ncoarse * 4 + 15) & ~15) + ((d.dir_entries * 2 + 15) & ~15);
        const int ns0 = ((max_len - 1) >> sshift) + 1;
        int grp = std::min(8, std::max(1, 192 / ns0));
        if (const char *e = std::getenv("GENIE_SAMPLE_GROUP")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) grp = v; }   // tuning
        int wl_cap = 0, pw = 0;
        for (; grp >= 1; grp--) {
            wl_cap = grp * max_len;
            pw = grp * g->qp_words * 8 + 8 * 8 + 32 + ((grp * g->fwd_stride + 15) & ~15) + ((wl_cap * 2 + 15) & ~15);
            if (dir16 + 16 * pw <= lds_cap / 2) break;
        }
        if (grp < 1) grp = 1;
        int leaf2 = 0;
        if (mode == GENIE_MODE_RMI) {
            const int cnt = d.rmi_off[d.nlev] - d.rmi_off[d.nlev - 1];
            int room = lds_cap / 2 - dir16 - 16 * pw;
            if (room < 0) room = 0;
            leaf2 = std::min(cnt, room / 16) * 16;
        }
        const int lds2 = dir16 + leaf2 + 16 * pw;
        if (lds2 <= lds_cap / 2) {
            g->sampled = wl_cap;
            g->grp = grp;
            g->sshift = sshift;
            g->lds = lds2;
            g->leaf_in_lds = leaf2 / 16;
            g->block = 16 * kWave;
            long long gr = (long long)cus * 2;
            const long long need2 = (N + 16 * grp - 1) / (16 * grp);
            if (gr > need2) gr = need2;
            if (gr < 1) gr = 1;
            g->grid = (int)gr;
            return GENIE_OK;
        }
    }
    g->grid = (int)grid;
    g->block = waves * kWave;
    g->lds = lds;
    g->leaf_in_lds = leaf_bytes / 16;     // number of leaf models staged
    return GENIE_OK;
}

struct Workspace {
    uint8_t *fwd;        // N x fwd_stride bytes
    RefRec *qp;          // qp_recs records of 16 bytes per read
    int32_t *status;     // used when the caller passes no status array
    uint8_t *kj;         // N x kj_row emitted (start | end << shift) entries of 2 or 4 bytes
    unsigned long long *hm;  // N x hm_words: K-mer hit mask per 64 positions + longest match
    int32_t *counts;     // used by the CSR entry point
    uint8_t *scan_tmp;   // scratch of the offsets scan (CSR entry point)
};

inline int64_t ws_align(int64_t x) { return (x + 255) & ~(int64_t)255; }

inline int64_t workspace_bytes_for(int64_t N, int max_len)
{
    Geometry g;
    shape_for(max_len, &g);
    return ws_align(N * (int64_t)g.fwd_stride) + ws_align(N * (int64_t)g.qp_recs * 16) + ws_align(N * 4) +
           ws_align(N * (int64_t)g.kj_row * (g.wide ? 4 : 2)) + ws_align(N * (int64_t)g.hm_words * 8) + ws_align(N * 4) +
           ws_align(compact_tmp_bytes(N)) + 256;
}

inline int carve_workspace(void *d_ws, int64_t ws_bytes, int64_t N, const Geometry &g, Workspace *ws)
{
    if (!d_ws || ws_bytes < workspace_bytes_for(N, g.max_len) || (reinterpret_cast<uintptr_t>(d_ws) & 255) != 0)
        return GENIE_E_CAPACITY;
    uint8_t *p = reinterpret_cast<uint8_t *>(d_ws);
    ws->fwd = p;
    p += ws_align(N * (int64_t)g.fwd_stride);
    ws->qp = reinterpret_cast<RefRec *>(p);
    p += ws_align(N * (int64_t)g.qp_recs * 16);
    ws->status = reinterpret_cast<int32_t *>(p);
    p += ws_align(N * 4);
    ws->kj = p;
    p += ws_align(N * (int64_t)g.kj_row * (g.wide ? 4 : 2));
    ws->hm = reinterpret_cast<unsigned long long *>(p);
    p += ws_align(N * (int64_t)g.hm_words * 8);
    ws->counts = reinterpret_cast<int32_t *>(p);
    p += ws_align(N * 4);
    ws->scan_tmp = p;
    return GENIE_OK;
}

struct CsrOut {
    int64_t *offsets = nullptr;      // non-null => write CSR rows to `rows`, else slots
    int32_t *rows = nullptr;
    int64_t cap_rows = 0;
};

template <int MODE, int NS, bool WIDE>
int launch_pipeline(const genie_index *ix, const Geometry &g, const uint8_t *d_reads, const int32_t *d_lens, int64_t N,
                    int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts, int32_t *d_slots, int32_t cap,
                    int32_t *d_status, const Workspace &ws, const CsrOut &csr, hipStream_t s)
{
    int32_t *st = d_status ? d_status : ws.status;
    int32_t *cnt = d_counts ? d_counts : ws.counts;
    constexpr bool CANPAIR = !WIDE;
    // the K-mer hash probe (GENIE_OPT_LUT_PROBE) is compiled in only where it is asked for
    constexpr bool CANPROBE = MODE == GENIE_MODE_LUT;
    const bool probe = CANPROBE && ix->opt_lut_probe;
    auto ka = (CANPAIR && g.pair) ? (probe ? match_stats_kernel<MODE, NS, WIDE, CANPAIR, CANPROBE> : match_stats_kernel<MODE, NS, WIDE, CANPAIR, false>)
                                  : (probe ? match_stats_kernel<MODE, NS, WIDE, false, CANPROBE> : match_stats_kernel<MODE, NS, WIDE, false, false>);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds));
    if (ix->ev_search_begin) HIP_TRY(hipEventRecord((hipEvent_t)ix->ev_search_begin, s));
    if (!WIDE && g.mt) {
        auto km = match_table_kernel;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(km), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds));
        hipLaunchKernelGGL(km, dim3(g.grid), dim3(g.block), g.lds, s, ix->dev, (int)MODE, d_reads, d_lens, (long long)N, stride,
                           fixed_len, reinterpret_cast<uint8_t *>(ws.fwd), g.fwd_stride, ws.qp, g.qp_recs, ws.hm, g.hm_words, st,
                           g.grp, g.max_len, (long long)sizeof(MatchRec) * ix->dev.mtab_entries, ix->opt_debug);
    } else if (WIDE && g.mt) {
        auto km = match_table_long_kernel;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(km), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds));
        hipLaunchKernelGGL(km, dim3(g.grid), dim3(g.block), g.lds, s, ix->dev, (int)MODE, d_reads, d_lens, (long long)N, stride,
                           fixed_len, reinterpret_cast<uint8_t *>(ws.fwd), g.fwd_stride, ws.qp, g.qp_recs, ws.hm, g.hm_words, st,
                           g.max_len, (long long)sizeof(MatchRec) * ix->dev.mtab_entries);
    } else if (WIDE && g.win) {
        auto kl = probe ? match_stats_sampled_long_kernel<MODE, CANPROBE> : match_stats_sampled_long_kernel<MODE, false>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds));
        hipLaunchKernelGGL(kl, dim3(g.grid), dim3(g.block), g.lds, s, ix->dev, d_reads, d_lens, (long long)N, stride, fixed_len,
                           reinterpret_cast<uint8_t *>(ws.fwd), g.fwd_stride, ws.qp, g.qp_recs, g.qp_words, ws.hm, g.hm_words, st,
                           g.leaf_in_lds, g.sshift, g.win);
    } else if (!WIDE && g.sampled) {
        // five reads per wave is what 150-base reads get: that instantiation has the group size folded in
        auto ks = g.grp == 5 ? (probe ? match_stats_sampled_kernel<MODE, CANPROBE, 5> : match_stats_sampled_kernel<MODE, false, 5>)
                             : (probe ? match_stats_sampled_kernel<MODE, CANPROBE, 0> : match_stats_sampled_kernel<MODE, false, 0>);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds));
        hipLaunchKernelGGL(ks, dim3(g.grid), dim3(g.block), g.lds, s, ix->dev, d_reads, d_lens, (long long)N, stride, fixed_len,
                           reinterpret_cast<uint8_t *>(ws.fwd), g.fwd_stride, ws.qp, g.qp_recs, g.qp_words, ws.hm, g.hm_words, st,
                           g.leaf_in_lds, g.sampled, g.sshift, g.grp);
    } else
    hipLaunchKernelGGL(ka, dim3(g.grid), dim3(g.block), g.lds, s, ix->dev, d_reads, d_lens, (long long)N, stride,
                       fixed_len, ws.fwd, g.fwd_stride, ws.qp, g.qp_recs, g.qp_words, ws.hm, g.hm_words, st, g.leaf_in_lds,
                       g.blocks_a);
    HIP_TRY(hipGetLastError());
    if (ix->ev_search_end) HIP_TRY(hipEventRecord((hipEvent_t)ix->ev_search_end, s));
    if (ix->opt_search_only) return GENIE_OK;        // GENIE_OPT_SEARCH_ONLY: timing experiments, workspace only
    auto kb = traverse_kernel<MODE, WIDE>;
    const int tb = 256;
    const int lds_b = WIDE ? 0 : tb * g.fwd_stride;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
    hipLaunchKernelGGL(kb, dim3((unsigned)((N + tb - 1) / tb)), dim3(tb), lds_b, s, ix->dev, d_lens, (long long)N,
                       fixed_len, min_len, ws.fwd, g.fwd_stride, ws.qp, g.qp_recs, ws.hm, g.hm_words, cnt, ws.kj, g.kj_row,
                       csr.offsets ? g.kj_row : cap, st);
    HIP_TRY(hipGetLastError());
    if (csr.offsets) {                       // offsets = exclusive scan of the counts (no slots involved)
        int rc = launch_compact(cnt, nullptr, N, g.kj_row, csr.offsets, nullptr, 0, ws.scan_tmp, s);
        if (rc) return rc;
    }
    // K_C: intervals + final rows, 16 lanes per read, persistent blocks with the directory in LDS
    const int lds_c = (ix->dev.dir_entries * 4 + 15) & ~15;
    const int cus = ix->num_cus > 0 ? ix->num_cus : 256;
    int blocks_c = 160 * 1024 / lds_c;
    blocks_c = blocks_c > 2 ? 2 : (blocks_c < 1 ? 1 : blocks_c);
    long long grid_c = (long long)cus * blocks_c;
    const long long need_c = (N + 63) / 64;
    if (grid_c > need_c) grid_c = need_c;
    if (csr.offsets) {
        auto kc = interval_kernel<true, WIDE>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, lds_c));
        hipLaunchKernelGGL(kc, dim3((unsigned)grid_c), dim3(1024), lds_c, s, ix->dev, (long long)N, cnt, ws.kj, g.kj_row,
                           ws.qp, g.qp_recs, reinterpret_cast<int4 *>(csr.rows), 0,
                           reinterpret_cast<const long long *>(csr.offsets), (long long)csr.cap_rows);
    } else {
        auto kc = interval_kernel<false, WIDE>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, lds_c));
        hipLaunchKernelGGL(kc, dim3((unsigned)grid_c), dim3(1024), lds_c, s, ix->dev, (long long)N, cnt, ws.kj, g.kj_row,
                           ws.qp, g.qp_recs, reinterpret_cast<int4 *>(d_slots), cap, nullptr, 0ll);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

template <int MODE>
int launch_find_mode(const genie_index *ix, const Geometry &g, const uint8_t *d_reads, const int32_t *d_lens, int64_t N,
                     int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts, int32_t *d_slots,
                     int32_t cap, int32_t *d_status, void *d_ws, int64_t ws_bytes, const CsrOut &csr, hipStream_t s)
{
    Workspace ws;
    int rc = carve_workspace(d_ws, ws_bytes, N, g, &ws);
    if (rc) return rc;
#define GENIE_PIPE(NS_, WIDE_)                                                                                        \
    return launch_pipeline<MODE, NS_, WIDE_>(ix, g, d_reads, d_lens, N, stride, fixed_len, min_len, d_counts, d_slots, \
                                             cap, d_status, ws, csr, s)
    if (g.wide) GENIE_PIPE(4, true);
    switch (g.ns) {
    case 1: GENIE_PIPE(1, false);
    case 2: GENIE_PIPE(2, false);
    case 3: GENIE_PIPE(3, false);
    case 4: GENIE_PIPE(4, false);
    }
#undef GENIE_PIPE
    return GENIE_E_INVALID;
}

}  // namespace

void set_hip_error(const char *what, int code)
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = std::string(what) + ": " + hipGetErrorString((hipError_t)code);
}

const char *last_hip_error()
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_err.c_str();
}

int find_smems_geometry(const genie_index *ix, int32_t mode, int32_t max_len, int32_t *grid, int32_t *block,
                        int32_t *lds_bytes)
{
    Geometry g;
    int rc = plan_find_smems(ix, mode, max_len, true, 1ll << 40, &g);
    if (rc) return rc;
    if (grid) *grid = g.grid;
    if (block) *block = g.block;
    if (lds_bytes) *lds_bytes = g.lds;
    return GENIE_OK;
}

// Name of the match-statistics kernel the plan picks (as rocprofv3 prints it, without the argument list).
int search_kernel_name(const genie_index *ix, int32_t mode, int32_t max_len, char *buf, int32_t cap)
{
    Geometry g;
    int rc = plan_find_smems(ix, mode, max_len, true, 1ll << 40, &g);
    if (rc) return rc;
    char tmp[160];
    if (g.mt) snprintf(tmp, sizeof tmp, g.wide ? "match_table_long_kernel" : "match_table_kernel");
    else if (g.win) snprintf(tmp, sizeof tmp, "match_stats_sampled_long_kernel<%d, %s>", mode, ix->opt_lut_probe && mode == GENIE_MODE_LUT ? "true" : "false");
    else if (g.sampled) snprintf(tmp, sizeof tmp, "match_stats_sampled_kernel<%d, %s, %d>", mode, ix->opt_lut_probe && mode == GENIE_MODE_LUT ? "true" : "false", g.grp == 5 ? 5 : 0);
    else snprintf(tmp, sizeof tmp, "match_stats_kernel<%d, %d, %s, %s, %s>", mode, g.ns, g.wide ? "true" : "false", g.pair ? "true" : "false",
                  ix->opt_lut_probe && mode == GENIE_MODE_LUT ? "true" : "false");
    if (!buf || cap < (int)strlen(tmp) + 1) return GENIE_E_CAPACITY;
    memcpy(buf, tmp, strlen(tmp) + 1);
    return GENIE_OK;
}

int64_t find_smems_workspace_bytes(int64_t N, int32_t max_len) { return workspace_bytes_for(N, max_len); }

void find_smems_workspace_rows(int32_t max_len, int32_t out[4])
{
    Geometry g;
    shape_for(max_len, &g);
    out[0] = g.fwd_stride;
    out[1] = g.qp_recs;
    out[2] = g.hm_words;
    out[3] = g.kj_row * (g.wide ? 4 : 2);
}

static int launch_find_any(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens, int64_t N,
                           int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts, int32_t *d_slots,
                           int32_t cap, int32_t *d_status, void *d_ws, int64_t ws_bytes, const CsrOut &csr, void *stream)
{
    Geometry g;
    // with ragged lengths `fixed_len` carries the maximum length (host contract)
    int rc = plan_find_smems(ix, mode, fixed_len, d_lens == nullptr, N, &g);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
    case GENIE_MODE_BWA:
        return launch_find_mode<GENIE_MODE_BWA>(ix, g, d_reads, d_lens, N, stride, fixed_len, min_len, d_counts, d_slots,
                                                cap, d_status, d_ws, ws_bytes, csr, s);
    case GENIE_MODE_LUT:
        return launch_find_mode<GENIE_MODE_LUT>(ix, g, d_reads, d_lens, N, stride, fixed_len, min_len, d_counts, d_slots,
                                                cap, d_status, d_ws, ws_bytes, csr, s);
    case GENIE_MODE_RMI:
        return launch_find_mode<GENIE_MODE_RMI>(ix, g, d_reads, d_lens, N, stride, fixed_len, min_len, d_counts, d_slots,
                                                cap, d_status, d_ws, ws_bytes, csr, s);
    }
    return GENIE_E_INVALID;
}

int launch_find_smems(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens, int64_t N,
                      int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts, int32_t *d_slots,
                      int32_t cap, int32_t *d_status, void *d_ws, int64_t ws_bytes, void *stream)
{
    if (N == 0) return GENIE_OK;
    return launch_find_any(ix, mode, d_reads, d_lens, N, stride, fixed_len, min_len, d_counts, d_slots, cap, d_status, d_ws,
                           ws_bytes, CsrOut{}, stream);
}

int launch_find_smems_csr(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens, int64_t N,
                          int32_t stride, int32_t fixed_len, int32_t min_len, int64_t *d_offsets, int32_t *d_rows,
                          int64_t out_cap_rows, int32_t *d_status, void *d_ws, int64_t ws_bytes, void *stream)
{
    if (N == 0) {
        HIP_TRY(hipMemsetAsync(d_offsets, 0, 8, (hipStream_t)stream));
        return GENIE_OK;
    }
    CsrOut csr;
    csr.offsets = d_offsets;
    csr.rows = d_rows;
    csr.cap_rows = out_cap_rows;
    return launch_find_any(ix, mode, d_reads, d_lens, N, stride, fixed_len, min_len, nullptr, nullptr, 0, d_status, d_ws,
                           ws_bytes, csr, stream);
}

int launch_sa_interval(const genie_index *ix, const uint8_t *d_pats, const int32_t *d_lens, int64_t N, int32_t stride,
                       int32_t fixed_len, int32_t *d_out, void *stream)
{
    if (N == 0) return GENIE_OK;
    const DevIndex &d = ix->dev;
    const int dir_bytes = (d.dir_entries * 4 + 15) & ~15;
    const int Lmax = std::max(32, (fixed_len + 31) / 32 * 32);
    const int per_wave = scratch_bytes(Lmax);
    int waves = (160 * 1024 - dir_bytes) / per_wave;
    if (waves < 1) return GENIE_E_TOO_LONG;
    waves = waves >= 4 ? 4 : (waves >= 2 ? 2 : 1);
    const int lds = dir_bytes + waves * per_wave;
    const int cus = ix->num_cus > 0 ? ix->num_cus : 256;
    long long grid = std::min<long long>((N + waves - 1) / waves, (long long)cus * std::max(1, (160 * 1024) / lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sa_interval_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(sa_interval_kernel, dim3((unsigned)grid), dim3(waves * kWave), lds, (hipStream_t)stream, ix->dev,
                       d_pats, d_lens, (long long)N, stride, fixed_len, Lmax, reinterpret_cast<int2 *>(d_out));
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int launch_seed_lookup(const genie_index *ix, int32_t mode, const uint8_t *d_kmers, int64_t N, int32_t *d_out,
                       double *d_pred, void *stream)
{
    if (N == 0) return GENIE_OK;
    const unsigned grid = (unsigned)((N + 255) / 256);
    if (mode == GENIE_MODE_LUT)
        hipLaunchKernelGGL(seed_lookup_kernel<GENIE_MODE_LUT>, dim3(grid), dim3(256), 0, (hipStream_t)stream, ix->dev,
                           d_kmers, (long long)N, reinterpret_cast<int2 *>(d_out), d_pred);
    else if (mode == GENIE_MODE_RMI)
        hipLaunchKernelGGL(seed_lookup_kernel<GENIE_MODE_RMI>, dim3(grid), dim3(256), 0, (hipStream_t)stream, ix->dev,
                           d_kmers, (long long)N, reinterpret_cast<int2 *>(d_out), d_pred);
    else
        return GENIE_E_INVALID;
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int64_t compact_tmp_bytes(int64_t N)
{
    const int64_t nblocks = (N + kScanBlock - 1) / kScanBlock;
    return (nblocks + 2) * 8 + 16;
}

int launch_compact(const int32_t *d_counts, const int32_t *d_slots, int64_t N, int32_t cap, int64_t *d_offsets,
                   int32_t *d_out, int64_t out_cap_rows, void *d_tmp, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        HIP_TRY(hipMemsetAsync(d_offsets, 0, 8, s));
        return GENIE_OK;
    }
    const long long nblocks = (N + kScanBlock - 1) / kScanBlock;
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(d_tmp);
    int *overflow = reinterpret_cast<int *>(sums + nblocks + 1);
    HIP_TRY(hipMemsetAsync(overflow, 0, 4, s));
    hipLaunchKernelGGL(compact_block_sums, dim3((unsigned)nblocks), dim3(kScanBlock), 0, s, d_counts, (long long)N, cap, sums);
    hipLaunchKernelGGL(compact_scan_sums, dim3(1), dim3(kScanBlock), 0, s, sums, nblocks);
    hipLaunchKernelGGL(compact_scatter, dim3((unsigned)nblocks), dim3(kScanBlock), 0, s, d_counts,
                       reinterpret_cast<const int4 *>(d_slots), (long long)N, cap, sums,
                       reinterpret_cast<long long *>(d_offsets), reinterpret_cast<int4 *>(d_out), (long long)out_cap_rows,
                       overflow);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

// ------------------------------------------------------------------ position resolution (rows -> coordinates)
namespace {

__global__ void __launch_bounds__(256) locate_count_kernel(const int32_t *__restrict__ lohi, int stride, long long S,
                                                           int32_t *__restrict__ counts)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S) return;
    const int lo = lohi[g * stride], hi = lohi[g * stride + 1];
    counts[g] = (lo >= 0 && hi >= lo) ? hi - lo + 1 : 0;
}

// ExactMatch.get_positions (ExactMatch.py:195-199): the suffix-array entries of rows lo..hi, 1-based, in
// row order.  One lane per interval (almost all hold one or two rows); an interval of more than 32 rows
// is copied by the whole wave.
__global__ void __launch_bounds__(256) locate_scatter_kernel(DevIndex ix, const int32_t *__restrict__ lohi, int stride,
                                                             long long S, const long long *__restrict__ offsets,
                                                             int32_t *__restrict__ positions, long long cap)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int lo = 0, cnt = 0;
    long long off = 0;
    if (g < S) {
        lo = lohi[g * stride];
        const int hi = lohi[g * stride + 1];
        cnt = (lo >= 0 && hi >= lo) ? hi - lo + 1 : 0;
        off = offsets[g];
    }
    const bool big = cnt > 32;
    if (!big)
        for (int i = 0; i < cnt; i++)
            if (off + i < cap) positions[off + i] = ix.sa[lo + i].s + 1;
    unsigned long long todo = __ballot(big);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int blo = __shfl(lo, src), bcnt = __shfl(cnt, src);
        const long long boff = __shfl(off, src);
        for (int i = lane; i < bcnt; i += kWave)
            if (boff + i < cap) positions[boff + i] = ix.sa[blo + i].s + 1;
    }
}

}  // namespace

int64_t locate_tmp_bytes(int64_t S) { return ws_align(S * 4) + ws_align(compact_tmp_bytes(S)) + 256; }

int launch_locate(const genie_index *ix, const int32_t *d_lohi, int32_t stride, int64_t S, int64_t *d_offsets,
                  int32_t *d_positions, int64_t cap, void *d_tmp, int64_t tmp_bytes, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (S == 0) {
        HIP_TRY(hipMemsetAsync(d_offsets, 0, 8, s));
        return GENIE_OK;
    }
    if (!d_tmp || tmp_bytes < locate_tmp_bytes(S) || (reinterpret_cast<uintptr_t>(d_tmp) & 255) != 0) return GENIE_E_CAPACITY;
    int32_t *counts = reinterpret_cast<int32_t *>(d_tmp);
    void *scan_tmp = reinterpret_cast<uint8_t *>(d_tmp) + ws_align(S * 4);
    const unsigned grid = (unsigned)((S + 255) / 256);
    hipLaunchKernelGGL(locate_count_kernel, dim3(grid), dim3(256), 0, s, d_lohi, stride, (long long)S, counts);
    int rc = launch_compact(counts, nullptr, S, 0x7FFFFFFF, d_offsets, nullptr, 0, scan_tmp, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(locate_scatter_kernel, dim3(grid), dim3(256), 0, s, ix->dev, d_lohi, stride, (long long)S,
                       reinterpret_cast<const long long *>(d_offsets), d_positions, (long long)cap);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

}  // namespace genie

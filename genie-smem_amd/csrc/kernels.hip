// kernels.hip -- gfx950 (MI355X, wave64) kernels of the batched SMEM finder and their launch code.
//
// Device building blocks (this file): packed 32-base windows, the P-mer prefix directory with its
// exact tail corrections, suffix comparison on inline-key suffix-array records, interval search,
// K-mer hash table probe, RMI prediction + last-mile search.
// The read pipeline K_A (match statistics, match_table_kernel.inc) -> K_B (traversal, one lane per read)
// -> K_C (intervals, one lane per SMEM; short_read_kernel.inc); the single-step kernels
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
// Issue priority of the calling wave: level `i` mod 4 (s_setprio takes an immediate).  A CU serves its oldest waves first; waves
// of a persistent kernel that change their level as they go share the CU evenly instead (match_table_kernel.inc).
__device__ __forceinline__ void set_wave_priority(int i)
{
    switch (i & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// Ask for the 64-byte lines of [p, p + bytes) through the SCALAR cache (p, bytes wave-uniform; p inside a live allocation).
// A CU's L1 handles a bounded amount of miss time (it reports "pending" stalls two thirds of a lookup kernel's life) and a
// line that comes from HBM / the Infinity Cache is in flight for 1300-1800 cycles, seven to nine times as long as an L2 hit --
// the streamed lines of a persistent kernel (its next inputs) therefore cost it more than their count suggests.  Scalar
// loads reach the L2 by another path (the scalar data cache's own miss handling), so a wave that asks for its NEXT
// iteration's lines this way finds them in the L2 when its vector loads come ("streamed lines").  The loaded dwords are dropped; the wait is part of the block because the destination register must not be reused
// while a load is in flight (the wave waits here instead of at its first vector load).
__device__ __forceinline__ void scalar_touch_lines(const void *p, uint32_t bytes)
{
    uint32_t d, off;
    asm volatile("s_mov_b32 %[off], 0\n"
                 "1:\n\t"
                 "s_load_dword %[d], %[p], %[off]\n\t"
                 "s_add_u32 %[off], %[off], 64\n\t"
                 "s_cmp_lt_u32 %[off], %[n]\n\t"
                 "s_cbranch_scc1 1b\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [d] "=&s"(d), [off] "=&s"(off)
                 : [p] "s"(reinterpret_cast<uint64_t>(p) & ~3ull), [n] "s"(bytes)
                 : "scc", "memory");
}

// LDS traffic between lanes of ONE wave: the LDS executes a wave's instructions in order, so
// only the compiler has to be stopped from reordering.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
// K_C -- every 32-base window is ONE 16-byte ALIGNED load (one vector-memory instruction per window instead of
// two 8-byte loads; round 2 rebuilt the two-load form and found it correct on this hardware too).
struct QWords {
    const uint64_t *p;
    __device__ __forceinline__ uint64_t win(int pos) const { return qwin(p, pos); }
};
struct QRecs {
    const RefRec *p;
    __device__ __forceinline__ uint64_t win(int pos) const { return rwin(p, pos); }
};
// QPlain: plain 64-bit words in global memory (short reads since round 3: half the bytes of the records): a window is one
// 16-byte load at an 8-byte-aligned address.
struct QPlain {
    const uint64_t *p;
    __device__ __forceinline__ uint64_t win(int pos) const
    {
        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
        typedef u64x2 u64x2_a8 __attribute__((aligned(8)));
        const u64x2 v = *reinterpret_cast<const u64x2_a8 *>(p + (pos >> 5));
        return funnel(v.x, v.y, (pos & 31) * 2);
    }
};

// ------------------------------------------------------------------ prefix directory (LDS)
// dir[x] = number of SA rows whose suffix is smaller than the P-mer string x.  Rows whose
// suffix has fewer than P bases ("tails", P-1 of them plus the '$' row) need the two
// corrections below; padtail[l] is the A-padded code of the tail of length l.
//
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

// The match-statistics kernel's hand-off rows leave with the non-temporal hint: 6 % off that kernel at 100 kb, 1 % at 1 Mb.
// (The same hint on its input loads changed nothing; on the suffix-array rows of its deep path it cut the L2's misses by a
// fifth and cost 12-18 % in time: tools/experiments/README.md.)
typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
typedef nt_u4 nt_u4_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ void store_nt(void *dst, uint4 v)
{
    __builtin_nontemporal_store(nt_u4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u4 *>(dst));
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
    for (uint32_t tries = 0; tries < ix.lut_slots; tries++) {              // bounded: a corrupt table cannot spin a wave
        const int4 v = *reinterpret_cast<const int4 *>(ix.lut + p);
        if (v.y < 0) return false;
        if ((uint32_t)v.x == code) { lo = v.y; hi = v.z; return true; }
        p = p + 1 == ix.lut_slots ? 0 : p + 1;
    }
    return false;
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

typedef int mt_v4i __attribute__((ext_vector_type(4)));

// Compact match table: the keys of rows lb + 5 .. of an entry of 7 .. 13 suffixes, from its overflow block `blk` (one
// 16-byte load): the smallest key ^ xk16 seen, and a bit (row - lb) for each of the `nov` rows whose key agrees with xk16
// under `km16`.
__device__ __forceinline__ void mt_ov_scan(const __amdgpu_buffer_rsrc_t ov, uint32_t blk, int nov, uint32_t xk16, uint32_t km16,
                                           uint32_t &xmin, uint32_t &mask)
{
    const mt_v4i v = __builtin_amdgcn_raw_buffer_load_b128(ov, (int)(blk << 4), 0, 0);
    const uint32_t d[4] = {(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
#pragma unroll
    for (int i = 0; i < kM16OvKeys; i++) {
        const uint32_t x = ((d[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) ^ xk16;
        if (i < nov) {
            xmin = min(xmin, x);
            mask |= (x & km16) == 0 ? 1u << (kM16Keys - 1 + i) : 0u;
        }
    }
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

// single block: exclusive scan of the traversal kernels' block sums in place (+ grand total at [nblocks]); eight
// values per thread and pass, so that 10^5 .. 10^6 sums (10^7 .. 10^8 reads) stay a few tens of microseconds
__global__ void __launch_bounds__(kScanBlock) scan_block_sums_kernel(unsigned long long *__restrict__ sums, long long nblocks)
{
    constexpr int kPer = 8;
    __shared__ unsigned long long wave_total[kScanBlock / kWave];
    __shared__ unsigned long long carry;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long base = 0; base < nblocks; base += (long long)kScanBlock * kPer) {
        const long long i0 = base + (long long)threadIdx.x * kPer;
        unsigned long long v[kPer], mine = 0;
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            v[e] = i0 + e < nblocks ? sums[i0 + e] : 0ull;
            mine += v[e];
        }
        unsigned long long inc = mine;                               // inclusive scan of `mine` over the wave ...
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned int lo = (unsigned)__shfl_up((int)(unsigned)inc, off, kWave);
            const unsigned int hi = (unsigned)__shfl_up((int)(unsigned)(inc >> 32), off, kWave);
            if (lane >= off) inc += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == kWave - 1) wave_total[wave] = inc;
        __syncthreads();
        unsigned long long before = carry, all = 0;                  // ... and over the block
        for (int w = 0; w < kScanBlock / kWave; w++) {
            const unsigned long long t = wave_total[w];
            before += w < wave ? t : 0ull;
            all += t;
        }
        unsigned long long run = before + inc - mine;
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            if (i0 + e < nblocks) sums[i0 + e] = run;
            run += v[e];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nblocks] = carry;
}

__global__ void __launch_bounds__(kScanBlock) compact_scatter(const int32_t *__restrict__ counts,
                                                              const int4 *__restrict__ slots, long long N, int cap,
                                                              const unsigned long long *__restrict__ block_sums,
                                                              long long *__restrict__ offsets, int4 *__restrict__ out,
                                                              long long out_cap_rows)
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
            /* rows beyond the capacity are dropped: the caller compares offsets[N] with it */
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

inline long long table_bytes(const genie_index *ix)
{
    return (long long)((ix->dev.flags & kFlagCompactTable) ? sizeof(MatchRec16) : sizeof(MatchRec)) * ix->dev.mtab_entries;
}

struct Geometry {
    int grid, block, lds;
    int grp;         // reads per wave iteration (long reads: 1)
    int wps;         // waves per SIMD the match-statistics kernel is built for (match_table_kernel<WPS>)
    int wide;        // reads longer than 255 bases: uint16 fwd[], K_B reads it from global memory
    int max_len;
    int fwd_stride;  // bytes per fwd[] row in the workspace
    int fwd_lds;     // bytes per fwd[] row in K_B's LDS copy (narrow): an odd number of dwords
    int qp_recs;     // 16-byte pieces of the packed read per read in the workspace (long reads: overlapping records {w[i], w[i+1]};
                     // short reads: plain words, two per piece)
    int qp_stride;   // 16-byte units per read SLOT: short reads keep the head of the (count, pairs) row behind the packed read
    int kj_row;      // emitted-pair entries per read
};

inline void shape_for(int max_len, Geometry *g)
{
    g->max_len = max_len;
    g->wide = max_len > 255;
    // long reads: overlapping records {w[i], w[i+1]}, one per 32 bases; short reads: plain words in 16-byte pieces, enough of them
    // for a window that starts up to 8 bases behind the read (the zero padding K_A and K_C both see)
    g->qp_recs = g->wide ? (max_len + 31) / 32 : ((max_len + 8 + 31) / 32 + 1 + 1) / 2;
    // rows are multiples of 16 bytes (K_A copies them out 16 bytes per lane); K_B's LDS rows are an odd number of
    // dwords, so that lanes reading the same offset of their own rows hit distinct banks
    g->fwd_stride = g->wide ? ((max_len * 2) + 15) & ~15 : std::max(16, (max_len + 15) & ~15);
    int dw = std::max(1, (max_len + 3) / 4);
    if ((dw & 1) == 0) dw++;
    g->fwd_lds = dw * 4;
    g->kj_row = (std::max(max_len, 1) + 1 + 7) & ~7;      // entry 0 = the count
    g->qp_stride = g->wide ? g->qp_recs : g->qp_recs + kPairHeadWords * 8 / 16;      // 80 bytes at 150 bases: four slots = five lines
}

int plan_find_smems(const genie_index *ix, int mode, int max_len, long long N, Geometry *g)
{
    (void)mode;
    const int lds_cap = 160 * 1024;
    shape_for(max_len, g);
    const int cus = ix->num_cus > 0 ? ix->num_cus : 256;
    const int wpb = 8;                       // waves per block (block-wide in LDS: the group counter and the quad table)
    long long per_block;
    if (g->wide) {
        g->grp = 1;
        g->lds = wpb * mt_long_wave_bytes(max_len, g->qp_recs) + 16;                  // + the block's counter of reads handed out
        per_block = wpb;
    } else {
        // `grp` reads per wave iteration: about kMtTarget positions (one round-1 pass of 3 x 64 quads), and no
        // more reads than one pack pass holds
        const int ml = std::max(max_len, 1);
        const int grp = ix->opt_group_positions > 0 ? ix->opt_group_positions / ml : kMtTarget / ml;
        g->grp = std::max(1, std::min(std::min(kMtMaxG, kWave / mt_pack_dwords(ml) * 4), grp));
        g->lds = wpb * mt_wave_bytes(g->grp, ml, g->qp_recs, g->fwd_stride) + 16 + mt_quad_table_bytes(g->grp, ml);   // + the block's group counter and quad table
        per_block = (long long)wpb * g->grp;
    }
    if (g->lds > lds_cap) return GENIE_E_TOO_LONG;
    g->block = wpb * kWave;
    // waves per SIMD the kernel is built for: 4 (two blocks per CU) where the table exceeds an XCD's L2; else 6 for the compact
    // table (74 registers, nothing spilled, three blocks per CU: 595-607 us against 609-611 us per 10^6 reads with the
    // 64-register build and its 10 spilled registers -- the miss queue is full with three blocks' requests) and 8 for the
    // 32-byte one (on the 16 MB table of a 2.5 Mb reference the six-wave build was no faster than the four-wave one: 6.22 vs 6.10 ms)
    g->wps = !g->wide && table_bytes(ix) > kTableFitsL2 ? 4 : ((ix->dev.flags & kFlagCompactTable) ? 6 : 8);
    int bpc = std::min(lds_cap / g->lds, 4 * g->wps / wpb);                     // resident blocks per CU
    if (ix->opt_search_blocks_per_cu > 0) bpc = std::min(bpc, ix->opt_search_blocks_per_cu);
    if (bpc < 1) bpc = 1;
    long long gr = (long long)cus * bpc;
    const long long need = (N + per_block - 1) / per_block;
    if (gr > need) gr = need;
    if (gr < 1) gr = 1;
    g->grid = (int)gr;
    return GENIE_OK;
}

struct Workspace {
    uint8_t *fwd;        // N x fwd_stride bytes
    RefRec *qp;          // qp_recs records of 16 bytes per read
    int32_t *status;     // used when the caller passes no status array
    uint8_t *kj;         // N x kj_row emitted (start | end << shift) entries of 2 or 4 bytes
    int32_t *counts;     // used by the CSR entry point
    uint8_t *scan_tmp;   // the traversal blocks' sums and their scan (CSR entry point)
};

inline int64_t ws_align(int64_t x) { return (x + 255) & ~(int64_t)255; }

// block sums of the traversal kernels (CSR form): one 64-bit word per block; the smallest block holds 16 reads
inline int64_t block_sums_bytes(int64_t N) { return (N / 16 + 3) * 8; }

inline int64_t workspace_bytes_for(int64_t N, int max_len)
{
    Geometry g;
    shape_for(max_len, &g);
    return ws_align(N * (int64_t)g.fwd_stride) + ws_align(N * (int64_t)g.qp_stride * 16) + ws_align(N * 4) +
           ws_align(N * (int64_t)g.kj_row * (g.wide ? 4 : 2)) + ws_align(N * 4) + ws_align(block_sums_bytes(N)) + 256;
}

inline int carve_workspace(void *d_ws, int64_t ws_bytes, int64_t N, const Geometry &g, Workspace *ws)
{
    if (!d_ws || ws_bytes < workspace_bytes_for(N, g.max_len) || (reinterpret_cast<uintptr_t>(d_ws) & 255) != 0)
        return GENIE_E_CAPACITY;
    uint8_t *p = reinterpret_cast<uint8_t *>(d_ws);
    ws->fwd = p;
    p += ws_align(N * (int64_t)g.fwd_stride);
    ws->qp = reinterpret_cast<RefRec *>(p);
    p += ws_align(N * (int64_t)g.qp_stride * 16);
    ws->status = reinterpret_cast<int32_t *>(p);
    p += ws_align(N * 4);
    ws->kj = p;
    p += ws_align(N * (int64_t)g.kj_row * (g.wide ? 4 : 2));
    ws->counts = reinterpret_cast<int32_t *>(p);
    p += ws_align(N * 4);
    ws->scan_tmp = p;
    return GENIE_OK;
}

// long reads: lanes per read of the traversal kernel by batch size (measured, from-ref reads on the 100 kb reference: below)
constexpr long long kLongTwoLaneReads = 32768;

struct CsrOut {
    int64_t *offsets = nullptr;      // non-null => write CSR rows to `rows`, else slots
    int32_t *rows = nullptr;
    int64_t cap_rows = 0;
    // genie_find_smems_packed: 2-bit packed reads in; 8-byte rows, a count byte and a status byte per read out.
    // `offsets` then points at the totals (word 0 = rows, word 1 = escapes); `rows` at the 8-byte rows.
    bool packed = false;
    int row_bytes = 8;               // 8 or 6 (genie_find_smems_packed6)
    uint8_t *counts8 = nullptr, *status8 = nullptr;
    int64_t *escapes = nullptr;
    int64_t cap_escapes = 0;
};

template <int MODE, bool WIDE>
int launch_pipeline(const genie_index *ix, const Geometry &g, const uint8_t *d_reads, const int32_t *d_lens, int64_t N,
                    int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts, int32_t *d_slots, int32_t cap,
                    int32_t *d_status, const Workspace &ws, const CsrOut &csr, hipStream_t s)
{
    int32_t *st = d_status ? d_status : ws.status;
    int32_t *cnt = d_counts ? d_counts : ws.counts;
    const bool c16 = (ix->dev.flags & kFlagCompactTable) != 0;
    const long long mtab_bytes = table_bytes(ix);
    if (ix->ev_search_begin) HIP_TRY(hipEventRecord((hipEvent_t)ix->ev_search_begin, s));
    if (WIDE) {
        auto km = c16 ? match_table_long_kernel<true> : match_table_long_kernel<false>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(km), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds));
        hipLaunchKernelGGL(km, dim3(g.grid), dim3(g.block), g.lds, s, ix->dev, (int)MODE, d_reads, d_lens, (long long)N, stride,
                           fixed_len, reinterpret_cast<uint8_t *>(ws.fwd), g.fwd_stride, ws.qp, g.qp_recs, st,
                           std::max(g.max_len, 1), mtab_bytes, ix->opt_search_all | ix->opt_scheduling << 16, ix->num_cus > 0 ? ix->num_cus : 256);
    } else {
        auto km = csr.packed ? (c16 ? (g.wps == 4 ? match_table_kernel<4, true, true> : match_table_kernel<6, true, true>)
                                    : (g.wps == 4 ? match_table_kernel<4, false, true> : match_table_kernel<8, false, true>))
                             : (c16 ? (g.wps == 4 ? match_table_kernel<4, true, false> : match_table_kernel<6, true, false>)
                                    : (g.wps == 4 ? match_table_kernel<4, false, false> : match_table_kernel<8, false, false>));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(km), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds));
        hipLaunchKernelGGL(km, dim3(g.grid), dim3(g.block), g.lds, s, ix->dev, (int)MODE, d_reads, d_lens, (long long)N, stride,
                           fixed_len, reinterpret_cast<uint8_t *>(ws.fwd), g.fwd_stride, ws.qp, g.qp_recs, g.qp_stride, st,
                           g.grp, std::max(g.max_len, 1), mtab_bytes, ix->opt_search_all | ix->opt_debug << 8 | ix->opt_scheduling << 16, ix->num_cus > 0 ? ix->num_cus : 256);
    }
    HIP_TRY(hipGetLastError());
    if (ix->ev_search_end) HIP_TRY(hipEventRecord((hipEvent_t)ix->ev_search_end, s));
    if (ix->opt_search_only) return GENIE_W_SEARCH_ONLY;   // timing experiments: no counts / offsets / rows were written
    // the head of a read's (count, pairs) row: behind its packed-read records (short reads) or the kj row itself
    uint8_t *head = WIDE ? ws.kj : reinterpret_cast<uint8_t *>(ws.qp) + g.qp_recs * 16;
    const int head_stride = WIDE ? g.kj_row * 4 : g.qp_stride * 16;
    const int tb = 256;
    int reads_per_block = tb;                // of the traversal kernel
    unsigned long long *bsums = csr.offsets ? reinterpret_cast<unsigned long long *>(ws.scan_tmp) : nullptr;
    if (WIDE) {
        // lanes per read: 2, each taking eight positions per pass from LDS windows of the row, when the batch is large enough to
        // give every SIMD several such waves; else 16 lanes of one position each (few reads: the chip needs the lanes).
        // Measured (from-ref reads, 100 kb reference, traversal kernel alone): 300 000 x 500 bases 329 -> 256 us, 75 000 x 2000
        // bases 340 -> 270 us with two lanes; 18 750 x 8000 bases 410 us with 16 lanes, 690 us with four lanes of eight positions.
        const int lanes = N >= kLongTwoLaneReads ? 2 : 16;
        auto kl = lanes == 2 ? traverse_long_kernel<MODE, 2, 8> : traverse_long_kernel<MODE, 16, 1>;
        hipLaunchKernelGGL(kl, dim3((unsigned)((N + tb / lanes - 1) / (tb / lanes))), dim3(tb), lanes == 2 ? (tb / lanes) * kLongRowBytes : 0, s, d_lens,
                           (long long)N, fixed_len, min_len, ws.fwd, g.fwd_stride, cnt, reinterpret_cast<uint32_t *>(ws.kj),
                           g.kj_row, csr.offsets ? g.kj_row : cap, st, bsums);
        reads_per_block = tb / lanes;
    } else {                                 // one lane per read, rows staged in LDS
        auto kb = traverse_kernel<MODE>;
        const int lds_b = tb * g.fwd_lds;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
        hipLaunchKernelGGL(kb, dim3((unsigned)((N + tb - 1) / tb)), dim3(tb), lds_b, s, ix->dev, d_lens, (long long)N,
                           fixed_len, min_len, ws.fwd, g.fwd_stride, g.fwd_lds, cnt, ws.kj, g.kj_row, head, head_stride,
                           csr.offsets ? g.kj_row : cap, st, bsums, csr.counts8, csr.status8);
    }
    HIP_TRY(hipGetLastError());
    int block_shift = 0;
    while ((1 << block_shift) < reads_per_block) block_shift++;
    if (csr.offsets) {                       // scan of the traversal blocks' sums; K_C adds the in-block part
        const long long nb = (N + reads_per_block - 1) / reads_per_block;
        hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(kScanBlock), 0, s, bsums, nb);
        HIP_TRY(hipGetLastError());
    }
    // K_C: intervals + final rows, 16 lanes per read (4 reads per wave pass), persistent blocks
    const int cus = ix->num_cus > 0 ? ix->num_cus : 256;
    // (half the blocks help on the 1 Mb reference, 1.50 -> 1.40 ms per 4 x 10^6 reads, but cost at 300 kb, 0.215 -> 0.250 ms
    // per 10^6, with the same 8 MB table: not worth a rule)
    long long grid_c = (long long)cus * (32 / kIvWaves);
    const long long need_c = (N + kIvWaves * 4 * kIvUnroll - 1) / (kIvWaves * 4 * kIvUnroll);
    if (grid_c > need_c) grid_c = need_c;
    RowEscapes esc{nullptr, nullptr, 0};
    const int sched_c = ((ix->opt_scheduling >> 2) & 1) | cus << 8;  // bit 0: no priority rotation; bits 8..: CUs (blocks per round)
    if (csr.packed) {
        esc.count = reinterpret_cast<unsigned long long *>(csr.offsets) + 1;
        esc.list = reinterpret_cast<long long *>(csr.escapes);
        esc.cap = csr.cap_escapes;
        HIP_TRY(hipMemsetAsync(csr.offsets, 0, 16, s));
        auto kp = csr.row_bytes == 6 ? (c16 ? interval_kernel<true, false, true, 2> : interval_kernel<true, false, false, 2>)
                                     : (c16 ? interval_kernel<true, false, true, 1> : interval_kernel<true, false, false, 1>);
        hipLaunchKernelGGL(kp, dim3((unsigned)grid_c), dim3(kIvWaves * kWave), 0, s, ix->dev, (long long)N,
                           ws.kj, g.kj_row, head, head_stride, ws.qp, g.qp_stride, reinterpret_cast<void *>(csr.rows), 0,
                           reinterpret_cast<long long *>(csr.offsets), (long long)csr.cap_rows, bsums, cnt, block_shift, esc, sched_c);
        HIP_TRY(hipGetLastError());
        return GENIE_OK;
    }
    auto kc = csr.offsets ? (c16 ? interval_kernel<true, WIDE, true> : interval_kernel<true, WIDE, false>)
                          : (c16 ? interval_kernel<false, WIDE, true> : interval_kernel<false, WIDE, false>);
    if (csr.offsets)
        hipLaunchKernelGGL(kc, dim3((unsigned)grid_c), dim3(kIvWaves * kWave), 0, s, ix->dev, (long long)N,
                           ws.kj, g.kj_row, head, head_stride, ws.qp, g.qp_stride, reinterpret_cast<void *>(csr.rows), 0,
                           reinterpret_cast<long long *>(csr.offsets), (long long)csr.cap_rows, bsums, cnt, block_shift, esc, sched_c);
    else
        hipLaunchKernelGGL(kc, dim3((unsigned)grid_c), dim3(kIvWaves * kWave), 0, s, ix->dev, (long long)N,
                           ws.kj, g.kj_row, head, head_stride, ws.qp, g.qp_stride, reinterpret_cast<void *>(d_slots), cap, nullptr, 0ll, nullptr, nullptr, 0, esc, sched_c);
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
    if (g.wide)
        return launch_pipeline<MODE, true>(ix, g, d_reads, d_lens, N, stride, fixed_len, min_len, d_counts, d_slots, cap, d_status, ws, csr, s);
    return launch_pipeline<MODE, false>(ix, g, d_reads, d_lens, N, stride, fixed_len, min_len, d_counts, d_slots, cap, d_status, ws, csr, s);
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
    int rc = plan_find_smems(ix, mode, max_len, 1ll << 40, &g);
    if (rc) return rc;
    if (grid) *grid = g.grid;
    if (block) *block = g.block;
    if (lds_bytes) *lds_bytes = g.lds;
    return GENIE_OK;
}

// Name of the match-statistics kernel the plan picks (as rocprofv3 prints it, without the argument list).
int search_kernel_name(const genie_index *ix, int32_t mode, int32_t max_len, char *buf, int32_t cap)
{
    (void)mode;
    // match_table_kernel<waves per SIMD, compact table, packed reads> (the unpacked-reads instance: what bench.py times)
    const bool big = table_bytes(ix) > kTableFitsL2, c16 = (ix->dev.flags & kFlagCompactTable) != 0;
    const char *name = max_len > 255 ? (c16 ? "match_table_long_kernel<true>" : "match_table_long_kernel<false>")
                                     : (big ? (c16 ? "match_table_kernel<4, true, false>" : "match_table_kernel<4, false, false>")
                                            : (c16 ? "match_table_kernel<6, true, false>" : "match_table_kernel<8, false, false>"));
    if (!buf || cap < (int)strlen(name) + 1) return GENIE_E_CAPACITY;
    memcpy(buf, name, strlen(name) + 1);
    return GENIE_OK;
}

int64_t find_smems_workspace_bytes(int64_t N, int32_t max_len) { return workspace_bytes_for(N, max_len); }

void find_smems_workspace_rows(int32_t max_len, int32_t out[4])
{
    Geometry g;
    shape_for(max_len, &g);
    out[0] = g.fwd_stride;
    out[1] = g.qp_recs;
    out[2] = 0;                      // (a longest-match word, until K_B bounded its searches by the previous SMEM)
    out[3] = g.kj_row * (g.wide ? 4 : 2);
}

static int launch_find_any(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens, int64_t N,
                           int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts, int32_t *d_slots,
                           int32_t cap, int32_t *d_status, void *d_ws, int64_t ws_bytes, const CsrOut &csr, void *stream)
{
    Geometry g;
    // with ragged lengths `fixed_len` carries the maximum length (host contract)
    int rc = plan_find_smems(ix, mode, fixed_len, N, &g);
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

int launch_find_smems_packed(const genie_index *ix, int32_t mode, const uint8_t *d_reads2, const int32_t *d_lens, int64_t N,
                             int32_t stride_bytes, int32_t fixed_len, int32_t min_len, uint8_t *d_counts8, uint8_t *d_status8,
                             void *d_rows8, int64_t out_cap_rows, int64_t *d_totals, int64_t *d_escapes, int64_t cap_escapes,
                             void *d_ws, int64_t ws_bytes, void *stream, int row_bytes)
{
    if (N == 0) {
        HIP_TRY(hipMemsetAsync(d_totals, 0, 16, (hipStream_t)stream));
        return GENIE_OK;
    }
    CsrOut csr;
    csr.offsets = d_totals;
    csr.rows = reinterpret_cast<int32_t *>(d_rows8);
    csr.cap_rows = out_cap_rows;
    csr.packed = true;
    csr.row_bytes = row_bytes;
    csr.counts8 = d_counts8;
    csr.status8 = d_status8;
    csr.escapes = d_escapes;
    csr.cap_escapes = cap_escapes;
    return launch_find_any(ix, mode, d_reads2, d_lens, N, stride_bytes, fixed_len, min_len, nullptr, nullptr, 0, nullptr, d_ws,
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
    hipLaunchKernelGGL(compact_block_sums, dim3((unsigned)nblocks), dim3(kScanBlock), 0, s, d_counts, (long long)N, cap, sums);
    hipLaunchKernelGGL(compact_scan_sums, dim3(1), dim3(kScanBlock), 0, s, sums, nblocks);
    hipLaunchKernelGGL(compact_scatter, dim3((unsigned)nblocks), dim3(kScanBlock), 0, s, d_counts,
                       reinterpret_cast<const int4 *>(d_slots), (long long)N, cap, sums,
                       reinterpret_cast<long long *>(d_offsets), reinterpret_cast<int4 *>(d_out), (long long)out_cap_rows);
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

// ------------------------------------------------------------------ image contents (genie_index_validate)
namespace {

// Every value of the image that a kernel later uses as a row number, an entry index or a suffix start is checked
// against the section sizes, so that a corrupt or stale image cannot send a load outside the image.  (What the rows
// MEAN -- that the suffix array is sorted, that keys belong to their rows -- is not checked: a wrong table gives wrong
// answers, not wild accesses.)
__global__ void __launch_bounds__(256) validate_image_kernel(DevIndex ix, unsigned int *__restrict__ bad)
{
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long rows = (long long)ix.n + 1;
    unsigned int err = 0;
    for (long long r = t0; r < rows; r += stride) {
        const int s = ix.sa[r].s;
        err |= (s < 0 || s > ix.n) ? 1u : 0u;
    }
    for (long long x = t0; x < ix.dir_entries; x += stride) {
        const uint32_t v = ix.dir[x];
        err |= (v > (uint32_t)rows || (x + 1 < ix.dir_entries && v > ix.dir[x + 1])) ? 2u : 0u;
    }
    const long long nb2 = 1ll << (2 * ix.P2);
    for (long long c = t0; c < nb2; c += stride) {
        const HeadRec h = ix.dir2[c];
        err |= ((long long)h.lb + (h.meta & ~kHeadShort) > rows) ? 4u : 0u;
        if (ix.flags & kFlagCompactTable) {
            const MatchRec16 m = reinterpret_cast<const MatchRec16 *>(ix.mtab)[c];
            const uint32_t lb = m.w0 & 0xFFFFFFu, cnt4 = (m.w0 >> 24) & 15u, nib = m.w0 >> 28;
            if (cnt4 == 0) {
                const uint32_t last = (uint32_t)m.key[0] | (uint32_t)m.key[1] << 16;
                err |= (nib >= 1 && (lb > last || last > (uint32_t)ix.n)) ? 8u : 0u;
            } else {
                const uint32_t nrow = cnt4 == kM16More ? 7u + nib : cnt4;
                err |= (cnt4 > kM16More || (long long)lb + nrow > rows) ? 8u : 0u;
                err |= (cnt4 == kM16More && (nib > 6u || m.key[kM16Keys - 1] == 0 || (int)m.key[kM16Keys - 1] >= ix.ov_entries)) ? 16u : 0u;
            }
        } else {
            const MatchRec m = ix.mtab[c];
            const uint32_t nrow = m.meta >> 24;
            if (nrow == 0) {
                err |= ((m.meta & 0xFFu) >= 1 && (m.lb > m.key[0] || m.key[0] > (uint32_t)ix.n)) ? 8u : 0u;
            } else {
                err |= ((long long)m.lb + nrow > rows || (m.meta & 0xFFu) != (uint32_t)ix.P2) ? 8u : 0u;
                if (m.meta & kMatchMore) {
                    const long long idx = m.key[kMatchKeys - 1], extra = ((long long)nrow - (kMatchKeys - 1) + 7) >> 3;
                    err |= (idx < nb2 || idx + extra > ix.mtab_entries) ? 16u : 0u;
                }
            }
        }
    }
    for (long long i = t0; i < (long long)ix.lut_slots; i += stride) {
        const LutSlot sl = ix.lut[i];
        err |= (sl.lo >= 0 && (sl.lo > sl.hi || sl.hi > ix.n)) ? 32u : 0u;
    }
    if (ix.rmi_err && ix.nlev > 0)
        for (long long i = t0; i < ix.rmi_off[ix.nlev] - ix.rmi_off[ix.nlev - 1]; i += stride) err |= ix.rmi_err[i] < 0 ? 64u : 0u;
    if (err) atomicOr(bad, err);
}

}  // namespace

int validate_image(const genie_index *ix, unsigned int *what, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    unsigned int *d_bad = nullptr, h_bad = 0;
    HIP_TRY(hipMalloc(&d_bad, sizeof(unsigned int)));
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(unsigned int), s);
    if (e == hipSuccess) {
        const int cus = ix->num_cus > 0 ? ix->num_cus : 256;
        hipLaunchKernelGGL(validate_image_kernel, dim3((unsigned)cus * 8), dim3(256), 0, s, ix->dev, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, d_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_bad);
    if (e != hipSuccess) { set_hip_error("validate_image", (int)e); return GENIE_E_HIP; }
    if (what) *what = h_bad;
    return h_bad ? GENIE_E_BAD_BLOB : GENIE_OK;
}

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

// index_host.cpp -- host-side index construction and (de)serialization.
//
// Replaces the reference's O(n^2) index builders on this path:
//   ExactMatch.create_bwt_matrix / create_fm_index   (reference SMEM/ExactMatch.py:22-68)
//   LUT.generate_lut                                  (reference SMEM/LUT.py:15-35)
// with: a suffix array by prefix doubling (O(n log^2 n)), a 2-bit packed reference, a P-mer
// prefix directory, and the K-mer table read straight off the suffix-array order.  No BWT /
// Occ matrix is built -- interval search on the suffix array yields the same row intervals as
// FM backward search (pinned by tests against the reference's own outputs).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <new>

#include "genie_internal.h"

namespace genie {

namespace {

struct KeyIdx {
    uint64_t key;
    int32_t idx;
};

inline bool key_less(const KeyIdx &a, const KeyIdx &b) { return a.key < b.key; }

// Suffix array of codes+"$" (n+1 rows, '$' smallest) by prefix doubling.
// Round 0 ranks suffixes by their first 16 symbols (3 bits each: '$'/past-end = 0, base+1),
// every later round sorts by (rank[i], rank[i+h]) and doubles h until all ranks are distinct.
void build_suffix_array(const uint8_t *codes, int64_t n, std::vector<int32_t> &sa0)
{
    const int64_t rows = n + 1;
    std::vector<KeyIdx> v((size_t)rows);
    std::vector<int32_t> rank((size_t)rows), tmp((size_t)rows);
    for (int64_t i = 0; i < rows; i++) {
        uint64_t k = 0;
        for (int j = 0; j < 16; j++) {
            int64_t p = i + j;
            k = (k << 3) | (p < n ? (uint64_t)codes[p] + 1 : 0);
        }
        v[(size_t)i] = {k, (int32_t)i};
    }
    std::sort(v.begin(), v.end(), key_less);
    int64_t h = 16;
    for (;;) {
        int32_t r = 0;
        for (int64_t i = 0; i < rows; i++) {
            if (i > 0 && v[(size_t)i].key != v[(size_t)i - 1].key) r++;
            tmp[(size_t)i] = r;
        }
        for (int64_t i = 0; i < rows; i++) rank[(size_t)v[(size_t)i].idx] = tmp[(size_t)i];
        if (r == rows - 1) break;
        for (int64_t i = 0; i < rows; i++) {
            int32_t idx = v[(size_t)i].idx;
            int64_t j = (int64_t)idx + h;
            uint64_t second = j < rows ? (uint64_t)rank[(size_t)j] + 1 : 0;   // past the end sorts first
            v[(size_t)i].key = ((uint64_t)rank[(size_t)idx] << 32) | second;
        }
        std::sort(v.begin(), v.end(), key_less);
        h *= 2;
    }
    sa0.resize((size_t)rows);
    for (int64_t i = 0; i < rows; i++) sa0[(size_t)i] = v[(size_t)i].idx;
}

// 32 bases per word, base j of the word in bits [62-2j, 63-2j]: unsigned integer order of two
// windows == lexicographic order of the bases, clz(x^y)/2 == common prefix length.
void pack_reference(const uint8_t *codes, int64_t n, std::vector<RefRec> &ref)
{
    const int64_t words = (n + 31) / 32 + 3;
    std::vector<uint64_t> w((size_t)words + 1, 0);
    for (int64_t i = 0; i < n; i++) w[(size_t)(i >> 5)] |= (uint64_t)codes[i] << (62 - 2 * (i & 31));
    ref.resize((size_t)words);
    for (int64_t i = 0; i < words; i++) ref[(size_t)i] = {w[(size_t)i], w[(size_t)i + 1]};
}

inline uint32_t code_at(const uint8_t *codes, int64_t s, int len)
{
    uint32_t c = 0;
    for (int j = 0; j < len; j++) c = (c << 2) | codes[s + j];
    return c;
}

inline uint64_t code_at64(const uint8_t *codes, int64_t s, int len)
{
    uint64_t c = 0;
    for (int j = 0; j < len; j++) c = (c << 2) | codes[s + j];
    return c;
}

// Is suffix a (0-based start) lexicographically smaller than suffix b?  ('$' smallest)
bool suffix_less(const uint8_t *codes, int64_t n, int64_t a, int64_t b)
{
    int64_t la = n - a, lb = n - b, m = la < lb ? la : lb;
    int c = m ? memcmp(codes + a, codes + b, (size_t)m) : 0;
    if (c) return c < 0;
    return la < lb;
}

}  // namespace

int build_host_index(const uint8_t *codes, int64_t n, const int32_t *sa_one_based, int32_t K, int32_t P,
                     int32_t dir2_bits, int32_t table_format, HostIndex **out)
{
    if (!codes || !out || n < 1 || n > 0x7ffffff0ll || K < 0 || K > GENIE_MAX_K) return GENIE_E_INVALID;
    if (table_format < 0 || table_format > 2 || (table_format == 2 && n >= kM16MaxN)) return GENIE_E_INVALID;
    if (P <= 0 || P > GENIE_MAX_DIR_BITS) P = GENIE_MAX_DIR_BITS;
    for (int64_t i = 0; i < n; i++)
        if (codes[i] > 3) return GENIE_E_ALPHABET;
    HostIndex *h = new (std::nothrow) HostIndex();
    if (!h) return GENIE_E_NOMEM;
    try {
        h->n = n;
        h->K = K;
        h->P = P;
        h->codes.assign(codes, codes + n);
        const int64_t rows = n + 1;
        if (sa_one_based) {
            // adopt the caller's suffix array after checking it is a sorted permutation
            h->sa0.resize((size_t)rows);
            std::vector<uint8_t> seen((size_t)rows, 0);
            for (int64_t r = 0; r < rows; r++) {
                int64_t s = (int64_t)sa_one_based[r] - 1;
                if (s < 0 || s > n || seen[(size_t)s]) { delete h; return GENIE_E_INVALID; }
                seen[(size_t)s] = 1;
                h->sa0[(size_t)r] = (int32_t)s;
            }
            for (int64_t r = 1; r < rows; r++)
                if (!suffix_less(codes, n, h->sa0[(size_t)r - 1], h->sa0[(size_t)r])) { delete h; return GENIE_E_INVALID; }
        } else {
            build_suffix_array(codes, n, h->sa0);
        }
        h->sa1.resize((size_t)rows);
        for (int64_t r = 0; r < rows; r++) h->sa1[(size_t)r] = h->sa0[(size_t)r] + 1;
        pack_reference(codes, n, h->ref);
        // device suffix-array records: start + the 32 bases after the P-base prefix
        h->sarec.resize((size_t)rows);
        for (int64_t r = 0; r < rows; r++) {
            const int64_t s = h->sa0[(size_t)r], pos = s + P;
            uint64_t key = 0;
            if (pos < n) {
                const RefRec &rr = h->ref[(size_t)(pos >> 5)];
                const int sh = (int)(pos & 31) * 2;
                key = sh ? (rr.w0 << sh) | (rr.w1 >> (64 - sh)) : rr.w0;
            }
            h->sarec[(size_t)r] = SaRec{(int32_t)s, 0, key};
        }

        // Prefix directory: dir[x] = number of rows whose suffix is lexicographically smaller
        // than the P-mer string x (x = 4^P: all rows).  A row whose suffix holds >= P bases and
        // starts with P-mer c is smaller than every x > c; a tail row t+"$" (fewer than P bases)
        // is smaller than every x >= t padded with A's (t$ < tA...A).
        const int64_t nb = (int64_t)1 << (2 * P);
        std::vector<uint32_t> d((size_t)nb + 2, 0);
        for (int64_t r = 0; r < rows; r++) {
            int64_t s = h->sa0[(size_t)r], avail = n - s;
            if (avail >= P) d[(size_t)code_at(codes, s, P) + 1]++;
            else d[(size_t)((uint64_t)code_at(codes, s, (int)avail) << (2 * (P - avail)))]++;
        }
        h->dir.resize((size_t)nb + 1);
        uint32_t acc = 0;
        for (int64_t x = 0; x <= nb; x++) { acc += d[(size_t)x]; h->dir[(size_t)x] = acc; }
        // can the LDS copy be 16-bit deltas against every 16th entry? (the sampled search stages it so)
        {
            bool ok = true;
            for (int64_t x = 0; x <= nb && ok; x++) ok = h->dir[(size_t)x] - h->dir[(size_t)(x & ~(int64_t)15)] <= 65535u;
            h->flags = ok ? kFlagDir16 : 0;
        }
        // padded codes of the tail suffixes (row 0 = '$' has length 0 and pads to 0)
        for (int l = 0; l < 8; l++) {
            if (l < P && l <= n) h->padtail[l] = code_at(codes, n - l, l) << (2 * (P - l));
            else h->padtail[l] = kNoTail;
        }

        // Second-level range table: for every P2-mer the exact rows [lb, ub) whose suffix starts with it
        // (rows sharing a prefix are contiguous).  P2 = smallest with 4^P2 >= n/4 (and > P), i.e. <= 4 rows
        // per entry on average.  The match table has the same index: per P2-mer the 16-base continuations
        // of (up to kMatchKeys of) its suffixes, in ONE 32-byte entry -- what the match-statistics kernel
        // reads instead of searching rows.
        {
            int P2 = P + 1;
            while (P2 < 12 && ((int64_t)1 << (2 * P2)) < n / 4) P2++;
            if (dir2_bits > P && dir2_bits <= 12) P2 = dir2_bits;       // build-time tuning (genie_index_create_ex)
            h->P2 = P2;
            const int64_t nb2 = (int64_t)1 << (2 * P2);
            // table form: the compact entries (genie_internal.h) wherever a row number fits their 24 bits -- they halve the
            // L2 misses of a table that does not fit an XCD's L2, and measured 3.5 % faster on one that does (100 kb) --
            // else the 32-byte ones
            const bool compact = table_format == 2 || (table_format == 0 && n < kM16MaxN);
            const int KB = compact ? 8 : 16;                            // bases per key
            if (compact) h->flags |= kFlagCompactTable;
            h->dir2.assign((size_t)nb2, HeadRec{0, 0, 0});
            if (compact) {
                h->mtab16.assign((size_t)nb2, MatchRec16{0, {0, 0, 0, 0, 0, 0}});
                h->ov.assign(1, MatchOv16{{0, 0, 0, 0, 0, 0, 0, 0}});       // block 0 is never referenced
            } else {
                h->mtab.assign((size_t)nb2, MatchRec{0, 0, {0, 0, 0, 0, 0, 0}});
            }
            std::vector<uint32_t> cnt((size_t)nb2, 0), first((size_t)nb2, 0), ovf(compact ? (size_t)nb2 : 0, 0);
            std::vector<uint8_t> cut((size_t)nb2, 0);                   // a suffix of the entry has fewer than P2 + KB bases
            std::vector<uint8_t> cut16(compact ? (size_t)nb2 : 0, 0);   // ... fewer than P2 + 16 (compact form: no wide keys)
            for (int64_t r = 0; r < rows; r++) {
                const int64_t s = h->sa0[(size_t)r];
                if (n - s < P2) continue;
                const uint64_t c = code_at64(codes, s, P2);
                HeadRec &e = h->dir2[(size_t)c];
                if ((e.meta & ~kHeadShort) == 0) {
                    e.lb = (uint32_t)r;
                    e.key = h->sarec[(size_t)r].key;
                    e.meta = (n - s < P + 32) ? kHeadShort : 0;
                    first[(size_t)c] = (uint32_t)r;
                }
                e.meta++;
                cnt[(size_t)c]++;
                if (n - s < P2 + KB) cut[(size_t)c] = 1;
                if (compact && n - s < P2 + 16) cut16[(size_t)c] = 1;
            }
            // keys: up to kMatchKeys in the entry itself; kMatchKeys+1 .. kMatchChainRows rows: kMatchKeys-1 in the
            // entry, its last slot = index of the overflow entries (8 keys each) appended behind the table.
            // Compact form: up to kM16Keys in the entry; 7 .. kM16MaxRows rows: five in the entry + an overflow block of eight.
            for (int64_t r = 0; r < rows; r++) {
                const int64_t s = h->sa0[(size_t)r];
                if (n - s < P2) continue;
                const uint64_t c = code_at64(codes, s, P2);
                const uint32_t rows_c = cnt[(size_t)c], k = (uint32_t)r - first[(size_t)c];
                uint32_t key = 0, key32 = 0;
                for (int j = 0; j < 16; j++) {
                    const int64_t p = s + P2 + j;
                    key32 |= (p < n ? (uint32_t)codes[p] : 0u) << (30 - 2 * j);
                }
                key = compact ? key32 >> 16 : key32;                      // the first KB bases
                if (compact) {
                    MatchRec16 &m = h->mtab16[(size_t)c];
                    if (rows_c <= 3u && !cut16[(size_t)c]) {                 // wide keys: three 32-bit keys in dwords 1..3
                        uint32_t *k32 = reinterpret_cast<uint32_t *>(m.key);
                        if (k == 0) k32[0] = k32[1] = k32[2] = key32;
                        k32[k] = key32;
                        continue;
                    }
                    const bool block = !cut[(size_t)c] && rows_c > (uint32_t)kM16Keys && rows_c <= (uint32_t)kM16MaxRows;
                    if (k == 0) {
                        for (int j = 0; j < kM16Keys; j++) m.key[j] = (uint16_t)key;                 // unused slots repeat key[0]
                        ovf[(size_t)c] = 0;
                        if (block && (int64_t)h->ov.size() < kM16MaxOv) ovf[(size_t)c] = (uint32_t)h->ov.size();
                        if (ovf[(size_t)c]) h->ov.push_back(MatchOv16{{0, 0, 0, 0, 0, 0, 0, 0}});
                    }
                    if (ovf[(size_t)c]) {
                        MatchOv16 &o = h->ov[ovf[(size_t)c]];
                        if (k < (uint32_t)kM16Keys - 1) m.key[k] = (uint16_t)key;
                        else {
                            if (k == (uint32_t)kM16Keys - 1) for (int j = 0; j < kM16OvKeys; j++) o.key[j] = (uint16_t)key;
                            o.key[k - (kM16Keys - 1)] = (uint16_t)key;
                        }
                    } else if (k < (uint32_t)kM16Keys) {
                        m.key[k] = (uint16_t)key;
                    }
                    continue;
                }
                h->mtab[(size_t)c].lb = first[(size_t)c];
                const bool chain = !cut[(size_t)c] && rows_c > (uint32_t)kMatchKeys && rows_c <= (uint32_t)kMatchChainRows;
                if (!chain) {
                    if (k < (uint32_t)kMatchKeys) h->mtab[(size_t)c].key[k] = key;
                    continue;
                }
                if (k == 0) {                                             // reserve the overflow entries, filled with key 0
                    const uint32_t extra = (rows_c - (kMatchKeys - 1) + 7) / 8;
                    h->mtab[(size_t)c].key[kMatchKeys - 1] = (uint32_t)h->mtab.size();
                    MatchRec fill;
                    uint32_t *fw = reinterpret_cast<uint32_t *>(&fill);
                    for (int j = 0; j < 8; j++) fw[j] = key;
                    for (uint32_t x = 0; x < extra; x++) h->mtab.push_back(fill);
                }
                if (k < (uint32_t)kMatchKeys - 1) {
                    h->mtab[(size_t)c].key[k] = key;
                } else {
                    const uint32_t o = k - (kMatchKeys - 1);
                    reinterpret_cast<uint32_t *>(&h->mtab[(size_t)h->mtab[(size_t)c].key[kMatchKeys - 1] + o / 8])[o % 8] = key;
                }
            }
            // the rows of every t-mer (t < P2) that occurs anywhere in the reference, its last bases included:
            // rows sharing a prefix are contiguous, so first row + count
            std::vector<std::vector<uint32_t>> tfirst((size_t)P2), tcnt((size_t)P2);
            for (int t = 1; t < P2; t++) {
                tfirst[(size_t)t].assign((size_t)1 << (2 * t), 0);
                tcnt[(size_t)t].assign((size_t)1 << (2 * t), 0);
            }
            for (int64_t r = 0; r < rows; r++) {
                const int64_t s = h->sa0[(size_t)r];
                const int tmax = (int)std::min<int64_t>(P2 - 1, n - s);
                if (tmax < 1) continue;
                const uint64_t c = code_at64(codes, s, tmax);
                for (int t = 1; t <= tmax; t++) {
                    const size_t x = (size_t)(c >> (2 * (tmax - t)));
                    if (tcnt[(size_t)t][x]++ == 0) tfirst[(size_t)t][x] = (uint32_t)r;
                }
            }
            for (int64_t c = 0; c < nb2; c++) {
                const uint32_t k = cnt[(size_t)c];
                // absent: the longest prefix that does occur, and (for the interval kernel) that prefix's rows
                int t = 0;
                uint32_t plb = 0, plast = 0;
                if (k == 0) {
                    t = P2 - 1;
                    while (t >= 1 && !tcnt[(size_t)t][(size_t)((uint64_t)c >> (2 * (P2 - t)))]) t--;
                    if (t >= 1) {
                        const size_t x = (size_t)((uint64_t)c >> (2 * (P2 - t)));
                        plb = tfirst[(size_t)t][x];
                        plast = plb + tcnt[(size_t)t][x] - 1;               // last row (inclusive)
                    }
                }
                if (compact) {
                    MatchRec16 &m = h->mtab16[(size_t)c];
                    if (k == 0) {
                        m.w0 = plb | ((uint32_t)t << 28);
                        m.key[0] = (uint16_t)(plast & 0xFFFFu);
                        m.key[1] = (uint16_t)(plast >> 16);
                    } else if (k <= 3u && !cut16[(size_t)c]) {
                        m.w0 = first[(size_t)c] | (k << 24) | (kM16Wide << 28);
                    } else if (k <= (uint32_t)kM16Keys) {
                        m.w0 = first[(size_t)c] | (k << 24) | ((cut[(size_t)c] ? kM16General : 0u) << 28);
                    } else if (ovf[(size_t)c]) {                          // 7 .. 13 suffixes, none cut short: five keys + a block
                        m.key[kM16Keys - 1] = (uint16_t)ovf[(size_t)c];
                        m.w0 = first[(size_t)c] | (kM16More << 24) | ((k - 7u) << 28);
                    } else {                                              // the rows decide
                        m.w0 = first[(size_t)c] | ((uint32_t)kM16Keys << 24) | (kM16General << 28);
                    }
                    continue;
                }
                MatchRec &m = h->mtab[(size_t)c];
                if (k == 0) {
                    m.meta = (uint32_t)t;                                // lmask = 0, flags = 0, rows = 0
                    m.lb = plb;
                    m.key[0] = plast;
                } else {
                    const bool chain = !cut[(size_t)c] && k > (uint32_t)kMatchKeys && k <= (uint32_t)kMatchChainRows;
                    for (uint32_t i = k; i < (uint32_t)kMatchKeys; i++) m.key[i] = m.key[0];
                    // an entry that cannot decide alone proves P2 bases, no more: more rows than keys, or a cut-short suffix
                    const uint32_t slow = (k > (uint32_t)kMatchKeys || cut[(size_t)c]) ? kMatchSlow | (chain ? kMatchMore : 0u) : 0u;
                    m.meta = slow | (uint32_t)P2 | (slow ? 0u : 0x1Fu << 8) | ((k < 255 ? k : 255u) << 24);
                }
            }
        }

        // K-mer table (LUT.generate_lut): rows sharing a K-mer prefix are contiguous in the SA.
        if (K > 0 && n >= K) {
            bool have = false;
            uint32_t cur = 0;
            for (int64_t r = 0; r < rows; r++) {
                int64_t s = h->sa0[(size_t)r];
                if (n - s < K) continue;
                uint32_t c = code_at(codes, s, K);
                if (!have || c != cur) {
                    h->lut_code.push_back(c);
                    h->lut_lo.push_back((int32_t)r);
                    h->lut_hi.push_back((int32_t)r);
                    cur = c;
                    have = true;
                } else {
                    h->lut_hi.back() = (int32_t)r;
                }
            }
            const uint64_t m = h->lut_code.size();
            uint64_t slots = 2 * m + 8;                            // load factor <= 1/2
            h->lut_slots.assign((size_t)slots, LutSlot{0, -1, -1, 0});
            for (uint64_t i = 0; i < m; i++) {
                uint32_t p = lut_hash(h->lut_code[(size_t)i], (uint32_t)slots);
                while (h->lut_slots[p].lo >= 0) p = p + 1 == slots ? 0 : p + 1;
                h->lut_slots[p] = LutSlot{h->lut_code[(size_t)i], h->lut_lo[(size_t)i], h->lut_hi[(size_t)i], 0};
            }
        } else {
            h->lut_slots.assign(8, LutSlot{0, -1, -1, 0});
        }
    } catch (const std::bad_alloc &) {
        delete h;
        return GENIE_E_NOMEM;
    }
    *out = h;
    return GENIE_OK;
}

static int64_t align_up(int64_t x) { return (x + kSectionAlign - 1) / kSectionAlign * kSectionAlign; }

void fill_header(const HostIndex &h, BlobHeader *hdr, int32_t image_flags)
{
    const bool no_seed = (image_flags & GENIE_IMAGE_NO_SEED_TABLE) != 0;
    memset(hdr, 0, sizeof(*hdr));
    hdr->magic = kMagic;
    hdr->version = kBlobVersion;
    hdr->header_bytes = GENIE_HEADER_BYTES;
    hdr->n = h.n;
    hdr->K = h.K;
    hdr->P = h.P;
    hdr->ref_recs = (int64_t)h.ref.size();
    hdr->dir_entries = (int64_t)h.dir.size();
    hdr->lut_slots = no_seed ? 8 : (int64_t)h.lut_slots.size();
    hdr->lut_keys = (int64_t)h.lut_code.size();
    hdr->rmi_models = (int64_t)h.rmi.size();
    hdr->P2 = h.P2;
    hdr->flags = h.flags | (no_seed ? kFlagNoSeedTable : 0);
    hdr->dir2_entries = (int64_t)h.dir2.size();
    hdr->nlev = h.nlev;
    for (int l = 0; l < GENIE_MAX_RMI_LEVELS; l++) {
        hdr->rmi_size[l] = h.rmi_size[l];
        hdr->rmi_scale[l] = h.rmi_scale[l];
    }
    for (int l = 0; l <= GENIE_MAX_RMI_LEVELS; l++) hdr->rmi_off[l] = h.rmi_off[l];
    for (int l = 0; l < 8; l++) hdr->padtail[l] = h.padtail[l];
    int64_t off = GENIE_HEADER_BYTES;
    hdr->off_sa = off;
    off = align_up(off + (int64_t)h.sarec.size() * (int64_t)sizeof(SaRec));
    hdr->off_ref = off;
    off = align_up(off + (int64_t)h.ref.size() * (int64_t)sizeof(RefRec));
    hdr->off_dir = off;
    off = align_up(off + (int64_t)h.dir.size() * 4);
    hdr->off_lut = off;
    off = align_up(off + hdr->lut_slots * (int64_t)sizeof(LutSlot));
    hdr->off_rmi = off;
    off = align_up(off + (int64_t)std::max<size_t>(h.rmi.size(), 1) * (int64_t)sizeof(RmiModel));
    hdr->off_dir2 = off;
    off = align_up(off + (int64_t)std::max<size_t>(h.dir2.size(), 1) * (int64_t)sizeof(HeadRec));
    hdr->off_rmi_err = off;
    hdr->rmi_err_entries = (int64_t)h.rmi_err.size();
    off = align_up(off + (int64_t)std::max<size_t>(h.rmi_err.size(), 1) * 4);
    hdr->off_mtab = off;
    const bool compact = (h.flags & kFlagCompactTable) != 0;
    hdr->mtab_entries = compact ? (int64_t)h.mtab16.size() : (int64_t)h.mtab.size();
    off = align_up(off + (compact ? (int64_t)h.mtab16.size() * (int64_t)sizeof(MatchRec16) : (int64_t)h.mtab.size() * (int64_t)sizeof(MatchRec)));
    hdr->off_ov = off;
    hdr->ov_entries = (int64_t)h.ov.size();
    off = align_up(off + (int64_t)std::max<size_t>(h.ov.size(), 1) * (int64_t)sizeof(MatchOv16));
    hdr->total_bytes = off;
}

// ------------------------------------------------------------------ native RMI training
// RMI_LUT.train_RMI (RMI_LUT.py:36-50) + RMI.fit (RMI.py:10-50) without scikit-learn: the
// (K-mer code, SA row) pairs of every row whose suffix holds a full K-mer, in SA order; per level and
// per non-empty expert the closed-form least-squares line to the reference's bucket-budget target
// (RMI.py:29-39); points routed with the reference's clamp(int(p)) (RMI.py:42-46); empty experts alias
// the root model (RMI.py:24-26).  Also records, per leaf model, the largest |int(prediction) - row|
// over the training pairs: a window of that radius around a prediction holds every row of any K-mer
// that occurs in the reference, which bounds the last-mile search.
namespace {
struct Line { double coef, icpt; };

Line fit_line(const std::vector<double> &x, const std::vector<double> &y)
{
    const size_t m = x.size();
    long double sx = 0, sy = 0;
    for (size_t i = 0; i < m; i++) { sx += x[i]; sy += y[i]; }
    const double xm = (double)(sx / m), ym = (double)(sy / m);
    long double var = 0, cov = 0;
    for (size_t i = 0; i < m; i++) {
        const long double dx = (long double)x[i] - xm;
        var += dx * dx;
        cov += dx * ((long double)y[i] - ym);
    }
    const double slope = var > 0 ? (double)(cov / var) : 0.0;
    return Line{slope, ym - slope * xm};
}

inline int64_t route(double p, int scale)
{
    if (!(p > 0.0)) return 0;
    if (p >= (double)scale) return scale - 1;
    return (int64_t)p;
}
}  // namespace

int train_rmi(HostIndex &h, int n_experts, const int32_t *experts, double *mean_abs_err, int32_t *max_abs_err)
{
    const int nlev = n_experts + 1;
    if (n_experts < 0 || nlev > GENIE_MAX_RMI_LEVELS || h.K < 1 || h.n < h.K) return GENIE_E_INVALID;
    for (int l = 0; l < n_experts; l++)
        if (experts[l] < 1 || experts[l] > (1 << 24)) return GENIE_E_INVALID;
    const int64_t rows = h.n + 1;
    std::vector<double> x, y;
    x.reserve((size_t)rows);
    y.reserve((size_t)rows);
    for (int64_t r = 0; r < rows; r++) {
        const int64_t s = h.sa0[(size_t)r];
        if (h.n - s < h.K) continue;
        x.push_back((double)code_at64(h.codes.data(), s, h.K));
        y.push_back((double)r);
    }
    const size_t m = x.size();
    if (m == 0) return GENIE_E_INVALID;
    std::vector<int32_t> scales(nlev), sizes(nlev);
    for (int l = 0; l < nlev; l++) {
        scales[l] = l < n_experts ? experts[l] : 1;
        sizes[l] = l == 0 ? 1 : scales[l - 1];
    }
    std::vector<std::vector<Line>> models(nlev);
    std::vector<int32_t> assign(m, 0), nxt(m, 0);
    std::vector<double> cx, cy;
    for (int l = 0; l < nlev; l++) {
        const int nb = sizes[l], scale = scales[l];
        // stable counting sort of the points by expert
        std::vector<int64_t> bounds((size_t)nb + 1, 0);
        for (size_t i = 0; i < m; i++) bounds[(size_t)assign[i] + 1]++;
        for (int b = 0; b < nb; b++) bounds[(size_t)b + 1] += bounds[(size_t)b];
        std::vector<int64_t> order(m), fill(bounds.begin(), bounds.end() - 1);
        for (size_t i = 0; i < m; i++) order[(size_t)fill[(size_t)assign[i]]++] = (int64_t)i;
        models[l].resize((size_t)nb);
        double allocated = 0.0;
        for (int b = 0; b < nb; b++) {
            const int64_t b0 = bounds[(size_t)b], b1 = bounds[(size_t)b + 1];
            if (b0 == b1) { models[l][(size_t)b] = models[0][0]; continue; }        // RMI.py:24-26
            cx.clear();
            cy.clear();
            double ymin = 1e300, ymax = -1e300;
            for (int64_t t = b0; t < b1; t++) {
                const size_t i = (size_t)order[(size_t)t];
                cx.push_back(x[i]);
                cy.push_back(y[i]);
                ymin = std::min(ymin, y[i]);
                ymax = std::max(ymax, y[i]);
            }
            if (l < nlev - 1) {                                                     // RMI.py:29-39
                const double span = ymax - ymin;
                double budget = 1.0;
                if (span != 0.0) {
                    for (double &v : cy) v = (v - ymin) / span;
                    budget = (double)(b1 - b0) * (double)scale / (double)m;
                }
                for (double &v : cy) v = v * budget + allocated;
                allocated += budget;
            }
            const Line ln = fit_line(cx, cy);
            models[l][(size_t)b] = ln;
            for (int64_t t = b0; t < b1; t++) {
                const size_t i = (size_t)order[(size_t)t];
                nxt[i] = (int32_t)route(ln.coef * x[i] + ln.icpt, scale);            // one rounding each, no FMA
            }
        }
        // which leaf model serves a point = its expert at the last level
        if (l < nlev - 1) assign.swap(nxt);
    }
    // install (the layout genie_index_set_rmi produces) + per-leaf error bounds
    h.nlev = nlev;
    int64_t off = 0;
    for (int l = 0; l < GENIE_MAX_RMI_LEVELS; l++) {
        h.rmi_size[l] = l < nlev ? sizes[l] : 0;
        h.rmi_scale[l] = l < nlev ? scales[l] : 0;
        h.rmi_off[l] = (int32_t)off;
        if (l < nlev) off += sizes[l];
    }
    for (int l = nlev; l <= GENIE_MAX_RMI_LEVELS; l++) h.rmi_off[l] = (int32_t)off;
    h.rmi.clear();
    for (int l = 0; l < nlev; l++)
        for (const Line &ln : models[l]) h.rmi.push_back(RmiModel{ln.coef, ln.icpt});
    const int nleaf = sizes[nlev - 1];
    h.rmi_err.assign((size_t)nleaf, 0);
    long double tot = 0;
    int32_t worst = 0;
    for (size_t i = 0; i < m; i++) {
        const int leaf = assign[i];
        const Line &ln = models[nlev - 1][(size_t)leaf];
        const double p = ln.coef * x[i] + ln.icpt;
        const int64_t r0 = !(p > 0.0) ? 0 : (p >= (double)rows ? rows - 1 : (int64_t)p);    // int(start_sa), clamped
        const int64_t e = r0 > (int64_t)y[i] ? r0 - (int64_t)y[i] : (int64_t)y[i] - r0;
        if (e > h.rmi_err[(size_t)leaf]) h.rmi_err[(size_t)leaf] = (int32_t)e;
        if (e > worst) worst = (int32_t)e;
        tot += (long double)e;
    }
    if (mean_abs_err) *mean_abs_err = (double)(tot / m);
    if (max_abs_err) *max_abs_err = worst;
    return GENIE_OK;
}

int serialize(const HostIndex &h, void *dst, int64_t cap, int32_t image_flags)
{
    BlobHeader hdr;
    fill_header(h, &hdr, image_flags);
    if (!dst || cap < hdr.total_bytes) return GENIE_E_INVALID;
    uint8_t *p = (uint8_t *)dst;
    memset(p, 0, (size_t)hdr.total_bytes);
    memcpy(p, &hdr, sizeof(hdr));
    memcpy(p + hdr.off_sa, h.sarec.data(), h.sarec.size() * sizeof(SaRec));
    memcpy(p + hdr.off_ref, h.ref.data(), h.ref.size() * sizeof(RefRec));
    memcpy(p + hdr.off_dir, h.dir.data(), h.dir.size() * 4);
    if (hdr.flags & kFlagNoSeedTable) {
        const LutSlot empty{0, -1, -1, 0};
        for (int64_t i = 0; i < hdr.lut_slots; i++) memcpy(p + hdr.off_lut + i * (int64_t)sizeof(LutSlot), &empty, sizeof(empty));
    } else {
        memcpy(p + hdr.off_lut, h.lut_slots.data(), h.lut_slots.size() * sizeof(LutSlot));
    }
    if (!h.rmi.empty()) memcpy(p + hdr.off_rmi, h.rmi.data(), h.rmi.size() * sizeof(RmiModel));
    if (!h.dir2.empty()) memcpy(p + hdr.off_dir2, h.dir2.data(), h.dir2.size() * sizeof(HeadRec));
    if (!h.rmi_err.empty()) memcpy(p + hdr.off_rmi_err, h.rmi_err.data(), h.rmi_err.size() * 4);
    if (h.flags & kFlagCompactTable) {
        memcpy(p + hdr.off_mtab, h.mtab16.data(), h.mtab16.size() * sizeof(MatchRec16));
        memcpy(p + hdr.off_ov, h.ov.data(), h.ov.size() * sizeof(MatchOv16));
    } else {
        memcpy(p + hdr.off_mtab, h.mtab.data(), h.mtab.size() * sizeof(MatchRec));
    }
    return GENIE_OK;
}

// A section [off, off + count * elem) must lie inside the image after the header, 16-byte aligned.
static bool section_ok(const BlobHeader &hdr, int64_t bytes, int64_t off, int64_t count, int64_t elem)
{
    if (off < GENIE_HEADER_BYTES || (off & 15) != 0 || count < 0 || elem <= 0) return false;
    if (count > (INT64_MAX - off) / elem) return false;
    const int64_t end = off + count * elem;
    return end <= hdr.total_bytes && end <= bytes;
}

int dev_index_from_header(const BlobHeader &hdr, const void *d_blob, int64_t bytes, DevIndex *out)
{
    if (hdr.magic != kMagic || hdr.version != kBlobVersion || hdr.header_bytes != GENIE_HEADER_BYTES)
        return GENIE_E_BAD_BLOB;
    if (bytes < hdr.total_bytes || hdr.total_bytes < GENIE_HEADER_BYTES) return GENIE_E_BAD_BLOB;
    if (hdr.P < 1 || hdr.P > GENIE_MAX_DIR_BITS || hdr.n < 1 || hdr.n > 0x7ffffff0ll) return GENIE_E_BAD_BLOB;
    if (hdr.K < 0 || hdr.K > GENIE_MAX_K) return GENIE_E_BAD_BLOB;
    if (hdr.dir_entries != ((int64_t)1 << (2 * hdr.P)) + 1) return GENIE_E_BAD_BLOB;
    const bool compact = (hdr.flags & kFlagCompactTable) != 0;
    if (hdr.P2 <= hdr.P || hdr.P2 > 12 || hdr.dir2_entries != ((int64_t)1 << (2 * hdr.P2)) ||
        hdr.mtab_entries < hdr.dir2_entries || hdr.mtab_entries > hdr.dir2_entries + hdr.n || hdr.mtab_entries > (1 << 26))
        return GENIE_E_BAD_BLOB;
    if (compact ? (hdr.mtab_entries != hdr.dir2_entries || hdr.n >= kM16MaxN || hdr.ov_entries < 1 || hdr.ov_entries > kM16MaxOv)
                : hdr.ov_entries != 0)
        return GENIE_E_BAD_BLOB;
    if (hdr.ref_recs < (hdr.n + 31) / 32 + 3) return GENIE_E_BAD_BLOB;
    // every section inside the image (a truncated or corrupt image must not become a wild device pointer)
    if (!section_ok(hdr, bytes, hdr.off_sa, hdr.n + 1, sizeof(SaRec)) ||
        !section_ok(hdr, bytes, hdr.off_ref, hdr.ref_recs, sizeof(RefRec)) ||
        !section_ok(hdr, bytes, hdr.off_dir, hdr.dir_entries, 4) ||
        !section_ok(hdr, bytes, hdr.off_lut, hdr.lut_slots, sizeof(LutSlot)) ||
        !section_ok(hdr, bytes, hdr.off_rmi, hdr.rmi_models, sizeof(RmiModel)) ||
        !section_ok(hdr, bytes, hdr.off_dir2, hdr.dir2_entries, sizeof(HeadRec)) ||
        !section_ok(hdr, bytes, hdr.off_rmi_err, hdr.rmi_err_entries, 4) ||
        !section_ok(hdr, bytes, hdr.off_mtab, hdr.mtab_entries, compact ? sizeof(MatchRec16) : sizeof(MatchRec)) ||
        !section_ok(hdr, bytes, hdr.off_ov, hdr.ov_entries, sizeof(MatchOv16)))
        return GENIE_E_BAD_BLOB;
    // the hash table needs an empty slot for every probe sequence to end
    if (hdr.lut_keys < 0 || hdr.lut_slots > 0xFFFFFFFFll || hdr.lut_slots < 1 ||
        (!(hdr.flags & kFlagNoSeedTable) && hdr.lut_slots < hdr.lut_keys + 1))
        return GENIE_E_BAD_BLOB;
    // RMI level table: level l has rmi_size[l] models at rmi_off[l]; level l + 1 has rmi_scale[l] of them
    if (hdr.nlev < 0 || hdr.nlev > GENIE_MAX_RMI_LEVELS) return GENIE_E_BAD_BLOB;
    if (hdr.nlev > 0) {
        if (hdr.rmi_off[0] != 0 || hdr.rmi_size[0] != 1) return GENIE_E_BAD_BLOB;
        for (int l = 0; l < hdr.nlev; l++) {
            if (hdr.rmi_size[l] < 1 || hdr.rmi_scale[l] < 1) return GENIE_E_BAD_BLOB;
            if (hdr.rmi_off[l + 1] != hdr.rmi_off[l] + hdr.rmi_size[l]) return GENIE_E_BAD_BLOB;
            if (l + 1 < hdr.nlev && hdr.rmi_size[l + 1] != hdr.rmi_scale[l]) return GENIE_E_BAD_BLOB;
        }
        if ((int64_t)hdr.rmi_off[hdr.nlev] != hdr.rmi_models) return GENIE_E_BAD_BLOB;
        if (hdr.rmi_err_entries != 0 && hdr.rmi_err_entries != hdr.rmi_size[hdr.nlev - 1]) return GENIE_E_BAD_BLOB;
    } else if (hdr.rmi_models != 0 || hdr.rmi_err_entries != 0) {
        return GENIE_E_BAD_BLOB;
    }
    if ((reinterpret_cast<uintptr_t>(d_blob) & 15) != 0) return GENIE_E_INVALID;
    const uint8_t *p = (const uint8_t *)d_blob;
    memset(out, 0, sizeof(*out));
    out->sa = (const SaRec *)(p + hdr.off_sa);
    out->ref = (const RefRec *)(p + hdr.off_ref);
    out->dir = (const uint32_t *)(p + hdr.off_dir);
    out->lut = (const LutSlot *)(p + hdr.off_lut);
    out->rmi = (const RmiModel *)(p + hdr.off_rmi);
    out->dir2 = (const HeadRec *)(p + hdr.off_dir2);
    out->mtab = (const MatchRec *)(p + hdr.off_mtab);
    out->mtab_entries = (int32_t)hdr.mtab_entries;
    out->ov = (const MatchOv16 *)(p + hdr.off_ov);
    out->ov_entries = (int32_t)hdr.ov_entries;
    out->P2 = hdr.P2;
    out->flags = hdr.flags;
    out->rmi_err = hdr.rmi_err_entries > 0 ? (const int32_t *)(p + hdr.off_rmi_err) : nullptr;
    out->n = (int32_t)hdr.n;
    out->K = hdr.K;
    out->P = hdr.P;
    out->dir_entries = (int32_t)hdr.dir_entries;
    out->lut_slots = (uint32_t)hdr.lut_slots;
    out->nlev = hdr.nlev;
    for (int l = 0; l < GENIE_MAX_RMI_LEVELS; l++) out->rmi_scale[l] = hdr.rmi_scale[l];
    for (int l = 0; l <= GENIE_MAX_RMI_LEVELS; l++) out->rmi_off[l] = hdr.rmi_off[l];
    for (int l = 0; l < 8; l++) out->padtail[l] = hdr.padtail[l];
    return GENIE_OK;
}

}  // namespace genie

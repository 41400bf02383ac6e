// genie_internal.h -- shared between the host index builder, the C ABI and the HIP kernels.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "genie_smem.h"

#ifndef __HIPCC__
struct uint2 { unsigned int x, y; };
#endif

namespace genie {

constexpr uint64_t kMagic = 0x58444947454e4547ull;  // "GENEGIDX"
constexpr uint32_t kBlobVersion = 9;
constexpr uint32_t kNoTail = 0xFFFFFFFFu;
constexpr int kSectionAlign = 256;

// Second-level range table entry: the rows [lb, lb + cnt) whose suffix starts with this P2-mer
// (cnt == 0: absent) and a copy of the FIRST row's inline key, so that one 16-byte load both
// bounds the search and serves as its first probe.  kHeadShort marks a first row whose suffix has
// fewer than P + 32 bases (its key is zero padded): such an entry is used as a range only.
struct HeadRec {
    uint32_t lb;
    uint32_t meta;        // cnt | kHeadShort
    uint64_t key;
};
static_assert(sizeof(HeadRec) == 16, "HeadRec layout");
constexpr uint32_t kHeadShort = 0x80000000u;
constexpr int32_t kFlagDir16 = 1;

// Match-table entry (one per P2-mer, 32 bytes = ONE aligned fetch): everything the match-statistics
// kernel needs to know about the suffixes that start with this P2-mer.
//   meta byte 0  base   P2 when the P2-mer occurs, else the longest prefix of it (0 .. P2-1) that occurs
//                       anywhere in the reference (also inside its last P2-1 bases)
//   meta byte 1  lmask  0x1F when the P2-mer occurs and the entry is not slow, else 0:
//                       longest match >= base + (lcp & lmask)
//   meta byte 2  flags  kMatchSlow: more than kMatchKeys suffixes, or one of them has fewer than
//                       P2 + 16 bases (its key is zero padded) -- the entry alone only proves `base`;
//                       kMatchMore (with kMatchSlow): kMatchKeys + 1 .. kMatchChainRows suffixes, none cut
//                       short: key[kMatchKeys - 1] is the index (in 32-byte entries from the table start) of
//                       overflow entries holding keys kMatchKeys - 1, kMatchKeys, ... eight per entry
//   meta byte 3  rows   min(number of suffixes, 255)
//   lb                  suffix-array row of the first of them (rows of one P2-mer are contiguous); an ABSENT
//                       P2-mer (rows = 0, base >= 1): lb .. key[0] are the rows of its longest occurring prefix
//   key[i]              the 16 bases that FOLLOW the first P2 bases of suffix-array row lb + i (so the keys
//                       ascend), packed like the reference (base j in bits [30-2j, 31-2j]); unused slots
//                       repeat key[0].  meta, lb, key[0..1] are the first 16 bytes: most positions need no
//                       more (two or fewer suffixes, or a query key that does not exceed key[1]).
// The longest match of a query position is  min(base + max_i lcp(query key, key[i]), bases left)  unless
// the entry is slow or a key agrees in all 16 bases with more of the read left: then the suffix-array
// rows lb + i whose keys agree (or, for a slow entry, a search of all its rows) decide.  The rows that hold
// a pattern of P2 .. P2 + 16 bases are lb + i for the keys that agree with it that far (interval search); a
// LONGER pattern that is known to occur and whose first P2 + 16 bases single out one key is that one row.
constexpr int kMatchKeys = 6;
constexpr uint32_t kMatchSlow = 1u << 16;
constexpr uint32_t kMatchMore = 1u << 17;
constexpr int kMatchChainRows = kMatchKeys - 1 + 8 * 3;      // longest chain: three overflow entries
struct MatchRec {
    uint32_t meta;
    uint32_t lb;
    uint32_t key[kMatchKeys];
};
static_assert(sizeof(MatchRec) == 32, "MatchRec must be one 32-byte fetch");

// COMPACT match-table entry (16 bytes), the default form (references of fewer than 2^24 bases): on a table that does not
// fit an XCD's L2 the match-statistics kernel is bound by its L2 MISSES (each one a trip through the fabric), so the
// table is halved -- keys hold the 8 bases that follow the P2-mer instead of 16 -- and a lookup is ONE 16-byte load.
//   w0 bits  0..23  lb     suffix-array row of the first suffix that starts with the P2-mer (n < 2^24)
//      bits 24..27  cnt4   0: the P2-mer does not occur; 1..6: that many suffixes, their keys in key[0..cnt4) (unused
//                          slots repeat key[0]); 7 (kM16More): 7 .. 13 suffixes -- key[0..4] hold the first five,
//                          key[5] is the index of a 16-byte overflow block with the keys of rows lb + 5 .. (unused slots
//                          repeat its first key), and nib = the number of suffixes - 7
//      bits 28..31  nib    cnt4 in 1..6: 0, or kM16General = the rows decide (a suffix of the entry has fewer than
//                          P2 + 8 bases, so its key is zero padded; or more than 13 suffixes; or no overflow block
//                          index left): the entry only proves P2 bases;  cnt4 == 0: the longest prefix of the P2-mer
//                          (0 .. P2-1 bases) that occurs anywhere in the reference, and then lb .. (key[0] | key[1] << 16)
//                          are the suffix-array rows of that prefix
//   key[i]  the 8 bases that follow the first P2 bases of suffix-array row lb + i, packed like the reference (base j in
//           bits [14-2j, 15-2j]); ascending.
//   WIDE KEYS: an entry of one to three suffixes, none cut short within P2 + 16 bases, spends its twelve key bytes on
//           three 32-bit keys of 16 bases instead (nib = kM16Wide; dwords 1..3; unused slots repeat the first): most
//           lookups on a small reference land on such entries, and a key that is exhausted after 8 bases costs two more
//           requests (the rows) where one of 16 bases decides.
// A match that exhausts a key (P2 + 8 bases, more of the read left) is decided by the suffix-array rows whose keys
// agree (inline 32-base key, then the packed reference).
constexpr int kM16Keys = 6;
constexpr uint32_t kM16More = 7;
constexpr uint32_t kM16General = 8;
constexpr uint32_t kM16Wide = 4;
constexpr int kM16OvKeys = 8;                                  // keys per overflow block
constexpr int kM16MaxRows = kM16Keys - 1 + kM16OvKeys;         // 13: the most suffixes an entry + its block describe
constexpr int64_t kM16MaxOv = 65536;                           // overflow blocks a 16-bit index reaches
struct MatchRec16 {
    uint32_t w0;
    uint16_t key[kM16Keys];
};
static_assert(sizeof(MatchRec16) == 16, "MatchRec16 must be one 16-byte load");
struct MatchOv16 {
    uint16_t key[kM16OvKeys];
};
constexpr int64_t kM16MaxN = (int64_t)1 << 24;
constexpr int32_t kFlagNoSeedTable = 4;        // the image carries no K-mer hash table (find_smems-only image): genie_seed_lookup(LUT) refuses
constexpr int32_t kFlagCompactTable = 2;       // BlobHeader.flags / DevIndex.flags: mtab holds MatchRec16 entries
constexpr int64_t kTableFitsL2 = 4ll << 20;    // bytes of table an XCD's L2 keeps

// One slot of the device K-mer hash table (the GPU form of the reference's `lut` dict,
// SMEM/LUT.py:33-35): key -> inclusive SA interval.  Empty slot: lo < 0.
struct LutSlot {
    uint32_t key;
    int32_t lo;
    int32_t hi;
    int32_t pad;
};
static_assert(sizeof(LutSlot) == 16, "LutSlot must be one 16-byte load");

// One linear model of the RMI (sklearn LinearRegression with one feature, SMEM/RMI.py:23,40).
struct RmiModel {
    double coef;
    double icpt;
};

// One suffix-array row: the 0-based suffix start and, inline, the 32 bases that FOLLOW the first P
// bases of that suffix (packed like the reference, zero padded past the end).  A binary-search
// probe is ONE aligned 16-byte load: the P-mer directory has already fixed the first P bases, the
// inline key decides the next 32 -- the packed reference is only touched when more than P+32
// bases are equal (repeats).
struct SaRec {
    int32_t s;
    int32_t pad;
    uint64_t key;
};
static_assert(sizeof(SaRec) == 16, "SaRec must be one 16-byte load");

// Two consecutive 32-base words of the packed reference, so that any 32-base window is ONE
// aligned 16-byte load: rec[i] = { W[i], W[i+1] }.
struct RefRec {
    uint64_t w0;
    uint64_t w1;
};

// Header at the start of a serialized index image (padded to GENIE_HEADER_BYTES).
struct BlobHeader {
    uint64_t magic;
    uint32_t version;
    uint32_t header_bytes;
    int64_t total_bytes;
    int64_t n;
    int32_t K;
    int32_t P;
    int64_t off_sa;       // SaRec  [n+1]        suffix start + inline 32-base key; row 0 = the '$' suffix
    int64_t off_ref;      // RefRec [ref_recs]   big-endian-in-word 2-bit bases, zero padded
    int64_t off_dir;      // uint32 [4^P + 1]    prefix directory (rows < P-mer string)
    int64_t off_lut;      // LutSlot[lut_slots]
    int64_t off_rmi;      // RmiModel[rmi_models]
    int64_t ref_recs;
    int64_t dir_entries;
    int64_t lut_slots;
    int64_t lut_keys;
    int64_t rmi_models;
    int32_t nlev;
    int32_t rmi_size[GENIE_MAX_RMI_LEVELS];
    int32_t rmi_scale[GENIE_MAX_RMI_LEVELS];
    int32_t rmi_off[GENIE_MAX_RMI_LEVELS + 1];
    uint32_t padtail[8];  // padtail[l] = code(last l bases) << 2(P-l) for l < P, kNoTail if l > n
    int64_t off_dir2;     // HeadRec [4^P2]: rows of every P2-mer + the inline key of the first of them
    int64_t dir2_entries;
    int32_t P2;           // 0 = no second-level table
    int32_t flags;        // kFlagDir16: every directory entry is within 65535 rows of entry (x & ~15)
    int64_t off_rmi_err;  // int32 [rmi_err_entries]: per leaf model, max |int(prediction) - row| over the training keys
    int64_t rmi_err_entries;   // 0 = no error table (model installed from coefficients)
    int64_t off_mtab;     // MatchRec [4^P2 + overflow entries], or MatchRec16 [4^P2] (flags & kFlagCompactTable)
    int64_t mtab_entries;
    int64_t off_ov;       // MatchOv16 [ov_entries]: overflow blocks of the compact table
    int64_t ov_entries;   // 0 for the 32-byte form, else >= 1
};
// The serialized header occupies GENIE_HEADER_BYTES; the struct is copied into its front.
static_assert(sizeof(BlobHeader) <= GENIE_HEADER_BYTES, "header size");

// What the kernels receive by value (kernarg): raw device pointers + scalars.
struct DevIndex {
    const SaRec *sa;
    const RefRec *ref;
    const uint32_t *dir;
    const LutSlot *lut;
    const RmiModel *rmi;
    const HeadRec *dir2;   // second-level range table (global, L2-resident), or null
    const int32_t *rmi_err; // per-leaf error bounds of a natively trained RMI, or null
    const MatchRec *mtab;  // match table, 4^P2 entries + overflow entries (MatchRec16 entries when flags & kFlagCompactTable)
    const MatchOv16 *ov;   // compact table: overflow blocks
    int32_t ov_entries;
    int32_t mtab_entries;
    int32_t flags;
    int32_t n;
    int32_t K;
    int32_t P;
    int32_t dir_entries;
    uint32_t lut_slots;
    int32_t nlev;
    int32_t rmi_scale[GENIE_MAX_RMI_LEVELS];
    int32_t rmi_off[GENIE_MAX_RMI_LEVELS + 1];
    uint32_t padtail[8];
    int32_t P2;
};

inline uint32_t lut_hash(uint32_t code, uint32_t slots)
{
    return (uint32_t)(((uint64_t)(code * 0x9E3779B1u) * (uint64_t)slots) >> 32);
}

struct HostIndex {
    int64_t n = 0;
    int32_t K = 0, P = 0;
    std::vector<uint8_t> codes;
    std::vector<int32_t> sa0;            // 0-based starts, row 0 = n
    std::vector<SaRec> sarec;            // device layout of the suffix array
    std::vector<int32_t> sa1;            // reference convention (1-based), for the host API
    std::vector<RefRec> ref;
    std::vector<uint32_t> dir;
    int32_t P2 = 0;
    std::vector<HeadRec> dir2;           // 4^P2 entries
    std::vector<MatchRec> mtab;          // 4^P2 entries (+ overflow), 32-byte form
    std::vector<MatchRec16> mtab16;      // 4^P2 entries, compact form (then mtab is empty)
    std::vector<MatchOv16> ov;           // compact form: overflow blocks
    int32_t flags = 0;
    std::vector<uint32_t> lut_code;      // sorted distinct K-mers
    std::vector<int32_t> lut_lo, lut_hi;
    std::vector<LutSlot> lut_slots;
    uint32_t padtail[8];
    int32_t nlev = 0;
    int32_t rmi_size[GENIE_MAX_RMI_LEVELS] = {0, 0, 0, 0};
    int32_t rmi_scale[GENIE_MAX_RMI_LEVELS] = {0, 0, 0, 0};
    int32_t rmi_off[GENIE_MAX_RMI_LEVELS + 1] = {0, 0, 0, 0, 0};
    std::vector<RmiModel> rmi;
    std::vector<int32_t> rmi_err;        // per leaf model (native training only)
};

// table_format: 0 = automatic (compact when n < kM16MaxN), 1 = 32-byte entries,
// 2 = compact entries
int build_host_index(const uint8_t *codes, int64_t n, const int32_t *sa_one_based, int32_t K, int32_t P,
                     int32_t dir2_bits, int32_t table_format, HostIndex **out);
// image_flags: GENIE_IMAGE_* (genie_smem.h)
void fill_header(const HostIndex &h, BlobHeader *hdr, int32_t image_flags = 0);
int serialize(const HostIndex &h, void *dst, int64_t cap, int32_t image_flags = 0);
int dev_index_from_header(const BlobHeader &hdr, const void *d_blob, int64_t bytes, DevIndex *out);

}  // namespace genie

struct genie_index {
    genie::HostIndex *host = nullptr;
    genie::BlobHeader hdr;
    bool has_hdr = false;
    genie::DevIndex dev;
    bool has_dev = false;
    int32_t device = -1;
    void *owned_blob = nullptr;      // hipMalloc'ed by genie_index_to_device
    int64_t blob_bytes = 0;
    int32_t num_cus = 0;
    int32_t opt_search_all = 0;      // GENIE_OPT_SEARCH_ALL
    int32_t opt_group_positions = 0; // GENIE_OPT_GROUP_POSITIONS (0 = default)
    int32_t opt_search_only = 0;     // GENIE_OPT_SEARCH_ONLY
    int32_t opt_scheduling = 0;      // GENIE_OPT_SCHEDULING bits (A/B timing; results unchanged)
    int32_t opt_debug = 0;           // experiments: stages of the search kernel switched off (results invalid)
    int32_t opt_search_blocks_per_cu = 0;   // GENIE_OPT_SEARCH_BLOCKS_PER_CU (0 = as many as fit)
    void *ev_search_begin = nullptr; // optional hipEvent_t pair bracketing the search kernel
    void *ev_search_end = nullptr;
};

// kernels.hip
namespace genie {
int launch_sa_interval(const genie_index *ix, const uint8_t *d_pats, const int32_t *d_lens, int64_t N,
                       int32_t stride, int32_t fixed_len, int32_t *d_out, void *stream);
int launch_seed_lookup(const genie_index *ix, int32_t mode, const uint8_t *d_kmers, int64_t N, int32_t *d_out,
                       double *d_pred, void *stream);
int launch_find_smems(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens,
                      int64_t N, int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts,
                      int32_t *d_slots, int32_t cap, int32_t *d_status, void *d_ws, int64_t ws_bytes, void *stream);
int64_t find_smems_workspace_bytes(int64_t N, int32_t max_len);
int launch_find_smems_csr(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens,
                          int64_t N, int32_t stride, int32_t fixed_len, int32_t min_len, int64_t *d_offsets,
                          int32_t *d_rows, int64_t out_cap_rows, int32_t *d_status, void *d_ws, int64_t ws_bytes,
                          void *stream);
int launch_find_smems_packed(const genie_index *ix, int32_t mode, const uint8_t *d_reads2, const int32_t *d_lens, int64_t N,
                             int32_t stride_bytes, int32_t fixed_len, int32_t min_len, uint8_t *d_counts8, uint8_t *d_status8,
                             void *d_rows8, int64_t out_cap_rows, int64_t *d_totals, int64_t *d_escapes, int64_t cap_escapes,
                             void *d_ws, int64_t ws_bytes, void *stream, int row_bytes = 8);
int launch_compact(const int32_t *d_counts, const int32_t *d_slots, int64_t N, int32_t cap, int64_t *d_offsets,
                   int32_t *d_out, int64_t out_cap_rows, void *d_tmp, void *stream);
int64_t compact_tmp_bytes(int64_t N);
int64_t locate_tmp_bytes(int64_t S);
int train_rmi(HostIndex &h, int n_experts, const int32_t *experts, double *mean_abs_err, int32_t *max_abs_err);
int launch_locate(const genie_index *ix, const int32_t *d_lohi, int32_t stride, int64_t S, int64_t *d_offsets,
                  int32_t *d_positions, int64_t cap, void *d_tmp, int64_t tmp_bytes, void *stream);
int find_smems_geometry(const genie_index *ix, int32_t mode, int32_t max_len, int32_t *grid, int32_t *block,
                        int32_t *lds_bytes);
void find_smems_workspace_rows(int32_t max_len, int32_t out[4]);
int search_kernel_name(const genie_index *ix, int32_t mode, int32_t max_len, char *buf, int32_t cap);
int validate_image(const genie_index *ix, unsigned int *what, void *stream);
void set_hip_error(const char *what, int code);
const char *last_hip_error();
}  // namespace genie

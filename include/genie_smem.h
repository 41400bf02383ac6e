/*
 * genie_smem.h -- C ABI of the MI355X-native batched SMEM finder (libgenie_smem.so).
 *
 * This is the drop-in boundary for the one hot path of jgkellymit/GENIE-SMEM: seed lookup,
 * suffix-array interval search and the SMEM extension loop.  The reference has no FFI (it is
 * in-process Python), so each entry point names the reference method it replaces; the Python
 * classes in genie-smem_amd/ bind these with ctypes and re-expose the reference's own method
 * names (see INTEGRATION.md for the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in any signature (`stream` is a
 *     hipStream_t passed as void*, 0 = the null stream);
 *   - pointers named d_* are DEVICE pointers owned by the caller (e.g. torch tensors);
 *     everything else is host memory;
 *   - bases are codes 0..3 in the sorted order of the reference alphabet (ACGT -> 0..3, the
 *     map of LUT.convert_seq_to_num, reference SMEM/LUT.py:37-48);
 *   - suffix-array rows: n+1 rows, row 0 is the '$' suffix; intervals are 0-based inclusive
 *     [lo, hi] exactly as ExactMatch.exact_match_back_prop returns them (SMEM/ExactMatch.py:151);
 *     an absent pattern is (-1, -1) where the reference returns the int -1;
 *   - every function returns 0 (GENIE_OK) or a negative genie_status; none throws.  (One positive code exists,
 *     GENIE_W_SEARCH_ONLY, returned only while the timing knob GENIE_OPT_SEARCH_ONLY is set.)
 *   - an index handle is immutable once opened on a device: any number of concurrent calls on
 *     distinct streams may share it.
 */
#ifndef GENIE_SMEM_H
#define GENIE_SMEM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GENIE_ABI_VERSION 2      /* 2: GENIE_W_SEARCH_ONLY, option codes renumbered since 1, workspace layout, image version */
#define GENIE_HEADER_BYTES 512      /* fixed-size header at the start of a serialized index */
#define GENIE_MAX_K 16              /* K-mer codes are 32-bit (2 bits per base) */
#define GENIE_MAX_DIR_BITS 7        /* P: prefix directory has 4^P + 1 entries (64 KB: genie_sa_interval stages it in LDS) */
#define GENIE_MAX_READ_LEN 8192     /* per-read scratch lives in LDS */
#define GENIE_MAX_RMI_LEVELS 4

typedef enum genie_status {
    GENIE_OK = 0,
    GENIE_E_INVALID = -1,      /* bad argument (null pointer, negative size, K out of range ...) */
    GENIE_E_ALPHABET = -2,     /* a base code > 3 in the reference */
    GENIE_E_NOMEM = -3,
    GENIE_E_NO_DEVICE = -4,    /* index has no device image (call genie_index_open / _to_device) */
    GENIE_E_HIP = -5,          /* HIP runtime error; see genie_last_hip_error() */
    GENIE_E_TOO_LONG = -6,     /* read/pattern longer than GENIE_MAX_READ_LEN */
    GENIE_E_NO_MODEL = -7,     /* RMI mode requested but no model was set */
    GENIE_E_BAD_BLOB = -8,     /* serialized index: wrong magic / version / size */
    GENIE_E_NO_LUT = -9,       /* LUT mode requested but the index was built with K = 0, or the image has no seed table */
    GENIE_E_CAPACITY = -10,    /* output buffer too small (compaction) */
    GENIE_W_SEARCH_ONLY = 1    /* GENIE_OPT_SEARCH_ONLY is set: only the match-statistics kernel was launched;
                                  counts / offsets / rows were NOT written */
} genie_status;

/* Per-read status written by genie_find_smems into d_status (0 = ok).  They mirror how the
 * reference fails on the same input (SURVEY.md section 8a "quirks"). */
enum {
    GENIE_READ_OK = 0,
    GENIE_READ_BAD_BASE = 1,   /* code > 3: reference raises KeyError (ExactMatch.py:139, LUT.py:47) */
    GENIE_READ_TOO_SHORT = 2,  /* len < K in LUT/RMI mode: reference mis-encodes (SMEM.py:26-28) */
    GENIE_READ_ABSENT_BASE = 3,/* a base that never occurs in the reference: reference raises
                                  KeyError('') (SMEM.py:39) or never terminates (SMEM.py:465,484) */
    GENIE_READ_OVERFLOW = 4    /* more SMEMs than `cap` slots; d_counts holds the true count */
};

/* SMEM traversal selector == which reference method is replaced. */
enum {
    GENIE_MODE_BWA = 0,        /* SMEM.get_SMEMS        (SMEM/SMEM.py:456-467) */
    GENIE_MODE_LUT = 1,        /* SMEM.get_smems_lut    (SMEM/SMEM.py:20-192)  */
    GENIE_MODE_RMI = 2         /* SMEM.get_smems_rmi    (SMEM/SMEM.py:206-384) */
};

typedef struct genie_index genie_index;

typedef struct genie_info {
    int64_t n;                 /* reference length in bases (without '$') */
    int32_t K;                 /* LUT / RMI key size (lut_size, prediction_size); 0 = none */
    int32_t dir_bits;          /* P */
    int64_t lut_keys;          /* distinct K-mers (len(lut.lut) in the reference) */
    int64_t lut_slots;         /* device hash-table slots */
    int32_t rmi_levels;        /* 0 = no model */
    int32_t has_host;          /* host arrays present (built here, not attached from a blob) */
    int32_t has_device;        /* device image bound */
    int32_t device;            /* HIP device ordinal of the image, -1 if none */
    int64_t blob_bytes;        /* size of the serialized / device image */
} genie_info;

/* ---------------------------------------------------------------------------------------
 * Index construction (host).  Replaces ExactMatch.create_fm_index + LUT.generate_lut
 * (SMEM/ExactMatch.py:22-33, SMEM/LUT.py:15-35): builds the suffix array of ref+"$", the
 * 2-bit packed reference, the P-mer prefix directory and (K > 0) the K-mer table.
 * No FM Occ/BWT is built: suffix-array bound search gives the same intervals (tests pin this).
 * ------------------------------------------------------------------------------------- */
int genie_index_create(const uint8_t *codes, int64_t n, int32_t K, int32_t dir_bits, genie_index **out);

/* Same, but adopt a suffix array supplied by the caller in the reference's JSON convention
 * (fm_index["suffix_array"]: n+1 entries, 1-based starts, row 0 = n+1; SMEM/ExactMatch.py:66).
 * Used by ExactMatch.load_fm_index on files the reference wrote. */
int genie_index_create_from_sa(const uint8_t *codes, int64_t n, const int32_t *sa_one_based, int32_t K,
                               int32_t dir_bits, genie_index **out);

/* genie_index_create / _from_sa (sa_one_based may be NULL) with the size of the per-P2-mer tables chosen by
 * the caller: table_bits = P2 in (dir_bits, 12] (anything else but 0 is GENIE_E_INVALID), 0 = automatic (smallest P2 with 4^P2 >= n/4: measured best on MI355X at n = 100 kb and 1 Mb).  A tuning
 * knob of the index image only: results do not depend on it.  The form of the per-P2-mer match table is chosen the same
 * way: GENIE_TABLE_WIDE / GENIE_TABLE_COMPACT ORed into table_bits, neither = automatic (compact: 16-byte entries, for
 * every reference of fewer than 2^24 bases). */
#define GENIE_TABLE_WIDE (1 << 8)    /* OR into table_bits: 32-byte entries with 16-base keys */
#define GENIE_TABLE_COMPACT (2 << 8) /* OR into table_bits: 16-byte entries with 8-base keys (n < 2^24) */
int genie_index_create_ex(const uint8_t *codes, int64_t n, const int32_t *sa_one_based, int32_t K, int32_t dir_bits,
                          int32_t table_bits, genie_index **out);

/* Install an RMI model (RMI.models after RMI.fit, SMEM/RMI.py:10-50): `nlev` levels, level l has
 * sizes[l] linear models (sizes[0] == 1) and the clamp scale scales[l] (= experts + [1], RMI.py:54);
 * coef / icpt are the per-level arrays concatenated.  Must precede serialize / to_device. */
int genie_index_set_rmi(genie_index *ix, int32_t nlev, const int32_t *sizes, const int32_t *scales,
                        const double *coef, const double *icpt);

/* Train the RMI natively (no scikit-learn): RMI_LUT.train_RMI + RMI.fit (SMEM/RMI_LUT.py:36-50,
 * SMEM/RMI.py:10-50) on the handle's own (K-mer code, SA row) pairs, `experts` as in
 * RMI_LUT(structure=experts, ...), e.g. {1000} or {10, 100}; closed-form least squares per expert with
 * the reference's bucket-budget targets.  Installs the model like genie_index_set_rmi and also records a
 * per-leaf error bound (max |int(prediction) - row| over the training pairs) that the device uses to
 * fence the last-mile search of genie_seed_lookup.  Outputs (optional): mean and max of that error.
 * Coefficients differ from scikit-learn's in the last bits, which only moves where a search starts. */
int genie_index_train_rmi(genie_index *ix, int32_t n_experts, const int32_t *experts, double *mean_abs_err,
                          int32_t *max_abs_err);
/* Export the installed model: coef / icpt hold all levels concatenated (1 + experts[0] + ... entries);
 * leaf_err (optional, natively trained models only) one bound per model of the last level. */
int genie_index_rmi_models(const genie_index *ix, double *coef, double *icpt, int32_t *leaf_err);

int genie_index_info(const genie_index *ix, genie_info *out);

/* Host views for the position-resolution helpers of the drop-in API
 * (ExactMatch.get_position(s) / exact_match, SMEM/ExactMatch.py:174-199).
 * 1-based values as the reference stores them; NULL when the handle has no host arrays. */
const int32_t *genie_index_suffix_array(const genie_index *ix);
/* Sorted distinct K-mer table (the reference's lut dict in key order): codes/lo/hi[lut_keys]. */
int genie_index_lut_arrays(const genie_index *ix, const uint32_t **codes, const int32_t **lo, const int32_t **hi);

/* Serialize to one flat, position-independent image (header + sections).  The caller uploads
 * it (e.g. torch.from_numpy(buf).cuda()) and -- multi-GPU -- broadcasts that ONE tensor over
 * RCCL; every rank then calls genie_index_open on its copy. */
int64_t genie_index_blob_bytes(const genie_index *ix);
int genie_index_serialize(const genie_index *ix, void *host_dst, int64_t cap);

/* The same with options.  GENIE_IMAGE_NO_SEED_TABLE leaves out the K-mer hash table (half of the image at 1 Mb): for ranks
 * that only run genie_find_smems* / genie_sa_interval / genie_locate -- genie_seed_lookup(LUT) on such an image returns
 * GENIE_E_NO_LUT.  (What a multi-GPU driver broadcasts.) */
#define GENIE_IMAGE_NO_SEED_TABLE 1
int64_t genie_index_image_bytes(const genie_index *ix, int32_t image_flags);
int genie_index_serialize_image(const genie_index *ix, int32_t image_flags, void *host_dst, int64_t cap);

/* Open a device-resident image.  `host_header` = the first GENIE_HEADER_BYTES of the same
 * image in host memory (ranks that received it by broadcast copy those bytes back).  The
 * image is NOT copied and must outlive the handle.  If `ix_inout` points at an existing handle
 * the image is bound to it (keeps the host arrays); otherwise a device-only handle is made. */
int genie_index_open(const void *host_header, const void *d_blob, int64_t blob_bytes, int32_t device,
                     genie_index **ix_inout);

/* Check the CONTENTS of an opened image on the device (genie_index_open checks the header only): every row number,
 * entry index and suffix start that a kernel will later follow must lie inside its section.  One pass over the image on
 * `stream`, then a synchronisation.  GENIE_E_BAD_BLOB when something is out of range (*what, optional, gets a bit per
 * section: 1 suffix array, 2 prefix directory, 4 range table, 8 / 16 match table / its overflow links, 32 K-mer table,
 * 64 RMI error bounds).  The drop-in calls it on every image it opens, also on one received by broadcast. */
int genie_index_validate(const genie_index *ix, uint32_t *what, void *stream);

/* Convenience for non-torch callers: hipMalloc + upload an image owned by the handle. */
int genie_index_to_device(genie_index *ix, int32_t device);

void genie_index_destroy(genie_index *ix);

/* ---------------------------------------------------------------------------------------
 * Hot path (device).  All launches are asynchronous on `stream`.
 * ------------------------------------------------------------------------------------- */

/* Batched ExactMatch.exact_match_back_prop (SMEM/ExactMatch.py:132-151):
 * pattern i = d_pats[i*stride .. +len_i) with len_i = d_lens ? d_lens[i] : fixed_len;
 * d_out_lohi[2i..2i+1] = inclusive SA interval, or (-1,-1) if absent; an empty pattern gives
 * (0, n) like the reference; a code > 3 gives (-2,-2) (reference: KeyError). */
int genie_sa_interval(const genie_index *ix, const uint8_t *d_pats, const int32_t *d_lens, int64_t N,
                      int32_t stride, int32_t fixed_len, int32_t *d_out_lohi, void *stream);

/* Batched seed lookup of one K-mer each (N rows of K codes, row stride K):
 * mode LUT: `lut[str(code)][0]` membership + interval (SMEM/SMEM.py:28-32,65-67);
 * mode RMI: RMI_LUT.get_suffix_rmi = predict + last-mile search (SMEM/RMI_LUT.py:67-184),
 * contract behaviour = the true interval; an absent K-mer is reported the reference's way,
 * lower > upper (lower = the row it would be inserted at).  LUT output as genie_sa_interval.
 * d_pred (may be NULL,
 * RMI only) receives the float64 prediction of RMI_LUT.rmi_predict (SMEM/RMI_LUT.py:53-63). */
int genie_seed_lookup(const genie_index *ix, int32_t mode, const uint8_t *d_kmers, int64_t N,
                      int32_t *d_out_lohi, double *d_pred, void *stream);

/* Batched SMEM discovery: replaces SMEM.get_SMEMS / get_smems_lut / get_smems_rmi.
 * Read r = d_reads[r*stride .. +len_r), len_r = d_lens ? d_lens[r] : fixed_len.
 * Writes d_counts[r] = number of SMEMs of read r (after the min_len filter, which the
 * reference applies in BWA mode only, SMEM.py:463; pass 1 otherwise) and
 * d_slots[(r*cap + t)*4 + {0,1,2,3}] = (start, end, lo, hi) of its t-th SMEM in the
 * reference's emission order: the substring read[start:end) and its SA interval [lo, hi].
 * d_status[r] (may be NULL) = GENIE_READ_* code.  `cap` slots per read (cap >= max read
 * length never overflows).
 * d_workspace: 256-byte aligned device scratch of genie_find_smems_workspace_bytes(N, max_len)
 * bytes (matching statistics, packed reads and emitted (start, end) pairs handed between
 * the kernels of the pipeline). */
int64_t genie_find_smems_workspace_bytes(int64_t N, int32_t max_len);
/* Per-read rows of that workspace, for traffic accounting: out[0] = bytes of a matching-statistics (fwd) row,
 * out[1] = 16-byte pieces of the packed read (reads of up to 255 bases: plain 64-bit words, two per piece; longer: overlapping
 * records {w[i], w[i+1]}), out[2] = 0 (reserved), out[3] = bytes of the emitted
 * (count, (start, end) pairs) row. */
int genie_find_smems_workspace_rows(int32_t max_len, int32_t *row_bytes4);
int genie_find_smems(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens,
                     int64_t N, int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts,
                     int32_t *d_slots, int32_t cap, int32_t *d_status, void *d_workspace, int64_t workspace_bytes,
                     void *stream);

/* Same discovery, CSR output in one call: d_offsets[N+1] = exclusive prefix sum of the per-read SMEM
 * counts, d_rows[4*t .. 4*t+3] = (start, end, lo, hi) of SMEM t, reads in input order, SMEMs in the
 * reference's emission order.  Every row is written exactly once; rows beyond out_cap_rows are
 * dropped (the caller compares d_offsets[N] with its capacity).  Flagged reads contribute no rows. */
int genie_find_smems_csr(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens,
                         int64_t N, int32_t stride, int32_t fixed_len, int32_t min_len, int64_t *d_offsets,
                         int32_t *d_rows, int64_t out_cap_rows, int32_t *d_status, void *d_workspace,
                         int64_t workspace_bytes, void *stream);

/* The same discovery for callers on the far side of a host link (SMEM.find_smems_* on host arrays): 2-bit packed reads in,
 * 8-byte rows out -- 40 instead of 150 bytes per 150-base read over PCIe, 8 instead of 16 per SMEM.  Reads of at most 255
 * bases.  Row r of d_reads2bit = stride_bytes bytes (a multiple of 4, >= 4 * ceil(max length / 16)): byte i holds bases
 * 4i .. 4i+3, base 4i in bits 7..6 (codes as above; bases past the read's length are ignored).  Output:
 *   d_counts8[r]  SMEMs of read r (flagged reads: 0);  d_status8[r] = its GENIE_READ_* code;
 *   d_rows8       dense, reads in input order, SMEMs in emission order (the CSR rows of genie_find_smems_csr; a read's
 *                 rows start at the sum of the counts before it), 8 bytes each: byte 0 start, byte 1 end, bytes 2..3
 *                 span = hi - lo (little endian), bytes 4..7 lo.  span == 0xFFFF means "65535 or more": that row's
 *                 index and its hi are also appended to d_escapes (int64 pairs: row, hi; unordered);
 *   d_totals[0]   rows in all (rows beyond out_cap_rows were dropped);  d_totals[1] = escapes in all (compare with
 *                 cap_escapes and call again with a larger list if it is exceeded).
 * The reference has no counterpart (its API is in-process Python strings); genie-smem_amd/packing.py holds the host side:
 * pack_reads() and unpack_rows() give back exactly the int32 (start, end, lo, hi) rows of genie_find_smems_csr. */
int genie_find_smems_packed(const genie_index *ix, int32_t mode, const uint8_t *d_reads2bit, const int32_t *d_lens, int64_t N,
                            int32_t stride_bytes, int32_t fixed_len, int32_t min_len, uint8_t *d_counts8, uint8_t *d_status8,
                            void *d_rows8, int64_t out_cap_rows, int64_t *d_totals, int64_t *d_escapes, int64_t cap_escapes,
                            void *d_workspace, int64_t workspace_bytes, void *stream);

/* The same with 6-byte rows (another quarter off the bytes that travel back: 66 instead of 88 per 150-base read), for
 * references below 2^24 bases (GENIE_E_TOO_LONG otherwise): byte 0 start, byte 1 end, bytes 2..4 lo (24 bits, little endian),
 * byte 5 span = hi - lo, 0xFF meaning "255 or more" (that row's index and hi are on d_escapes).  d_rows6: 2-byte aligned.
 * Everything else as genie_find_smems_packed; packing.unpack_rows(..., row_bytes=6) is the host side. */
int genie_find_smems_packed6(const genie_index *ix, int32_t mode, const uint8_t *d_reads2bit, const int32_t *d_lens, int64_t N,
                             int32_t stride_bytes, int32_t fixed_len, int32_t min_len, uint8_t *d_counts8, uint8_t *d_status8,
                             void *d_rows6, int64_t out_cap_rows, int64_t *d_totals, int64_t *d_escapes, int64_t cap_escapes,
                             void *d_workspace, int64_t workspace_bytes, void *stream);

/* Compact the slotted output to CSR: d_offsets[N+1] (exclusive prefix sum of min(count,cap))
 * and d_out[total*4].  d_tmp: scratch of genie_compact_tmp_bytes(N) bytes. */
int64_t genie_compact_tmp_bytes(int64_t N);
int genie_compact_smems(const int32_t *d_counts, const int32_t *d_slots, int64_t N, int32_t cap,
                        int64_t *d_offsets, int32_t *d_out, int64_t out_cap_rows, void *d_tmp, void *stream);

/* Rows -> reference coordinates: ExactMatch.get_positions (SMEM/ExactMatch.py:195-199) for S intervals at
 * once.  Interval t is (d_lohi[t*stride], d_lohi[t*stride + 1]) = inclusive rows (lo, hi); lo < 0 or
 * hi < lo (absent) contributes nothing.  Pass the (lo, hi) columns of a find_smems result as
 * d_rows + 2 with stride 4, or a genie_sa_interval result with stride 2.  Output CSR:
 * d_pos_offsets[S+1], d_positions[...] = the suffix-array entries of rows lo..hi, 1-based like the
 * reference's, in row order (ExactMatch.exact_match sorts them: :174-192).  Entries beyond cap_positions
 * are dropped (compare d_pos_offsets[S] with the capacity).  d_tmp: 256-byte aligned scratch of
 * genie_locate_tmp_bytes(S) bytes. */
int64_t genie_locate_tmp_bytes(int64_t S);
int genie_locate(const genie_index *ix, const int32_t *d_lohi, int32_t stride, int64_t S, int64_t *d_pos_offsets,
                 int32_t *d_positions, int64_t cap_positions, void *d_tmp, int64_t tmp_bytes, void *stream);

/* Launch-time options of an index handle (none of them changes results).
 * GENIE_OPT_SEARCH_ALL (default 0): the matching statistics fwd[] are non-decreasing along a read, so by default
 *   the match-statistics kernel looks up every 4th position and the three between two of them only where their
 *   values differ.  Value 1 looks up every position (differential testing, A/B timing).
 * GENIE_OPT_GROUP_POSITIONS (default 0 = built-in): read positions a wave works on per iteration (tuning).
 * GENIE_OPT_SEARCH_BLOCKS_PER_CU (default 0 = as many as fit): cap on resident blocks of that kernel (tuning).
 * GENIE_OPT_SEARCH_ONLY (default 0): launch the match-statistics kernel only -- outputs are NOT produced and the
 *   find_smems entry points return GENIE_W_SEARCH_ONLY instead of GENIE_OK; for timing that kernel alone.
 * GENIE_OPT_SEARCH_STAGES_OFF (default 0; honoured only while GENIE_OPT_SEARCH_ONLY is set, so never on a run that
 *   produces output): bit mask of stages of that kernel to skip -- 1 slow list, 2 round 2, 4 rounds 1+2, 8 packed-read
 *   records, 32 fwd rows; the stage ablation of DESIGN.md section 4 (tools/ka_sweep.sh).
 * GENIE_OPT_SCHEDULING (default 0; A/B timing, results unchanged): bit mask -- 1: the match-statistics kernel gives every
 *   wave a fixed share of the read groups instead of handing them out per block; 2: its waves keep one issue priority
 *   instead of rotating it; 4: the same for the interval kernel; 8: the match-statistics kernels do not ask for their next
 *   group's input rows through the scalar cache ahead of time. */
enum { GENIE_OPT_SEARCH_ALL = 2, GENIE_OPT_GROUP_POSITIONS = 4, GENIE_OPT_SEARCH_ONLY = 5, GENIE_OPT_SEARCH_BLOCKS_PER_CU = 6,
       GENIE_OPT_SEARCH_STAGES_OFF = 7, GENIE_OPT_SCHEDULING = 8 };
int genie_index_set_option(genie_index *ix, int32_t option, int32_t value);

/* Profiling hook: two hipEvent_t (as void*, created by the caller with timing enabled) that the next
 * genie_find_smems calls record on their stream immediately before and after the dominant kernel of
 * the path (the match-statistics kernel, match_table_kernel / match_table_long_kernel).
 * Pass NULLs to stop.  Not thread-safe with concurrent launches on the same handle. */
int genie_index_set_stage_events(genie_index *ix, void *ev_search_begin, void *ev_search_end);

/* Launch geometry actually used by genie_find_smems for (mode, max read length): for reports. */
int genie_launch_info(const genie_index *ix, int32_t mode, int32_t max_len, int32_t *grid, int32_t *block,
                      int32_t *lds_bytes);

/* Name of the match-statistics (dominant) kernel genie_find_smems launches for (mode, max read length), as a
 * profiler prints it without the argument list: for reports that look the kernel up in a rocprofv3 trace. */
int genie_search_kernel_name(const genie_index *ix, int32_t mode, int32_t max_len, char *buf, int32_t cap);

const char *genie_strerror(int status);
const char *genie_last_hip_error(void);
int genie_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GENIE_SMEM_H */

/*
 * smem_oracle.h -- CPU oracle for the GENIE-SMEM hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (genie-smem_amd/)
 * never links, imports or calls it.
 *
 * It is a plain-C restatement of the reference's algorithms, function by function
 * (each one cites the reference file:line it follows), deliberately kept in the
 * reference's own shape -- FM-index backward search, forward/backward extension that
 * re-searches the growing string, the K-frame state machine with explicit position
 * lists -- so that it is an independent check on the GPU path, which computes the same
 * answers by a different route (suffix-array bound searches + matching statistics).
 *
 * Parity pin: tests/test_oracle_golden.py checks this oracle against golden vectors that
 * tests/golden/make_golden.py produced by running the unmodified reference in the build
 * container (see that script's header for how).
 *
 * Conventions (same as the reference):
 *   - bases are codes 0..3 in the sorted order of the reference's alphabet (ACGT -> 0..3,
 *     LUT.py:37-48); '$' is the terminator and sorts lowest;
 *   - the suffix array has n+1 rows, row 0 is the '$' suffix; intervals are 0-based,
 *     inclusive (ExactMatch.py:151); positions are 1-based (ExactMatch.py:66);
 *   - modes: 0 = BWA-SMEM (SMEM.py:456), 1 = LUT-SMEM (SMEM.py:20), 2 = RMI-SMEM (SMEM.py:206).
 */
#ifndef SMEM_ORACLE_H
#define SMEM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_index orc_index;

enum {
    ORC_OK = 0,
    ORC_ABSENT = -1,    /* pattern does not occur (the reference returns int -1) */
    ORC_EKEY = -2,      /* base not in the reference's alphabet (reference: KeyError) */
    ORC_ESHORT = -3,    /* read shorter than K in LUT/RMI mode (reference mis-encodes; rejected) */
    ORC_ERUNAWAY = -4,  /* reference would loop forever / KeyError('') (base absent from ref) */
    ORC_ECAP = -5,      /* output capacity too small */
    ORC_ENOMODEL = -6,  /* RMI mode without a model */
    ORC_ERECURSE = -7   /* compat RMI search: the reference would hit RecursionError */
};

/* Build SA + FM index (+ K-mer LUT when K > 0) for `codes[0..n)`. */
orc_index *orc_index_build(const uint8_t *codes, int64_t n, int K);
void orc_index_free(orc_index *ix);

int64_t orc_n(const orc_index *ix);
int orc_K(const orc_index *ix);
/* 1-based suffix array exactly as the reference stores it (row 0 = n+1). */
const int32_t *orc_suffix_array(const orc_index *ix);
/* FM pieces, reference layout: bwt code per row (4 = '$'), count_dic first rows, Occ. */
const uint8_t *orc_bwt(const orc_index *ix);
int32_t orc_count(const orc_index *ix, int code /*0..3, 4='$', 5=''*/);
const int32_t *orc_occ(const orc_index *ix, int code /*0..3, 4='$'*/);

/* LUT (LUT.py:15-35) as sorted arrays: codes[m], lo[m], hi[m]; positions of entry i are
 * suffix_array[lo[i]..hi[i]] (1-based, SA order), as LUT.py:33 stores them. */
int64_t orc_lut_size(const orc_index *ix);
const uint32_t *orc_lut_codes(const orc_index *ix);
const int32_t *orc_lut_lo(const orc_index *ix);
const int32_t *orc_lut_hi(const orc_index *ix);

/* ExactMatch.exact_match_back_prop (ExactMatch.py:132-151). */
int orc_back_prop(const orc_index *ix, const uint8_t *pat, int m, int32_t *lo, int32_t *hi);
/* ExactMatch.exact_match_back_prop_add_one (ExactMatch.py:155-171). */
int orc_back_prop_add_one(const orc_index *ix, int code, int32_t *lo, int32_t *hi);

/* RMI model (RMI.py:52-69): nlev levels; level l has sizes[l] models (sizes[0] == 1) and the
 * clamp scale scales[l] (= experts + [1]).  coef/icpt are the concatenated per-level arrays. */
int orc_set_rmi(orc_index *ix, int nlev, const int32_t *sizes, const int32_t *scales,
                const double *coef, const double *icpt);
/* RMI_LUT.rmi_predict (RMI_LUT.py:53-63): float64 prediction for one K-mer code. */
double orc_rmi_predict(const orc_index *ix, uint64_t code);
/* K-mer -> interval through the RMI: predict, then last-mile search.
 * compat = 0: contract behaviour -- always the true interval (SURVEY 8a, A8 decision);
 * compat = 1: literal replay of RMI_LUT.exponential_search/binary_search (RMI_LUT.py:95-184)
 *             including its defects; returns ORC_ERECURSE where the reference recurses forever.
 * Output is the reference's convention: absent <=> *lower > *upper. */
int orc_rmi_suffix(const orc_index *ix, const uint8_t *kmer, int compat, int64_t *lower, int64_t *upper);

/* One read.  out receives rows (start, end, lo, hi) in emission order; returns the number
 * of SMEMs (>= 0) or a negative ORC_E* code. */
int orc_find_smems(const orc_index *ix, int mode, const uint8_t *read, int L, int min_len,
                   int32_t *out, int cap);
/* Batch over reads (OpenMP when nthreads > 1): reads[N][stride], lens may be NULL (=> L). */
void orc_find_smems_batch(const orc_index *ix, int mode, const uint8_t *reads, int64_t N, int32_t stride,
                          const int32_t *lens, int32_t L, int min_len, int32_t *counts, int32_t *out,
                          int cap, int nthreads);

#ifdef __cplusplus
}
#endif
#endif

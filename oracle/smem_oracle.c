/*
 * smem_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see smem_oracle.h).
 *
 * Plain-C restatement of the reference's hot path, kept in the reference's own shape.
 * Every function names the reference lines it follows.  Pinned against the golden vectors
 * under tests/golden/ (produced by running the unmodified reference).
 */
#define _GNU_SOURCE
#include "smem_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct orc_index {
    int64_t n;
    int K;
    uint8_t *codes;      /* n bases */
    int32_t *sa1;        /* n+1 rows, 1-based start positions, row 0 = n+1 (ExactMatch.py:66) */
    uint8_t *bwt;        /* n+1, last column of the sorted rotations; 4 = '$' */
    int32_t *occ[5];     /* inclusive cumulative counts per BWT char (ExactMatch.py:71-90) */
    int32_t cnt[6];      /* count_dic: first row whose first column is the char; [5] = n+1 */
    int64_t lut_m;
    uint32_t *lut_code;
    int32_t *lut_lo, *lut_hi;
    int nlev;
    int32_t *sizes, *scales;
    int64_t *lvl_off;
    double *coef, *icpt;
};

/* ------------------------------------------------------------------ suffix array + FM */

typedef struct { uint32_t code; int32_t lo, hi; } ent_t;

static int cmp_ent(const void *a, const void *b)
{
    uint32_t x = ((const ent_t *)a)->code, y = ((const ent_t *)b)->code;
    return x < y ? -1 : x > y;
}

/* ExactMatch.create_bwt_matrix (ExactMatch.py:52-68) sorts all n+1 rotations of ref+"$".
 * Because '$' is unique and smallest, rotation order == suffix order, so we sort suffixes. */
static int cmp_suffix(const void *pa, const void *pb, void *arg)
{
    const orc_index *ix = (const orc_index *)arg;
    int64_t a = *(const int32_t *)pa, b = *(const int32_t *)pb;
    if (a == b) return 0;
    int64_t la = ix->n - a, lb = ix->n - b;
    int64_t m = la < lb ? la : lb;
    int c = m ? memcmp(ix->codes + a, ix->codes + b, (size_t)m) : 0;
    if (c) return c;
    return la < lb ? -1 : 1;               /* the shorter one meets '$' first */
}

orc_index *orc_index_build(const uint8_t *codes, int64_t n, int K)
{
    orc_index *ix = (orc_index *)calloc(1, sizeof(*ix));
    ix->n = n;
    ix->K = K;
    ix->codes = (uint8_t *)malloc((size_t)n + 1);
    memcpy(ix->codes, codes, (size_t)n);
    int64_t rows = n + 1;
    int32_t *start = (int32_t *)malloc(sizeof(int32_t) * rows);
    for (int64_t i = 0; i < rows; i++) start[i] = (int32_t)i;
    qsort_r(start, (size_t)rows, sizeof(int32_t), cmp_suffix, ix);
    ix->sa1 = (int32_t *)malloc(sizeof(int32_t) * rows);
    ix->bwt = (uint8_t *)malloc((size_t)rows);
    for (int c = 0; c < 5; c++) ix->occ[c] = (int32_t *)calloc((size_t)rows, sizeof(int32_t));
    for (int c = 0; c < 6; c++) ix->cnt[c] = -1;
    int32_t run[5] = {0, 0, 0, 0, 0};
    for (int64_t r = 0; r < rows; r++) {
        int64_t s = start[r];
        ix->sa1[r] = (int32_t)(s + 1);                     /* ref_size - rotation.index("$") */
        int last = s == 0 ? 4 : codes[s - 1];              /* rotation[-1] */
        int first = s == n ? 4 : codes[s];                 /* rotation[0]  */
        ix->bwt[r] = (uint8_t)last;
        run[last]++;
        for (int c = 0; c < 5; c++) ix->occ[c][r] = run[c];
        if (ix->cnt[first] < 0) ix->cnt[first] = (int32_t)r; /* create_count_dic :93-101 */
    }
    ix->cnt[5] = (int32_t)rows;                            /* count_dic[""] */
    free(start);

    /* LUT.generate_lut (LUT.py:15-35): every K-mer of the reference -> (interval, positions).
     * As in the reference the interval comes from exact_match_back_prop of the K-mer. */
    if (K > 0 && n >= K) {
        int64_t cntk = n - K + 1;
        ent_t *e = (ent_t *)malloc(sizeof(ent_t) * cntk);
        for (int64_t p = 0; p < cntk; p++) {
            uint32_t code = 0;
            for (int j = 0; j < K; j++) code = (code << 2) | codes[p + j]; /* LUT.py:37-48 */
            e[p].code = code;
            orc_back_prop(ix, codes + p, K, &e[p].lo, &e[p].hi);
        }
        /* sort by code (stable order irrelevant: equal codes have equal intervals) */
        qsort(e, (size_t)cntk, sizeof(ent_t), cmp_ent);
        int64_t m = 0;
        for (int64_t i = 0; i < cntk; i++)
            if (i == 0 || e[i].code != e[i - 1].code) e[m++] = e[i];
        ix->lut_m = m;
        ix->lut_code = (uint32_t *)malloc(sizeof(uint32_t) * m);
        ix->lut_lo = (int32_t *)malloc(sizeof(int32_t) * m);
        ix->lut_hi = (int32_t *)malloc(sizeof(int32_t) * m);
        for (int64_t i = 0; i < m; i++) {
            ix->lut_code[i] = e[i].code;
            ix->lut_lo[i] = e[i].lo;
            ix->lut_hi[i] = e[i].hi;
        }
        free(e);
    }
    return ix;
}

void orc_index_free(orc_index *ix)
{
    if (!ix) return;
    free(ix->codes); free(ix->sa1); free(ix->bwt);
    for (int c = 0; c < 5; c++) free(ix->occ[c]);
    free(ix->lut_code); free(ix->lut_lo); free(ix->lut_hi);
    free(ix->sizes); free(ix->scales); free(ix->lvl_off); free(ix->coef); free(ix->icpt);
    free(ix);
}

int64_t orc_n(const orc_index *ix) { return ix->n; }
int orc_K(const orc_index *ix) { return ix->K; }
const int32_t *orc_suffix_array(const orc_index *ix) { return ix->sa1; }
const uint8_t *orc_bwt(const orc_index *ix) { return ix->bwt; }
int32_t orc_count(const orc_index *ix, int code) { return ix->cnt[code]; }
const int32_t *orc_occ(const orc_index *ix, int code) { return ix->occ[code]; }
int64_t orc_lut_size(const orc_index *ix) { return ix->lut_m; }
const uint32_t *orc_lut_codes(const orc_index *ix) { return ix->lut_code; }
const int32_t *orc_lut_lo(const orc_index *ix) { return ix->lut_lo; }
const int32_t *orc_lut_hi(const orc_index *ix) { return ix->lut_hi; }

/* ------------------------------------------------------------------ A1: backward search */

/* One backward-search step, shared by both entry points below (ExactMatch.py:139-149 and
 * :159-169 are the same arithmetic): start/end are the reference's 1-based values. */
static int back_step(const orc_index *ix, int c, int32_t *start, int32_t *end)
{
    if (c < 0 || c > 3 || ix->cnt[c] < 0) return ORC_EKEY;      /* count_dic[char] KeyError */
    int32_t cc = ix->cnt[c];
    if (*start - 1 <= 0) *start = cc + 1;
    else *start = cc + 1 + ix->occ[c][*start - 2];
    *end = cc + ix->occ[c][*end - 1];
    return *start > *end ? ORC_ABSENT : ORC_OK;
}

/* ExactMatch.exact_match_back_prop (ExactMatch.py:132-151) */
int orc_back_prop(const orc_index *ix, const uint8_t *pat, int m, int32_t *lo, int32_t *hi)
{
    int32_t start = 1, end = ix->cnt[5];
    for (int i = m - 1; i >= 0; i--) {
        int rc = back_step(ix, pat[i], &start, &end);
        if (rc) { *lo = *hi = -1; return rc; }
    }
    *lo = start - 1;
    *hi = end - 1;
    return ORC_OK;
}

/* ExactMatch.exact_match_back_prop_add_one (ExactMatch.py:155-171) */
int orc_back_prop_add_one(const orc_index *ix, int code, int32_t *lo, int32_t *hi)
{
    int32_t start = *lo + 1, end = *hi + 1;
    int rc = back_step(ix, code, &start, &end);
    if (rc) { *lo = *hi = -1; return rc; }
    *lo = start - 1;
    *hi = end - 1;
    return ORC_OK;
}

/* ------------------------------------------------------------------ A6: LUT lookup */

/* `encoded_sub in self.lut.lut` / `self.lut.lut[encoded_sub]` (SMEM.py:28-32, 65-67) */
static int64_t lut_find(const orc_index *ix, const uint8_t *q, int K)
{
    uint32_t code = 0;
    for (int j = 0; j < K; j++) {
        if (q[j] > 3) return -2;                                  /* LUT.py:47 KeyError */
        code = (code << 2) | q[j];
    }
    int64_t lo = 0, hi = ix->lut_m;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (ix->lut_code[mid] < code) lo = mid + 1; else hi = mid;
    }
    return (lo < ix->lut_m && ix->lut_code[lo] == code) ? lo : -1;
}

/* ------------------------------------------------------------------ A8: RMI */

int orc_set_rmi(orc_index *ix, int nlev, const int32_t *sizes, const int32_t *scales,
                const double *coef, const double *icpt)
{
    free(ix->sizes); free(ix->scales); free(ix->lvl_off); free(ix->coef); free(ix->icpt);
    ix->nlev = nlev;
    ix->sizes = (int32_t *)malloc(sizeof(int32_t) * nlev);
    ix->scales = (int32_t *)malloc(sizeof(int32_t) * nlev);
    ix->lvl_off = (int64_t *)malloc(sizeof(int64_t) * (nlev + 1));
    int64_t tot = 0;
    for (int l = 0; l < nlev; l++) {
        ix->sizes[l] = sizes[l];
        ix->scales[l] = scales[l];
        ix->lvl_off[l] = tot;
        tot += sizes[l];
    }
    ix->lvl_off[nlev] = tot;
    ix->coef = (double *)malloc(sizeof(double) * tot);
    ix->icpt = (double *)malloc(sizeof(double) * tot);
    memcpy(ix->coef, coef, sizeof(double) * tot);
    memcpy(ix->icpt, icpt, sizeof(double) * tot);
    return ORC_OK;
}

/* RMI.predict (RMI.py:52-69) for a single key: at each level p = coef*x + intercept
 * (sklearn LinearRegression.predict, one feature), next expert = min(scale-1, max(0, int(p)))
 * (:66).  Compiled with -ffp-contract=off: a multiply then an add, each rounded once. */
double orc_rmi_predict(const orc_index *ix, uint64_t code)
{
    double x = (double)code, p = 0.0;
    int64_t idx = 0;
    for (int l = 0; l < ix->nlev; l++) {
        int64_t o = ix->lvl_off[l] + idx;
        double prod = ix->coef[o] * x;
        p = prod + ix->icpt[o];
        int32_t scale = ix->scales[l];
        if (!(p > 0.0)) idx = 0;                                   /* max(0, int(p)); int() truncates */
        else if (p >= (double)scale) idx = scale - 1;
        else idx = (int64_t)p;
    }
    return p;
}

/* Compare the K-mer with the suffix of SA row r over its first K symbols, '$' smallest.
 * <0: suffix < kmer, 0: kmer is a prefix of the suffix, >0: suffix > kmer. */
static int cmp_row_kmer(const orc_index *ix, int64_t r, const uint8_t *kmer, int K)
{
    int64_t s = (int64_t)ix->sa1[r] - 1, avail = ix->n - s;
    int64_t m = avail < K ? avail : K;
    int c = m ? memcmp(ix->codes + s, kmer, (size_t)m) : 0;
    if (c) return c;
    return avail < K ? -1 : 0;
}

/* get_ref_seq (RMI_LUT.py:89-92) with Python list-index semantics for `ind`:
 * returns 0 and *s (0-based start) for a K-mer row, 1 for "None", -1 for IndexError. */
static int py_ref_seq(const orc_index *ix, int64_t ind, int64_t *s)
{
    int64_t rows = ix->n + 1;
    if (ind < 0) ind += rows;
    if (ind < 0 || ind >= rows) return -1;
    int64_t p = ix->sa1[ind];
    if (p - 1 + ix->K > ix->n) return 1;
    *s = p - 1;
    return 0;
}

#define PY_NONE 1
#define PY_ERR (-1)

/* RMI_LUT.binary_search (RMI_LUT.py:95-133), literal, with a recursion guard.
 * rc: 0 ok, ORC_ERECURSE, or 1 for a Python TypeError/IndexError. */
static int compat_bsearch(const orc_index *ix, const uint8_t *q, int64_t lower, int64_t upper, int strict,
                          int depth, int64_t *res)
{
    int K = ix->K;
    int64_t s;
    int st;
    if (depth > 900) return ORC_ERECURSE;
    if (lower == upper) { *res = lower; return 0; }
    if (upper - lower == 1) {
        if (strict) {
            st = py_ref_seq(ix, upper, &s);
            if (st == PY_ERR) return 1;
            *res = (st == 0 && memcmp(ix->codes + s, q, (size_t)K) == 0) ? upper : lower;
        } else {
            st = py_ref_seq(ix, lower, &s);
            if (st == PY_ERR) return 1;
            *res = (st == 0 && memcmp(ix->codes + s, q, (size_t)K) == 0) ? lower : upper;
        }
        return 0;
    }
    int64_t sum = lower + upper;
    int64_t mid = sum >= 0 ? sum / 2 : -((-sum + 1) / 2);          /* Python floor division */
    st = py_ref_seq(ix, mid, &s);
    if (st == PY_ERR) return 1;
    while (st == PY_NONE && mid > lower) {
        mid -= 1;
        st = py_ref_seq(ix, mid, &s);
        if (st == PY_ERR) return 1;
        if (mid == lower) {
            if (strict) {
                int64_t su;
                int stu = py_ref_seq(ix, upper, &su);
                if (stu == PY_ERR) return 1;
                *res = (stu == 0 && memcmp(ix->codes + su, q, (size_t)K) == 0) ? upper : lower;
            } else {
                *res = (st == 0 && memcmp(ix->codes + s, q, (size_t)K) == 0) ? lower : upper;
            }
            return 0;
        }
    }
    if (st == PY_NONE) return 1;                                    /* None < str: TypeError */
    int c = memcmp(ix->codes + s, q, (size_t)K);
    if (c < 0 || (c == 0 && strict)) return compat_bsearch(ix, q, mid, upper, strict, depth + 1, res);
    return compat_bsearch(ix, q, lower, mid, strict, depth + 1, res);
}

/* RMI_LUT.exponential_search (RMI_LUT.py:136-184), literal. */
static int compat_exp_search(const orc_index *ix, const uint8_t *q, int64_t start_sa, int64_t *lo, int64_t *hi)
{
    int K = ix->K;
    int64_t rows = ix->n + 1, s;
    int have_l = 0, have_u = 0;
    int64_t lower = 0, upper = 0;
    int st = py_ref_seq(ix, start_sa, &s);
    long guard = 0;
    while (st == PY_NONE) {
        start_sa += 1;
        st = py_ref_seq(ix, start_sa, &s);
        if (++guard > rows + 2) return 1;
    }
    if (st == PY_ERR) return 1;
    int c = memcmp(ix->codes + s, q, (size_t)K);
    if (c < 0) { lower = start_sa; have_l = 1; }
    else if (c > 0) { upper = start_sa; have_u = 1; }
    int64_t window = 1;
    if (!have_u) {
        while (start_sa + window < ix->n + 1) {
            int64_t ind = start_sa + window;
            window *= 2;
            st = py_ref_seq(ix, ind, &s);
            while (st == PY_NONE) { ind += 1; st = py_ref_seq(ix, ind, &s); }
            if (st == PY_ERR) return 1;
            c = memcmp(ix->codes + s, q, (size_t)K);
            if (c > 0) { upper = ind; have_u = 1; break; }
            if (c < 0) { lower = ind; have_l = 1; }
        }
    }
    window = 1;
    if (!have_l) {
        while (start_sa - window >= 0) {
            int64_t ind = start_sa - window;
            window *= 2;
            st = py_ref_seq(ix, ind, &s);
            guard = 0;
            while (st == PY_NONE) {
                ind -= 1;
                st = py_ref_seq(ix, ind, &s);
                if (++guard > 2 * rows + 2) return 1;
            }
            if (st == PY_ERR) return 1;
            c = memcmp(ix->codes + s, q, (size_t)K);
            if (c < 0) { lower = ind; have_l = 1; break; }
            if (c > 0) { upper = ind; have_u = 1; }
        }
    }
    if (!have_l) lower = 0;
    if (!have_u) upper = rows - 1;
    int rc = compat_bsearch(ix, q, lower, upper, 0, 0, lo);
    if (rc) return rc;
    return compat_bsearch(ix, q, lower, upper, 1, 0, hi);
}

/* RMI_LUT.get_suffix_rmi (RMI_LUT.py:67-78): predict, int() truncate, last-mile search. */
int orc_rmi_suffix(const orc_index *ix, const uint8_t *kmer, int compat, int64_t *lower, int64_t *upper)
{
    if (!ix->nlev) return ORC_ENOMODEL;
    int K = ix->K;
    uint64_t code = 0;
    for (int j = 0; j < K; j++) {
        if (kmer[j] > 3) return ORC_EKEY;                           /* RMI_LUT.py:60 KeyError */
        code = (code << 2) | kmer[j];
    }
    double p = orc_rmi_predict(ix, code);
    int64_t rows = ix->n + 1;
    if (compat) {
        int64_t start = (int64_t)p;                                  /* int(start_sa) truncates */
        return compat_exp_search(ix, kmer, start, lower, upper);
    }
    /* Contract behaviour: the true interval.  Gallop out from the predicted
     * row to bracket the K-mer, then bound-search inside the bracket. */
    int64_t r = p > 0.0 ? (p >= (double)rows ? rows - 1 : (int64_t)p) : 0;
    int64_t lo = r, hi = r, step = 1;
    while (lo > 0 && cmp_row_kmer(ix, lo, kmer, K) >= 0) { lo = lo - step < 0 ? 0 : lo - step; step *= 2; }
    step = 1;
    while (hi < rows - 1 && cmp_row_kmer(ix, hi, kmer, K) <= 0) { hi = hi + step > rows - 1 ? rows - 1 : hi + step; step *= 2; }
    int64_t a = lo, b = hi + 1;                                      /* first row with suffix >= kmer */
    while (a < b) { int64_t m = (a + b) >> 1; if (cmp_row_kmer(ix, m, kmer, K) < 0) a = m + 1; else b = m; }
    int64_t first = a;
    a = first; b = hi + 1;                                           /* first row with suffix > kmer */
    while (a < b) { int64_t m = (a + b) >> 1; if (cmp_row_kmer(ix, m, kmer, K) <= 0) a = m + 1; else b = m; }
    *lower = first;
    *upper = a - 1;
    return ORC_OK;
}

/* ------------------------------------------------------------------ A3/A4: extensions */

typedef struct { int32_t lo, hi; } iv_t;

/* The reference's `forward_matches` dict {string: interval}; every key starts at the same
 * query index `key_start`, so a key is identified by its end index j (ascending = dict order). */
typedef struct {
    int n, key_start;
    int *j;
    iv_t *iv;
} fmatch_t;

/* A seed frame's K-mer source: mode 1 = LUT entry, mode 2 = RMI prediction. */
typedef struct { int hit; iv_t iv; } seed_t;

/* SMEM.forward_extension (SMEM.py:425-443).  `key_start` is where `largest` (the seed
 * K-mer, possibly empty) begins; extension proceeds from `start_index`.  Each step
 * re-searches the whole growing string (get_suffix_index -> exact_match_back_prop, :435).
 * Returns the end index of the longest match (== key_start when it is the empty string),
 * or a negative error. */
static int forward_extension(const orc_index *ix, const uint8_t *q, int L, int key_start, int start_index,
                             int seeded, iv_t seed_iv, fmatch_t *fm)
{
    fm->n = 0;
    fm->key_start = key_start;
    int longest = key_start;
    if (seeded) {                                                   /* :427-428 */
        fm->j[0] = start_index;
        fm->iv[0] = seed_iv;
        fm->n = 1;
        longest = start_index;
    }
    for (int i = start_index + 1; i <= L; i++) {                    /* :431 */
        iv_t iv;
        int rc = orc_back_prop(ix, q + key_start, i - key_start, &iv.lo, &iv.hi);
        if (rc == ORC_EKEY) return rc;
        if (rc == ORC_ABSENT) return i - 1;                         /* :437-438 currentSearch[:-1] */
        fm->j[fm->n] = i;
        fm->iv[fm->n] = iv;
        fm->n++;
        longest = i;
    }
    return longest;                                                 /* :443 hit end of query */
}

typedef struct { int len, k, end; iv_t iv; int valid; } cand_t;   /* (string q[k:end], interval, end) */

/* SMEM.backward_extension (SMEM.py:389-423). */
static int backward_extension(const orc_index *ix, const uint8_t *q, int start_index, const fmatch_t *fm,
                              cand_t *out)
{
    int largest_len = 0, largest_k = 0, end_index = -1;
    iv_t largest_iv = {-1, -1};
    int have_iv = 0;
    int lf_len = 0, lf_idx = -1;
    for (int t = 0; t < fm->n; t++) {                               /* for key in forward_matches */
        int j = fm->j[t];
        int keylen = j - fm->key_start;
        if (keylen > lf_len) { lf_len = keylen; lf_idx = t; }       /* :397-398 */
        iv_t iv = {-1, -1};
        int first = 1;
        for (int i = start_index - 1; i >= 0; i--) {                /* :402 */
            int rc;
            if (first) { rc = orc_back_prop(ix, q + i, j - i, &iv.lo, &iv.hi); first = 0; }   /* :406 */
            else rc = orc_back_prop_add_one(ix, q[i], &iv.lo, &iv.hi);                        /* :408 */
            if (rc == ORC_EKEY) return rc;
            if (rc == ORC_ABSENT) break;                            /* :410-411 */
            if (j - i > largest_len) {                              /* :413 strict > */
                largest_len = j - i;
                largest_k = i;
                largest_iv = iv;
                have_iv = 1;
                end_index = start_index + keylen;
            }
        }
    }
    if (lf_len > largest_len) {                                     /* :418-421 */
        largest_len = lf_len;
        largest_k = start_index;
        largest_iv = fm->iv[lf_idx];
        have_iv = 1;
        end_index = start_index + lf_len;
    }
    out->len = largest_len;
    out->k = largest_k;
    out->end = end_index;
    out->iv = largest_iv;
    out->valid = have_iv;
    return ORC_OK;
}

/* SMEM.get_SMEM_at_index (SMEM.py:469-484). */
static int smem_at_index(const orc_index *ix, const uint8_t *q, int L, int start, fmatch_t *fm, cand_t *out)
{
    iv_t none = {-1, -1};
    int fend = forward_extension(ix, q, L, start, start, 0, none, fm);
    if (fend < 0) return fend;
    int rc = backward_extension(ix, q, start, fm, out);
    if (rc) return rc;
    int flen = fend - start;
    if (flen > out->len) {                                          /* :481-482 */
        out->len = flen;
        out->k = start;
        out->end = flen + start;
        out->iv = fm->iv[fm->n - 1];
        out->valid = 1;
    }
    if (out->len == 0) return ORC_ERUNAWAY;     /* ["", None, -1]: reference never terminates */
    return ORC_OK;
}

/* SMEM.check_sequential (SMEM.py:196-202) on the two position lists (1-based SA values). */
static int check_sequential(const orc_index *ix, iv_t a, iv_t b)
{
    for (int32_t r1 = a.lo; r1 <= a.hi; r1++)
        for (int32_t r2 = b.lo; r2 <= b.hi; r2++)
            if (ix->sa1[r1] + 1 == ix->sa1[r2]) return 1;
    return 0;
}

/* The seed lookup of one frame: LUT membership + interval (SMEM.py:65-67) or the RMI
 * prediction with its `pred[1] >= pred[0]` hit test (SMEM.py:251-253). */
static int seed_lookup(const orc_index *ix, int mode, const uint8_t *kmer, seed_t *s)
{
    s->hit = 0;
    s->iv.lo = s->iv.hi = -1;
    if (mode == 1) {
        int64_t e = lut_find(ix, kmer, ix->K);
        if (e == -2) return ORC_EKEY;
        if (e >= 0) { s->hit = 1; s->iv.lo = ix->lut_lo[e]; s->iv.hi = ix->lut_hi[e]; }
        return ORC_OK;
    }
    int64_t lo, hi;
    int rc = orc_rmi_suffix(ix, kmer, 0, &lo, &hi);
    if (rc) return rc;
    if (hi >= lo) { s->hit = 1; s->iv.lo = (int32_t)lo; s->iv.hi = (int32_t)hi; }
    return ORC_OK;
}

#define EMIT(S, E, IV)                                                       \
    do {                                                                     \
        if (cnt >= cap) return ORC_ECAP;                                     \
        out[4 * cnt + 0] = (S); out[4 * cnt + 1] = (E);                      \
        out[4 * cnt + 2] = (IV).lo; out[4 * cnt + 3] = (IV).hi; cnt++;       \
    } while (0)

/* `if current is None or len(x) >= len(current)` -- the offer rule used throughout
 * get_smems_lut (e.g. SMEM.py:81, 88, 98, 114, 119, 135, 141, 155, 162, 168). */
#define OFFER(LEN, KK, EE, IV)                                               \
    do {                                                                     \
        if (!have || (LEN) >= cur.len) {                                     \
            cur.len = (LEN); cur.k = (KK); cur.end = (EE); cur.iv = (IV); have = 1; \
        }                                                                    \
    } while (0)

/* SMEM.get_SMEMS (SMEM.py:456-467). */
static int find_bwa(const orc_index *ix, const uint8_t *q, int L, int min_len, int32_t *out, int cap,
                    fmatch_t *fm)
{
    int cnt = 0, i = 0;
    while (i < L) {
        cand_t c;
        int rc = smem_at_index(ix, q, L, i, fm, &c);
        if (rc) return rc;
        if (c.len >= min_len) EMIT(c.k, c.end, c.iv);               /* :463-464 */
        i = c.end;                                                   /* :465 */
    }
    return cnt;
}

/* SMEM.get_smems_lut (SMEM.py:20-192) and get_smems_rmi (SMEM.py:206-384): one state
 * machine, the seed source differs (mode). */
static int find_seeded(const orc_index *ix, int mode, const uint8_t *q, int L, int32_t *out, int cap,
                       fmatch_t *fm)
{
    const int K = ix->K;
    int cnt = 0;
    if (L < K) return ORC_ESHORT;       /* SMEM.py:26-28 would mis-encode a short first K-mer */
    seed_t sd;
    int rc = seed_lookup(ix, mode, q, &sd);
    if (rc) return rc;
    iv_t none = {-1, -1};
    int fend;
    if (sd.hit) fend = forward_extension(ix, q, L, 0, K, 1, sd.iv, fm);      /* :31-32 */
    else fend = forward_extension(ix, q, L, 0, 0, 0, none, fm);              /* :35-36 */
    if (fend < 0) return fend;
    if (fend == 0) return ORC_ERUNAWAY;                 /* forward_match[0][""] KeyError, :39 */
    EMIT(0, fend, fm->iv[fm->n - 1]);                                          /* :38-39 */
    int prev_len = fend, end = fend;                                           /* :43-48 */

    while (end < L) {                                                          /* :49 */
        /* prev_frame: state 0 = None, 1 = (), 2 = frame(c, iv, forward flag); backward flag is
         * always True in the reference (:70, 73, 106, 124). */
        int pstate = 0, pc = -1, pfwd = 0;
        iv_t piv = none;
        cand_t cur = {0, 0, -1, none, 0};
        int have = 0;
        int prev_start = end - prev_len;                                       /* :54 */
        for (int i = 0; i < K; i++) {                                          /* :56 */
            if (i >= prev_len) continue;                                       /* :57-58 */
            int c = end - i;
            if (c + K > L) continue;                                           /* :62-63 */
            rc = seed_lookup(ix, mode, q + c, &sd);
            if (rc) return rc;
            if (sd.hit) {                                                      /* :67 Case 1,2,4 */
                if (pstate == 0) { pstate = 2; pc = c; piv = sd.iv; pfwd = 1; }         /* :68-70 */
                else if (pstate == 1) { pstate = 2; pc = c; piv = sd.iv; pfwd = 0; }    /* :71-73 */
                else {
                    if (check_sequential(ix, sd.iv, piv)) {                    /* :75 Case 1 */
                        if (pfwd) {                                            /* :77-84 */
                            int fe = forward_extension(ix, q, L, pc, pc + K, 1, piv, fm);
                            if (fe < 0) return fe;
                            cand_t b;
                            rc = backward_extension(ix, q, pc, fm, &b);
                            if (rc) return rc;
                            OFFER(b.len, b.k, b.end, b.iv);
                        } else {                                               /* :92-101 */
                            if (have && (pc - prev_start) + K < cur.len) continue;   /* :94-95 */
                            fm->n = 1; fm->key_start = pc; fm->j[0] = pc + K; fm->iv[0] = piv;
                            cand_t b;
                            rc = backward_extension(ix, q, pc, fm, &b);
                            if (rc) return rc;
                            OFFER(b.len, b.k, b.end, b.iv);
                        }
                        pc = c; piv = sd.iv; pfwd = 0;                          /* :106 */
                    } else {                                                   /* :108 Case 2 */
                        if (pfwd) {                                            /* :110-117 */
                            int fe = forward_extension(ix, q, L, pc, pc + K, 1, piv, fm);
                            if (fe < 0) return fe;
                            OFFER(fe - pc, pc, fe, fm->iv[fm->n - 1]);
                        } else {                                               /* :118-122 current */
                            OFFER(K, c, c + K, sd.iv);
                        }
                        pc = c; piv = sd.iv; pfwd = 0;                          /* :124 */
                    }
                }
            } else {
                if (pstate != 2) pstate = 1;                                   /* :126-128 */
                else {                                                         /* :129 Case 3 */
                    if (pfwd) {                                                /* :130-138 */
                        int fe = forward_extension(ix, q, L, pc, pc + K, 1, piv, fm);
                        if (fe < 0) return fe;
                        OFFER(fe - pc, pc, fe, fm->iv[fm->n - 1]);
                    } else {                                                   /* :140-144 */
                        OFFER(K, pc, pc + K, piv);
                    }
                    pstate = 1;                                                /* :146 */
                }
            }
        }
        if (pstate == 2) {                                                     /* :149 last frame */
            if (pfwd) {                                                        /* :150-158 */
                int fe = forward_extension(ix, q, L, pc, pc + K, 1, piv, fm);
                if (fe < 0) return fe;
                cand_t b;
                rc = backward_extension(ix, q, pc, fm, &b);
                if (rc) return rc;
                OFFER(b.len, b.k, b.end, b.iv);
            } else {                                                           /* :166-171 */
                fm->n = 1; fm->key_start = pc; fm->j[0] = pc + K; fm->iv[0] = piv;
                cand_t b;
                rc = backward_extension(ix, q, pc, fm, &b);
                if (rc) return rc;
                OFFER(b.len, b.k, b.end, b.iv);
            }
        }
        if (!have) {                                                           /* :175-179 */
            cand_t c2;
            rc = smem_at_index(ix, q, L, end, fm, &c2);
            if (rc) return rc;
            EMIT(c2.k, c2.end, c2.iv);
            end = c2.end;
            prev_len = c2.len;
        } else {                                                               /* :183-186 */
            EMIT(cur.k, cur.end, cur.iv);
            end = cur.end;
            prev_len = cur.len;
        }
    }
    return cnt;
}

int orc_find_smems(const orc_index *ix, int mode, const uint8_t *read, int L, int min_len, int32_t *out, int cap)
{
    if (mode == 2 && !ix->nlev) return ORC_ENOMODEL;
    for (int i = 0; i < L; i++)
        if (read[i] > 3) return ORC_EKEY;
    fmatch_t fm;
    fm.j = (int *)malloc(sizeof(int) * (size_t)(L + 2));
    fm.iv = (iv_t *)malloc(sizeof(iv_t) * (size_t)(L + 2));
    int rc = mode == 0 ? find_bwa(ix, read, L, min_len, out, cap, &fm)
                       : find_seeded(ix, mode, read, L, out, cap, &fm);
    free(fm.j);
    free(fm.iv);
    return rc;
}

void orc_find_smems_batch(const orc_index *ix, int mode, const uint8_t *reads, int64_t N, int32_t stride,
                          const int32_t *lens, int32_t L, int min_len, int32_t *counts, int32_t *out,
                          int cap, int nthreads)
{
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
#endif
    for (int64_t r = 0; r < N; r++) {
        int len = lens ? lens[r] : L;
        counts[r] = orc_find_smems(ix, mode, reads + r * (int64_t)stride, len, min_len,
                                   out + r * (int64_t)cap * 4, cap);
    }
}

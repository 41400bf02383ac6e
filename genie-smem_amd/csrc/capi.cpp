// capi.cpp -- the extern "C" surface declared in include/genie_smem.h.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <new>

#include "genie_internal.h"

using namespace genie;

namespace {

int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return GENIE_OK;
    set_hip_error(what, (int)e);
    return GENIE_E_HIP;
}

int query_cus(int device)
{
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return cus;
}

int make_index(const uint8_t *codes, int64_t n, const int32_t *sa1, int32_t K, int32_t P, int32_t P2, int32_t fmt, genie_index **out)
{
    if (!out) return GENIE_E_INVALID;
    HostIndex *h = nullptr;
    int rc = build_host_index(codes, n, sa1, K, P, P2, fmt, &h);
    if (rc) return rc;
    genie_index *ix = new (std::nothrow) genie_index();
    if (!ix) { delete h; return GENIE_E_NOMEM; }
    ix->host = h;
    fill_header(*h, &ix->hdr);
    ix->has_hdr = true;
    *out = ix;
    return GENIE_OK;
}

}  // namespace

static int ready(const genie_index *ix);

extern "C" {

int genie_abi_version(void) { return GENIE_ABI_VERSION; }

int genie_index_create(const uint8_t *codes, int64_t n, int32_t K, int32_t dir_bits, genie_index **out)
{
    return make_index(codes, n, nullptr, K, dir_bits, 0, 0, out);
}

int genie_index_create_ex(const uint8_t *codes, int64_t n, const int32_t *sa_one_based, int32_t K, int32_t dir_bits,
                          int32_t table_bits, genie_index **out)
{
    const int32_t P = (dir_bits <= 0 || dir_bits > GENIE_MAX_DIR_BITS) ? GENIE_MAX_DIR_BITS : dir_bits;   // as build_host_index
    const int32_t fmt = table_bits >> 8;                      // GENIE_TABLE_WIDE / GENIE_TABLE_COMPACT, 0 = automatic
    table_bits &= 0xFF;
    if (fmt < 0 || fmt > 2) return GENIE_E_INVALID;
    if (table_bits != 0 && (table_bits <= P || table_bits > 12)) return GENIE_E_INVALID;
    return make_index(codes, n, sa_one_based, K, dir_bits, table_bits, fmt, out);
}

int genie_index_create_from_sa(const uint8_t *codes, int64_t n, const int32_t *sa_one_based, int32_t K,
                               int32_t dir_bits, genie_index **out)
{
    if (!sa_one_based) return GENIE_E_INVALID;
    return make_index(codes, n, sa_one_based, K, dir_bits, 0, 0, out);
}

int genie_index_set_rmi(genie_index *ix, int32_t nlev, const int32_t *sizes, const int32_t *scales,
                        const double *coef, const double *icpt)
{
    if (!ix || !ix->host || ix->has_dev) return GENIE_E_INVALID;
    if (nlev < 1 || nlev > GENIE_MAX_RMI_LEVELS || !sizes || !scales || !coef || !icpt) return GENIE_E_INVALID;
    if (sizes[0] != 1) return GENIE_E_INVALID;
    HostIndex &h = *ix->host;
    int64_t tot = 0;
    for (int l = 0; l < nlev; l++) {
        if (sizes[l] < 1 || scales[l] < 1) return GENIE_E_INVALID;
        if (l + 1 < nlev && sizes[l + 1] != scales[l]) return GENIE_E_INVALID;   // next level has `scale` experts
        tot += sizes[l];
    }
    h.nlev = nlev;
    int64_t off = 0;
    for (int l = 0; l < GENIE_MAX_RMI_LEVELS; l++) {
        h.rmi_size[l] = l < nlev ? sizes[l] : 0;
        h.rmi_scale[l] = l < nlev ? scales[l] : 0;
        h.rmi_off[l] = (int32_t)off;
        if (l < nlev) off += sizes[l];
    }
    h.rmi_off[GENIE_MAX_RMI_LEVELS] = (int32_t)off;
    for (int l = nlev; l <= GENIE_MAX_RMI_LEVELS; l++) h.rmi_off[l] = (int32_t)tot;
    h.rmi.resize((size_t)tot);
    for (int64_t i = 0; i < tot; i++) h.rmi[(size_t)i] = RmiModel{coef[i], icpt[i]};
    h.rmi_err.clear();                       // error bounds belong to a natively trained model
    fill_header(h, &ix->hdr);
    return GENIE_OK;
}

int genie_index_train_rmi(genie_index *ix, int32_t n_experts, const int32_t *experts, double *mean_abs_err,
                          int32_t *max_abs_err)
{
    if (!ix || !ix->host || ix->has_dev || (n_experts > 0 && !experts)) return GENIE_E_INVALID;
    int rc = train_rmi(*ix->host, n_experts, experts, mean_abs_err, max_abs_err);
    if (rc) return rc;
    fill_header(*ix->host, &ix->hdr);
    return GENIE_OK;
}

int genie_index_rmi_models(const genie_index *ix, double *coef, double *icpt, int32_t *leaf_err)
{
    if (!ix || !ix->host || ix->host->nlev < 1) return GENIE_E_INVALID;
    const HostIndex &h = *ix->host;
    for (size_t i = 0; i < h.rmi.size(); i++) {
        if (coef) coef[i] = h.rmi[i].coef;
        if (icpt) icpt[i] = h.rmi[i].icpt;
    }
    if (leaf_err) {
        if (h.rmi_err.empty()) return GENIE_E_INVALID;
        for (size_t i = 0; i < h.rmi_err.size(); i++) leaf_err[i] = h.rmi_err[i];
    }
    return GENIE_OK;
}

int genie_index_info(const genie_index *ix, genie_info *out)
{
    if (!ix || !out || !ix->has_hdr) return GENIE_E_INVALID;
    out->n = ix->hdr.n;
    out->K = ix->hdr.K;
    out->dir_bits = ix->hdr.P;
    out->lut_keys = ix->hdr.lut_keys;
    out->lut_slots = ix->hdr.lut_slots;
    out->rmi_levels = ix->hdr.nlev;
    out->has_host = ix->host != nullptr;
    out->has_device = ix->has_dev;
    out->device = ix->device;
    out->blob_bytes = ix->hdr.total_bytes;
    return GENIE_OK;
}

const int32_t *genie_index_suffix_array(const genie_index *ix)
{
    return (ix && ix->host) ? ix->host->sa1.data() : nullptr;
}

int genie_index_lut_arrays(const genie_index *ix, const uint32_t **codes, const int32_t **lo, const int32_t **hi)
{
    if (!ix || !ix->host) return GENIE_E_INVALID;
    if (codes) *codes = ix->host->lut_code.data();
    if (lo) *lo = ix->host->lut_lo.data();
    if (hi) *hi = ix->host->lut_hi.data();
    return GENIE_OK;
}

int64_t genie_index_blob_bytes(const genie_index *ix)
{
    return (ix && ix->has_hdr) ? ix->hdr.total_bytes : (int64_t)GENIE_E_INVALID;
}

int genie_index_serialize(const genie_index *ix, void *host_dst, int64_t cap)
{
    if (!ix || !ix->host) return GENIE_E_INVALID;
    return serialize(*ix->host, host_dst, cap);
}

int64_t genie_index_image_bytes(const genie_index *ix, int32_t image_flags)
{
    if (!ix || !ix->host || (image_flags & ~GENIE_IMAGE_NO_SEED_TABLE)) return (int64_t)GENIE_E_INVALID;
    BlobHeader hdr;
    fill_header(*ix->host, &hdr, image_flags);
    return hdr.total_bytes;
}

int genie_index_serialize_image(const genie_index *ix, int32_t image_flags, void *host_dst, int64_t cap)
{
    if (!ix || !ix->host || (image_flags & ~GENIE_IMAGE_NO_SEED_TABLE)) return GENIE_E_INVALID;
    return serialize(*ix->host, host_dst, cap, image_flags);
}

int genie_index_open(const void *host_header, const void *d_blob, int64_t blob_bytes, int32_t device,
                     genie_index **ix_inout)
{
    if (!host_header || !d_blob || !ix_inout) return GENIE_E_INVALID;
    BlobHeader hdr;
    memcpy(&hdr, host_header, sizeof(hdr));
    DevIndex dev;
    int rc = dev_index_from_header(hdr, d_blob, blob_bytes, &dev);
    if (rc) return rc;
    genie_index *ix = *ix_inout;
    if (!ix) {
        ix = new (std::nothrow) genie_index();
        if (!ix) return GENIE_E_NOMEM;
    }
    ix->hdr = hdr;
    ix->has_hdr = true;
    ix->dev = dev;
    ix->has_dev = true;
    ix->device = device;
    ix->blob_bytes = blob_bytes;
    ix->num_cus = query_cus(device);
    *ix_inout = ix;
    return GENIE_OK;
}

int genie_index_validate(const genie_index *ix, uint32_t *what, void *stream)
{
    int rc = ready(ix);
    if (rc) return rc;
    unsigned int w = 0;
    rc = validate_image(ix, &w, stream);
    if (what) *what = w;
    return rc;
}

int genie_index_to_device(genie_index *ix, int32_t device)
{
    if (!ix || !ix->host) return GENIE_E_INVALID;
    int rc = check_hip(hipSetDevice(device), "hipSetDevice");
    if (rc) return rc;
    const int64_t bytes = ix->hdr.total_bytes;
    void *host = malloc((size_t)bytes);
    if (!host) return GENIE_E_NOMEM;
    rc = serialize(*ix->host, host, bytes);
    void *d = nullptr;
    if (!rc) rc = check_hip(hipMalloc(&d, (size_t)bytes), "hipMalloc(index image)");
    if (!rc) rc = check_hip(hipMemcpy(d, host, (size_t)bytes, hipMemcpyHostToDevice), "hipMemcpy(index image)");
    if (!rc) {
        if (ix->owned_blob) (void)hipFree(ix->owned_blob);
        ix->owned_blob = d;
        rc = genie_index_open(host, d, bytes, device, &ix);
    } else if (d) {
        (void)hipFree(d);
    }
    free(host);
    return rc;
}

void genie_index_destroy(genie_index *ix)
{
    if (!ix) return;
    if (ix->owned_blob) (void)hipFree(ix->owned_blob);
    delete ix->host;
    delete ix;
}

static int ready(const genie_index *ix)
{
    if (!ix) return GENIE_E_INVALID;
    if (!ix->has_dev) return GENIE_E_NO_DEVICE;
    return GENIE_OK;
}

int genie_sa_interval(const genie_index *ix, const uint8_t *d_pats, const int32_t *d_lens, int64_t N,
                      int32_t stride, int32_t fixed_len, int32_t *d_out_lohi, void *stream)
{
    int rc = ready(ix);
    if (rc) return rc;
    if (N < 0 || stride < 0 || fixed_len < 0 || (N > 0 && (!d_pats || !d_out_lohi))) return GENIE_E_INVALID;
    if (fixed_len > GENIE_MAX_READ_LEN) return GENIE_E_TOO_LONG;
    return launch_sa_interval(ix, d_pats, d_lens, N, stride, fixed_len, d_out_lohi, stream);
}

int genie_seed_lookup(const genie_index *ix, int32_t mode, const uint8_t *d_kmers, int64_t N,
                      int32_t *d_out_lohi, double *d_pred, void *stream)
{
    int rc = ready(ix);
    if (rc) return rc;
    if (N < 0 || (N > 0 && (!d_kmers || !d_out_lohi))) return GENIE_E_INVALID;
    if (ix->dev.K < 1) return GENIE_E_NO_LUT;
    if (mode == GENIE_MODE_LUT && (ix->dev.flags & kFlagNoSeedTable)) return GENIE_E_NO_LUT;     // a find_smems-only image
    if (mode == GENIE_MODE_RMI && ix->dev.nlev < 1) return GENIE_E_NO_MODEL;
    return launch_seed_lookup(ix, mode, d_kmers, N, d_out_lohi, d_pred, stream);
}

int genie_find_smems(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens,
                     int64_t N, int32_t stride, int32_t fixed_len, int32_t min_len, int32_t *d_counts,
                     int32_t *d_slots, int32_t cap, int32_t *d_status, void *d_workspace, int64_t workspace_bytes,
                     void *stream)
{
    int rc = ready(ix);
    if (rc) return rc;
    if (N < 0 || stride < 0 || fixed_len < 0 || cap < 1 || (N > 0 && (!d_reads || !d_counts || !d_slots)))
        return GENIE_E_INVALID;
    if (mode < GENIE_MODE_BWA || mode > GENIE_MODE_RMI) return GENIE_E_INVALID;
    if (fixed_len > GENIE_MAX_READ_LEN) return GENIE_E_TOO_LONG;
    if (mode != GENIE_MODE_BWA && ix->dev.K < 1) return GENIE_E_NO_LUT;
    if (mode == GENIE_MODE_RMI && ix->dev.nlev < 1) return GENIE_E_NO_MODEL;
    if ((reinterpret_cast<uintptr_t>(d_slots) & 15) != 0) return GENIE_E_INVALID;
    return launch_find_smems(ix, mode, d_reads, d_lens, N, stride, fixed_len, min_len, d_counts, d_slots, cap,
                             d_status, d_workspace, workspace_bytes, stream);
}

int genie_find_smems_csr(const genie_index *ix, int32_t mode, const uint8_t *d_reads, const int32_t *d_lens,
                         int64_t N, int32_t stride, int32_t fixed_len, int32_t min_len, int64_t *d_offsets,
                         int32_t *d_rows, int64_t out_cap_rows, int32_t *d_status, void *d_workspace,
                         int64_t workspace_bytes, void *stream)
{
    int rc = ready(ix);
    if (rc) return rc;
    if (N < 0 || stride < 0 || fixed_len < 0 || out_cap_rows < 0 || !d_offsets || (N > 0 && (!d_reads || !d_rows)))
        return GENIE_E_INVALID;
    if (mode < GENIE_MODE_BWA || mode > GENIE_MODE_RMI) return GENIE_E_INVALID;
    if (fixed_len > GENIE_MAX_READ_LEN) return GENIE_E_TOO_LONG;
    if (mode != GENIE_MODE_BWA && ix->dev.K < 1) return GENIE_E_NO_LUT;
    if (mode == GENIE_MODE_RMI && ix->dev.nlev < 1) return GENIE_E_NO_MODEL;
    if ((reinterpret_cast<uintptr_t>(d_rows) & 15) != 0) return GENIE_E_INVALID;
    return launch_find_smems_csr(ix, mode, d_reads, d_lens, N, stride, fixed_len, min_len, d_offsets, d_rows, out_cap_rows,
                                 d_status, d_workspace, workspace_bytes, stream);
}

static int find_smems_packed_any(const genie_index *ix, int32_t mode, const uint8_t *d_reads2bit, const int32_t *d_lens, int64_t N,
                                 int32_t stride_bytes, int32_t fixed_len, int32_t min_len, uint8_t *d_counts8, uint8_t *d_status8,
                                 void *d_rows, int64_t out_cap_rows, int64_t *d_totals, int64_t *d_escapes, int64_t cap_escapes,
                                 void *d_workspace, int64_t workspace_bytes, void *stream, int row_bytes)
{
    int rc = ready(ix);
    if (rc) return rc;
    if (N < 0 || stride_bytes < 0 || fixed_len < 0 || out_cap_rows < 0 || cap_escapes < 0 || !d_totals ||
        (N > 0 && (!d_reads2bit || !d_rows || !d_counts8 || !d_status8)) || (cap_escapes > 0 && !d_escapes))
        return GENIE_E_INVALID;
    if (mode < GENIE_MODE_BWA || mode > GENIE_MODE_RMI) return GENIE_E_INVALID;
    if (fixed_len > 255) return GENIE_E_TOO_LONG;                       // start / end are bytes in the compact rows
    if (row_bytes == 6 && (int64_t)ix->dev.n + 1 >= (1ll << 24)) return GENIE_E_TOO_LONG;       // lo is 24 bits in the 6-byte rows
    if ((stride_bytes & 3) != 0 || stride_bytes < 4 * ((fixed_len + 15) / 16)) return GENIE_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(d_reads2bit) & 3) != 0 || (reinterpret_cast<uintptr_t>(d_rows) & (row_bytes == 6 ? 1 : 7)) != 0 ||
        (reinterpret_cast<uintptr_t>(d_totals) & 7) != 0)
        return GENIE_E_INVALID;
    if (mode != GENIE_MODE_BWA && ix->dev.K < 1) return GENIE_E_NO_LUT;
    if (mode == GENIE_MODE_RMI && ix->dev.nlev < 1) return GENIE_E_NO_MODEL;
    return launch_find_smems_packed(ix, mode, d_reads2bit, d_lens, N, stride_bytes, fixed_len, min_len, d_counts8, d_status8,
                                    d_rows, out_cap_rows, d_totals, d_escapes, cap_escapes, d_workspace, workspace_bytes, stream, row_bytes);
}

int genie_find_smems_packed(const genie_index *ix, int32_t mode, const uint8_t *d_reads2bit, const int32_t *d_lens, int64_t N,
                            int32_t stride_bytes, int32_t fixed_len, int32_t min_len, uint8_t *d_counts8, uint8_t *d_status8,
                            void *d_rows8, int64_t out_cap_rows, int64_t *d_totals, int64_t *d_escapes, int64_t cap_escapes,
                            void *d_workspace, int64_t workspace_bytes, void *stream)
{
    return find_smems_packed_any(ix, mode, d_reads2bit, d_lens, N, stride_bytes, fixed_len, min_len, d_counts8, d_status8, d_rows8,
                                 out_cap_rows, d_totals, d_escapes, cap_escapes, d_workspace, workspace_bytes, stream, 8);
}

int genie_find_smems_packed6(const genie_index *ix, int32_t mode, const uint8_t *d_reads2bit, const int32_t *d_lens, int64_t N,
                             int32_t stride_bytes, int32_t fixed_len, int32_t min_len, uint8_t *d_counts8, uint8_t *d_status8,
                             void *d_rows6, int64_t out_cap_rows, int64_t *d_totals, int64_t *d_escapes, int64_t cap_escapes,
                             void *d_workspace, int64_t workspace_bytes, void *stream)
{
    return find_smems_packed_any(ix, mode, d_reads2bit, d_lens, N, stride_bytes, fixed_len, min_len, d_counts8, d_status8, d_rows6,
                                 out_cap_rows, d_totals, d_escapes, cap_escapes, d_workspace, workspace_bytes, stream, 6);
}

int64_t genie_find_smems_workspace_bytes(int64_t N, int32_t max_len)
{
    if (N < 0 || max_len < 0) return (int64_t)GENIE_E_INVALID;
    return find_smems_workspace_bytes(N, max_len);
}

int genie_find_smems_workspace_rows(int32_t max_len, int32_t *row_bytes4)
{
    if (max_len < 0 || max_len > GENIE_MAX_READ_LEN || !row_bytes4) return GENIE_E_INVALID;
    find_smems_workspace_rows(max_len, row_bytes4);
    return GENIE_OK;
}

int64_t genie_compact_tmp_bytes(int64_t N) { return N < 0 ? (int64_t)GENIE_E_INVALID : compact_tmp_bytes(N); }

int genie_compact_smems(const int32_t *d_counts, const int32_t *d_slots, int64_t N, int32_t cap,
                        int64_t *d_offsets, int32_t *d_out, int64_t out_cap_rows, void *d_tmp, void *stream)
{
    if (N < 0 || cap < 1 || !d_offsets || !d_tmp || (N > 0 && (!d_counts || !d_slots))) return GENIE_E_INVALID;
    return launch_compact(d_counts, d_slots, N, cap, d_offsets, d_out, out_cap_rows, d_tmp, stream);
}

int64_t genie_locate_tmp_bytes(int64_t S) { return S < 0 ? 0 : locate_tmp_bytes(S); }

int genie_locate(const genie_index *ix, const int32_t *d_lohi, int32_t stride, int64_t S, int64_t *d_pos_offsets,
                 int32_t *d_positions, int64_t cap_positions, void *d_tmp, int64_t tmp_bytes, void *stream)
{
    int rc = ready(ix);
    if (rc) return rc;
    if (S < 0 || stride < 2 || !d_pos_offsets || cap_positions < 0 || (S > 0 && !d_lohi) ||
        (cap_positions > 0 && !d_positions))
        return GENIE_E_INVALID;
    return launch_locate(ix, d_lohi, stride, S, d_pos_offsets, d_positions, cap_positions, d_tmp, tmp_bytes, stream);
}

int genie_launch_info(const genie_index *ix, int32_t mode, int32_t max_len, int32_t *grid, int32_t *block,
                      int32_t *lds_bytes)
{
    int rc = ready(ix);
    if (rc) return rc;
    return find_smems_geometry(ix, mode, max_len, grid, block, lds_bytes);
}

int genie_search_kernel_name(const genie_index *ix, int32_t mode, int32_t max_len, char *buf, int32_t cap)
{
    int rc = ready(ix);
    if (rc) return rc;
    if (mode < GENIE_MODE_BWA || mode > GENIE_MODE_RMI) return GENIE_E_INVALID;
    return search_kernel_name(ix, mode, max_len, buf, cap);
}

int genie_index_set_option(genie_index *ix, int32_t option, int32_t value)
{
    if (!ix) return GENIE_E_INVALID;
    switch (option) {
    case GENIE_OPT_SEARCH_ALL: ix->opt_search_all = value != 0; return GENIE_OK;
    case GENIE_OPT_GROUP_POSITIONS: ix->opt_group_positions = value > 0 ? value : 0; return GENIE_OK;
    case GENIE_OPT_SEARCH_ONLY: ix->opt_search_only = value != 0; if (!value) ix->opt_debug = 0; return GENIE_OK;
    case GENIE_OPT_SCHEDULING: ix->opt_scheduling = value & 15; return GENIE_OK;
    case GENIE_OPT_SEARCH_STAGES_OFF: ix->opt_debug = ix->opt_search_only ? (value & 63) : 0; return GENIE_OK;
    case GENIE_OPT_SEARCH_BLOCKS_PER_CU: ix->opt_search_blocks_per_cu = value > 0 ? value : 0; return GENIE_OK;
    default: return GENIE_E_INVALID;
    }
}

int genie_index_set_stage_events(genie_index *ix, void *ev_search_begin, void *ev_search_end)
{
    if (!ix) return GENIE_E_INVALID;
    ix->ev_search_begin = ev_search_begin;
    ix->ev_search_end = ev_search_end;
    return GENIE_OK;
}

const char *genie_strerror(int status)
{
    switch (status) {
    case GENIE_OK: return "ok";
    case GENIE_E_INVALID: return "invalid argument";
    case GENIE_E_ALPHABET: return "base code outside 0..3";
    case GENIE_E_NOMEM: return "out of host memory";
    case GENIE_E_NO_DEVICE: return "index has no device image";
    case GENIE_E_HIP: return "HIP runtime error";
    case GENIE_E_TOO_LONG: return "read longer than the kernel supports";
    case GENIE_E_NO_MODEL: return "no RMI model installed";
    case GENIE_E_BAD_BLOB: return "not a serialized GENIE index";
    case GENIE_E_NO_LUT: return "no K-mer table (index built with K = 0, or an image serialized without its seed table)";
    case GENIE_E_CAPACITY: return "output capacity too small";
    case GENIE_W_SEARCH_ONLY: return "GENIE_OPT_SEARCH_ONLY is set: no outputs were produced";
    default: return "unknown status";
    }
}

const char *genie_last_hip_error(void) { return last_hip_error(); }

}  // extern "C"

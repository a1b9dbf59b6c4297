// svdq_api.hip -- extern "C" entry points of libsvdq_hip.so (declared in include/svdq.h).
// Host-side only: builds the unit / parameter tables of a plan and enqueues the kernels.

#include "svdq_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void svdq_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            svdq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SVDQ_EHIP;                                                           \
        }                                                                               \
    } while (0)

extern "C" int svdq_abi_version(void) { return SVDQ_ABI_VERSION; }
extern "C" const char *svdq_last_error(void) { return g_err; }

static int32_t auto_unit_rows(int64_t D, int32_t n_tasks) {
    // Measured on MI355X (bench.py --unit-rows sweeps, tools/shard_one.py): for a whole ViT-L-14 / ViT-B-32 model
    // 4096..8192-row units are equal within noise (smaller ones multiply the fp64 partial slots the two small kernels
    // must reduce), but one rank's share of the 8-GPU run has only ~4 700 units of 8192 rows for 5 120 resident waves --
    // a single partial wave, 0.765 ms against 0.699 ms with 4096-row units.  The unit size is a function of the
    // tensor alone (never of what else is in the plan), so a tensor's artifacts are the same bits in any batch and on
    // any number of GPUs.  Small tensors get >= 4 units when they have the rows for it, never less than 4 blocks each.
    // N > 16: a unit's partial matrices are N x N doubles (3.2 KB at N = 20) and the two reduction kernels read all of
    // them -- 237 MB per launch for ViT-L-14 x 20 with 4096-row units, bandwidth-bound; 8192-row units halve that
    // (0.39 -> 0.31 ms for the four small launches, the streaming passes unchanged).  Still a function of (tensor, N) alone.
    int64_t ur = n_tasks > 16 ? 8192 : 4096;
    if (D < 4 * ur) ur = svdq_align_up((D + 3) / 4, SVDQ_BLK_ROWS);
    if (ur < 4 * SVDQ_BLK_ROWS) ur = 4 * SVDQ_BLK_ROWS;
    return (int32_t)ur;
}

extern "C" int svdq_plan_create(svdq_plan **out, int32_t n_tasks, int32_t n_params, const int64_t *rows,
                                const svdq_config *cfg) {
    if (!out || !rows || !cfg) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    if (n_tasks < 1 || n_tasks > SVDQ_MAX_TASKS) {
        svdq_set_error("n_tasks must be in [1, %d], got %d", SVDQ_MAX_TASKS, n_tasks);
        return SVDQ_EINVAL;
    }
    if (n_params < 1) {
        svdq_set_error("Empty delta list");  // basis.py:285
        return SVDQ_EINVAL;
    }
    // SVDHybridConfig.__post_init__ (config.py:207-234)
    if (!(cfg->energy_threshold > 0.f && cfg->energy_threshold <= 1.f)) {
        svdq_set_error("Energy threshold must be in (0, 1], got %g", cfg->energy_threshold);
        return SVDQ_EINVAL;
    }
    if (cfg->low_bits < 1 || cfg->low_bits > 8) {
        svdq_set_error("Low bits must be in [1, 8], got %d", cfg->low_bits);
        return SVDQ_EINVAL;
    }
    if (cfg->rtvq_stages < 1 || cfg->rtvq_stages > SVDQ_MAX_STAGES) {
        svdq_set_error("RTVQ stages must be in [1, %d], got %d", SVDQ_MAX_STAGES, cfg->rtvq_stages);
        return SVDQ_EINVAL;
    }
    if (cfg->unit_rows < 0 || cfg->unit_rows % SVDQ_BLK_ROWS != 0) {
        svdq_set_error("unit_rows must be 0 or a multiple of %d", SVDQ_BLK_ROWS);
        return SVDQ_EINVAL;
    }
    for (int p = 0; p < n_params; ++p)
        if (rows[p] < 1) {
            svdq_set_error("parameter %d has %lld rows", p, (long long)rows[p]);
            return SVDQ_EINVAL;
        }

    svdq_plan *pl = (svdq_plan *)calloc(1, sizeof(svdq_plan));
    pl->n_tasks = n_tasks;
    pl->n_params = n_params;
    pl->cfg = *cfg;
    pl->ntp = svdq_ntp(n_tasks);
    pl->pack = pl->ntp <= 8 ? 2 : 1;
    const int es = cfg->fp16 ? 2 : 4;
    const int64_t N = n_tasks;

    pl->h_params = (SvdqParam *)calloc(n_params, sizeof(SvdqParam));
    int64_t n_units = 0, basis_bytes = 0, mean_floats = 0;
    for (int p = 0; p < n_params; ++p) {
        const int64_t D = rows[p];
        const int32_t ur = cfg->unit_rows > 0 ? cfg->unit_rows : auto_unit_rows(D, n_tasks);
        const int64_t cnt = (D + ur - 1) / ur;
        SvdqParam &pd = pl->h_params[p];
        pd.rows = D;
        pd.slab_off = basis_bytes;
        pd.mean_off = mean_floats;
        pd.unit_begin = (int32_t)n_units;
        pd.unit_count = (int32_t)cnt;
        n_units += cnt;
        const int64_t r = D < N ? D : N;
        basis_bytes += svdq_align_up(D * r * es, 256) + 256;  // + room for the aligned U_low start
        mean_floats += svdq_align_up(D, 64);
    }
    if (n_units > 0x7fffffff / 4) {
        svdq_set_error("too many work units (%lld)", (long long)n_units);
        free(pl->h_params);
        free(pl);
        return SVDQ_EINVAL;
    }
    pl->n_units = (int32_t)n_units;
    pl->n_slots = pl->n_units * pl->pack;
    pl->h_units = (SvdqUnit *)calloc(n_units, sizeof(SvdqUnit));
    for (int p = 0; p < n_params; ++p) {
        const int64_t D = rows[p];
        const int32_t ur = cfg->unit_rows > 0 ? cfg->unit_rows : auto_unit_rows(D, n_tasks);
        const SvdqParam &pd = pl->h_params[p];
        for (int32_t i = 0; i < pd.unit_count; ++i) {
            SvdqUnit &u = pl->h_units[pd.unit_begin + i];
            u.param = p;
            u.row0 = (int64_t)i * ur;
            u.nrows = (int32_t)((D - u.row0) < ur ? (D - u.row0) : ur);
        }
    }

    // workspace
    const int64_t nn = N * N;
    int64_t off = 0;
    pl->ws_gram_off = off;
    off += svdq_align_up((int64_t)pl->n_slots * nn * 8, 256);
    pl->ws_cpart_off = off;
    off += svdq_align_up((int64_t)pl->n_slots * nn * 8, 256);
    pl->ws_w_off = off;
    off += svdq_align_up((int64_t)n_params * (nn + 4) * 4, 256);  // W [N][N] + {spike, null column, -, -}
    pl->ws_c0_off = off;
    off += svdq_align_up((int64_t)n_params * nn * 8, 256);        // closed-form coefficients of the unrounded basis
    pl->ws_gram2_off = off;
    off += svdq_align_up((int64_t)n_params * SVDQ_RC * nn * 8, 256);  // level-2 partials (k_reduce)
    pl->ws_cpart2_off = off;
    off += svdq_align_up((int64_t)n_params * SVDQ_RC * nn * 8, 256);
    pl->ws_flag_off = off;
    off += svdq_align_up((int64_t)n_params * 4, 256);                 // N > 16: parameters that need the fp64 Gram pass
    pl->sizes.workspace_bytes = off;
    pl->sizes.basis_bytes = basis_bytes;
    pl->sizes.mean_floats = mean_floats;
    pl->sizes.n_units = pl->n_units;
    pl->sizes.n_slots = pl->n_slots;

    // small artifacts
    const int64_t P = n_params, S = cfg->rtvq_stages;
    svdq_small_layout &L = pl->small;
    off = 0;
    L.sigma_off = off;  off += svdq_align_up(P * N * 4, 64);
    L.k_off = off;      off += svdq_align_up(P * 4, 64);
    L.r_off = off;      off += svdq_align_up(P * 4, 64);
    L.energy_off = off; off += svdq_align_up(P * 4, 64);
    L.rows_off = off;   off += svdq_align_up(P * 8, 64);
    L.chigh_off = off;  off += svdq_align_up(P * nn * 2, 64);
    L.codes_off = off;  off += svdq_align_up(P * N * S * N, 64);
    L.scale_off = off;  off += svdq_align_up(P * N * S * 4, 64);
    L.zp_off = off;     off += svdq_align_up(P * N * S * 4, 64);
    L.rnorm_off = off;  off += svdq_align_up(P * N * S * 4, 64);
    L.coef_off = off;   off += svdq_align_up(P * nn * 4, 64);
    L.status_off = off; off += 64;
    L.total_bytes = off;
    pl->sizes.small_bytes = off;

    hipError_t e = hipMalloc((void **)&pl->d_params, sizeof(SvdqParam) * n_params);
    if (e == hipSuccess) e = hipMalloc((void **)&pl->d_units, sizeof(SvdqUnit) * n_units);
    if (e == hipSuccess)
        e = hipMemcpy(pl->d_params, pl->h_params, sizeof(SvdqParam) * n_params, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d_units, pl->h_units, sizeof(SvdqUnit) * n_units, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        svdq_set_error("plan table upload failed: %s", hipGetErrorString(e));
        svdq_plan_destroy(pl);
        return SVDQ_EHIP;
    }
    *out = pl;
    return SVDQ_OK;
}

extern "C" void svdq_plan_destroy(svdq_plan *pl) {
    if (!pl) return;
    if (pl->d_params) (void)hipFree(pl->d_params);
    if (pl->d_units) (void)hipFree(pl->d_units);
    if (pl->d_bits) (void)hipFree(pl->d_bits);
    free(pl->h_params);
    free(pl->h_units);
    free(pl);
}

extern "C" int svdq_plan_set_low_bits(svdq_plan *pl, const int32_t *bits) {
    if (!pl) return SVDQ_EINVAL;
    if (!bits) {   // back to one width for the whole plan
        if (pl->d_bits) (void)hipFree(pl->d_bits);
        pl->d_bits = nullptr;
        return SVDQ_OK;
    }
    for (int p = 0; p < pl->n_params; ++p)
        if (bits[p] < 1 || bits[p] > 8) {
            svdq_set_error("Low bits must be in [1, 8], got %d (parameter %d)", bits[p], p);
            return SVDQ_EINVAL;
        }
    if (!pl->d_bits) HIP_TRY(hipMalloc((void **)&pl->d_bits, sizeof(int32_t) * pl->n_params));
    HIP_TRY(hipMemcpy(pl->d_bits, bits, sizeof(int32_t) * pl->n_params, hipMemcpyHostToDevice));
    return SVDQ_OK;
}

extern "C" int svdq_plan_sizes(const svdq_plan *pl, svdq_sizes *out) {
    if (!pl || !out) return SVDQ_EINVAL;
    *out = pl->sizes;
    return SVDQ_OK;
}

extern "C" int svdq_plan_small_layout(const svdq_plan *pl, svdq_small_layout *out) {
    if (!pl || !out) return SVDQ_EINVAL;
    *out = pl->small;
    return SVDQ_OK;
}

extern "C" int svdq_plan_basis_layout(const svdq_plan *pl, int64_t *slab_off, int64_t *mean_off) {
    if (!pl) return SVDQ_EINVAL;
    for (int p = 0; p < pl->n_params; ++p) {
        if (slab_off) slab_off[p] = pl->h_params[p].slab_off;
        if (mean_off) mean_off[p] = pl->h_params[p].mean_off;
    }
    return SVDQ_OK;
}

static inline uint8_t *ws(void *base, int64_t off) { return reinterpret_cast<uint8_t *>(base) + off; }

static int check_range(const svdq_plan *pl, int32_t param0, int32_t nparams) {
    if (param0 < 0 || nparams < 1 || param0 + nparams > pl->n_params) {
        svdq_set_error("parameter range [%d, %d) outside [0, %d)", param0, param0 + nparams, pl->n_params);
        return SVDQ_EINVAL;
    }
    return SVDQ_OK;
}

static void unit_range(const svdq_plan *pl, int32_t param0, int32_t nparams, int *u0, int *nu) {
    const SvdqParam &a = pl->h_params[param0], &b = pl->h_params[param0 + nparams - 1];
    *u0 = a.unit_begin;
    *nu = b.unit_begin + b.unit_count - a.unit_begin;
}

static int gram_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace, int32_t param0,
                      int32_t nparams, const void *idx, void *stream, const void *base = nullptr,
                      const int64_t *ustart = nullptr) {
    if (!pl || !ptrs || !workspace) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    if (int rc = check_range(pl, param0, nparams)) return rc;
    int u0, nu;
    unit_range(pl, param0, nparams, &u0, &nu);
    // N <= 16: exact products on the fp64 MFMA (pass 1 stays HBM-bound); N > 16: fp32 products first, the fp64 pass only
    // for the parameters the eigen-stage flags (eig_range).  cfg.reserved bit 1 keeps fp32 products throughout (A/B).
    const int f64 = (pl->ntp <= 16 && !(pl->cfg.reserved & 2)) ? 1 : 0;
    return svdq_launch_gram(pl, ptrs, rows_dev, reinterpret_cast<double *>(ws(workspace, pl->ws_gram_off)), u0, nu,
                            pl->cfg.center, idx, base, f64, nullptr, (hipStream_t)stream, ustart);
}

extern "C" int svdq_gram_center_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace,
                                      int32_t param0, int32_t nparams, void *stream) {
    return gram_range(pl, ptrs, rows_dev, workspace, param0, nparams, nullptr, stream);
}

extern "C" int svdq_task_gram(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace,
                              double *out_gram, void *stream) {
    if (!pl || !ptrs || !workspace || !out_gram) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    double *part = reinterpret_cast<double *>(ws(workspace, pl->ws_gram_off));
    double *part2 = reinterpret_cast<double *>(ws(workspace, pl->ws_gram2_off));
    if (int rc = svdq_launch_gram(pl, ptrs, rows_dev, part, 0, pl->n_units, /*center=*/0, nullptr, nullptr,
                                  pl->ntp <= 16, nullptr, st))
        return rc;
    if (int rc = svdq_launch_reduce(pl, part, part2, 0, pl->n_params, nullptr, st)) return rc;
    return svdq_launch_gram_total(pl, part2, out_gram, st);
}

static int eig_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace, void *small,
                     int32_t param0, int32_t nparams, const void *idx, void *stream, const void *base = nullptr,
                     const int64_t *ustart = nullptr) {
    if (!pl || !ptrs || !workspace || !small) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    if (int rc = check_range(pl, param0, nparams)) return rc;
    hipStream_t st = (hipStream_t)stream;
    double *part = reinterpret_cast<double *>(ws(workspace, pl->ws_gram_off));
    double *part2 = reinterpret_cast<double *>(ws(workspace, pl->ws_gram2_off));
    float *W = reinterpret_cast<float *>(ws(workspace, pl->ws_w_off));
    double *c0 = reinterpret_cast<double *>(ws(workspace, pl->ws_c0_off));
    uint8_t *sm = reinterpret_cast<uint8_t *>(small);
    const bool refine = pl->ntp > 16 && !(pl->cfg.reserved & 2);
    int32_t *flags = refine ? reinterpret_cast<int32_t *>(ws(workspace, pl->ws_flag_off)) : nullptr;
    if (int rc = svdq_launch_reduce(pl, part, part2, param0, nparams, nullptr, st)) return rc;
    if (int rc = svdq_launch_eig(pl, ptrs, rows_dev, part2, W, c0, sm, param0, nparams, idx, base, nullptr, flags, st,
                                 ustart))
        return rc;
    if (!refine) return SVDQ_OK;
    // N > 16: the parameters whose spectrum reaches into the band fp32-product sums do not resolve are accumulated
    // again with exact products (v_mfma_f64_16x16x4_f64) and solved again; units of all other parameters return at
    // once, so a batch without such a parameter pays three near-empty launches
    int u0, nu;
    unit_range(pl, param0, nparams, &u0, &nu);
    if (int rc = svdq_launch_gram(pl, ptrs, rows_dev, part, u0, nu, pl->cfg.center, idx, base, 1, flags, st, ustart))
        return rc;
    if (int rc = svdq_launch_reduce(pl, part, part2, param0, nparams, flags, st)) return rc;
    return svdq_launch_eig(pl, ptrs, rows_dev, part2, W, c0, sm, param0, nparams, idx, base, flags, nullptr, st, ustart);
}

extern "C" int svdq_eig_rank_select_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev,
                                          void *workspace, void *small, int32_t param0, int32_t nparams,
                                          void *stream) {
    return eig_range(pl, ptrs, rows_dev, workspace, small, param0, nparams, nullptr, stream);
}

static int bp_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace, const void *small,
                    void *basis, float *mean, int32_t param0, int32_t nparams, const void *idx, void *stream,
                    const void *base = nullptr, const int64_t *ustart = nullptr) {
    if (!pl || !ptrs || !workspace || !small || !basis) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(basis) & 255) != 0) {
        svdq_set_error("basis buffer must be 256-byte aligned");
        return SVDQ_EINVAL;
    }
    if (int rc = check_range(pl, param0, nparams)) return rc;
    int u0, nu;
    unit_range(pl, param0, nparams, &u0, &nu);
    const uint8_t *sm = reinterpret_cast<const uint8_t *>(small);
    return svdq_launch_basis_project(pl, ptrs, rows_dev, reinterpret_cast<const float *>(ws(workspace, pl->ws_w_off)),
                                     reinterpret_cast<const int32_t *>(sm + pl->small.k_off),
                                     reinterpret_cast<const int32_t *>(sm + pl->small.r_off),
                                     reinterpret_cast<uint8_t *>(basis), mean,
                                     reinterpret_cast<double *>(ws(workspace, pl->ws_cpart_off)), u0, nu,
                                     pl->cfg.reserved & 5, idx, base, (hipStream_t)stream, ustart);
}

extern "C" int svdq_basis_project_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev,
                                        void *workspace, const void *small, void *basis, float *mean, int32_t param0,
                                        int32_t nparams, void *stream) {
    return bp_range(pl, ptrs, rows_dev, workspace, small, basis, mean, param0, nparams, nullptr, stream);
}

extern "C" int svdq_coeff_quantize_range(const svdq_plan *pl, void *workspace, void *small, int32_t param0,
                                         int32_t nparams, void *stream) {
    if (!pl || !workspace || !small) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    if (int rc = check_range(pl, param0, nparams)) return rc;
    if (int rc = svdq_launch_reduce(pl, reinterpret_cast<const double *>(ws(workspace, pl->ws_cpart_off)),
                                    reinterpret_cast<double *>(ws(workspace, pl->ws_cpart2_off)), param0, nparams,
                                    nullptr, (hipStream_t)stream))
        return rc;
    return svdq_launch_coeff(pl, reinterpret_cast<const double *>(ws(workspace, pl->ws_cpart2_off)),
                             reinterpret_cast<const double *>(ws(workspace, pl->ws_c0_off)),
                             reinterpret_cast<uint8_t *>(small), param0, nparams, (hipStream_t)stream);
}

extern "C" int svdq_gram_center(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace,
                                void *stream) {
    if (!pl) return SVDQ_EINVAL;
    return svdq_gram_center_range(pl, ptrs, rows_dev, workspace, 0, pl->n_params, stream);
}

extern "C" int svdq_eig_rank_select(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace,
                                    void *small, void *stream) {
    if (!pl) return SVDQ_EINVAL;
    return svdq_eig_rank_select_range(pl, ptrs, rows_dev, workspace, small, 0, pl->n_params, stream);
}

extern "C" int svdq_basis_project(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace,
                                  const void *small, void *basis, float *mean, void *stream) {
    if (!pl) return SVDQ_EINVAL;
    return svdq_basis_project_range(pl, ptrs, rows_dev, workspace, small, basis, mean, 0, pl->n_params, stream);
}

extern "C" int svdq_coeff_quantize(const svdq_plan *pl, void *workspace, void *small, void *stream) {
    if (!pl) return SVDQ_EINVAL;
    return svdq_coeff_quantize_range(pl, workspace, small, 0, pl->n_params, stream);
}

extern "C" int svdq_compress(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace,
                             void *small, void *basis, float *mean, void *stream) {
    if (pl && small)
        HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(small) + pl->small.status_off, 0, sizeof(int32_t),
                               (hipStream_t)stream));
    int rc = svdq_gram_center(pl, ptrs, rows_dev, workspace, stream);
    if (rc == SVDQ_OK) rc = svdq_eig_rank_select(pl, ptrs, rows_dev, workspace, small, stream);
    if (rc == SVDQ_OK) rc = svdq_basis_project(pl, ptrs, rows_dev, workspace, small, basis, mean, stream);
    if (rc == SVDQ_OK) rc = svdq_coeff_quantize(pl, workspace, small, stream);
    return rc;
}

// Masked parameters without a compaction pass: delta_ptrs name the ORIGINAL (full-size) tensors,
// index_ptrs[p] the ascending source positions of parameter p's selected elements (svdq_maskset_indices),
// rows_dev[p] how many there are.  Outputs (basis rows, mean, coefficients) are those of the compacted
// tensors, bit for bit.
extern "C" int svdq_compress_gather(const svdq_plan *pl, const void *ptrs, const void *index_ptrs,
                                    const int64_t *rows_dev, void *workspace, void *small, void *basis, float *mean,
                                    void *stream) {
    if (!pl || !index_ptrs || !rows_dev) {
        svdq_set_error("svdq_compress_gather: plan, index_ptrs and rows_dev are required");
        return SVDQ_EINVAL;
    }
    if (small)
        HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(small) + pl->small.status_off, 0, sizeof(int32_t),
                               (hipStream_t)stream));
    int rc = gram_range(pl, ptrs, rows_dev, workspace, 0, pl->n_params, index_ptrs, stream);
    if (rc == SVDQ_OK) rc = eig_range(pl, ptrs, rows_dev, workspace, small, 0, pl->n_params, index_ptrs, stream);
    if (rc == SVDQ_OK)
        rc = bp_range(pl, ptrs, rows_dev, workspace, small, basis, mean, 0, pl->n_params, index_ptrs, stream);
    if (rc == SVDQ_OK) rc = svdq_coeff_quantize(pl, workspace, small, stream);
    return rc;
}

// Step 0 folded into the path: delta_ptrs of svdq_compress are replaced by the FINE-TUNED tensors and one base
// tensor per parameter; finetuned - base is formed in registers in both streaming passes (and for row 0 in the
// eigen-stage), so the task vectors are never materialised: 2 x 4(N+1) + 2N + 4 bytes per row instead of
// 4(2N+1) for the ingest plus 10N + 4 for the compression.
extern "C" int svdq_compress_from_base(const svdq_plan *pl, const void *finetuned_ptrs, const void *base_ptrs,
                                       const int64_t *rows_dev, void *workspace, void *small, void *basis, float *mean,
                                       void *stream) {
    if (!pl || !base_ptrs) {
        svdq_set_error("svdq_compress_from_base: plan and base_ptrs are required");
        return SVDQ_EINVAL;
    }
    if (small)
        HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(small) + pl->small.status_off, 0, sizeof(int32_t),
                               (hipStream_t)stream));
    int rc = gram_range(pl, finetuned_ptrs, rows_dev, workspace, 0, pl->n_params, nullptr, stream, base_ptrs);
    if (rc == SVDQ_OK)
        rc = eig_range(pl, finetuned_ptrs, rows_dev, workspace, small, 0, pl->n_params, nullptr, stream, base_ptrs);
    if (rc == SVDQ_OK)
        rc = bp_range(pl, finetuned_ptrs, rows_dev, workspace, small, basis, mean, 0, pl->n_params, nullptr, stream,
                      base_ptrs);
    if (rc == SVDQ_OK) rc = svdq_coeff_quantize(pl, workspace, small, stream);
    return rc;
}

// Both at once: masked parameters straight from checkpoints.  finetuned_ptrs name the ORIGINAL (full-size) fine-tuned
// tensors, base_ptrs the base tensors, index_ptrs the selected positions; finetuned[idx] - base[idx] is formed in
// registers in both passes.  Bit-identical to svdq_ingest followed by svdq_compress_gather.
extern "C" int svdq_compress_gather_from_base(const svdq_plan *pl, const void *finetuned_ptrs, const void *base_ptrs,
                                              const void *index_ptrs, const int64_t *rows_dev, void *workspace,
                                              void *small, void *basis, float *mean, void *stream) {
    if (!pl || !base_ptrs || !index_ptrs || !rows_dev) {
        svdq_set_error("svdq_compress_gather_from_base: plan, base_ptrs, index_ptrs and rows_dev are required");
        return SVDQ_EINVAL;
    }
    if (small)
        HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(small) + pl->small.status_off, 0, sizeof(int32_t),
                               (hipStream_t)stream));
    int rc = gram_range(pl, finetuned_ptrs, rows_dev, workspace, 0, pl->n_params, index_ptrs, stream, base_ptrs);
    if (rc == SVDQ_OK)
        rc = eig_range(pl, finetuned_ptrs, rows_dev, workspace, small, 0, pl->n_params, index_ptrs, stream, base_ptrs);
    if (rc == SVDQ_OK)
        rc = bp_range(pl, finetuned_ptrs, rows_dev, workspace, small, basis, mean, 0, pl->n_params, index_ptrs, stream,
                      base_ptrs);
    if (rc == SVDQ_OK) rc = svdq_coeff_quantize(pl, workspace, small, stream);
    return rc;
}

// Masked parameters WITHOUT index lists (the default for dense masks; reference cli.py:324-341 +
// mask_loader.py:651-709 applied inside the two passes): delta_ptrs name the ORIGINAL (full-size) tensors, mask_ptrs[p]
// the combined mask of parameter p as bool bytes, unit_start the source position of every work unit's first row
// (svdq_maskset_unit_starts / _combine_starts; its bit 62 selects the cleared elements -- the noise region) and
// rows_dev[p] how many rows parameter p has.  The passes walk the source rows, read the mask beside them and compact
// the selected rows into the LDS strip on the fly: 4 N + 1 bytes per source row and pass, no index lists, no compacted
// copies.  Outputs are those of svdq_compress on the compacted tensors, bit for bit.  Above 16 tasks both passes take
// their one-wave kernels (svdq_project_walk.hip): slower per byte than the two-wave kernels the index-list route runs.
extern "C" int svdq_compress_masked(const svdq_plan *pl, const void *ptrs, const void *mask_ptrs,
                                    const int64_t *unit_start, const int64_t *rows_dev, void *workspace, void *small,
                                    void *basis, float *mean, void *stream) {
    if (!pl || !mask_ptrs || !unit_start || !rows_dev) {
        svdq_set_error("svdq_compress_masked: plan, mask_ptrs, unit_start and rows_dev are required");
        return SVDQ_EINVAL;
    }
    if (small)
        HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(small) + pl->small.status_off, 0, sizeof(int32_t),
                               (hipStream_t)stream));
    int rc = gram_range(pl, ptrs, rows_dev, workspace, 0, pl->n_params, mask_ptrs, stream, nullptr, unit_start);
    if (rc == SVDQ_OK)
        rc = eig_range(pl, ptrs, rows_dev, workspace, small, 0, pl->n_params, mask_ptrs, stream, nullptr, unit_start);
    if (rc == SVDQ_OK)
        rc = bp_range(pl, ptrs, rows_dev, workspace, small, basis, mean, 0, pl->n_params, mask_ptrs, stream, nullptr,
                      unit_start);
    if (rc == SVDQ_OK) rc = svdq_coeff_quantize(pl, workspace, small, stream);
    return rc;
}

// The same straight from checkpoints: finetuned[row] - base[row] is formed in registers in both passes.
// Bit-identical to svdq_ingest followed by svdq_compress_masked.
extern "C" int svdq_compress_masked_from_base(const svdq_plan *pl, const void *finetuned_ptrs, const void *base_ptrs,
                                              const void *mask_ptrs, const int64_t *unit_start, const int64_t *rows_dev,
                                              void *workspace, void *small, void *basis, float *mean, void *stream) {
    if (!pl || !base_ptrs || !mask_ptrs || !unit_start || !rows_dev) {
        svdq_set_error("svdq_compress_masked_from_base: plan, base_ptrs, mask_ptrs, unit_start and rows_dev are required");
        return SVDQ_EINVAL;
    }
    if (pl->ntp > 16) {
        svdq_set_error("the mask-walk mode covers N <= 16 tasks (got %d): use the index lists "
                       "(svdq_compress_gather_from_base)", pl->n_tasks);
        return SVDQ_EUNSUPPORTED;
    }
    if (small)
        HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(small) + pl->small.status_off, 0, sizeof(int32_t),
                               (hipStream_t)stream));
    int rc = gram_range(pl, finetuned_ptrs, rows_dev, workspace, 0, pl->n_params, mask_ptrs, stream, base_ptrs, unit_start);
    if (rc == SVDQ_OK)
        rc = eig_range(pl, finetuned_ptrs, rows_dev, workspace, small, 0, pl->n_params, mask_ptrs, stream, base_ptrs,
                       unit_start);
    if (rc == SVDQ_OK)
        rc = bp_range(pl, finetuned_ptrs, rows_dev, workspace, small, basis, mean, 0, pl->n_params, mask_ptrs, stream,
                      base_ptrs, unit_start);
    if (rc == SVDQ_OK) rc = svdq_coeff_quantize(pl, workspace, small, stream);
    return rc;
}

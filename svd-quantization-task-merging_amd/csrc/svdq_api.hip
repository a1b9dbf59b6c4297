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

static int32_t auto_unit_rows(int64_t D) {
    // Measured on MI355X, ViT-L-14 x 8 (bench.py --unit-rows sweep): 4096..8192-row units are best --
    // smaller units multiply the fp64 partial slots the two small kernels must reduce, larger ones
    // leave a long tail on the 256 CUs x ~12 resident waves.  Small tensors get >= 4 units when they
    // have the rows for it, never less than 4 blocks per unit.
    int64_t ur = 8192;
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
        const int32_t ur = cfg->unit_rows > 0 ? cfg->unit_rows : auto_unit_rows(D);
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
        const int32_t ur = cfg->unit_rows > 0 ? cfg->unit_rows : auto_unit_rows(D);
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

    // fused persistent schedule: the item queue.  Gram items in parameter order; the projection items of a
    // parameter become eligible once `lag` bytes of Gram work have been queued after the parameter's last Gram
    // item (time for its eigen-stage), and are then interleaved one (two when many wait) per Gram item.
    pl->fused = (cfg->reserved >> 2) & 1;
    if (pl->fused) {
        int64_t lag = (int64_t)((cfg->reserved >> 8) & 0xffff) * 1000000;
        if (lag <= 0) lag = 192 * 1000000LL;
        pl->n_items = 2 * pl->n_units;
        int32_t *items = (int32_t *)malloc(sizeof(int32_t) * (size_t)pl->n_items);
        int32_t *bq = (int32_t *)malloc(sizeof(int32_t) * (size_t)pl->n_units);   // FIFO of eligible projection units
        int64_t *done_pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_params);
        int64_t pos = 0;
        int n = 0, bq_head = 0, bq_tail = 0, next_pending = 0;  // parameters [next_pending, p) wait for their lag
        for (int p = 0; p < n_params; ++p) {
            const SvdqParam &pd = pl->h_params[p];
            for (int32_t i = 0; i < pd.unit_count; ++i) {
                const int32_t u = pd.unit_begin + i;
                items[n++] = u;
                pos += (int64_t)pl->h_units[u].nrows * N * 4;
                while (next_pending < p && pos - done_pos[next_pending] >= lag) {
                    const SvdqParam &q = pl->h_params[next_pending++];
                    for (int32_t j = 0; j < q.unit_count; ++j) bq[bq_tail++] = q.unit_begin + j;
                }
                if (bq_head < bq_tail) items[n++] = (int32_t)(0x80000000u | (uint32_t)bq[bq_head++]);
                if (bq_tail - bq_head > 256) items[n++] = (int32_t)(0x80000000u | (uint32_t)bq[bq_head++]);
            }
            done_pos[p] = pos;
        }
        while (next_pending < n_params) {
            const SvdqParam &q = pl->h_params[next_pending++];
            for (int32_t j = 0; j < q.unit_count; ++j) bq[bq_tail++] = q.unit_begin + j;
        }
        while (bq_head < bq_tail) items[n++] = (int32_t)(0x80000000u | (uint32_t)bq[bq_head++]);
        free(bq);
        free(done_pos);
        pl->ctl_bytes = (int64_t)sizeof(int32_t) * (4 + (int64_t)n_params * (SVDQ_RC + 2));
        hipError_t fe = (n == pl->n_items) ? hipSuccess : hipErrorUnknown;
        if (fe == hipSuccess) fe = hipMalloc((void **)&pl->d_items, sizeof(int32_t) * (size_t)pl->n_items);
        if (fe == hipSuccess) fe = hipMalloc((void **)&pl->d_ctl, (size_t)pl->ctl_bytes);
        if (fe == hipSuccess)
            fe = hipMemcpy(pl->d_items, items, sizeof(int32_t) * (size_t)pl->n_items, hipMemcpyHostToDevice);
        free(items);
        if (fe != hipSuccess) {
            svdq_set_error("fused schedule setup failed: %s", hipGetErrorString(fe));
            svdq_plan_destroy(pl);
            return SVDQ_EHIP;
        }
    }

    // cache-resident pipeline: groups of consecutive parameters with >= group_mb MB of input
    const int group_mb = pl->fused ? 0 : (cfg->reserved >> 8) & 0xffff;
    pl->lag = (cfg->reserved >> 4) & 0xf;
    if (pl->lag < 1) pl->lag = 2;
    if (group_mb > 0) {
        pl->grp_p0 = (int32_t *)calloc(n_params, sizeof(int32_t));
        pl->grp_n = (int32_t *)calloc(n_params, sizeof(int32_t));
        int p0 = 0;
        double acc = 0.0;
        for (int p = 0; p < n_params; ++p) {
            acc += (double)rows[p] * N * 4.0 / 1e6;
            if (acc >= group_mb || p == n_params - 1) {
                pl->grp_p0[pl->n_groups] = p0;
                pl->grp_n[pl->n_groups] = p + 1 - p0;
                ++pl->n_groups;
                p0 = p + 1;
                acc = 0.0;
            }
        }
        pl->ev_gram = (hipEvent_t *)calloc(pl->n_groups, sizeof(hipEvent_t));
        pl->ev_eig = (hipEvent_t *)calloc(pl->n_groups, sizeof(hipEvent_t));
        pl->ev_bp = (hipEvent_t *)calloc(pl->n_groups, sizeof(hipEvent_t));
        hipError_t ee = hipSuccess;
        for (int i = 0; i < SVDQ_NSIDE && ee == hipSuccess; ++i)
            ee = hipStreamCreateWithFlags(&pl->side[i], hipStreamNonBlocking);
        for (int g = 0; g < pl->n_groups && ee == hipSuccess; ++g) {
            ee = hipEventCreateWithFlags(&pl->ev_gram[g], hipEventDisableTiming);
            if (ee == hipSuccess) ee = hipEventCreateWithFlags(&pl->ev_eig[g], hipEventDisableTiming);
            if (ee == hipSuccess) ee = hipEventCreateWithFlags(&pl->ev_bp[g], hipEventDisableTiming);
        }
        if (ee == hipSuccess) ee = hipEventCreateWithFlags(&pl->ev_start, hipEventDisableTiming);
        if (ee == hipSuccess) ee = hipStreamCreateWithFlags(&pl->gram_stream, hipStreamNonBlocking);
        if (ee != hipSuccess) {
            svdq_set_error("pipeline stream/event creation failed: %s", hipGetErrorString(ee));
            svdq_plan_destroy(pl);
            return SVDQ_EHIP;
        }
    }

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
    if (pl->d_items) (void)hipFree(pl->d_items);
    if (pl->d_ctl) (void)hipFree(pl->d_ctl);
    for (int g = 0; g < pl->n_groups; ++g) {
        if (pl->ev_gram && pl->ev_gram[g]) (void)hipEventDestroy(pl->ev_gram[g]);
        if (pl->ev_eig && pl->ev_eig[g]) (void)hipEventDestroy(pl->ev_eig[g]);
        if (pl->ev_bp && pl->ev_bp[g]) (void)hipEventDestroy(pl->ev_bp[g]);
    }
    if (pl->ev_start) (void)hipEventDestroy(pl->ev_start);
    if (pl->gram_stream) (void)hipStreamDestroy(pl->gram_stream);
    free(pl->ev_bp);
    for (int i = 0; i < SVDQ_NSIDE; ++i)
        if (pl->side[i]) (void)hipStreamDestroy(pl->side[i]);
    free(pl->grp_p0);
    free(pl->grp_n);
    free(pl->ev_gram);
    free(pl->ev_eig);
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
                      int32_t nparams, const void *idx, void *stream, const void *base = nullptr) {
    if (!pl || !ptrs || !workspace) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    if (int rc = check_range(pl, param0, nparams)) return rc;
    int u0, nu;
    unit_range(pl, param0, nparams, &u0, &nu);
    return svdq_launch_gram(pl, ptrs, rows_dev, reinterpret_cast<double *>(ws(workspace, pl->ws_gram_off)), u0, nu,
                            pl->cfg.center, idx, base, (hipStream_t)stream);
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
    if (int rc = svdq_launch_gram(pl, ptrs, rows_dev, part, 0, pl->n_units, /*center=*/0, nullptr, nullptr, st)) return rc;
    if (int rc = svdq_launch_reduce(pl, part, part2, 0, pl->n_params, st)) return rc;
    return svdq_launch_gram_total(pl, part2, out_gram, st);
}

static int eig_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace, void *small,
                     int32_t param0, int32_t nparams, const void *idx, void *stream, const void *base = nullptr) {
    if (!pl || !ptrs || !workspace || !small) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    if (int rc = check_range(pl, param0, nparams)) return rc;
    if (int rc = svdq_launch_reduce(pl, reinterpret_cast<const double *>(ws(workspace, pl->ws_gram_off)),
                                    reinterpret_cast<double *>(ws(workspace, pl->ws_gram2_off)), param0, nparams,
                                    (hipStream_t)stream))
        return rc;
    return svdq_launch_eig(pl, ptrs, rows_dev, reinterpret_cast<const double *>(ws(workspace, pl->ws_gram2_off)),
                           reinterpret_cast<float *>(ws(workspace, pl->ws_w_off)),
                           reinterpret_cast<double *>(ws(workspace, pl->ws_c0_off)), reinterpret_cast<uint8_t *>(small),
                           param0, nparams, idx, base, (hipStream_t)stream);
}

extern "C" int svdq_eig_rank_select_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev,
                                          void *workspace, void *small, int32_t param0, int32_t nparams,
                                          void *stream) {
    return eig_range(pl, ptrs, rows_dev, workspace, small, param0, nparams, nullptr, stream);
}

static int bp_range(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace, const void *small,
                    void *basis, float *mean, int32_t param0, int32_t nparams, const void *idx, void *stream,
                    const void *base = nullptr) {
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
                                     pl->cfg.reserved & 1, idx, base, (hipStream_t)stream);
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
                                    (hipStream_t)stream))
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

// Cache-resident schedule over groups of consecutive parameters (>= group_mb MB of deltas each):
//   gram stream   : gram(0) gram(1) ...            gram(g) waits for bp(g - lag)  (back-pressure)
//   side streams  : eig(g) as soon as gram(g) is done (independent solves, dealt over SVDQ_NSIDE streams)
//   caller stream : bp(g) as soon as eig(g) is done, finally the coefficient epilogue
// Two streaming kernels (one gram, one basis_project) are in flight at any time, so the ramp and tail
// of per-group launches overlap, and basis_project(g) runs while group g's deltas are still (mostly)
// resident in the 256 MiB Infinity Cache.  Everything is joined back into the caller's stream.
static int compress_pipelined(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace,
                              void *small, void *basis, float *mean, hipStream_t main) {
    const int G = pl->n_groups;
    HIP_TRY(hipEventRecord(pl->ev_start, main));
    HIP_TRY(hipStreamWaitEvent(pl->gram_stream, pl->ev_start, 0));
    for (int i = 0; i < SVDQ_NSIDE; ++i) HIP_TRY(hipStreamWaitEvent(pl->side[i], pl->ev_start, 0));
    for (int g = 0; g < G; ++g) {
        if (g >= pl->lag) HIP_TRY(hipStreamWaitEvent(pl->gram_stream, pl->ev_bp[g - pl->lag], 0));
        int rc = svdq_gram_center_range(pl, ptrs, rows_dev, workspace, pl->grp_p0[g], pl->grp_n[g], pl->gram_stream);
        if (rc != SVDQ_OK) return rc;
        HIP_TRY(hipEventRecord(pl->ev_gram[g], pl->gram_stream));
        hipStream_t sd = pl->side[g % SVDQ_NSIDE];
        HIP_TRY(hipStreamWaitEvent(sd, pl->ev_gram[g], 0));
        rc = svdq_eig_rank_select_range(pl, ptrs, rows_dev, workspace, small, pl->grp_p0[g], pl->grp_n[g], sd);
        if (rc != SVDQ_OK) return rc;
        HIP_TRY(hipEventRecord(pl->ev_eig[g], sd));
        HIP_TRY(hipStreamWaitEvent(main, pl->ev_eig[g], 0));
        rc = svdq_basis_project_range(pl, ptrs, rows_dev, workspace, small, basis, mean, pl->grp_p0[g], pl->grp_n[g],
                                      main);
        if (rc != SVDQ_OK) return rc;
        HIP_TRY(hipEventRecord(pl->ev_bp[g], main));
    }
    return svdq_coeff_quantize(pl, workspace, small, main);
}

// One persistent launch for gram + eig + basis_project (k_fused), then the coefficient epilogue.
static int compress_fused(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace, void *small,
                          void *basis, float *mean, hipStream_t st) {
    if (!ptrs || !workspace || !small || !basis) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(basis) & 255) != 0) {
        svdq_set_error("basis buffer must be 256-byte aligned");
        return SVDQ_EINVAL;
    }
    uint8_t *sm = reinterpret_cast<uint8_t *>(small);
    HIP_TRY(hipMemsetAsync(pl->d_ctl, 0, (size_t)pl->ctl_bytes, st));
    int rc = svdq_launch_fused(pl, ptrs, rows_dev, reinterpret_cast<double *>(ws(workspace, pl->ws_gram_off)),
                               reinterpret_cast<double *>(ws(workspace, pl->ws_gram2_off)),
                               reinterpret_cast<float *>(ws(workspace, pl->ws_w_off)),
                               reinterpret_cast<double *>(ws(workspace, pl->ws_c0_off)), sm,
                               reinterpret_cast<uint8_t *>(basis), mean,
                               reinterpret_cast<double *>(ws(workspace, pl->ws_cpart_off)), st);
    if (rc != SVDQ_OK) return rc;
    HIP_TRY(hipMemcpyAsync(sm + pl->small.status_off, pl->d_ctl + 1, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return svdq_coeff_quantize(pl, workspace, small, st);
}

extern "C" int svdq_compress(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, void *workspace,
                             void *small, void *basis, float *mean, void *stream) {
    if (pl && pl->fused) return compress_fused(pl, ptrs, rows_dev, workspace, small, basis, mean, (hipStream_t)stream);
    if (pl && small)
        HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(small) + pl->small.status_off, 0, sizeof(int32_t),
                               (hipStream_t)stream));
    if (pl && pl->n_groups > 1)
        return compress_pipelined(pl, ptrs, rows_dev, workspace, small, basis, mean, (hipStream_t)stream);
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

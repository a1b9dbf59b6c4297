// svdq_merge.hip -- the consumers of a plan's artifacts, batched over the plan.
// Compiled with -ffp-contract=off: every product and sum below is rounded where the reference's torch ops round.
//
//   k_merge_coeff        dequantize_and_average (merge.py:61-141; RTVQQuantizer.dequantize rtvq.py:85-103) for every
//                        parameter of the plan at once, straight from the packed small-artifact buffer the compressor
//                        left in HBM (codes, scale, zero_point, c_high): no host copy, no per-task launch.  One
//                        coefficient vector per (parameter, set); a set is "all tasks" (merge_all_parameters),
//                        one cluster (merge_with_clustering merge.py:555-626) or one task (diagnostics).
//   k_merge_reconstruct  reconstruct_from_coefficients (merge.py:144-194) over the plan's unit table in ONE streaming
//                        launch: out = ((U_high c_high + U_low c_low) + mean) * scale per set, the sets combined with
//                        their shares (merge_cluster_results clustering.py:374-425 / apply_weights_to_tensors
//                        weighting.py:332-372 -- the merge is linear, so any number of clusters costs one pass over U),
//                        + base (apply_merged_deltas merge.py:429-552) when asked.  HBM-bound: reads the basis once
//                        (e (k + nl) B/row), mean and base, writes 4 B/row.
//   k_diag / k_diag_finish  compute_parameter_diagnostics' inner loop (diagnostics.py:186-215) for all N tasks of a
//                        parameter in one pass over U and the N deltas: N error tuples per parameter.
//
// Per-row arithmetic is that of k_reconstruct / k_recon_error (svdq_elem.hip): fp32 fma chains from 0 over the columns
// in order, hi + lo, + mean, * scale -- so a parameter's merged rows are the same bits as the per-parameter route's.

#include "svdq_common.h"
#include <hip/hip_fp16.h>

#define MRG_MAX_SETS 8

// ------------------------------------------------------------------------------------ coefficients
// weights [P or 1][n_sets][N]: weight of task t inside set s, renormalised over the set's present tasks by the caller
// exactly as the reference does on the host (merge.py:123-124); < 0 = the task is not in the set.
// order   [P or 1][N]: task indices in the order the reference adds them (sorted task names, merge.py:89); NULL = 0..N-1.
// cbar    [P][n_sets][N] out: columns 0..k-1 the averaged c_high, k..r-1 the averaged dequantized c_low, 0 beyond.
__global__ __launch_bounds__(64) void k_merge_coeff(int NT, int stages, int n_sets, int per_param,
                                                    const int32_t *__restrict__ k_in, const int32_t *__restrict__ r_in,
                                                    const uint16_t *__restrict__ chigh, const uint8_t *__restrict__ codes,
                                                    const float *__restrict__ scale, const float *__restrict__ zp,
                                                    const float *__restrict__ weights, const int32_t *__restrict__ order,
                                                    float *__restrict__ cbar) {
    const int p = blockIdx.x, i = threadIdx.x, n = NT;
    const int k = k_in[p], r = r_in[p];
    const float *w = weights + (per_param ? (size_t)p * n_sets * n : 0);
    const int32_t *ord = order ? order + (per_param ? (size_t)p * n : 0) : nullptr;
    if (i >= n) return;
    for (int s = 0; s < n_sets; ++s) {
        float acc = 0.f;
        if (i < r) {
            for (int tt = 0; tt < n; ++tt) {
                const int t = ord ? ord[tt] : tt;
                const float wt = w[s * n + t];
                if (wt < 0.f) continue;
                float c;
                if (i < k) {
                    c = __half2float(__ushort_as_half(chigh[((size_t)p * n + t) * n + i]));
                } else {
                    // zeros + sum over stages of (q - zero_point) / scale (rtvq.py:29-36, :85-103)
                    c = 0.f;
                    const size_t sb = ((size_t)p * n + t) * stages;
                    for (int st = 0; st < stages; ++st) {
                        const float q = (float)codes[(sb + st) * n + (i - k)];
                        c = __fadd_rn(c, __fdiv_rn(__fsub_rn(q, zp[sb + st]), scale[sb + st]));
                    }
                }
                acc = __fadd_rn(acc, __fmul_rn(c, wt));      // (stack * w).sum(0), task by task
            }
        }
        cbar[((size_t)p * n_sets + s) * n + i] = acc;
    }
}

// ------------------------------------------------------------------------------------ streaming pass
template <bool U16> struct UElem;
template <> struct UElem<true> { using type = __half; };
template <> struct UElem<false> { using type = float; };
__device__ __forceinline__ float u_val(const __half *u, int i) { return __half2float(u[i]); }
__device__ __forceinline__ float u_val(const float *u, int i) { return u[i]; }

// global -> LDS, 16 B per lane, whole tile contiguous (the mirror image of pass 2's copy_out)
__device__ __forceinline__ void copy_in(void *lds_dst, const uint8_t *gsrc, int nbytes, int lane) {
    const int nvec = nbytes >> 4;
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(gsrc);
    f32x4 *d4 = reinterpret_cast<f32x4 *>(lds_dst);
    for (int i = lane; i < nvec; i += 64) d4[i] = s4[i];
    uint8_t *db = reinterpret_cast<uint8_t *>(lds_dst);
    for (int b = (nvec << 4) + 2 * lane; b < nbytes; b += 128)
        *reinterpret_cast<uint16_t *>(db + b) = *reinterpret_cast<const uint16_t *>(gsrc + b);
}

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wavefront per work unit of the plan; 256-row blocks; lane l owns rows l, 64 + l, 128 + l, 192 + l of a block (the
// [256, k] / [256, nl] tiles arrive in LDS with 16-byte loads; a lane's row reads are then conflict-free).  Columns
// outermost: one coefficient read serves the lane's four rows, and the four fma chains are independent.
// NS = compiled-in number of sets (1, 2, 4, 8 >= n_sets).
template <bool U16, int NS>
__global__ __launch_bounds__(64) void k_merge_reconstruct(const SvdqParam *__restrict__ params,
                                                          const SvdqUnit *__restrict__ units,
                                                          const int64_t *__restrict__ rows_dev, int NT, int n_sets,
                                                          int per_param, const int32_t *__restrict__ k_in,
                                                          const int32_t *__restrict__ r_in,
                                                          const uint8_t *__restrict__ basis,
                                                          const float *__restrict__ meanbuf,
                                                          const float *__restrict__ cbar,
                                                          const float *__restrict__ set_share,
                                                          const float *__restrict__ scale_tab,
                                                          const float *const *__restrict__ base_ptrs,
                                                          float *const *__restrict__ out_ptrs) {
    using T = typename UElem<U16>::type;
    constexpr int ES = U16 ? 2 : 4;
    // dynamic LDS: [256 x N] basis elements (the U_high tile, then the U_low tile), then the coefficient sets
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    T *UT = reinterpret_cast<T *>(lds_raw);
    float *C = reinterpret_cast<float *>(lds_raw + (size_t)SVDQ_BLK_ROWS * NT * ES);   // [column][NS]
    float *SH = C + NS * NT;
    const int lane = threadIdx.x;
    const SvdqUnit ud = units[blockIdx.x];
    const int p = ud.param;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    const int64_t r_begin = ud.row0;
    int64_t r_end = r_begin + ud.nrows;
    if (r_end > D) r_end = D;
    if (r_begin >= r_end) return;
    const int k = k_in[p], r = r_in[p], nl = r - k, n = NT;
    for (int e = lane; e < NS * n; e += 64) {      // transposed: the NS coefficients of a column side by side
        const int i = e / NS, s = e % NS;
        C[e] = s < n_sets ? cbar[((size_t)p * n_sets + s) * n + i] : 0.f;
    }
    if (lane < NS)
        SH[lane] = (set_share && lane < n_sets) ? set_share[(per_param ? (size_t)p * n_sets : 0) + lane] : -1.f;
    const float scale = scale_tab ? scale_tab[p] : 1.f;
    const uint8_t *slab = basis + params[p].slab_off;
    const uint8_t *gUh = slab;
    const uint8_t *gUl = slab + svdq_align_up(D * (int64_t)k * ES, 256);
    const float *gmean = meanbuf ? meanbuf + params[p].mean_off : nullptr;
    const float *gbase = base_ptrs ? base_ptrs[p] : nullptr;
    float *gout = out_ptrs[p];
    T *Uh = UT, *Ul = UT + SVDQ_BLK_ROWS * k;

    for (int64_t rb = r_begin; rb < r_end; rb += SVDQ_BLK_ROWS) {
        const int rows_blk = (int)((D - rb < SVDQ_BLK_ROWS) ? (D - rb) : SVDQ_BLK_ROWS);
        if (k > 0) copy_in(Uh, gUh + rb * (int64_t)k * ES, rows_blk * k * ES, lane);
        if (nl > 0) copy_in(Ul, gUl + rb * (int64_t)nl * ES, rows_blk * nl * ES, lane);
        float mv[4], bv[4];
        int rl[4];      // the lane's rows, clamped into the block (results of clamped rows are not stored)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int q = 64 * m + lane;
            rl[m] = q < rows_blk ? q : rows_blk - 1;
            mv[m] = gmean ? gmean[rb + rl[m]] : 0.f;
            bv[m] = gbase ? gbase[rb + rl[m]] : 0.f;
        }
        lds_fence();
        float hi[4][NS], lo[4][NS];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int s = 0; s < NS; ++s) hi[m][s] = lo[m][s] = 0.f;
        for (int i = 0; i < k; ++i) {
            float c[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) c[s] = C[i * NS + s];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float u = u_val(Uh, rl[m] * k + i);
#pragma unroll
                for (int s = 0; s < NS; ++s) hi[m][s] = fmaf(u, c[s], hi[m][s]);
            }
        }
        for (int j = 0; j < nl; ++j) {
            float c[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) c[s] = C[(k + j) * NS + s];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float u = u_val(Ul, rl[m] * nl + j);
#pragma unroll
                for (int s = 0; s < NS; ++s) lo[m][s] = fmaf(u, c[s], lo[m][s]);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float res = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float v = __fadd_rn(hi[m][s], lo[m][s]);
                if (gmean) v = __fadd_rn(v, mv[m]);
                v = __fmul_rn(v, scale);
                if (set_share) {
                    if (SH[s] >= 0.f) res = __fadd_rn(res, __fmul_rn(v, SH[s]));   // (stack * w).sum(0), set by set
                } else if (s == 0) {
                    res = v;
                }
            }
            if (gbase) res = __fadd_rn(bv[m], res);      // base + delta (merge.py:429-552)
            if (64 * m + lane < rows_blk) gout[rb + 64 * m + lane] = res;
        }
        lds_fence();
    }
}

// ------------------------------------------------------------------------------------ diagnostics
// One pass over U and the N task deltas of a parameter: per task t the reconstruction U_high c_high[t] + U_low c_low[t]
// (+ mean when add_mean: the reference's diagnostics do NOT add it back) is formed per row and compared with
// delta_t -- N x {sum e^2, sum x^2, sum rec^2, sum |e|, max |e|} per unit, reduced per parameter in a fixed order.
struct DiagPart {
    double se, sx, sr, sa, mx;
};

template <int NTP, bool U16>
__global__ __launch_bounds__(64) void k_diag(const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
                                             const float *const *__restrict__ ptrs,
                                             const int64_t *__restrict__ rows_dev, int NT,
                                             const int32_t *__restrict__ k_in, const int32_t *__restrict__ r_in,
                                             const uint8_t *__restrict__ basis, const float *__restrict__ meanbuf,
                                             int add_mean, const float *__restrict__ ctask /* [P][N][N] */,
                                             DiagPart *__restrict__ part /* [n_units][N] */) {
    using T = typename UElem<U16>::type;
    constexpr int ES = U16 ? 2 : 4;
    __shared__ __attribute__((aligned(16))) T UT[SVDQ_BLK_ROWS * NTP];
    __shared__ float C[NTP * NTP];
    const int lane = threadIdx.x, n = NT;
    const SvdqUnit ud = units[blockIdx.x];
    const int p = ud.param;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    const int64_t r_begin = ud.row0;
    int64_t r_end = r_begin + ud.nrows;
    if (r_end > D) r_end = D;
    const int k = k_in[p], r = r_in[p], nl = r - k;
    double se[NTP], sx[NTP], sr[NTP], sa[NTP];
    float mx[NTP];
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
        se[t] = sx[t] = sr[t] = sa[t] = 0.0;
        mx[t] = 0.f;
    }
    if (r_begin < r_end) {
        for (int e = lane; e < n * n; e += 64) C[e] = ctask[(size_t)p * n * n + e];
        const uint8_t *slab = basis + params[p].slab_off;
        const uint8_t *gUh = slab;
        const uint8_t *gUl = slab + svdq_align_up(D * (int64_t)k * ES, 256);
        const float *gmean = (add_mean && meanbuf) ? meanbuf + params[p].mean_off : nullptr;
        const float *dp[NTP];
#pragma unroll
        for (int t = 0; t < NTP; ++t) dp[t] = ptrs[(size_t)p * n + (t < n ? t : n - 1)];
        T *Uh = UT, *Ul = UT + SVDQ_BLK_ROWS * k;
        for (int64_t rb = r_begin; rb < r_end; rb += SVDQ_BLK_ROWS) {
            const int rows_blk = (int)((D - rb < SVDQ_BLK_ROWS) ? (D - rb) : SVDQ_BLK_ROWS);
            if (k > 0) copy_in(Uh, gUh + rb * (int64_t)k * ES, rows_blk * k * ES, lane);
            if (nl > 0) copy_in(Ul, gUl + rb * (int64_t)nl * ES, rows_blk * nl * ES, lane);
            lds_fence();
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int rl = 64 * m + lane;
                if (rl < rows_blk) {
                    float x[NTP];
#pragma unroll
                    for (int t = 0; t < NTP; ++t) x[t] = (t < n) ? dp[t][rb + rl] : 0.f;
                    const float mval = gmean ? gmean[rb + rl] : 0.f;
                    float hi[NTP], lo[NTP];
#pragma unroll
                    for (int t = 0; t < NTP; ++t) hi[t] = lo[t] = 0.f;
                    for (int i = 0; i < k; ++i) {
                        const float u = u_val(Uh, rl * k + i);
#pragma unroll
                        for (int t = 0; t < NTP; ++t) hi[t] = fmaf(u, C[(t < n ? t : 0) * n + i], hi[t]);
                    }
                    for (int j = 0; j < nl; ++j) {
                        const float u = u_val(Ul, rl * nl + j);
#pragma unroll
                        for (int t = 0; t < NTP; ++t) lo[t] = fmaf(u, C[(t < n ? t : 0) * n + k + j], lo[t]);
                    }
#pragma unroll
                    for (int t = 0; t < NTP; ++t) {
                        float rec = __fadd_rn(hi[t], lo[t]);
                        if (gmean) rec = __fadd_rn(rec, mval);
                        const float e = __fsub_rn(x[t], rec);
                        se[t] += (double)e * e;
                        sx[t] += (double)x[t] * x[t];
                        sr[t] += (double)rec * rec;
                        sa[t] += fabs((double)e);
                        mx[t] = fmaxf(mx[t], fabsf(e));
                        if (e != e) mx[t] = e;      // NaN propagates like torch.max
                    }
                }
            }
            lds_fence();
        }
    }
    // wave reduction in a fixed order (xor butterflies), lane 0 writes the unit's partials
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
        double a = se[t], b = sx[t], c = sr[t], d = sa[t];
        float q = mx[t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
            c += __shfl_xor(c, off);
            d += __shfl_xor(d, off);
            const float o = __shfl_xor(q, off);
            q = (q != q) ? q : ((o != o) ? o : (o > q ? o : q));
        }
        if (lane == 0 && t < n) {
            DiagPart &dst = part[(size_t)blockIdx.x * n + t];
            dst.se = a;
            dst.sx = b;
            dst.sr = c;
            dst.sa = d;
            dst.mx = (double)q;
        }
    }
}

// out [P][N][6] = absolute_error, relative_error, max_absolute_error, mean_absolute_error, original_norm,
// reconstructed_norm (diagnostics.py:72-117; fp32 norms like the reference's tensors)
__global__ __launch_bounds__(64) void k_diag_finish(const SvdqParam *__restrict__ params,
                                                    const int64_t *__restrict__ rows_dev, int NT,
                                                    const DiagPart *__restrict__ part, double *__restrict__ out) {
    const int p = blockIdx.x, t = blockIdx.y, lane = threadIdx.x, n = NT;
    const SvdqParam pd = params[p];
    const int64_t rows = rows_dev ? rows_dev[p] : pd.rows;
    double se = 0.0, sx = 0.0, sr = 0.0, sa = 0.0, mx = 0.0;
    for (int u = lane; u < pd.unit_count; u += 64) {      // lane-strided, then a fixed butterfly
        const DiagPart v = part[(size_t)(pd.unit_begin + u) * n + t];
        se += v.se;
        sx += v.sx;
        sr += v.sr;
        sa += v.sa;
        mx = (mx != mx) ? mx : ((v.mx != v.mx) ? v.mx : (v.mx > mx ? v.mx : mx));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        se += __shfl_xor(se, off);
        sx += __shfl_xor(sx, off);
        sr += __shfl_xor(sr, off);
        sa += __shfl_xor(sa, off);
        const double o = __shfl_xor(mx, off);
        mx = (mx != mx) ? mx : ((o != o) ? o : (o > mx ? o : mx));
    }
    if (lane == 0) {
        double *o6 = out + ((size_t)p * n + t) * 6;
        const float en = (float)sqrt(se), on = (float)sqrt(sx);
        o6[0] = (double)en;
        o6[1] = on > 1e-10f ? (double)en / (double)on : 0.0;
        o6[2] = mx;
        o6[3] = rows > 0 ? (double)(float)(sa / (double)rows) : 0.0;
        o6[4] = (double)on;
        o6[5] = (double)(float)sqrt(sr);
    }
}

// ------------------------------------------------------------------------------------ entry points
static int merge_args_ok(const svdq_plan *pl, const void *small, int32_t n_sets, const char *who) {
    if (!pl || !small) {
        svdq_set_error("%s: plan and small are required", who);
        return SVDQ_EINVAL;
    }
    if (n_sets < 1 || n_sets > SVDQ_MAX_TASKS) {
        svdq_set_error("%s: n_sets must be in [1, %d], got %d", who, SVDQ_MAX_TASKS, n_sets);
        return SVDQ_EINVAL;
    }
    return SVDQ_OK;
}

extern "C" int64_t svdq_merge_work_bytes(const svdq_plan *pl, int32_t n_sets) {
    if (!pl || n_sets < 1) return 0;
    return svdq_align_up((int64_t)pl->n_params * n_sets * pl->n_tasks * 4, 256);
}

extern "C" int svdq_merge_coeffs(const svdq_plan *pl, const void *small, const float *weights, const int32_t *order,
                                 int32_t n_sets, int32_t per_param, float *cbar, void *stream) {
    if (int rc = merge_args_ok(pl, small, n_sets, "svdq_merge_coeffs")) return rc;
    if (!weights || !cbar) {
        svdq_set_error("svdq_merge_coeffs: weights and cbar are required");
        return SVDQ_EINVAL;
    }
    const svdq_small_layout &L = pl->small;
    const uint8_t *sm = reinterpret_cast<const uint8_t *>(small);
    hipLaunchKernelGGL(k_merge_coeff, dim3(pl->n_params), dim3(64), 0, (hipStream_t)stream, pl->n_tasks,
                       pl->cfg.rtvq_stages, n_sets, per_param, reinterpret_cast<const int32_t *>(sm + L.k_off),
                       reinterpret_cast<const int32_t *>(sm + L.r_off), reinterpret_cast<const uint16_t *>(sm + L.chigh_off),
                       sm + L.codes_off, reinterpret_cast<const float *>(sm + L.scale_off),
                       reinterpret_cast<const float *>(sm + L.zp_off), weights, order, cbar);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_merge_reconstruct(const svdq_plan *pl, const int64_t *rows_dev, const void *small, const void *basis,
                                      const float *mean, const float *cbar, int32_t n_sets, int32_t per_param,
                                      const float *set_share, const float *scale, const void *base_ptrs,
                                      const void *out_ptrs, void *stream) {
    if (int rc = merge_args_ok(pl, small, n_sets, "svdq_merge_reconstruct")) return rc;
    if (!basis || !cbar || !out_ptrs) {
        svdq_set_error("svdq_merge_reconstruct: basis, cbar and out_ptrs are required");
        return SVDQ_EINVAL;
    }
    if (n_sets > MRG_MAX_SETS) {
        svdq_set_error("svdq_merge_reconstruct: at most %d sets (clusters) per pass, got %d", MRG_MAX_SETS, n_sets);
        return SVDQ_EUNSUPPORTED;
    }
    if (n_sets > 1 && !set_share) {
        svdq_set_error("svdq_merge_reconstruct: set_share is required when n_sets > 1");
        return SVDQ_EINVAL;
    }
    const svdq_small_layout &L = pl->small;
    const uint8_t *sm = reinterpret_cast<const uint8_t *>(small);
    auto kk = reinterpret_cast<const int32_t *>(sm + L.k_off), rr = reinterpret_cast<const int32_t *>(sm + L.r_off);
    auto bp = reinterpret_cast<const float *const *>(base_ptrs);
    auto op = reinterpret_cast<float *const *>(out_ptrs);
    hipStream_t st = (hipStream_t)stream;
    const int ns = n_sets == 1 ? 1 : (n_sets == 2 ? 2 : (n_sets <= 4 ? 4 : 8));
    const size_t lds = (size_t)SVDQ_BLK_ROWS * pl->n_tasks * (pl->cfg.fp16 ? 2 : 4) + (size_t)(ns * pl->n_tasks + ns) * 4;
    const uint8_t *bs = reinterpret_cast<const uint8_t *>(basis);
    const float *mn = pl->cfg.center ? mean : nullptr;
#define SVDQ_MRG_LAUNCH(F16, NS_)                                                                                      \
    hipLaunchKernelGGL((k_merge_reconstruct<F16, NS_>), dim3(pl->n_units), dim3(64), lds, st, pl->d_params, pl->d_units, \
                       rows_dev, pl->n_tasks, n_sets, per_param, kk, rr, bs, mn, cbar, set_share, scale, bp, op)
    if (pl->cfg.fp16) {
        switch (ns) {
            case 1: SVDQ_MRG_LAUNCH(true, 1); break;
            case 2: SVDQ_MRG_LAUNCH(true, 2); break;
            case 4: SVDQ_MRG_LAUNCH(true, 4); break;
            default: SVDQ_MRG_LAUNCH(true, 8); break;
        }
    } else {
        switch (ns) {
            case 1: SVDQ_MRG_LAUNCH(false, 1); break;
            case 2: SVDQ_MRG_LAUNCH(false, 2); break;
            case 4: SVDQ_MRG_LAUNCH(false, 4); break;
            default: SVDQ_MRG_LAUNCH(false, 8); break;
        }
    }
#undef SVDQ_MRG_LAUNCH
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_merge(const svdq_plan *pl, const int64_t *rows_dev, const void *small, const void *basis,
                          const float *mean, const float *weights, const int32_t *order, int32_t n_sets,
                          int32_t per_param, const float *set_share, const float *scale, const void *base_ptrs,
                          const void *out_ptrs, void *work, void *stream) {
    if (!work) {
        svdq_set_error("svdq_merge: work is required (svdq_merge_work_bytes)");
        return SVDQ_EINVAL;
    }
    float *cbar = reinterpret_cast<float *>(work);
    if (int rc = svdq_merge_coeffs(pl, small, weights, order, n_sets, per_param, cbar, stream)) return rc;
    return svdq_merge_reconstruct(pl, rows_dev, small, basis, mean, cbar, n_sets, per_param, set_share, scale, base_ptrs,
                                  out_ptrs, stream);
}

extern "C" int64_t svdq_diagnostics_work_bytes(const svdq_plan *pl) {
    if (!pl) return 0;
    const int64_t n = pl->n_tasks;
    return svdq_align_up((int64_t)pl->n_params * n * n * 4, 256) +              // per-task coefficients
           svdq_align_up(n * n * 4, 256) +                                      // one-hot weights [N][N]
           svdq_align_up((int64_t)pl->n_units * n * (int64_t)sizeof(DiagPart), 256);
}

__global__ void k_one_hot(int n, float *w) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * n) w[e] = (e / n == e % n) ? 1.f : -1.f;
}

template <int NTP>
static void launch_diag(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, const int32_t *kk,
                        const int32_t *rr, const uint8_t *basis, const float *mean, int add_mean, const float *ctask,
                        DiagPart *part, hipStream_t st) {
    auto pp = reinterpret_cast<const float *const *>(ptrs);
    if (pl->cfg.fp16)
        hipLaunchKernelGGL((k_diag<NTP, true>), dim3(pl->n_units), dim3(64), 0, st, pl->d_params, pl->d_units, pp,
                           rows_dev, pl->n_tasks, kk, rr, basis, mean, add_mean, ctask, part);
    else
        hipLaunchKernelGGL((k_diag<NTP, false>), dim3(pl->n_units), dim3(64), 0, st, pl->d_params, pl->d_units, pp,
                           rows_dev, pl->n_tasks, kk, rr, basis, mean, add_mean, ctask, part);
}

extern "C" int svdq_diagnostics(const svdq_plan *pl, const void *delta_ptrs, const int64_t *rows_dev, const void *small,
                                const void *basis, const float *mean, int32_t add_mean, double *out, void *work,
                                void *stream) {
    if (!pl || !delta_ptrs || !small || !basis || !out || !work) {
        svdq_set_error("svdq_diagnostics: bad argument");
        return SVDQ_EINVAL;
    }
    const int64_t n = pl->n_tasks;
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    float *ctask = reinterpret_cast<float *>(wb);
    float *onehot = reinterpret_cast<float *>(wb + svdq_align_up((int64_t)pl->n_params * n * n * 4, 256));
    DiagPart *part = reinterpret_cast<DiagPart *>(wb + svdq_align_up((int64_t)pl->n_params * n * n * 4, 256) +
                                                  svdq_align_up(n * n * 4, 256));
    // per-task coefficients = the "average" of one task with weight 1: sets = tasks, one-hot weights
    hipLaunchKernelGGL(k_one_hot, dim3(((int)(n * n) + 255) / 256), dim3(256), 0, st, (int)n, onehot);
    if (int rc = svdq_merge_coeffs(pl, small, onehot, nullptr, (int32_t)n, 0, ctask, stream)) return rc;
    const svdq_small_layout &L = pl->small;
    const uint8_t *sm = reinterpret_cast<const uint8_t *>(small);
    auto kk = reinterpret_cast<const int32_t *>(sm + L.k_off), rr = reinterpret_cast<const int32_t *>(sm + L.r_off);
    auto bs = reinterpret_cast<const uint8_t *>(basis);
#define SVDQ_DIAG_CASE(N_) \
    case N_: launch_diag<N_>(pl, delta_ptrs, rows_dev, kk, rr, bs, mean, add_mean, ctask, part, st); break
    switch (pl->ntp) {
        SVDQ_DIAG_CASE(4); SVDQ_DIAG_CASE(8); SVDQ_DIAG_CASE(12); SVDQ_DIAG_CASE(16);
        SVDQ_DIAG_CASE(20); SVDQ_DIAG_CASE(24); SVDQ_DIAG_CASE(28); SVDQ_DIAG_CASE(32);
        default:
            svdq_set_error("unsupported padded task count %d", pl->ntp);
            return SVDQ_EUNSUPPORTED;
    }
#undef SVDQ_DIAG_CASE
    hipLaunchKernelGGL(k_diag_finish, dim3(pl->n_params, (int)n), dim3(64), 0, st, pl->d_params, rows_dev, (int)n, part, out);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

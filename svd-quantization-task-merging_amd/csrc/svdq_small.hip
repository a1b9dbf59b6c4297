// svdq_small.hip -- the two latency-bound kernels between / after the streaming passes.
// Compiled with -ffp-contract=off: the quantizer must round scale*x and +zero_point separately.
//
//   k_eig    per parameter: deterministic fp64 reduction of the Gram partials, cyclic Jacobi
//            eigen-solve of the N x N Gram in fp64, sigma = sqrt(lambda), the reference's fp32
//            energy / rank rule (basis.py:116-156, 159-213, :367), W = V Sigma^-1.
//   k_coeff  per parameter: deterministic fp64 reduction of the projection partials,
//            c_high -> fp16 (compress.py:44-47), multi-stage affine quantization of c_low
//            (rtvq.py:4-82) -- one lane per task, n = r-k <= 31 scalars each.

#include "svdq_common.h"
#include <hip/hip_fp16.h>

#define EIG_THREADS 256
#define LDN 33  // padded leading dimension of the N x N LDS matrices (N <= 32)

// Sum partial matrices [slot][nn] over slots [s0, s1) into out[nn] (LDS), fixed order:
// wave w takes slots s0+w, s0+w+4, ... ; the four wave sums are then added 0+1+2+3.
__device__ void reduce_partials(const double *__restrict__ part, int s0, int s1, int nn, double *red /*[4][1024]*/,
                                double *out /*[nn]*/) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    for (int e = l; e < nn; e += 64) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.0;
        int s = s0 + w;
        for (; s + 28 < s1; s += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += part[(size_t)(s + 4 * u) * nn + e];
        }
        for (; s < s1; s += 4) a[0] += part[(size_t)s * nn + e];
        red[w * 1024 + e] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();
    for (int e = tid; e < nn; e += EIG_THREADS)
        out[e] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
    __syncthreads();
}

__global__ __launch_bounds__(EIG_THREADS) void k_eig(const SvdqParam *__restrict__ params,
                                                     const float *const *__restrict__ ptrs,
                                                     const int64_t *__restrict__ rows_dev, int NT, int pack,
                                                     int center, float thr, int max_rank, const double *__restrict__ gram_part,
                                                     float *__restrict__ Wtab, double *__restrict__ c0_out, int param0,
                                                     float *__restrict__ sigma_out,
                                                     int32_t *__restrict__ k_out, int32_t *__restrict__ r_out,
                                                     float *__restrict__ energy_out, int64_t *__restrict__ rows_out) {
    __shared__ double red[4 * 1024];
    __shared__ double G[1024];
    __shared__ double Gd[1024];
    __shared__ double xc0[32];
    __shared__ int s_i0;
    __shared__ double s_spike;
    __shared__ double A[32 * LDN];
    __shared__ double V[32 * LDN];
    __shared__ double lam[32];
    __shared__ double rowoff[32];
    __shared__ int order[32];
    __shared__ double sgn[32];
    __shared__ double sig[32];
    __shared__ int done;
    __shared__ int pr_p[16], pr_q[16];
    __shared__ double pr_c[16], pr_s[16];

    const int p = param0 + blockIdx.x, tid = threadIdx.x, n = NT;
    const SvdqParam pd = params[p];
    const int64_t D = rows_dev ? rows_dev[p] : pd.rows;
    reduce_partials(gram_part, pd.unit_begin * pack, (pd.unit_begin + pd.unit_count) * pack, n * n, red, G);

    // Centred rows sum to zero, so 1/sqrt(N) is an exact null vector of Tc.  The fp32-product Gram
    // only resolves sigma down to ~1e-4 sigma_0, so deflate that direction explicitly in fp64:
    // G <- C G C, C = I - 11^T/N.  (LAPACK reports ~1e-7 sigma_0 noise there; we report ~0.)
    if (center) {
        if (tid < n) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += 0.5 * (G[tid * n + j] + G[j * n + tid]);
            rowoff[tid] = s / n;
        }
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += rowoff[j];
            lam[0] = s / n;
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += EIG_THREADS) {
        const int i = e / n, j = e % n;
        // G is symmetric by construction (same products, same order); average anyway.
        double a = 0.5 * (G[i * n + j] + G[j * n + i]);
        if (center) a = a - rowoff[i] - rowoff[j] + lam[0];
        Gd[i * n + j] = a;  // deflated Gram, kept for the completion column's coefficients
        A[i * LDN + j] = a;
        V[i * LDN + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();

    // Parallel-order cyclic Jacobi: a round-robin tournament pairs all indices into M = ceil(n/2)
    // disjoint (p,q) per round (n_even - 1 rounds per sweep); the M rotations of a round commute, so
    // they are applied together: A <- A J (columns), then A <- J^T A (rows), V <- V J.
    // 3 barriers per ROUND instead of 2 per rotation: ~5x less latency at n = 8, ~10x at n = 32.
    const int M = (n + 1) >> 1, ne = 2 * M;
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid < n) {
            double off = 0.0;
            for (int j = 0; j < n; ++j)
                if (j != tid) off += A[tid * LDN + j] * A[tid * LDN + j];
            rowoff[tid] = off;
        }
        __syncthreads();
        if (tid == 0) {
            double off = 0.0, dg = 0.0;
            for (int j = 0; j < n; ++j) {
                off += rowoff[j];
                dg += A[j * LDN + j] * A[j * LDN + j];
            }
            done = (off <= 1e-30 * dg) || (dg == 0.0);
        }
        __syncthreads();
        if (done || n < 2) break;
        for (int rd = 0; rd < ne - 1; ++rd) {
            if (tid < M) {
                int a, b;
                if (tid == 0) {
                    a = ne - 1;
                    b = rd;
                } else {
                    a = (rd + tid) % (ne - 1);
                    b = (rd - tid + (ne - 1)) % (ne - 1);
                }
                const int pp = a < b ? a : b, q = a < b ? b : a;
                double cs = 1.0, sn = 0.0;
                if (q < n) {
                    const double app = A[pp * LDN + pp], aqq = A[q * LDN + q], apq = A[pp * LDN + q];
                    if (apq != 0.0) {
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        cs = 1.0 / sqrt(1.0 + t * t);
                        sn = t * cs;
                    }
                }
                pr_p[tid] = pp;
                pr_q[tid] = q;
                pr_c[tid] = cs;
                pr_s[tid] = sn;
            }
            __syncthreads();
            for (int e = tid; e < n * M; e += EIG_THREADS) {  // columns of A and V
                const int i = e / M, m = e % M;
                const int pp = pr_p[m], q = pr_q[m];
                if (q < n) {
                    const double cs = pr_c[m], sn = pr_s[m];
                    const double x = A[i * LDN + pp], y = A[i * LDN + q];
                    A[i * LDN + pp] = cs * x - sn * y;
                    A[i * LDN + q] = sn * x + cs * y;
                    const double vx = V[i * LDN + pp], vy = V[i * LDN + q];
                    V[i * LDN + pp] = cs * vx - sn * vy;
                    V[i * LDN + q] = sn * vx + cs * vy;
                }
            }
            __syncthreads();
            for (int e = tid; e < n * M; e += EIG_THREADS) {  // rows of A
                const int j = e / M, m = e % M;
                const int pp = pr_p[m], q = pr_q[m];
                if (q < n) {
                    const double cs = pr_c[m], sn = pr_s[m];
                    const double x = A[pp * LDN + j], y = A[q * LDN + j];
                    A[pp * LDN + j] = (j == q) ? 0.0 : cs * x - sn * y;
                    A[q * LDN + j] = (j == pp) ? 0.0 : sn * x + cs * y;
                }
            }
            __syncthreads();
        }
    }

    // sort descending (stable on ties), sign convention: largest-|v| component positive
    if (tid < n) lam[tid] = A[tid * LDN + tid];
    __syncthreads();
    if (tid < n) {
        int rank = 0;
        for (int i = 0; i < n; ++i)
            if (lam[i] > lam[tid] || (lam[i] == lam[tid] && i < tid)) ++rank;
        order[rank] = tid;
    }
    __syncthreads();
    if (tid < n) {
        const int col = order[tid];
        const double l = lam[col];
        sig[tid] = l > 0.0 ? sqrt(l) : 0.0;
        double best = 0.0, bv = 1.0;
        for (int j = 0; j < n; ++j) {
            const double x = V[j * LDN + col];
            if (fabs(x) > best) {
                best = fabs(x);
                bv = x;
            }
        }
        sgn[tid] = bv < 0.0 ? -1.0 : 1.0;
    }
    __syncthreads();

    const int r = (int)(D < (int64_t)n ? D : (int64_t)n);
    if (tid == 0) {
        // basis.py:147-156 and :199-211, fp32 like the reference (threshold compared as fp32)
        float S[32], cum[32];
        float total = 0.f;
        for (int i = 0; i < r; ++i) {
            S[i] = (float)sig[i];
            total += S[i] * S[i];
        }
        if (total < 1e-10f) {
            for (int i = 0; i < r; ++i) cum[i] = 1.f;
        } else {
            float run = 0.f;
            for (int i = 0; i < r; ++i) {
                run += S[i] * S[i];
                cum[i] = run / total;
            }
        }
        int kk = 1;
        for (int i = 0; i < r; ++i)
            if (cum[i] < thr) ++kk;
        if (kk < 1) kk = 1;
        if (max_rank > 0 && kk > max_rank) kk = max_rank;
        if (kk > r) kk = r;
        for (int i = 0; i < n; ++i) sigma_out[(size_t)p * n + i] = (i < r) ? S[i] : 0.f;
        k_out[p] = kk;
        r_out[p] = r;
        energy_out[p] = (kk > 0 && r > 0) ? cum[kk - 1] : 0.f;
        rows_out[p] = D;
    }
    // W[t][i] = sgn_i V[t][order[i]] / sigma_i.  Directions with sigma_i <= 1e-6 sigma_0 are below the
    // fp32 resolution of the data (LAPACK returns an arbitrary unit vector orthogonal to the rest
    // there).  The first such direction -- the one centring always creates -- gets an explicit
    // orthonormal completion u = (e_0 - U U[0,:]^T) / norm, i.e. one more W column
    // w[t] = -(sum_j W[t][j] U[0][j]) / norm plus a spike 1/norm at row 0 (added in pass 2);
    // any further null directions are zero columns ("null directions").
    const double s0 = sig[0];
    for (int e = tid; e < n * n; e += EIG_THREADS) {
        const int t = e / n, i = e % n;
        double wv = 0.0;
        if (i < r && sig[i] > 1e-6 * s0 && sig[i] > 0.0) wv = sgn[i] * V[t * LDN + order[i]] / sig[i];
        A[t * LDN + i] = wv;  // A is free after the sweeps
    }
    float *aux = Wtab + (size_t)p * (n * n + 4) + n * n;
    if (tid < n && D > 0) lam[tid] = (double)ptrs[(size_t)p * n + tid][0];  // row 0 of every task
    __syncthreads();
    if (tid == 0) {
        int i0 = -1;
        if (s0 > 0.0 && D > 0)
            for (int i = 0; i < r; ++i)
                if (!(sig[i] > 1e-6 * s0)) {
                    i0 = i;
                    break;
                }
        float spike = 0.f;
        if (i0 >= 0) {
            // centre row 0 exactly as the streaming kernels do (fp32, task order, one divide)
            float sum = 0.f;
            for (int t = 0; t < n; ++t) sum += (float)lam[t];
            const float mean0 = center ? sum / (float)n : 0.f;
            double norm2 = 1.0;
            for (int t = 0; t < n; ++t) xc0[t] = (double)((float)lam[t] - mean0);
            for (int j = 0; j < r; ++j) {
                double u = 0.0;
                for (int t = 0; t < n; ++t) u += xc0[t] * A[t * LDN + j];
                rowoff[j] = u;  // U[0][j]
                norm2 -= u * u;
            }
            if (norm2 > 0.25) {
                const double inv = 1.0 / sqrt(norm2);
                for (int t = 0; t < n; ++t) {
                    double acc = 0.0;
                    for (int j = 0; j < r; ++j) acc += A[t * LDN + j] * rowoff[j];
                    lam[t] = -inv * acc;  // lam (row 0 of the tasks) is already folded into xc0
                }
                for (int t = 0; t < n; ++t) A[t * LDN + i0] = lam[t];
                spike = (float)inv;
            } else {
                i0 = -1;
            }
        }
        aux[0] = spike;
        aux[1] = (float)i0;
        aux[2] = 0.f;
        aux[3] = 0.f;
        s_i0 = i0;
        s_spike = (double)spike;
    }
    __syncthreads();
    // W (fp32) and the closed-form coefficients c0[t][i] = u_i^T xc_t of the UNROUNDED basis:
    //   real direction:      sigma_i * v_i[t]           (U^T Tc = Sigma V^T)
    //   completion column:   w^T Gd[:,t] + spike * xc_t[row 0]
    //   zero column:         0
    // pass 2 adds the fp16-rounding correction E^T Tc on top (k_coeff sums both).
    for (int e = tid; e < n * n; e += EIG_THREADS) {
        const int t = e / n, i = e % n;
        Wtab[(size_t)p * (n * n + 4) + e] = (float)A[t * LDN + i];
        double cv = 0.0;
        if (i < r) {
            if (i == s_i0) {
                for (int t2 = 0; t2 < n; ++t2) cv += A[t2 * LDN + i] * Gd[t2 * n + t];
                cv += s_spike * xc0[t];
            } else if (sig[i] > 1e-6 * s0 && sig[i] > 0.0) {
                cv = sig[i] * sgn[i] * V[t * LDN + order[i]];
            }
        }
        c0_out[(size_t)p * n * n + e] = cv;
    }
}

// ------------------------------------------------------------------------------------ epilogue
__device__ __forceinline__ float f_min_nan(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ float f_max_nan(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

__global__ __launch_bounds__(EIG_THREADS) void k_coeff(const SvdqParam *__restrict__ params, int NT, int pack,
                                                       int bits, int stages, const double *__restrict__ cpart,
                                                       const double *__restrict__ c0_in,
                                                       const int32_t *__restrict__ k_in,
                                                       const int32_t *__restrict__ r_in,
                                                       float *__restrict__ coef_out, uint16_t *__restrict__ chigh_out,
                                                       uint8_t *__restrict__ codes_out, float *__restrict__ scale_out,
                                                       float *__restrict__ zp_out, float *__restrict__ rnorm_out, int param0) {
    __shared__ double red[4 * 1024];
    __shared__ double C[1024];
    __shared__ float res[32 * LDN];

    const int p = param0 + blockIdx.x, tid = threadIdx.x, n = NT;
    const SvdqParam pd = params[p];
    reduce_partials(cpart, pd.unit_begin * pack, (pd.unit_begin + pd.unit_count) * pack, n * n, red, C);

    const int k = k_in[p], r = r_in[p];
    const int nl = r - k;
    // c[t][i] fp32, c_high fp16 (round-to-nearest-even), zero padding past the valid prefix
    for (int e = tid; e < n * n; e += EIG_THREADS) {
        const int t = e / n, i = e % n;
        const float cv = (float)(c0_in[(size_t)p * n * n + e] + C[e]);
        coef_out[(size_t)p * n * n + e] = cv;
        chigh_out[(size_t)p * n * n + e] = (i < k) ? __half_as_ushort(__float2half_rn(cv)) : (uint16_t)0;
        if (i >= k && i < r) res[t * LDN + (i - k)] = cv;
    }
    __syncthreads();

    if (tid < n) {
        const int t = tid;
        float *x = res + t * LDN;
        const float qmax = (float)((1 << bits) - 1);
        const size_t sbase = ((size_t)p * n + t) * stages;
        for (int s = 0; s < stages; ++s) {
            uint8_t *cdst = codes_out + (sbase + s) * n;
            if (nl <= 0) {
                for (int j = 0; j < n; ++j) cdst[j] = 0;
                scale_out[sbase + s] = 0.f;
                zp_out[sbase + s] = 0.f;
                rnorm_out[sbase + s] = 0.f;
                continue;
            }
            double ss = 0.0;
            float mn = x[0], mx = x[0];
            for (int j = 0; j < nl; ++j) {
                ss += (double)x[j] * (double)x[j];
                mn = f_min_nan(mn, x[j]);
                mx = f_max_nan(mx, x[j]);
            }
            // rtvq.py:17-18: python-int / Tensor == Tensor.reciprocal() * int -> two roundings
            const float range = __fsub_rn(mx, mn);
            const float recip = __fdiv_rn(1.0f, range);
            const float scale = __fmul_rn(recip, qmax);
            const float zp = __fmul_rn(-1.0f, rintf(__fmul_rn(scale, mn)));
            for (int j = 0; j < nl; ++j) {
                float vq = rintf(__fadd_rn(__fmul_rn(scale, x[j]), zp));
                uint8_t q;
                if (vq != vq) {
                    q = 0;
                } else {
                    vq = vq < 0.f ? 0.f : (vq > qmax ? qmax : vq);
                    q = (uint8_t)vq;
                }
                cdst[j] = q;
                const float deq = __fdiv_rn(__fsub_rn((float)q, zp), scale);
                x[j] = __fsub_rn(x[j], deq);
            }
            for (int j = nl; j < n; ++j) cdst[j] = 0;
            scale_out[sbase + s] = scale;
            zp_out[sbase + s] = zp;
            rnorm_out[sbase + s] = (float)sqrt(ss);
        }
    }
}

// ------------------------------------------------------------------------------------ launchers
int svdq_launch_eig(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, const double *gram_part, float *W,
                    double *c0, uint8_t *small,
                    int param0, int nparams, hipStream_t st) {
    const svdq_small_layout &L = pl->small;
    hipLaunchKernelGGL(k_eig, dim3(nparams), dim3(EIG_THREADS), 0, st, pl->d_params,
                       reinterpret_cast<const float *const *>(ptrs), rows_dev, pl->n_tasks,
                       pl->pack, pl->cfg.center, pl->cfg.energy_threshold, pl->cfg.max_rank, gram_part, W, c0, param0,
                       reinterpret_cast<float *>(small + L.sigma_off), reinterpret_cast<int32_t *>(small + L.k_off),
                       reinterpret_cast<int32_t *>(small + L.r_off), reinterpret_cast<float *>(small + L.energy_off),
                       reinterpret_cast<int64_t *>(small + L.rows_off));
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

int svdq_launch_coeff(const svdq_plan *pl, const double *cpart, const double *c0, uint8_t *small, int param0,
                      int nparams, hipStream_t st) {
    const svdq_small_layout &L = pl->small;
    hipLaunchKernelGGL(k_coeff, dim3(nparams), dim3(EIG_THREADS), 0, st, pl->d_params, pl->n_tasks, pl->pack,
                       pl->cfg.low_bits, pl->cfg.rtvq_stages, cpart, c0, reinterpret_cast<const int32_t *>(small + L.k_off),
                       reinterpret_cast<const int32_t *>(small + L.r_off), reinterpret_cast<float *>(small + L.coef_off),
                       reinterpret_cast<uint16_t *>(small + L.chigh_off), small + L.codes_off,
                       reinterpret_cast<float *>(small + L.scale_off), reinterpret_cast<float *>(small + L.zp_off),
                       reinterpret_cast<float *>(small + L.rnorm_off), param0);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

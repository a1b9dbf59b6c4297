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

// First-level reduction, spread over the chip: chunk c of parameter p sums its share of the unit
// slots (fixed order inside the chunk) into part2[(p*SVDQ_RC + c)][nn]; k_eig / k_coeff then only add
// SVDQ_RC partials per parameter.  Keeps the whole reduction deterministic and off one CU.
__global__ __launch_bounds__(EIG_THREADS) void k_reduce(const SvdqParam *__restrict__ params, int NT, int pack,
                                                        const double *__restrict__ part, double *__restrict__ part2,
                                                        int param0) {
    __shared__ double red[4 * 1024];
    __shared__ double out[1024];
    const int p = param0 + blockIdx.x, c = blockIdx.y, nn = NT * NT;
    const SvdqParam pd = params[p];
    const int s0 = pd.unit_begin * pack, ns = pd.unit_count * pack;
    const int per = (ns + SVDQ_RC - 1) / SVDQ_RC;
    int a = s0 + c * per, b = a + per;
    if (b > s0 + ns) b = s0 + ns;
    if (a > b) a = b;
    reduce_partials(part, a, b, nn, red, out);
    for (int e = threadIdx.x; e < nn; e += EIG_THREADS) part2[((size_t)p * SVDQ_RC + c) * nn + e] = out[e];
}

int svdq_launch_reduce(const svdq_plan *pl, const double *part, double *part2, int param0, int nparams,
                       hipStream_t st) {
    hipLaunchKernelGGL(k_reduce, dim3(nparams, SVDQ_RC), dim3(EIG_THREADS), 0, st, pl->d_params, pl->n_tasks, pl->pack,
                       part, part2, param0);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// Rotation (c, s) that annihilates a_pq.  The ANGLE only has to be good enough to make the sweep
// converge (it is seeded in fp32: one v_rcp/v_sqrt instead of two fp64 divides and two fp64 square
// roots on the critical path of every round); ORTHOGONALITY must hold to fp64, so c = (1+t^2)^-1/2 is
// refined by Newton steps in fp64 and s = t c.
__device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double &cs, double &sn) {
    cs = 1.0;
    sn = 0.0;
    if (apq == 0.0) return;
    const float o = (float)(2.0 * apq);
    const float d = (float)(aqq - app);
    double td;
    if (o != 0.f && fabsf(d) < 3.0e38f) {
        // 1-ulp hardware approximations are plenty for the angle
        const float tau = d * __builtin_amdgcn_rcpf(o);
        const float at = fabsf(tau);
        const float t = (at > 1.0e18f) ? 0.5f * __builtin_amdgcn_rcpf(at)
                                       : __builtin_amdgcn_rcpf(at + __builtin_amdgcn_sqrtf(1.0f + at * at));
        td = (double)(tau >= 0.f ? t : -t);
    } else {  // fp32 under/overflow of the operands: the slow exact path (rare)
        const double tau = (aqq - app) / (2.0 * apq);
        td = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    }
    const double x = 1.0 + td * td;  // in [1, 2]
    double r = (double)__builtin_amdgcn_rsqf((float)x);  // ~1e-7; two Newton steps -> fp64
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    cs = r;
    sn = td * r;
}

// One-wavefront workgroups need no s_barrier: LDS operations of a wave execute in order, so only the
// compiler has to be kept from moving accesses across the phase boundary.
template <int THREADS>
__device__ __forceinline__ void phase_sync() {
    if constexpr (THREADS == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// THREADS = 64 for n <= 8 (one wavefront: barriers cost nothing), 256 otherwise.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_eig(const SvdqParam *__restrict__ params,
                                                 const float *const *__restrict__ ptrs,
                                                 const int64_t *__restrict__ rows_dev, int NT, int center, float thr,
                                                 int max_rank, const double *__restrict__ gram_part2,
                                                 float *__restrict__ Wtab, double *__restrict__ c0_out, int param0,
                                                 float *__restrict__ sigma_out, int32_t *__restrict__ k_out,
                                                 int32_t *__restrict__ r_out, float *__restrict__ energy_out,
                                                 int64_t *__restrict__ rows_out) {
    __shared__ double Gd[1024];        // (deflated) Gram, kept for the completion column
    __shared__ double A[32 * LDN];     // working matrix, later W in fp64
    __shared__ double V[32 * LDN];
    __shared__ double lam[32], rowoff[32], rowdg[32], sgn[32], sig[32], xc0[32], u0[32];
    __shared__ int order[32];
    __shared__ int s_i0;

    const int p = param0 + blockIdx.x, tid = threadIdx.x, n = NT, nn = NT * NT;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;

    // fixed-order sum of the SVDQ_RC level-2 partials
    for (int e = tid; e < nn; e += THREADS) {
        const double *src = gram_part2 + (size_t)p * SVDQ_RC * nn + e;
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < SVDQ_RC; ++c) a += src[(size_t)c * nn];
        Gd[e] = a;
    }
    __syncthreads();

    // Centred rows sum to zero, so 1/sqrt(N) is an exact null vector of Tc.  The fp32-product Gram only
    // resolves sigma down to ~1e-4 sigma_0, so deflate that direction explicitly in fp64:
    // G <- C G C, C = I - 11^T/N.  (LAPACK reports ~1e-7 sigma_0 noise there; we report ~0.)
    if (center) {
        if (tid < n) {
            double sm = 0.0;
            for (int j = 0; j < n; ++j) sm += 0.5 * (Gd[tid * n + j] + Gd[j * n + tid]);
            rowoff[tid] = sm / n;
        }
        __syncthreads();
        double tot = 0.0;
        for (int j = 0; j < n; ++j) tot += rowoff[j];
        tot /= n;
        for (int e = tid; e < nn; e += THREADS) {
            const int i = e / n, j = e % n;
            A[i * LDN + j] = 0.5 * (Gd[i * n + j] + Gd[j * n + i]) - rowoff[i] - rowoff[j] + tot;
        }
    } else {
        for (int e = tid; e < nn; e += THREADS) {
            const int i = e / n, j = e % n;
            A[i * LDN + j] = 0.5 * (Gd[i * n + j] + Gd[j * n + i]);
        }
    }
    __syncthreads();
    for (int e = tid; e < nn; e += THREADS) {
        const int i = e / n, j = e % n;
        Gd[e] = A[i * LDN + j];
        V[i * LDN + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();

    // Parallel-order cyclic Jacobi: a round-robin tournament pairs all indices into M = ceil(n/2)
    // disjoint (p,q) per round (ne - 1 rounds per sweep); the M rotations of a round commute, so they
    // are applied together: A <- A J (columns), then A <- J^T A (rows), V <- V J.
    const int M = (n + 1) >> 1, ne = 2 * M;
    // fixed work assignment: item e = (idx, m) -> thread e % THREADS; n*M <= 512, so <= 2 items/thread
    // for THREADS = 256 and exactly <= 1 for THREADS = 64 (n <= 8).  No division inside the sweeps.
    constexpr int ITEMS = (THREADS == 64) ? 1 : 2;
    int it_idx[ITEMS], it_m[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const int e = tid + u * THREADS;
        it_idx[u] = (e < n * M) ? e / M : -1;
        it_m[u] = (e < n * M) ? e % M : 0;
    }
    for (int sweep = 0; sweep < 40 && n >= 2; ++sweep) {
        if (tid < n) {
            double off = 0.0;
            for (int j = 0; j < n; ++j) {
                const double a = A[tid * LDN + j];
                off += (j != tid) ? a * a : 0.0;
            }
            rowoff[tid] = off;
            rowdg[tid] = A[tid * LDN + tid] * A[tid * LDN + tid];
        }
        phase_sync<THREADS>();
        double off = 0.0, dg = 0.0;  // every thread adds the n row sums in the same order: uniform decision
        for (int j = 0; j < n; ++j) {
            off += rowoff[j];
            dg += rowdg[j];
        }
        if (off <= 1e-30 * dg || dg == 0.0) break;
        for (int rd = 0; rd < ne - 1; ++rd) {
            // every thread derives the rotation of ITS pair itself (same inputs -> same bits in all of
            // the pair's threads) and keeps (c, s) in registers for the column and the row phase
            int pp[ITEMS], qq[ITEMS];
            double cs[ITEMS], sn[ITEMS];
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) {
                const int m = it_m[u];
                int a2, b2;
                if (m == 0) {
                    a2 = ne - 1;
                    b2 = rd;
                } else {
                    a2 = rd + m;
                    if (a2 >= ne - 1) a2 -= ne - 1;
                    b2 = rd - m;
                    if (b2 < 0) b2 += ne - 1;
                }
                pp[u] = a2 < b2 ? a2 : b2;
                qq[u] = a2 < b2 ? b2 : a2;
                cs[u] = 1.0;
                sn[u] = 0.0;
                if (it_idx[u] >= 0 && qq[u] < n)
                    jacobi_cs(A[pp[u] * LDN + pp[u]], A[qq[u] * LDN + qq[u]], A[pp[u] * LDN + qq[u]], cs[u], sn[u]);
            }
            phase_sync<THREADS>();  // everybody has read the 2x2 blocks before anybody rotates
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) {  // columns of A and V: A <- A J, V <- V J
                const int i = it_idx[u];
                if (i >= 0 && qq[u] < n) {
                    const double x = A[i * LDN + pp[u]], y = A[i * LDN + qq[u]];
                    A[i * LDN + pp[u]] = cs[u] * x - sn[u] * y;
                    A[i * LDN + qq[u]] = sn[u] * x + cs[u] * y;
                    const double vx = V[i * LDN + pp[u]], vy = V[i * LDN + qq[u]];
                    V[i * LDN + pp[u]] = cs[u] * vx - sn[u] * vy;
                    V[i * LDN + qq[u]] = sn[u] * vx + cs[u] * vy;
                }
            }
            phase_sync<THREADS>();
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) {  // rows of A: A <- J^T A
                const int j = it_idx[u];
                if (j >= 0 && qq[u] < n) {
                    const double x = A[pp[u] * LDN + j], y = A[qq[u] * LDN + j];
                    A[pp[u] * LDN + j] = cs[u] * x - sn[u] * y;
                    A[qq[u] * LDN + j] = sn[u] * x + cs[u] * y;
                }
            }
            phase_sync<THREADS>();
        }
    }
    __syncthreads();

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
        // first direction below the fp32 resolution of the data, if any
        int i0 = -1;
        if (sig[0] > 0.0 && D > 0)
            for (int i = 0; i < r; ++i)
                if (!(sig[i] > 1e-6 * sig[0])) {
                    i0 = i;
                    break;
                }
        s_i0 = i0;
    }
    // W[t][i] = sgn_i V[t][order[i]] / sigma_i for the resolved directions, 0 otherwise (A is free now)
    const double s0 = sig[0];
    for (int e = tid; e < nn; e += THREADS) {
        const int t = e / n, i = e % n;
        double wv = 0.0;
        if (i < r && sig[i] > 1e-6 * s0 && sig[i] > 0.0) wv = sgn[i] * V[t * LDN + order[i]] / sig[i];
        A[t * LDN + i] = wv;
    }
    if (tid < n && D > 0) lam[tid] = (double)ptrs[(size_t)p * n + tid][0];  // row 0 of every task
    __syncthreads();

    // Orthonormal completion of the first null direction (the one centring always creates; LAPACK
    // returns an arbitrary orthonormal vector there): u = (e_0 - U U[0,:]^T) / norm, i.e. one more W
    // column w[t] = -(sum_j W[t][j] U[0][j]) / norm plus a spike 1/norm at row 0 (added in pass 2).
    // Any further null directions stay zero columns ("null directions").
    const int i0 = s_i0;
    double spike = 0.0;
    if (i0 >= 0) {  // uniform
        // centre row 0 exactly as the streaming kernels do (fp32, task order, one divide)
        float sum = 0.f;
        for (int t = 0; t < n; ++t) sum += (float)lam[t];
        const float mean0 = center ? sum / (float)n : 0.f;
        if (tid < n) xc0[tid] = (double)((float)lam[tid] - mean0);
        __syncthreads();
        if (tid < r) {
            double u = 0.0;
            for (int t = 0; t < n; ++t) u += xc0[t] * A[t * LDN + tid];
            u0[tid] = u;  // U[0][tid]
        }
        __syncthreads();
        double norm2 = 1.0;
        for (int j = 0; j < r; ++j) norm2 -= u0[j] * u0[j];
        if (norm2 > 0.25) {  // uniform
            spike = 1.0 / sqrt(norm2);
            double acc = 0.0;
            if (tid < n)
                for (int j = 0; j < r; ++j) acc += A[tid * LDN + j] * u0[j];
            __syncthreads();
            if (tid < n) A[tid * LDN + i0] = -spike * acc;
        }
    }
    __syncthreads();
    const bool have_col = spike != 0.0;
    if (tid == 0) {
        float *aux = Wtab + (size_t)p * (nn + 4) + nn;
        aux[0] = (float)spike;
        aux[1] = have_col ? (float)i0 : -1.f;
        aux[2] = 0.f;
        aux[3] = 0.f;
    }
    // W (fp32) and the closed-form coefficients c0[t][i] = u_i^T xc_t of the UNROUNDED basis:
    //   resolved direction:  sigma_i * v_i[t]           (U^T Tc = Sigma V^T)
    //   completion column:   w^T Gd[:,t] + spike * xc_t[row 0]
    //   zero column:         0
    // pass 2 adds the fp16-rounding correction E^T Tc on top (k_coeff sums both).
    for (int e = tid; e < nn; e += THREADS) {
        const int t = e / n, i = e % n;
        Wtab[(size_t)p * (nn + 4) + e] = (float)A[t * LDN + i];
        double cv = 0.0;
        if (i < r) {
            if (have_col && i == i0) {
                for (int t2 = 0; t2 < n; ++t2) cv += A[t2 * LDN + i] * Gd[t2 * n + t];
                cv += spike * xc0[t];
            } else if (sig[i] > 1e-6 * s0 && sig[i] > 0.0) {
                cv = sig[i] * sgn[i] * V[t * LDN + order[i]];
            }
        }
        c0_out[(size_t)p * nn + e] = cv;
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
    (void)pack;
    (void)pd;
    reduce_partials(cpart, p * SVDQ_RC, (p + 1) * SVDQ_RC, n * n, red, C);  // level-2 partials of k_reduce

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
int svdq_launch_eig(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, const double *gram_part2, float *W,
                    double *c0, uint8_t *small, int param0, int nparams, hipStream_t st) {
    const svdq_small_layout &L = pl->small;
    auto pp = reinterpret_cast<const float *const *>(ptrs);
    float *sg = reinterpret_cast<float *>(small + L.sigma_off);
    int32_t *kk = reinterpret_cast<int32_t *>(small + L.k_off), *rr = reinterpret_cast<int32_t *>(small + L.r_off);
    float *en = reinterpret_cast<float *>(small + L.energy_off);
    int64_t *ro = reinterpret_cast<int64_t *>(small + L.rows_off);
    if (pl->n_tasks <= 8)
        hipLaunchKernelGGL(k_eig<64>, dim3(nparams), dim3(64), 0, st, pl->d_params, pp, rows_dev, pl->n_tasks,
                           pl->cfg.center, pl->cfg.energy_threshold, pl->cfg.max_rank, gram_part2, W, c0, param0, sg,
                           kk, rr, en, ro);
    else
        hipLaunchKernelGGL(k_eig<256>, dim3(nparams), dim3(256), 0, st, pl->d_params, pp, rows_dev, pl->n_tasks,
                           pl->cfg.center, pl->cfg.energy_threshold, pl->cfg.max_rank, gram_part2, W, c0, param0, sg,
                           kk, rr, en, ro);
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

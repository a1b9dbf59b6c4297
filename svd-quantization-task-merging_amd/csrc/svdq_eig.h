// svdq_eig.h -- the per-parameter eigen-stage (fixed-order sum of the level-2 Gram partials, null-vector
// deflation, parallel-order Jacobi in fp64, the reference's fp32 rank rule, W = V Sigma^-1, orthonormal
// completion column, closed-form coefficients) as a device function shared by k_eig and the fused kernel.
#pragma once
#include "svdq_common.h"

#define LDN 33  // padded leading dimension of N x N LDS matrices sized for N <= 32 (k_coeff)

// phase stamps for tools/probe/eig_time.hip (a diagnostic build defines EIG_STAMP; nothing is emitted otherwise)
#ifndef EIG_STAMP
#define EIG_STAMP(i)
#endif

// Rotation (c, s) that annihilates a_pq.  The ANGLE only has to be good enough to make the sweep
// converge (it is seeded in fp32: one v_rcp/v_sqrt instead of two fp64 divides and two fp64 square
// roots on the critical path of every round); ORTHOGONALITY must hold to fp64, so c = (1+t^2)^-1/2 is
// refined by Newton steps in fp64 and s = t c.
__device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double &cs, double &sn) {
#pragma clang fp contract(off)  // same bits whichever translation unit (and -ffp-contract) this is inlined into
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
// LDS scratch eig_param<THREADS, NMAX> needs (bytes), for the caller that provides it
#define SVDQ_EIG_LDS_BYTES(NMAX) (((NMAX) * (NMAX) + 2 * (NMAX) * ((NMAX) + 1) + 7 * (NMAX)) * 8 + ((NMAX) + 1) * 4)

// Canonical fixed-order sum of partial slots [a, b) for entry e: eight interleaved accumulators, then a
// fixed tree.  k_reduce and the fused kernel both use it, so the two schedules give identical bits.
__device__ __forceinline__ double chunk_sum(const double *__restrict__ part, int a, int b, int nn, int e) {
#pragma clang fp contract(off)
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    int s = a;
    for (; s + 8 <= b; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(s + u) * nn + e];
    }
    for (; s < b; ++s) acc[0] += part[(size_t)s * nn + e];
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

// The whole per-parameter eigen-stage as a device function over caller-provided LDS scratch, so that
// the stand-alone kernel (k_eig) and the persistent fused kernel share it.  THREADS threads of ONE
// workgroup must call it together (tid = thread index in [0, THREADS)).
template <int THREADS, int NMAX>
__device__ void eig_param(double *__restrict__ lds, int p, int tid, int64_t D,
                          const float *const *__restrict__ ptrs, int NT, int center, float thr, int max_rank,
                          const double *__restrict__ gram_part2, float *__restrict__ Wtab,
                          double *__restrict__ c0_out, float *__restrict__ sigma_out, int32_t *__restrict__ k_out,
                          int32_t *__restrict__ r_out, float *__restrict__ energy_out,
                          int64_t *__restrict__ rows_out, int64_t row0_pos = 0,
                          const float *const *__restrict__ base_ptrs = nullptr,
                          int32_t *__restrict__ refine_out = nullptr, double resolve = 1e-6) {
#pragma clang fp contract(off)
    constexpr int LDX = NMAX + 1;      // padded leading dimension
    double *Gd = lds;                  // [NMAX*NMAX] (deflated) Gram, kept for the completion column
    double *A = Gd + NMAX * NMAX;      // [NMAX*LDX] working matrix, later W in fp64
    double *V = A + NMAX * LDX;
    double *lam = V + NMAX * LDX, *rowoff = lam + NMAX, *rowdg = rowoff + NMAX, *sgn = rowdg + NMAX,
           *sig = sgn + NMAX, *xc0 = sig + NMAX, *u0 = xc0 + NMAX;
    int *order = reinterpret_cast<int *>(u0 + NMAX);
    int &s_i0 = order[NMAX];

    const int n = NT, nn = NT * NT;
    EIG_STAMP(0);

    // fixed-order sum of the SVDQ_RC level-2 partials
    for (int e = tid; e < nn; e += THREADS) {
        const double *src = gram_part2 + (size_t)p * SVDQ_RC * nn + e;
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < SVDQ_RC; ++c) a += src[(size_t)c * nn];
        Gd[e] = a;
    }
    phase_sync<THREADS>();
    EIG_STAMP(1);

    // Centred rows sum to zero, so 1/sqrt(N) is an exact null vector of Tc.  Deflate that direction explicitly in fp64
    // (the fp32 rounding of the centred rows leaves ~1e-7 sigma_0 of it in the data):
    // G <- C G C, C = I - 11^T/N.  (LAPACK reports ~1e-7 sigma_0 noise there; we report ~0.)
    if (center) {
        if (tid < n) {
            double sm = 0.0;
            for (int j = 0; j < n; ++j) sm += 0.5 * (Gd[tid * n + j] + Gd[j * n + tid]);
            rowoff[tid] = sm / n;
        }
        phase_sync<THREADS>();
        double tot = 0.0;
        for (int j = 0; j < n; ++j) tot += rowoff[j];
        tot /= n;
        for (int e = tid; e < nn; e += THREADS) {
            const int i = e / n, j = e % n;
            A[i * LDX + j] = 0.5 * (Gd[i * n + j] + Gd[j * n + i]) - rowoff[i] - rowoff[j] + tot;
        }
    } else {
        for (int e = tid; e < nn; e += THREADS) {
            const int i = e / n, j = e % n;
            A[i * LDX + j] = 0.5 * (Gd[i * n + j] + Gd[j * n + i]);
        }
    }
    phase_sync<THREADS>();
    for (int e = tid; e < nn; e += THREADS) {
        const int i = e / n, j = e % n;
        Gd[e] = A[i * LDX + j];
        V[i * LDX + j] = (i == j) ? 1.0 : 0.0;
    }
    phase_sync<THREADS>();

    EIG_STAMP(2);
    // Parallel-order cyclic Jacobi: a round-robin tournament pairs all indices into M = ceil(n/2)
    // disjoint (p,q) per round (ne - 1 rounds per sweep); the M rotations of a round commute, so they
    // are applied together: A <- A J (columns), then A <- J^T A (rows), V <- V J.
    const int M = (n + 1) >> 1, ne = 2 * M;
    // fixed work assignment: item e = (idx, m) -> thread e % THREADS; n*M <= 512, so <= 2 items/thread
    // for THREADS = 256 and exactly <= 1 for THREADS = 64 (n <= 8).  No division inside the sweeps.
    constexpr int ITEMS = (NMAX * ((NMAX + 1) / 2) + THREADS - 1) / THREADS;
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
                const double a = A[tid * LDX + j];
                off += (j != tid) ? a * a : 0.0;
            }
            rowoff[tid] = off;
            rowdg[tid] = A[tid * LDX + tid] * A[tid * LDX + tid];
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
                    jacobi_cs(A[pp[u] * LDX + pp[u]], A[qq[u] * LDX + qq[u]], A[pp[u] * LDX + qq[u]], cs[u], sn[u]);
            }
            phase_sync<THREADS>();  // everybody has read the 2x2 blocks before anybody rotates
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) {  // columns of A and V: A <- A J, V <- V J
                const int i = it_idx[u];
                if (i >= 0 && qq[u] < n) {
                    const double x = A[i * LDX + pp[u]], y = A[i * LDX + qq[u]];
                    A[i * LDX + pp[u]] = cs[u] * x - sn[u] * y;
                    A[i * LDX + qq[u]] = sn[u] * x + cs[u] * y;
                    const double vx = V[i * LDX + pp[u]], vy = V[i * LDX + qq[u]];
                    V[i * LDX + pp[u]] = cs[u] * vx - sn[u] * vy;
                    V[i * LDX + qq[u]] = sn[u] * vx + cs[u] * vy;
                }
            }
            phase_sync<THREADS>();
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) {  // rows of A: A <- J^T A
                const int j = it_idx[u];
                if (j >= 0 && qq[u] < n) {
                    const double x = A[pp[u] * LDX + j], y = A[qq[u] * LDX + j];
                    A[pp[u] * LDX + j] = cs[u] * x - sn[u] * y;
                    A[qq[u] * LDX + j] = sn[u] * x + cs[u] * y;
                }
            }
            phase_sync<THREADS>();
        }
    }
    phase_sync<THREADS>();
    EIG_STAMP(3);

    // sort descending (stable on ties), sign convention: largest-|v| component positive
    if (tid < n) lam[tid] = A[tid * LDX + tid];
    phase_sync<THREADS>();
    if (tid < n) {
        int rank = 0;
        for (int i = 0; i < n; ++i)
            if (lam[i] > lam[tid] || (lam[i] == lam[tid] && i < tid)) ++rank;
        order[rank] = tid;
    }
    phase_sync<THREADS>();
    if (tid < n) {
        const int col = order[tid];
        const double l = lam[col];
        sig[tid] = l > 0.0 ? sqrt(l) : 0.0;
        double best = 0.0, bv = 1.0;
        for (int j = 0; j < n; ++j) {
            const double x = V[j * LDX + col];
            if (fabs(x) > best) {
                best = fabs(x);
                bv = x;
            }
        }
        sgn[tid] = bv < 0.0 ? -1.0 : 1.0;
    }
    phase_sync<THREADS>();

    const int r = (int)(D < (int64_t)n ? D : (int64_t)n);
    EIG_STAMP(4);
    if (tid == 0) {
        // basis.py:147-156 and :199-211, fp32 like the reference (threshold compared as fp32)
        float S[NMAX], cum[NMAX];
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
        // first direction below the resolution of this Gram (exact products: the fp32 resolution of the data), if any
        int i0 = -1;
        if (sig[0] > 0.0 && D > 0)
            for (int i = 0; i < r; ++i)
                if (!(sig[i] > resolve * sig[0])) {
                    i0 = i;
                    break;
                }
        s_i0 = i0;
        // N > 16 accumulates fp32 products first: a sigma between the noise floor of the explicitly deflated null
        // direction (~1e-8 sigma_0) and 2e-2 sigma_0 may be off by more than rtol 2e-5 -> ask for the fp64 pass
        // ... and so may one the fp32-product sums measured as NEGATIVE (clipped to sigma = 0 above): a true lambda up
        // to ~1e-7 lambda_0 can come out below zero, and would then pass for a null direction.  A centred stack of
        // D >= n rows has exactly one null direction (the deflated one), anything else none: every surplus
        // direction at or below the noise floor asks for the fp64 pass too.
        if (refine_out) {
            int need = 0, nulls = 0;
            for (int i = 1; i < r; ++i)
                if (sig[i] > 3e-7 * sig[0] && sig[i] < 2e-2 * sig[0]) need = 1;
            for (int i = 0; i < r; ++i)
                if (!(sig[i] > 3e-7 * sig[0])) ++nulls;
            if (nulls > ((center && D >= (int64_t)n) ? 1 : 0) && sig[0] > 0.0) need = 1;
            refine_out[p] = need;
        }
    }
    EIG_STAMP(5);
    // W[t][i] = sgn_i V[t][order[i]] / sigma_i for the resolved directions, 0 otherwise (A is free now)
    const double s0 = sig[0];
    for (int e = tid; e < nn; e += THREADS) {
        const int t = e / n, i = e % n;
        double wv = 0.0;
        if (i < r && sig[i] > resolve * s0 && sig[i] > 0.0) wv = sgn[i] * V[t * LDX + order[i]] / sig[i];
        A[t * LDX + i] = wv;
    }
    // row 0 of every task (gather / walk mode: the first selected element of the tensor, row0_pos)
    if (tid < n && D > 0) {
        const int64_t i0 = row0_pos;
        float x0 = ptrs[(size_t)p * n + tid][i0];
        if (base_ptrs) x0 = x0 - base_ptrs[p][i0];   // minus-base mode: the delta is formed exactly as in the passes
        lam[tid] = (double)x0;
    }
    phase_sync<THREADS>();
    EIG_STAMP(6);

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
        phase_sync<THREADS>();
        if (tid < r) {
            double u = 0.0;
            for (int t = 0; t < n; ++t) u += xc0[t] * A[t * LDX + tid];
            u0[tid] = u;  // U[0][tid]
        }
        phase_sync<THREADS>();
        double norm2 = 1.0;
        for (int j = 0; j < r; ++j) norm2 -= u0[j] * u0[j];
        if (norm2 > 0.25) {  // uniform
            spike = 1.0 / sqrt(norm2);
            double acc = 0.0;
            if (tid < n)
                for (int j = 0; j < r; ++j) acc += A[tid * LDX + j] * u0[j];
            phase_sync<THREADS>();
            if (tid < n) A[tid * LDX + i0] = -spike * acc;
        }
    }
    phase_sync<THREADS>();
    EIG_STAMP(7);
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
        Wtab[(size_t)p * (nn + 4) + e] = (float)A[t * LDX + i];
        double cv = 0.0;
        if (i < r) {
            if (have_col && i == i0) {
                for (int t2 = 0; t2 < n; ++t2) cv += A[t2 * LDX + i] * Gd[t2 * n + t];
                cv += spike * xc0[t];
            } else if (sig[i] > resolve * s0 && sig[i] > 0.0) {
                cv = sig[i] * sgn[i] * V[t * LDX + order[i]];
            }
        }
        c0_out[(size_t)p * nn + e] = cv;
    }
    EIG_STAMP(8);
}


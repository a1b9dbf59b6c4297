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
#include "svdq_eig.h"
#include <hip/hip_fp16.h>

#define EIG_THREADS 256

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

// First-level reduction, spread over the chip: chunk c of parameter p sums its share of the unit slots into
// part2[(p*SVDQ_RC + c)][nn]; k_eig / k_coeff then only add SVDQ_RC partials per parameter.  Keeps the whole reduction
// deterministic and off one CU.  Latency-bound (a chunk of a 4 M-row tensor is 128 slots of N x N doubles, each thread
// adds its entry of every slot): 256 threads = G groups of >= nn threads; group g takes slots a + g, a + g + G, ...
// with eight loads in flight per thread, the G group sums meet in LDS in group order.  Fixed order, so the same bits
// in every run and every batch; 4 dependent memory round trips per chunk where the 64-thread version had 16.
#define RED_THREADS 256
__global__ __launch_bounds__(RED_THREADS) void k_reduce(const SvdqParam *__restrict__ params, int NT, int pack,
                                                        const double *__restrict__ part, double *__restrict__ part2,
                                                        int param0, const int32_t *__restrict__ only) {
#pragma clang fp contract(off)
    __shared__ double red[RED_THREADS];
    const int p = param0 + blockIdx.x, c = blockIdx.y, nn = NT * NT;
    if (only && !only[p]) return;
    const SvdqParam pd = params[p];
    const int per = (pd.unit_count + SVDQ_RC - 1) / SVDQ_RC;  // units per chunk
    int ua = c * per, ub = ua + per;
    if (ub > pd.unit_count) ub = pd.unit_count;
    if (ua > ub) ua = ub;
    const int a = (pd.unit_begin + ua) * pack, b = (pd.unit_begin + ub) * pack;
    const int width = nn <= 64 ? 64 : (nn <= 128 ? 128 : RED_THREADS);   // threads per group
    const int G = RED_THREADS / width;
    const int g = threadIdx.x / width, l = threadIdx.x % width;
    double *dst = part2 + ((size_t)p * SVDQ_RC + c) * nn;
    for (int e0 = 0; e0 < nn; e0 += width) {      // one round unless nn > 256
        const int e = e0 + l;
        double acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.0;
        if (e < nn) {
            int s = a + g;
            for (; s + 7 * G < b; s += 8 * G) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(s + u * G) * nn + e];
            }
            for (; s < b; s += G) acc[0] += part[(size_t)s * nn + e];
        }
        const double mine = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        if (G == 1) {
            if (e < nn) dst[e] = mine;
        } else {
            red[threadIdx.x] = mine;
            __syncthreads();
            if (g == 0 && e < nn) {
                double t = red[l];
                for (int j = 1; j < G; ++j) t += red[j * width + l];
                dst[e] = t;
            }
            __syncthreads();
        }
    }
}

int svdq_launch_reduce(const svdq_plan *pl, const double *part, double *part2, int param0, int nparams,
                       const int32_t *only, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce, dim3(nparams, SVDQ_RC), dim3(RED_THREADS), 0, st, pl->d_params, pl->n_tasks, pl->pack,
                       part, part2, param0, only);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// Sum of the level-2 partials over all parameters and chunks, fixed order: 256 threads, thread j of an
// entry's group takes partials j, j+G, ... (chunk_sum order within), then a fixed tree over the group.
__global__ __launch_bounds__(256) void k_gram_total(int nparams, int nn, const double *__restrict__ part2,
                                                    double *__restrict__ out) {
    __shared__ double red[256];
    const int total = nparams * SVDQ_RC;
    for (int e0 = 0; e0 < nn; e0 += 16) {   // 16 entries x 16 partial-sums per pass
        const int e = e0 + (threadIdx.x & 15), j = threadIdx.x >> 4;
        const int per = (total + 15) / 16;
        int a = j * per, b = a + per;
        if (b > total) b = total;
        if (a > b) a = b;
        red[threadIdx.x] = e < nn ? chunk_sum(part2, a, b, nn, e) : 0.0;
        __syncthreads();
        for (int off = 128; off >= 16; off >>= 1) {
            if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x < 16 && e < nn) out[e] = red[threadIdx.x];
        __syncthreads();
    }
}

int svdq_launch_gram_total(const svdq_plan *pl, const double *part2, double *out, hipStream_t st) {
    hipLaunchKernelGGL(k_gram_total, dim3(1), dim3(256), 0, st, pl->n_params, pl->n_tasks * pl->n_tasks, part2, out);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

template <int THREADS, int NMAX>
__global__ __launch_bounds__(THREADS) void k_eig(const SvdqParam *__restrict__ params,
                                                 const float *const *__restrict__ ptrs,
                                                 const int64_t *__restrict__ rows_dev, int NT, int center, float thr,
                                                 int max_rank, const double *__restrict__ gram_part2,
                                                 float *__restrict__ Wtab, double *__restrict__ c0_out, int param0,
                                                 float *__restrict__ sigma_out, int32_t *__restrict__ k_out,
                                                 int32_t *__restrict__ r_out, float *__restrict__ energy_out,
                                                 int64_t *__restrict__ rows_out,
                                                 const int32_t *const *__restrict__ idx_ptrs,
                                                 const float *const *__restrict__ base_ptrs,
                                                 const int32_t *__restrict__ only, int32_t *__restrict__ refine_out,
                                                 float resolve, const int64_t *__restrict__ ustart) {
    __shared__ __attribute__((aligned(16))) double lds[SVDQ_EIG_LDS_BYTES(NMAX) / 8 + 1];
    const int p = param0 + blockIdx.x;
    if (only && !only[p]) return;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    // source position of the parameter's first row: 0, the first entry of its index list (gather mode), or the start
    // of its first work unit (walk mode; bit 62 is the mask polarity)
    int64_t row0 = 0;
    if (D > 0) {
        if (idx_ptrs) row0 = idx_ptrs[p][0];
        else if (ustart) row0 = ustart[params[p].unit_begin] & ((1ll << 62) - 1);
    }
    eig_param<THREADS, NMAX>(lds, p, threadIdx.x, D, ptrs, NT, center, thr, max_rank, gram_part2, Wtab, c0_out, sigma_out,
                             k_out, r_out, energy_out, rows_out, row0, base_ptrs, refine_out, (double)resolve);
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
                                                       float *__restrict__ zp_out, float *__restrict__ rnorm_out, int param0,
                                                       const int32_t *__restrict__ bits_tab) {
    __shared__ double red[4 * 1024];
    __shared__ double C[1024];
    __shared__ float res[32 * LDN];

    const int p = param0 + blockIdx.x, tid = threadIdx.x, n = NT;
    const SvdqParam pd = params[p];
    if (bits_tab) bits = bits_tab[p];   // per-parameter code width (config #5's mixed 8-bit / 2-bit run)
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
                    double *c0, uint8_t *small, int param0, int nparams, const void *idx, const void *base,
                    const int32_t *only, int32_t *refine_out, hipStream_t st, const int64_t *ustart) {
    if (ustart) idx = nullptr;   // walk mode: idx names mask bytes, not index lists
    // smallest sigma / sigma_0 the Gram behind gram_part2 resolves: exact-product (fp64 MFMA) sums reach the fp32
    // resolution of the data; fp32-product sums (cfg.reserved bit 1, and N > 16 before its refinement) do not
    const bool exact = (pl->ntp <= 16 && !(pl->cfg.reserved & 2)) || only != nullptr;
    const float resolve = exact ? 1e-6f : 3e-4f;
    auto ip = reinterpret_cast<const int32_t *const *>(idx);
    auto bpp = reinterpret_cast<const float *const *>(base);
    const svdq_small_layout &L = pl->small;
    auto pp = reinterpret_cast<const float *const *>(ptrs);
    float *sg = reinterpret_cast<float *>(small + L.sigma_off);
    int32_t *kk = reinterpret_cast<int32_t *>(small + L.k_off), *rr = reinterpret_cast<int32_t *>(small + L.r_off);
    float *en = reinterpret_cast<float *>(small + L.energy_off);
    int64_t *ro = reinterpret_cast<int64_t *>(small + L.rows_off);
    if (pl->n_tasks <= 8)
        hipLaunchKernelGGL((k_eig<64, 8>), dim3(nparams), dim3(64), 0, st, pl->d_params, pp, rows_dev, pl->n_tasks,
                           pl->cfg.center, pl->cfg.energy_threshold, pl->cfg.max_rank, gram_part2, W, c0, param0, sg,
                           kk, rr, en, ro, ip, bpp, only, refine_out, resolve, ustart);
    else
        hipLaunchKernelGGL((k_eig<256, 32>), dim3(nparams), dim3(256), 0, st, pl->d_params, pp, rows_dev, pl->n_tasks,
                           pl->cfg.center, pl->cfg.energy_threshold, pl->cfg.max_rank, gram_part2, W, c0, param0, sg,
                           kk, rr, en, ro, ip, bpp, only, refine_out, resolve, ustart);
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
                       reinterpret_cast<float *>(small + L.rnorm_off), param0, pl->d_bits);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// svdq_elem.hip -- element-wise / scan kernels: the standalone multi-stage quantizer on large
// tensors (RTVQQuantizer.quantize/dequantize, rtvq.py:39-139) and the mask operators
// (mask_loader.py:412-485, 651-709).  Compiled with -ffp-contract=off.
//
// Quantizer on n elements, S stages, all on device, no host round trip:
//   k_rtvq_stats   x -> per-block {min, max, has_nan, sum of squares}        (read 4n)
//   per stage s:
//     k_rtvq_params  block partials -> scale_s, zero_point_s, residual_norm_s (fixed order)
//     k_rtvq_apply   residual -> codes_s (1 B/elem), next residual, next stage's partials
//                                                                            (read 4n, write 5n)
// Algorithmic bytes n*(4+S); this schedule moves n*(4 + S*9 - 4) because stage s+1's min/max
// depends on stage s's scale.  A one-launch form for tensors that fit the register file (x read
// once, grid-wide meetings between the stages) was built and measured in round 4: 3-8x SLOWER -- a meeting of 10^3
// workgroups across eight XCDs costs 15-80 us against ~2 us for a kernel boundary (profiles/r04_rtvq_one_launch_experiment.txt).

#include "svdq_common.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>

#define ELT_THREADS 256
#define RTVQ_MAX_BLOCKS 2048

struct RtvqPartial {
    float mn, mx;
    int has_nan, pad;
    double ss;
};

typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void stat_acc(float x, float &mn, float &mx, int &has_nan, double &ss) {
    if (x != x) has_nan = 1;
    mn = x < mn ? x : mn;
    mx = x > mx ? x : mx;
    ss += (double)x * (double)x;
}

// block-level combine of per-thread stats, result written by thread 0
__device__ void stats_block_reduce(float mn, float mx, int has_nan, double ss, RtvqPartial *dst) {
    __shared__ float s_mn[ELT_THREADS], s_mx[ELT_THREADS];
    __shared__ int s_nan[ELT_THREADS];
    __shared__ double s_ss[ELT_THREADS];
    const int tid = threadIdx.x;
    s_mn[tid] = mn;
    s_mx[tid] = mx;
    s_nan[tid] = has_nan;
    s_ss[tid] = ss;
    __syncthreads();
    for (int off = ELT_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) {
            s_mn[tid] = s_mn[tid + off] < s_mn[tid] ? s_mn[tid + off] : s_mn[tid];
            s_mx[tid] = s_mx[tid + off] > s_mx[tid] ? s_mx[tid + off] : s_mx[tid];
            s_nan[tid] |= s_nan[tid + off];
            s_ss[tid] += s_ss[tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        dst->mn = s_mn[0];
        dst->mx = s_mx[0];
        dst->has_nan = s_nan[0];
        dst->pad = 0;
        dst->ss = s_ss[0];
    }
    __syncthreads();
}

__global__ __launch_bounds__(ELT_THREADS) void k_rtvq_stats(const float *__restrict__ x, int64_t n,
                                                            RtvqPartial *__restrict__ part) {
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * ELT_THREADS;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    int has_nan = 0;
    double ss = 0.0;
    int64_t i = (int64_t)blockIdx.x * ELT_THREADS + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {  // four independent 16-byte loads in flight per thread
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const f32x4 *>(x)[i + q * stride];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) stat_acc(v[q][e], mn, mx, has_nan, ss);
    }
    for (; i < nvec; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) stat_acc(v[e], mn, mx, has_nan, ss);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) stat_acc(x[(nvec << 2) + threadIdx.x], mn, mx, has_nan, ss);
    stats_block_reduce(mn, mx, has_nan, ss, part + blockIdx.x);
}

// rtvq.py:10-18 for one stage, from the block partials of the previous pass.  Every block of the apply
// kernel recomputes it (<= 2048 partials, L2-resident; fixed order => the same bits in every block and a
// deterministic norm), which removes a dependent single-block launch per stage.
__device__ void stage_params(const RtvqPartial *__restrict__ part, int nblk, int bits, float &scale, float &zp,
                             float &rnorm) {
    __shared__ RtvqPartial s_tot;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    int has_nan = 0;
    double ss = 0.0;
    for (int b = threadIdx.x; b < nblk; b += ELT_THREADS) {
        const RtvqPartial q = part[b];
        mn = q.mn < mn ? q.mn : mn;
        mx = q.mx > mx ? q.mx : mx;
        has_nan |= q.has_nan;
        ss += q.ss;
    }
    stats_block_reduce(mn, mx, has_nan, ss, &s_tot);
    mn = s_tot.mn;
    mx = s_tot.mx;
    if (s_tot.has_nan) {  // torch min/max propagate NaN
        mn = __builtin_nanf("");
        mx = mn;
    }
    const float qmax = (float)((1 << bits) - 1);
    // python-int / Tensor == Tensor.reciprocal() * int: two roundings (tests/golden/rtvq_cases.npz)
    scale = __fmul_rn(__fdiv_rn(1.0f, __fsub_rn(mx, mn)), qmax);
    zp = __fmul_rn(-1.0f, rintf(__fmul_rn(scale, mn)));
    rnorm = (float)sqrt(s_tot.ss);
    __syncthreads();  // s_tot and the reduction scratch are reused by the caller's own reduction
}

__device__ __forceinline__ unsigned char quant_one(float x, float scale, float zp, float qmax, float &res) {
    float vq = rintf(__fadd_rn(__fmul_rn(scale, x), zp));   // rtvq.py:20, two roundings
    unsigned char q;
    if (vq != vq) {
        q = 0;                                               // NaN -> code 0 (reference CPU cast)
    } else {
        vq = vq < 0.f ? 0.f : (vq > qmax ? qmax : vq);
        q = (unsigned char)vq;
    }
    const float deq = __fdiv_rn(__fsub_rn((float)q, zp), scale);   // rtvq.py:35
    res = __fsub_rn(x, deq);                                       // rtvq.py:65
    return q;
}

template <bool LAST>
__global__ __launch_bounds__(ELT_THREADS) void k_rtvq_apply(const float *__restrict__ rin, float *__restrict__ rout,
                                                            int64_t n, const RtvqPartial *__restrict__ part_in,
                                                            int nblk_in, int stage, int bits,
                                                            uint8_t *__restrict__ codes,
                                                            RtvqPartial *__restrict__ part_out,
                                                            float *__restrict__ scale_out, float *__restrict__ zp_out,
                                                            float *__restrict__ rnorm_out) {
    float scale, zp, rnorm;
    stage_params(part_in, nblk_in, bits, scale, zp, rnorm);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scale_out[stage] = scale;
        zp_out[stage] = zp;
        rnorm_out[stage] = rnorm;
    }
    const float qmax = (float)((1 << bits) - 1);
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * ELT_THREADS;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    int has_nan = 0;
    double ss = 0.0;
    int64_t i = (int64_t)blockIdx.x * ELT_THREADS + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const f32x4 *>(rin)[i + q * stride];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 r;
            u8x4 c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float re;
                c[e] = quant_one(v[q][e], scale, zp, qmax, re);
                r[e] = re;
                if (!LAST) stat_acc(re, mn, mx, has_nan, ss);
            }
            reinterpret_cast<u8x4 *>(codes)[i + q * stride] = c;
            if (!LAST) reinterpret_cast<f32x4 *>(rout)[i + q * stride] = r;
        }
    }
    for (; i < nvec; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(rin)[i];
        f32x4 r;
        u8x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float re;
            q[e] = quant_one(v[e], scale, zp, qmax, re);
            r[e] = re;
            if (!LAST) stat_acc(re, mn, mx, has_nan, ss);
        }
        reinterpret_cast<u8x4 *>(codes)[i] = q;
        if (!LAST) reinterpret_cast<f32x4 *>(rout)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t j = (nvec << 2) + threadIdx.x;
        float re;
        codes[j] = quant_one(rin[j], scale, zp, qmax, re);
        if (!LAST) {
            rout[j] = re;
            stat_acc(re, mn, mx, has_nan, ss);
        }
    }
    if (!LAST) stats_block_reduce(mn, mx, has_nan, ss, part_out + blockIdx.x);
}

// rtvq.py:85-103: ((0 + deq_0) + deq_1) + ...
__global__ __launch_bounds__(ELT_THREADS) void k_rtvq_dequant(const uint8_t *__restrict__ codes, int64_t cstride,
                                                              int64_t n, int stages,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ zp, float *__restrict__ out) {
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * ELT_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * ELT_THREADS + threadIdx.x; i < nvec; i += stride) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < stages; ++s) {
            const u8x4 q = reinterpret_cast<const u8x4 *>(codes + (size_t)s * cstride)[i];
            const float sc = scale[s], z = zp[s];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __fadd_rn(acc[e], __fdiv_rn(__fsub_rn((float)q[e], z), sc));
        }
        reinterpret_cast<f32x4 *>(out)[i] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nvec << 2) + threadIdx.x;
        float acc = 0.f;
        for (int s = 0; s < stages; ++s)
            acc = __fadd_rn(acc, __fdiv_rn(__fsub_rn((float)codes[(size_t)s * cstride + i], zp[s]), scale[s]));
        out[i] = acc;
    }
}

static int rtvq_grid(int64_t n) {
    int64_t b = ((n >> 2) + ELT_THREADS * 4 - 1) / (ELT_THREADS * 4);
    if (b < 1) b = 1;
    if (b > RTVQ_MAX_BLOCKS) b = RTVQ_MAX_BLOCKS;
    return (int)b;
}

extern "C" int64_t svdq_rtvq_work_bytes(int64_t n) {
    return svdq_align_up(n * 4, 256) + 2 * (int64_t)RTVQ_MAX_BLOCKS * sizeof(RtvqPartial) + 256;  // residual + ping-pong partials
}

extern "C" int svdq_rtvq_quantize(const float *x, int64_t n, int32_t bits, int32_t stages, uint8_t *codes,
                                  int64_t code_stride, float *scale, float *zp, float *rnorm, void *work,
                                  void *stream) {
    if (n < 1 || !x || !codes || !scale || !zp || !rnorm || !work) {
        svdq_set_error("svdq_rtvq_quantize: bad argument (n must be >= 1; empty tensors are the caller's job)");
        return SVDQ_EINVAL;
    }
    if (bits < 1 || bits > 8 || stages < 1 || stages > SVDQ_MAX_STAGES) {
        svdq_set_error("svdq_rtvq_quantize: bits in [1,8], stages in [1,%d]", SVDQ_MAX_STAGES);
        return SVDQ_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(work) & 255) ||
        (reinterpret_cast<uintptr_t>(codes) & 3) || (code_stride & 3) || code_stride < n) {
        svdq_set_error("svdq_rtvq_quantize: x 16-byte, work 256-byte, codes 4-byte aligned; code_stride %% 4 == 0, >= n");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    float *res = reinterpret_cast<float *>(wb);
    RtvqPartial *part[2];
    part[0] = reinterpret_cast<RtvqPartial *>(wb + svdq_align_up(n * 4, 256));
    part[1] = part[0] + RTVQ_MAX_BLOCKS;
    const int grid = rtvq_grid(n);
    // 1 + S launches: statistics of x, then per stage "parameters (recomputed per block) + codes + residual +
    // statistics of the residual"
    hipLaunchKernelGGL(k_rtvq_stats, dim3(grid), dim3(ELT_THREADS), 0, st, x, n, part[0]);
    for (int s = 0; s < stages; ++s) {
        const float *rin = (s == 0) ? x : res;
        if (s == stages - 1)
            hipLaunchKernelGGL((k_rtvq_apply<true>), dim3(grid), dim3(ELT_THREADS), 0, st, rin, res, n, part[s & 1], grid,
                               s, bits, codes + (size_t)s * code_stride, part[(s + 1) & 1], scale, zp, rnorm);
        else
            hipLaunchKernelGGL((k_rtvq_apply<false>), dim3(grid), dim3(ELT_THREADS), 0, st, rin, res, n, part[s & 1], grid,
                               s, bits, codes + (size_t)s * code_stride, part[(s + 1) & 1], scale, zp, rnorm);
    }
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_rtvq_dequantize(const uint8_t *codes, int64_t code_stride, int64_t n, int32_t stages,
                                    const float *scale, const float *zp, float *out, void *stream) {
    if (n < 1 || !codes || !scale || !zp || !out || stages < 1) {
        svdq_set_error("svdq_rtvq_dequantize: bad argument");
        return SVDQ_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(codes) & 3) || (reinterpret_cast<uintptr_t>(out) & 15) || (code_stride & 3) ||
        code_stride < n) {
        svdq_set_error("svdq_rtvq_dequantize: codes 4-byte / out 16-byte aligned; code_stride %% 4 == 0, >= n");
        return SVDQ_EINVAL;
    }
    hipLaunchKernelGGL(k_rtvq_dequant, dim3(rtvq_grid(n)), dim3(ELT_THREADS), 0, (hipStream_t)stream, codes,
                       code_stride, n, stages, scale, zp, out);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// ---- qbit = 16 (rtvq.py:22-25): single stage, int16 codes.  The reference casts the clamped fp32 value (0 .. 65535) to
// torch.int16; on the CPU that goes through a 32-bit integer and keeps the low 16 bits, so codes above 32767 come out
// negative -- reproduced as is (tests/golden/rtvq_cases.npz holds the reference's values).
__global__ __launch_bounds__(ELT_THREADS) void k_asym16_apply(const float *__restrict__ x, int64_t n,
                                                              const RtvqPartial *__restrict__ part, int nblk,
                                                              int16_t *__restrict__ codes, float *__restrict__ scale_out,
                                                              float *__restrict__ zp_out) {
    float scale, zp, rnorm;
    stage_params(part, nblk, 16, scale, zp, rnorm);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scale_out[0] = scale;
        zp_out[0] = zp;
    }
    const int64_t stride = (int64_t)gridDim.x * ELT_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * ELT_THREADS + threadIdx.x; i < n; i += stride) {
        float vq = rintf(__fadd_rn(__fmul_rn(scale, x[i]), zp));
        int32_t q = 0;
        if (vq == vq) {
            vq = vq < 0.f ? 0.f : (vq > 65535.f ? 65535.f : vq);
            q = (int32_t)vq;
        }
        codes[i] = (int16_t)(uint16_t)q;
    }
}

__global__ __launch_bounds__(ELT_THREADS) void k_asym16_dequant(const int16_t *__restrict__ codes, int64_t n,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ zp, float *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * ELT_THREADS;
    const float sc = scale[0], z = zp[0];
    for (int64_t i = (int64_t)blockIdx.x * ELT_THREADS + threadIdx.x; i < n; i += stride)
        out[i] = __fdiv_rn(__fsub_rn((float)codes[i], z), sc);   // rtvq.py:35 on the int16 values as they are
}

extern "C" int svdq_asym16_quantize(const float *x, int64_t n, int16_t *codes, float *scale, float *zp, void *work,
                                    void *stream) {
    if (n < 1 || !x || !codes || !scale || !zp || !work || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (reinterpret_cast<uintptr_t>(work) & 255)) {
        svdq_set_error("svdq_asym16_quantize: bad argument (n >= 1, x 16-byte aligned, work 256-byte aligned)");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    RtvqPartial *part = reinterpret_cast<RtvqPartial *>(reinterpret_cast<uint8_t *>(work) + svdq_align_up(n * 4, 256));
    const int grid = rtvq_grid(n);
    hipLaunchKernelGGL(k_rtvq_stats, dim3(grid), dim3(ELT_THREADS), 0, st, x, n, part);
    hipLaunchKernelGGL(k_asym16_apply, dim3(grid), dim3(ELT_THREADS), 0, st, x, n, part, grid, codes, scale, zp);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_asym16_dequantize(const int16_t *codes, int64_t n, const float *scale, const float *zp, float *out,
                                      void *stream) {
    if (n < 1 || !codes || !scale || !zp || !out) {
        svdq_set_error("svdq_asym16_dequantize: bad argument");
        return SVDQ_EINVAL;
    }
    hipLaunchKernelGGL(k_asym16_dequant, dim3(rtvq_grid(n)), dim3(ELT_THREADS), 0, (hipStream_t)stream, codes, n, scale,
                       zp, out);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// ------------------------------------------------------------------------------------ masks
#define MASK_TILE 2048  // elements per block: 256 threads x 8

// ---- tile helpers shared by the single-parameter and the batched mask kernels ------------------
// torch.bool storage is one byte per element holding 0 or 1, so 8 mask bytes can be voted on at once:
// byte lanes of a 64-bit word accumulate the per-element counts (n_masks <= 32 < 256, no carries).
__device__ __forceinline__ unsigned long long load_mask8(const uint8_t *p, int lim) {
    unsigned long long w = 0;
    if (lim == 8 && (reinterpret_cast<uintptr_t>(p) & 7) == 0) return *reinterpret_cast<const unsigned long long *>(p);
    for (int e = 0; e < lim; ++e) w |= (unsigned long long)(p[e] != 0) << (8 * e);
    return w;
}

// strategy: low byte = SVDQ_MASK_*; for the majority vote bits 8.. hold (votes needed + 1), 0 = the default threshold 0.5
__device__ __forceinline__ unsigned long long vote8(unsigned long long acc, int n_masks, int strategy) {
    unsigned long long out = 0;
    const int kind = strategy & 0xff, need1 = strategy >> 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int v = (int)((acc >> (8 * e)) & 0xff);
        int bit;
        if (kind == SVDQ_MASK_UNION)
            bit = v > 0;
        else if (kind == SVDQ_MASK_INTERSECTION)
            bit = v == n_masks;
        else
            bit = need1 ? (v >= need1 - 1) : (2 * v >= n_masks);  // vote_sum >= threshold * len(masks), mask_loader.py:483
        out |= (unsigned long long)bit << (8 * e);
    }
    return out;
}

// sum over the 256 threads of a block, valid in thread 0: wave shuffles, then one LDS hop for the 4 wave sums
__device__ __forceinline__ unsigned block_sum_u32(unsigned v) {
    __shared__ unsigned s_w[ELT_THREADS / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned tot = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < ELT_THREADS / 64; ++w) tot += s_w[w];
    }
    __syncthreads();
    return tot;
}

// one 2048-element tile: combined mask out, number of selected elements returned by thread 0 via *cnt_out
__device__ void combine_tile(const uint8_t *const *masks, int n_masks, int strategy, int64_t tile, int64_t numel,
                             uint8_t *out, unsigned *cnt_out) {
    const int64_t base = tile * MASK_TILE + (int64_t)threadIdx.x * 8;
    unsigned cnt = 0;
    if (base < numel) {
        const int lim = (int)((numel - base) < 8 ? (numel - base) : 8);
        unsigned long long acc = 0;
        for (int m0 = 0; m0 < n_masks; m0 += 8) {   // 8 independent loads in flight per thread
            unsigned long long w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (m0 + j < n_masks) ? load_mask8(masks[m0 + j] + base, lim) : 0ull;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += w[j];
        }
        const unsigned long long res = vote8(acc, n_masks, strategy);
        if (lim == 8 && (reinterpret_cast<uintptr_t>(out + base) & 7) == 0) {
            *reinterpret_cast<unsigned long long *>(out + base) = res;
        } else {
            for (int e = 0; e < lim; ++e) out[base + e] = (uint8_t)((res >> (8 * e)) & 1);
        }
        cnt = (unsigned)__popcll(lim == 8 ? res : (res & ((1ull << (8 * lim)) - 1)));
    }
    const unsigned tot = block_sum_u32(cnt);
    if (threadIdx.x == 0) *cnt_out = tot;
}

// one tile of an order-preserving compaction for n_src buffers: every source tile is loaded with
// 16-B loads, compacted through LDS (selected -> one image, unselected -> a second one when asked)
// and written out as ONE contiguous, coalesced run per region.
__device__ void scatter_tile(const uint8_t *mask, int64_t tile, int64_t numel, const float *const *src,
                             float *const *dst_true, float *const *dst_false, int n_src,
                             unsigned long long tile_off) {
    __shared__ unsigned s[ELT_THREADS];
    __shared__ float lt[MASK_TILE], lf[MASK_TILE];
    const int tid = threadIdx.x;
    const int64_t tbase = tile * MASK_TILE;
    const int64_t base = tbase + (int64_t)tid * 8;
    const int lim = base < numel ? (int)((numel - base) < 8 ? (numel - base) : 8) : 0;
    const unsigned long long w = lim ? load_mask8(mask + base, lim) : 0;
    unsigned sel = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) sel |= (unsigned)((w >> (8 * e)) & 1) << e;
    const unsigned cnt = __popc(sel);
    s[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < ELT_THREADS; off <<= 1) {
        unsigned add = tid >= off ? s[tid - off] : 0;
        __syncthreads();
        s[tid] += add;
        __syncthreads();
    }
    const unsigned tot_true = s[ELT_THREADS - 1];
    const int tile_n = (int)((numel - tbase) < MASK_TILE ? (numel - tbase) : MASK_TILE);
    const unsigned tot_false = (unsigned)tile_n - tot_true;
    const unsigned pt = s[tid] - cnt;                 // rank of this thread's first selected element
    const unsigned pf = (unsigned)(tid * 8) - pt;     // ... and of its first unselected one
    const unsigned long long f_off = (unsigned long long)tbase - tile_off;
    for (int m = 0; m < n_src; ++m) {
        const float *sp = src[m] + base;
        float v[8];
        if (lim == 8 && (reinterpret_cast<uintptr_t>(sp) & 15) == 0) {
            const f32x4 a = reinterpret_cast<const f32x4 *>(sp)[0], b = reinterpret_cast<const f32x4 *>(sp)[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = e < lim ? sp[e] : 0.f;
        }
        unsigned jt = pt, jf = pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e < lim) {
                if (sel & (1u << e))
                    lt[jt++] = v[e];
                else
                    lf[jf++] = v[e];
            }
        }
        __syncthreads();
        if (dst_true) {
            float *dt = dst_true[m] + tile_off;
            for (unsigned i = tid; i < tot_true; i += ELT_THREADS) dt[i] = lt[i];
        }
        if (dst_false) {
            float *df = dst_false[m] + f_off;
            for (unsigned i = tid; i < tot_false; i += ELT_THREADS) df[i] = lf[i];
        }
        __syncthreads();
    }
}

// mask_loader.py:412-485.  torch.bool storage is one byte per element, 0 or 1.
__global__ __launch_bounds__(ELT_THREADS) void k_mask_combine(const uint8_t *const *__restrict__ masks, int n_masks,
                                                              int64_t numel, int strategy, uint8_t *__restrict__ out,
                                                              unsigned long long *__restrict__ count) {
    __shared__ unsigned cnt;
    combine_tile(masks, n_masks, strategy, blockIdx.x, numel, out, &cnt);
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(count, (unsigned long long)cnt);
}

__global__ __launch_bounds__(ELT_THREADS) void k_mask_count(const uint8_t *__restrict__ mask, int invert, int64_t numel,
                                                            unsigned *__restrict__ tile_counts) {
    const int64_t base = (int64_t)blockIdx.x * MASK_TILE + (int64_t)threadIdx.x * 8;
    unsigned cnt = 0;
    for (int e = 0; e < 8; ++e)
        if (base + e < numel) cnt += ((mask[base + e] != 0) != (invert != 0)) ? 1u : 0u;
    const unsigned tot = block_sum_u32(cnt);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// exclusive scan of the tile counts (one block; ntiles <= a few thousand), total -> count_out
__global__ __launch_bounds__(1024) void k_mask_scan(const unsigned *__restrict__ tile_counts, int ntiles,
                                                    unsigned long long *__restrict__ tile_offsets,
                                                    long long *__restrict__ count_out) {
    __shared__ unsigned long long s[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned long long v = i < ntiles ? tile_counts[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            unsigned long long add = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < ntiles) tile_offsets[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = (long long)carry;
}

// mask_loader.py:675-679 / :706-709: flat[mask] (or flat[~mask]) for n_src buffers sharing one mask
__global__ __launch_bounds__(ELT_THREADS) void k_mask_scatter(const float *const *__restrict__ src,
                                                              float *const *__restrict__ dst, int n_src,
                                                              const uint8_t *__restrict__ mask, int invert,
                                                              int64_t numel,
                                                              const unsigned long long *__restrict__ tile_offsets) {
    // tile_offsets were scanned over the SELECTED count (mask != invert); scatter_tile wants the offset of the
    // True elements, so for invert the roles of its two outputs are swapped
    const unsigned long long off = tile_offsets[blockIdx.x];
    if (!invert) {
        scatter_tile(mask, blockIdx.x, numel, src, dst, nullptr, n_src, off);
    } else {
        const unsigned long long true_off = (unsigned long long)blockIdx.x * MASK_TILE - off;
        scatter_tile(mask, blockIdx.x, numel, src, nullptr, dst, n_src, true_off);
    }
}

extern "C" int64_t svdq_mask_work_bytes(int64_t numel) {
    const int64_t ntiles = (numel + MASK_TILE - 1) / MASK_TILE;
    return svdq_align_up(ntiles * 4, 256) + svdq_align_up(ntiles * 8, 256) + 256;
}

extern "C" int svdq_mask_combine(const void *mask_ptrs, int32_t n_masks, int64_t numel, int32_t strategy, uint8_t *out,
                                 int64_t *count, void *work, void *stream) {
    (void)work;
    if (n_masks < 1) {
        svdq_set_error("Empty mask list");  // mask_loader.py:425
        return SVDQ_EINVAL;
    }
    if ((strategy & 0xff) > SVDQ_MASK_MAJORITY || strategy < 0 ||
        ((strategy >> 8) != 0 && (strategy & 0xff) != SVDQ_MASK_MAJORITY) || (strategy >> 8) > SVDQ_MAX_TASKS + 2) {
        svdq_set_error("Unknown mask strategy: %d", strategy);  // mask_loader.py:611
        return SVDQ_EINVAL;
    }
    if (!mask_ptrs || !out || !count || numel < 1) {
        svdq_set_error("svdq_mask_combine: bad argument");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, 8, st) != hipSuccess) return SVDQ_EHIP;
    const int ntiles = (int)((numel + MASK_TILE - 1) / MASK_TILE);
    hipLaunchKernelGGL(k_mask_combine, dim3(ntiles), dim3(ELT_THREADS), 0, st,
                       reinterpret_cast<const uint8_t *const *>(mask_ptrs), n_masks, numel, strategy, out,
                       reinterpret_cast<unsigned long long *>(count));
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_mask_compact(const void *src_ptrs, const void *dst_ptrs, int32_t n_src, const uint8_t *mask,
                                 int32_t invert, int64_t numel, int64_t *count, void *work, void *stream) {
    if (!src_ptrs || !dst_ptrs || !mask || !count || !work || n_src < 1 || numel < 1) {
        svdq_set_error("svdq_mask_compact: bad argument");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (int)((numel + MASK_TILE - 1) / MASK_TILE);
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets = reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ntiles * 4, 256));
    hipLaunchKernelGGL(k_mask_count, dim3(ntiles), dim3(ELT_THREADS), 0, st, mask, invert, numel, tile_counts);
    hipLaunchKernelGGL(k_mask_scan, dim3(1), dim3(1024), 0, st, tile_counts, ntiles, tile_offsets,
                       reinterpret_cast<long long *>(count));
    hipLaunchKernelGGL(k_mask_scatter, dim3(ntiles), dim3(ELT_THREADS), 0, st,
                       reinterpret_cast<const float *const *>(src_ptrs), reinterpret_cast<float *const *>(dst_ptrs),
                       n_src, mask, invert, numel, tile_offsets);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// ------------------------------------------------------------------------------------ projection
// project_to_basis (compress.py:6-21) with the mean subtraction of compress.py:35-37 folded in,
// for callers that bring their own basis: c[i] = sum_d float(U[d][i]) * (delta[d] - mean[d]).
// Thread per row (adjacent threads read adjacent rows of the row-major [D,k] / [D,nl] arrays),
// fp32 per thread, fp64 across threads and blocks, fixed-order final sum => deterministic.
#define PROJ_BLOCKS 1024

__device__ __forceinline__ float u_load(const __half *u, int64_t i) { return __half2float(u[i]); }
__device__ __forceinline__ float u_load(const float *u, int64_t i) { return u[i]; }

template <typename T>
__global__ __launch_bounds__(ELT_THREADS) void k_project(const T *__restrict__ uh, const T *__restrict__ ul,
                                                         int64_t rows, int k, int nl,
                                                         const float *__restrict__ delta,
                                                         const float *__restrict__ mean,
                                                         double *__restrict__ part /*[grid][32]*/) {
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    const int64_t stride = (int64_t)gridDim.x * ELT_THREADS;
    for (int64_t d = (int64_t)blockIdx.x * ELT_THREADS + threadIdx.x; d < rows; d += stride) {
        float x = delta[d];
        if (mean) x = __fsub_rn(x, mean[d]);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i < k)
                acc[i] = fmaf(u_load(uh, d * k + i), x, acc[i]);
            else if (i < k + nl)
                acc[i] = fmaf(u_load(ul, d * nl + (i - k)), x, acc[i]);
        }
    }
    __shared__ double red[ELT_THREADS];
    for (int i = 0; i < k + nl; ++i) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j == i) v = (double)acc[j];
        red[threadIdx.x] = v;
        __syncthreads();
        for (int off = ELT_THREADS / 2; off > 0; off >>= 1) {
            if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * 32 + i] = red[0];
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_project_finish(const double *__restrict__ part, int nblk, int ncols,
                                                       float *__restrict__ c_out) {
    const int i = threadIdx.x;
    if (i >= ncols) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * 32 + i];
    c_out[i] = (float)s;
}

static int project_grid(int64_t rows) {
    int64_t b = (rows + ELT_THREADS * 8 - 1) / (ELT_THREADS * 8);
    if (b < 1) b = 1;
    if (b > PROJ_BLOCKS) b = PROJ_BLOCKS;
    return (int)b;
}

extern "C" int64_t svdq_project_work_bytes(int64_t rows, int32_t ncols) {
    (void)ncols;
    return (int64_t)project_grid(rows) * 32 * 8 + 256;
}

extern "C" int svdq_project(const void *u_high, const void *u_low, int32_t u_fp16, int64_t rows, int32_t k, int32_t nl,
                            const float *delta, const float *mean, float *c_out, void *work, void *stream) {
    if (rows < 1 || k < 0 || nl < 0 || k + nl < 1 || k + nl > 32 || !delta || !c_out || !work ||
        (k > 0 && !u_high) || (nl > 0 && !u_low)) {
        svdq_set_error("svdq_project: bad argument (1 <= k + nl <= 32)");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int grid = project_grid(rows);
    double *part = reinterpret_cast<double *>(work);
    if (u_fp16)
        hipLaunchKernelGGL((k_project<__half>), dim3(grid), dim3(ELT_THREADS), 0, st,
                           reinterpret_cast<const __half *>(u_high), reinterpret_cast<const __half *>(u_low), rows, k,
                           nl, delta, mean, part);
    else
        hipLaunchKernelGGL((k_project<float>), dim3(grid), dim3(ELT_THREADS), 0, st,
                           reinterpret_cast<const float *>(u_high), reinterpret_cast<const float *>(u_low), rows, k, nl,
                           delta, mean, part);
    hipLaunchKernelGGL(k_project_finish, dim3(1), dim3(64), 0, st, part, grid, k + nl, c_out);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// ------------------------------------------------------------------------------------ merge consumers
// reconstruct_from_coefficients (merge.py:144-194): out = ((U_high c_high + U_low c_low) + mean) * scale.
// HBM-bound: reads the fp16 (or fp32) basis once (e*(k+nl) B/row) + mean, writes 4 B/row.
// Thread per row; adjacent threads read adjacent rows of the row-major [D,k] / [D,nl] arrays.
template <typename T>
__global__ __launch_bounds__(ELT_THREADS) void k_reconstruct(const T *__restrict__ uh, const T *__restrict__ ul,
                                                             int64_t rows, int k, int nl,
                                                             const float *__restrict__ coef,
                                                             const float *__restrict__ mean, float scale,
                                                             float *__restrict__ out) {
    __shared__ float c[32];
    if (threadIdx.x < k + nl) c[threadIdx.x] = coef[threadIdx.x];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * ELT_THREADS;
    for (int64_t d = (int64_t)blockIdx.x * ELT_THREADS + threadIdx.x; d < rows; d += stride) {
        float hi = 0.f, lo = 0.f;
        for (int i = 0; i < k; ++i) hi = fmaf(u_load(uh, d * k + i), c[i], hi);
        for (int j = 0; j < nl; ++j) lo = fmaf(u_load(ul, d * nl + j), c[k + j], lo);
        float v = __fadd_rn(hi, lo);
        if (mean) v = __fadd_rn(v, mean[d]);
        out[d] = __fmul_rn(v, scale);
    }
}

extern "C" int svdq_reconstruct(const void *u_high, const void *u_low, int32_t u_fp16, int64_t rows, int32_t k,
                                int32_t nl, const float *coef, const float *mean, float scale, float *out,
                                void *stream) {
    if (rows < 1 || k < 0 || nl < 0 || k + nl > 32 || !out || (k + nl > 0 && !coef) || (k > 0 && !u_high) ||
        (nl > 0 && !u_low)) {
        svdq_set_error("svdq_reconstruct: bad argument (k + nl <= 32)");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int grid = project_grid(rows);
    if (u_fp16)
        hipLaunchKernelGGL((k_reconstruct<__half>), dim3(grid), dim3(ELT_THREADS), 0, st,
                           reinterpret_cast<const __half *>(u_high), reinterpret_cast<const __half *>(u_low), rows, k,
                           nl, coef, mean, scale, out);
    else
        hipLaunchKernelGGL((k_reconstruct<float>), dim3(grid), dim3(ELT_THREADS), 0, st,
                           reinterpret_cast<const float *>(u_high), reinterpret_cast<const float *>(u_low), rows, k, nl,
                           coef, mean, scale, out);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// reconstruct_from_masked (mask_loader.py:712-763): out = zeros; out[mask] = signal; out[~mask] = noise.
// Inverse of the compaction: same tile counts + scan, then a gather by rank.
__global__ __launch_bounds__(ELT_THREADS) void k_mask_expand(const float *__restrict__ sig,
                                                             const float *__restrict__ noise,
                                                             const uint8_t *__restrict__ mask, int64_t numel,
                                                             const unsigned long long *__restrict__ tile_offsets,
                                                             float *__restrict__ out) {
    const int64_t base = (int64_t)blockIdx.x * MASK_TILE + (int64_t)threadIdx.x * 8;
    unsigned sel = 0, cnt = 0, n = 0;
    for (int e = 0; e < 8; ++e)
        if (base + e < numel) {
            ++n;
            if (mask[base + e] != 0) {
                sel |= 1u << e;
                ++cnt;
            }
        }
    __shared__ unsigned s[ELT_THREADS];
    s[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 1; off < ELT_THREADS; off <<= 1) {
        unsigned add = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += add;
        __syncthreads();
    }
    if (!n) return;
    const unsigned long long t0 = tile_offsets[blockIdx.x] + (s[threadIdx.x] - cnt);  // rank among True
    const unsigned long long f0 = (unsigned long long)base - t0;                       // rank among False
    unsigned jt = 0, jf = 0;
    for (unsigned e = 0; e < n; ++e) {
        float v;
        if (sel & (1u << e))
            v = sig[t0 + jt++];
        else
            v = noise ? noise[f0 + jf++] : 0.f;
        out[base + e] = v;
    }
}

extern "C" int svdq_mask_expand(const float *signal, const float *noise, const uint8_t *mask, int64_t numel,
                                float *out, void *work, void *stream) {
    if (!signal || !mask || !out || !work || numel < 1) {
        svdq_set_error("svdq_mask_expand: bad argument");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (int)((numel + MASK_TILE - 1) / MASK_TILE);
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets = reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ntiles * 4, 256));
    long long *total = reinterpret_cast<long long *>(wb + svdq_align_up((int64_t)ntiles * 4, 256) +
                                                     svdq_align_up((int64_t)ntiles * 8, 256));
    hipLaunchKernelGGL(k_mask_count, dim3(ntiles), dim3(ELT_THREADS), 0, st, mask, 0, numel, tile_counts);
    hipLaunchKernelGGL(k_mask_scan, dim3(1), dim3(1024), 0, st, tile_counts, ntiles, tile_offsets, total);
    hipLaunchKernelGGL(k_mask_expand, dim3(ntiles), dim3(ELT_THREADS), 0, st, signal, noise, mask, numel, tile_offsets,
                       out);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}


// ------------------------------------------------------------------------------------ batched masks
// The same operators over a ragged SET of parameters in a handful of launches (cli.py runs them once
// per parameter and task): a tile table maps every 2048-element tile to (parameter,
// tile-in-parameter); per-parameter counts stay on the device and feed rows_dev of the plan.
struct svdq_maskset {
    int32_t n_params, n_tiles;
    int64_t *h_numel;
    int32_t *h_tile_begin;
    int64_t *d_numel;        // [Q]
    int32_t *d_tile_begin;   // [Q+1]
    int32_t *d_tile_param;   // [n_tiles]
};

extern "C" int svdq_maskset_create(svdq_maskset **out, int32_t n_params, const int64_t *numel) {
    if (!out || !numel || n_params < 1) {
        svdq_set_error("svdq_maskset_create: bad argument");
        return SVDQ_EINVAL;
    }
    svdq_maskset *ms = (svdq_maskset *)calloc(1, sizeof(svdq_maskset));
    ms->n_params = n_params;
    ms->h_numel = (int64_t *)calloc(n_params, sizeof(int64_t));
    ms->h_tile_begin = (int32_t *)calloc(n_params + 1, sizeof(int32_t));
    int64_t tiles = 0;
    for (int q = 0; q < n_params; ++q) {
        if (numel[q] < 1) {
            svdq_set_error("svdq_maskset_create: parameter %d has %lld elements", q, (long long)numel[q]);
            free(ms->h_numel);
            free(ms->h_tile_begin);
            free(ms);
            return SVDQ_EINVAL;
        }
        ms->h_numel[q] = numel[q];
        ms->h_tile_begin[q] = (int32_t)tiles;
        tiles += (numel[q] + MASK_TILE - 1) / MASK_TILE;
    }
    ms->h_tile_begin[n_params] = (int32_t)tiles;
    ms->n_tiles = (int32_t)tiles;
    int32_t *tp = (int32_t *)malloc(sizeof(int32_t) * tiles);
    for (int q = 0; q < n_params; ++q)
        for (int t = ms->h_tile_begin[q]; t < ms->h_tile_begin[q + 1]; ++t) tp[t] = q;
    hipError_t e = hipMalloc((void **)&ms->d_numel, sizeof(int64_t) * n_params);
    if (e == hipSuccess) e = hipMalloc((void **)&ms->d_tile_begin, sizeof(int32_t) * (n_params + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&ms->d_tile_param, sizeof(int32_t) * tiles);
    if (e == hipSuccess) e = hipMemcpy(ms->d_numel, ms->h_numel, sizeof(int64_t) * n_params, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(ms->d_tile_begin, ms->h_tile_begin, sizeof(int32_t) * (n_params + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ms->d_tile_param, tp, sizeof(int32_t) * tiles, hipMemcpyHostToDevice);
    free(tp);
    if (e != hipSuccess) {
        svdq_set_error("svdq_maskset_create: %s", hipGetErrorString(e));
        svdq_maskset_destroy(ms);
        return SVDQ_EHIP;
    }
    *out = ms;
    return SVDQ_OK;
}

extern "C" void svdq_maskset_destroy(svdq_maskset *ms) {
    if (!ms) return;
    if (ms->d_numel) (void)hipFree(ms->d_numel);
    if (ms->d_tile_begin) (void)hipFree(ms->d_tile_begin);
    if (ms->d_tile_param) (void)hipFree(ms->d_tile_param);
    free(ms->h_numel);
    free(ms->h_tile_begin);
    free(ms);
}

extern "C" int64_t svdq_maskset_work_bytes(const svdq_maskset *ms) {
    if (!ms) return 0;
    return svdq_align_up((int64_t)ms->n_tiles * 4, 256) + svdq_align_up((int64_t)ms->n_tiles * 8, 256) + 256;
}

__global__ __launch_bounds__(ELT_THREADS) void k_maskset_combine(const int32_t *__restrict__ tile_param,
                                                                 const int32_t *__restrict__ tile_begin,
                                                                 const int64_t *__restrict__ numel_tab,
                                                                 const uint8_t *const *__restrict__ masks, int n_masks,
                                                                 int strategy, uint8_t *const *__restrict__ outs,
                                                                 unsigned long long *__restrict__ counts,
                                                                 unsigned *__restrict__ tile_counts) {
    const int q = tile_param[blockIdx.x];
    __shared__ unsigned cnt;
    combine_tile(masks + (size_t)q * n_masks, n_masks, strategy, blockIdx.x - tile_begin[q], numel_tab[q], outs[q], &cnt);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (tile_counts) tile_counts[blockIdx.x] = cnt;       // feeds the index build without a counting pass
        else if (cnt) atomicAdd(&counts[q], (unsigned long long)cnt);
    }
}

__global__ __launch_bounds__(ELT_THREADS) void k_maskset_count(const int32_t *__restrict__ tile_param,
                                                               const int32_t *__restrict__ tile_begin,
                                                               const int64_t *__restrict__ numel_tab,
                                                               const uint8_t *const *__restrict__ masks,
                                                               unsigned *__restrict__ tile_counts) {
    const int q = tile_param[blockIdx.x];
    const int64_t numel = numel_tab[q];
    const uint8_t *mask = masks[q];
    const int64_t base = (int64_t)(blockIdx.x - tile_begin[q]) * MASK_TILE + (int64_t)threadIdx.x * 8;
    unsigned cnt = 0;
    if (base < numel) {
        const int lim = (int)((numel - base) < 8 ? (numel - base) : 8);
        unsigned long long w = load_mask8(mask + base, lim);      // one 8-byte load; bytes are 0 or 1
        w = (w | (w >> 1) | (w >> 2) | (w >> 3) | (w >> 4) | (w >> 5) | (w >> 6) | (w >> 7)) & 0x0101010101010101ull;
        cnt = (unsigned)__popcll(w);
    }
    const unsigned tot = block_sum_u32(cnt);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// one block per parameter: exclusive scan of ITS tile counts; true / false totals out
__global__ __launch_bounds__(1024) void k_maskset_scan(const int32_t *__restrict__ tile_begin,
                                                       const int64_t *__restrict__ numel_tab,
                                                       const unsigned *__restrict__ tile_counts,
                                                       unsigned long long *__restrict__ tile_offsets,
                                                       long long *__restrict__ count_true,
                                                       long long *__restrict__ count_false) {
    __shared__ unsigned long long s[1024];
    __shared__ unsigned long long carry;
    const int q = blockIdx.x, t0 = tile_begin[q], nt = tile_begin[q + 1] - t0;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nt; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned long long v = i < nt ? tile_counts[t0 + i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            unsigned long long add = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nt) tile_offsets[t0 + i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (count_true) count_true[q] = (long long)carry;
        if (count_false) count_false[q] = (long long)numel_tab[q] - (long long)carry;
    }
}

// flat[mask] -> dst_true, flat[~mask] -> dst_false (optional), for n_src buffers per parameter, one pass
__global__ __launch_bounds__(ELT_THREADS) void k_maskset_scatter(const int32_t *__restrict__ tile_param,
                                                                 const int32_t *__restrict__ tile_begin,
                                                                 const int64_t *__restrict__ numel_tab,
                                                                 const uint8_t *const *__restrict__ masks,
                                                                 const float *const *__restrict__ src,
                                                                 float *const *__restrict__ dst_true,
                                                                 float *const *__restrict__ dst_false, int n_src,
                                                                 const unsigned long long *__restrict__ tile_offsets) {
    const int q = tile_param[blockIdx.x];
    scatter_tile(masks[q], blockIdx.x - tile_begin[q], numel_tab[q], src + (size_t)q * n_src,
                 dst_true + (size_t)q * n_src, dst_false ? dst_false + (size_t)q * n_src : nullptr, n_src,
                 tile_offsets[blockIdx.x]);
}

extern "C" int svdq_maskset_combine(const svdq_maskset *ms, const void *mask_ptrs, int32_t n_masks, int32_t strategy,
                                    const void *out_ptrs, int64_t *counts, void *stream) {
    if (!ms || !mask_ptrs || !out_ptrs || !counts) {
        svdq_set_error("svdq_maskset_combine: bad argument");
        return SVDQ_EINVAL;
    }
    if (n_masks < 1) {
        svdq_set_error("Empty mask list");
        return SVDQ_EINVAL;
    }
    if (strategy < SVDQ_MASK_UNION || strategy > SVDQ_MASK_MAJORITY) {
        svdq_set_error("Unknown mask strategy: %d", strategy);
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(int64_t) * ms->n_params, st) != hipSuccess) return SVDQ_EHIP;
    hipLaunchKernelGGL(k_maskset_combine, dim3(ms->n_tiles), dim3(ELT_THREADS), 0, st, ms->d_tile_param,
                       ms->d_tile_begin, ms->d_numel, reinterpret_cast<const uint8_t *const *>(mask_ptrs), n_masks,
                       strategy, reinterpret_cast<uint8_t *const *>(out_ptrs),
                       reinterpret_cast<unsigned long long *>(counts), (unsigned *)nullptr);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_maskset_compact(const svdq_maskset *ms, const void *mask_ptrs, const void *src_ptrs,
                                    const void *dst_true_ptrs, const void *dst_false_ptrs, int32_t n_src,
                                    int64_t *count_true, int64_t *count_false, void *work, void *stream) {
    if (!ms || !mask_ptrs || !src_ptrs || !dst_true_ptrs || !work || n_src < 1) {
        svdq_set_error("svdq_maskset_compact: bad argument");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets =
        reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ms->n_tiles * 4, 256));
    auto mp = reinterpret_cast<const uint8_t *const *>(mask_ptrs);
    hipLaunchKernelGGL(k_maskset_count, dim3(ms->n_tiles), dim3(ELT_THREADS), 0, st, ms->d_tile_param, ms->d_tile_begin,
                       ms->d_numel, mp, tile_counts);
    hipLaunchKernelGGL(k_maskset_scan, dim3(ms->n_params), dim3(1024), 0, st, ms->d_tile_begin, ms->d_numel, tile_counts,
                       tile_offsets, reinterpret_cast<long long *>(count_true),
                       reinterpret_cast<long long *>(count_false));
    hipLaunchKernelGGL(k_maskset_scatter, dim3(ms->n_tiles), dim3(ELT_THREADS), 0, st, ms->d_tile_param,
                       ms->d_tile_begin, ms->d_numel, mp, reinterpret_cast<const float *const *>(src_ptrs),
                       reinterpret_cast<float *const *>(dst_true_ptrs),
                       reinterpret_cast<float *const *>(dst_false_ptrs), n_src, tile_offsets);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// inclusive prefix sum over the 256 threads of a block (wave shuffles + one LDS hop); *total = block sum
__device__ __forceinline__ unsigned block_scan_u32(unsigned v, unsigned *total) {
    __shared__ unsigned s_w[ELT_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    if (lane == 63) s_w[w] = v;
    __syncthreads();
    unsigned add = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < ELT_THREADS / 64; ++i) {
        if (i < w) add += s_w[i];
        tot += s_w[i];
    }
    __syncthreads();
    *total = tot;
    return v + add;
}

// Index lists for the gather mode of the streaming passes (svdq_compress_gather): the ascending flat positions
// of the selected (and, when asked, of the unselected) elements of every parameter.  Same tile scan as the
// compaction; 4 B written per element instead of 4 N B read + 4 N B written.
__device__ void maskset_index_tile(int gtile, const int32_t *__restrict__ tile_param,
                                                               const int32_t *__restrict__ tile_begin,
                                                               const int64_t *__restrict__ numel_tab,
                                                               const uint8_t *const *__restrict__ masks,
                                                               int32_t *const *__restrict__ idx_true,
                                                               int32_t *const *__restrict__ idx_false,
                                                               const unsigned long long *__restrict__ tile_offsets) {
    __shared__ int32_t lt[MASK_TILE], lf[MASK_TILE];
    const int q = tile_param[gtile];
    const int64_t tile = gtile - tile_begin[q], numel = numel_tab[q];
    const uint8_t *mask = masks[q];
    const int tid = threadIdx.x;
    const int64_t tbase = tile * MASK_TILE;
    const int64_t base = tbase + (int64_t)tid * 8;
    const int lim = base < numel ? (int)((numel - base) < 8 ? (numel - base) : 8) : 0;
    const unsigned long long w = lim ? load_mask8(mask + base, lim) : 0;
    unsigned sel = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) sel |= (unsigned)((w >> (8 * e)) & 1) << e;
    const unsigned cnt = __popc(sel);
    unsigned tot_true;
    const unsigned incl = block_scan_u32(cnt, &tot_true);
    const int tile_n = (int)((numel - tbase) < MASK_TILE ? (numel - tbase) : MASK_TILE);
    const unsigned tot_false = (unsigned)tile_n - tot_true;
    unsigned jt = incl - cnt, jf = (unsigned)(tid * 8) - jt;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e < lim) {
            if (sel & (1u << e))
                lt[jt++] = (int32_t)(base + e);
            else
                lf[jf++] = (int32_t)(base + e);
        }
    }
    __syncthreads();
    const unsigned long long t_off = tile_offsets[gtile];
    int32_t *dt = idx_true[q] + t_off;
    for (unsigned i = tid; i < tot_true; i += ELT_THREADS) dt[i] = lt[i];
    if (idx_false) {
        int32_t *df = idx_false[q] + ((unsigned long long)tbase - t_off);
        for (unsigned i = tid; i < tot_false; i += ELT_THREADS) df[i] = lf[i];
    }
    __syncthreads();   // the LDS images are reused by the block's next tile
}

// MASK_TPB consecutive tiles per block: these kernels do a few hundred bytes of work per thread and tile, so the
// block start-up dominates when every tile is its own block
#define MASK_TPB 4
__global__ __launch_bounds__(ELT_THREADS) void k_maskset_index(const int32_t *__restrict__ tile_param,
                                                               const int32_t *__restrict__ tile_begin,
                                                               const int64_t *__restrict__ numel_tab,
                                                               const uint8_t *const *__restrict__ masks,
                                                               int32_t *const *__restrict__ idx_true,
                                                               int32_t *const *__restrict__ idx_false,
                                                               const unsigned long long *__restrict__ tile_offsets,
                                                               int n_tiles) {
    for (int i = 0; i < MASK_TPB; ++i) {
        const int gtile = blockIdx.x * MASK_TPB + i;
        if (gtile >= n_tiles) break;
        maskset_index_tile(gtile, tile_param, tile_begin, numel_tab, masks, idx_true, idx_false, tile_offsets);
    }
}

extern "C" int svdq_maskset_indices(const svdq_maskset *ms, const void *mask_ptrs, const void *idx_true_ptrs,
                                    const void *idx_false_ptrs, int64_t *count_true, int64_t *count_false, void *work,
                                    void *stream) {
    if (!ms || !mask_ptrs || !idx_true_ptrs || !count_true || !work) {
        svdq_set_error("svdq_maskset_indices: bad argument");
        return SVDQ_EINVAL;
    }
    for (int q = 0; q < ms->n_params; ++q)
        if (ms->h_numel[q] > 0x7fffffffLL) {
            svdq_set_error("svdq_maskset_indices: parameter %d has more than 2^31-1 elements", q);
            return SVDQ_EUNSUPPORTED;
        }
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets =
        reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ms->n_tiles * 4, 256));
    auto mp = reinterpret_cast<const uint8_t *const *>(mask_ptrs);
    hipLaunchKernelGGL(k_maskset_count, dim3(ms->n_tiles), dim3(ELT_THREADS), 0, st, ms->d_tile_param, ms->d_tile_begin,
                       ms->d_numel, mp, tile_counts);
    hipLaunchKernelGGL(k_maskset_scan, dim3(ms->n_params), dim3(1024), 0, st, ms->d_tile_begin, ms->d_numel, tile_counts,
                       tile_offsets, reinterpret_cast<long long *>(count_true),
                       reinterpret_cast<long long *>(count_false));
    hipLaunchKernelGGL(k_maskset_index, dim3((ms->n_tiles + MASK_TPB - 1) / MASK_TPB), dim3(ELT_THREADS), 0, st, ms->d_tile_param, ms->d_tile_begin,
                       ms->d_numel, mp, reinterpret_cast<int32_t *const *>(idx_true_ptrs),
                       reinterpret_cast<int32_t *const *>(idx_false_ptrs), tile_offsets, ms->n_tiles);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_maskset_combine_indices(const svdq_maskset *ms, const void *mask_ptrs, int32_t n_masks,
                                            int32_t strategy, const void *out_ptrs, const void *idx_true_ptrs,
                                            const void *idx_false_ptrs, int64_t *count_true, int64_t *count_false,
                                            void *work, void *stream) {
    if (!ms || !mask_ptrs || !out_ptrs || !idx_true_ptrs || !count_true || !work || n_masks < 1) {
        svdq_set_error(n_masks < 1 ? "Empty mask list" : "svdq_maskset_combine_indices: bad argument");
        return SVDQ_EINVAL;
    }
    if (strategy < 0 || strategy > 2) {
        svdq_set_error("Unknown mask strategy: %d", strategy);
        return SVDQ_EINVAL;
    }
    for (int q = 0; q < ms->n_params; ++q)
        if (ms->h_numel[q] > 0x7fffffffLL) {
            svdq_set_error("svdq_maskset_combine_indices: parameter %d has more than 2^31-1 elements", q);
            return SVDQ_EUNSUPPORTED;
        }
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets =
        reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ms->n_tiles * 4, 256));
    hipLaunchKernelGGL(k_maskset_combine, dim3(ms->n_tiles), dim3(ELT_THREADS), 0, st, ms->d_tile_param,
                       ms->d_tile_begin, ms->d_numel, reinterpret_cast<const uint8_t *const *>(mask_ptrs), n_masks,
                       strategy, reinterpret_cast<uint8_t *const *>(out_ptrs), (unsigned long long *)nullptr,
                       tile_counts);
    hipLaunchKernelGGL(k_maskset_scan, dim3(ms->n_params), dim3(1024), 0, st, ms->d_tile_begin, ms->d_numel, tile_counts,
                       tile_offsets, reinterpret_cast<long long *>(count_true),
                       reinterpret_cast<long long *>(count_false));
    hipLaunchKernelGGL(k_maskset_index, dim3((ms->n_tiles + MASK_TPB - 1) / MASK_TPB), dim3(ELT_THREADS), 0, st, ms->d_tile_param, ms->d_tile_begin,
                       ms->d_numel, reinterpret_cast<const uint8_t *const *>(out_ptrs),
                       reinterpret_cast<int32_t *const *>(idx_true_ptrs),
                       reinterpret_cast<int32_t *const *>(idx_false_ptrs), tile_offsets, ms->n_tiles);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// Tall masks as they are distributed (mask_loader.py:125-206): ONE bit per element of the flattened state dict,
// numpy.packbits order (first element = most significant bit), one stream per task.  Combining them from the
// packed form reads N/8 bytes per element instead of N: parameter q's elements are bits bit_off[q] + e of every
// stream.  Output: the combined mask as bool bytes (what the reference hands on), tile counts for the index build.
__device__ void maskset_combine_packed_tile(int gtile, const int32_t *__restrict__ tile_param,
                                                                        const int32_t *__restrict__ tile_begin,
                                                                        const int64_t *__restrict__ numel_tab,
                                                                        const uint8_t *const *__restrict__ streams,
                                                                        const int64_t *__restrict__ bit_off,
                                                                        const int64_t *__restrict__ stream_bytes,
                                                                        int n_masks, int strategy,
                                                                        uint8_t *const *__restrict__ outs,
                                                                        unsigned *__restrict__ tile_counts) {
    // the tile's packed bytes of every stream are fetched with 16-byte loads into LDS first (17 chunks cover the 257
    // bytes a 2048-element tile can touch at an arbitrary bit offset); byte loads per thread would waste the bus
    __shared__ __attribute__((aligned(16))) uint8_t S[SVDQ_MAX_TASKS][17 * 16];
    const int q = tile_param[gtile];
    const int64_t numel = numel_tab[q];
    const int64_t tile0 = (int64_t)(gtile - tile_begin[q]) * MASK_TILE;
    const int64_t base = tile0 + (int64_t)threadIdx.x * 8;
    const int64_t c0 = ((bit_off[q] + tile0) >> 3) & ~(int64_t)15;   // first 16-byte chunk of this tile in every stream
    for (int i = threadIdx.x; i < n_masks * 17; i += ELT_THREADS) {
        const int m = i / 17, c = i - m * 17;
        const int64_t b = c0 + 16 * c;
        const int64_t nb = stream_bytes[m];
        uint4 v = {0u, 0u, 0u, 0u};
        if (b + 16 <= nb) {
            v = *reinterpret_cast<const uint4 *>(streams[m] + b);
        } else if (b < nb) {
            uint8_t tmp[16];
            for (int k = 0; k < 16; ++k) tmp[k] = (b + k < nb) ? streams[m][b + k] : (uint8_t)0;
            v = *reinterpret_cast<const uint4 *>(tmp);
        }
        *reinterpret_cast<uint4 *>(&S[m][16 * c]) = v;
    }
    __syncthreads();
    unsigned cnt = 0;
    if (base < numel) {
        const int lim = (int)((numel - base) < 8 ? (numel - base) : 8);
        const int64_t pos = bit_off[q] + base;   // bit position of this thread's first element in every stream
        const int byte = (int)((pos >> 3) - c0);
        const int sh = (int)(pos & 7);
        unsigned char cnts[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) cnts[j] = 0;
        unsigned any = 0, all = 0xff;
        for (int m = 0; m < n_masks; ++m) {
            const unsigned w = ((unsigned)S[m][byte] << 8) | S[m][byte + 1];   // byte + 1 <= 16 * 17 - 1: in the image
            const unsigned bits = (w >> (8 - sh)) & 0xff;   // 8 elements, first element = bit 7
            any |= bits;
            all &= bits;
            if (strategy == SVDQ_MASK_MAJORITY) {
#pragma unroll
                for (int j = 0; j < 8; ++j) cnts[j] += (bits >> (7 - j)) & 1;
            }
        }
        unsigned long long res = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned bit;
            if (strategy == SVDQ_MASK_UNION)
                bit = (any >> (7 - j)) & 1;
            else if (strategy == SVDQ_MASK_INTERSECTION)
                bit = (all >> (7 - j)) & 1;
            else
                bit = 2 * cnts[j] >= n_masks;
            if (j < lim) res |= (unsigned long long)bit << (8 * j);
        }
        uint8_t *out = outs[q];
        if (lim == 8 && (reinterpret_cast<uintptr_t>(out + base) & 7) == 0) {
            *reinterpret_cast<unsigned long long *>(out + base) = res;
        } else {
            for (int j = 0; j < lim; ++j) out[base + j] = (uint8_t)((res >> (8 * j)) & 1);
        }
        cnt = (unsigned)__popcll(res);
    }
    const unsigned tot = block_sum_u32(cnt);
    if (threadIdx.x == 0) tile_counts[gtile] = tot;
    __syncthreads();   // the LDS image is reused by the block's next tile
}

__global__ __launch_bounds__(ELT_THREADS) void k_maskset_combine_packed(const int32_t *__restrict__ tile_param,
                                                                        const int32_t *__restrict__ tile_begin,
                                                                        const int64_t *__restrict__ numel_tab,
                                                                        const uint8_t *const *__restrict__ streams,
                                                                        const int64_t *__restrict__ bit_off,
                                                                        const int64_t *__restrict__ stream_bytes,
                                                                        int n_masks, int strategy,
                                                                        uint8_t *const *__restrict__ outs,
                                                                        unsigned *__restrict__ tile_counts, int n_tiles) {
    for (int i = 0; i < MASK_TPB; ++i) {
        const int gtile = blockIdx.x * MASK_TPB + i;
        if (gtile >= n_tiles) break;
        maskset_combine_packed_tile(gtile, tile_param, tile_begin, numel_tab, streams, bit_off, stream_bytes, n_masks,
                                    strategy, outs, tile_counts);
    }
}

extern "C" int svdq_maskset_combine_packed_indices(const svdq_maskset *ms, const void *stream_ptrs,
                                                   const int64_t *stream_bytes, const int64_t *bit_offsets,
                                                   int32_t n_masks, int32_t strategy, const void *out_ptrs,
                                                   const void *idx_true_ptrs, const void *idx_false_ptrs,
                                                   int64_t *count_true, int64_t *count_false, void *work,
                                                   void *stream) {
    if (!ms || !stream_ptrs || !stream_bytes || !bit_offsets || !out_ptrs || !idx_true_ptrs || !count_true || !work ||
        n_masks < 1) {
        svdq_set_error(n_masks < 1 ? "Empty mask list" : "svdq_maskset_combine_packed_indices: bad argument");
        return SVDQ_EINVAL;
    }
    if (n_masks > SVDQ_MAX_TASKS) {
        svdq_set_error("at most %d packed mask streams are supported, got %d", SVDQ_MAX_TASKS, n_masks);
        return SVDQ_EINVAL;
    }
    if (strategy < 0 || strategy > 2) {
        svdq_set_error("Unknown mask strategy: %d", strategy);
        return SVDQ_EINVAL;
    }
    for (int q = 0; q < ms->n_params; ++q)
        if (ms->h_numel[q] > 0x7fffffffLL) {
            svdq_set_error("svdq_maskset_combine_packed_indices: parameter %d has more than 2^31-1 elements", q);
            return SVDQ_EUNSUPPORTED;
        }
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets =
        reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ms->n_tiles * 4, 256));
    hipLaunchKernelGGL(k_maskset_combine_packed, dim3((ms->n_tiles + MASK_TPB - 1) / MASK_TPB), dim3(ELT_THREADS), 0, st, ms->d_tile_param,
                       ms->d_tile_begin, ms->d_numel, reinterpret_cast<const uint8_t *const *>(stream_ptrs), bit_offsets,
                       stream_bytes, n_masks, strategy, reinterpret_cast<uint8_t *const *>(out_ptrs), tile_counts, ms->n_tiles);
    hipLaunchKernelGGL(k_maskset_scan, dim3(ms->n_params), dim3(1024), 0, st, ms->d_tile_begin, ms->d_numel, tile_counts,
                       tile_offsets, reinterpret_cast<long long *>(count_true),
                       reinterpret_cast<long long *>(count_false));
    hipLaunchKernelGGL(k_maskset_index, dim3((ms->n_tiles + MASK_TPB - 1) / MASK_TPB), dim3(ELT_THREADS), 0, st, ms->d_tile_param, ms->d_tile_begin,
                       ms->d_numel, reinterpret_cast<const uint8_t *const *>(out_ptrs),
                       reinterpret_cast<int32_t *const *>(idx_true_ptrs),
                       reinterpret_cast<int32_t *const *>(idx_false_ptrs), tile_offsets, ms->n_tiles);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// ------------------------------------------------------------------------------------ walk mode: unit starts
// The mask-walk mode of the streaming passes (svdq_compress_masked, svdq_stream.h "walk mode") needs, for every work
// unit of the plan, the SOURCE position of the unit's first row -- the element of rank unit.row0 among the selected
// (or, for the noise region, the cleared) elements of its mask.  One wavefront per unit: a 64-ary search over the
// exclusive tile offsets of the scan, then one pass over the 2048 mask bytes of the tile that holds the element.
// entry_map: NULL (plan parameter p uses mask p, selected elements) or [n_params] int32: low 31 bits = which mask of the
// set, bit 31 = the cleared elements.  Units past rows_dev[p] get numel (nothing to walk).  Bit 62 of every start
// carries the polarity to the streaming kernels.
#define SVDQ_WALK_INV (1ll << 62)
__global__ __launch_bounds__(64) void k_maskset_unit_starts(const SvdqParam *__restrict__ params,
                                                            const SvdqUnit *__restrict__ units,
                                                            const int64_t *__restrict__ rows_dev,
                                                            const int32_t *__restrict__ entry_map,
                                                            const int32_t *__restrict__ tile_begin,
                                                            const int64_t *__restrict__ numel_tab,
                                                            const uint8_t *const *__restrict__ masks,
                                                            const unsigned long long *__restrict__ tile_offsets,
                                                            int64_t *__restrict__ ustart) {
    const int u = blockIdx.x, lane = threadIdx.x;
    const SvdqUnit ud = units[u];
    const int p = ud.param;
    const unsigned em = entry_map ? (unsigned)entry_map[p] : (unsigned)p;
    const int q = (int)(em & 0x7fffffffu), inv = (int)(em >> 31);
    const int64_t numel = numel_tab[q];
    const int64_t tag = inv ? SVDQ_WALK_INV : 0;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    if (ud.row0 >= D) {
        if (lane == 0) ustart[u] = numel | tag;
        return;
    }
    const unsigned long long target = (unsigned long long)ud.row0;
    const int t0 = tile_begin[q], nt = tile_begin[q + 1] - t0;
    // off(i) = selected elements in front of tile i
    auto off = [&](int i) -> unsigned long long {
        const unsigned long long t = tile_offsets[t0 + i];
        return inv ? (unsigned long long)i * MASK_TILE - t : t;
    };
    int lo = 0, hi = nt;   // invariant: off(lo) <= target, the answer is the largest such tile in [lo, hi)
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) / 64;
        const int i = lo + lane * step;
        const bool ok = i < hi && off(i) <= target;
        const int j = (int)__popcll(__ballot(ok)) - 1;   // off is monotone: lanes 0..j pass, lane 0 always does
        const int nlo = lo + (j < 0 ? 0 : j) * step;
        hi = (nlo + step < hi) ? nlo + step : hi;
        lo = nlo;
    }
    const unsigned rem = (unsigned)(target - off(lo));   // rank inside the tile
    const int64_t tb = (int64_t)lo * MASK_TILE;
    const uint8_t *m = masks[q] + tb + 32 * lane;
    unsigned bits = 0;
    const int64_t left = numel - (tb + 32 * lane);
    if (left >= 32 && (reinterpret_cast<uintptr_t>(m) & 7) == 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned long long x = reinterpret_cast<const unsigned long long *>(m)[w];
#pragma unroll
            for (int e = 0; e < 8; ++e) bits |= (unsigned)(((x >> (8 * e)) & 0xff) != 0) << (8 * w + e);
        }
    } else {
        for (int e = 0; e < 32; ++e)
            if (e < left) bits |= (unsigned)(m[e] != 0) << e;
    }
    if (inv) {
        bits = ~bits;
        if (left < 32) bits &= left > 0 ? ((1u << left) - 1u) : 0u;
    }
    const unsigned cnt = __popc(bits);
    unsigned incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned excl = incl - cnt;
    const bool mine = excl <= rem && rem < incl;
    if (mine) {
        unsigned b = bits;
        for (unsigned i = excl; i < rem; ++i) b &= b - 1;   // drop the set bits in front of ours
        ustart[u] = (tb + 32 * lane + (__ffs(b) - 1)) | tag;
    }
    if (__ballot(mine) == 0 && lane == 0) ustart[u] = numel | tag;   // counts and mask disagree: nothing to walk
}

static int maskset_check_plan(const svdq_maskset *ms, const svdq_plan *pl, const char *who, bool identity) {
    if (!ms || !pl) {
        svdq_set_error("%s: bad argument", who);
        return SVDQ_EINVAL;
    }
    if (identity) {
        if (pl->n_params != ms->n_params) {
            svdq_set_error("%s: the plan has %d parameters, the mask set %d", who, pl->n_params, ms->n_params);
            return SVDQ_EINVAL;
        }
        for (int q = 0; q < ms->n_params; ++q)
            if (pl->h_params[q].rows != ms->h_numel[q]) {
                svdq_set_error("Shape mismatch: parameter %d has %lld elements in the plan and %lld in the mask set", q,
                               (long long)pl->h_params[q].rows, (long long)ms->h_numel[q]);
                return SVDQ_EINVAL;
            }
    }
    return SVDQ_OK;
}

static void launch_unit_starts(const svdq_maskset *ms, const svdq_plan *pl, const uint8_t *const *mp,
                               const int32_t *entry_map, const int64_t *rows_dev,
                               const unsigned long long *tile_offsets, int64_t *unit_start, hipStream_t st) {
    hipLaunchKernelGGL(k_maskset_unit_starts, dim3(pl->n_units), dim3(64), 0, st, pl->d_params, pl->d_units, rows_dev,
                       entry_map, ms->d_tile_begin, ms->d_numel, mp, tile_offsets, unit_start);
}

extern "C" int svdq_maskset_count_scan(const svdq_maskset *ms, const void *mask_ptrs, int64_t *count_true,
                                       int64_t *count_false, void *work, void *stream) {
    if (!ms || !mask_ptrs || !count_true || !work) {
        svdq_set_error("svdq_maskset_count_scan: bad argument");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets =
        reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ms->n_tiles * 4, 256));
    hipLaunchKernelGGL(k_maskset_count, dim3(ms->n_tiles), dim3(ELT_THREADS), 0, st, ms->d_tile_param, ms->d_tile_begin,
                       ms->d_numel, reinterpret_cast<const uint8_t *const *>(mask_ptrs), tile_counts);
    hipLaunchKernelGGL(k_maskset_scan, dim3(ms->n_params), dim3(1024), 0, st, ms->d_tile_begin, ms->d_numel, tile_counts,
                       tile_offsets, reinterpret_cast<long long *>(count_true),
                       reinterpret_cast<long long *>(count_false));
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_maskset_unit_starts(const svdq_maskset *ms, const svdq_plan *pl, const void *mask_ptrs,
                                        const int32_t *entry_map, const int64_t *rows_dev, const void *work,
                                        int64_t *unit_start, void *stream) {
    if (!mask_ptrs || !rows_dev || !work || !unit_start) {
        svdq_set_error("svdq_maskset_unit_starts: bad argument");
        return SVDQ_EINVAL;
    }
    if (int rc = maskset_check_plan(ms, pl, "svdq_maskset_unit_starts", entry_map == nullptr)) return rc;
    const uint8_t *wb = reinterpret_cast<const uint8_t *>(work);
    auto tile_offsets = reinterpret_cast<const unsigned long long *>(wb + svdq_align_up((int64_t)ms->n_tiles * 4, 256));
    launch_unit_starts(ms, pl, reinterpret_cast<const uint8_t *const *>(mask_ptrs), entry_map, rows_dev, tile_offsets,
                       unit_start, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_maskset_combine_starts(const svdq_maskset *ms, const svdq_plan *pl, const void *mask_ptrs,
                                           int32_t n_masks, int32_t strategy, const void *out_ptrs, int64_t *count_true,
                                           int64_t *count_false, void *work, int64_t *unit_start, void *stream) {
    if (!mask_ptrs || !out_ptrs || !count_true || !work || !unit_start || n_masks < 1) {
        svdq_set_error(n_masks < 1 ? "Empty mask list" : "svdq_maskset_combine_starts: bad argument");
        return SVDQ_EINVAL;
    }
    if (strategy < 0 || strategy > 2) {
        svdq_set_error("Unknown mask strategy: %d", strategy);
        return SVDQ_EINVAL;
    }
    if (int rc = maskset_check_plan(ms, pl, "svdq_maskset_combine_starts", true)) return rc;
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets =
        reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ms->n_tiles * 4, 256));
    auto op = reinterpret_cast<uint8_t *const *>(out_ptrs);
    hipLaunchKernelGGL(k_maskset_combine, dim3(ms->n_tiles), dim3(ELT_THREADS), 0, st, ms->d_tile_param,
                       ms->d_tile_begin, ms->d_numel, reinterpret_cast<const uint8_t *const *>(mask_ptrs), n_masks,
                       strategy, op, (unsigned long long *)nullptr, tile_counts);
    hipLaunchKernelGGL(k_maskset_scan, dim3(ms->n_params), dim3(1024), 0, st, ms->d_tile_begin, ms->d_numel, tile_counts,
                       tile_offsets, reinterpret_cast<long long *>(count_true),
                       reinterpret_cast<long long *>(count_false));
    launch_unit_starts(ms, pl, reinterpret_cast<const uint8_t *const *>(out_ptrs), nullptr, count_true, tile_offsets,
                       unit_start, st);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_maskset_combine_packed_starts(const svdq_maskset *ms, const svdq_plan *pl, const void *stream_ptrs,
                                                  const int64_t *stream_bytes, const int64_t *bit_offsets,
                                                  int32_t n_masks, int32_t strategy, const void *out_ptrs,
                                                  int64_t *count_true, int64_t *count_false, void *work,
                                                  int64_t *unit_start, void *stream) {
    if (!stream_ptrs || !stream_bytes || !bit_offsets || !out_ptrs || !count_true || !work || !unit_start ||
        n_masks < 1) {
        svdq_set_error(n_masks < 1 ? "Empty mask list" : "svdq_maskset_combine_packed_starts: bad argument");
        return SVDQ_EINVAL;
    }
    if (n_masks > SVDQ_MAX_TASKS) {
        svdq_set_error("at most %d packed mask streams are supported, got %d", SVDQ_MAX_TASKS, n_masks);
        return SVDQ_EINVAL;
    }
    if (strategy < 0 || strategy > 2) {
        svdq_set_error("Unknown mask strategy: %d", strategy);
        return SVDQ_EINVAL;
    }
    if (int rc = maskset_check_plan(ms, pl, "svdq_maskset_combine_packed_starts", true)) return rc;
    hipStream_t st = (hipStream_t)stream;
    uint8_t *wb = reinterpret_cast<uint8_t *>(work);
    unsigned *tile_counts = reinterpret_cast<unsigned *>(wb);
    unsigned long long *tile_offsets =
        reinterpret_cast<unsigned long long *>(wb + svdq_align_up((int64_t)ms->n_tiles * 4, 256));
    hipLaunchKernelGGL(k_maskset_combine_packed, dim3((ms->n_tiles + MASK_TPB - 1) / MASK_TPB), dim3(ELT_THREADS), 0, st,
                       ms->d_tile_param, ms->d_tile_begin, ms->d_numel,
                       reinterpret_cast<const uint8_t *const *>(stream_ptrs), bit_offsets, stream_bytes, n_masks,
                       strategy, reinterpret_cast<uint8_t *const *>(out_ptrs), tile_counts, ms->n_tiles);
    hipLaunchKernelGGL(k_maskset_scan, dim3(ms->n_params), dim3(1024), 0, st, ms->d_tile_begin, ms->d_numel, tile_counts,
                       tile_offsets, reinterpret_cast<long long *>(count_true),
                       reinterpret_cast<long long *>(count_false));
    launch_unit_starts(ms, pl, reinterpret_cast<const uint8_t *const *>(out_ptrs), nullptr, count_true, tile_offsets,
                       unit_start, st);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// ------------------------------------------------------------------------------------ diagnostics
// compute_reconstruction_error (diagnostics.py:72-117) fused with the reconstruction it is applied to in
// compute_parameter_diagnostics (diagnostics.py:205-215): rec = U_high c_high + U_low c_low (+ mean when
// given -- the reference's diagnostics do NOT add it back) is never materialised.
// out[6] = {||x-rec||, ||x-rec||/||x|| (0 when ||x|| <= 1e-10), max|x-rec|, mean|x-rec|, ||x||, ||rec||}
struct ErrPart {
    double se, sx, sr, sa, mx;
};

template <typename T>
__global__ __launch_bounds__(ELT_THREADS) void k_recon_error(const T *__restrict__ uh, const T *__restrict__ ul,
                                                             int64_t rows, int k, int nl,
                                                             const float *__restrict__ coef,
                                                             const float *__restrict__ mean,
                                                             const float *__restrict__ recon_in,
                                                             const float *__restrict__ orig,
                                                             ErrPart *__restrict__ part) {
    __shared__ float c[32];
    if (threadIdx.x < k + nl) c[threadIdx.x] = coef[threadIdx.x];
    __syncthreads();
    double se = 0.0, sx = 0.0, sr = 0.0, sa = 0.0;
    float mx = 0.f;
    const int64_t stride = (int64_t)gridDim.x * ELT_THREADS;
    for (int64_t d = (int64_t)blockIdx.x * ELT_THREADS + threadIdx.x; d < rows; d += stride) {
        float rec;
        if (recon_in) {
            rec = recon_in[d];
        } else {
            float hi = 0.f, lo = 0.f;
            for (int i = 0; i < k; ++i) hi = fmaf(u_load(uh, d * k + i), c[i], hi);
            for (int j = 0; j < nl; ++j) lo = fmaf(u_load(ul, d * nl + j), c[k + j], lo);
            rec = __fadd_rn(hi, lo);
            if (mean) rec = __fadd_rn(rec, mean[d]);
        }
        const float x = orig[d];
        const float e = __fsub_rn(x, rec);
        se += (double)e * e;
        sx += (double)x * x;
        sr += (double)rec * rec;
        sa += fabs((double)e);
        mx = fmaxf(mx, fabsf(e));
        if (e != e) mx = e;  // NaN propagates like torch.max
    }
    __shared__ double r0[ELT_THREADS], r1[ELT_THREADS], r2[ELT_THREADS], r3[ELT_THREADS], r4[ELT_THREADS];
    const int tid = threadIdx.x;
    r0[tid] = se; r1[tid] = sx; r2[tid] = sr; r3[tid] = sa; r4[tid] = (double)mx;
    __syncthreads();
    for (int off = ELT_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) {
            r0[tid] += r0[tid + off]; r1[tid] += r1[tid + off]; r2[tid] += r2[tid + off]; r3[tid] += r3[tid + off];
            const double a = r4[tid], b = r4[tid + off];
            r4[tid] = (a != a) ? a : ((b != b) ? b : (b > a ? b : a));
        }
        __syncthreads();
    }
    if (tid == 0) {
        part[blockIdx.x].se = r0[0]; part[blockIdx.x].sx = r1[0]; part[blockIdx.x].sr = r2[0];
        part[blockIdx.x].sa = r3[0]; part[blockIdx.x].mx = r4[0];
    }
}

__global__ __launch_bounds__(64) void k_recon_error_finish(const ErrPart *__restrict__ part, int nblk, int64_t rows,
                                                           double *__restrict__ out) {
    if (threadIdx.x != 0) return;
    double se = 0.0, sx = 0.0, sr = 0.0, sa = 0.0, mx = 0.0;
    for (int b = 0; b < nblk; ++b) {
        se += part[b].se; sx += part[b].sx; sr += part[b].sr; sa += part[b].sa;
        const double v = part[b].mx;
        mx = (mx != mx) ? mx : ((v != v) ? v : (v > mx ? v : mx));
    }
    // the reference works on fp32 tensors: norms are fp32 values
    const float en = (float)sqrt(se), on = (float)sqrt(sx);
    out[0] = (double)en;
    out[1] = on > 1e-10f ? (double)en / (double)on : 0.0;
    out[2] = mx;
    out[3] = (double)(float)(sa / (double)rows);
    out[4] = (double)on;
    out[5] = (double)(float)sqrt(sr);
}

extern "C" int64_t svdq_recon_error_work_bytes(int64_t rows) { return (int64_t)project_grid(rows) * sizeof(ErrPart) + 256; }

extern "C" int svdq_recon_error(const void *u_high, const void *u_low, int32_t u_fp16, int64_t rows, int32_t k,
                                int32_t nl, const float *coef, const float *mean, const float *recon,
                                const float *orig, double *out6, void *work, void *stream) {
    const bool fused = recon == nullptr;
    if (rows < 1 || !orig || !out6 || !work || (fused && (k < 0 || nl < 0 || k + nl > 32 || (k + nl > 0 && !coef) ||
                                                          (k > 0 && !u_high) || (nl > 0 && !u_low)))) {
        svdq_set_error("svdq_recon_error: bad argument");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int grid = project_grid(rows);
    ErrPart *part = reinterpret_cast<ErrPart *>(work);
    if (!fused) {
        k = 0;
        nl = 0;
    }
    if (u_fp16 && fused)
        hipLaunchKernelGGL((k_recon_error<__half>), dim3(grid), dim3(ELT_THREADS), 0, st,
                           reinterpret_cast<const __half *>(u_high), reinterpret_cast<const __half *>(u_low), rows, k,
                           nl, coef, mean, recon, orig, part);
    else
        hipLaunchKernelGGL((k_recon_error<float>), dim3(grid), dim3(ELT_THREADS), 0, st,
                           reinterpret_cast<const float *>(u_high), reinterpret_cast<const float *>(u_low), rows, k, nl,
                           coef, mean, recon, orig, part);
    hipLaunchKernelGGL(k_recon_error_finish, dim3(1), dim3(64), 0, st, part, grid, rows, out6);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

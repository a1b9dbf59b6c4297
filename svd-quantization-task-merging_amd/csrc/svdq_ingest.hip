// svdq_ingest.hip -- the step before the hot path, batched over a plan's
// ragged parameter list x N tasks with the same unit table the streaming passes use:
//
//   k_ingest        delta[p][t][i] = finetuned[p][t][i] - base[p][i]     (task_vector_loader.py:103-141,
//                   task_vectors.py TaskVector.__init__): base is read ONCE for the N tasks, the
//                   deltas land directly in the per-task buffers the compressor's pointer table names.
//                   Optionally emits the per-unit min / max of each delta so that whole-tensor
//                   quantization ("TVQ") needs no extra statistics pass.
//   k_tvq_stats     the same statistics for tensors that are already resident
//   k_tvq_params    per (parameter, task): scale / zero-point of asymmetric_quantization
//                   (quantization_utils.py:76-99) or absmax_quantization (:60-73)
//   k_tvq_apply     codes[p][t][i]  (uint8 asymmetric, int8 absmax)
//   k_tvq_dequant   dequantize_asymmetric / dequantize_absmax (:137-172, :102-134), optionally fused
//                   with "+ base" (QuantizedTaskVector.apply_to, QuantizedBaseAndTaskVector.dequantize)
//
// All of it is HBM-bound elementwise work: one wavefront per unit (run of 256-row blocks), 16-byte
// accesses, every input byte read once.  Compiled with -ffp-contract=off; the quantizer arithmetic
// uses explicit round-to-nearest intrinsics so that codes are bit-identical to the reference.

#include "svdq_common.h"

typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));

struct TvqPartial {
    float mn, mx;
    int32_t has_nan, pad;
};

#define AS1(T, p) ((const __attribute__((address_space(1))) T *)(p))

__device__ __forceinline__ void mm_acc(float x, float &mn, float &mx, int &has_nan) {
    if (x != x) has_nan = 1;
    mn = x < mn ? x : mn;
    mx = x > mx ? x : mx;
}

__device__ __forceinline__ void wave_minmax(float &mn, float &mx, int &has_nan) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
        const int c = __shfl_xor(has_nan, off);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        has_nan |= c;
    }
}

// ------------------------------------------------------------------------------------ ingest
template <bool STATS>
__global__ __launch_bounds__(64) void k_ingest(const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
                                               int NT, const float *const *__restrict__ base_ptrs,
                                               const float *const *__restrict__ ft_ptrs,
                                               float *const *__restrict__ delta_ptrs, TvqPartial *__restrict__ part) {
    const int uidx = blockIdx.x, lane = threadIdx.x;
    const SvdqUnit u = units[uidx];
    const int p = u.param;
    const int64_t D = params[p].rows;
    const float *base = base_ptrs[p];
    const int64_t r0 = u.row0, r1 = r0 + u.nrows;  // r1 <= D
    const int64_t v1 = (r1 == D) ? (D & ~(int64_t)3) : r1;  // vector part ends at a multiple of 4
    for (int t0 = 0; t0 < NT; t0 += 8) {
        const int tn = NT - t0 < 8 ? NT - t0 : 8;
        const float *ft[8];
        float *dl[8];
        float mn[8], mx[8];
        int nan[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + (j < tn ? j : 0);
            ft[j] = ft_ptrs[(size_t)p * NT + t];
            dl[j] = delta_ptrs[(size_t)p * NT + t];
            mn[j] = __builtin_inff();
            mx[j] = -__builtin_inff();
            nan[j] = 0;
        }
        for (int64_t i = r0 + 4 * lane; i < v1; i += 256) {
            const f32x4 b = *AS1(f32x4, base + i);
            f32x4 f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < tn) f[j] = *AS1(f32x4, ft[j] + i);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < tn) {
                    const f32x4 d = f[j] - b;
                    *reinterpret_cast<f32x4 *>(dl[j] + i) = d;
                    if (STATS) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) mm_acc(d[e], mn[j], mx[j], nan[j]);
                    }
                }
        }
        if (r1 == D && lane < (int)(D & 3) && v1 >= r0) {  // scalar tail of the parameter's last unit
            const int64_t i = v1 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < tn) {
                    const float d = ft[j][i] - base[i];
                    dl[j][i] = d;
                    if (STATS) mm_acc(d, mn[j], mx[j], nan[j]);
                }
        }
        if (STATS) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < tn) {
                    wave_minmax(mn[j], mx[j], nan[j]);
                    if (lane == 0) part[(size_t)uidx * NT + t0 + j] = TvqPartial{mn[j], mx[j], nan[j], 0};
                }
        }
    }
}

// ------------------------------------------------------------------------------------ TVQ
__global__ __launch_bounds__(64) void k_tvq_stats(const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
                                                  int NT, const float *const *__restrict__ x_ptrs,
                                                  TvqPartial *__restrict__ part) {
    const int uidx = blockIdx.x, lane = threadIdx.x;
    const SvdqUnit u = units[uidx];
    const int p = u.param;
    const int64_t D = params[p].rows;
    const int64_t r0 = u.row0, r1 = r0 + u.nrows;
    const int64_t v1 = (r1 == D) ? (D & ~(int64_t)3) : r1;
    for (int t = 0; t < NT; ++t) {
        const float *x = x_ptrs[(size_t)p * NT + t];
        float mn = __builtin_inff(), mx = -__builtin_inff();
        int nan = 0;
        int64_t i = r0 + 4 * lane;
        for (; i + 768 < v1; i += 1024) {  // 4 independent 16-byte loads in flight per lane
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *AS1(f32x4, x + i + 256 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) mm_acc(v[q][e], mn, mx, nan);
        }
        for (; i < v1; i += 256) {
            const f32x4 v = *AS1(f32x4, x + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) mm_acc(v[e], mn, mx, nan);
        }
        if (r1 == D && lane < (int)(D & 3) && v1 >= r0) mm_acc(x[v1 + lane], mn, mx, nan);
        wave_minmax(mn, mx, nan);
        if (lane == 0) part[(size_t)uidx * NT + t] = TvqPartial{mn, mx, nan, 0};
    }
}

// one thread per (parameter, task); mode 0 = asymmetric, 1 = absmax
__global__ __launch_bounds__(64) void k_tvq_params(const SvdqParam *__restrict__ params, int nparams, int NT, int mode,
                                                   int bits, const TvqPartial *__restrict__ part,
                                                   float *__restrict__ scale_out, float *__restrict__ zp_out) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= nparams * NT) return;
    const int p = idx / NT, t = idx - p * NT;
    const SvdqParam pd = params[p];
    float mn = __builtin_inff(), mx = -__builtin_inff();
    int has_nan = 0;
    for (int u = 0; u < pd.unit_count; ++u) {
        const TvqPartial q = part[(size_t)(pd.unit_begin + u) * NT + t];
        mn = q.mn < mn ? q.mn : mn;
        mx = q.mx > mx ? q.mx : mx;
        has_nan |= q.has_nan;
    }
    if (has_nan) {  // torch min / max / abs().max() propagate NaN
        mn = __builtin_nanf("");
        mx = mn;
    }
    if (mode == 0) {
        const float qmax = (float)((1 << bits) - 1);
        // python-int / Tensor == Tensor.reciprocal() * int: two roundings (tests/golden/rtvq_cases.npz)
        const float scale = __fmul_rn(__fdiv_rn(1.0f, __fsub_rn(mx, mn)), qmax);
        scale_out[idx] = scale;
        zp_out[idx] = __fmul_rn(-1.0f, rintf(__fmul_rn(scale, mn)));
    } else {
        const float qmax = (float)((1 << (bits - 1)) - 1);
        float amax = fabsf(mn) > fabsf(mx) ? fabsf(mn) : fabsf(mx);
        if (has_nan) amax = mn;
        scale_out[idx] = __fmul_rn(__fdiv_rn(1.0f, amax), qmax);
        if (zp_out) zp_out[idx] = 0.f;
    }
}

__device__ __forceinline__ unsigned char tvq_code(float x, float scale, float zp, float qlo, float qhi, int mode) {
    float vq = mode == 0 ? rintf(__fadd_rn(__fmul_rn(scale, x), zp)) : rintf(__fmul_rn(scale, x));
    if (vq != vq) return 0;  // NaN -> 0 (reference CPU cast)
    vq = vq < qlo ? qlo : (vq > qhi ? qhi : vq);
    return mode == 0 ? (unsigned char)vq : (unsigned char)(signed char)vq;
}

// absmax with qbit = 16: int16 codes (quantization_utils.py:67-70), no clamp in the reference -- the int16 range
// only guards the cast
typedef short i16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ short tvq_code16(float x, float scale) {
    float vq = rintf(__fmul_rn(scale, x));
    if (vq != vq) return 0;
    vq = vq < -32768.f ? -32768.f : (vq > 32767.f ? 32767.f : vq);
    return (short)vq;
}

__global__ __launch_bounds__(64) void k_tvq_apply16(const SvdqParam *__restrict__ params,
                                                    const SvdqUnit *__restrict__ units, int NT,
                                                    const float *const *__restrict__ x_ptrs,
                                                    const float *__restrict__ scale_in,
                                                    short *const *__restrict__ code_ptrs) {
    const int uidx = blockIdx.x, lane = threadIdx.x;
    const SvdqUnit u = units[uidx];
    const int p = u.param;
    const int64_t D = params[p].rows;
    const int64_t r0 = u.row0, r1 = r0 + u.nrows;
    const int64_t v1 = (r1 == D) ? (D & ~(int64_t)3) : r1;
    for (int t = 0; t < NT; ++t) {
        const float *x = x_ptrs[(size_t)p * NT + t];
        short *c = code_ptrs[(size_t)p * NT + t];
        const float scale = scale_in[(size_t)p * NT + t];
        for (int64_t i = r0 + 4 * lane; i < v1; i += 256) {
            const f32x4 v = *AS1(f32x4, x + i);
            i16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = tvq_code16(v[e], scale);
            *reinterpret_cast<i16x4 *>(c + i) = o;
        }
        if (r1 == D && lane < (int)(D & 3)) c[v1 + lane] = tvq_code16(x[v1 + lane], scale);
    }
}

__global__ __launch_bounds__(64) void k_tvq_dequant16(const SvdqParam *__restrict__ params,
                                                      const SvdqUnit *__restrict__ units, int NT,
                                                      const short *const *__restrict__ code_ptrs,
                                                      const float *__restrict__ scale_in,
                                                      const float *const *__restrict__ add_ptrs,
                                                      float *const *__restrict__ out_ptrs) {
    const int uidx = blockIdx.x, lane = threadIdx.x;
    const SvdqUnit u = units[uidx];
    const int p = u.param;
    const int64_t D = params[p].rows;
    const int64_t r0 = u.row0, r1 = r0 + u.nrows;
    const int64_t v1 = (r1 == D) ? (D & ~(int64_t)3) : r1;
    const float *add = add_ptrs ? add_ptrs[p] : nullptr;
    for (int t = 0; t < NT; ++t) {
        const short *c = code_ptrs[(size_t)p * NT + t];
        float *o = out_ptrs[(size_t)p * NT + t];
        const float scale = scale_in[(size_t)p * NT + t];
        for (int64_t i = r0 + 4 * lane; i < v1; i += 256) {
            const i16x4 q = *AS1(i16x4, c + i);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __fmul_rn((float)q[e], scale);
            if (add) {
                const f32x4 a = *AS1(f32x4, add + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(a[e], v[e]);
            }
            *reinterpret_cast<f32x4 *>(o + i) = v;
        }
        if (r1 == D && lane < (int)(D & 3)) {
            const int64_t i = v1 + lane;
            float v = __fmul_rn((float)c[i], scale);
            if (add) v = __fadd_rn(add[i], v);
            o[i] = v;
        }
    }
}

__global__ __launch_bounds__(64) void k_tvq_apply(const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
                                                  int NT, int mode, int bits, const float *const *__restrict__ x_ptrs,
                                                  const float *__restrict__ scale_in, const float *__restrict__ zp_in,
                                                  uint8_t *const *__restrict__ code_ptrs) {
    const int uidx = blockIdx.x, lane = threadIdx.x;
    const SvdqUnit u = units[uidx];
    const int p = u.param;
    const int64_t D = params[p].rows;
    const int64_t r0 = u.row0, r1 = r0 + u.nrows;
    const int64_t v1 = (r1 == D) ? (D & ~(int64_t)3) : r1;
    // asymmetric: clamp(0, 2^b - 1) (quantization_utils.py:92); absmax has no clamp in the reference -- the
    // int8 range only guards the cast
    const float qlo = mode == 0 ? 0.f : -128.f, qhi = mode == 0 ? (float)((1 << bits) - 1) : 127.f;
    for (int t = 0; t < NT; ++t) {
        const float *x = x_ptrs[(size_t)p * NT + t];
        uint8_t *c = code_ptrs[(size_t)p * NT + t];
        const float scale = scale_in[(size_t)p * NT + t], zp = mode == 0 ? zp_in[(size_t)p * NT + t] : 0.f;
        int64_t i = r0 + 4 * lane;
        for (; i + 768 < v1; i += 1024) {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *AS1(f32x4, x + i + 256 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u8x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = tvq_code(v[q][e], scale, zp, qlo, qhi, mode);
                *reinterpret_cast<u8x4 *>(c + i + 256 * q) = o;
            }
        }
        for (; i < v1; i += 256) {
            const f32x4 v = *AS1(f32x4, x + i);
            u8x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = tvq_code(v[e], scale, zp, qlo, qhi, mode);
            *reinterpret_cast<u8x4 *>(c + i) = o;
        }
        if (r1 == D && lane < (int)(D & 3) && v1 >= r0) c[v1 + lane] = tvq_code(x[v1 + lane], scale, zp, qlo, qhi, mode);
    }
}

__device__ __forceinline__ float tvq_value(unsigned char q, float scale, float zp, int mode) {
    // dequantize_asymmetric: (q - zp) / scale; dequantize_absmax: q * scale (sic, quantization_utils.py:102-134)
    return mode == 0 ? __fdiv_rn(__fsub_rn((float)q, zp), scale) : __fmul_rn((float)(signed char)q, scale);
}

__global__ __launch_bounds__(64) void k_tvq_dequant(const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
                                                    int NT, int mode, const uint8_t *const *__restrict__ code_ptrs,
                                                    const float *__restrict__ scale_in, const float *__restrict__ zp_in,
                                                    const float *const *__restrict__ add_ptrs,
                                                    float *const *__restrict__ out_ptrs) {
    const int uidx = blockIdx.x, lane = threadIdx.x;
    const SvdqUnit u = units[uidx];
    const int p = u.param;
    const int64_t D = params[p].rows;
    const int64_t r0 = u.row0, r1 = r0 + u.nrows;
    const int64_t v1 = (r1 == D) ? (D & ~(int64_t)3) : r1;
    const float *add = add_ptrs ? add_ptrs[p] : nullptr;
    for (int t = 0; t < NT; ++t) {
        const uint8_t *c = code_ptrs[(size_t)p * NT + t];
        float *o = out_ptrs[(size_t)p * NT + t];
        const float scale = scale_in[(size_t)p * NT + t], zp = mode == 0 ? zp_in[(size_t)p * NT + t] : 0.f;
        int64_t i = r0 + 4 * lane;
        for (; i + 768 < v1; i += 1024) {  // 4 code words (and addends) in flight per lane
            u8x4 q[4];
            f32x4 a[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                q[w] = *AS1(u8x4, c + i + 256 * w);
                if (add) a[w] = *AS1(f32x4, add + i + 256 * w);
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = tvq_value(q[w][e], scale, zp, mode);
                    if (add) v[e] = __fadd_rn(a[w][e], v[e]);
                }
                *reinterpret_cast<f32x4 *>(o + i + 256 * w) = v;
            }
        }
        for (; i < v1; i += 256) {
            const u8x4 q = *AS1(u8x4, c + i);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tvq_value(q[e], scale, zp, mode);
            if (add) {
                const f32x4 a = *AS1(f32x4, add + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(a[e], v[e]);
            }
            *reinterpret_cast<f32x4 *>(o + i) = v;
        }
        if (r1 == D && lane < (int)(D & 3) && v1 >= r0) {
            const int64_t i = v1 + lane;
            float v = tvq_value(c[i], scale, zp, mode);
            if (add) v = __fadd_rn(add[i], v);
            o[i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------ entry points
extern "C" int64_t svdq_tvq_work_bytes(const svdq_plan *pl) {
    if (!pl) return 0;
    return svdq_align_up((int64_t)pl->n_units * pl->n_tasks * sizeof(TvqPartial), 256);
}

static int tvq_check(const svdq_plan *pl, int32_t mode, int32_t bits) {
    if (!pl) {
        svdq_set_error("null plan");
        return SVDQ_EINVAL;
    }
    if (mode != 0 && mode != 1) {
        svdq_set_error("Unknown quantization method %d (0 = asymmetric, 1 = absmax)", mode);
        return SVDQ_EINVAL;
    }
    if (mode == 1 && bits == 16) return SVDQ_OK;  // int16 codes
    if (bits < 1 || bits > 8 || (mode == 1 && bits < 2)) {
        svdq_set_error("qbit must be in [%d, 8]%s, got %d (asymmetric int16 codes are not implemented)",
                       mode == 1 ? 2 : 1, mode == 1 ? " or 16" : "", bits);
        return SVDQ_EUNSUPPORTED;
    }
    return SVDQ_OK;
}

extern "C" int svdq_ingest(const svdq_plan *pl, const void *base_ptrs, const void *finetuned_ptrs,
                           const void *delta_ptrs, void *stats_work, void *stream) {
    if (!pl || !base_ptrs || !finetuned_ptrs || !delta_ptrs) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    auto bp = reinterpret_cast<const float *const *>(base_ptrs);
    auto fp = reinterpret_cast<const float *const *>(finetuned_ptrs);
    auto dp = reinterpret_cast<float *const *>(delta_ptrs);
    if (stats_work)
        hipLaunchKernelGGL(k_ingest<true>, dim3(pl->n_units), dim3(64), 0, (hipStream_t)stream, pl->d_params, pl->d_units,
                           pl->n_tasks, bp, fp, dp, reinterpret_cast<TvqPartial *>(stats_work));
    else
        hipLaunchKernelGGL(k_ingest<false>, dim3(pl->n_units), dim3(64), 0, (hipStream_t)stream, pl->d_params,
                           pl->d_units, pl->n_tasks, bp, fp, dp, (TvqPartial *)nullptr);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_tvq_quantize(const svdq_plan *pl, const void *x_ptrs, int32_t mode, int32_t bits,
                                 const void *code_ptrs, float *scale, float *zero_point, void *work,
                                 int32_t stats_ready, void *stream) {
    if (int rc = tvq_check(pl, mode, bits)) return rc;
    if (!x_ptrs || !code_ptrs || !scale || !work || (mode == 0 && !zero_point)) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    auto xp = reinterpret_cast<const float *const *>(x_ptrs);
    TvqPartial *part = reinterpret_cast<TvqPartial *>(work);
    if (!stats_ready)
        hipLaunchKernelGGL(k_tvq_stats, dim3(pl->n_units), dim3(64), 0, st, pl->d_params, pl->d_units, pl->n_tasks, xp,
                           part);
    const int nt = pl->n_params * pl->n_tasks;
    hipLaunchKernelGGL(k_tvq_params, dim3((nt + 63) / 64), dim3(64), 0, st, pl->d_params, pl->n_params, pl->n_tasks, mode,
                       bits, part, scale, zero_point);
    if (bits == 16)
        hipLaunchKernelGGL(k_tvq_apply16, dim3(pl->n_units), dim3(64), 0, st, pl->d_params, pl->d_units, pl->n_tasks, xp,
                           scale, reinterpret_cast<short *const *>(code_ptrs));
    else
        hipLaunchKernelGGL(k_tvq_apply, dim3(pl->n_units), dim3(64), 0, st, pl->d_params, pl->d_units, pl->n_tasks, mode,
                           bits, xp, scale, zero_point, reinterpret_cast<uint8_t *const *>(code_ptrs));
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_tvq_dequantize(const svdq_plan *pl, const void *code_ptrs, int32_t mode, const float *scale,
                                   const float *zero_point, const void *add_ptrs, const void *out_ptrs,
                                   void *stream) {
    if (mode == 2) {  // absmax with int16 codes
        if (!pl || !code_ptrs || !scale || !out_ptrs) {
            svdq_set_error("null argument");
            return SVDQ_EINVAL;
        }
        hipLaunchKernelGGL(k_tvq_dequant16, dim3(pl->n_units), dim3(64), 0, (hipStream_t)stream, pl->d_params,
                           pl->d_units, pl->n_tasks, reinterpret_cast<const short *const *>(code_ptrs), scale,
                           reinterpret_cast<const float *const *>(add_ptrs), reinterpret_cast<float *const *>(out_ptrs));
        return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
    }
    if (int rc = tvq_check(pl, mode, 8)) return rc;
    if (!code_ptrs || !scale || !out_ptrs || (mode == 0 && !zero_point)) {
        svdq_set_error("null argument");
        return SVDQ_EINVAL;
    }
    hipLaunchKernelGGL(k_tvq_dequant, dim3(pl->n_units), dim3(64), 0, (hipStream_t)stream, pl->d_params, pl->d_units,
                       pl->n_tasks, mode, reinterpret_cast<const uint8_t *const *>(code_ptrs), scale, zero_point,
                       reinterpret_cast<const float *const *>(add_ptrs), reinterpret_cast<float *const *>(out_ptrs));
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

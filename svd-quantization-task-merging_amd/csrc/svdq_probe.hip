// svdq_probe.hip -- measurement only: what this box's HBM delivers to plain streaming kernels with the access shape
// of the two passes (16 B per lane, 1 KiB per wave instruction, 8 loads in flight per lane).  bench.py quotes the
// path's roofline fraction against the 8 TB/s specification AND against these measured ceilings.
//   mode 0  read-only   (pass 1's shape: everything loaded, 4 bytes per workgroup stored)
//   mode 1  copy        (1 byte stored per byte loaded)
//   mode 2  read 8 : write 5   (pass 2's mix at N = 8: 32 B of deltas in, 16 B of fp16 basis + 4 B of mean out)

#include "svdq_common.h"

typedef const __attribute__((address_space(1))) f32x4 pgf32x4;

// One wavefront (= one 64-thread workgroup, like the two passes) walks a contiguous chunk in 8 KiB steps: eight 16-B
// loads per lane in flight, each wave instruction covering 1 KiB.
#define PROBE_CHUNK_VECS (16384)   // 256 KiB per wave

template <int MODE>
__global__ __launch_bounds__(64) void k_probe(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, int64_t nvec) {
    constexpr int INFL = 8;
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * PROBE_CHUNK_VECS;
    int64_t c1 = c0 + PROBE_CHUNK_VECS;
    if (c1 > nvec) c1 = nvec;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int64_t i = c0;
    for (; i + INFL * 64 <= c1; i += INFL * 64) {
        f32x4 v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; ++u) v[u] = *(pgf32x4 *)(src + i + u * 64 + lane);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int u = 0; u < INFL; ++u) acc += v[u];
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < INFL; ++u) dst[i + u * 64 + lane] = v[u];
        } else {   // 8 KiB in, 5 KiB out, written densely
            const int64_t o = i / 8 * 5;
#pragma unroll
            for (int u = 0; u < 5; ++u) dst[o + u * 64 + lane] = v[u] + v[u + 3];
        }
    }
    for (i += lane; i < c1; i += 64) {   // tail of the last chunk
        const f32x4 v = src[i];
        if constexpr (MODE == 0) acc += v;
        else if constexpr (MODE == 1) dst[i] = v;
    }
    if constexpr (MODE == 0) {
        float s = (acc.x + acc.y) + (acc.z + acc.w);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) reinterpret_cast<float *>(dst)[blockIdx.x] = s;
    }
}

extern "C" int svdq_hbm_probe(int32_t mode, const void *src_dev, void *dst_dev, int64_t bytes, void *stream) {
    if (!src_dev || !dst_dev || bytes < 16 || (bytes & 15) || mode < 0 || mode > 2) {
        svdq_set_error("svdq_hbm_probe: mode 0..2, 16-byte multiple of bytes, non-null buffers");
        return SVDQ_EINVAL;
    }
    if (((uintptr_t)src_dev | (uintptr_t)dst_dev) & 15) {
        svdq_set_error("svdq_hbm_probe: buffers must be 16-byte aligned");
        return SVDQ_EINVAL;
    }
    const int64_t nvec = bytes / 16;
    const int grid = (int)((nvec + PROBE_CHUNK_VECS - 1) / PROBE_CHUNK_VECS);
    auto s = reinterpret_cast<const f32x4 *>(src_dev);
    auto d = reinterpret_cast<f32x4 *>(dst_dev);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(64), 0, st, s, d, nvec);
    else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(64), 0, st, s, d, nvec);
    else hipLaunchKernelGGL(k_probe<2>, dim3(grid), dim3(64), 0, st, s, d, nvec);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

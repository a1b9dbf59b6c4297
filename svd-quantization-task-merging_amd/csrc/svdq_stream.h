// svdq_stream.h -- shared by the two HBM-streaming passes of the SVD-Hybrid compressor (gfx950 / CDNA4):
// svdq_gram.hip (pass 1) and svdq_project.hip (pass 2).  Block loading, centring, the mask-walk helpers.
//
//   k_gram          pass 1: G = Tc^T Tc          (reference basis.py:63-113 + the reduction half of
//                                                  torch.linalg.svd, basis.py:216-249)
//   k_basis_project pass 2: U = Tc W -> fp16, mean, c = fp16(U)^T Tc
//                                                 (basis.py:363-364, cli.py:354-361, compress.py:6-21,35-40)
//
// Both walk the same 256-row blocks.  One wavefront (= one 64-thread workgroup, so LDS is
// wave-private and no cross-wave barrier exists) owns a unit of consecutive blocks:
//
//   global --16 B/lane, 1 KiB contiguous per task per instruction--> VGPR (next block prefetched)
//          --row mean over tasks, subtract--> LDS  X[task][row]  (centred, zero past the end)
//          --ds_read in the two MFMA operand layouts--> v_mfma_f32_16x16x4_f32
//
// MFMA operand layouts (16x16x4 f32: lane l supplies A[l&15][l>>4] and B[l>>4][l&15],
// holds D[4*(l>>4)+reg][l&15]):
//   "task on slot, row on k"  value X[task l&15][row 4*(l>>4)+e]   one ds_read_b128 = 4 MFMA steps
//        Gram:        D[m][n] += X[m][row] * X[n][row]   (A and B are the SAME register)
//        projection:  B operand; A operand is the rounded U tile straight out of the U-MFMA
//                     accumulator (its D layout is exactly "U column on slot, row on k").
//   "row on slot, task on k"  value X[task 4s+(l>>4)][row l&15]    ds_read_b32 per k-step s
//        U = Tc W:    A operand; B operand W[4s+(l>>4)][l&15] lives in registers.
//   Row-set packing (N <= 8): slots 0-7 carry tasks for one 16-row set, slots 8-15 the same tasks
//   for the next 16 rows; the two diagonal 8x8 blocks of D are two independent partial sums.
//
// Accumulation: fp32 inside a block (64 MFMA k-steps), fp64 across blocks and units.

#pragma once
// SVDQ_UNIT_STAMPS (diagnostic builds only, tools/unit_timeline.py): every work unit of the two streaming passes records
// when its wavefront started and ended (s_memrealtime, 100 MHz) and on which XCD it ran, so that the ramp-up, the
// steady state and the tail of a launch can be drawn.  Nothing is emitted otherwise.
#ifdef SVDQ_UNIT_STAMPS
#define SVDQ_STAMP_DECL(name) static __device__ unsigned long long *name = nullptr;
#define SVDQ_STAMP_BEGIN() const unsigned long long stamp_t0_ = wall_clock64()
#define SVDQ_STAMP_END(name, u)                                                       \
    do {                                                                              \
        if (name && (threadIdx.x == 0)) {                                             \
            unsigned hw = 0;                                                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));         \
            name[3 * (size_t)(u)] = stamp_t0_;                                        \
            name[3 * (size_t)(u) + 1] = wall_clock64();                               \
            name[3 * (size_t)(u) + 2] = hw;                                           \
        }                                                                             \
    } while (0)
#else
#define SVDQ_STAMP_DECL(name)
#define SVDQ_STAMP_BEGIN()
#define SVDQ_STAMP_END(name, u)
#endif

#include "svdq_common.h"
#include <hip/hip_fp16.h>

#define XS SVDQ_XS
#ifndef SVDQ_EXP_ULOW_SHIFT
#define SVDQ_EXP_ULOW_SHIFT 0   // experiment builds only (tools/placement_probe7.py): U_low written this many bytes further
#endif
#ifndef SVDQ_UNROLL_BP
#define SVDQ_UNROLL_BP 8
#endif
#ifndef SVDQ_UNROLL_BP2
#define SVDQ_UNROLL_BP2 4  // sub-tile-pair loop of the two-wave pass 2 (N > 16)
#endif
#ifndef SVDQ_UNROLL_GRAM
#define SVDQ_UNROLL_GRAM 8
#endif
#ifndef SVDQ_UNROLL_GRAM_P1
#define SVDQ_UNROLL_GRAM_P1 16  // sub-tile loop of the unpacked (N > 8) Gram variants
#endif
#ifndef SVDQ_NT_LOADS
#define SVDQ_NT_LOADS 0
#endif
#ifndef SVDQ_NT_STORES
#define SVDQ_NT_STORES 0
#endif
#ifndef SVDQ_PREFETCH2
#define SVDQ_PREFETCH2 0  // 1: two register sets, loads two blocks ahead (measured: no gain, fewer waves)
#endif
#define PRAGMA_(x) _Pragma(#x)
#define UNROLL_N(n) PRAGMA_(unroll n)

// Which work unit a workgroup takes.  Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b % 8),
// and consecutive units are consecutive row ranges of one tensor.  order bit 2 ("XCD-chunked"): XCD x walks ONE
// contiguous eighth of the unit list, so the address window its L2 and its translation caches see at any time is an
// eighth of what the interleaved order gives.  bit 0: reversed unit order (measurement).
__device__ __forceinline__ int unit_of_block(int b, int n, int order) {
    int u = b;
    if (order & 4) {
        const int q = n >> 3, rem = n & 7, x = b & 7;
        u = x * q + (x < rem ? x : rem) + (b >> 3);
    }
    return (order & 1) ? n - 1 - u : u;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One wavefront per workgroup: LDS operations of a wave execute in order, so phases only need the
// COMPILER kept from moving LDS accesses across the boundary.  __syncthreads() would also emit
// s_waitcnt vmcnt(0), draining this wave's global stores and prefetched loads once per block.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64: operands as the f32 form (one value per lane: A[l&15][l>>4], B[l>>4][l&15]); the
// accumulator layout differs: lane l holds D[(l>>4) + 4*reg][l&15].  A product of two fp32 values is exact in
// fp64, so a Gram accumulated this way carries only the ~1e-16 rounding of the running sums.
__device__ __forceinline__ f64x4 mfma4d(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 blocks, D_b[i][j] += A_b[i] B_b[j]; lane l belongs to block l / 4,
// supplies A_b[l % 4] and B_b[l % 4] and holds D_b[register][l % 4] (tools/probe/mfma_layout.hip); 8 cycles.
__device__ __forceinline__ f32x4 mfma_4x4x1(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 zero4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}

// Task-delta pointers come out of a device table, so the compiler only knows them as generic
// ("flat") pointers; flat loads count on BOTH vmcnt and lgkmcnt, which would make every LDS wait
// also drain the next block's prefetch.  Cast them to the global address space explicitly.
typedef const __attribute__((address_space(1))) float gfloat;
typedef const __attribute__((address_space(1))) f32x4 gf32x4;

// Issue the 16-B loads of one 256-row block: lane l takes rows rb+4l..rb+4l+3 of every task.
// Full blocks take the unconditional path (no per-load branch, all loads in flight together);
// only the last block of a parameter takes the guarded one.
template <int NTP>
__device__ __forceinline__ void load_block(f32x4 (&v)[NTP], gfloat *(&bp)[NTP], int64_t rb,
                                           int64_t D, int lane) {
    const int64_t r = rb + 4 * lane;
    if (rb + SVDQ_BLK_ROWS <= D) {
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
#if SVDQ_NT_LOADS
            v[t] = __builtin_nontemporal_load(reinterpret_cast<gf32x4 *>(bp[t] + r));
#else
            v[t] = *reinterpret_cast<gf32x4 *>(bp[t] + r);
#endif
        }
    } else {
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            f32x4 o = zero4();
            if (r < D) o.x = bp[t][r];
            if (r + 1 < D) o.y = bp[t][r + 1];
            if (r + 2 < D) o.z = bp[t][r + 2];
            if (r + 3 < D) o.w = bp[t][r + 3];
            v[t] = o;
        }
    }
}

// Gather mode (masked parameters without a compaction pass): the kernels walk the COMPACTED row space and
// fetch row j of every task from source element idx[j] (ascending positions of the set mask bits, built
// once per mask by svdq_maskset_indices).  Outputs stay exactly as in the contiguous mode.  Indices are
// loaded one block ahead of the data they address, so the data loads never wait for them.
typedef const __attribute__((address_space(1))) int32_t gint;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) i32x4 gi32x4;

// In gather mode lane l owns rows l, 64+l, 128+l, 192+l of a block (not 4l..4l+3): the four dword loads of a
// task then each read 64 CONSECUTIVE compacted rows -- a nearly contiguous 256-byte run of the source -- instead
// of every fourth row of a 1 KiB span, which would make each instruction touch the same 16 cache lines.
__device__ __forceinline__ i32x4 load_idx(gint *idx, int64_t rb, int64_t D, int lane) {
    const int64_t r = rb + lane;
    i32x4 o = {-1, -1, -1, -1};
    if (rb + SVDQ_BLK_ROWS <= D) {
        o.x = idx[r];
        o.y = idx[r + 64];
        o.z = idx[r + 128];
        o.w = idx[r + 192];
    } else {
        if (r < D) o.x = idx[r];
        if (r + 64 < D) o.y = idx[r + 64];
        if (r + 128 < D) o.z = idx[r + 128];
        if (r + 192 < D) o.w = idx[r + 192];
    }
    return o;
}

// "Minus base" mode (svdq_compress_from_base): the task tensors are FINE-TUNED weights and the delta
// finetuned - base is formed in registers, so the task vectors are never written to or read back from HBM.
__device__ __forceinline__ f32x4 load_base(gfloat *b, int64_t rb, int64_t D, int lane) {
    const int64_t r = rb + 4 * lane;
    if (rb + SVDQ_BLK_ROWS <= D) return *reinterpret_cast<gf32x4 *>(b + r);
    f32x4 o = zero4();
    if (r < D) o.x = b[r];
    if (r + 1 < D) o.y = b[r + 1];
    if (r + 2 < D) o.z = b[r + 2];
    if (r + 3 < D) o.w = b[r + 3];
    return o;
}

// base rows of a gathered block (minus-base mode combined with gather mode): same indices as the task rows
__device__ __forceinline__ f32x4 load_base_gather(gfloat *b, const i32x4 &ix, bool full) {
    f32x4 o = zero4();
    if (full) {
        o.x = b[ix.x];
        o.y = b[ix.y];
        o.z = b[ix.z];
        o.w = b[ix.w];
    } else {
        if (ix.x >= 0) o.x = b[ix.x];
        if (ix.y >= 0) o.y = b[ix.y];
        if (ix.z >= 0) o.z = b[ix.z];
        if (ix.w >= 0) o.w = b[ix.w];
    }
    return o;
}

template <int NTP>
__device__ __forceinline__ void load_block_gather(f32x4 (&v)[NTP], gfloat *(&bp)[NTP], const i32x4 &ix, bool full) {
    if (full) {
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            f32x4 o;
            o.x = bp[t][ix.x];
            o.y = bp[t][ix.y];
            o.z = bp[t][ix.z];
            o.w = bp[t][ix.w];
            v[t] = o;
        }
    } else {
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            f32x4 o = zero4();
            if (ix.x >= 0) o.x = bp[t][ix.x];
            if (ix.y >= 0) o.y = bp[t][ix.y];
            if (ix.z >= 0) o.z = bp[t][ix.z];
            if (ix.w >= 0) o.w = bp[t][ix.w];
            v[t] = o;
        }
    }
}

// Row mean over the NT real tasks (sum in task order, then one fp32 divide: basis.py:109).  Component-wise over the
// lane's four rows, whichever rows those are.
template <int NTP>
__device__ __forceinline__ f32x4 row_mean(const f32x4 (&v)[NTP], int NT, int center) {
    f32x4 s = zero4();
    if constexpr (NTP > 16) {
        // same association as the two-wave pass 2 (each wave sums half of the tasks, then h0 + h1): pass 1 and
        // pass 2 must centre with the same mean, bit for bit
        f32x4 h[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            h[w] = zero4();
#pragma unroll
            for (int i = 0; i < NTP / 2; ++i) {
                const int t = w * (NTP / 2) + i;
                h[w] += (t < NT) ? v[t] : zero4();
            }
        }
        s = h[0] + h[1];
    } else {
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            f32x4 x = (t < NT) ? v[t] : zero4();
            s += x;
        }
    }
    f32x4 mean = zero4();
    if (center) {
        const float n = (float)NT;
        mean.x = s.x / n;
        mean.y = s.y / n;
        mean.z = s.z / n;
        mean.w = s.w / n;
    }
    return mean;
}

// Subtract the row mean (basis.py:111) and park the centred strip in LDS.  Padded tasks are stored as 0.
// STRIDED (gather mode): component e of a lane's vector is row 64 e + lane, not 4 lane + e.
template <int NTP, bool STRIDED = false>
__device__ __forceinline__ f32x4 center_store(const f32x4 (&v)[NTP], int NT, int center, float *X, int lane) {
    const f32x4 mean = row_mean<NTP>(v, NT, center);
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
        f32x4 xc = (t < NT) ? (v[t] - mean) : zero4();
        if constexpr (STRIDED) {
#pragma unroll
            for (int e = 0; e < 4; ++e) X[t * XS + 64 * e + lane] = xc[e];
        } else {
            *reinterpret_cast<f32x4 *>(X + t * XS + 4 * lane) = xc;
        }
    }
    return mean;
}

// ------------------------------------------------------------------------------------ walk mode (masked parameters)
// MODE bit 2.  Masked parameters WITHOUT index lists (reference mask_loader.py:651-709 applied inside the passes):
// a unit still owns a run of 256-row blocks of the COMPACTED row space -- so every artifact bit is where the
// compacted / gather modes put it -- but it reaches them by walking the SOURCE tensor from the position of its first
// selected element (ustart[unit], found once per mask by svdq_maskset_*_starts from the tile scan):
//   * chunk = 256 consecutive source rows; lane l owns rows src + 64 e + l (e = 0..3), so every dword load of a task
//     reads 256 contiguous bytes and every mask load 64 contiguous bytes -- 4 N + 1 bytes per source row and pass,
//     nothing per selected row (the index lists cost 4 bytes per selected row and pass on top of the rows themselves);
//   * the four ballots of "row selected" give every selected row its rank in the chunk (s_bcnt / v_mbcnt, no scan
//     through LDS); rank + rows already in the strip = its row in the current block;
//   * centred values are scattered into the strip with ds_write_b32 -- consecutive lanes hold consecutive selected
//     rows, so the writes are conflict-free; rows that overflow the block wait in registers until the block has been
//     consumed (phase B below) and then open the next one.
// Loads past the unit's last source row (the next unit's start, or the end of the tensor) are masked off, so
// neighbouring units do not fetch each other's rows beyond the sector they share.
typedef const __attribute__((address_space(1))) uint8_t gbyte;
#define SVDQ_WALK_INV (1ll << 62)      // ustart[u] bit 62: select the CLEARED mask elements (the noise region)

__device__ __forceinline__ int lanes_below(unsigned long long bal) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
}

struct WalkSel {
    bool sel[4];   // this lane's row e is selected
    int pos[4];    // its row in the strip, counted from the start of the current block (may be >= 256)
    int total;     // rows in the strip once the chunk is in (wave-uniform)
};

// the chunk's data: row src + 64 e + lane of every task (and of the base tensor, minus-base mode) + the mask bytes
template <int NTP, bool SUB>
__device__ __forceinline__ void walk_load(f32x4 (&v)[NTP], f32x4 &vb, unsigned (&mk)[4], gfloat *(&bp)[NTP],
                                          gfloat *gbase, gbyte *gmask, int64_t src, int64_t src_end, int lane) {
    const int64_t r = src + lane;
    if (src + SVDQ_BLK_ROWS <= src_end) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mk[e] = gmask[r + 64 * e];
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            f32x4 o;
            o.x = bp[t][r];
            o.y = bp[t][r + 64];
            o.z = bp[t][r + 128];
            o.w = bp[t][r + 192];
            v[t] = o;
        }
        if constexpr (SUB) {
            vb.x = gbase[r];
            vb.y = gbase[r + 64];
            vb.z = gbase[r + 128];
            vb.w = gbase[r + 192];
        }
    } else {
        bool in[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            in[e] = r + 64 * e < src_end;
            mk[e] = in[e] ? (unsigned)gmask[r + 64 * e] : 0x100u;   // 0x100: past the end, selected by neither polarity
        }
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            f32x4 o = zero4();
            if (in[0]) o.x = bp[t][r];
            if (in[1]) o.y = bp[t][r + 64];
            if (in[2]) o.z = bp[t][r + 128];
            if (in[3]) o.w = bp[t][r + 192];
            v[t] = o;
        }
        if constexpr (SUB) {
            vb = zero4();
            if (in[0]) vb.x = gbase[r];
            if (in[1]) vb.y = gbase[r + 64];
            if (in[2]) vb.z = gbase[r + 128];
            if (in[3]) vb.w = gbase[r + 192];
        }
    }
}

// ranks of the chunk's selected rows (ascending source position = e-major, lane-minor)
__device__ __forceinline__ WalkSel walk_select(const unsigned (&mk)[4], int inv, int fill) {
    WalkSel w;
    int base = fill;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        w.sel[e] = inv ? (mk[e] == 0u) : (mk[e] != 0u && mk[e] != 0x100u);
        const unsigned long long bal = __ballot(w.sel[e]);
        w.pos[e] = base + lanes_below(bal);
        base += (int)__popcll(bal);
    }
    w.total = base;
    return w;
}

// strip rows [lo, lo + 256) of the chunk: phase A (lo = 0) completes the current block, phase B (lo = 256) opens the next
template <int NTP>
__device__ __forceinline__ void walk_scatter(float *X, const f32x4 (&xc)[NTP], const WalkSel &w, int lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int q = w.pos[e] - lo;
        if (w.sel[e] && (unsigned)q < (unsigned)SVDQ_BLK_ROWS) {
#pragma unroll
            for (int t = 0; t < NTP; ++t) X[t * XS + q] = xc[t][e];
        }
    }
}

// rows [fill, 256) of the strip <- 0 (the last, partial block of a parameter)
template <int NTP>
__device__ __forceinline__ void walk_zero_tail(float *X, int fill, int lane) {
    for (int q = fill + lane; q < SVDQ_BLK_ROWS; q += 64) {
#pragma unroll
        for (int t = 0; t < NTP; ++t) X[t * XS + q] = 0.f;
    }
}


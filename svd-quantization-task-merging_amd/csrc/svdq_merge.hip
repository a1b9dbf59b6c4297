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
//   k_merge_expand       the same for MASKED regions with reconstruct_from_masked (mask_loader.py:712-763) inside: walks the
//                        source rows with the combined mask byte beside them, writes every merged row at its source
//                        position (signal and noise regions each their own rows of the full tensor).
//   k_diag / k_diag_finish  compute_parameter_diagnostics' inner loop (diagnostics.py:186-215) for all N tasks of a
//                        parameter in one pass over U and the N deltas: N error tuples per parameter; walk mode =
//                        masked regions, apply_mask_to_tensor (mask_loader.py:651-679) inside the pass.
//
// Per-row arithmetic is that of k_reconstruct / k_recon_error (svdq_elem.hip): fp32 fma chains from 0 over the columns
// in order, hi + lo, + mean, * scale -- so a parameter's merged rows are the same bits as the per-parameter route's
// (the diagnostics' sums run in another order: equal to the last digits of the fp64 accumulators).

#include "svdq_common.h"
#include <hip/hip_fp16.h>

#define MRG_MAX_SETS 8

// pointers read out of device tables are generic to the compiler: without the address space it emits flat_load /
// flat_store, which count on lgkmcnt as well and so make every LDS wait a wait for HBM
typedef const __attribute__((address_space(1))) float mg_gfloat;
typedef __attribute__((address_space(1))) float mg_gfloat_w;
typedef const __attribute__((address_space(1))) uint8_t mg_gbyte;
typedef const __attribute__((address_space(1))) f32x4 mg_gf32x4;

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

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// where a block's staged basis rows sit (wave-uniform): the two parts are fetched with aligned 16-byte loads and kept
// row-major, one after the other, in LDS
struct UStage {
    int nvh, nv;     // 16-byte vectors of the U_high part, of both parts
    int offh, offl;  // element offset of the block's first row inside each part
    int64_t a0h, a0l;
};
template <int ES>
__device__ __forceinline__ UStage ustage_plan(int64_t c0, int nr, int k, int nl) {
    UStage u;
    const int64_t b0h = c0 * k * ES, b1h = (c0 + nr) * (int64_t)k * ES;
    const int64_t b0l = c0 * nl * ES, b1l = (c0 + nr) * (int64_t)nl * ES;
    u.a0h = b0h & ~15ll;
    u.a0l = b0l & ~15ll;
    u.nvh = k > 0 ? (int)((b1h - u.a0h + 15) >> 4) : 0;
    u.nv = u.nvh + (nl > 0 ? (int)((b1l - u.a0l + 15) >> 4) : 0);
    u.offh = (int)(b0h - u.a0h) / ES;
    u.offl = (int)(b0l - u.a0l) / ES;
    return u;
}

// One wavefront per work unit of the plan.  Blocks of RB = 64 RPL rows (256 for N <= 16, 128 above: the block's basis
// rows then fit 9 (fp16) / 17 (fp32) 16-byte registers per lane); lane l owns rows l, 64 + l, ... of a block (conflict-free
// row reads from the row-major LDS image).  Software pipeline, one block deep: while block b is computed from LDS, the
// loads of block b + 1 -- its run of basis rows (16-byte loads), its mean and base rows -- are in flight into registers
// (round 3 loaded a tile, fenced, computed, fenced: nothing was in flight during the compute phase and the counters showed
// 84 % of the wave cycles waiting).  Columns outermost: one coefficient read serves the lane's rows, the fma chains are
// independent.  NS = compiled-in number of sets (1, 2, 4, 8 >= n_sets).
template <bool U16, int NS, int RPL>
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
    constexpr int RB = 64 * RPL;
    constexpr int NMAX = RPL == 4 ? 16 : 32;                          // tasks this block size is launched for
    constexpr int SV = (RB * NMAX * ES / 16 + 2 + 63) / 64;          // 16-byte vectors of one block's basis rows per lane
    // dynamic LDS: the staged basis rows (both parts + alignment slack), then the coefficient sets
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x, n = NT;
    const int ubytes = (int)svdq_align_up((int64_t)RB * n * ES + 48, 16);
    float *C = reinterpret_cast<float *>(lds_raw + ubytes);   // [column][NS]
    float *SH = C + NS * n;
    const SvdqUnit ud = units[blockIdx.x];
    const int p = ud.param;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    const int64_t r_begin = ud.row0;
    int64_t r_end = r_begin + ud.nrows;
    if (r_end > D) r_end = D;
    if (r_begin >= r_end) return;
    const int k = k_in[p], r = r_in[p], nl = r - k;
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
    mg_gfloat *gmean = meanbuf ? (mg_gfloat *)(meanbuf + params[p].mean_off) : nullptr;
    mg_gfloat *gbase = base_ptrs ? (mg_gfloat *)base_ptrs[p] : nullptr;
    mg_gfloat_w *gout = (mg_gfloat_w *)out_ptrs[p];

    // ---- the loads of one block into registers
    f32x4 ureg[SV];
    float mpf[RPL] = {}, bpf[RPL] = {};
    auto prefetch = [&](int64_t rb) {
        const int nr = (int)((r_end - rb < RB) ? (r_end - rb) : RB);
        const UStage us = ustage_plan<ES>(rb, nr, k, nl);
#pragma unroll
        for (int s = 0; s < SV; ++s) {
            const int v = lane + 64 * s;
            if (v < us.nv) {
                const uint8_t *gp = (v < us.nvh) ? gUh + us.a0h + 16ll * v : gUl + us.a0l + 16ll * (v - us.nvh);
                ureg[s] = *(mg_gf32x4 *)gp;
            }
        }
#pragma unroll
        for (int m = 0; m < RPL; ++m) {      // rows past the block's end are clamped into it (their results are not stored)
            const int q = 64 * m + lane;
            const int64_t row = rb + (q < nr ? q : nr - 1);
            if (gmean) mpf[m] = gmean[row];
            if (gbase) bpf[m] = gbase[row];
        }
    };
    prefetch(r_begin);
    for (int64_t rb = r_begin; rb < r_end; rb += RB) {
        const int rows_blk = (int)((r_end - rb < RB) ? (r_end - rb) : RB);
        // ---- registers -> LDS
        const UStage cur = ustage_plan<ES>(rb, rows_blk, k, nl);
#pragma unroll
        for (int s = 0; s < SV; ++s) {
            const int v = lane + 64 * s;
            if (v < cur.nv) reinterpret_cast<f32x4 *>(lds_raw)[v] = ureg[s];
        }
        float mv[RPL], bv[RPL];
        int rl[RPL];      // the lane's rows, clamped into the block
#pragma unroll
        for (int m = 0; m < RPL; ++m) {
            const int q = 64 * m + lane;
            rl[m] = q < rows_blk ? q : rows_blk - 1;
            mv[m] = mpf[m];
            bv[m] = bpf[m];
        }
        lds_fence();
        // ---- the next block's loads
        if (rb + RB < r_end) prefetch(rb + RB);
        // ---- compute
        const T *Uh = reinterpret_cast<const T *>(lds_raw) + cur.offh;
        const T *Ul = reinterpret_cast<const T *>(lds_raw + 16 * cur.nvh) + cur.offl;
        float hi[RPL][NS], lo[RPL][NS];
#pragma unroll
        for (int m = 0; m < RPL; ++m)
#pragma unroll
            for (int s = 0; s < NS; ++s) hi[m][s] = lo[m][s] = 0.f;
        for (int i = 0; i < k; ++i) {
            float c[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) c[s] = C[i * NS + s];
#pragma unroll
            for (int m = 0; m < RPL; ++m) {
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
            for (int m = 0; m < RPL; ++m) {
                const float u = u_val(Ul, rl[m] * nl + j);
#pragma unroll
                for (int s = 0; s < NS; ++s) lo[m][s] = fmaf(u, c[s], lo[m][s]);
            }
        }
#pragma unroll
        for (int m = 0; m < RPL; ++m) {
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
        lds_fence();      // the basis rows are rewritten next
    }
}

// ------------------------------------------------------------------------------------ masked parameters: source walk
// reconstruct_from_masked (mask_loader.py:712-763) fused into the merge: the artifacts of a masked region describe the
// COMPACTED rows, the merged tensor wants them back at their source positions.  A unit of the plan owns the source rows
// from its first selected element (ustart[u], svdq_maskset_*_starts -- the table the mask-walk compressor uses) to the
// next unit's; the first unit of a parameter starts at 0, the last ends with the tensor.  Per chunk of 256 source rows
// (lane l owns rows src + 64 e + l: every load and store is 256 contiguous bytes):
//   mask bytes -> four ballots -> compacted row of every selected source row (no index list, no scan through memory);
//   the chunk's run of basis rows [cpos, cpos + selected) is contiguous in U_high / U_low: staged in LDS by 16-byte loads;
//   selected rows get the merged value (same per-row arithmetic as k_merge_reconstruct), the others 0 when ``fill``
//   (no noise region writes them) or are left to the noise region's launch; base is added at the SOURCE row.
#define MRG_POS_MASK ((1ll << 62) - 1)      // ustart: bit 62 = the region takes the cleared mask elements

struct SrcRange {
    int64_t lo, hi;
    int inv;
};
__device__ __forceinline__ SrcRange unit_source_range(const SvdqParam &pd, int u, const int64_t *__restrict__ ustart) {
    const int64_t us = ustart[u];
    SrcRange s;
    s.inv = (int)((us >> 62) & 1);
    s.lo = (u == pd.unit_begin) ? 0 : (us & MRG_POS_MASK);
    s.hi = (u == pd.unit_begin + pd.unit_count - 1) ? pd.rows : (ustart[u + 1] & MRG_POS_MASK);
    return s;
}

// One wavefront per work unit; chunk = RB = 64 RPL SOURCE rows (256 for N <= 16, 128 above), lane l owns rows
// src + 64 e + l.  The same one-block software pipeline as k_merge_reconstruct: while chunk c is computed from LDS, chunk
// c + 1's mask bytes and base rows and -- from the compacted position where chunk c ends, known once c's mask has been
// counted -- the next RB basis rows and mean values (clamped to the unit; how many of them chunk c + 1 selects is not
// known before its mask is, so a full block is fetched) are in flight into registers.
template <bool U16, int NS, int RPL>
__global__ __launch_bounds__(64) void k_merge_expand(const SvdqParam *__restrict__ params,
                                                     const SvdqUnit *__restrict__ units,
                                                     const int64_t *__restrict__ rows_dev, int NT, int n_sets,
                                                     int per_param, const int32_t *__restrict__ k_in,
                                                     const int32_t *__restrict__ r_in,
                                                     const uint8_t *__restrict__ basis,
                                                     const float *__restrict__ meanbuf, const float *__restrict__ cbar,
                                                     const float *__restrict__ set_share,
                                                     const float *__restrict__ scale_tab,
                                                     const uint8_t *const *__restrict__ mask_ptrs,
                                                     const int64_t *__restrict__ ustart,
                                                     const int32_t *__restrict__ fill_tab,
                                                     const float *const *__restrict__ base_ptrs,
                                                     float *const *__restrict__ out_ptrs) {
    using T = typename UElem<U16>::type;
    constexpr int ES = U16 ? 2 : 4;
    constexpr int RB = 64 * RPL;
    constexpr int NMAX = RPL == 4 ? 16 : 32;
    constexpr int SV = (RB * NMAX * ES / 16 + 2 + 63) / 64;
    // dynamic LDS: the staged basis rows, the staged mean values, the coefficient sets
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x, u = blockIdx.x, n = NT;
    const int ubytes = (int)svdq_align_up((int64_t)RB * n * ES + 48, 16);
    float *M = reinterpret_cast<float *>(lds_raw + ubytes);      // [RB]
    float *C = M + RB;                                           // [column][NS]
    float *SH = C + NS * n;
    const SvdqUnit ud = units[u];
    const int p = ud.param;
    mg_gfloat_w *gout = (mg_gfloat_w *)out_ptrs[p];
    if (!gout) return;      // an entry the caller does not merge
    const SvdqParam pd = params[p];
    const SrcRange sr = unit_source_range(pd, u, ustart);
    if (sr.lo >= sr.hi) return;
    const int64_t D = rows_dev ? rows_dev[p] : pd.rows;      // compacted rows of the region
    int64_t cpos = ud.row0;
    int64_t cend = ud.row0 + ud.nrows;
    if (cend > D) cend = D;
    if (cpos > cend) cpos = cend;
    const int k = k_in[p], r = r_in[p], nl = r - k;
    for (int e = lane; e < NS * n; e += 64) {
        const int i = e / NS, s = e % NS;
        C[e] = s < n_sets ? cbar[((size_t)p * n_sets + s) * n + i] : 0.f;
    }
    if (lane < NS)
        SH[lane] = (set_share && lane < n_sets) ? set_share[(per_param ? (size_t)p * n_sets : 0) + lane] : -1.f;
    const float scale = scale_tab ? scale_tab[p] : 1.f;
    const bool fill = fill_tab ? fill_tab[p] != 0 : false;
    const uint8_t *slab = basis + pd.slab_off;
    const uint8_t *gUh = slab;
    const uint8_t *gUl = slab + svdq_align_up(D * (int64_t)k * ES, 256);
    mg_gfloat *gmean = meanbuf ? (mg_gfloat *)(meanbuf + pd.mean_off) : nullptr;
    mg_gfloat *gbase = base_ptrs ? (mg_gfloat *)base_ptrs[p] : nullptr;
    mg_gbyte *gmask = (mg_gbyte *)mask_ptrs[p];

    // ---- the loads of one chunk into registers
    unsigned mk[RPL];      // mask bytes; 0x100 = past the end of the range: selected by neither polarity
    float bpf[RPL] = {}, mpf[RPL] = {};
    f32x4 ureg[SV];
    auto prefetch = [&](int64_t s0, int64_t c0) {
        const int64_t r0 = s0 + lane;
        if (s0 + RB <= sr.hi) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                mk[e] = (unsigned)gmask[r0 + 64 * e];
                if (gbase) bpf[e] = gbase[r0 + 64 * e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                const bool in = r0 + 64 * e < sr.hi;
                mk[e] = in ? (unsigned)gmask[r0 + 64 * e] : 0x100u;
                if (gbase) bpf[e] = in ? gbase[r0 + 64 * e] : 0.f;
            }
        }
        const int nr = (int)((cend - c0 < RB) ? (cend - c0) : RB);
        if (gmean) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) mpf[e] = (64 * e + lane < nr) ? gmean[c0 + 64 * e + lane] : 0.f;
        }
        const UStage us = ustage_plan<ES>(c0, nr, k, nl);
#pragma unroll
        for (int s = 0; s < SV; ++s) {
            const int v = lane + 64 * s;
            if (v < us.nv) {
                const uint8_t *gp = (v < us.nvh) ? gUh + us.a0h + 16ll * v : gUl + us.a0l + 16ll * (v - us.nvh);
                ureg[s] = *(mg_gf32x4 *)gp;
            }
        }
    };
    prefetch(sr.lo, cpos);
    for (int64_t src = sr.lo; src < sr.hi; src += RB) {
        // ---- the chunk's selection: rank of each of the lane's rows among the chunk's selected rows, and their number
        bool in[RPL], sel[RPL];
        int rank[RPL], count;
        {
            int base = 0;
            const int64_t room = cend - cpos;
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                in[e] = mk[e] != 0x100u;
                const bool sb = sr.inv ? (mk[e] == 0u) : (mk[e] != 0u && mk[e] != 0x100u);
                const unsigned long long bal = __ballot(sb);
                rank[e] = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                sel[e] = sb && rank[e] < room;      // never past the unit's compacted rows, whatever the mask says
                base += (int)__popcll(bal);
            }
            count = base < room ? base : (int)room;
        }
        // ---- registers -> LDS
        const UStage cur = ustage_plan<ES>(cpos, (int)((cend - cpos < RB) ? (cend - cpos) : RB), k, nl);
#pragma unroll
        for (int s = 0; s < SV; ++s) {
            const int v = lane + 64 * s;
            if (v < cur.nv) reinterpret_cast<f32x4 *>(lds_raw)[v] = ureg[s];
        }
        if (gmean) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) M[64 * e + lane] = mpf[e];
        }
        float bv[RPL];
#pragma unroll
        for (int m = 0; m < RPL; ++m) bv[m] = bpf[m];
        lds_fence();
        // ---- the next chunk's loads
        if (src + RB < sr.hi) prefetch(src + RB, cpos + count);
        // ---- compute
        float res[RPL];
#pragma unroll
        for (int m = 0; m < RPL; ++m) res[m] = 0.f;
        if (count > 0) {
            const T *Uh = reinterpret_cast<const T *>(lds_raw) + cur.offh;
            const T *Ul = reinterpret_cast<const T *>(lds_raw + 16 * cur.nvh) + cur.offl;
            int rl[RPL];
            float mv[RPL];
#pragma unroll
            for (int m = 0; m < RPL; ++m) {
                rl[m] = sel[m] ? rank[m] : 0;
                mv[m] = gmean ? M[rl[m]] : 0.f;
            }
            float hi[RPL][NS], lo[RPL][NS];
#pragma unroll
            for (int m = 0; m < RPL; ++m)
#pragma unroll
                for (int s = 0; s < NS; ++s) hi[m][s] = lo[m][s] = 0.f;
            for (int i = 0; i < k; ++i) {
                float c[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) c[s] = C[i * NS + s];
#pragma unroll
                for (int m = 0; m < RPL; ++m) {
                    const float uv = u_val(Uh, rl[m] * k + i);
#pragma unroll
                    for (int s = 0; s < NS; ++s) hi[m][s] = fmaf(uv, c[s], hi[m][s]);
                }
            }
            for (int j = 0; j < nl; ++j) {
                float c[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) c[s] = C[(k + j) * NS + s];
#pragma unroll
                for (int m = 0; m < RPL; ++m) {
                    const float uv = u_val(Ul, rl[m] * nl + j);
#pragma unroll
                    for (int s = 0; s < NS; ++s) lo[m][s] = fmaf(uv, c[s], lo[m][s]);
                }
            }
#pragma unroll
            for (int m = 0; m < RPL; ++m) {
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float v = __fadd_rn(hi[m][s], lo[m][s]);
                    if (gmean) v = __fadd_rn(v, mv[m]);
                    v = __fmul_rn(v, scale);
                    if (set_share) {
                        if (SH[s] >= 0.f) acc = __fadd_rn(acc, __fmul_rn(v, SH[s]));
                    } else if (s == 0) {
                        acc = v;
                    }
                }
                res[m] = sel[m] ? acc : 0.f;
            }
        }
#pragma unroll
        for (int m = 0; m < RPL; ++m) {
            if (in[m] && (sel[m] || fill)) {
                const float v = gbase ? __fadd_rn(bv[m], res[m]) : res[m];      // base + delta (merge.py:429-552)
                gout[src + 64 * m + lane] = v;
            }
        }
        lds_fence();
        cpos += count;
    }
}

// ------------------------------------------------------------------------------------ diagnostics
// One pass over U and the N task deltas of a parameter: per task t the reconstruction U_high c_high[t] + U_low c_low[t]
// (+ mean when add_mean: the reference's diagnostics do NOT add it back) is formed per row and compared with
// delta_t -- N x {sum e^2, sum x^2, sum rec^2, sum |e|, max |e|} per unit, reduced per parameter in a fixed order.
struct DiagPart {
    double se, sx, sr, sa, mx;
};

// One wavefront per work unit.  The reconstruction tile rec[row][task] = sum_c U[row][c] C[c][task] is formed on the
// matrix pipe (v_mfma_f32_16x16x4_f32: rows on M, tasks on N, basis columns on K), the vector ALU only sees the
// epilogue: e = x - rec, packed fmas for the three sums of squares, |e| into the L1 sum and the maximum.  In the result
// layout lane l holds rows 4 (l >> 4) .. + 3 of the 16-row tile for ONE task (l & 15), so a lane's running sums belong to
// one task per 16-task tile -- 5 sums per lane instead of 5 N.
//   block   = RB rows (256 for N <= 8, 128 above): global -> registers one block ahead (the N deltas with vector loads,
//             1 KiB / 512 B contiguous per task and instruction; the run of basis rows with 16-byte loads), registers -> LDS
//             (X[task][row] strips, the two basis parts row-major as they are stored), wave-level fence, issue the next
//             block's loads, compute from LDS.
//   K slots : slot (step s, lane group g = l >> 4) carries one basis column; the SAME map is used for the A operand (U)
//             and the B operand (C, in registers for the whole unit), so any map is a valid contraction.  High part: column
//             KSH g + s (KSH = ceil(k / 4) steps), low part: column KSL g + (s - KSH): a lane's columns of a part are
//             consecutive elements of its row, each step one ds_read at a per-step address that advances by a constant
//             per tile.  Slots past a part's columns carry C = 0 (their U operand is a finite staged value).
//   N <= 8  : SETS = 2 (N <= 4: SETS = 4) -- several 16-row sets share one tile.  SETS = 2: K slots 0..7 carry set A's columns, 8..15 set B's; result columns
//             0..7 are the tasks for set A's rows, 8..15 the same tasks for set B's (C is zero in the off-diagonal
//             blocks), so no result register is idle.
//   N > 16  : two task tiles per row tile (the A operand is shared).
// Sums: fp32 inside a block (two chains of <= 16 terms per lane), fp64 across blocks; rows past the block's count take
// the masked form of the tile (exact zeros).  WALK: the block is a chunk of RB SOURCE rows of a masked region
// (k_merge_expand's walk); selected rows are written to X at their rank, so the compute phase is the same loop over
// `count` compacted rows; the next chunk's basis run starts where this one ends, RB rows (clamped to the unit) are fetched.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x2 mg_gf32x2;

__device__ __forceinline__ float lds_u(const uint8_t *base, int byte_off, __half) {
    return __half2float(*reinterpret_cast<const __half *>(base + byte_off));
}
__device__ __forceinline__ float lds_u(const uint8_t *base, int byte_off, float) {
    return *reinterpret_cast<const float *>(base + byte_off);
}

template <int NTP, int RPL_, int SETS> struct DiagGeom {
    static constexpr int RPL = RPL_;                     // rows per lane and block (4 when row sets are packed)
    static constexpr int RB = 64 * RPL;                  // rows per block
    static constexpr int TT = NTP > 16 ? 2 : 1;          // 16-task tiles
    static constexpr int TROWS = 16 * SETS;              // rows per MFMA tile
    static constexpr int KMAX = SETS > 1 ? 4 : NTP / 4 + 1; // K steps of 4 slots
};
__host__ __device__ constexpr int diag_xs(int rb, bool walk) { return rb + 4 + (walk ? 64 : 0); }   // X strip stride (floats)

template <bool B> struct DiagBool { static constexpr bool value = B; };

// FULL: the plan has exactly NTP tasks -- no "task t is real" tests (each one is a scalar-register pair the compiler
// keeps across the loop; with twenty of them the task pointers are pushed out of the scalar registers)
#ifdef SVDQ_DIAG_WPE
#define SVDQ_DIAG_ATTR __attribute__((amdgpu_waves_per_eu(SVDQ_DIAG_WPE, SVDQ_DIAG_WPE)))
#else
#define SVDQ_DIAG_ATTR
#endif
template <int NTP, int RPL_, int SETS, bool FULL, bool U16, bool WALK>
__global__ __launch_bounds__(64) SVDQ_DIAG_ATTR void k_diag(const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
                                             const float *const *__restrict__ ptrs,
                                             const uint8_t *const *__restrict__ mask_ptrs,
                                             const int64_t *__restrict__ ustart, const int64_t *__restrict__ rows_dev,
                                             int NT, const int32_t *__restrict__ k_in, const int32_t *__restrict__ r_in,
                                             const uint8_t *__restrict__ basis, const float *__restrict__ meanbuf,
                                             int add_mean, const float *__restrict__ ctask /* [P][N][N] */,
                                             DiagPart *__restrict__ part /* [n_units][N] */) {
    using T = typename UElem<U16>::type;
    using G = DiagGeom<NTP, RPL_, SETS>;
    static_assert(SETS == 1 || (RPL_ == 4 && NTP * SETS <= 16), "packed row sets: 16 / SETS tasks each, 256-row blocks");
    constexpr bool PACK = SETS > 1;
    constexpr int TPS = 16 / SETS;      // tasks (and K slots) per row set
    constexpr int SPS = 4 / SETS;       // K steps per row set
    constexpr int ES = U16 ? 2 : 4;
    constexpr int RPL = G::RPL, RB = G::RB, TT = G::TT, TROWS = G::TROWS, KMAX = G::KMAX;
    constexpr int XSD = diag_xs(RB, WALK);
    constexpr int SV = (RB * NTP * ES / 16 + 2 + 63) / 64;      // 16-byte vectors of one block's basis rows per lane
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x, n = FULL ? NTP : NT, u = blockIdx.x;
    const int g = lane >> 4, col = lane & 15;
    const int ubytes = (int)svdq_align_up((int64_t)RB * n * ES + 48, 16);
    uint8_t *Ubuf = lds_raw;
    float *X = reinterpret_cast<float *>(lds_raw + ubytes);      // [NTP tasks + mean][XSD]
    const SvdqUnit ud = units[u];
    const int p = ud.param;
    const SvdqParam pd = params[p];
    const int64_t D = rows_dev ? rows_dev[p] : pd.rows;
    int64_t cpos = ud.row0;
    int64_t cend = ud.row0 + ud.nrows;
    if (cend > D) cend = D;
    // the task whose column this lane holds in each task tile (packed: every row set carries tasks 0 .. TPS - 1)
    int task[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) task[tt] = PACK ? (col % TPS) : col + 16 * tt;
    double se[TT], sx[TT], sr[TT], sa[TT];
    float mx[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        se[tt] = sx[tt] = sr[tt] = sa[tt] = 0.0;
        mx[tt] = 0.f;
    }
    if (cpos < cend) {      // wave-uniform
        const int k = k_in[p], r = r_in[p], nl = r - k;
        const int KSH = PACK ? 0 : (k + 3) >> 2, KSL = PACK ? 0 : (nl + 3) >> 2;
        const int KT = PACK ? 4 : KSH + KSL;
        // basis column of K slot (step s, this lane's group), -1 = the slot is idle
        auto slot_col = [&](int s) -> int {
            if constexpr (PACK) {      // step s belongs to row set s / SPS; the lane group's columns are consecutive
                const int c = SPS * g + (s % SPS);
                return c < r ? c : -1;
            } else {
                if (s < KSH) {
                    const int c = KSH * g + s;
                    return c < k ? c : -1;
                }
                const int c = KSL * g + (s - KSH);
                return (s < KT && c < nl) ? k + c : -1;
            }
        };
        // B operand: the coefficients of the lane's task(s), one register per step, for the whole unit
        float creg[TT][KMAX];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const int c = slot_col(s);
                bool on = c >= 0 && task[tt] < n;
                if constexpr (PACK) on = on && ((s / SPS) == (col / TPS));      // a set's slots feed its own result columns only
                creg[tt][s] = on ? ctask[(size_t)p * n * n + (size_t)task[tt] * n + c] : 0.f;
            }
        const uint8_t *slab = basis + pd.slab_off;
        const uint8_t *gUh = slab;
        const uint8_t *gUl = slab + svdq_align_up(D * (int64_t)k * ES, 256);
        mg_gfloat *gmean = (add_mean && meanbuf) ? (mg_gfloat *)(meanbuf + pd.mean_off) : nullptr;
        mg_gbyte *gmask = WALK ? (mg_gbyte *)mask_ptrs[p] : nullptr;
        mg_gfloat *dp[NTP];
#pragma unroll
        for (int t = 0; t < NTP; ++t) dp[t] = (mg_gfloat *)ptrs[(size_t)p * n + (t < n ? t : 0)];
        int64_t src = cpos, src_hi = cend;      // plain: source rows = compacted rows
        int inv = 0;
        if constexpr (WALK) {
            const SrcRange rg = unit_source_range(pd, u, ustart);
            src = ustart[u] & MRG_POS_MASK;      // nothing is selected in front of the unit's first element
            src_hi = rg.hi;
            inv = rg.inv;
        }
        // ---- the loads of one block into registers
        float xpf[NTP][RPL] = {}, mpf[RPL] = {};
        unsigned mkpf[RPL] = {};
        f32x4 ureg[SV];
        auto prefetch = [&](int64_t s0, int64_t c0) {
            if constexpr (WALK) {      // lane l owns source rows s0 + 64 e + l
                const int64_t r0 = s0 + lane;
                if (s0 + RB <= src_hi) {
#pragma unroll
                    for (int e = 0; e < RPL; ++e) mkpf[e] = (unsigned)gmask[r0 + 64 * e];
#pragma unroll
                    for (int t = 0; t < NTP; ++t)
                        if (t < n) {      // wave-uniform
#pragma unroll
                            for (int e = 0; e < RPL; ++e) xpf[t][e] = dp[t][r0 + 64 * e];
                        }
                } else {
                    bool in[RPL];
#pragma unroll
                    for (int e = 0; e < RPL; ++e) {
                        in[e] = r0 + 64 * e < src_hi;
                        mkpf[e] = in[e] ? (unsigned)gmask[r0 + 64 * e] : 0x100u;
                    }
#pragma unroll
                    for (int t = 0; t < NTP; ++t)
                        if (t < n) {
#pragma unroll
                            for (int e = 0; e < RPL; ++e) xpf[t][e] = in[e] ? dp[t][r0 + 64 * e] : 0.f;
                        }
                }
            } else {                   // lane l owns rows s0 + RPL l .. + RPL - 1: one vector load per task
                const int64_t r0 = s0 + RPL * lane;
                if (s0 + RB <= src_hi) {
#pragma unroll
                    for (int t = 0; t < NTP; ++t)
                        if (t < n) {
                            if constexpr (RPL == 4) {
                                const f32x4 v = *reinterpret_cast<mg_gf32x4 *>(dp[t] + r0);
                                xpf[t][0] = v.x, xpf[t][1] = v.y, xpf[t][2] = v.z, xpf[t][3] = v.w;
                            } else if constexpr (RPL == 2) {
                                const f32x2 v = *reinterpret_cast<mg_gf32x2 *>(dp[t] + r0);
                                xpf[t][0] = v.x, xpf[t][1] = v.y;
                            } else {
                                xpf[t][0] = dp[t][r0];
                            }
                        }
                } else {
#pragma unroll
                    for (int t = 0; t < NTP; ++t)
                        if (t < n) {
#pragma unroll
                            for (int e = 0; e < RPL; ++e) xpf[t][e] = (r0 + e < src_hi) ? dp[t][r0 + e] : 0.f;
                        }
                }
            }
            const int nr = (int)((cend - c0 < RB) ? (cend - c0) : RB);
            if (gmean) {      // (add_mean only) compacted rows c0 + 64 e + lane
#pragma unroll
                for (int e = 0; e < RPL; ++e) mpf[e] = (64 * e + lane < nr) ? gmean[c0 + 64 * e + lane] : 0.f;
            }
            const UStage us = ustage_plan<ES>(c0, nr, k, nl);
#pragma unroll
            for (int s = 0; s < SV; ++s) {
                const int v = lane + 64 * s;
                if (v < us.nv) {
                    const uint8_t *gp = (v < us.nvh) ? gUh + us.a0h + 16ll * v : gUl + us.a0l + 16ll * (v - us.nvh);
                    ureg[s] = *(mg_gf32x4 *)gp;
                }
            }
        };
        prefetch(src, cpos);
        while (true) {
            // ---- selection of the block whose data sits in the registers
            int count;
            int slot[RPL];      // WALK: where the lane's rows go in the strips
            if constexpr (WALK) {
                int base = 0;
                const int64_t room = cend - cpos;
#pragma unroll
                for (int e = 0; e < RPL; ++e) {
                    const bool s = inv ? (mkpf[e] == 0u) : (mkpf[e] != 0u && mkpf[e] != 0x100u);
                    const unsigned long long bal = __ballot(s);
                    const int rank = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    // a selected row goes to its rank, any other to the lane's dump slot (no branch around the writes)
                    slot[e] = (s && rank < room) ? rank : RB + 4 + lane;
                    base += (int)__popcll(bal);
                }
                count = base < room ? base : (int)room;
            } else {
                count = (int)((src_hi - src < RB) ? (src_hi - src) : RB);
            }
            // ---- registers -> LDS (the staging layout is a function of the block's first row: recomputed, not kept)
            const UStage cur = ustage_plan<ES>(cpos, (int)((cend - cpos < RB) ? (cend - cpos) : RB), k, nl);
#pragma unroll
            for (int s = 0; s < SV; ++s) {
                const int v = lane + 64 * s;
                if (v < cur.nv) reinterpret_cast<f32x4 *>(Ubuf)[v] = ureg[s];
            }
            if (gmean) {
#pragma unroll
                for (int e = 0; e < RPL; ++e) X[NTP * XSD + 64 * e + lane] = mpf[e];
            }
#pragma unroll
            for (int t = 0; t < NTP; ++t) {      // strips past the plan's tasks hold don't-cares (their C is zero)
                if constexpr (WALK) {
#pragma unroll
                    for (int e = 0; e < RPL; ++e) X[t * XSD + slot[e]] = xpf[t][e];
                } else if constexpr (RPL == 4) {
                    f32x4 v = {xpf[t][0], xpf[t][1], xpf[t][2], xpf[t][3]};
                    *reinterpret_cast<f32x4 *>(X + t * XSD + 4 * lane) = v;
                } else if constexpr (RPL == 2) {
                    f32x2 v = {xpf[t][0], xpf[t][1]};
                    *reinterpret_cast<f32x2 *>(X + t * XSD + 2 * lane) = v;
                } else {
                    X[t * XSD + lane] = xpf[t][0];
                }
            }
            lds_fence();
            // ---- the next block's loads
            const int64_t nsrc = src + RB, ncpos = cpos + count;
            const bool more = nsrc < src_hi && ncpos < cend;
            if (more) prefetch(nsrc, ncpos);
            // ---- compute
            if (count > 0) {
                // per step: LDS byte address of the lane's element in the first tile, advance per tile
                int ua[KMAX], ust[KMAX];
                {
                    const int baseH = cur.offh * ES, baseL = 16 * cur.nvh + cur.offl * ES;
#pragma unroll
                    for (int s = 0; s < KMAX; ++s) {
                        int c = slot_col(s);
                        if (c < 0) c = 0;                                   // an idle slot reads a staged value (times C = 0)
                        const int row = col + (PACK ? 16 * (s / SPS) : 0);
                        const bool hi = c < k;
                        const int w = hi ? k : nl;
                        ua[s] = (hi ? baseH : baseL) + (row * w + (hi ? c : c - k)) * ES;
                        ust[s] = TROWS * w * ES;
                    }
                }
                f32x2 se2[TT], sx2[TT], sr2[TT];
                float sa0[TT], sa1[TT];
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    se2[tt] = sx2[tt] = sr2[tt] = f32x2{0.f, 0.f};
                    sa0[tt] = sa1[tt] = 0.f;
                }
                // the A operands of one tile: all KMAX steps, no branch -- a step past the parameter's columns reads a staged
                // value against C = 0 (at most one such step unless the plan has fewer tasks than the variant is padded to)
                auto load_a = [&](float (&a)[KMAX]) {
#pragma unroll
                    for (int s = 0; s < KMAX; ++s) {
                        a[s] = lds_u(Ubuf, ua[s], T{});
                        ua[s] += ust[s];
                    }
                };
                auto tile = [&](int ti, const float (&a_in)[KMAX], auto masked_c) {
                    constexpr bool MASKED = decltype(masked_c)::value;
                    float a[KMAX];
#pragma unroll
                    for (int s = 0; s < KMAX; ++s) {
                        a[s] = a_in[s];
                        if constexpr (MASKED) {
                            // the block's last tile: rows past its count were never staged; whatever sits there must not
                            // reach the matrix pipe -- a NaN times a zero coefficient is a NaN in a VALID row's sum (packed:
                            // set B's rows feed set A's result columns through the zero blocks of C)
                            const int arow = ti * TROWS + col + (PACK ? 16 * (s / SPS) : 0);
                            a[s] = arow < count ? a[s] : 0.f;
                        }
                    }
                    const int rb = ti * TROWS + (PACK ? 16 * (col / TPS) : 0) + 4 * g;      // the lane's four rows
                    f32x4 x[TT], acc[TT];
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) {
                        x[tt] = *reinterpret_cast<const f32x4 *>(X + task[tt] * XSD + rb);
                        acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int s = 0; s < KMAX; ++s)      // the task tiles' chains are independent: interleaved
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt)
                            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], creg[tt][s], acc[tt], 0, 0, 0);
                    if (gmean) {
                        const f32x4 m = *reinterpret_cast<const f32x4 *>(X + NTP * XSD + rb);
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt) {
                            acc[tt].x = __fadd_rn(acc[tt].x, m.x), acc[tt].y = __fadd_rn(acc[tt].y, m.y);
                            acc[tt].z = __fadd_rn(acc[tt].z, m.z), acc[tt].w = __fadd_rn(acc[tt].w, m.w);
                        }
                    }
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) {
                        if constexpr (MASKED) {      // rows past the block's count contribute exact zeros
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const bool ok = rb + v < count;
                                x[tt][v] = ok ? x[tt][v] : 0.f;
                                acc[tt][v] = ok ? acc[tt][v] : 0.f;
                            }
                        }
                        const f32x2 m1 = {-1.f, -1.f};
                        const f32x2 x01 = {x[tt].x, x[tt].y}, x23 = {x[tt].z, x[tt].w};
                        const f32x2 c01 = {acc[tt].x, acc[tt].y}, c23 = {acc[tt].z, acc[tt].w};
                        const f32x2 e01 = __builtin_elementwise_fma(c01, m1, x01);      // x - rec, one rounding
                        const f32x2 e23 = __builtin_elementwise_fma(c23, m1, x23);
                        se2[tt] = __builtin_elementwise_fma(e01, e01, se2[tt]);
                        se2[tt] = __builtin_elementwise_fma(e23, e23, se2[tt]);
                        sx2[tt] = __builtin_elementwise_fma(x01, x01, sx2[tt]);
                        sx2[tt] = __builtin_elementwise_fma(x23, x23, sx2[tt]);
                        sr2[tt] = __builtin_elementwise_fma(c01, c01, sr2[tt]);
                        sr2[tt] = __builtin_elementwise_fma(c23, c23, sr2[tt]);
                        sa0[tt] += fabsf(e01.x) + fabsf(e01.y);
                        sa1[tt] += fabsf(e23.x) + fabsf(e23.y);
                        // a NaN error reaches max through se (k_diag_finish)
                        mx[tt] = fmaxf(mx[tt], fmaxf(fmaxf(fabsf(e01.x), fabsf(e01.y)), fmaxf(fabsf(e23.x), fabsf(e23.y))));
                    }
                };
                // the next tile's operands are read before this tile's chain is issued (the read one past the last tile
                // stays inside the workgroup's LDS: the strips follow the basis rows)
                const int nfull = count / TROWS;
                float a0[KMAX], a1[KMAX];
                load_a(a0);
                for (int ti = 0; ti < nfull; ++ti) {
                    load_a(a1);
                    tile(ti, a0, DiagBool<false>{});
#pragma unroll
                    for (int s = 0; s < KMAX; ++s) a0[s] = a1[s];
                }
                if (nfull * TROWS < count) tile(nfull, a0, DiagBool<true>{});
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    se[tt] += (double)(se2[tt].x + se2[tt].y);
                    sx[tt] += (double)(sx2[tt].x + sx2[tt].y);
                    sr[tt] += (double)(sr2[tt].x + sr2[tt].y);
                    sa[tt] += (double)(sa0[tt] + sa1[tt]);
                }
            }
            if (!more) break;
            src = nsrc;
            cpos = ncpos;
            lds_fence();      // the strips and the basis rows are rewritten next
        }
    }
    // the lanes that hold one task's columns meet in a fixed order (xor butterflies); one lane writes the unit's partial
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        double qa = se[tt], qb = sx[tt], qc = sr[tt], qd = sa[tt];
        float q = mx[tt];
#pragma unroll
        for (int off = 32; off >= (PACK ? TPS : 16); off >>= 1) {
            qa += __shfl_xor(qa, off);
            qb += __shfl_xor(qb, off);
            qc += __shfl_xor(qc, off);
            qd += __shfl_xor(qd, off);
            const float o = __shfl_xor(q, off);
            q = (q != q) ? q : ((o != o) ? o : (o > q ? o : q));
        }
        if (lane < (PACK ? TPS : 16) && task[tt] < n) {
            DiagPart &dst = part[(size_t)blockIdx.x * n + task[tt]];
            dst.se = qa;
            dst.sx = qb;
            dst.sr = qc;
            dst.sa = qd;
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
        if (se != se) mx = se;      // an error that is NaN: torch.max propagates it (the units keep fmax of the others)
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

// the streaming launch of svdq_merge_reconstruct (mask_ptrs == NULL) and svdq_merge_masked (source walk)
static int launch_reconstruct(const char *who, const svdq_plan *pl, const int64_t *rows_dev, const void *small,
                              const void *basis, const float *mean, const float *cbar, int32_t n_sets,
                              int32_t per_param, const float *set_share, const float *scale, const void *mask_ptrs,
                              const int64_t *unit_start, const int32_t *fill, const void *base_ptrs,
                              const void *out_ptrs, void *stream) {
    if (int rc = merge_args_ok(pl, small, n_sets, who)) return rc;
    if (!basis || !cbar || !out_ptrs) {
        svdq_set_error("%s: basis, cbar and out_ptrs are required", who);
        return SVDQ_EINVAL;
    }
    if (n_sets > MRG_MAX_SETS) {
        svdq_set_error("%s: at most %d sets (clusters) per pass, got %d", who, MRG_MAX_SETS, n_sets);
        return SVDQ_EUNSUPPORTED;
    }
    if (n_sets > 1 && !set_share) {
        svdq_set_error("%s: set_share is required when n_sets > 1", who);
        return SVDQ_EINVAL;
    }
    const svdq_small_layout &L = pl->small;
    const uint8_t *sm = reinterpret_cast<const uint8_t *>(small);
    auto kk = reinterpret_cast<const int32_t *>(sm + L.k_off), rr = reinterpret_cast<const int32_t *>(sm + L.r_off);
    auto bp = reinterpret_cast<const float *const *>(base_ptrs);
    auto op = reinterpret_cast<float *const *>(out_ptrs);
    auto mp = reinterpret_cast<const uint8_t *const *>(mask_ptrs);
    hipStream_t st = (hipStream_t)stream;
    const int ns = n_sets == 1 ? 1 : (n_sets == 2 ? 2 : (n_sets <= 4 ? 4 : 8));
    const int rpl = pl->n_tasks <= 16 ? 4 : 2;      // rows per lane and block (256 / 128 rows: a block's basis rows stay <= 17 vectors per lane)
    const size_t lds = (size_t)svdq_align_up((int64_t)64 * rpl * pl->n_tasks * (pl->cfg.fp16 ? 2 : 4) + 48, 16) +
                       (mp ? (size_t)64 * rpl * 4 : 0) + (size_t)(ns * pl->n_tasks + ns) * 4;
    const uint8_t *bs = reinterpret_cast<const uint8_t *>(basis);
    const float *mn = pl->cfg.center ? mean : nullptr;
#define SVDQ_MRG_LAUNCH(F16, NS_)                                                                                      \
    do {                                                                                                               \
        if (mp && rpl == 4)                                                                                            \
            hipLaunchKernelGGL((k_merge_expand<F16, NS_, 4>), dim3(pl->n_units), dim3(64), lds, st, pl->d_params,       \
                               pl->d_units, rows_dev, pl->n_tasks, n_sets, per_param, kk, rr, bs, mn, cbar, set_share,  \
                               scale, mp, unit_start, fill, bp, op);                                                   \
        else if (mp)                                                                                                   \
            hipLaunchKernelGGL((k_merge_expand<F16, NS_, 2>), dim3(pl->n_units), dim3(64), lds, st, pl->d_params,       \
                               pl->d_units, rows_dev, pl->n_tasks, n_sets, per_param, kk, rr, bs, mn, cbar, set_share,  \
                               scale, mp, unit_start, fill, bp, op);                                                   \
        else if (rpl == 4)                                                                                             \
            hipLaunchKernelGGL((k_merge_reconstruct<F16, NS_, 4>), dim3(pl->n_units), dim3(64), lds, st, pl->d_params,  \
                               pl->d_units, rows_dev, pl->n_tasks, n_sets, per_param, kk, rr, bs, mn, cbar, set_share,  \
                               scale, bp, op);                                                                         \
        else                                                                                                           \
            hipLaunchKernelGGL((k_merge_reconstruct<F16, NS_, 2>), dim3(pl->n_units), dim3(64), lds, st, pl->d_params,  \
                               pl->d_units, rows_dev, pl->n_tasks, n_sets, per_param, kk, rr, bs, mn, cbar, set_share,  \
                               scale, bp, op);                                                                         \
    } while (0)
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

extern "C" int svdq_merge_reconstruct(const svdq_plan *pl, const int64_t *rows_dev, const void *small, const void *basis,
                                      const float *mean, const float *cbar, int32_t n_sets, int32_t per_param,
                                      const float *set_share, const float *scale, const void *base_ptrs,
                                      const void *out_ptrs, void *stream) {
    return launch_reconstruct("svdq_merge_reconstruct", pl, rows_dev, small, basis, mean, cbar, n_sets, per_param,
                              set_share, scale, nullptr, nullptr, nullptr, base_ptrs, out_ptrs, stream);
}

extern "C" int svdq_merge_masked(const svdq_plan *pl, const int64_t *rows_dev, const void *small, const void *basis,
                                 const float *mean, const float *weights, const int32_t *order, int32_t n_sets,
                                 int32_t per_param, const float *set_share, const float *scale, const void *mask_ptrs,
                                 const int64_t *unit_start, const int32_t *fill, const void *base_ptrs,
                                 const void *out_ptrs, void *work, void *stream) {
    if (!work || !mask_ptrs || !unit_start || !rows_dev) {
        svdq_set_error("svdq_merge_masked: work, mask_ptrs, unit_start and rows_dev are required");
        return SVDQ_EINVAL;
    }
    float *cbar = reinterpret_cast<float *>(work);
    if (int rc = svdq_merge_coeffs(pl, small, weights, order, n_sets, per_param, cbar, stream)) return rc;
    return launch_reconstruct("svdq_merge_masked", pl, rows_dev, small, basis, mean, cbar, n_sets, per_param, set_share,
                              scale, mask_ptrs, unit_start, fill, base_ptrs, out_ptrs, stream);
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

#ifndef SVDQ_DIAG_RPL_MID
#define SVDQ_DIAG_RPL_MID 2      // rows per lane and block of the 9..24-task variants (A/B builds)
#endif
template <int NTP, int RPL, int SETS, bool FULL>
static void launch_diag(const svdq_plan *pl, const void *ptrs, const void *mask_ptrs, const int64_t *unit_start,
                        const int64_t *rows_dev, const int32_t *kk, const int32_t *rr, const uint8_t *basis,
                        const float *mean, int add_mean, const float *ctask, DiagPart *part, hipStream_t st) {
    auto pp = reinterpret_cast<const float *const *>(ptrs);
    auto mp = reinterpret_cast<const uint8_t *const *>(mask_ptrs);
    using G = DiagGeom<NTP, RPL, SETS>;
    const int n = pl->n_tasks, es = pl->cfg.fp16 ? 2 : 4;
    const size_t lds = (size_t)svdq_align_up((int64_t)G::RB * n * es + 48, 16) +
                       (size_t)(NTP + 1) * diag_xs(G::RB, mp != nullptr) * 4;
#define SVDQ_DIAG_LAUNCH(F16, WALK_)                                                                                   \
    hipLaunchKernelGGL((k_diag<NTP, RPL, SETS, FULL, F16, WALK_>), dim3(pl->n_units), dim3(64), lds, st, pl->d_params, \
                       pl->d_units, pp, mp, unit_start, rows_dev, n, kk, rr, basis, mean, add_mean, ctask, part)
    if (pl->cfg.fp16) {
        if (mp) SVDQ_DIAG_LAUNCH(true, true); else SVDQ_DIAG_LAUNCH(true, false);
    } else {
        if (mp) SVDQ_DIAG_LAUNCH(false, true); else SVDQ_DIAG_LAUNCH(false, false);
    }
#undef SVDQ_DIAG_LAUNCH
}

static int run_diagnostics(const char *who, const svdq_plan *pl, const void *delta_ptrs, const void *mask_ptrs,
                           const int64_t *unit_start, const int64_t *rows_dev, const void *small, const void *basis,
                           const float *mean, int32_t add_mean, double *out, void *work, void *stream) {
    if (!pl || !delta_ptrs || !small || !basis || !out || !work) {
        svdq_set_error("%s: bad argument", who);
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
    // variants by padded task count (the prefetch registers and the X strips are sized by it), each also for plans with
    // exactly that many tasks
#define SVDQ_DIAG_ARGS pl, delta_ptrs, mask_ptrs, unit_start, rows_dev, kk, rr, bs, mean, add_mean, ctask, part, st
#define SVDQ_DIAG_PICK(NTP_, RPL_, PACK_)                                                                              \
    do {                                                                                                               \
        if (n == NTP_) launch_diag<NTP_, RPL_, PACK_, true>(SVDQ_DIAG_ARGS);                                            \
        else launch_diag<NTP_, RPL_, PACK_, false>(SVDQ_DIAG_ARGS);                                                     \
    } while (0)
    if (n <= 4) SVDQ_DIAG_PICK(4, 4, 4);
    else if (n <= 8) SVDQ_DIAG_PICK(8, 4, 2);
    else if (n <= 12) SVDQ_DIAG_PICK(12, SVDQ_DIAG_RPL_MID, 1);
    else if (n <= 16) SVDQ_DIAG_PICK(16, SVDQ_DIAG_RPL_MID, 1);
    else if (n <= 20) SVDQ_DIAG_PICK(20, SVDQ_DIAG_RPL_MID, 1);
    else if (n <= 24) SVDQ_DIAG_PICK(24, SVDQ_DIAG_RPL_MID, 1);
    else if (n <= 28) SVDQ_DIAG_PICK(28, 1, 1);
    else if (n <= 32) SVDQ_DIAG_PICK(32, 1, 1);
    else {
        svdq_set_error("%s: unsupported task count %d", who, (int)n);
        return SVDQ_EUNSUPPORTED;
    }
#undef SVDQ_DIAG_PICK
#undef SVDQ_DIAG_ARGS
    hipLaunchKernelGGL(k_diag_finish, dim3(pl->n_params, (int)n), dim3(64), 0, st, pl->d_params, rows_dev, (int)n, part, out);
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

extern "C" int svdq_diagnostics(const svdq_plan *pl, const void *delta_ptrs, const int64_t *rows_dev, const void *small,
                                const void *basis, const float *mean, int32_t add_mean, double *out, void *work,
                                void *stream) {
    return run_diagnostics("svdq_diagnostics", pl, delta_ptrs, nullptr, nullptr, rows_dev, small, basis, mean, add_mean,
                           out, work, stream);
}

extern "C" int svdq_diagnostics_masked(const svdq_plan *pl, const void *delta_ptrs, const void *mask_ptrs,
                                       const int64_t *unit_start, const int64_t *rows_dev, const void *small,
                                       const void *basis, const float *mean, int32_t add_mean, double *out, void *work,
                                       void *stream) {
    if (!mask_ptrs || !unit_start || !rows_dev) {
        svdq_set_error("svdq_diagnostics_masked: mask_ptrs, unit_start and rows_dev are required");
        return SVDQ_EINVAL;
    }
    return run_diagnostics("svdq_diagnostics_masked", pl, delta_ptrs, mask_ptrs, unit_start, rows_dev, small, basis, mean,
                           add_mean, out, work, stream);
}

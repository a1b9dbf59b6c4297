// svdq_project.hip -- pass 2 of the SVD-Hybrid compressor: U = Tc W -> fp16, mean, c = fp16(U)^T Tc (k_basis_project and its
// N > 16 forms).  Helpers and the layout notes: svdq_stream.h.
#include "svdq_project_unit.h"

// ------------------------------------------------------------------------------------ N > 16: two waves
// For 16 < N <= 32 the single-wave pass 2 needs 300+ registers (one wave per SIMD, nothing to overlap the
// memory phases with).  This variant runs TWO wavefronts per workgroup on one shared LDS strip:
//   * each wave loads and centres HALF of the tasks (half the prefetch registers); the row sums of the two
//     halves meet in LDS (mean = (sum of wave 0's tasks + sum of wave 1's tasks) / N, in that order);
//   * pass 2: wave w computes the 16-column block w of U (and of the rounding-correction MFMA), so the
//     accumulator-side registers halve as well; both stage into the same output images.
// (A two-wave pass 1 -- each wave the full 2x2-blocked Gram over half of the sub-tiles -- was measured 10 %
// SLOWER than the single-wave k_gram at N = 20 and is not kept.  A four-wave pass 2 -- columns x row halves, a
// quarter of the tasks loaded per wave -- is 5 % faster at N = 32 but 5 % slower at N = 20: not kept either.)
// Barriers are s_barrier after an LDS-only wait: a full __syncthreads() would also drain the prefetch.
__device__ __forceinline__ void wg_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int HT, bool GATHER>
__device__ __forceinline__ f32x4 center_store_half(const f32x4 (&v)[HT], int t0, int NT, int center, float *X,
                                                   f32x4 *SUM, int wv, int lane) {
    f32x4 s = zero4();
#pragma unroll
    for (int i = 0; i < HT; ++i) {
        const f32x4 x = (t0 + i < NT) ? v[i] : zero4();
        s += x;
    }
    SUM[wv * 64 + lane] = s;
    wg_sync();
    const f32x4 tot = SUM[lane] + SUM[64 + lane];
    f32x4 mean = zero4();
    if (center) {
        const float n = (float)NT;
        mean.x = tot.x / n;
        mean.y = tot.y / n;
        mean.z = tot.z / n;
        mean.w = tot.w / n;
    }
#pragma unroll
    for (int i = 0; i < HT; ++i) {
        const f32x4 xc = (t0 + i < NT) ? (v[i] - mean) : zero4();
        if constexpr (GATHER) {
#pragma unroll
            for (int e = 0; e < 4; ++e) X[(t0 + i) * XS + 64 * e + lane] = xc[e];
        } else {
            *reinterpret_cast<f32x4 *>(X + (t0 + i) * XS + 4 * lane) = xc;
        }
    }
    return mean;
}

// first block + one-block-ahead loads of one wave's HT tasks (contiguous or through the index list)
template <int HT, int MODE>
struct HalfLoader {
    static constexpr bool GATHER = (MODE & 1) != 0, SUB = (MODE & 2) != 0;
    gfloat *bp[HT];
    gint *gidx;
    gfloat *gbase;
    i32x4 ixn;
    f32x4 vb;
    int64_t D, r_end;
    int lane;
    __device__ __forceinline__ void apply(f32x4 (&v)[HT]) {  // minus-base mode: v holds fine-tuned rows
        if constexpr (SUB) {
#pragma unroll
            for (int i = 0; i < HT; ++i) v[i] = v[i] - vb;
        }
    }
    __device__ __forceinline__ void first(f32x4 (&v)[HT], int64_t r_begin) {
        ixn = i32x4{-1, -1, -1, -1};
        vb = zero4();
        if (r_begin >= r_end) return;
        if constexpr (GATHER) {
            const i32x4 ix0 = load_idx(gidx, r_begin, D, lane);
            load_block_gather<HT>(v, bp, ix0, r_begin + SVDQ_BLK_ROWS <= D);
            if constexpr (SUB) vb = load_base_gather(gbase, ix0, r_begin + SVDQ_BLK_ROWS <= D);
            if (r_begin + SVDQ_BLK_ROWS < r_end) ixn = load_idx(gidx, r_begin + SVDQ_BLK_ROWS, D, lane);
        } else {
            if constexpr (SUB) vb = load_base(gbase, r_begin, D, lane);
            load_block<HT>(v, bp, r_begin, D, lane);
        }
    }
    __device__ __forceinline__ void next(f32x4 (&v)[HT], int64_t rb) {  // data of block rb + 256
        if (rb + SVDQ_BLK_ROWS >= r_end) return;
        if constexpr (GATHER) {
            load_block_gather<HT>(v, bp, ixn, rb + 2 * SVDQ_BLK_ROWS <= D);
            if constexpr (SUB) vb = load_base_gather(gbase, ixn, rb + 2 * SVDQ_BLK_ROWS <= D);
            if (rb + 2 * SVDQ_BLK_ROWS < r_end) ixn = load_idx(gidx, rb + 2 * SVDQ_BLK_ROWS, D, lane);
        } else {
            load_block<HT>(v, bp, rb + SVDQ_BLK_ROWS, D, lane);
            if constexpr (SUB) vb = load_base(gbase, rb + SVDQ_BLK_ROWS, D, lane);
        }
    }
};

__device__ __forceinline__ void copy_out_wg(const void *lds_src, uint8_t *gdst, int nbytes, int tid) {
    const int nvec = nbytes >> 4;
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(lds_src);
    f32x4 *d4 = reinterpret_cast<f32x4 *>(gdst);
    for (int i = tid; i < nvec; i += 128) d4[i] = s4[i];
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(lds_src);
    for (int b = (nvec << 4) + 2 * tid; b < nbytes; b += 256)
        *reinterpret_cast<uint16_t *>(gdst + b) = *reinterpret_cast<const uint16_t *>(sb + b);
}

template <int NTP, bool OUT16, int MODE>
__global__ __launch_bounds__(128) void k_basis_project2(
    const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
    const float *const *__restrict__ ptrs, const int64_t *__restrict__ rows_dev, int NT, int center,
    const float *__restrict__ Wtab, const int32_t *__restrict__ k_dev, const int32_t *__restrict__ r_dev,
    uint8_t *__restrict__ basis, float *__restrict__ meanbuf, double *__restrict__ cpart, int unit0, int reverse,
    const void *const *__restrict__ aux, const void *const *__restrict__ aux2) {
    static_assert(NTP > 16 && NTP <= 32 && NTP % 4 == 0, "two-wave variant is for 16 < N <= 32");
    constexpr bool GATHER = (MODE & 1) != 0;
    using out_t = typename OutT<OUT16>::type;
    constexpr int HT = NTP / 2;
    constexpr int KS = NTP / 4;
    constexpr int ES = OUT16 ? 2 : 4;
    __shared__ __attribute__((aligned(16))) float X[NTP * XS];
    __shared__ __attribute__((aligned(16))) out_t OUT[SVDQ_BLK_ROWS * NTP + 16];
    __shared__ __attribute__((aligned(16))) f32x4 SUM[128];

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int uidx = unit0 + unit_of_block((int)blockIdx.x, (int)gridDim.x, reverse);
    const SvdqUnit ud = units[uidx];
    const int p = ud.param;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    const int64_t r_begin = ud.row0;
    int64_t r_end = r_begin + ud.nrows;
    if (r_end > D) r_end = D;
    const int k = k_dev[p], r = r_dev[p], nl = r - k;
    const int t0 = wv * HT;

    HalfLoader<HT, MODE> ld;
#pragma unroll
    for (int i = 0; i < HT; ++i) ld.bp[i] = (gfloat *)ptrs[(size_t)p * NT + (t0 + i < NT ? t0 + i : NT - 1)];
    ld.gidx = GATHER ? (gint *)aux[p] : nullptr;
    ld.gbase = (MODE & 2) ? (gfloat *)aux2[p] : nullptr;
    ld.D = D;
    ld.r_end = r_end;
    ld.lane = lane;

    const int c = lane & 15, g = lane >> 4;
    // this wave's 16-column block of W = V Sigma^-1: B-operand registers
    const float *Wp = Wtab + (size_t)p * (NT * NT + 4);
    const float spike = Wp[NT * NT];
    const int nullcol = (int)Wp[NT * NT + 1];
    const int icol = 16 * wv + c;  // the U column this lane owns
    float w[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = 4 * s + g;
        w[s] = (t < NT && icol < NT) ? Wp[t * NT + icol] : 0.f;
    }
    out_t *const OUTh = OUT;
    out_t *const OUTl = OUT + SVDQ_BLK_ROWS * k;
    out_t *const DUMP = OUT + SVDQ_BLK_ROWS * NTP;
    const bool cvalid = icol < r;
    out_t *const colbase = !cvalid ? DUMP : (icol < k ? OUTh + icol : OUTl + (icol - k));
    const int colstride = !cvalid ? 0 : (icol < k ? k : nl);

    uint8_t *slab = basis + params[p].slab_off;
    uint8_t *gUh = slab;
    uint8_t *gUl = slab + svdq_align_up(D * (int64_t)k * ES, 256) + SVDQ_EXP_ULOW_SHIFT;
    float *gmean = (center && meanbuf) ? meanbuf + params[p].mean_off : nullptr;

    double caccd[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) caccd[i][e] = 0.0;

    // projection B operand ("task on slot, row on k") for the two 16-task blocks
    const float *xb_ptr[2];
    bool xb_ok[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int t = 16 * nb + c;
        xb_ok[nb] = t < NTP;
        xb_ptr[nb] = X + (xb_ok[nb] ? t : 0) * XS + 4 * g;
    }

    f32x4 v[HT];
    ld.first(v, r_begin);
    for (int64_t rb = r_begin; rb < r_end; rb += SVDQ_BLK_ROWS) {
        ld.apply(v);
        const f32x4 mean = center_store_half<HT, GATHER>(v, t0, NT, center, X, SUM, wv, lane);
        if (gmean && wv == 0) {
            if constexpr (GATHER) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (rb + 64 * e + lane < D) gmean[rb + 64 * e + lane] = mean[e];
            } else {
                const int64_t rr = rb + 4 * lane;
                if (rr + 3 < D) {
                    *reinterpret_cast<f32x4 *>(gmean + rr) = mean;
                } else {
                    if (rr < D) gmean[rr] = mean.x;
                    if (rr + 1 < D) gmean[rr + 1] = mean.y;
                    if (rr + 2 < D) gmean[rr + 2] = mean.z;
                }
            }
        }
        wg_sync();
        ld.next(v, rb);

        f32x4 cf[2];
        cf[0] = zero4();
        cf[1] = zero4();
UNROLL_N(SVDQ_UNROLL_BP2)
        for (int jj = 0; jj < 8; ++jj) {
            f32x4 err[2];
            f32x4 xb[2][2];
            float xa[2][KS];      // the pair's A operands, read before the first MFMA (see k_basis_project)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int s = 0; s < KS; ++s) xa[s2][s] = X[(4 * s + g) * XS + 16 * (2 * jj + s2) + c];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int j = 2 * jj + s2;
                f32x4 u = zero4();
#pragma unroll
                for (int s = 0; s < KS; ++s) u = mfma4(xa[s2][s], w[s], u);
                if (j == 0 && rb == 0 && g == 0 && icol == nullcol) u[0] += spike;  // completion column, row 0
                out_t *dst = colbase + (16 * j + 4 * g) * colstride;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (OUT16) {
                        const __half h = __float2half_rn(u[e]);
                        dst[e * colstride] = h;
                        err[s2][e] = __half2float(h) - u[e];
                    } else {
                        dst[e * colstride] = u[e];
                    }
                }
                if constexpr (OUT16) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        xb[nb][s2] = *reinterpret_cast<const f32x4 *>(xb_ptr[nb] + 16 * j);
                        if (!xb_ok[nb]) xb[nb][s2] = zero4();
                    }
                }
            }
            if constexpr (OUT16) {
                const bf16x8 ea = pack_bf16(err[0], err[1]);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) cf[nb] = mfma_bf16(ea, pack_bf16(xb[nb][0], xb[nb][1]), cf[nb]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) caccd[i][e] += (double)cf[i][e];
        wg_sync();

        const int rows_blk = (int)((D - rb < SVDQ_BLK_ROWS) ? (D - rb) : SVDQ_BLK_ROWS);
        if (k > 0) copy_out_wg(OUTh, gUh + rb * (int64_t)k * ES, rows_blk * k * ES, threadIdx.x);
        if (nl > 0) copy_out_wg(OUTl, gUl + rb * (int64_t)nl * ES, rows_blk * nl * ES, threadIdx.x);
        wg_sync();
    }

    // rounding-correction partials: cpart[unit][t*NT + i]; lane (c,g) holds D[m = U column 16 wv + 4g+e][n = task]
    double *dst = cpart + (size_t)uidx * NT * NT;
#pragma unroll
    for (int nbt = 0; nbt < 2; ++nbt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 16 * wv + 4 * g + e;
            const int t = 16 * nbt + c;
            if (i < NT && t < NT) dst[t * NT + i] = caccd[nbt][e];
        }
}

// ------------------------------------------------------------------------------------ 16 < N <= 24: one wave, 4x4 blocks
// At N = 17..20 the two-wave kernel above pays for 32 column slots where 20 are real: its second wave runs a full
// 16-column MFMA tile, and above all the full fp16 staging / rounding-error arithmetic, for 4 columns -- and that
// vector work, not memory, sets its time (4.4 TB/s at N = 20; 910 vector instructions per block and wave = 1 820 per
// 256 rows).  This variant has no idle column slots (1 500 vector instructions per 256 rows; ViT-L-14 x 20: pass 2
// 8.66 -> 8.0 ms, 4.35 -> 4.7 TB/s.  At N = 21..24, eight 4x4-block columns, it is slower than the two-wave kernel,
// which stays in charge there):
//   * one wavefront per workgroup on HALF blocks of 128 rows (lane = 2 rows per task: the 40 prefetch registers of
//     the two-wave kernel without its barriers and row-sum exchange);
//   * columns 0..15: v_mfma_f32_16x16x4_f32 per 16-row sub-tile as before;
//   * columns 16..: v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4x4 blocks, block b = rows 4b..4b+3 of a 64-row
//     group, A = the strip value of ONE task for those rows (lane l reads X[task][64 grp + l]), B = W[task][16 + 4q + j],
//     one instruction per task: 64 rows x 4 columns per chain, every result register a real output
//     (lane (b, j) register i = U[row 4b + i][col 16 + 4q + j]), 8 cycles per instruction;
//   * their rounding correction E^T Tc with v_mfma_f32_4x4x4_16b_bf16: B = the lane's own four errors (rows 4b..4b+3
//     of its column), A = four rows of four tasks from the strip; block partial sums meet in a shuffle at the end.
// Layouts of the two small-block MFMAs were determined on the device (tools/probe/mfma_layout.hip): lane l belongs to
// block l / 4, supplies A[i = l % 4] and B[j = l % 4], and holds D[i = register][j = l % 4].
// The fp32 dot products run over the tasks in the same order as in the 16x16x4 chain (one fused multiply-add each).
#define SVDQ_HB 128    // rows per half block
#ifndef SVDQ_XSH
#define SVDQ_XSH 144   // LDS row stride (floats) of one task's 128-row strip: 128 + 16 (the four task rows a 16x16x4
                       // operand read touches land in different banks: 8.08 -> 8.00 ms at N = 20 against 132)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x2 gf32x2;
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 pack_bf16x4(const f32x4 &x) {
    union { __bf16 h[4]; s16x4 v; } u;
#pragma unroll
    for (int e = 0; e < 4; ++e) u.h[e] = (__bf16)x[e];
    return u.v;
}

#ifndef SVDQ_Q_WAVES
#define SVDQ_Q_WAVES 1
#endif
// FULL: the plan has exactly NTP tasks (N = 20, N = 24).  With N a run-time value every "task t is real" test is a
// lane mask the compiler keeps in a scalar-register pair across the loop -- twenty of them, which is what pushed the
// twenty task pointers out of the scalar registers (a v_readlane pair + a 64-bit vector add per load) and put a
// v_cndmask on every value of the centring (round 3: ~200 of ~1 000 wave instructions per half block).
template <int NTP, bool OUT16, int MODE, bool FULL>
__global__ __launch_bounds__(64, SVDQ_Q_WAVES) void k_basis_project_q(
    const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
    const float *const *__restrict__ ptrs, const int64_t *__restrict__ rows_dev, int NT_arg, int center,
    const float *__restrict__ Wtab, const int32_t *__restrict__ k_dev, const int32_t *__restrict__ r_dev,
    uint8_t *__restrict__ basis, float *__restrict__ meanbuf, double *__restrict__ cpart, int unit0, int reverse,
    const void *const *__restrict__ aux, const void *const *__restrict__ aux2) {
    static_assert(NTP == 20 || NTP == 24, "the 4x4-block variant covers 16 < N <= 24");
    const int NT = FULL ? NTP : NT_arg;
    constexpr bool GATHER = (MODE & 1) != 0, SUB = (MODE & 2) != 0;
    using out_t = typename OutT<OUT16>::type;
    constexpr int KS = NTP / 4;    // k-steps of the 16x16x4 chain = task groups of the 4x4x4 correction
    constexpr int N1 = NTP - 16;   // columns handled by the 4x4 blocks
    constexpr int G1 = N1 / 4;     // groups of four such columns
    constexpr int ES = OUT16 ? 2 : 4;
    constexpr int HB = SVDQ_HB, XH = SVDQ_XSH;
    __shared__ __attribute__((aligned(16))) float X[(NTP + 1) * XH];   // + one row of zeros (task slots past NTP)
    __shared__ __attribute__((aligned(16))) out_t OUT[HB * NTP + 16];
    __shared__ float W1[NTP * N1];   // W[task][16 + q]: B operands of the 4x4x1 chain

    const int lane = threadIdx.x & 63;
    const int uidx = unit0 + unit_of_block((int)blockIdx.x, (int)gridDim.x, reverse);
    const SvdqUnit ud = units[uidx];
    const int p = ud.param;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    const int64_t r_begin = ud.row0;
    int64_t r_end = r_begin + ud.nrows;
    if (r_end > D) r_end = D;
    const int k = k_dev[p], r = r_dev[p], nl = r - k;

    // Twenty 64-bit task pointers held across the loop would push the kernel over its scalar-register budget (the
    // compiler then parks them in vector-register lanes and pays a v_readlane pair per use): they are re-read from the
    // table for every half block instead (scalar loads, cached), and lanes address with a 32-bit offset from the
    // unit's first row.
#ifndef SVDQ_Q_PTR_RELOAD
#define SVDQ_Q_PTR_RELOAD 0   // measured: the scalar loads in front of every half block cost more than the readlanes
#endif
    const float *const *ptab = ptrs + (size_t)p * NT;
    gfloat *bp[NTP];
#pragma unroll
    for (int t = 0; t < NTP; ++t) bp[t] = (gfloat *)ptab[t < NT ? t : NT - 1];
    gint *gidx = GATHER ? (gint *)aux[p] : nullptr;
    gfloat *gbase = SUB ? (gfloat *)aux2[p] : nullptr;

    const int c = lane & 15, g = lane >> 4;     // 16x16 tile coordinates
    const int b4 = lane >> 2, j4 = lane & 3;    // 4x4 block, column inside the block
    const float *Wp = Wtab + (size_t)p * (NT * NT + 4);
    const float spike = Wp[NT * NT];
    const int nullcol = (int)Wp[NT * NT + 1];
    float w0[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = 4 * s + g;
        w0[s] = (t < NT && c < NT) ? Wp[t * NT + c] : 0.f;
    }
    for (int e = lane; e < NTP * N1; e += 64) {
        const int t = e / N1, q = e % N1;
        W1[e] = (t < NT && 16 + q < NT) ? Wp[t * NT + 16 + q] : 0.f;
    }
    for (int e = lane; e < XH; e += 64) X[NTP * XH + e] = 0.f;
    // projection B operand ("task on slot, rows on k") of the two 16-task blocks: task 16 nb + c, or the zero row
    const float *xbrow[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) xbrow[nb] = X + (16 * nb + c < NTP ? 16 * nb + c : NTP) * XH + 4 * g;
    // staging: every lane owns one column of the 16-column tile and one column of each 4-column group
    out_t *const OUTh = OUT;
    out_t *const OUTl = OUT + HB * k;
    out_t *const DUMP = OUT + HB * NTP;
    auto col_base = [&](int i) -> out_t * { return i >= r ? DUMP : (i < k ? OUTh + i : OUTl + (i - k)); };
    auto col_stride = [&](int i) -> int { return i >= r ? 0 : (i < k ? k : nl); };
    out_t *const cb0 = col_base(c);
    const int cs0 = col_stride(c);
    out_t *cb1[G1];
    int cs1[G1];
#pragma unroll
    for (int q = 0; q < G1; ++q) {
        cb1[q] = col_base(16 + 4 * q + j4);
        cs1[q] = col_stride(16 + 4 * q + j4);
    }
    uint8_t *slab = basis + params[p].slab_off;
    uint8_t *gUh = slab;
    uint8_t *gUl = slab + svdq_align_up(D * (int64_t)k * ES, 256) + SVDQ_EXP_ULOW_SHIFT;
    float *gmean = (center && meanbuf) ? meanbuf + params[p].mean_off : nullptr;

    double caccd[2][4];          // columns 0..15 x tasks (two 16-task blocks): fp32 inside a half block, fp64 across
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) caccd[i][e] = 0.0;
    f32x4 cacc1[G1][KS];         // columns 16..: [task 4 tg + i][col 16 + 4q + j4], per 4-row block, over the unit
#pragma unroll
    for (int q = 0; q < G1; ++q)
#pragma unroll
        for (int tg = 0; tg < KS; ++tg) cacc1[q][tg] = zero4();

    // loads of one half block: lane owns rows 2 lane, 2 lane + 1 (gather mode: rows lane, 64 + lane)
    f32x2 v[NTP];
    f32x2 vb = {0.f, 0.f};
    int ixa = -1, ixb = -1;      // gather mode: source positions of the NEXT half block's two rows
    auto load_idx2 = [&](int64_t rb) {
        const int64_t ra = rb + lane, rc = rb + 64 + lane;
        ixa = ra < D ? gidx[ra] : -1;
        ixb = rc < D ? gidx[rc] : -1;
    };
    auto load_half = [&](int64_t rb) {
        // (the index depends on rb only formally: it keeps the pointer loads inside the loop)
        const int hop = (int)(rb >> 62);
        if constexpr (GATHER) {
#pragma unroll
            for (int t = 0; t < NTP; ++t) {
                gfloat *bt = SVDQ_Q_PTR_RELOAD ? (gfloat *)ptab[(t < NT ? t : NT - 1) + hop] : bp[t];
                f32x2 o = {0.f, 0.f};
                if (ixa >= 0) o.x = bt[ixa];
                if (ixb >= 0) o.y = bt[ixb];
                v[t] = o;
            }
            if constexpr (SUB) {
                vb = f32x2{0.f, 0.f};
                if (ixa >= 0) vb.x = gbase[ixa];
                if (ixb >= 0) vb.y = gbase[ixb];
            }
        } else {
            const int64_t rr = rb + 2 * lane;
            // BYTE offset past the unit's first row, 32 bits: scalar base + 32-bit vector offset is an addressing mode
            // of global_load (a 64-bit element offset makes every load a 64-bit vector add first)
            const uint32_t boff = ((uint32_t)(rb - r_begin) + 2u * (uint32_t)lane) * 4u;
            if (rb + HB <= D) {
#pragma unroll
                for (int t = 0; t < NTP; ++t) {
                    gfloat *bt = (SVDQ_Q_PTR_RELOAD ? (gfloat *)ptab[(t < NT ? t : NT - 1) + hop] : bp[t]) + r_begin;
                    v[t] = *reinterpret_cast<gf32x2 *>(reinterpret_cast<const __attribute__((address_space(1))) char *>(bt) + boff);
                }
                if constexpr (SUB) vb = *reinterpret_cast<gf32x2 *>(gbase + rr);
            } else {
#pragma unroll
                for (int t = 0; t < NTP; ++t) {
                    gfloat *bt = SVDQ_Q_PTR_RELOAD ? (gfloat *)ptab[(t < NT ? t : NT - 1) + hop] : bp[t];
                    f32x2 o = {0.f, 0.f};
                    if (rr < D) o.x = bt[rr];
                    if (rr + 1 < D) o.y = bt[rr + 1];
                    v[t] = o;
                }
                if constexpr (SUB) {
                    vb = f32x2{0.f, 0.f};
                    if (rr < D) vb.x = gbase[rr];
                    if (rr + 1 < D) vb.y = gbase[rr + 1];
                }
            }
        }
    };
    if (r_begin < r_end) {
        if constexpr (GATHER) load_idx2(r_begin);
        load_half(r_begin);
        if constexpr (GATHER) {
            if (r_begin + HB < r_end) load_idx2(r_begin + HB);
        }
    }
    wave_sync();   // W1 is in LDS
    float w1r[G1][NTP];      // B operands of the 4x4x1 chains: this lane's column of every group, constant over the unit
#pragma unroll
    for (int q = 0; q < G1; ++q)
#pragma unroll
        for (int t = 0; t < NTP; ++t) w1r[q][t] = W1[t * N1 + 4 * q + j4];

    for (int64_t rb = r_begin; rb < r_end; rb += HB) {
        // ---- centre (same association as pass 1 at N > 16: two halves of the tasks), park the strip, write the mean
        if constexpr (SUB) {
#pragma unroll
            for (int t = 0; t < NTP; ++t) v[t] = v[t] - vb;
        }
        f32x2 h0 = {0.f, 0.f}, h1 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NTP / 2; ++i) {
            h0 += (i < NT) ? v[i] : f32x2{0.f, 0.f};
            h1 += (NTP / 2 + i < NT) ? v[NTP / 2 + i] : f32x2{0.f, 0.f};
        }
        const f32x2 tot = h0 + h1;
        f32x2 mean = {0.f, 0.f};
        if (center) {
            const float n = (float)NT;
            mean.x = tot.x / n;
            mean.y = tot.y / n;
        }
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            const f32x2 xc = (t < NT) ? (v[t] - mean) : f32x2{0.f, 0.f};
            if constexpr (GATHER) {
                X[t * XH + lane] = xc.x;
                X[t * XH + 64 + lane] = xc.y;
            } else {
                *reinterpret_cast<f32x2 *>(X + t * XH + 2 * lane) = xc;
            }
        }
        if (gmean) {
            if constexpr (GATHER) {
                if (rb + lane < D) gmean[rb + lane] = mean.x;
                if (rb + 64 + lane < D) gmean[rb + 64 + lane] = mean.y;
            } else {
                const int64_t rr = rb + 2 * lane;
                if (rr + 1 < D) *reinterpret_cast<f32x2 *>(gmean + rr) = mean;
                else if (rr < D) gmean[rr] = mean.x;
            }
        }
        wave_sync();
        if (rb + HB < r_end) {   // next half block in flight while this one is computed
            load_half(rb + HB);
            if constexpr (GATHER) {
                if (rb + 2 * HB < r_end) load_idx2(rb + 2 * HB);
            }
        }

        // ---- columns 0..15: eight 16-row sub-tiles
        f32x4 cf[2];
        cf[0] = zero4();
        cf[1] = zero4();
#pragma unroll
        for (int jj = 0; jj < HB / 32; ++jj) {
            f32x4 err[2];
            f32x4 xb[2][2];
            // the A operands of BOTH sub-tiles of the pair are read before the first MFMA: one LDS round trip per pair
            // instead of one per MFMA (the chain ds_read -> wait -> mfma, ten times, left the wave idle for most of it)
            float xa[2][KS];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int s = 0; s < KS; ++s) xa[s2][s] = X[(4 * s + g) * XH + 16 * (2 * jj + s2) + c];
            if constexpr (OUT16) {      // ... and the projection's B operands with them
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        xb[nb][s2] = *reinterpret_cast<const f32x4 *>(xbrow[nb] + 16 * (2 * jj + s2));
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int j = 2 * jj + s2;
                f32x4 u = zero4();
#pragma unroll
                for (int s = 0; s < KS; ++s) u = mfma4(xa[s2][s], w0[s], u);
                if (j == 0 && rb == 0 && g == 0 && c == nullcol) u[0] += spike;   // completion column, row 0
                out_t *dst = cb0 + (16 * j + 4 * g) * cs0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (OUT16) {
                        const __half hh = __float2half_rn(u[e]);
                        dst[e * cs0] = hh;
                        err[s2][e] = __half2float(hh) - u[e];
                    } else {
                        dst[e * cs0] = u[e];
                    }
                }
            }
            if constexpr (OUT16) {
                const bf16x8 ea = pack_bf16(err[0], err[1]);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) cf[nb] = mfma_bf16(ea, pack_bf16(xb[nb][0], xb[nb][1]), cf[nb]);
            }
        }
        // ---- columns 16..: two 64-row groups of sixteen 4x4 blocks
#pragma unroll
        for (int grp = 0; grp < HB / 64; ++grp) {
            float xr[NTP];      // this lane's row of every task: read once per group, before the MFMA chains
#pragma unroll
            for (int t = 0; t < NTP; ++t) xr[t] = X[t * XH + 64 * grp + lane];
            f32x4 xq[KS];       // A operands of the 4x4x4 correction: four rows of four tasks per task group
            if constexpr (OUT16) {
#pragma unroll
                for (int tg = 0; tg < KS; ++tg)
                    xq[tg] = *reinterpret_cast<const f32x4 *>(X + (4 * tg + j4) * XH + 64 * grp + 4 * b4);
            }
#pragma unroll
            for (int q = 0; q < G1; ++q) {
                f32x4 u = zero4();
#pragma unroll
                for (int t = 0; t < NTP; ++t) u = mfma_4x4x1(xr[t], w1r[q][t], u);
                if (grp == 0 && rb == 0 && b4 == 0 && 16 + 4 * q + j4 == nullcol) u[0] += spike;
                out_t *dst = cb1[q] + (64 * grp + 4 * b4) * cs1[q];
                f32x4 er;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (OUT16) {
                        const __half hh = __float2half_rn(u[i]);
                        dst[i * cs1[q]] = hh;
                        er[i] = __half2float(hh) - u[i];
                    } else {
                        dst[i * cs1[q]] = u[i];
                    }
                }
                if constexpr (OUT16) {
                    const s16x4 eb = pack_bf16x4(er);
#pragma unroll
                    for (int tg = 0; tg < KS; ++tg)
                        cacc1[q][tg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(pack_bf16x4(xq[tg]), eb, cacc1[q][tg], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) caccd[i][e] += (double)cf[i][e];
        wave_sync();
        const int rows_blk = (int)((D - rb < HB) ? (D - rb) : HB);
        if (k > 0) copy_out(OUTh, gUh + rb * (int64_t)k * ES, rows_blk * k * ES, lane);
        if (nl > 0) copy_out(OUTl, gUl + rb * (int64_t)nl * ES, rows_blk * nl * ES, lane);
        wave_sync();
    }

    // rounding-correction partials: cpart[unit][t * NT + i] = sum over rows of E[row][i] * Xc[t][row]
    double *dst = cpart + (size_t)uidx * NT * NT;
#pragma unroll
    for (int nbt = 0; nbt < 2; ++nbt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // lane (c, g) holds D[m = column 4g + e][n = task 16 nbt + c]
            const int i = 4 * g + e, t = 16 * nbt + c;
            if (i < NT && t < NT) dst[t * NT + i] = caccd[nbt][e];
        }
#pragma unroll
    for (int q = 0; q < G1; ++q)
#pragma unroll
        for (int tg = 0; tg < KS; ++tg)
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // lane (b, j) register i holds [task 4 tg + i][col 16 + 4q + j] of block b
                double x = (double)cacc1[q][tg][i];
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) x += __shfl_xor(x, off);
                const int t = 4 * tg + i, col = 16 + 4 * q + j4;
                if (b4 == 0 && t < NT && col < NT) dst[t * NT + col] = x;
            }
}

// ------------------------------------------------------------------------------------ launchers
template <int NTP, bool F16>
static int launch_bp_mode(const svdq_plan *pl, const float *const *pp, const int64_t *rows_dev, const float *W,
                          const int32_t *k_dev, const int32_t *r_dev, uint8_t *basis, float *mean, double *cpart,
                          int unit0, int nunits, int reverse, const void *idx, const void *base,
                          const int64_t *ustart, hipStream_t st) {
    auto ai = (const void *const *)idx, ab = (const void *const *)base;
    if (ustart) {
        if constexpr (NTP <= 16) {
#define SVDQ_LAUNCH_BPW(M, FULL_)                                                                                     \
    hipLaunchKernelGGL((k_basis_project<NTP, F16, M, FULL_>), dim3(nunits), dim3(64), 0, st, pl->d_params, pl->d_units, \
                       pp, rows_dev, pl->n_tasks, pl->cfg.center, W, k_dev, r_dev, basis, mean, cpart, unit0, reverse, \
                       ai, ab, ustart)
            if (pl->n_tasks == NTP) {
                if (base) SVDQ_LAUNCH_BPW(6, true); else SVDQ_LAUNCH_BPW(4, true);
            } else {
                if (base) SVDQ_LAUNCH_BPW(6, false); else SVDQ_LAUNCH_BPW(4, false);
            }
#undef SVDQ_LAUNCH_BPW
            return SVDQ_OK;
        }
        if (base) {
            svdq_set_error("the mask walk straight from checkpoints covers N <= 16 tasks (got %d): use the index lists "
                           "(svdq_compress_gather_from_base)", pl->n_tasks);
            return SVDQ_EUNSUPPORTED;
        }
        return svdq_launch_basis_project_walk32(pl, pp, rows_dev, W, k_dev, r_dev, basis, mean, cpart, unit0, nunits, reverse,
                                                ai, ustart, st);   // the one-wave kernel: svdq_project_walk.hip
    }
#define SVDQ_LAUNCH_BP(M)                                                                                             \
    do {                                                                                                              \
        if constexpr (NTP == 20) {   /* N = 21..24: measured slower than the two-wave kernel (11.9 against 10.0 ms) */ \
            if (!(pl->cfg.reserved & 8)) {   /* bit 3: the two-wave kernel instead (A/B) */                           \
                if (pl->n_tasks == NTP)                                                                               \
                    hipLaunchKernelGGL((k_basis_project_q<NTP, F16, M, true>), dim3(nunits), dim3(64), 0, st,         \
                                       pl->d_params, pl->d_units, pp, rows_dev, pl->n_tasks, pl->cfg.center, W, k_dev, \
                                       r_dev, basis, mean, cpart, unit0, reverse, ai, ab);                            \
                else                                                                                                  \
                    hipLaunchKernelGGL((k_basis_project_q<NTP, F16, M, false>), dim3(nunits), dim3(64), 0, st,        \
                                       pl->d_params, pl->d_units, pp, rows_dev, pl->n_tasks, pl->cfg.center, W, k_dev, \
                                       r_dev, basis, mean, cpart, unit0, reverse, ai, ab);                            \
                break;                                                                                                \
            }                                                                                                         \
        }                                                                                                             \
        if constexpr (NTP > 16)                                                                                       \
            hipLaunchKernelGGL((k_basis_project2<NTP, F16, M>), dim3(nunits), dim3(128), 0, st, pl->d_params,         \
                               pl->d_units, pp, rows_dev, pl->n_tasks, pl->cfg.center, W, k_dev, r_dev, basis, mean,  \
                               cpart, unit0, reverse, ai, ab);                                                        \
        else if (pl->n_tasks == NTP)                                                                                  \
            hipLaunchKernelGGL((k_basis_project<NTP, F16, M, true>), dim3(nunits), dim3(64), 0, st, pl->d_params,     \
                               pl->d_units, pp, rows_dev, pl->n_tasks, pl->cfg.center, W, k_dev, r_dev, basis, mean,  \
                               cpart, unit0, reverse, ai, ab, (const int64_t *)nullptr);                              \
        else                                                                                                          \
            hipLaunchKernelGGL((k_basis_project<NTP, F16, M, false>), dim3(nunits), dim3(64), 0, st, pl->d_params,    \
                               pl->d_units, pp, rows_dev, pl->n_tasks, pl->cfg.center, W, k_dev, r_dev, basis, mean,  \
                               cpart, unit0, reverse, ai, ab, (const int64_t *)nullptr);                              \
    } while (0)
    switch ((idx ? 1 : 0) | (base ? 2 : 0)) {
        case 0: SVDQ_LAUNCH_BP(0); break;
        case 1: SVDQ_LAUNCH_BP(1); break;
        case 2: SVDQ_LAUNCH_BP(2); break;
        default: SVDQ_LAUNCH_BP(3); break;
    }
#undef SVDQ_LAUNCH_BP
    return SVDQ_OK;
}

template <int NTP>
static int launch_bp_t(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, const float *W,
                       const int32_t *k_dev, const int32_t *r_dev, uint8_t *basis, float *mean, double *cpart,
                       int unit0, int nunits, int reverse, const void *idx, const void *base, const int64_t *ustart,
                       hipStream_t st) {
    auto pp = reinterpret_cast<const float *const *>(ptrs);
    int rc;
    if (pl->cfg.fp16)
        rc = launch_bp_mode<NTP, true>(pl, pp, rows_dev, W, k_dev, r_dev, basis, mean, cpart, unit0, nunits, reverse, idx, base, ustart, st);
    else
        rc = launch_bp_mode<NTP, false>(pl, pp, rows_dev, W, k_dev, r_dev, basis, mean, cpart, unit0, nunits, reverse, idx, base, ustart, st);
    if (rc != SVDQ_OK) return rc;
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

int svdq_launch_basis_project(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, const float *W,
                              const int32_t *k_dev, const int32_t *r_dev, uint8_t *basis, float *mean,
                              double *cpart, int unit0, int nunits, int reverse, const void *idx, const void *base,
                              hipStream_t st, const int64_t *ustart) {
#define SVDQ_BP_CASE(n) \
    case n: return launch_bp_t<n>(pl, ptrs, rows_dev, W, k_dev, r_dev, basis, mean, cpart, unit0, nunits, reverse, idx, base, ustart, st)
    switch (pl->ntp) {
        SVDQ_BP_CASE(4); SVDQ_BP_CASE(8); SVDQ_BP_CASE(12); SVDQ_BP_CASE(16);
        SVDQ_BP_CASE(20); SVDQ_BP_CASE(24); SVDQ_BP_CASE(28); SVDQ_BP_CASE(32);
    }
#undef SVDQ_BP_CASE
    svdq_set_error("unsupported padded task count %d", pl->ntp);
    return SVDQ_EUNSUPPORTED;
}


#ifdef SVDQ_UNIT_STAMPS
extern "C" int svdq_debug_stamps_project(unsigned long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(svdq_stamps_project), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

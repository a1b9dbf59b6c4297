// svdq_project_unit.h -- the one-wave form of pass 2 (bp_unit / k_basis_project): what svdq_project.hip launches for
// N <= 16 in every mode, and svdq_project_walk.hip for the mask walk at 16 < N <= 32.  Layout notes: svdq_stream.h.
#pragma once
#include "svdq_stream.h"

// ------------------------------------------------------------------------------------ pass 2
__device__ __forceinline__ void copy_out(const void *lds_src, uint8_t *gdst, int nbytes, int lane) {
    const int nvec = nbytes >> 4;
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(lds_src);
    f32x4 *d4 = reinterpret_cast<f32x4 *>(gdst);
    for (int i = lane; i < nvec; i += 64) {
#if SVDQ_NT_STORES
        __builtin_nontemporal_store(s4[i], &d4[i]);
#else
        d4[i] = s4[i];
#endif
    }
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(lds_src);
    for (int b = (nvec << 4) + 2 * lane; b < nbytes; b += 128)
        *reinterpret_cast<uint16_t *>(gdst + b) = *reinterpret_cast<const uint16_t *>(sb + b);
}

template <bool OUT16> struct OutT;
template <> struct OutT<true> { using type = __half; };
template <> struct OutT<false> { using type = float; };

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 pack_bf16(const f32x4 &lo, const f32x4 &hi) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = (__bf16)lo[e];
        o[4 + e] = (__bf16)hi[e];
    }
    return o;
}

// Projection onto the ROUNDED basis without eight more f32 MFMAs per 32 rows:
//   c = fp16(U)^T Xc = U^T Xc + E^T Xc,   E = fp16(U) - U.
// U^T Xc is known in closed form from the eigen-solve (sigma_i V[t][i], k_eig writes it), so only the
// rounding correction E^T Xc -- 2^-12 of the signal -- is accumulated here, with bf16 operands on
// v_mfma_f32_16x16x32_bf16 (K = 32 rows per instruction): bf16's 2^-9 relative operand error on a
// 2^-12 term is 2^-21 of c, below the fp32 accumulation noise of the direct product.
// One work unit of pass 2 by ONE wavefront.  X: NTP*XS floats, OUT: SVDQ_BLK_ROWS*NTP + 16 elements of
// wave-private LDS.
// FULL: the plan has exactly NTP tasks -- every "task t is real" test folds away (see k_basis_project_q)
template <int NTP, bool OUT16, int MODE = 0, bool FULL = false>
__device__ __forceinline__ void bp_unit(
    float *X, typename OutT<OUT16>::type *OUT, int uidx, const SvdqParam *__restrict__ params,
    const SvdqUnit *__restrict__ units, const float *const *__restrict__ ptrs,
    const int64_t *__restrict__ rows_dev, int NT_arg, int center, const float *__restrict__ Wtab,
    const int32_t *__restrict__ k_dev, const int32_t *__restrict__ r_dev, uint8_t *__restrict__ basis,
    float *__restrict__ meanbuf, double *__restrict__ cpart, const void *const *__restrict__ aux = nullptr,
    const void *const *__restrict__ aux2 = nullptr, const int64_t *__restrict__ ustart = nullptr) {
    constexpr bool GATHER = (MODE & 1) != 0, SUB = (MODE & 2) != 0, WALK = (MODE & 4) != 0;
    static_assert(!(GATHER && WALK), "index lists and the mask walk are alternatives");
    static_assert(!(MODE != 0 && SVDQ_PREFETCH2), "gather / minus-base support the one-block-ahead pipeline only");
    const int NT = FULL ? NTP : NT_arg;
    constexpr int PACK = (NTP <= 8) ? 2 : 1;
    constexpr int NB = (NTP + 15) / 16;
    constexpr int KS = NTP / 4;
    constexpr int NCB = NB * NB;
    constexpr int ES = OUT16 ? 2 : 4;
    constexpr int TROWS = (PACK == 2) ? 32 : 16;  // rows per MFMA sub-tile
    constexpr int NPAIR = SVDQ_BLK_ROWS / (2 * TROWS);
    using out_t = typename OutT<OUT16>::type;

    const int lane = threadIdx.x & 63;
    const SvdqUnit ud = units[uidx];
    const int p = ud.param;
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    const int64_t r_begin = ud.row0;
    int64_t r_end = r_begin + ud.nrows;
    if (r_end > D) r_end = D;

    const int k = k_dev[p];
    const int r = r_dev[p];
    const int nl = r - k;

    gfloat *bp[NTP];
#pragma unroll
    for (int t = 0; t < NTP; ++t) bp[t] = (gfloat *)ptrs[(size_t)p * NT + (t < NT ? t : NT - 1)];

    const int c = lane & 15, g = lane >> 4;

    // W = V Sigma^-1 (columns >= r and unresolved directions are zero), B-operand registers.
    const float *Wp = Wtab + (size_t)p * (NT * NT + 4);
    const float spike = Wp[NT * NT];
    const int nullcol = (int)Wp[NT * NT + 1];
    float w[KS][NB];
    float whi[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = 4 * s + g;
        if constexpr (PACK == 2) {
            const int i = c & 7;
            const float val = (t < NT && i < NT) ? Wp[t * NT + i] : 0.f;
            w[s][0] = (c < 8) ? val : 0.f;
            whi[s] = (c >= 8) ? val : 0.f;
        } else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int i = 16 * nb + c;
                w[s][nb] = (t < NT && i < NT) ? Wp[t * NT + i] : 0.f;
            }
            whi[s] = 0.f;
        }
    }

    // Branch-free staging of the fp16 tile: every lane owns one U column per 16-slot block and writes
    // it into the U_high or the U_low image; lanes without a column write to the dump slot.
    out_t *const OUTh = OUT;
    out_t *const OUTl = OUT + SVDQ_BLK_ROWS * k;
    out_t *const DUMP = OUT + SVDQ_BLK_ROWS * NTP;
    out_t *colbase[NB];
    int colstride[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int i = (PACK == 2) ? (c & 7) : 16 * nb + c;
        const bool valid = (PACK == 2) ? ((c & 7) < NTP && i < r) : (i < r);
        colbase[nb] = !valid ? DUMP : (i < k ? OUTh + i : OUTl + (i - k));
        colstride[nb] = !valid ? 0 : (i < k ? k : nl);
    }

    uint8_t *slab = basis + params[p].slab_off;
    uint8_t *gUh = slab;
    uint8_t *gUl = slab + svdq_align_up(D * (int64_t)k * ES, 256) + SVDQ_EXP_ULOW_SHIFT;
    float *gmean = (center && meanbuf) ? meanbuf + params[p].mean_off : nullptr;

    double caccd[NCB][4];
#pragma unroll
    for (int i = 0; i < NCB; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) caccd[i][e] = 0.0;

    // The loads of the next block (or next two, SVDQ_PREFETCH2) are in flight while a block is computed.
    constexpr int AHEAD = SVDQ_PREFETCH2 ? 2 : 1;
    f32x4 v0[NTP];
    gint *gidx = nullptr;
    i32x4 ixn = {-1, -1, -1, -1};  // indices of the block after the one whose data is in flight
    gfloat *gbase = nullptr;
    f32x4 vb = zero4();  // base rows of the block whose fine-tuned rows sit in v
    if constexpr (SUB) gbase = (gfloat *)aux2[p];
    if constexpr (WALK) {
        // loads are issued by the walk loop below
    } else if constexpr (GATHER) {
        gidx = (gint *)aux[p];
        if (r_begin < r_end) {
            const i32x4 ix0 = load_idx(gidx, r_begin, D, lane);
            load_block_gather<NTP>(v0, bp, ix0, r_begin + SVDQ_BLK_ROWS <= D);
            if constexpr (SUB) vb = load_base_gather(gbase, ix0, r_begin + SVDQ_BLK_ROWS <= D);
            if (r_begin + SVDQ_BLK_ROWS < r_end) ixn = load_idx(gidx, r_begin + SVDQ_BLK_ROWS, D, lane);
        }
    } else {
        if (r_begin < r_end) load_block<NTP>(v0, bp, r_begin, D, lane);
        if constexpr (SUB) {
            if (r_begin < r_end) vb = load_base(gbase, r_begin, D, lane);
        }
    }
#if SVDQ_PREFETCH2
    f32x4 v1[NTP];
    if (r_begin + SVDQ_BLK_ROWS < r_end) load_block<NTP>(v1, bp, r_begin + SVDQ_BLK_ROWS, D, lane);
#endif

    // one block out of the strip: U tiles, fp16 staging, rounding-correction MFMAs, the two row-major stores
    auto compute = [&](int64_t rb) {
        f32x4 cf[NCB];
#pragma unroll
        for (int i = 0; i < NCB; ++i) cf[i] = zero4();

        // projection B operand ("task on slot, row on k"): per 16-slot block, which strip and rows
        const float *xb_ptr[NB];
        bool xb_ok[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int t = (PACK == 2) ? (c & 7) : 16 * nb + c;
            xb_ok[nb] = t < NTP;
            xb_ptr[nb] = X + (xb_ok[nb] ? t : 0) * XS + ((PACK == 2) ? 16 * (c >> 3) : 0) + 4 * g;
        }

UNROLL_N(SVDQ_UNROLL_BP)
        for (int jj = 0; jj < NPAIR; ++jj) {
            f32x4 err[NB][2];  // E = fp16(U) - U for the two sub-tiles of this pair
            f32x4 xb[NB][2];
            // every LDS operand of the pair is read before its first MFMA (one round trip per pair; read one by one in
            // front of each MFMA the chain ds_read -> wait -> mfma leaves a wave idle for most of the sub-tile, which the
            // two or three waves per SIMD of N > 8 cannot cover: N = 20 pass 2 7.99 -> 7.7 ms)
            float xa[2][KS][PACK];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int h = 0; h < PACK; ++h) xa[s2][s][h] = X[(4 * s + g) * XS + TROWS * (2 * jj + s2) + 16 * h + c];
            if constexpr (OUT16) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        xb[nb][s2] = *reinterpret_cast<const f32x4 *>(xb_ptr[nb] + TROWS * (2 * jj + s2));
                        if (!xb_ok[nb]) xb[nb][s2] = zero4();
                    }
            }
            // (alternating the two sub-tiles' accumulation chains -- a dependent MFMA then issues 64 cycles after its
            // predecessor instead of right behind it -- measured nothing: 7.31 / 7.36 ms at N = 20, 5.39 / 5.53 at N = 16)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int j = 2 * jj + s2;
                f32x4 u[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) u[nb] = zero4();
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if constexpr (PACK == 2) {
                        u[0] = mfma4(xa[s2][s][0], w[s][0], u[0]);
                        u[0] = mfma4(xa[s2][s][1], whi[s], u[0]);
                    } else {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) u[nb] = mfma4(xa[s2][s][0], w[s][nb], u[nb]);
                    }
                }
                // row 0 of the completion column (see k_eig)
                if (j == 0 && rb == 0 && g == 0) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int i = (PACK == 2) ? c : 16 * nb + c;  // PACK: row 0 lives on slots 0-7 only
                        if (i == nullcol) u[nb][0] += spike;
                    }
                }
                const int row0 = TROWS * j + ((PACK == 2) ? 16 * (c >> 3) : 0) + 4 * g;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    out_t *dst = colbase[nb] + row0 * colstride[nb];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (OUT16) {
                            const __half h = __float2half_rn(u[nb][e]);
                            dst[e * colstride[nb]] = h;
                            err[nb][s2][e] = __half2float(h) - u[nb][e];
                        } else {
                            dst[e * colstride[nb]] = u[nb][e];
                        }
                    }
                }
            }
            if constexpr (OUT16) {
                bf16x8 ea[NB], xv[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    ea[nb] = pack_bf16(err[nb][0], err[nb][1]);
                    xv[nb] = pack_bf16(xb[nb][0], xb[nb][1]);
                }
#pragma unroll
                for (int nbi = 0; nbi < NB; ++nbi)
#pragma unroll
                    for (int nbt = 0; nbt < NB; ++nbt)
                        cf[nbi * NB + nbt] = mfma_bf16(ea[nbi], xv[nbt], cf[nbi * NB + nbt]);
            }
        }
#pragma unroll
        for (int i = 0; i < NCB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) caccd[i][e] += (double)cf[i][e];
        wave_sync();

        // stream the two row-major output tiles of this block out of LDS, 16 B per lane
        const int rows_blk = (int)((D - rb < SVDQ_BLK_ROWS) ? (D - rb) : SVDQ_BLK_ROWS);
#ifndef SVDQ_ABLATE_STORES
        if (k > 0) copy_out(OUTh, gUh + rb * (int64_t)k * ES, rows_blk * k * ES, lane);
        if (nl > 0) copy_out(OUTl, gUl + rb * (int64_t)nl * ES, rows_blk * nl * ES, lane);
#else
        if (rows_blk < 0) copy_out(OUTh, gUh, 16, lane);  // diagnostic build: keep OUT live, skip the stores
#endif
        wave_sync();
    };

    if constexpr (WALK) {
        // walk the source rows from this unit's first selected element (see "walk mode" above)
        gbyte *gmask = (gbyte *)aux[p];
        const int64_t Dsrc = params[p].rows;
        const int64_t us = ustart[uidx];
        const int inv = (us & SVDQ_WALK_INV) ? 1 : 0;
        int64_t src = us & (SVDQ_WALK_INV - 1);
        int64_t src_end = Dsrc;      // where the next unit's rows begin
        if (uidx + 1 < params[p].unit_begin + params[p].unit_count) src_end = ustart[uidx + 1] & (SVDQ_WALK_INV - 1);
        if (src_end > Dsrc) src_end = Dsrc;
        const int need = (r_begin < r_end) ? (int)(r_end - r_begin) : 0;
        int produced = 0, fill = 0;
        int64_t rb = r_begin;
        // (A second register set taking the NEXT chunk's loads while this one is ingested and its block computed was
        // measured: 0.850-0.868 ms against 0.860 ms for pass 2 of ViT-B-16 x 8 on the same box, with 11 spilled dwords
        // to stay at three waves per SIMD -- the twelve waves of a CU already keep the loads flowing.  Not kept.)
        unsigned mk[4];
        bool have = need > 0 && src < src_end;
        if (have) walk_load<NTP, SUB>(v0, vb, mk, bp, gbase, gmask, src, src_end, lane);
        bool more = need > 0;
        while (more) {
            const int fill0 = fill;
            WalkSel w;
            w.total = SVDQ_BLK_ROWS;      // no chunk left: flush the partial block
            if (have) {
                w = walk_select(mk, inv, fill0);
                if constexpr (SUB) {
#pragma unroll
                    for (int t = 0; t < NTP; ++t) v0[t] = v0[t] - vb;
                }
                const f32x4 mean = row_mean<NTP>(v0, NT, center);
#pragma unroll
                for (int t = 0; t < NTP; ++t) v0[t] = (t < NT) ? (v0[t] - mean) : zero4();
                if (gmean) {   // mean of the compacted rows (strip position 0 = row rb): consecutive lanes, consecutive rows
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (w.sel[e] && rb + w.pos[e] < D) gmean[rb + w.pos[e]] = mean[e];
                }
                walk_scatter<NTP>(X, v0, w, 0);
            } else {
                walk_zero_tail<NTP>(X, fill0, lane);
            }
            if (w.total >= SVDQ_BLK_ROWS) {
                wave_sync();
                compute(rb);
                rb += SVDQ_BLK_ROWS;
                if (have) walk_scatter<NTP>(X, v0, w, SVDQ_BLK_ROWS);
                fill = have ? w.total - SVDQ_BLK_ROWS : 0;
            } else {
                fill = w.total;
            }
            if (have) {
                produced += w.total - fill0;
                src += SVDQ_BLK_ROWS;
            }
            have = have && produced < need && src < src_end;
            if (have) walk_load<NTP, SUB>(v0, vb, mk, bp, gbase, gmask, src, src_end, lane);
            more = have || fill > 0;
        }
    } else {
        auto do_block = [&](f32x4 (&v)[NTP], int64_t rb) {
            if constexpr (SUB) {
#pragma unroll
                for (int t = 0; t < NTP; ++t) v[t] = v[t] - vb;
            }
            const f32x4 mean = center_store<NTP, GATHER>(v, NT, center, X, lane);
#ifdef SVDQ_ABLATE_STORES
            if (gmean && D < 0) {
#else
            if (gmean) {
#endif
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
            wave_sync();
            if (rb + AHEAD * SVDQ_BLK_ROWS < r_end) {
                if constexpr (GATHER) {
                    load_block_gather<NTP>(v, bp, ixn, rb + 2 * SVDQ_BLK_ROWS <= D);
                    if constexpr (SUB) vb = load_base_gather(gbase, ixn, rb + 2 * SVDQ_BLK_ROWS <= D);
                    if (rb + 2 * SVDQ_BLK_ROWS < r_end) ixn = load_idx(gidx, rb + 2 * SVDQ_BLK_ROWS, D, lane);
                } else {
                    load_block<NTP>(v, bp, rb + AHEAD * SVDQ_BLK_ROWS, D, lane);
                    if constexpr (SUB) vb = load_base(gbase, rb + AHEAD * SVDQ_BLK_ROWS, D, lane);
                }
            }
            compute(rb);
        };
        for (int64_t rb = r_begin; rb < r_end; rb += AHEAD * SVDQ_BLK_ROWS) {
            do_block(v0, rb);
#if SVDQ_PREFETCH2
            if (rb + SVDQ_BLK_ROWS < r_end) do_block(v1, rb + SVDQ_BLK_ROWS);
#endif
        }
    }

    // rounding-correction partials: cpart[slot][t*NT + i]; lane (c,g) holds D[m = U column][n = task]
    const int NN = NT * NT;
    if constexpr (PACK == 2) {
        const int rs = c >> 3, t = c & 7;
        double *dst = cpart + ((size_t)uidx * 2 + rs) * NN;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 4 * g + e;
            if ((m >> 3) == rs && (m & 7) < NT && t < NT) dst[t * NT + (m & 7)] = caccd[0][e];
        }
    } else {
        double *dst = cpart + (size_t)uidx * NN;
#pragma unroll
        for (int nbi = 0; nbi < NB; ++nbi)
#pragma unroll
            for (int nbt = 0; nbt < NB; ++nbt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * nbi + 4 * g + e;
                    const int t = 16 * nbt + c;
                    if (i < NT && t < NT) dst[t * NT + i] = caccd[nbi * NB + nbt][e];
                }
    }
}

SVDQ_STAMP_DECL(svdq_stamps_project)
template <int NTP, bool OUT16, int MODE, bool FULL>
__global__ __launch_bounds__(64) void k_basis_project(
    const SvdqParam *__restrict__ params, const SvdqUnit *__restrict__ units,
    const float *const *__restrict__ ptrs, const int64_t *__restrict__ rows_dev, int NT, int center,
    const float *__restrict__ Wtab, const int32_t *__restrict__ k_dev, const int32_t *__restrict__ r_dev,
    uint8_t *__restrict__ basis, float *__restrict__ meanbuf, double *__restrict__ cpart, int unit0, int reverse,
    const void *const *__restrict__ aux, const void *const *__restrict__ aux2, const int64_t *__restrict__ ustart) {
    using out_t = typename OutT<OUT16>::type;
    __shared__ __attribute__((aligned(16))) float X[NTP * XS];
    __shared__ __attribute__((aligned(16))) out_t OUT[SVDQ_BLK_ROWS * NTP + 16];  // +16: dump slot for idle lanes
    SVDQ_STAMP_BEGIN();
    const int uidx = unit0 + unit_of_block((int)blockIdx.x, (int)gridDim.x, reverse);
    bp_unit<NTP, OUT16, MODE, FULL>(X, OUT, uidx, params, units, ptrs, rows_dev, NT, center, Wtab, k_dev, r_dev, basis,
                                    meanbuf, cpart, aux, aux2, ustart);
    SVDQ_STAMP_END(svdq_stamps_project, uidx);
}


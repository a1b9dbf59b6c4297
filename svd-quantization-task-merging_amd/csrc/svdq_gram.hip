// svdq_gram.hip -- pass 1 of the SVD-Hybrid compressor: G = Tc^T Tc (k_gram).  Helpers and the layout notes: svdq_stream.h.
#include "svdq_stream.h"

// ------------------------------------------------------------------------------------ pass 1
// One work unit of pass 1 (a run of 256-row blocks of one parameter) by ONE wavefront.
// X: NTP*XS floats of wave-private LDS.
// MODE bit 0: gather through an index list (aux[p] = int32 list); bit 1: the task tensors are fine-tuned weights and a
// base tensor is subtracted in registers (aux2[p] = base).  0 = contiguous task vectors, 3 = both.
// F64: the products are accumulated by v_mfma_f64_16x16x4_f64 (exact fp32 x fp32 products, fp64 running sums over the
// whole unit), which resolves singular values down to ~1e-6 sigma_0 like the reference's LAPACK path; the fp32 form
// (fp32 sums inside a 256-row block) only reaches ~1e-3..1e-4 sigma_0.  Pass 1 is HBM-bound for N <= 16, so the fp64
// form is the default there; N > 16 would become MFMA-bound and keeps fp32 products.
template <int NTP, int MODE = 0, bool F64 = false, bool FULL = false>
__device__ __forceinline__ void gram_unit(float *X, int uidx, const SvdqParam *__restrict__ params,
                                          const SvdqUnit *__restrict__ units,
                                          const float *const *__restrict__ ptrs,
                                          const int64_t *__restrict__ rows_dev, int NT_arg, int center,
                                          double *__restrict__ gram_part,
                                          const void *const *__restrict__ aux = nullptr,
                                          const int32_t *__restrict__ only = nullptr,
                                          const void *const *__restrict__ aux2 = nullptr,
                                          const int64_t *__restrict__ ustart = nullptr) {
    constexpr bool GATHER = (MODE & 1) != 0, SUB = (MODE & 2) != 0, WALK = (MODE & 4) != 0;
    static_assert(!(GATHER && WALK), "index lists and the mask walk are alternatives");
    static_assert(!(MODE != 0 && SVDQ_PREFETCH2), "gather / minus-base support the one-block-ahead pipeline only");
    const int NT = FULL ? NTP : NT_arg;      // FULL: the plan has exactly NTP tasks, the "task t is real" tests fold away
    constexpr int PACK = (NTP <= 8) ? 2 : 1;
    constexpr int NB = (NTP + 15) / 16;
    // N = 17..20: of the 2x2-blocked Gram only AA is a full 16 x 16 tile; AB is 16 x 4 and BB 4 x 4.  A 16x16x4 MFMA per
    // k-step for each of them wastes 3/4 and 15/16 of the matrix pipe, and pass 1 at N = 20 is MFMA-issue-bound
    // (SQ_WAIT_INST_ANY 69 %, MFMA busy 58 %).  They are taken by v_mfma_f32_4x4x1_16b_f32 instead -- sixteen 4x4 outer
    // products per instruction, 8 cycles:  AB: block (mg, rg) = tasks 4mg..4mg+3 x tasks 16..19 on row 4rg + e of the
    // sub-tile (four instructions per 16 rows);  BB: block b = row b of the sub-tile (one instruction per 16 rows).
    // 2 688 matrix-pipe cycles per 256-row block instead of 4 096 plus the vector-ALU corner.
    constexpr bool VBB = (NTP == 20) && !F64;
    constexpr int NACC = (NB == 1 || VBB) ? 1 : 3;  // AA | AA, AB, BB
    // N = 5..8 with exact products: the 8 x 8 Gram is four 4 x 4 tiles, which is exactly one v_mfma_f64_4x4x4_4b_f64
    // (four blocks, K = 4 rows, 16 cycles) -- the 16x16x4 form spends 64 cycles on two useful 8 x 8 corners of a 16 x 16
    // tile.  Layout probed on the device (tools/probe/mfma_f64_layout.hip): lane l = (k = l >> 4, b = (l >> 2) & 3,
    // x = l & 3) supplies A_b[x][k] and B_b[k][x] and holds D_b[i = l >> 4][j = l & 3]; block b = tile (b >> 1, b & 1).
    // The lane's k selects rows 4k..4k+3 of a 16-row sub-tile (one 16-byte LDS read per operand), the four
    // instructions of a sub-tile take one of them each.  1 024 instead of 2 048 matrix-pipe cycles per block.
    // The same instruction serves every N <= 16 (T = NTP / 4 tile rows): only the T (T + 1) / 2 tiles on or above the
    // diagonal are computed, four per instruction -- T = 1: the four blocks take four different 16-row groups of the
    // one tile (summed at the end); T = 2: all four tiles in one instruction; T = 3: six tiles in two instructions;
    // (T = 4: ten tiles in three, was measured SLOWER than the 16x16x4 form -- 4.62 against 3.58 ms at ViT-L-14 x 16:
    // every 4x4x4 instruction needs two converted operands per lane, eight times the v_cvt_f64_f32 work per output --
    // so N = 13..16 stays on 16x16x4.)  Matrix-pipe cycles per four rows: 4 / 16 / 32 against 64 for the 16x16x4 form,
    // which at N <= 4 made pass 1 matrix-bound (1.52 -> 0.52 ms for the 2.4 GB of ViT-L-14 x 2; N = 4: 1.58 -> 0.83;
    // N = 12: 3.21 -> 2.98).
    constexpr bool Q64 = F64 && (NTP <= 12);
    constexpr int TQ = NTP / 4;
    constexpr int NTILE = TQ * (TQ + 1) / 2;
    constexpr int NSET = (TQ == 2) ? 1 : (NTILE + 3) / 4;   // T = 2 keeps its redundant (1,0) tile: one instruction anyway

    const int lane = threadIdx.x & 63;
    const SvdqUnit ud = units[uidx];
    const int p = ud.param;
    if (only && !only[p]) return;  // refinement pass (N > 16): only the parameters the eigen-stage flagged
    const int64_t D = rows_dev ? rows_dev[p] : params[p].rows;
    const int64_t r_begin = ud.row0;
    int64_t r_end = r_begin + ud.nrows;
    if (r_end > D) r_end = D;

    gfloat *bp[NTP];
#pragma unroll
    for (int t = 0; t < NTP; ++t) bp[t] = (gfloat *)ptrs[(size_t)p * NT + (t < NT ? t : NT - 1)];

    const int c = lane & 15, g = lane >> 4;
    double accd[NACC][4];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) accd[i][e] = 0.0;

#ifndef SVDQ_GRAM64_CHAINS
#define SVDQ_GRAM64_CHAINS 1
#endif
    constexpr int QC = (NB == 1) ? SVDQ_GRAM64_CHAINS : 1;  // F64: independent accumulation chains per block
    f64x4 accq[NACC * QC];
#pragma unroll
    for (int i = 0; i < NACC * QC; ++i) accq[i] = f64x4{0.0, 0.0, 0.0, 0.0};

    double q64[Q64 ? NSET : 1];   // Q64: this lane's Gram entries (one per instruction of a step), over the whole unit
#ifndef SVDQ_Q64_CHAINS
#define SVDQ_Q64_CHAINS 2   // independent accumulation chains per entry (even / odd k-steps), added at the end of the unit
#endif
    double q64b[(Q64 && SVDQ_Q64_CHAINS == 2) ? NSET : 1];
#pragma unroll
    for (int i = 0; i < (Q64 ? NSET : 1); ++i) q64[i] = 0.0;
#pragma unroll
    for (int i = 0; i < ((Q64 && SVDQ_Q64_CHAINS == 2) ? NSET : 1); ++i) q64b[i] = 0.0;
    // tile (ti <= tj) number q in row-major order of the upper triangle
    auto tile_of = [](int q, int &ti, int &tj) {
        ti = 0;
        int rowlen = TQ;
        while (q >= rowlen) {
            q -= rowlen;
            --rowlen;
            ++ti;
        }
        tj = ti + q;
    };
    double qd[2][4];  // VBB: AB and BB block partials of the 4x4x1 chains (fp32 inside a block, fp64 across)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) qd[i][e] = 0.0;
    const int b4 = lane >> 2, i4 = lane & 3, mg4 = b4 & 3, rg4 = b4 >> 2;

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

    // the MFMA phase over the strip of one block (ends with the barrier that frees the strip)
    auto compute = [&]() {
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = zero4();
        f32x4 qab = zero4(), qbb = zero4();

        if constexpr (Q64) {
            const int kq = lane >> 4, bq = (lane >> 2) & 3, xq = lane & 3;
            if constexpr (TQ == 1) {
                // one tile: block b takes rows 16b..16b+15 of every 64-row group; A and B are the same value
                const float *pa = X + xq * XS + 16 * bq + 4 * kq;
UNROLL_N(4)
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(pa + 64 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double ad = (double)a[e];
                        if (SVDQ_Q64_CHAINS == 2 && (e & 1))
                            q64b[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ad, ad, q64b[0], 0, 0, 0);
                        else
                            q64[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ad, ad, q64[0], 0, 0, 0);
                    }
                }
            } else {
                const float *pa[NSET], *pb[NSET];
#pragma unroll
                for (int st = 0; st < NSET; ++st) {
                    int ti, tj;
                    if constexpr (TQ == 2) {
                        ti = bq >> 1;
                        tj = bq & 1;
                    } else {
                        const int q = 4 * st + bq;
                        tile_of(q < NTILE ? q : NTILE - 1, ti, tj);   // spare blocks repeat the last tile, unused
                    }
                    pa[st] = X + (4 * ti + xq) * XS + 4 * kq;
                    pb[st] = X + (4 * tj + xq) * XS + 4 * kq;
                }
UNROLL_N(SVDQ_UNROLL_GRAM)
                for (int j = 0; j < 16; ++j) {
#pragma unroll
                    for (int st = 0; st < NSET; ++st) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(pa[st] + 16 * j);
                        const f32x4 b = *reinterpret_cast<const f32x4 *>(pb[st] + 16 * j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (SVDQ_Q64_CHAINS == 2 && (e & 1))
                                q64b[st] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)a[e], (double)b[e], q64b[st], 0, 0, 0);
                            else
                                q64[st] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)a[e], (double)b[e], q64[st], 0, 0, 0);
                        }
                    }
                }
            }
        } else if constexpr (PACK == 2) {
            const int t = c & 7;
            const bool valid = t < NTP;
            const float *xr = X + (valid ? t : 0) * XS + 16 * (c >> 3) + 4 * g;
UNROLL_N(SVDQ_UNROLL_GRAM)
            for (int j = 0; j < 8; ++j) {
                f32x4 a = *reinterpret_cast<const f32x4 *>(xr + 32 * j);
                if (!valid) a = zero4();
                if constexpr (F64) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double ad = (double)a[e];
                        accq[e % QC] = mfma4d(ad, ad, accq[e % QC]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0] = mfma4(a[e], a[e], acc[0]);
                }
            }
        } else {
            const bool v0ok = c < NTP;
            const bool v1ok = (NB == 2) && (16 + c < NTP);
            const float *x0 = X + (v0ok ? c : 0) * XS + 4 * g;
            const float *x1 = X + (v1ok ? 16 + c : 0) * XS + 4 * g;
UNROLL_N(SVDQ_UNROLL_GRAM_P1)
            for (int j = 0; j < 16; ++j) {
                f32x4 a0 = *reinterpret_cast<const f32x4 *>(x0 + 16 * j);
                if (!v0ok) a0 = zero4();
                if constexpr (F64 && NB == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double ad = (double)a0[e];
                        accq[e % QC] = mfma4d(ad, ad, accq[e % QC]);
                    }
                } else if constexpr (F64) {
                    f32x4 a1 = *reinterpret_cast<const f32x4 *>(x1 + 16 * j);
                    if (!v1ok) a1 = zero4();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double d0 = (double)a0[e], d1 = (double)a1[e];
                        accq[0] = mfma4d(d0, d0, accq[0]);
                        accq[1] = mfma4d(d0, d1, accq[1]);
                        accq[2] = mfma4d(d1, d1, accq[2]);
                    }
                } else if constexpr (NB == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0] = mfma4(a0[e], a0[e], acc[0]);
                } else if constexpr (VBB) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0] = mfma4(a0[e], a0[e], acc[0]);
                    const f32x4 xa = *reinterpret_cast<const f32x4 *>(X + (4 * mg4 + i4) * XS + 16 * j + 4 * rg4);
                    const f32x4 xq = *reinterpret_cast<const f32x4 *>(X + (16 + i4) * XS + 16 * j + 4 * rg4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) qab = mfma_4x4x1(xa[e], xq[e], qab);
                    const float xr = X[(16 + i4) * XS + 16 * j + b4];
                    qbb = mfma_4x4x1(xr, xr, qbb);
                } else {
                    f32x4 a1 = *reinterpret_cast<const f32x4 *>(x1 + 16 * j);
                    if (!v1ok) a1 = zero4();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = mfma4(a0[e], a0[e], acc[0]);
                        acc[1] = mfma4(a0[e], a1[e], acc[1]);
                        acc[2] = mfma4(a1[e], a1[e], acc[2]);
                    }
                }
            }
        }
        if constexpr (!F64) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) accd[i][e] += (double)acc[i][e];
            if constexpr (VBB) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qd[0][e] += (double)qab[e];
                    qd[1][e] += (double)qbb[e];
                }
            }
        }
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
                walk_scatter<NTP>(X, v0, w, 0);
            } else {
                walk_zero_tail<NTP>(X, fill0, lane);
            }
            if (w.total >= SVDQ_BLK_ROWS) {
                wave_sync();
                compute();
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
            center_store<NTP, GATHER>(v, NT, center, X, lane);
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
            compute();
        };
        for (int64_t rb = r_begin; rb < r_end; rb += AHEAD * SVDQ_BLK_ROWS) {
            do_block(v0, rb);
#if SVDQ_PREFETCH2
            if (rb + SVDQ_BLK_ROWS < r_end) do_block(v1, rb + SVDQ_BLK_ROWS);
#endif
        }
    }
    if constexpr (F64) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double t = accq[i * QC][e];
#pragma unroll
                for (int q = 1; q < QC; ++q) t += accq[i * QC + q][e];
                accd[i][e] = t;
            }
    }

    // One fp64 partial per slot, dense [NT][NT].  Lane (c,g) holds D[4g+e][c] (fp32 MFMA) or D[g+4e][c] (fp64 MFMA).
    const int NN = NT * NT;
    if constexpr (Q64 && SVDQ_Q64_CHAINS == 2) {
#pragma unroll
        for (int st = 0; st < NSET; ++st) q64[st] += q64b[st];
    }
    if constexpr (Q64) {   // everything in the unit's first slot; N <= 8 has a second slot per unit: zeros
        const int i = lane >> 4, bq = (lane >> 2) & 3, jq = lane & 3;
        double *dst = gram_part + (size_t)uidx * PACK * NN;
        if constexpr (TQ == 1) {
            double x = q64[0];
            x += __shfl_xor(x, 4);
            x += __shfl_xor(x, 8);
            if (bq == 0 && i < NT && jq < NT) {
                dst[i * NT + jq] = x;
                dst[NN + i * NT + jq] = 0.0;
            }
        } else if constexpr (TQ == 2) {
            const int m = 4 * (bq >> 1) + i, n = 4 * (bq & 1) + jq;
            if (m < NT && n < NT) {
                dst[m * NT + n] = q64[0];
                dst[NN + m * NT + n] = 0.0;
            }
        } else {
#pragma unroll
            for (int st = 0; st < NSET; ++st) {
                const int q = 4 * st + bq;
                if (q < NTILE) {
                    int ti, tj;
                    tile_of(q, ti, tj);
                    const int m = 4 * ti + i, n = 4 * tj + jq;
                    if (m < NT && n < NT) {
                        dst[m * NT + n] = q64[st];
                        if (ti != tj) dst[n * NT + m] = q64[st];
                    }
                }
            }
        }
    } else if constexpr (PACK == 2) {
        const int rs = c >> 3, n = c & 7;
        double *dst = gram_part + ((size_t)uidx * 2 + rs) * NN;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = F64 ? g + 4 * e : 4 * g + e;
            if ((m >> 3) == rs && (m & 7) < NT && n < NT) dst[(m & 7) * NT + n] = accd[0][e];
        }
    } else {
        double *dst = gram_part + (size_t)uidx * NN;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = F64 ? g + 4 * e : 4 * g + e;
            if (m < NT && c < NT) dst[m * NT + c] = accd[0][e];
            if constexpr (NB == 2 && !VBB) {
                if (m < NT && 16 + c < NT) {
                    dst[m * NT + 16 + c] = accd[1][e];
                    dst[(16 + c) * NT + m] = accd[1][e];
                }
                if (16 + m < NT && 16 + c < NT) dst[(16 + m) * NT + 16 + c] = accd[2][e];
            }
        }
        if constexpr (VBB) {   // block partials meet in fixed-order shuffles
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // AB: lane (block (mg, rg), j) register e = [task 4mg + e][task 16 + j] over the rows of row group rg
                double x = qd[0][e];
                x += __shfl_xor(x, 16);
                x += __shfl_xor(x, 32);
                const int m = 4 * mg4 + e;
                if (rg4 == 0 && m < NT && 16 + i4 < NT) {
                    dst[m * NT + 16 + i4] = x;
                    dst[(16 + i4) * NT + m] = x;
                }
                // BB: lane (block b, j) register e = [task 16 + e][task 16 + j] over the rows b mod 16
                double y = qd[1][e];
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) y += __shfl_xor(y, off);
                if (b4 == 0 && 16 + e < NT && 16 + i4 < NT) dst[(16 + e) * NT + 16 + i4] = y;
            }
        }
    }
}

// second launch bound = waves per SIMD the register allocation must leave room for: the fp64 accumulators would
// otherwise cost the N <= 8 and the N <= 16 kernels one resident wave each (measured: -9 % bandwidth)
#ifndef SVDQ_GRAM64_WAVES8
#define SVDQ_GRAM64_WAVES8 5
#endif
#ifndef SVDQ_GRAM64_WAVES16
#define SVDQ_GRAM64_WAVES16 3
#endif
SVDQ_STAMP_DECL(svdq_stamps_gram)
template <int NTP, int MODE, bool F64, bool FULL>
__global__ __launch_bounds__(64, (F64 && (MODE == 0 || MODE == 4) && NTP <= 16) ? (NTP <= 8 ? SVDQ_GRAM64_WAVES8 : SVDQ_GRAM64_WAVES16) : 1) void k_gram(const SvdqParam *__restrict__ params,
                                             const SvdqUnit *__restrict__ units,
                                             const float *const *__restrict__ ptrs,
                                             const int64_t *__restrict__ rows_dev, int NT, int center,
                                             double *__restrict__ gram_part, int unit0,
                                             const void *const *__restrict__ aux,
                                             const int32_t *__restrict__ only,
                                             const void *const *__restrict__ aux2, int order,
                                             const int64_t *__restrict__ ustart) {
    __shared__ __attribute__((aligned(16))) float X[NTP * XS];
    SVDQ_STAMP_BEGIN();
    const int uidx = unit0 + unit_of_block((int)blockIdx.x, (int)gridDim.x, order);
    gram_unit<NTP, MODE, F64, FULL>(X, uidx, params, units, ptrs, rows_dev, NT, center, gram_part, aux, only, aux2, ustart);
    SVDQ_STAMP_END(svdq_stamps_gram, uidx);
}

// ------------------------------------------------------------------------------------ launchers
// idx: NULL or the device table of index lists (gather mode); base: NULL or the device table of base tensors
// (minus-base mode); both may be given (masked parameters straight from checkpoints).  ustart: NULL, or the per-unit
// source start positions of the walk mode -- idx is then the device table of combined MASK byte tensors.
template <int NTP>
static int launch_gram_t(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, double *gram_part,
                         int unit0, int nunits, int center, const void *idx, const void *base, int f64,
                         const int32_t *only, const int64_t *ustart, hipStream_t st) {
    auto pp = reinterpret_cast<const float *const *>(ptrs);
    auto ai = (const void *const *)idx, ab = (const void *const *)base;
#define SVDQ_LAUNCH_GRAM_(M, F, FULL_)                                                                                     \
    hipLaunchKernelGGL((k_gram<NTP, M, F, FULL_>), dim3(nunits), dim3(64), 0, st, pl->d_params, pl->d_units, pp, rows_dev, \
                       pl->n_tasks, center, gram_part, unit0, ai, only, ab, pl->cfg.reserved & 4, ustart)
    // the plain and the mask-walk mode have a variant for plans with exactly NTP tasks
#define SVDQ_LAUNCH_GRAM(M, F)                                                                                       \
    do {                                                                                                             \
        if constexpr ((M) == 0 || (M) == 4) {                                                                        \
            if (pl->n_tasks == NTP) {                                                                                \
                SVDQ_LAUNCH_GRAM_(M, F, true);                                                                       \
                break;                                                                                               \
            }                                                                                                        \
        }                                                                                                            \
        SVDQ_LAUNCH_GRAM_(M, F, false);                                                                              \
    } while (0)
    const int mode = ustart ? (4 | (base ? 2 : 0)) : ((idx ? 1 : 0) | (base ? 2 : 0));
    if constexpr (NTP <= 16) {
        if (mode & 4) {      // walk mode exists for the one-wave kernels (N <= 16), always with the default Gram
            if (f64) {
                if (mode == 4) SVDQ_LAUNCH_GRAM(4, true); else SVDQ_LAUNCH_GRAM(6, true);
            } else {
                if (mode == 4) SVDQ_LAUNCH_GRAM(4, false); else SVDQ_LAUNCH_GRAM(6, false);
            }
            return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
        }
    }
    if (mode & 4) {      // 16 < N <= 32: the same one-wave kernel walks (pass 2: svdq_project_walk.hip); not from checkpoints
        if (mode != 4) {
            svdq_set_error("the mask walk straight from checkpoints covers N <= 16 tasks (got %d): use the index lists "
                           "(svdq_compress_gather_from_base)", pl->n_tasks);
            return SVDQ_EUNSUPPORTED;
        }
        if (f64) SVDQ_LAUNCH_GRAM_(4, true, false); else SVDQ_LAUNCH_GRAM_(4, false, false);
        return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
    }
    if (f64) {
        switch (mode) {
            case 0: SVDQ_LAUNCH_GRAM(0, true); break;
            case 1: SVDQ_LAUNCH_GRAM(1, true); break;
            case 2: SVDQ_LAUNCH_GRAM(2, true); break;
            default: SVDQ_LAUNCH_GRAM(3, true); break;
        }
    } else {
        switch (mode) {
            case 0: SVDQ_LAUNCH_GRAM(0, false); break;
            case 1: SVDQ_LAUNCH_GRAM(1, false); break;
            case 2: SVDQ_LAUNCH_GRAM(2, false); break;
            default: SVDQ_LAUNCH_GRAM(3, false); break;
        }
    }
#undef SVDQ_LAUNCH_GRAM
#undef SVDQ_LAUNCH_GRAM_
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

// f64: accumulate the products with v_mfma_f64_16x16x4_f64 (exact) instead of fp32 MFMA; only: NULL, or a device
// table [n_params] -- units of parameters whose entry is 0 return at once (the refinement pass of N > 16)
int svdq_launch_gram(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, double *gram_part,
                     int unit0, int nunits, int center, const void *idx, const void *base, int f64,
                     const int32_t *only, hipStream_t st, const int64_t *ustart) {
#define SVDQ_GRAM_CASE(n) \
    case n: return launch_gram_t<n>(pl, ptrs, rows_dev, gram_part, unit0, nunits, center, idx, base, f64, only, ustart, st)
    switch (pl->ntp) {
        SVDQ_GRAM_CASE(4); SVDQ_GRAM_CASE(8); SVDQ_GRAM_CASE(12); SVDQ_GRAM_CASE(16);
        SVDQ_GRAM_CASE(20); SVDQ_GRAM_CASE(24); SVDQ_GRAM_CASE(28); SVDQ_GRAM_CASE(32);
    }
#undef SVDQ_GRAM_CASE
    svdq_set_error("unsupported padded task count %d", pl->ntp);
    return SVDQ_EUNSUPPORTED;
}


#ifdef SVDQ_UNIT_STAMPS
extern "C" int svdq_debug_stamps_gram(unsigned long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(svdq_stamps_gram), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

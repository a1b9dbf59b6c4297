// svdq_common.h -- shared declarations of the gfx950 SVD-Hybrid compressor library.
//
// Work decomposition:
//   parameter p : D_p rows (elements) x N tasks; the N task deltas are N separate fp32 buffers.
//   block       : 256 consecutive rows of one parameter, staged (centred) in LDS as [task][row].
//   unit        : a run of blocks of one parameter handled by ONE wavefront (= one 64-thread
//                 workgroup).  Each unit emits one fp64 partial N x N matrix per "slot".
//   slot        : with row-set packing (N <= 8) a 16x16 MFMA tile carries two independent
//                 8-task row sets, so a unit has 2 slots; otherwise 1.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svdq.h"

#define SVDQ_BLK_ROWS 256
#define SVDQ_RC 16  // level-2 partial chunks per parameter (k_reduce)
#ifndef SVDQ_XS
#define SVDQ_XS 260  // LDS row stride (floats) of one task's 256-row strip: 256 + pad, multiple of 4 (16-B alignment)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SvdqParam {     // one per parameter (device table)
    int64_t rows;      // D_p (upper bound when rows_dev overrides)
    int64_t slab_off;  // byte offset of the U slab in the packed basis buffer (256-aligned)
    int64_t mean_off;  // float offset in the packed mean buffer (64-aligned)
    int32_t unit_begin;
    int32_t unit_count;
};

struct SvdqUnit {      // one per work unit (device table)
    int32_t param;
    int32_t nrows;     // rows covered by this unit (multiple of 256 except the last unit of a parameter)
    int64_t row0;
};

struct svdq_plan {
    int32_t n_tasks, n_params, n_units, n_slots, ntp, pack;
    svdq_config cfg;
    svdq_sizes sizes;
    svdq_small_layout small;
    // workspace offsets (bytes)
    int64_t ws_gram_off, ws_cpart_off, ws_w_off, ws_c0_off, ws_gram2_off, ws_cpart2_off, ws_flag_off;
    // host copies
    SvdqParam *h_params;
    SvdqUnit *h_units;
    // device tables
    SvdqParam *d_params;
    SvdqUnit *d_units;
    int32_t *d_bits;  // optional per-parameter low_bits (svdq_plan_set_low_bits), NULL = cfg.low_bits everywhere
};

__host__ __device__ static inline int64_t svdq_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// padded task count the streaming kernels are instantiated for (multiple of 4, <= 32)
static inline int svdq_ntp(int n) { return (n + 3) / 4 * 4; }

void svdq_set_error(const char *fmt, ...);

// launchers (defined in the .hip files)
// idx: NULL, or a device table [n_params] of int32 index lists (gather mode, see svdq_compress_gather);
// base: NULL, or a device table [n_params] of base tensors (minus-base mode, see svdq_compress_from_base);
// only: NULL, or a device table [n_params] of int32 -- parameters whose entry is 0 are skipped (refinement pass)
// ustart: NULL, or the walk mode's per-unit source start positions [n_units] (svdq_maskset_*_starts); idx then names
// the combined MASK byte tensors instead of index lists (svdq_compress_masked)
int svdq_launch_gram(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, double *gram_part,
                     int unit0, int nunits, int center, const void *idx, const void *base, int f64,
                     const int32_t *only, hipStream_t st, const int64_t *ustart = nullptr);
int svdq_launch_gram_total(const svdq_plan *pl, const double *part2, double *out, hipStream_t st);
int svdq_launch_basis_project(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, const float *W,
                              const int32_t *k_dev, const int32_t *r_dev, uint8_t *basis, float *mean,
                              double *cpart, int unit0, int nunits, int reverse, const void *idx, const void *base,
                              hipStream_t st, const int64_t *ustart = nullptr);
// pass 2 of the mask-walk mode for 16 < N <= 32 (svdq_project_walk.hip)
int svdq_launch_basis_project_walk32(const svdq_plan *pl, const float *const *pp, const int64_t *rows_dev, const float *W,
                                     const int32_t *k_dev, const int32_t *r_dev, uint8_t *basis, float *mean, double *cpart,
                                     int unit0, int nunits, int reverse, const void *const *masks, const int64_t *ustart,
                                     hipStream_t st);
// refine_out: NULL, or a device table [n_params] that receives 1 where a singular value lies in the band the fp32-product
// Gram does not resolve (then the caller re-accumulates those parameters in fp64 and calls again with only = that table)
int svdq_launch_eig(const svdq_plan *pl, const void *ptrs, const int64_t *rows_dev, const double *gram_part, float *W,
                    double *c0, uint8_t *small, int param0, int nparams, const void *idx, const void *base,
                    const int32_t *only, int32_t *refine_out, hipStream_t st, const int64_t *ustart = nullptr);
int svdq_launch_reduce(const svdq_plan *pl, const double *part, double *part2, int param0, int nparams,
                       const int32_t *only, hipStream_t st);
int svdq_launch_coeff(const svdq_plan *pl, const double *cpart, const double *c0, uint8_t *small, int param0,
                      int nparams, hipStream_t st);

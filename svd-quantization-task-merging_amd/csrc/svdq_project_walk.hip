// svdq_project_walk.hip -- pass 2 in the mask-walk mode for 16 < N <= 32 tasks (svdq_compress_masked).
// The two-wave kernels that serve N > 16 otherwise (svdq_project.hip) live on a one-block-ahead prefetch at 218-243
// registers; the walk keeps a chunk's overflow rows in registers across the block's compute phase and cannot share that
// budget.  So the walk takes the ONE-wave kernel here -- the form N <= 16 uses, the whole strip and
// both 16-column halves in one wavefront: 256 vector + 95-152 accumulation registers, one wave per SIMD, no scratch.
// Slower per byte than the two-wave kernels, but it reads the mask byte beside the rows instead of a 4-byte index per
// selected row, builds no index lists, and hands the batched consumers the same unit starts; artifacts bit-identical to
// the index-list route.  A translation unit of its own so that the build compiles it beside svdq_project.hip.
#include "svdq_project_unit.h"

template <int NTP>
static void launch_walk(const svdq_plan *pl, const float *const *pp, const int64_t *rows_dev, const float *W,
                        const int32_t *k_dev, const int32_t *r_dev, uint8_t *basis, float *mean, double *cpart, int unit0,
                        int nunits, int reverse, const void *const *masks, const int64_t *ustart, hipStream_t st) {
    if (pl->cfg.fp16)
        hipLaunchKernelGGL((k_basis_project<NTP, true, 4, false>), dim3(nunits), dim3(64), 0, st, pl->d_params, pl->d_units, pp,
                           rows_dev, pl->n_tasks, pl->cfg.center, W, k_dev, r_dev, basis, mean, cpart, unit0, reverse, masks,
                           (const void *const *)nullptr, ustart);
    else
        hipLaunchKernelGGL((k_basis_project<NTP, false, 4, false>), dim3(nunits), dim3(64), 0, st, pl->d_params, pl->d_units, pp,
                           rows_dev, pl->n_tasks, pl->cfg.center, W, k_dev, r_dev, basis, mean, cpart, unit0, reverse, masks,
                           (const void *const *)nullptr, ustart);
}

int svdq_launch_basis_project_walk32(const svdq_plan *pl, const float *const *pp, const int64_t *rows_dev, const float *W,
                                     const int32_t *k_dev, const int32_t *r_dev, uint8_t *basis, float *mean, double *cpart,
                                     int unit0, int nunits, int reverse, const void *const *masks, const int64_t *ustart,
                                     hipStream_t st) {
    switch (pl->ntp) {
        case 20: launch_walk<20>(pl, pp, rows_dev, W, k_dev, r_dev, basis, mean, cpart, unit0, nunits, reverse, masks, ustart, st); break;
        case 24: launch_walk<24>(pl, pp, rows_dev, W, k_dev, r_dev, basis, mean, cpart, unit0, nunits, reverse, masks, ustart, st); break;
        case 28: launch_walk<28>(pl, pp, rows_dev, W, k_dev, r_dev, basis, mean, cpart, unit0, nunits, reverse, masks, ustart, st); break;
        case 32: launch_walk<32>(pl, pp, rows_dev, W, k_dev, r_dev, basis, mean, cpart, unit0, nunits, reverse, masks, ustart, st); break;
        default:
            svdq_set_error("svdq_launch_basis_project_walk32: padded task count %d", pl->ntp);
            return SVDQ_EUNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? SVDQ_OK : SVDQ_EHIP;
}

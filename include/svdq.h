/*
 * include/svdq.h -- C ABI of libsvdq_hip.so: the MI355X (gfx950) SVD-Hybrid task-vector
 * compressor hot path.  extern "C", plain pointers and sizes, no framework types.
 *
 * The reference (mgradyn/SVD-Quantization-Task-Merging) is pure Python and has NO FFI /
 * operator interface for this path: its boundary is a set of
 * Python callables exchanging dicts.  Each entry point below states which of those
 * callables (reference file:line) it replaces; the Python host layer in
 * svd-quantization-task-merging_amd/ rebuilds the reference's exact signatures and dict
 * layouts on top of this ABI through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - Every pointer named *_dev is a DEVICE pointer (HBM).  Nothing here copies to or from
 *    the host except svdq_plan_create (tables, once) -- no entry point synchronises the
 *    stream or the device; all work is enqueued on `stream` (a hipStream_t passed as void*).
 *  - Return value: SVDQ_OK or a negative SVDQ_E* code; svdq_last_error() gives text.
 *  - Task delta buffers are fp32, contiguous, one buffer per (parameter, task), base
 *    address 16-byte aligned; the [D,N] stack of basis.py:103 is never materialised.
 *  - N (tasks) <= SVDQ_MAX_TASKS.
 */
#ifndef SVDQ_H
#define SVDQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVDQ_ABI_VERSION 1
#define SVDQ_MAX_TASKS 32
#define SVDQ_MAX_STAGES 8

enum {
    SVDQ_OK = 0,
    SVDQ_EINVAL = -1,     /* bad argument (the reference raises ValueError) */
    SVDQ_EHIP = -2,       /* HIP runtime error */
    SVDQ_EUNSUPPORTED = -3
};

enum { SVDQ_MASK_UNION = 0, SVDQ_MASK_INTERSECTION = 1, SVDQ_MASK_MAJORITY = 2 };

/* Run-wide settings; field names follow SVDHybridConfig (reference src/svd_hybrid/config.py:157-205). */
typedef struct svdq_config {
    float   energy_threshold;  /* svd_energy_threshold, (0,1]                       */
    int32_t max_rank;          /* svd_max_rank, <= 0 means None                      */
    int32_t center;            /* svd_center                                         */
    int32_t fp16;              /* svd_fp16: basis stored (and projected with) fp16   */
    int32_t low_bits;          /* svd_low_bits 1..8                                  */
    int32_t rtvq_stages;       /* svd_rtvq_stages 1..SVDQ_MAX_STAGES                 */
    int32_t unit_rows;         /* 0 = auto; rows per work unit (multiple of 256)     */
    int32_t reserved;          /* measurement switches, 0 in production.  bit 0: reverse unit order in pass 2;
                                  bit 1: fp32-product Gram for every N (the round-1 kernel; sigma then only resolves
                                  down to ~3e-4 sigma_0 and smaller ones are treated as null directions);
                                  bit 2: XCD-chunked unit order (XCD x walks a contiguous eighth of the units; measured: no change);
                                  bit 3: N = 17..20: the two-wave pass 2 instead of the one-wave 4x4-block kernel. */
} svdq_config;

/* Byte sizes / strides the caller needs to allocate outputs (all device memory). */
typedef struct svdq_sizes {
    int64_t workspace_bytes;   /* scratch: Gram / projection partials, W, ranks               */
    int64_t basis_bytes;       /* packed U buffer (see svdq_plan_basis_layout)                 */
    int64_t mean_floats;       /* packed mean buffer, floats                                   */
    int64_t small_bytes;       /* packed small artifacts (see svdq_small_layout)               */
    int32_t n_units;           /* work units (one wavefront each) in passes 1 and 2            */
    int32_t n_slots;           /* partial-sum slots                                            */
} svdq_sizes;

/* Offsets (bytes) of the typed arrays inside the packed "small artifacts" buffer.
 * P = parameters, N = tasks, S = stages.  Everything a caller needs on the host after a
 * run is in this one buffer, so one D2H copy fetches it. */
typedef struct svdq_small_layout {
    int64_t sigma_off;      /* float   [P][N]       singular values, descending (first r valid)   */
    int64_t k_off;          /* int32   [P]          selected rank k                                */
    int64_t r_off;          /* int32   [P]          r = min(D, N) = number of singular values      */
    int64_t energy_off;     /* float   [P]          energy_retained = cum_energy[k-1]              */
    int64_t rows_off;       /* int64   [P]          rows actually processed (D, or mask.sum())     */
    int64_t chigh_off;      /* uint16  [P][N][N]    fp16 bits of c_high per task (first k valid)   */
    int64_t codes_off;      /* uint8   [P][N][S][N] RTVQ codes of c_low per task (first r-k valid) */
    int64_t scale_off;      /* float   [P][N][S]                                                   */
    int64_t zp_off;         /* float   [P][N][S]                                                   */
    int64_t rnorm_off;      /* float   [P][N][S]    residual norm before each stage                */
    int64_t coef_off;       /* float   [P][N][N]    fp32 coefficients c[t][i] before rounding      */
    int64_t status_off;     /* int32   [1]          written by svdq_compress: 0 = ok, 1 = fused schedule timed out */
    int64_t total_bytes;
} svdq_small_layout;

typedef struct svdq_plan svdq_plan;

/* ---- library ---------------------------------------------------------------------------- */
int         svdq_abi_version(void);
const char *svdq_last_error(void);

/* ---- plan: a ragged batch of P parameter tensors x N tasks -------------------------------
 * rows[p] = D_p, the element count of parameter p (upper bound when a mask is applied on
 * device).  Replaces the Python loop state of cli.py:317 (Step 4) / compress.py:187 (Step 5).
 * Allocates and fills small device tables (hipMalloc + hipMemcpy): call outside timed code. */
int  svdq_plan_create(svdq_plan **plan, int32_t n_tasks, int32_t n_params, const int64_t *rows,
                      const svdq_config *cfg);
void svdq_plan_destroy(svdq_plan *plan);
/* Per-parameter code width for the low-energy coefficients (host array [n_params], each in [1, 8]); NULL
 * restores cfg.low_bits for every parameter.  The reference builds ONE RTVQQuantizer(config.svd_low_bits, ...)
 * per run (compress.py:180-183); BASELINE config #5's "mixed 8-bit / 2-bit" run is two quantizer instances over
 * a partition of the parameters, which this expresses inside one plan. */
int  svdq_plan_set_low_bits(svdq_plan *plan, const int32_t *bits);
int  svdq_plan_sizes(const svdq_plan *plan, svdq_sizes *out);
int  svdq_plan_small_layout(const svdq_plan *plan, svdq_small_layout *out);
/* Per parameter: byte offset of its slab in the packed basis buffer and float offset of its
 * mean in the packed mean buffer.  Inside a slab (rows = rows processed, e = 2 or 4 bytes):
 *   U_high [rows, k]     row-major at slab + 0
 *   U_low  [rows, r-k]   row-major at slab + align256(rows * k * e)
 * i.e. the two .contiguous() tensors of basis.py:363-364, already cast per cli.py:354-361. */
int  svdq_plan_basis_layout(const svdq_plan *plan, int64_t *slab_off_bytes /*[P]*/,
                            int64_t *mean_off_floats /*[P]*/);

/* ---- pass 1: centred Gram  (stack_and_center basis.py:63-113 + first half of compute_svd
 *      basis.py:216-249).  G_p = Tc^T Tc accumulated by fp32 MFMA per 256-row block and fp64
 *      across blocks; one fp64 partial per work unit goes to the workspace.
 *      delta_ptrs_dev: device array [P*N] of const float* (parameter-major, task order =
 *      column order of basis.py:103).  rows_dev: optional device int64 [P] overriding rows
 *      (mask.sum() computed on device); NULL = plan rows. */
int svdq_gram_center(const svdq_plan *plan, const void *delta_ptrs_dev, const int64_t *rows_dev,
                     void *workspace_dev, void *stream);

/* ---- N x N eigen-solve + rank selection  (second half of compute_svd; compute_energy_spectrum
 *      basis.py:116-156; select_rank basis.py:159-213; energy_retained basis.py:367).
 *      Cyclic Jacobi in fp64, sigma = sqrt(lambda) -> fp32, fp32 cumsum rule of the reference.
 *      Writes sigma / k / r / energy / rows into small_dev and W = V Sigma^-1 into the workspace.
 *      Reads row 0 of every task (delta_ptrs_dev) to build the orthonormal completion column of the
 *      null direction that centring creates (LAPACK returns an arbitrary orthonormal vector there). */
int svdq_eig_rank_select(const svdq_plan *plan, const void *delta_ptrs_dev, const int64_t *rows_dev,
                         void *workspace_dev, void *small_dev, void *stream);

/* ---- pass 2: basis + projection  (U = Tc W and the split of basis.py:363-364, the fp16 cast
 *      of cli.py:354-361, mean of basis.py:109, and project_to_basis compress.py:6-21 for all
 *      N tasks against the ROUNDED basis, fused in one streaming pass).
 *      Writes U_high/U_low slabs into basis_dev, means into mean_dev (NULL allowed when
 *      center == 0) and fp64 projection partials into the workspace. */
int svdq_basis_project(const svdq_plan *plan, const void *delta_ptrs_dev, const int64_t *rows_dev,
                       void *workspace_dev, const void *small_dev, void *basis_dev, float *mean_dev,
                       void *stream);

/* ---- coefficient epilogue  (compress_single_task compress.py:42-56: c_high -> fp16,
 *      c_low -> RTVQQuantizer.quantize rtvq.py:39-82, for every (parameter, task)). */
int svdq_coeff_quantize(const svdq_plan *plan, void *workspace_dev, void *small_dev, void *stream);

/* ---- the same four stages restricted to parameters [param0, param0 + nparams) of the plan.
 *      They let a caller pipeline groups of parameters (gram of group g+1 while the eigen-solve
 *      of group g runs on a second stream), which keeps a group's deltas resident in the 256 MiB
 *      Infinity Cache between its two streaming passes ("cache-resident pipeline"). */
int svdq_gram_center_range(const svdq_plan *plan, const void *delta_ptrs_dev, const int64_t *rows_dev,
                           void *workspace_dev, int32_t param0, int32_t nparams, void *stream);
int svdq_eig_rank_select_range(const svdq_plan *plan, const void *delta_ptrs_dev, const int64_t *rows_dev,
                               void *workspace_dev, void *small_dev, int32_t param0, int32_t nparams,
                               void *stream);
int svdq_basis_project_range(const svdq_plan *plan, const void *delta_ptrs_dev, const int64_t *rows_dev,
                             void *workspace_dev, const void *small_dev, void *basis_dev, float *mean_dev,
                             int32_t param0, int32_t nparams, void *stream);
int svdq_coeff_quantize_range(const svdq_plan *plan, void *workspace_dev, void *small_dev, int32_t param0,
                              int32_t nparams, void *stream);

/* Uncentred Gram of the CONCATENATED task vectors: out_gram[N*N] (device, fp64, row-major) =
 * sum over parameters of T_p^T T_p.  Replaces the [N, sum D] host feature matrix of
 * flatten_task_vectors (src/svd_hybrid/clustering.py:55-120): every quantity cluster_tasks (:198-245)
 * and compute_cluster_statistics (:263-316) need is a function of these N*N inner products.
 * Overwrites the pass-1 partials in `workspace`; multi-GPU callers all-reduce out_gram. */
int svdq_task_gram(const svdq_plan *plan, const void *delta_ptrs, const int64_t *rows_dev, void *workspace,
                   double *out_gram, void *stream);

/* svdq_compress for MASKED parameters without a compaction pass (replaces the 2*N boolean-index gathers per
 * parameter of cli.py:333 / compress.py:144, i.e. apply_mask_to_tensor mask_loader.py:651-679 feeding
 * construct_masked_basis and compress_masked_regions): delta_ptrs name the original full-size tensors,
 * index_ptrs is a device table [n_params] of int32 lists -- the ascending flat positions of the selected
 * elements of each parameter (svdq_maskset_indices) -- and rows_dev[p] their number (<= the plan's rows[p]).
 * Artifacts are those of svdq_compress on the compacted tensors, bit for bit. */
int svdq_compress_gather(const svdq_plan *plan, const void *delta_ptrs, const void *index_ptrs,
                         const int64_t *rows_dev, void *workspace_dev, void *small_dev, void *basis_dev,
                         float *mean_dev, void *stream);

/* svdq_compress straight from checkpoints: finetuned_ptrs [n_params * n_tasks] name the FINE-TUNED tensors,
 * base_ptrs [n_params] the base model's; delta = finetuned - base (compute_task_vector,
 * task_vector_loader.py:103-141) is formed in registers inside the streaming passes, so Step 1's task vectors
 * (cli.py:241-262) are never written to or read back from HBM.  Artifacts are those of svdq_ingest followed by
 * svdq_compress, bit for bit. */
int svdq_compress_from_base(const svdq_plan *plan, const void *finetuned_ptrs, const void *base_ptrs,
                            const int64_t *rows_dev, void *workspace_dev, void *small_dev, void *basis_dev,
                            float *mean_dev, void *stream);
/*      svdq_compress_gather_from_base: both at once -- masked parameters straight from checkpoints (cli.py Step 1 +
 *      the flat[mask] selections of Steps 4 and 5): finetuned[idx] - base[idx] is formed in registers.  Bit-identical to
 *      svdq_ingest followed by svdq_compress_gather. */
int svdq_compress_gather_from_base(const svdq_plan *plan, const void *finetuned_ptrs_dev, const void *base_ptrs_dev,
                                   const void *index_ptrs_dev, const int64_t *rows_dev, void *workspace, void *small,
                                   void *basis, float *mean, void *stream);

/* svdq_compress for MASKED parameters without index lists and without compacted copies (the default for dense masks;
 * the same reference calls as svdq_compress_gather: apply_mask_to_tensor / get_unmasked_portion mask_loader.py:651-709
 * inside the loops of cli.py:324-341 and compress.py:140-155).  delta_ptrs name the original full-size tensors,
 * mask_ptrs [n_params] the combined mask of each parameter as bool bytes (combine_masks mask_loader.py:488-648),
 * unit_start [n_units] the source position of every work unit's first row (svdq_maskset_unit_starts or
 * svdq_maskset_combine_starts; bit 62 set = the unit takes the CLEARED elements, i.e. the noise region), rows_dev[p] the
 * number of selected rows.  Both passes walk the source rows with contiguous loads, read the mask byte beside them and
 * compact the selected rows into the LDS strip: 4 N + 1 bytes per source row and pass against (4 N + 4) per selected row
 * through index lists.  Artifacts are those of svdq_compress on the compacted tensors, bit for bit.  Above 16 tasks both
 * passes run their one-wave kernels (one wavefront per SIMD: measured 20-24 % slower than svdq_compress_gather, whose index
 * lists feed the two-wave kernels; for N = 17..20 bit-identical to the two-wave kernel, plan flag 8).  The _from_base form
 * covers N <= 16 (SVDQ_EUNSUPPORTED above: use svdq_compress_gather_from_base). */
int svdq_compress_masked(const svdq_plan *plan, const void *delta_ptrs, const void *mask_ptrs,
                         const int64_t *unit_start, const int64_t *rows_dev, void *workspace_dev, void *small_dev,
                         void *basis_dev, float *mean_dev, void *stream);
/*      svdq_compress_masked_from_base: the same straight from checkpoints (finetuned[row] - base[row] formed in
 *      registers, compute_task_vector task_vector_loader.py:103-141).  Bit-identical to svdq_ingest + svdq_compress_masked. */
int svdq_compress_masked_from_base(const svdq_plan *plan, const void *finetuned_ptrs_dev, const void *base_ptrs_dev,
                                   const void *mask_ptrs_dev, const int64_t *unit_start, const int64_t *rows_dev,
                                   void *workspace, void *small, void *basis, float *mean, void *stream);

/* ---- the step before the path: task-vector ingest and whole-tensor quantization ("TVQ"),
 * batched over a plan's parameters x tasks.  All pointer tables are DEVICE arrays of device addresses,
 * parameter-major ([p * n_tasks + t]); fp32 buffers 16-byte aligned, code buffers 4-byte aligned. ---- */

/* delta[p][t] = finetuned[p][t] - base[p] in one pass, base read once for the N tasks.
 * Replaces compute_task_vector (src/svd_hybrid/task_vector_loader.py:103-141) and the delta loop of
 * TaskVector.__init__ (task_vectors.py:98-, "delta = finetuned_param - pretrained_param").
 * base_ptrs: [n_params].  stats_work: NULL, or svdq_tvq_work_bytes() bytes that receive the per-unit
 * min/max of every delta so that svdq_tvq_quantize(..., stats_ready = 1) skips its statistics pass. */
int svdq_ingest(const svdq_plan *plan, const void *base_ptrs, const void *finetuned_ptrs, const void *delta_ptrs,
                void *stats_work, void *stream);

int64_t svdq_tvq_work_bytes(const svdq_plan *plan);

/* Whole-tensor quantization of every (parameter, task) tensor of the plan.
 * mode 0: asymmetric_quantization (quantization_utils.py:76-99) -> uint8 codes, scale, zero_point;
 * mode 1: absmax_quantization (quantization_utils.py:60-73)     -> int8 codes, scale.
 * As used by QuantizedFinetunedModel / QuantizedBaseAndTaskVector (task_vectors.py:764-1010).
 * scale / zero_point: device float [n_params * n_tasks].  bits in [1, 8] (absmax: [2, 8], or 16 -> int16 codes,
 * code buffers then 8-byte aligned; dequantize those with mode 2). */
int svdq_tvq_quantize(const svdq_plan *plan, const void *x_ptrs, int32_t mode, int32_t bits, const void *code_ptrs,
                      float *scale, float *zero_point, void *work, int32_t stats_ready, void *stream);

/* dequantize_asymmetric (quantization_utils.py:137-172) / dequantize_absmax (:102-134, which MULTIPLIES by
 * the scale -- kept as is), optionally fused with "+ add[p]" (QuantizedTaskVector.apply_to,
 * task_vectors.py:638-761; QuantizedBaseAndTaskVector.dequantize).  add_ptrs: NULL or [n_params]. */
int svdq_tvq_dequantize(const svdq_plan *plan, const void *code_ptrs, int32_t mode, const float *scale,
                        const float *zero_point, const void *add_ptrs, const void *out_ptrs, void *stream);

/* ---- all four stages back to back on one stream (cli.py Step 4 + Step 5 for the batch) ---- */
int svdq_compress(const svdq_plan *plan, const void *delta_ptrs_dev, const int64_t *rows_dev,
                  void *workspace_dev, void *small_dev, void *basis_dev, float *mean_dev, void *stream);

/* ---- standalone quantizer on one large tensor  (RTVQQuantizer.quantize / dequantize
 *      rtvq.py:106-139; asymmetric_quantization rtvq.py:4-27 == quantization_utils.py:76-99).
 *      codes_dev: uint8 [stages][code_stride] (code_stride % 4 == 0, >= n; first n of each
 *      row valid); scale/zp/rnorm_dev: float [stages];
 *      work_dev: svdq_rtvq_work_bytes(n) bytes of scratch. */
int64_t svdq_rtvq_work_bytes(int64_t n);
int svdq_rtvq_quantize(const float *x_dev, int64_t n, int32_t bits, int32_t stages,
                       uint8_t *codes_dev, int64_t code_stride, float *scale_dev, float *zp_dev,
                       float *rnorm_dev, void *work_dev, void *stream);
int svdq_rtvq_dequantize(const uint8_t *codes_dev, int64_t code_stride, int64_t n, int32_t stages,
                         const float *scale_dev, const float *zp_dev, float *out_dev, void *stream);

/*      qbit = 16 (asymmetric_quantization rtvq.py:22-25 only; SVDHybridConfig allows at most 8 bits): one stage, int16
 *      codes exactly as the reference's CPU cast leaves them (values above 32767 wrap negative), and the matching
 *      asymmetric_dequantization on int16 input.  work_dev: svdq_rtvq_work_bytes(n) bytes. */
int svdq_asym16_quantize(const float *x_dev, int64_t n, int16_t *codes_dev, float *scale_dev, float *zp_dev,
                         void *work_dev, void *stream);
int svdq_asym16_dequantize(const int16_t *codes_dev, int64_t n, const float *scale_dev, const float *zp_dev,
                           float *out_dev, void *stream);

/* ---- standalone projection for callers that bring their own basis
 *      (project_to_basis compress.py:6-21 with the mean subtraction of compress.py:35-37):
 *      c_out[i] = sum_d float(U[d][i]) * (delta[d] - mean[d]),  i < k from U_high [rows,k],
 *      then i < k+nl from U_low [rows,nl]; both row-major, fp16 (u_fp16 != 0) or fp32;
 *      mean may be NULL; k + nl <= 32.  work_dev: svdq_project_work_bytes(rows, k+nl) bytes. */
int64_t svdq_project_work_bytes(int64_t rows, int32_t ncols);
int svdq_project(const void *u_high_dev, const void *u_low_dev, int32_t u_fp16, int64_t rows,
                 int32_t k, int32_t nl, const float *delta_dev, const float *mean_dev,
                 float *c_out_dev, void *work_dev, void *stream);

/* ---- masks  (compute_union/intersection/majority_mask mask_loader.py:412-485;
 *      apply_mask_to_tensor / get_unmasked_portion mask_loader.py:651-709).
 *      mask_ptrs_dev: device array [n_masks] of const uint8_t* (torch.bool storage).
 *      svdq_mask_combine writes the combined mask (0/1 bytes) and its popcount.  strategy: SVDQ_MASK_*; for
 *      SVDQ_MASK_MAJORITY bits 8.. may carry (votes needed + 1) for compute_majority_mask(threshold != 0.5)
 *      (mask_loader.py:456-485: vote_sum >= threshold * len(masks), compared in fp32); 0 there = threshold 0.5.
 *      svdq_mask_compact: order-preserving compaction of n_src fp32 buffers under one mask
 *      (invert != 0 selects the False positions); count_dev receives the selected count.
 *      work_dev: svdq_mask_work_bytes(numel) bytes. */
int64_t svdq_mask_work_bytes(int64_t numel);
int svdq_mask_combine(const void *mask_ptrs_dev, int32_t n_masks, int64_t numel, int32_t strategy,
                      uint8_t *out_dev, int64_t *count_dev, void *work_dev, void *stream);
int svdq_mask_compact(const void *src_ptrs_dev, const void *dst_ptrs_dev, int32_t n_src,
                      const uint8_t *mask_dev, int32_t invert, int64_t numel, int64_t *count_dev,
                      void *work_dev, void *stream);

/* ---- the mask operators over a ragged SET of parameters in a handful of launches (the reference runs
 *      them once per parameter and task: cli.py:324-341, compress.py:140-155).  numel[q] = elements of
 *      masked parameter q.  Tables are device arrays of device pointers:
 *        mask_ptrs [Q*n_masks] uint8 per-task masks (combine) / [Q] combined masks (compact)
 *        out_ptrs  [Q] combined mask outputs;  src/dst [Q*n_src] fp32 buffers (dst sized numel[q])
 *      counts stay on the device (int64 [Q]): mask.sum() and (~mask).sum(); they feed rows_dev.
 *      svdq_maskset_compact writes flat[mask] to dst_true and, when dst_false_ptrs != NULL,
 *      flat[~mask] to dst_false in the same pass.  work_dev: svdq_maskset_work_bytes() bytes. */
typedef struct svdq_maskset svdq_maskset;
int     svdq_maskset_create(svdq_maskset **ms, int32_t n_params, const int64_t *numel);
void    svdq_maskset_destroy(svdq_maskset *ms);
int64_t svdq_maskset_work_bytes(const svdq_maskset *ms);
int     svdq_maskset_combine(const svdq_maskset *ms, const void *mask_ptrs_dev, int32_t n_masks, int32_t strategy,
                             const void *out_ptrs_dev, int64_t *counts_dev, void *stream);
int     svdq_maskset_compact(const svdq_maskset *ms, const void *mask_ptrs_dev, const void *src_ptrs_dev,
                             const void *dst_true_ptrs_dev, const void *dst_false_ptrs_dev, int32_t n_src,
                             int64_t *count_true_dev, int64_t *count_false_dev, void *work_dev, void *stream);

/* Index lists instead of compacted copies: idx_true_ptrs[q] (int32, room for numel[q] entries, 16-byte aligned)
 * receives the ascending flat positions of the set elements of mask q, idx_false_ptrs[q] (NULL table = skip)
 * those of the cleared ones; counts as in svdq_maskset_compact.  This is what `flat[mask]` / `flat[~mask]`
 * (mask_loader.py:651-709) select; svdq_compress_gather reads the task deltas through these lists, so the
 * 2*N compacted copies per parameter never exist. */
int     svdq_maskset_indices(const svdq_maskset *ms, const void *mask_ptrs_dev, const void *idx_true_ptrs_dev,
                             const void *idx_false_ptrs_dev, int64_t *count_true_dev, int64_t *count_false_dev,
                             void *work_dev, void *stream);

/* svdq_maskset_combine followed by svdq_maskset_indices on the combined masks in 3 launches: the combine pass
 * already counts every tile, so the separate counting pass over the combined masks is skipped.
 * (combine_masks mask_loader.py:488-648 feeding the flat[mask] selections of cli.py:333 / compress.py:144.) */
int     svdq_maskset_combine_indices(const svdq_maskset *ms, const void *mask_ptrs_dev, int32_t n_masks,
                                     int32_t strategy, const void *out_ptrs_dev, const void *idx_true_ptrs_dev,
                                     const void *idx_false_ptrs_dev, int64_t *count_true_dev,
                                     int64_t *count_false_dev, void *work_dev, void *stream);

/* The same from BIT-PACKED tall masks, the form the reference's mask files have (load_tall_mask_file,
 * mask_loader.py:125-206: one numpy.packbits stream per task over the flattened state dict, first element = most
 * significant bit): stream_ptrs_dev [n_masks] device streams, stream_bytes_dev [n_masks] their lengths,
 * bit_offsets_dev [n_params] the bit position of each parameter's first element inside every stream.  Reads
 * n_masks / 8 bytes per element instead of n_masks; writes the combined bool masks, index lists and counts. */
int     svdq_maskset_combine_packed_indices(const svdq_maskset *ms, const void *stream_ptrs_dev,
                                            const int64_t *stream_bytes_dev, const int64_t *bit_offsets_dev,
                                            int32_t n_masks, int32_t strategy, const void *out_ptrs_dev,
                                            const void *idx_true_ptrs_dev, const void *idx_false_ptrs_dev,
                                            int64_t *count_true_dev, int64_t *count_false_dev, void *work_dev,
                                            void *stream);

/* Unit starts for svdq_compress_masked instead of index lists (same reference calls: mask.sum() and flat[mask] of
 * cli.py:332-333 / compress.py:143-144, mask_loader.py:651-709).
 *   svdq_maskset_count_scan: per-tile counts + per-parameter exclusive scan of already combined masks (2 launches);
 *     counts as in svdq_maskset_compact; the tile offsets stay in work_dev for svdq_maskset_unit_starts.
 *   svdq_maskset_unit_starts: unit_start_dev[u] (int64 [plan n_units]) = source position of the element of rank
 *     unit.row0 among the selected elements of the unit's mask, numel when the unit lies past rows_dev[p].
 *     entry_map_dev: NULL (plan parameter p uses mask p; the plan's rows must equal the set's numel) or int32
 *     [plan n_params]: low 31 bits = index of the mask in the set, bit 31 = take the CLEARED elements (the noise
 *     region of cli.py:336-338).  rows_dev [plan n_params] as handed to svdq_compress_masked.
 *   svdq_maskset_combine_starts / _combine_packed_starts: combine (bool-byte / bit-packed per-task masks, as
 *     svdq_maskset_combine_indices / _combine_packed_indices) + scan + unit starts in 3 launches, identity map,
 *     rows = count_true. */
int     svdq_maskset_count_scan(const svdq_maskset *ms, const void *mask_ptrs_dev, int64_t *count_true_dev,
                                int64_t *count_false_dev, void *work_dev, void *stream);
int     svdq_maskset_unit_starts(const svdq_maskset *ms, const svdq_plan *plan, const void *mask_ptrs_dev,
                                 const int32_t *entry_map_dev, const int64_t *rows_dev, const void *work_dev,
                                 int64_t *unit_start_dev, void *stream);
int     svdq_maskset_combine_starts(const svdq_maskset *ms, const svdq_plan *plan, const void *mask_ptrs_dev,
                                    int32_t n_masks, int32_t strategy, const void *out_ptrs_dev,
                                    int64_t *count_true_dev, int64_t *count_false_dev, void *work_dev,
                                    int64_t *unit_start_dev, void *stream);
int     svdq_maskset_combine_packed_starts(const svdq_maskset *ms, const svdq_plan *plan, const void *stream_ptrs_dev,
                                           const int64_t *stream_bytes_dev, const int64_t *bit_offsets_dev,
                                           int32_t n_masks, int32_t strategy, const void *out_ptrs_dev,
                                           int64_t *count_true_dev, int64_t *count_false_dev, void *work_dev,
                                           int64_t *unit_start_dev, void *stream);

/* ---- merge consumers (the parity reconstruction of R14)
 *      svdq_reconstruct: reconstruct_from_coefficients (merge.py:144-194):
 *        out[d] = ((sum_i U_high[d][i] c[i] + sum_j U_low[d][j] c[k+j]) + mean[d]) * scale
 *        coef_dev: float [k+nl] on the device; mean_dev may be NULL; scale = noise_shrink or 1.
 *      svdq_mask_expand: reconstruct_from_masked (mask_loader.py:712-763): out = 0; out[mask] = signal;
 *        out[~mask] = noise (noise_dev may be NULL).  work_dev: svdq_mask_work_bytes(numel) bytes. */
int svdq_reconstruct(const void *u_high_dev, const void *u_low_dev, int32_t u_fp16, int64_t rows, int32_t k,
                     int32_t nl, const float *coef_dev, const float *mean_dev, float scale, float *out_dev,
                     void *stream);
int svdq_mask_expand(const float *signal_dev, const float *noise_dev, const uint8_t *mask_dev, int64_t numel,
                     float *out_dev, void *work_dev, void *stream);

/* ---- diagnostics: compute_reconstruction_error (diagnostics.py:72-117).
 *      recon_dev != NULL: metrics of (orig - recon) for two given vectors.
 *      recon_dev == NULL: the reconstruction U_high c_high + U_low c_low (+ mean_dev if not NULL; the
 *      reference's compute_parameter_diagnostics passes none: diagnostics.py:210-212) is formed
 *      on the fly and never stored.  out6_dev (double[6]) = absolute_error, relative_error,
 *      max_absolute_error, mean_absolute_error, original_norm, reconstructed_norm.
 *      work_dev: svdq_recon_error_work_bytes(rows) bytes. */
int64_t svdq_recon_error_work_bytes(int64_t rows);
int svdq_recon_error(const void *u_high_dev, const void *u_low_dev, int32_t u_fp16, int64_t rows, int32_t k,
                     int32_t nl, const float *coef_dev, const float *mean_dev, const float *recon_dev,
                     const float *orig_dev, double *out6_dev, void *work_dev, void *stream);

/* ---- the same consumers for a WHOLE PLAN, straight from the buffers the compressor left in HBM (no host copy of the
 *      payloads, no per-parameter / per-task launch): what cli.py Steps 7-9 do parameter by parameter --
 *      merge_all_parameters (merge.py:304-426) -> merge_parameter (:197-301) -> dequantize_and_average (:61-141, with
 *      RTVQQuantizer.dequantize rtvq.py:85-103) + reconstruct_from_coefficients (:144-194); merge_with_clustering
 *      (:555-626) + merge_cluster_results (clustering.py:374-425); apply_merged_deltas (merge.py:429-552);
 *      compute_parameter_diagnostics (diagnostics.py:120-231).
 *   A "set" is a group of tasks averaged together: one set of all tasks (merge_all_parameters), one per cluster
 *   (merge_with_clustering) or one per task (diagnostics).
 *   weights_dev: float [n_sets][N] (per_param = 0) or [P][n_sets][N] (per_param = 1): weight of task t in set s,
 *     renormalised over the set's present tasks as the reference does on the host (merge.py:123-124); < 0 = not in the set.
 *   order_dev:   NULL or int32 [N] / [P][N]: task indices in the order of the sorted task names (merge.py:89).
 *   svdq_merge_coeffs: cbar_dev float [P][n_sets][N] = averaged c_high (columns < k) and dequantized c_low (k..r-1).
 *   svdq_merge_reconstruct: one streaming launch over the plan's units,
 *       out[p][d] = sum_s share[s] * (((U_high c_s,high + U_low c_s,low)[d] + mean[d]) * scale[p])  (+ base[p][d])
 *     set_share_dev: NULL (n_sets = 1: no weighting) or float [n_sets] / [P][n_sets]: the clusters' shares as
 *     apply_weights_to_tensors leaves them (weighting.py:332-372: shares / shares.sum() over ALL clusters -- in
 *     merge_with_clustering a cluster none of whose members holds the parameter still contributes its zeros,
 *     merge.py:297-299,555-626); < 0 = the set does not hold the parameter: skipped, the others are NOT renormalised;
 *     n_sets <= 8.
 *     scale_dev: NULL or float [P] (noise_shrink of the noise regions, merge.py:284); base_ptrs_dev: NULL or [P] base
 *     tensors (then out = base + delta); out_ptrs_dev [P] fp32 outputs of rows[p] elements (compacted rows for masked
 *     parameters: svdq_mask_expand scatters them).  A parameter's rows are the bits svdq_reconstruct gives.
 *   svdq_merge = both; work_dev: svdq_merge_work_bytes(plan, n_sets).
 *   svdq_diagnostics: out_dev double [P][N][6], the six numbers of svdq_recon_error for every (parameter, task) from
 *     one pass over U and the N deltas (delta_ptrs_dev as for svdq_compress); add_mean = 0 reproduces the reference
 *     (SURVEY Q1).  The reconstruction U_high c_high + U_low c_low is formed in fp32 (matrix pipe), the sums in fp64: every
 *     number is within the fp32 forward-error bound of diagnostics.py:210-212 of the formula evaluated in fp64 on the same
 *     artifacts (tests/test_hip_diagnostics.py).  work_dev: svdq_diagnostics_work_bytes(plan). */
int64_t svdq_merge_work_bytes(const svdq_plan *plan, int32_t n_sets);
int svdq_merge_coeffs(const svdq_plan *plan, const void *small_dev, const float *weights_dev, const int32_t *order_dev,
                      int32_t n_sets, int32_t per_param, float *cbar_dev, void *stream);
int svdq_merge_reconstruct(const svdq_plan *plan, const int64_t *rows_dev, const void *small_dev, const void *basis_dev,
                           const float *mean_dev, const float *cbar_dev, int32_t n_sets, int32_t per_param,
                           const float *set_share_dev, const float *scale_dev, const void *base_ptrs_dev,
                           const void *out_ptrs_dev, void *stream);
int svdq_merge(const svdq_plan *plan, const int64_t *rows_dev, const void *small_dev, const void *basis_dev,
               const float *mean_dev, const float *weights_dev, const int32_t *order_dev, int32_t n_sets,
               int32_t per_param, const float *set_share_dev, const float *scale_dev, const void *base_ptrs_dev,
               const void *out_ptrs_dev, void *work_dev, void *stream);
int64_t svdq_diagnostics_work_bytes(const svdq_plan *plan);
int svdq_diagnostics(const svdq_plan *plan, const void *delta_ptrs_dev, const int64_t *rows_dev, const void *small_dev,
                     const void *basis_dev, const float *mean_dev, int32_t add_mean, double *out_dev, void *work_dev,
                     void *stream);

/* ---- the same for MASKED regions, with reconstruct_from_masked (mask_loader.py:712-763: zeros; result[mask] = signal;
 *      result[~mask] = noise) and apply_mask_to_tensor (:651-679, the "original" of the masked diagnostics,
 *      diagnostics.py:186-199) inside the streaming launch: the plan's artifacts describe the COMPACTED rows of every
 *      region; the launch walks the SOURCE rows with the combined mask byte beside them (mask_ptrs_dev [P], unit_start_dev
 *      [plan units] from svdq_maskset_unit_starts / *_starts; bit 62 of a start = the region is the CLEARED elements),
 *      finds each selected row's compacted row by wave ballots and reads the contiguous run of basis rows it needs.
 *   svdq_merge_masked: out_ptrs_dev [P] = FULL tensors (params[p].rows elements; NULL = leave this entry alone); every
 *     selected source row gets the merged value of its compacted row (+ base at the source row); fill_dev NULL or int32
 *     [P]: 1 = the entry also writes 0 (+ base) to the rows its region does not select -- for a signal region without a
 *     noise region; with one, the noise entry (its own plan entry, inverted polarity, scale = noise_shrink) writes
 *     them.  Same bits as svdq_merge + svdq_mask_expand (+ base).
 *   svdq_diagnostics_masked: delta_ptrs_dev = the N unmasked task deltas per entry. */
int svdq_merge_masked(const svdq_plan *plan, const int64_t *rows_dev, const void *small_dev, const void *basis_dev,
                      const float *mean_dev, const float *weights_dev, const int32_t *order_dev, int32_t n_sets,
                      int32_t per_param, const float *set_share_dev, const float *scale_dev, const void *mask_ptrs_dev,
                      const int64_t *unit_start_dev, const int32_t *fill_dev, const void *base_ptrs_dev,
                      const void *out_ptrs_dev, void *work_dev, void *stream);
int svdq_diagnostics_masked(const svdq_plan *plan, const void *delta_ptrs_dev, const void *mask_ptrs_dev,
                            const int64_t *unit_start_dev, const int64_t *rows_dev, const void *small_dev,
                            const void *basis_dev, const float *mean_dev, int32_t add_mean, double *out_dev,
                            void *work_dev, void *stream);

/* ---- measurement aid (no reference counterpart; SURVEY.md section 8d asks for "a measured device-copy ceiling on
 *      the box" beside the 8 TB/s specification): plain streaming kernels with the access shape of the two passes.
 *      mode 0: read `bytes` from src_dev (dst_dev receives one float per 256 KiB read); mode 1: copy `bytes`;
 *      mode 2: read `bytes`, write 5/8 of that (pass 2's read : write mix at N = 8).  dst_dev must hold `bytes`. */
int svdq_hbm_probe(int32_t mode, const void *src_dev, void *dst_dev, int64_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVDQ_H */

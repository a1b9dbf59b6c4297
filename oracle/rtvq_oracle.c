/*
 * oracle/rtvq_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's multi-stage residual affine (min/max)
 * quantizer ("RTVQ").  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path never does.
 *
 * Reference lines restated (paths relative to /root/reference):
 *   src/svd_hybrid/rtvq.py:4-27    asymmetric_quantization   (== quantization_utils.py:76-99)
 *   src/svd_hybrid/rtvq.py:29-36   asymmetric_dequantization (== quantization_utils.py:137-172)
 *   src/svd_hybrid/rtvq.py:39-82   multistage_residual_quantization
 *   src/svd_hybrid/rtvq.py:85-103  multistage_residual_dequantization
 *
 * Arithmetic contract (what torch's CPU kernels do, one IEEE fp32 rounding per op):
 *   mn, mx        = min / max over x, NaN-propagating
 *   scale         = (1 / (mx - mn)) * (2^b - 1)         TWO roundings: python-int / Tensor is
 *                                                        Tensor.__rtruediv__ = reciprocal() * int
 *                                                        (pinned by tests/golden/rtvq_cases.npz);
 *                                                        no epsilon guard
 *   zero_point    = -1 * rint(scale * mn)               rint = round-half-to-even
 *   q             = clamp(rint(scale * x + zp), 0, 2^b-1)   mul and add rounded separately (no FMA)
 *   deq           = (float(q) - zp) / scale             true fp32 division
 *   residual     -= deq
 *   residual_norm = ||residual||_2 before the stage     (summation order is not pinned: rtol)
 * Degenerate input (mx == mn, one element, all equal) gives scale = inf and NaN
 * downstream exactly as the reference does; NaN converts to code 0.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * Pinned by tests/test_oracle_golden.py against vectors produced by the reference itself.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static float f32_min_nanprop(const float *x, size_t n) {
    float m = x[0];
    for (size_t i = 0; i < n; ++i) {
        if (isnan(x[i])) return x[i];
        if (x[i] < m) m = x[i];
    }
    return m;
}

static float f32_max_nanprop(const float *x, size_t n) {
    float m = x[0];
    for (size_t i = 0; i < n; ++i) {
        if (isnan(x[i])) return x[i];
        if (x[i] > m) m = x[i];
    }
    return m;
}

static uint8_t code_from_float(float v, float qmax) {
    /* clamp propagates NaN; the NaN -> integer cast lands on 0 on the reference's CPU path */
    if (isnan(v)) return 0;
    if (v < 0.0f) v = 0.0f;
    if (v > qmax) v = qmax;
    return (uint8_t)v;
}

/* rtvq.py:4-27.  n >= 1.  Returns scale/zero_point through pointers. */
void oracle_asym_quantize(const float *x, size_t n, int bits,
                          uint8_t *q, float *scale_out, float *zp_out) {
    volatile float mn = f32_min_nanprop(x, n);
    volatile float mx = f32_max_nanprop(x, n);
    const float qmax = (float)((1 << bits) - 1);
    volatile float range = mx - mn;
    volatile float recip = 1.0f / range;
    volatile float scale = recip * qmax;
    volatile float smn = scale * mn;
    volatile float zp = -1.0f * rintf(smn);
    for (size_t i = 0; i < n; ++i) {
        volatile float a = scale * x[i];
        volatile float b = a + zp;
        q[i] = code_from_float(rintf(b), qmax);
    }
    *scale_out = scale;
    *zp_out = zp;
}

/* rtvq.py:29-36 */
void oracle_asym_dequantize(const uint8_t *q, size_t n, float scale, float zp, float *out) {
    for (size_t i = 0; i < n; ++i) {
        volatile float d = (float)q[i] - zp;
        out[i] = d / scale;
    }
}

/*
 * rtvq.py:39-82.  codes is [stages][n], scale/zp/norm are [stages].
 * work is a caller-provided scratch of n floats (the running residual).
 * n == 0 is the caller's job (the reference returns an empty payload list).
 */
void oracle_rtvq_quantize(const float *x, size_t n, int bits, int stages,
                          uint8_t *codes, float *scale, float *zp, float *norm,
                          float *work) {
    for (size_t i = 0; i < n; ++i) work[i] = x[i];
    for (int s = 0; s < stages; ++s) {
        double ss = 0.0;
        for (size_t i = 0; i < n; ++i) ss += (double)work[i] * (double)work[i];
        norm[s] = (float)sqrt(ss);
        oracle_asym_quantize(work, n, bits, codes + (size_t)s * n, &scale[s], &zp[s]);
        for (size_t i = 0; i < n; ++i) {
            volatile float d = (float)codes[(size_t)s * n + i] - zp[s];
            volatile float deq = d / scale[s];
            work[i] = work[i] - deq;
        }
    }
}

/* rtvq.py:85-103: result = ((0 + deq_0) + deq_1) + ... in stage order */
void oracle_rtvq_dequantize(const uint8_t *codes, size_t n, int stages,
                            const float *scale, const float *zp, float *out) {
    for (size_t i = 0; i < n; ++i) out[i] = 0.0f;
    for (int s = 0; s < stages; ++s) {
        for (size_t i = 0; i < n; ++i) {
            volatile float d = (float)codes[(size_t)s * n + i] - zp[s];
            volatile float deq = d / scale[s];
            out[i] = out[i] + deq;
        }
    }
}

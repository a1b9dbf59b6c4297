// Empirical operand / result layouts of the small-block MFMAs on gfx950 (run on the GPU box):
//   v_mfma_f32_4x4x1_16b_f32   (16 blocks, D_b[4x4] += A_b[4x1] B_b[1x4])
//   v_mfma_f32_4x4x4_16b_f16   (16 blocks, D_b[4x4] += A_b[4x4] B_b[4x4], four halfs per lane and operand)
//   v_mfma_f32_4x4x4_16b_bf16? (bf16_1k form: four bf16 per lane and operand)
// Method: one lane (or one lane and one k) supplies 1.0, everything else 0; print which result registers light up.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe_4x4x1(float *out, int la, int lb) {
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(l == la ? 1.f : 0.f, l == lb ? 1.f : 0.f, c, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}
__global__ void probe_4x4x4_f16(float *out, int la, int ka, int lb, int kb) {
    const int l = threadIdx.x;
    f16x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (l == la) a[ka] = (_Float16)1.f;
    if (l == lb) b[kb] = (_Float16)1.f;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}
__global__ void probe_4x4x4_bf16(float *out, int la, int ka, int lb, int kb) {
    const int l = threadIdx.x;
    s16x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (l == la) a[ka] = (short)0x3f80;   // bf16 1.0
    if (l == lb) b[kb] = (short)0x3f80;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}
// timing: back-to-back dependent / independent issue
template <int WHICH>
__global__ void timing(float *out, long long *cyc) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    const float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x;
    f16x4 ha = {(_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a}, hb = ha;
    s16x4 sa = {1, 2, 3, 4}, sb = sa;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) {
        if (WHICH == 0) { c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
                          c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0); }
        if (WHICH == 1) { c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
                          c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); }
        if (WHICH == 2) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0); }
        if (WHICH == 3) { c0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sa, sb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sa, sb, c1, 0, 0, 0);
                          c2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sa, sb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sa, sb, c3, 0, 0, 0); }
        if (WHICH == 4) { c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c1, 0, 0, 0);
                          c2 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c3, 0, 0, 0); }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = c0 + c1 + c2 + c3;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

static void show(const char *name, const float *h) {
    printf("%s:", name);
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (h[l * 4 + e] != 0.f) printf(" (lane %d reg %d)=%g", l, e, h[l * 4 + e]);
    printf("\n");
}
int main() {
    float *d; hipMalloc(&d, 64 * 4 * 4); float h[256]; long long *dc; hipMalloc(&dc, 8); long long hc;
    // 4x4x1: A lane la, B lane lb -> which block / i / j
    int pairs[][2] = {{0, 0}, {1, 0}, {0, 1}, {2, 3}, {4, 4}, {5, 6}, {4, 0}, {17, 18}, {63, 60}};
    for (auto &p : pairs) {
        hipLaunchKernelGGL(probe_4x4x1, dim3(1), dim3(64), 0, 0, d, p[0], p[1]); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        char nm[64]; sprintf(nm, "4x4x1 A@lane%d B@lane%d", p[0], p[1]); show(nm, h);
    }
    int q[][4] = {{0, 0, 0, 0}, {0, 1, 0, 1}, {0, 1, 0, 0}, {1, 2, 3, 2}, {5, 3, 6, 3}, {4, 0, 0, 0}, {21, 2, 22, 2}};
    for (auto &p : q) {
        hipLaunchKernelGGL(probe_4x4x4_f16, dim3(1), dim3(64), 0, 0, d, p[0], p[1], p[2], p[3]); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        char nm[96]; sprintf(nm, "4x4x4f16 A@lane%d[k%d] B@lane%d[k%d]", p[0], p[1], p[2], p[3]); show(nm, h);
        hipLaunchKernelGGL(probe_4x4x4_bf16, dim3(1), dim3(64), 0, 0, d, p[0], p[1], p[2], p[3]); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        sprintf(nm, "4x4x4bf16_1k A@lane%d[k%d] B@lane%d[k%d]", p[0], p[1], p[2], p[3]); show(nm, h);
    }
    const char *tn[] = {"4x4x1 f32 x4 independent", "4x4x1 f32 x4 dependent", "16x16x4 f32 x4 independent", "4x4x4 bf16_1k x4 independent", "4x4x4 f16 x4 independent"};
    for (int w = 0; w < 5; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            if (w == 0) hipLaunchKernelGGL(timing<0>, dim3(1), dim3(64), 0, 0, d, dc);
            if (w == 1) hipLaunchKernelGGL(timing<1>, dim3(1), dim3(64), 0, 0, d, dc);
            if (w == 2) hipLaunchKernelGGL(timing<2>, dim3(1), dim3(64), 0, 0, d, dc);
            if (w == 3) hipLaunchKernelGGL(timing<3>, dim3(1), dim3(64), 0, 0, d, dc);
            if (w == 4) hipLaunchKernelGGL(timing<4>, dim3(1), dim3(64), 0, 0, d, dc);
            hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
        }
        printf("%s: %.1f s_memtime ticks per MFMA (1024 MFMAs)\n", tn[w], (double)hc / 1024.0);
    }
    return 0;
}

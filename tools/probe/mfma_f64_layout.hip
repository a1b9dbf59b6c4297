// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result layout and issue time (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out, int la, int lb) {
    const int l = threadIdx.x;
    double c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, c, 0, 0, 0);
    out[l] = c;
}
template <int WHICH>
__global__ void timing(double *out, long long *cyc) {
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    f64x4 d0 = {0, 0, 0, 0}, d1 = d0;
    const double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) {
        if (WHICH == 0) { c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
                          c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0); }
        if (WHICH == 1) { c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                          c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); }
        if (WHICH == 2) { d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d1, 0, 0, 0);
                          d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d1, 0, 0, 0); }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0 + c1 + c2 + c3 + d0[0] + d1[1];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; hipMalloc(&d, 64 * 8); double h[64]; long long *dc; hipMalloc(&dc, 8); long long hc;
    int pairs[][2] = {{0, 0}, {1, 0}, {0, 1}, {4, 0}, {0, 4}, {4, 4}, {5, 6}, {8, 4}, {12, 12}, {16, 16}, {17, 22}, {20, 16}, {63, 60}, {3, 12}, {12, 3}};
    for (auto &p : pairs) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p[0], p[1]); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("f64 4x4x4 A@lane%d B@lane%d:", p[0], p[1]);
        for (int l = 0; l < 64; ++l) if (h[l] != 0.0) printf(" lane %d=%g", l, h[l]);
        printf("\n");
    }
    const char *tn[] = {"f64 4x4x4_4b x4 independent", "f64 4x4x4_4b x4 dependent", "f64 16x16x4 x4 (2 chains)"};
    for (int w = 0; w < 3; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            if (w == 0) hipLaunchKernelGGL(timing<0>, dim3(1), dim3(64), 0, 0, d, dc);
            if (w == 1) hipLaunchKernelGGL(timing<1>, dim3(1), dim3(64), 0, 0, d, dc);
            if (w == 2) hipLaunchKernelGGL(timing<2>, dim3(1), dim3(64), 0, 0, d, dc);
            hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
        }
        printf("%s: %.1f ticks per MFMA\n", tn[w], (double)hc / 1024.0);
    }
    return 0;
}

// eig_time.hip -- where the per-parameter eigen-stage (csrc/svdq_eig.h: eig_param) spends its time.
// Build and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I svd-quantization-task-merging_amd/csrc tools/probe/eig_time.hip -o /tmp/eig_time && /tmp/eig_time
// One wavefront (N <= 8) or 256 threads (N > 8) per parameter, P parameters in one launch like k_eig; s_memtime stamps
// at the phase boundaries of eig_param (EIG_STAMP) land in a buffer of their own; the median parameter is printed.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long *g_stamps;   // [P][16]
__device__ int g_sweeps_dummy;
#define EIG_STAMP(i)                                                                         \
    do {                                                                                     \
        if (tid == 0) g_stamps[(size_t)p * 16 + (i)] = __builtin_amdgcn_s_memtime();         \
    } while (0)
#include "svdq_eig.h"

void svdq_set_error(const char *, ...) {}

template <int THREADS, int NMAX>
__global__ __launch_bounds__(THREADS) void k_probe_eig(const float *const *ptrs, int NT, const double *part2, float *W,
                                                       double *c0, float *sigma, int32_t *k, int32_t *r, float *en,
                                                       int64_t *rows, int64_t D) {
    __shared__ __attribute__((aligned(16))) double lds[SVDQ_EIG_LDS_BYTES(NMAX) / 8 + 1];
    eig_param<THREADS, NMAX>(lds, blockIdx.x, threadIdx.x, D, ptrs, NT, 1, 0.9f, 64, part2, W, c0, sigma, k, r, en,
                             rows, 0, nullptr, nullptr, 1e-6);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8, P = argc > 2 ? atoi(argv[2]) : 296;
    const int64_t D = 1 << 20;
    const int nn = N * N;
    // Gram partials of a decaying spectrum: G = B diag(s) B^T split over SVDQ_RC chunks
    std::vector<double> part((size_t)P * SVDQ_RC * nn);
    srand(1);
    for (int p = 0; p < P; ++p) {
        std::vector<double> B(N * N);
        for (auto &x : B) x = (rand() / (double)RAND_MAX - 0.5);
        for (int c = 0; c < SVDQ_RC; ++c)
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    double g = 0;
                    for (int q = 0; q < N; ++q) g += B[i * N + q] * B[j * N + q] * (q < 3 ? 1.0 / (1 << (2 * q)) : 1e-3);
                    part[((size_t)p * SVDQ_RC + c) * nn + i * N + j] = g / SVDQ_RC;
                }
    }
    double *d_part;
    float *d_W, *d_sigma, *d_en, *d_rows0;
    double *d_c0;
    int32_t *d_k, *d_r;
    int64_t *d_rows;
    unsigned long long *d_st;
    const float **d_ptrs;
    hipMalloc(&d_part, part.size() * 8);
    hipMemcpy(d_part, part.data(), part.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&d_W, (size_t)P * (nn + 4) * 4);
    hipMalloc(&d_c0, (size_t)P * nn * 8);
    hipMalloc(&d_sigma, (size_t)P * N * 4);
    hipMalloc(&d_en, P * 4);
    hipMalloc(&d_k, P * 4);
    hipMalloc(&d_r, P * 4);
    hipMalloc(&d_rows, P * 8);
    hipMalloc(&d_st, (size_t)P * 16 * 8);
    hipMalloc(&d_rows0, 4096);
    hipMemset(d_rows0, 0, 4096);
    std::vector<const float *> hp((size_t)P * N, d_rows0);
    hipMalloc(&d_ptrs, hp.size() * 8);
    hipMemcpy(d_ptrs, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (N <= 8)
            hipLaunchKernelGGL((k_probe_eig<64, 8>), dim3(P), dim3(64), 0, 0, d_ptrs, N, d_part, d_W, d_c0, d_sigma, d_k,
                               d_r, d_en, d_rows, D);
        else
            hipLaunchKernelGGL((k_probe_eig<256, 32>), dim3(P), dim3(256), 0, 0, d_ptrs, N, d_part, d_W, d_c0, d_sigma,
                               d_k, d_r, d_en, d_rows, D);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> st((size_t)P * 16);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    const char *names[8] = {"sum partials", "deflate", "jacobi", "sort+sign", "rank rule", "W + row0 load", "completion", "W/c0 out"};
    printf("N=%d P=%d  kernel %.1f us (last of 5 launches)\n", N, P, ms * 1e3);
    for (int i = 0; i < 8; ++i) {
        std::vector<double> d;
        for (int p = 0; p < P; ++p) d.push_back((double)(st[(size_t)p * 16 + i + 1] - st[(size_t)p * 16 + i]));
        std::sort(d.begin(), d.end());
        printf("  %-16s median %8.0f cycles (%.2f us at 100 MHz ticks -> see note)  max %8.0f\n", names[i], d[P / 2],
               d[P / 2] / 100.0, d[P - 1]);
    }
    std::vector<double> tot;
    for (int p = 0; p < P; ++p) tot.push_back((double)(st[(size_t)p * 16 + 8] - st[(size_t)p * 16]));
    std::sort(tot.begin(), tot.end());
    printf("  total            median %8.0f ticks  max %8.0f   (s_memtime ticks; the launch time above calibrates them)\n",
           tot[P / 2], tot[P - 1]);
    return 0;
}

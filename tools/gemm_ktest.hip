// Experiment: how the forward GEMM's rate depends on the length of its K loop and on its epilogue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../streamz_amd/csrc -o gemm_ktest gemm_ktest.hip
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm_kernel.h"

using namespace sz::gemm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

template <int AM, int EPI, int TM>
float run(GemmArgs g, int iters)
{
    dim3 grid((g.N + BN - 1) / BN, (g.M + TM - 1) / TM, 1);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_kernel<AM, B_KROWS, EPI, 0, TM, 32, 0>), grid, dim3(kGemmThreads), 0, 0, g);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_kernel<AM, B_KROWS, EPI, 0, TM, 32, 0>), grid, dim3(kGemmThreads), 0, 0, g);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.0f / iters;
}

int main()
{
    const int M = 393216, N = 256, KMAX = 2048;
    float *A, *B, *C, *bias;
    CK(hipMalloc((void**)&A, (size_t)M * KMAX * 4));
    CK(hipMalloc((void**)&B, (size_t)KMAX * N * 4));
    CK(hipMalloc((void**)&C, (size_t)M * N * 4));
    CK(hipMalloc((void**)&bias, N * 4));
    CK(hipMemset(A, 0, (size_t)M * KMAX * 4));
    CK(hipMemset(B, 0, (size_t)KMAX * N * 4));
    CK(hipMemset(bias, 0, N * 4));
    for (int K : {128, 256, 512, 1024, 2048}) {
        GemmArgs g{};
        g.A = A; g.lda = K; g.B = B; g.ldb = N; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.k_chunk = K; g.bias = bias;
        g.a_vec = 1; g.b_vec = 1;
        const double fl = 2.0 * M * N * K;
        const float t_tanh = run<A_ROWS, EPI_BIAS_TANH, 128>(g, 10);
        const float t_bias = run<A_ROWS, EPI_BIAS, 128>(g, 10);
        GemmArgs h = g;
        h.lda = M;  // A stored [K][M]
        const float t_kr = run<A_KROWS, EPI_BIAS, 128>(h, 10);
        std::printf("K=%4d  rows+tanh %.1f us %.1f TF | rows+bias %.1f us %.1f TF | krows+bias %.1f us %.1f TF\n", K, t_tanh,
                    fl / t_tanh * 1e-6, t_bias, fl / t_bias * 1e-6, t_kr, fl / t_kr * 1e-6);
    }
    return 0;
}

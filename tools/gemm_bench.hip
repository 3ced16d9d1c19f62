// Tuning harness for streamz_amd/csrc/gemm_kernel.h: times the K-tile / double-buffer variants
// of the GEMM kernel on the shapes of one 32 768-row training chunk.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../streamz_amd/csrc -o gemm_bench gemm_bench.hip
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm_kernel.h"

using namespace sz::gemm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

template <int AM, int BMd, int EPI, int TM, int KT, int DB>
float run(GemmArgs g, int splits, int iters)
{
    dim3 grid((g.N + BN - 1) / BN, (g.M + TM - 1) / TM, splits);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_kernel<AM, BMd, EPI, 0, TM, KT, DB>), grid, dim3(kGemmThreads), 0, 0, g);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_kernel<AM, BMd, EPI, 0, TM, KT, DB>), grid, dim3(kGemmThreads), 0, 0, g);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.0f / iters;
}

template <int KT, int DB>
void variant(float* X, float* H1, float* H2, float* D2, float* D1, float* W1, float* W2, float* slab, float* bias, int R)
{
    const int H1n = 512, H2n = 256, IN = 60;
    GemmArgs g{};
    auto vec = [](const void* p, int ld) { return (((uintptr_t)p & 15) == 0 && (ld & 3) == 0) ? 1 : 0; };
    // fwd1
    g = GemmArgs{}; g.A = X; g.lda = IN; g.B = W1; g.ldb = H1n; g.C = H1; g.ldc = H1n; g.M = R; g.N = H1n; g.K = IN; g.k_chunk = IN; g.bias = bias;
    g.a_vec = vec(g.A, g.lda); g.b_vec = vec(g.B, g.ldb);
    float t1 = run<A_ROWS, B_KROWS, EPI_BIAS_RELU, 128, KT, DB>(g, 1, 20);
    // fwd2
    g = GemmArgs{}; g.A = H1; g.lda = H1n; g.B = W2; g.ldb = H2n; g.C = H2; g.ldc = H2n; g.M = R; g.N = H2n; g.K = H1n; g.k_chunk = H1n; g.bias = bias;
    g.a_vec = vec(g.A, g.lda); g.b_vec = vec(g.B, g.ldb);
    float t2 = run<A_ROWS, B_KROWS, EPI_BIAS_TANH, 128, KT, DB>(g, 1, 20);
    // delta1
    g = GemmArgs{}; g.A = D2; g.lda = H2n; g.B = W2 /* read as w2^T [256][512] */; g.ldb = H1n; g.C = D1; g.ldc = H1n; g.M = R; g.N = H1n; g.K = H2n; g.k_chunk = H2n; g.aux = H1; g.ldaux = H1n;
    g.a_vec = vec(g.A, g.lda); g.b_vec = vec(g.B, g.ldb);
    float t3 = run<A_ROWS, B_KROWS, EPI_MUL_DRELU, 128, KT, DB>(g, 1, 20);
    // dw2
    int splits = 64, kc = (R + splits - 1) / splits; kc = (kc + KT - 1) / KT * KT;
    g = GemmArgs{}; g.A = H1; g.lda = H1n; g.B = D2; g.ldb = H2n; g.C = slab; g.ldc = H2n; g.M = H1n; g.N = H2n; g.K = R; g.k_chunk = kc;
    g.a_vec = vec(g.A, g.lda); g.b_vec = vec(g.B, g.ldb);
    float t4 = run<A_KROWS, B_KROWS, EPI_SLAB, 128, KT, DB>(g, (R + kc - 1) / kc, 20);
    // dw1 with the ones row
    g = GemmArgs{}; g.A = X; g.lda = IN; g.B = D1; g.ldb = H1n; g.C = slab; g.ldc = H1n; g.M = IN + 1; g.N = H1n; g.K = R; g.k_chunk = kc; g.ones_row = IN;
    g.a_vec = vec(g.A, g.lda); g.b_vec = vec(g.B, g.ldb);
    float t5 = run<A_KROWS, B_KROWS, EPI_SLAB, 64, KT, DB>(g, (R + kc - 1) / kc, 20);
    const double f2 = 2.0 * R * 512 * 256;
    std::printf("KT=%d DB=%d : fwd1 %.1f us | fwd2 %.1f us (%.1f TF) | delta1 %.1f us (%.1f TF) | dw2 %.1f us (%.1f TF) | dw1 %.1f us\n",
                KT, DB, t1, t2, f2 / t2 * 1e-6, t3, f2 / t3 * 1e-6, t4, f2 / t4 * 1e-6, t5);
}

int main(int argc, char** argv)
{
    const int R = argc > 1 ? std::atoi(argv[1]) : 32768;
    auto dalloc = [](size_t n) { float* p; CK(hipMalloc((void**)&p, n * sizeof(float))); return p; };
    float *X = dalloc((size_t)R * 60), *H1 = dalloc((size_t)R * 512), *H2 = dalloc((size_t)R * 256), *D2 = dalloc((size_t)R * 256),
          *D1 = dalloc((size_t)R * 512), *W1 = dalloc(60 * 512), *W2 = dalloc(512 * 256), *slab = dalloc((size_t)64 * 513 * 512), *bias = dalloc(512);
    std::vector<float> h((size_t)R * 512 + 8192);
    std::srand(1);
    for (auto& v : h) v = (std::rand() / (float)RAND_MAX) - 0.5f;
    CK(hipMemcpy(X, h.data(), (size_t)R * 60 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(H1, h.data(), (size_t)R * 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(H2, h.data(), (size_t)R * 256 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(D2, h.data() + 1000, (size_t)R * 256 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(D1, h.data() + 2000, (size_t)R * 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W1, h.data() + 3000, 60 * 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W2, h.data() + 4000, 512 * 256 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data() + 5000, 512 * 4, hipMemcpyHostToDevice));
    std::printf("R=%d\n", R);
    for (int rep = 0; rep < 2; ++rep) {
        variant<16, 0>(X, H1, H2, D2, D1, W1, W2, slab, bias, R);
        if (argc > 2) {
            variant<16, 1>(X, H1, H2, D2, D1, W1, W2, slab, bias, R);
            variant<32, 0>(X, H1, H2, D2, D1, W1, W2, slab, bias, R);
            variant<32, 1>(X, H1, H2, D2, D1, W1, W2, slab, bias, R);
        }
    }
    return 0;
}

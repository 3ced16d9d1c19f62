// Timing harness of the fused vote kernel (streamz_amd/csrc/infer_kernel.h) on random tables, for ablations:
// build with -DSZ_INFER_ABLATE=<bits> (1: no first-layer MFMAs, 2: no second-layer MFMAs, 4: no output layer /
// softmax / vote, 8, 16, 32, 64: see infer_kernel.h) to see what each part of a tile costs.  Results of an ablated build are meaningless.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../streamz_amd/csrc -I../include -o infer_bench infer_bench.hip
//   ./infer_bench [rows] [mt]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "infer_kernel.h"

#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    using namespace sz::infer;
    const uint64_t rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 8192;
    const int mt = argc > 2 ? std::atoi(argv[2]) : 1;
    const int H1 = 512, H2 = 256, C = 16, in = 60;
    std::mt19937 gen(1);
    std::uniform_real_distribution<float> ud(-0.5f, 0.5f);
    std::vector<float> x(rows * in), tab(table_floats(H1, H2)), b(H2 + C);
    for (auto& v : x) v = ud(gen) * 4;
    for (auto& v : tab) v = ud(gen);
    for (auto& v : b) v = ud(gen);
    float *dx, *dt, *db; unsigned long long* dc;
    HK(hipMalloc((void**)&dx, x.size() * 4)); HK(hipMalloc((void**)&dt, tab.size() * 4)); HK(hipMalloc((void**)&db, b.size() * 4));
    HK(hipMalloc((void**)&dc, 64 * 8));
    HK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemset(dc, 0, 64 * 8));
    Args a{};
    a.x = dx; a.rows = rows; a.in = in; a.tab = dt; a.b2 = db; a.b3 = db + H2; a.H1 = H1; a.H2 = H2; a.C = C; a.threshold = 0.1f;
    a.counts = dc; a.psums = nullptr; a.tail_off = (int)act_floats(mt, H1, 1);
    const size_t lds = lds_bytes(mt, H1, 1, in);
    const void* fn = mt == 2 ? (const void*)infer_vote_kernel<2, 1> : (const void*)infer_vote_kernel<1, 1>;
    HK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t tiles = (rows + 32 * mt - 1) / (32 * mt);
    const unsigned blocks = (unsigned)(tiles < 256 ? tiles : 256);
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    const int reps = rows > 1000000 ? 20 : 2000;
    for (int pass = 0; pass < 2; ++pass) {
        HK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            if (mt == 2) hipLaunchKernelGGL((infer_vote_kernel<2, 1>), dim3(blocks), dim3(kThreads), lds, 0, a);
            else hipLaunchKernelGGL((infer_vote_kernel<1, 1>), dim3(blocks), dim3(kThreads), lds, 0, a);
        }
        HK(hipEventRecord(e1));
        HK(hipEventSynchronize(e1));
    }
    float ms = 0;
    HK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, flop = 2.0 * (64.0 * H1 + (double)H1 * H2 + 32.0 * H2) * rows;
    std::printf("ablate %d rows %llu mt %d: %.2f us per launch (back to back), %.1f M rows/s, %.1f TFLOP/s issued\n",
#ifdef SZ_INFER_ABLATE
                SZ_INFER_ABLATE,
#else
                0,
#endif
                (unsigned long long)rows, mt, us, rows / us, flop / us / 1e6);
    return 0;
}

// Pure-MFMA ceiling of this device: back-to-back matrix instructions on register operands (no
// memory traffic inside the loop), every CU busy, random operand bits.  Prints, per instruction
// shape and per waves-per-SIMD, the sustained TFLOP/s over >= 2 s of launches and the shader clock
// the chip held inside the kernel (s_memtime ticks per 100 MHz s_memrealtime tick).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 16;  // MFMAs per loop trip and accumulator set

__device__ inline float rnd(uint32_t s)
{
    s = s * 747796405u + 2891336453u;
    s = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
    s = (s >> 22) ^ s;
    return (float)(int32_t)s * (1.0f / 2147483648.0f);
}

struct Stamp { uint64_t clk0, clk1, real0, real1; };

enum Shape { F32_32x32x2, F32_16x16x4, BF16_32x32x16, BF16_16x16x32 };

template <int SHAPE>
__global__ __launch_bounds__(512) void peak_kernel(int trips, float* sink, Stamp* stamps)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float total = 0.0f;
    if constexpr (SHAPE == F32_32x32x2) {
        f16v acc[4] = {};
        float a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = rnd(id * 8 + i); b[i] = rnd(id * 8 + 4 + i); }
        for (int t = 0; t < trips; ++t)
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 3], b[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) total += acc[i][j];
    } else if constexpr (SHAPE == F32_16x16x4) {
        f4 acc[8] = {};
        float a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = rnd(id * 8 + i); b[i] = rnd(id * 8 + 4 + i); }
        for (int t = 0; t < trips; ++t)
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 3], b[(u >> 2) & 3], acc[u & 7], 0, 0, 0);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) total += acc[i][j];
    } else if constexpr (SHAPE == BF16_32x32x16) {
        f16v acc[4] = {};
        bf8 a[2], b[2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)rnd(id * 32 + i * 8 + j); b[i][j] = (__bf16)rnd(id * 32 + 16 + i * 8 + j); }
        for (int t = 0; t < trips; ++t)
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 1], b[(u >> 1) & 1], acc[u & 3], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) total += acc[i][j];
    } else {
        f4 acc[8] = {};
        bf8 a[2], b[2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)rnd(id * 32 + i * 8 + j); b[i][j] = (__bf16)rnd(id * 32 + 16 + i * 8 + j); }
        for (int t = 0; t < trips; ++t)
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u & 1], b[(u >> 1) & 1], acc[u & 7], 0, 0, 0);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) total += acc[i][j];
    }
    if (threadIdx.x == 0) {
        Stamp s;
        s.clk0 = c0; s.real0 = r0;
        s.clk1 = __builtin_amdgcn_s_memtime(); s.real1 = __builtin_amdgcn_s_memrealtime();
        stamps[blockIdx.x] = s;
    }
    if (total == 12345.678f) sink[id] = total;  // keeps the accumulators alive
}

struct Case { const char* name; int shape; double flop_per_mfma; };

template <int SHAPE>
void run(const Case& c, int cus, int waves_per_simd, float* sink, Stamp* d_stamps, std::vector<Stamp>& h_stamps)
{
    const int threads = 256 * waves_per_simd, blocks = cus, trips = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) peak_kernel<SHAPE><<<blocks, threads>>>(trips, sink, d_stamps);
    CHECK(hipDeviceSynchronize());
    double total_ms = 0;
    int launches = 0;
    while (total_ms < 2000.0) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) peak_kernel<SHAPE><<<blocks, threads>>>(trips, sink, d_stamps);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        launches += 10;
    }
    CHECK(hipMemcpy(h_stamps.data(), d_stamps, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int b = 0; b < blocks; ++b) {
        const Stamp& s = h_stamps[b];
        if (s.real1 > s.real0) ghz.push_back((double)(s.clk1 - s.clk0) / (double)(s.real1 - s.real0) * 0.1);
    }
    std::sort(ghz.begin(), ghz.end());
    const double clock = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    const double mfma = (double)launches * blocks * (threads / 64) * (double)trips * kUnroll;
    const double tf = mfma * c.flop_per_mfma / (total_ms * 1e-3) * 1e-12;
    const double cyc = clock * 1e9 * (total_ms * 1e-3) / ((double)launches * trips * kUnroll * waves_per_simd);
    printf("%-14s waves/SIMD %d  %8.1f TFLOP/s  in-kernel clock %.3f GHz  %.1f cycles per MFMA per SIMD (launch overhead included)\n",
           c.name, waves_per_simd, tf, clock, cyc);
    fflush(stdout);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clockRate %.0f MHz\n", p.gcnArchName, cus, p.clockRate * 1e-3);
    float* sink;
    Stamp* d_stamps;
    CHECK(hipMalloc(&sink, sizeof(float) * cus * 512));
    CHECK(hipMalloc(&d_stamps, sizeof(Stamp) * cus));
    std::vector<Stamp> h(cus);
    const Case cases[4] = {{"f32 32x32x2", F32_32x32x2, 2.0 * 32 * 32 * 2}, {"f32 16x16x4", F32_16x16x4, 2.0 * 16 * 16 * 4},
                           {"bf16 32x32x16", BF16_32x32x16, 2.0 * 32 * 32 * 16}, {"bf16 16x16x32", BF16_16x16x32, 2.0 * 16 * 16 * 32}};
    for (int w = 1; w <= 2; ++w) {
        run<F32_32x32x2>(cases[0], cus, w, sink, d_stamps, h);
        run<F32_16x16x4>(cases[1], cus, w, sink, d_stamps, h);
        run<BF16_32x32x16>(cases[2], cus, w, sink, d_stamps, h);
        run<BF16_16x16x32>(cases[3], cus, w, sink, d_stamps, h);
    }
    return 0;
}

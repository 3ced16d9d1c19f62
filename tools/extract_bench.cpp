// Torch-free driver of the extraction entry points, for rocprofv3 counter passes: `clips` clips of
// `seconds` s of pseudo-random PCM, device-resident, extracted `repeats` times through the C ABI.
//   hipcc -O2 -std=c++17 -I../include -o extract_bench extract_bench.cpp -L../streamz_amd -lstreamz_hip -Wl,-rpath,'$ORIGIN/../streamz_amd'
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "streamz_hip.h"

#define CK(x) do { if ((x) != 0) { std::printf("failed %s: %s\n", #x, szh_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const uint32_t clips = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 200;
    const double seconds = argc > 2 ? std::atof(argv[2]) : 60.0;
    const int repeats = argc > 3 ? std::atoi(argv[3]) : 3;
    const uint64_t n = (uint64_t)(seconds * 44100.0), n_al = (n + 7) / 8 * 8;
    CK(szh_init(0));
    szh_extractor* ext = nullptr;
    CK(szh_extractor_create(&ext));
    std::vector<int16_t> pcm(n_al * clips);
    uint32_t s = 12345u;
    for (auto& v : pcm) { s = s * 1664525u + 1013904223u; v = (int16_t)((int)(s >> 17) - 16384); }
    std::vector<uint64_t> off(clips), len(clips, n);
    for (uint32_t i = 0; i < clips; ++i) off[i] = i * n_al;
    szh_plan* plan = nullptr;
    CK(szh_plan_create(ext, off.data(), len.data(), clips, &plan));
    const uint64_t nw = szh_plan_total_windows(plan);
    int16_t* d_pcm; float* d_feat;
    HK(hipMalloc((void**)&d_pcm, pcm.size() * 2));
    HK(hipMalloc((void**)&d_feat, nw * 60 * 4));
    HK(hipMemcpy(d_pcm, pcm.data(), pcm.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
    CK(szh_plan_run_dev(plan, d_pcm, d_feat, nullptr));
    HK(hipDeviceSynchronize());
    HK(hipEventRecord(a));
    for (int r = 0; r < repeats; ++r) CK(szh_plan_run_dev(plan, d_pcm, d_feat, nullptr));
    HK(hipEventRecord(b));
    HK(hipEventSynchronize(b));
    float ms = 0;
    HK(hipEventElapsedTime(&ms, a, b));
    std::printf("%llu windows, %.3f ms per extraction, %.1f M windows/s\n", (unsigned long long)nw, ms / repeats,
                nw * repeats / (ms * 1e3));
    szh_plan_destroy(plan);
    szh_extractor_destroy(ext);
    return 0;
}

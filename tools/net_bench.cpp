// Torch-free driver of the training entry points, for rocprofv3 counter passes: one epoch-sized
// gradient accumulation over random feature rows through the C ABI, repeated.
//   hipcc -O2 -std=c++17 -I../include -o net_bench net_bench.cpp -L../streamz_amd -lstreamz_hip -Wl,-rpath,'$ORIGIN/../streamz_amd'
//   ./net_bench [rows] [classes] [repeats]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "streamz_hip.h"

#define CK(x) do { if ((x) != 0) { std::printf("failed %s: %s\n", #x, szh_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const uint64_t rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 49152 * 4;
    const uint32_t classes = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 16;
    const int repeats = argc > 3 ? std::atoi(argv[3]) : 3;
    CK(szh_init(0));
    szh_net* net = nullptr;
    CK(szh_net_create(60, 512, 256, classes, nullptr, 7, &net));
    std::vector<float> x(rows * 60);
    std::vector<int32_t> lab(rows);
    std::mt19937 gen(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : x) v = nd(gen);
    for (auto& v : lab) v = (int32_t)(gen() % classes);
    float* dx; int32_t* dl;
    HK(hipMalloc((void**)&dx, x.size() * 4));
    HK(hipMalloc((void**)&dl, lab.size() * 4));
    HK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dl, lab.data(), lab.size() * 4, hipMemcpyHostToDevice));
    for (int r = 0; r < repeats + 1; ++r) {
        if (r == 1) szh_profile_enable(1);
        CK(szh_net_zero_grad_dev(net, nullptr));
        CK(szh_net_accumulate_dev(net, dx, dl, rows, 0.2f, 11 + r, nullptr));
        CK(szh_net_apply_grad_dev(net, 0.01f, nullptr));
        HK(hipDeviceSynchronize());
    }
    std::vector<char> buf(65536);
    szh_profile_report(buf.data(), buf.size());
    std::printf("%s\n", buf.data());
    szh_net_destroy(net);
    return 0;
}

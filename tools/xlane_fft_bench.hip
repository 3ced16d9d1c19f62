// What would a 400-point complex FFT cost with 16 lanes per window and the second stage done ACROSS the lanes
// of a DPP row instead of through LDS?  400 = 25 x 16: lane n2 (0..15) of a row holds z[16 n1 + n2], n1 = 0..24;
//   stage 1  a 25-point DFT per lane (5 x 5, twiddles between the two 5-point passes), then w400^(n2 k1),
//            the lane's 25 outer twiddles resident in registers;
//   stage 2  for each of the 25 k1 a 16-point DFT over the row's lanes: four radix-2 passes, the partner's value
//            fetched by a DPP move (quad_perm for distance 1 and 2, two bank-masked row rotations for 4,
//            row_ror:8 for 8); lane l ends up with Z[k1 + 25 bitrev4(l)].
// The program checks the result against a direct DFT in double precision, then times stage 1 + stage 2, stage 1
// alone and stage 2 alone, and prints SIMD cycles per window (time x clock x 1024 SIMDs / windows), the unit in
// which the extraction kernel's phases are known (mfcc_base_kernel: 1 583 per window in all).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../streamz_amd/csrc -o bin/xlane_fft_bench xlane_fft_bench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fftmath.h"

using sz::v2f;

#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_into(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
// the value of lane (l ^ D) of the same row of 16
template <int D>
__device__ __forceinline__ float from_xor(float v)
{
    if (D == 1) return dpp_into<0xB1, 0xf>(v, v);   // quad_perm [1,0,3,2]
    if (D == 2) return dpp_into<0x4E, 0xf>(v, v);   // quad_perm [2,3,0,1]
    if (D == 8) return dpp_into<0x128, 0xf>(v, v);  // row_ror:8
    // distance 4: lanes 0-3 and 8-11 take from four lanes up, the others from four lanes down
    float t = dpp_into<0x12C, 0x5>(v, v);            // row_ror:12 into banks 0 and 2: lane i <- lane i + 4
    return dpp_into<0x124, 0xA>(t, v);               // row_ror:4 into banks 1 and 3:  lane i <- lane i - 4
}
template <int D>
__device__ __forceinline__ v2f from_xor2(v2f v) { return sz::mk(from_xor<D>(v[0]), from_xor<D>(v[1])); }

struct LaneTables {
    v2f outer[25];   // w400^(n2 k1) for this lane's n2
    v2f st8, st4, st2;  // the twiddle this lane applies after the passes at distance 8, 4 and 2 (1 for the lower partner)
    v2f sg8, sg4, sg2, sg1;  // +1 for the lower partner of a pass, -1 for the upper one, in both halves
};

__device__ __forceinline__ v2f cexp_turns(int num, int den)
{
    float s, c;
    sincospif(-2.0f * (float)num / (float)den, &s, &c);
    return sz::mk(c, s);
}

__device__ __forceinline__ void make_tables(int l, LaneTables& t)
{
#pragma unroll
    for (int k1 = 0; k1 < 25; ++k1) t.outer[k1] = cexp_turns((l * k1) % 400, 400);
    const v2f one = sz::mk(1.0f, 0.0f);
    t.st8 = (l & 8) ? cexp_turns(l & 7, 16) : one;
    t.st4 = (l & 4) ? cexp_turns(2 * (l & 3), 16) : one;
    t.st2 = (l & 2) ? cexp_turns(4 * (l & 1), 16) : one;
    t.sg8 = (l & 8) ? sz::mk(-1, -1) : sz::mk(1, 1);
    t.sg4 = (l & 4) ? sz::mk(-1, -1) : sz::mk(1, 1);
    t.sg2 = (l & 2) ? sz::mk(-1, -1) : sz::mk(1, 1);
    t.sg1 = (l & 1) ? sz::mk(-1, -1) : sz::mk(1, 1);
}

// 25-point DFT, n1 = 5 a + b, k1 = c + 5 d
__device__ __forceinline__ void dft25(v2f (&x)[25])
{
    v2f t[5][5];
#pragma unroll
    for (int b = 0; b < 5; ++b) sz::dft5(x[b], x[5 + b], x[10 + b], x[15 + b], x[20 + b], t[b][0], t[b][1], t[b][2], t[b][3], t[b][4]);
#pragma unroll
    for (int b = 1; b < 5; ++b)
#pragma unroll
        for (int c = 1; c < 5; ++c) {
            const double ang = -2.0 * 3.14159265358979323846 * (b * c) / 25.0;
            const float wr = (float)__builtin_cos(ang), wi = (float)__builtin_sin(ang);
            const v2f a = t[b][c];
            t[b][c] = sz::mk(a[0] * wr - a[1] * wi, a[0] * wi + a[1] * wr);
        }
#pragma unroll
    for (int c = 0; c < 5; ++c) sz::dft5(t[0][c], t[1][c], t[2][c], t[3][c], t[4][c], x[c], x[c + 5], x[c + 10], x[c + 15], x[c + 20]);
}

__device__ __forceinline__ void stage1(v2f (&x)[25], const LaneTables& t)
{
    dft25(x);
#pragma unroll
    for (int k1 = 0; k1 < 25; ++k1) x[k1] = sz::cmul(x[k1], t.outer[k1]);
}

template <int D>
__device__ __forceinline__ v2f pass(v2f own, v2f sg, v2f tw, bool twiddle)
{
    const v2f other = from_xor2<D>(own);
    const v2f s = own * sg + other;   // lower partner: own + other; upper partner: other - own
    return twiddle ? sz::cmul(s, tw) : s;
}
__device__ __forceinline__ void stage2(v2f (&x)[25], const LaneTables& t)
{
#pragma unroll
    for (int k1 = 0; k1 < 25; ++k1) {
        v2f v = x[k1];
        v = pass<8>(v, t.sg8, t.st8, true);
        v = pass<4>(v, t.sg4, t.st4, true);
        v = pass<2>(v, t.sg2, t.st2, true);
        v = pass<1>(v, t.sg1, t.st2, false);
        x[k1] = v;
    }
}

// mode 0: stage 1 + stage 2, 1: stage 1 only, 2: stage 2 only.  z / out: [windows][400] complex.
template <int MODE>
__global__ __launch_bounds__(256) void fft_kernel(const float2* __restrict__ z, float2* __restrict__ out, int reps, float damp)
{
    const int l = threadIdx.x & 15;
    const size_t win = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    LaneTables t;
    make_tables(l, t);
    v2f x[25];
#pragma unroll
    for (int n1 = 0; n1 < 25; ++n1) { const float2 v = z[win * 400 + 16 * n1 + l]; x[n1] = sz::mk(v.x, v.y); }
    for (int r = 0; r < reps; ++r) {
        if (MODE != 2) stage1(x, t);
        if (MODE != 1) stage2(x, t);
        if (r + 1 < reps) {
#pragma unroll
            for (int i = 0; i < 25; ++i) x[i] = x[i] * sz::mk(damp, damp);  // keeps repeated transforms finite (timing runs only)
        }
    }
    int br = ((l & 1) << 3) | ((l & 2) << 1) | ((l & 4) >> 1) | ((l & 8) >> 3);
#pragma unroll
    for (int k1 = 0; k1 < 25; ++k1) out[win * 400 + k1 + 25 * br] = make_float2(x[k1][0], x[k1][1]);
}

template <int MODE>
static int time_mode(const float2* dz, float2* dout, int blocks, int reps, double windows, const char* what)
{
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fft_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, dz, dout, reps, 0.01f);
    HK(hipEventRecord(e0));
    hipLaunchKernelGGL(fft_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, dz, dout, reps, 0.01f);
    HK(hipEventRecord(e1));
    HK(hipEventSynchronize(e1));
    float ms = 0;
    HK(hipEventElapsedTime(&ms, e0, e1));
    hipLaunchKernelGGL(fft_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, dz, dout, 1, 0.01f);
    HK(hipEventRecord(e0));
    hipLaunchKernelGGL(fft_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, dz, dout, 1, 0.01f);
    HK(hipEventRecord(e1));
    HK(hipEventSynchronize(e1));
    float ms1 = 0;
    HK(hipEventElapsedTime(&ms1, e0, e1));
    const double per = (ms - ms1) * 1e-3 / (reps - 1) / windows;  // seconds per window-transform, loads and stores taken out
    std::printf("%-22s %8.3f ms for %d repetitions -> %.1f SIMD cycles per window at 2.4 GHz on 1024 SIMDs\n", what, ms, reps,
                per * 2.4e9 * 1024.0);
    return 0;
}

int main()
{
    const int blocks = 256 * 12, windows = blocks * 16;   // 16 windows per workgroup of 256 lanes
    std::vector<float2> z((size_t)windows * 400), got(z.size());
    unsigned s = 12345;
    for (auto& v : z) { s = s * 1664525u + 1013904223u; v.x = (float)((int)(s >> 8) % 2001 - 1000) / 1000.0f; s = s * 1664525u + 1013904223u; v.y = (float)((int)(s >> 8) % 2001 - 1000) / 1000.0f; }
    float2 *dz, *dout;
    HK(hipMalloc((void**)&dz, z.size() * sizeof(float2)));
    HK(hipMalloc((void**)&dout, z.size() * sizeof(float2)));
    HK(hipMemcpy(dz, z.data(), z.size() * sizeof(float2), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fft_kernel<0>, dim3(blocks), dim3(256), 0, 0, dz, dout, 1, 1.0f);
    HK(hipMemcpy(got.data(), dout, z.size() * sizeof(float2), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int w : {0, 7, windows - 1}) {
        for (int k = 0; k < 400; ++k) {
            std::complex<double> acc = 0;
            for (int n = 0; n < 400; ++n)
                acc += std::complex<double>(z[(size_t)w * 400 + n].x, z[(size_t)w * 400 + n].y) *
                       std::polar(1.0, -2.0 * M_PI * (double)((n * k) % 400) / 400.0);
            worst = std::fmax(worst, std::abs(acc - std::complex<double>(got[(size_t)w * 400 + k].x, got[(size_t)w * 400 + k].y)));
        }
    }
    std::printf("largest deviation from the direct DFT over three windows: %.3g (inputs in [-1, 1])\n", worst);
    if (!(worst < 1e-3)) { std::printf("WRONG RESULT\n"); return 2; }
    const int reps = 200;
    if (time_mode<0>(dz, dout, blocks, reps, windows, "stage 1 + stage 2")) return 1;
    if (time_mode<1>(dz, dout, blocks, reps, windows, "stage 1 (25-point)")) return 1;
    if (time_mode<2>(dz, dout, blocks, reps, windows, "stage 2 (across lanes)")) return 1;
    return 0;
}

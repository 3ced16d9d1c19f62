// Do vector instructions of one wave run while another wave of the same SIMD keeps the matrix pipe busy?
// A workgroup is 4 waves (one per SIMD), two workgroups share a compute unit (256 registers each).  Three
// launches of 512 workgroups on 256 compute units:
//   M  every workgroup runs a chain of v_mfma_f32_32x32x2_f32 (two accumulators, back to back)
//   V  every workgroup runs a loop of dependent-free packed multiply-adds of the same duration
//   MV even workgroups run M's loop, odd ones V's (one of each per compute unit if the dispatcher pairs them;
//      the kernel records which wave slots they got)
// If the two kinds of work overlap, MV takes about as long as the longer of M/2 and V/2 ... M and V; if they
// exclude each other on a SIMD, MV takes (M + V) / 2.  Times in microseconds.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o bin/coexec_bench coexec_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// kind: 0 all matrix, 1 all vector, 2 by parity of the workgroup, 3 by parity of the workgroup's position in its
// XCD's sequence (workgroups go to the 8 XCDs round-robin)
__global__ __launch_bounds__(256, 2) void coexec_kernel(int kind, int m_iters, int v_iters, float* out, unsigned* slots)
{
    const unsigned b = blockIdx.x;
    const bool vec = kind == 1 || (kind == 2 && (b & 1)) || (kind == 3 && ((b >> 3) & 1)) || (kind == 4 && ((b >> 8) & 1));
    // HW_ID (register 4, all 32 bits: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) and XCC_ID (register 20)
    if (threadIdx.x == 0) { slots[2 * b] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | (vec ? 0x80000000u : 0u); slots[2 * b + 1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }
    float r = 0.0f;
    if (!vec) {
        f32x16 a, c;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[i] = 0.0f; c[i] = 0.0f; }
        const float x = (float)threadIdx.x * 1e-3f, y = 1.0f;
        for (int it = 0; it < m_iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c, 0, 0, 0);
            }
        }
        r = a[0] + c[5];
    } else {
        f32x2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x + i, 1.0f};
        const f32x2 m = {0.999f, 1.001f}, d = {1e-3f, -1e-3f};
        for (int it = 0; it < v_iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * m + d;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += v[i][0] + v[i][1];
    }
    if (r == 123.456f) out[b] = r;
}

static int run(int kind, int m_iters, int v_iters, float* d_out, unsigned* d_slots, const char* what, float* us_out)
{
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    hipLaunchKernelGGL(coexec_kernel, dim3(512), dim3(256), 0, 0, kind, m_iters, v_iters, d_out, d_slots);
    HK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(coexec_kernel, dim3(512), dim3(256), 0, 0, kind, m_iters, v_iters, d_out, d_slots);
    HK(hipEventRecord(e1));
    HK(hipEventSynchronize(e1));
    float ms = 0;
    HK(hipEventElapsedTime(&ms, e0, e1));
    *us_out = ms * 1e3f / 5;
    std::printf("%-44s %9.1f us\n", what, *us_out);
    return 0;
}

int main()
{
    float* d_out; unsigned* d_slots;
    HK(hipMalloc((void**)&d_out, 512 * 4)); HK(hipMalloc((void**)&d_slots, 1024 * 4));
    const int m_iters = 4000;   // 64 000 matrix instructions per wave, 64 cycles each: 1.7 ms alone on its SIMD
    float m, v, mv, mv3;
    if (run(0, m_iters, 0, d_out, d_slots, "M: matrix chains in both workgroups", &m)) return 1;
    // a vector loop of about the same length: 32 packed instructions per iteration
    int v_iters = 16000;
    if (run(1, 0, v_iters, d_out, d_slots, "V: vector loops in both workgroups", &v)) return 1;
    v_iters = (int)(v_iters * (m / v));
    if (run(1, 0, v_iters, d_out, d_slots, "V, length matched to M", &v)) return 1;
    if (run(2, m_iters, v_iters, d_out, d_slots, "MV: matrix in even, vector in odd workgroups", &mv)) return 1;
    auto mixed_cus = [&](const char* what) -> int {
        std::vector<unsigned> slots(1024);
        HK(hipMemcpy(slots.data(), d_slots, 1024 * 4, hipMemcpyDeviceToHost));
        int kinds[8 * 8 * 2 * 16] = {};  // [xcc][se][sh][cu]: bit 0 a matrix workgroup ran there, bit 1 a vector one
        for (int b = 0; b < 512; ++b) {
            const unsigned h = slots[2 * b], x = slots[2 * b + 1] & 7u;
            const unsigned cu = (h >> 8) & 15u, sh = (h >> 12) & 1u, se = (h >> 13) & 7u;
            kinds[((x * 8 + se) * 2 + sh) * 16 + cu] |= (h >> 31) ? 2 : 1;
        }
        int n_used = 0, n_mixed = 0;
        for (int k : kinds) { n_used += k != 0; n_mixed += k == 3; }
        std::printf("  (%s: %d compute units used, %d of them ran both kinds)\n", what, n_used, n_mixed);
        return 0;
    };
    if (mixed_cus("by workgroup parity")) return 1;
    if (run(3, m_iters, v_iters, d_out, d_slots, "MV: kinds alternate within an XCD's sequence", &mv3)) return 1;
    if (mixed_cus("alternating within an XCD's sequence")) return 1;
    float mv4;
    if (run(4, m_iters, v_iters, d_out, d_slots, "MV: first 256 workgroups matrix, the rest vector", &mv4)) return 1;
    if (mixed_cus("first half / second half")) return 1;
    std::printf("if the kinds excluded each other: %.1f us; if they overlapped completely: %.1f us\n", (m + v) / 2, (m > v ? m : v) / 2);
    return 0;
}

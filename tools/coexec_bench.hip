// Do vector instructions of one wave run while another wave of the same SIMD keeps the matrix pipe busy?
// A workgroup is 4 waves (one per SIMD), two workgroups share a compute unit (256 registers each).  Three
// launches of 512 workgroups on 256 compute units:
//   M  every workgroup runs a chain of v_mfma_f32_32x32x2_f32 (two accumulators, back to back); with argument
//      "bf16" v_mfma_f32_32x32x16_bf16 instead
//   V  every workgroup runs a loop of dependent-free packed multiply-adds of the same duration; with argument
//      "int" 32-bit integer xor / shift / add instead
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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF16, bool INT>
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
        bf16x8 xb, yb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(x + i); yb[i] = (__bf16)1.0f; }
        for (int it = 0; it < m_iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (BF16) {
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yb, xb, c, 0, 0, 0);
                } else {
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c, 0, 0, 0);
                }
            }
        }
        r = a[0] + c[5];
    } else {
        f32x2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x + i, 1.0f};
        const f32x2 m = {0.999f, 1.001f}, d = {1e-3f, -1e-3f};
        unsigned u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
        for (int it = 0; it < v_iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (INT) u[i] = (u[i] ^ 0x9e3779b9u) + (u[i] >> 3);   // (integer instructions only)
                    else v[i] = v[i] * m + d;
                }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += v[i][0] + v[i][1] + (float)u[i];
    }
    if (r == 123.456f) out[b] = r;
}

template <bool BF16, bool INT>
static int run(int kind, int m_iters, int v_iters, float* d_out, unsigned* d_slots, const char* what, float* us_out)
{
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    hipLaunchKernelGGL((coexec_kernel<BF16, INT>), dim3(512), dim3(256), 0, 0, kind, m_iters, v_iters, d_out, d_slots);
    HK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((coexec_kernel<BF16, INT>), dim3(512), dim3(256), 0, 0, kind, m_iters, v_iters, d_out, d_slots);
    HK(hipEventRecord(e1));
    HK(hipEventSynchronize(e1));
    float ms = 0;
    HK(hipEventElapsedTime(&ms, e0, e1));
    *us_out = ms * 1e3f / 5;
    std::printf("%-44s %9.1f us\n", what, *us_out);
    return 0;
}

template <bool BF16, bool INT>
static int all(int m_iters)
{
    float* d_out; unsigned* d_slots;
    HK(hipMalloc((void**)&d_out, 512 * 4)); HK(hipMalloc((void**)&d_slots, 1024 * 4));
    std::printf("matrix instruction: %s, vector instruction: %s\n", BF16 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32",
                INT ? "32-bit integer xor, shift, add" : "v_pk_fma_f32");
    float m, v, mv4;
    if (run<BF16, INT>(0, m_iters, 0, d_out, d_slots, "M: matrix chains in both workgroups", &m)) return 1;
    int v_iters = 16000;
    if (run<BF16, INT>(1, 0, v_iters, d_out, d_slots, "V: vector loops in both workgroups", &v)) return 1;
    v_iters = (int)(v_iters * (m / v));
    if (run<BF16, INT>(1, 0, v_iters, d_out, d_slots, "V, length matched to M", &v)) return 1;
    if (run<BF16, INT>(4, m_iters, v_iters, d_out, d_slots, "MV: first 256 workgroups matrix, the rest vector", &mv4)) return 1;
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
    std::printf("  (%d compute units used, %d of them ran one workgroup of each kind)\n", n_used, n_mixed);
    std::printf("  if the kinds excluded each other: %.1f us; if they overlapped completely: %.1f us\n", (m + v) / 2, (m > v ? m : v) / 2);
    (void)hipFree(d_out); (void)hipFree(d_slots);
    return 0;
}

int main()
{
    if (all<false, false>(4000)) return 1;
    if (all<false, true>(4000)) return 1;
    if (all<true, false>(8000)) return 1;
    if (all<true, true>(8000)) return 1;
    return 0;
}

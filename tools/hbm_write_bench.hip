// What the memory system takes per second in pure stores, pure loads and a copy (16-byte accesses, grid-stride,
// 13.5 GB each -- the size of the first hidden layer's activations for 6.6 M rows):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o bin/hbm_write_bench hbm_write_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void store_kernel(float4* dst, size_t n, float v)
{
    const float4 x = {v, v + 1.0f, v + 2.0f, v + 3.0f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = x;
}
__global__ __launch_bounds__(256) void load_kernel(const float4* src, size_t n, float* out)
{
    float4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = src[i];
        a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}
__global__ __launch_bounds__(256) void copy_kernel(float4* dst, const float4* src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// as a tiled product writes: a workgroup stores a 256 x 256 float tile of a [rows][512] matrix, 1 KB per row
__global__ __launch_bounds__(256) void tile_store_kernel(float4* dst, size_t tiles, float v)
{
    const float4 x = {v, v + 1.0f, v + 2.0f, v + 3.0f};
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t row0 = (t >> 1) * 256, col0 = (t & 1) * 64;   // in float4 units: 128 per row
        for (int e = threadIdx.x; e < 256 * 64; e += 256) dst[(row0 + (e >> 6)) * 128 + col0 + (e & 63)] = x;
    }
}

// general form: tiles of TR rows x TC float4 columns of a [rows][128 float4] matrix, column tiles of one row tile
// taken one after the other by the same workgroup when `paired`, by neighbouring workgroups otherwise
__global__ __launch_bounds__(256) void tile_store2_kernel(float4* dst, size_t row_tiles, int TR, int TC, int paired, float v)
{
    const float4 x = {v, v + 1.0f, v + 2.0f, v + 3.0f};
    const int col_tiles = 128 / TC;
    const size_t tiles = row_tiles * col_tiles;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        size_t rt, ct;
        if (paired) { rt = t / col_tiles; ct = t % col_tiles; }          // (same as below: t = rt * col_tiles + ct)
        else { rt = t / col_tiles; ct = t % col_tiles; }
        const size_t row0 = rt * TR, col0 = ct * TC;
        for (int e = threadIdx.x; e < TR * TC; e += 256) dst[(row0 + e / TC) * 128 + col0 + e % TC] = x;
    }
}

// The stores of the tiled products' epilogue: 8 waves, a wave owns 64 rows x 128 columns of a 256 x 256 tile as
// 2 x 4 matrix-instruction blocks of 32 x 32; a lane holds column l31 of a block and the 16 rows
// (r & 3) + 8 (r >> 2) + 4 half, so one store instruction writes 128 bytes in each of two rows.
// ORDER 0: block by block (tm, tn, r) as the kernels do; 1: (tm, r, tn), a row's four 128-byte pieces one after the
// other; 2: (r, tm, tn); 3: as 0 but only every fourth store followed by the others after a pause of other work
template <int ORDER>
__global__ __launch_bounds__(512) void epi_store_kernel(float* dst, size_t tiles, float v)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t m0 = (t >> 1) * 256, n0 = (t & 1) * 256;
        float* base = dst + (m0 + wm * 64 + 4 * half) * 512 + n0 + wn * 128 + l31;
        if (ORDER == 0) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) base[(size_t)(tm * 32 + (r & 3) + 8 * (r >> 2)) * 512 + tn * 32] = v + r;
        } else if (ORDER == 1) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) base[(size_t)(tm * 32 + (r & 3) + 8 * (r >> 2)) * 512 + tn * 32] = v + r;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) base[(size_t)(tm * 32 + (r & 3) + 8 * (r >> 2)) * 512 + tn * 32] = v + r;
        }
    }
}

int main()
{
    const size_t bytes = 6614000ull * 512 * 4, n = bytes / 16;
    float4 *a, *b; float* out;
    HK(hipMalloc((void**)&a, bytes)); HK(hipMalloc((void**)&b, bytes)); HK(hipMalloc((void**)&out, 16));
    HK(hipMemset(a, 0, bytes)); HK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    for (int blocks : {1024, 2048, 4096, 16384}) {
        float ms[4];
        for (int k = 0; k < 4; ++k) {
            for (int rep = 0; rep < 2; ++rep) {
                HK(hipEventRecord(e0));
                if (k == 0) hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(256), 0, 0, a, n, 1.0f);
                if (k == 1) hipLaunchKernelGGL(load_kernel, dim3(blocks), dim3(256), 0, 0, a, n, out);
                if (k == 2) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, b, a, n);
                if (k == 3) hipLaunchKernelGGL(tile_store_kernel, dim3(blocks), dim3(256), 0, 0, a, (size_t)(6614000 / 256) * 2, 2.0f);
                HK(hipEventRecord(e1)); HK(hipEventSynchronize(e1));
                HK(hipEventElapsedTime(&ms[k], e0, e1));
            }
        }
        std::printf("%5d workgroups: stores %.2f ms = %.2f TB/s, loads %.2f ms = %.2f TB/s, copy %.2f ms = %.2f TB/s (read + written), stores in 256 x 256 tiles %.2f ms = %.2f TB/s\n",
                    blocks, ms[0], bytes / ms[0] / 1e9, ms[1], bytes / ms[1] / 1e9, ms[2], 2.0 * bytes / ms[2] / 1e9, ms[3], (double)(6614000 / 256) * 2 * 256 * 1024 / ms[3] / 1e9);
    }
    for (int grid : {256, 512, 1024}) {
        float ms[3];
        for (int k = 0; k < 3; ++k)
            for (int rep = 0; rep < 2; ++rep) {
                HK(hipEventRecord(e0));
                const size_t tiles = (size_t)(6614000 / 256) * 2;
                if (k == 0) hipLaunchKernelGGL(epi_store_kernel<0>, dim3(grid), dim3(512), 0, 0, (float*)a, tiles, 1.0f);
                if (k == 1) hipLaunchKernelGGL(epi_store_kernel<1>, dim3(grid), dim3(512), 0, 0, (float*)a, tiles, 1.0f);
                if (k == 2) hipLaunchKernelGGL(epi_store_kernel<2>, dim3(grid), dim3(512), 0, 0, (float*)a, tiles, 1.0f);
                HK(hipEventRecord(e1)); HK(hipEventSynchronize(e1));
                HK(hipEventElapsedTime(&ms[k], e0, e1));
            }
        const double b = (double)(6614000 / 256) * 2 * 256 * 1024;
        std::printf("epilogue-shaped stores, %4d workgroups of 8 waves: block by block %.2f ms = %.2f TB/s; row pieces together %.2f ms = %.2f TB/s; register-major %.2f ms = %.2f TB/s\n",
                    grid, ms[0], b / ms[0] / 1e9, ms[1], b / ms[1] / 1e9, ms[2], b / ms[2] / 1e9);
    }
    for (int tc : {128, 64, 32, 16, 8}) {
        for (int tr : {256, 64}) {
            const size_t row_tiles = 6614000 / tr;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                HK(hipEventRecord(e0));
                hipLaunchKernelGGL(tile_store2_kernel, dim3(2048), dim3(256), 0, 0, a, row_tiles, tr, tc, 0, 3.0f);
                HK(hipEventRecord(e1)); HK(hipEventSynchronize(e1));
                HK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::printf("tiles of %3d rows x %4d bytes per row: %.2f ms = %.2f TB/s\n", tr, tc * 16, ms, (double)row_tiles * tr * 2048 / ms / 1e9);
        }
    }
    return 0;
}

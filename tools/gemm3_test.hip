// Correctness and timing harness of the split-bf16 GEMM (streamz_amd/csrc/gemm_bf16x3.h) next to the
// f32 kernel, on the three large products of a training chunk.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Istreamz_amd/csrc -o tools/bin/gemm3_test tools/gemm3_test.hip
//   tools/bin/gemm3_test [rows for the timing part] [hidden1] [hidden2]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "gemm_bf16x3.h"

using namespace sz::gemm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

template <class T> T* dev(size_t n) { T* p; CK(hipMalloc((void**)&p, n * sizeof(T))); return p; }
template <class T> T* upload(const std::vector<T>& h) { T* p = dev<T>(h.size()); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }
std::vector<float> download(const float* p, size_t n) { std::vector<float> h(n); CK(hipMemcpy(h.data(), p, n * sizeof(float), hipMemcpyDeviceToHost)); return h; }

std::vector<float> rnd(size_t n, float scale, unsigned seed, bool relu = false)
{
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> u(-scale, scale);
    std::vector<float> v(n);
    for (auto& x : v) { x = u(g); if (relu && x < 0) x = 0; }
    return v;
}

uint32_t* split_dev(const float* src, size_t count)
{
    uint32_t* d = dev<uint32_t>(3 * count / 2);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((count / 2 + 255) / 256)), dim3(256), 0, 0, src, (uint64_t)count, d, (uint64_t)(count / 2));
    CK(hipGetLastError());
    return d;
}

double maxerr(const std::vector<float>& got, const std::vector<double>& ref)
{
    double e = 0;
    for (size_t i = 0; i < ref.size(); ++i) e = std::fmax(e, std::fabs((double)got[i] - ref[i]));
    return e;
}

template <int ASRC, int BSRC, int EPI, int TAG = 0>
void launch_w(GemmArgs g, int splits, int max_blocks = 256)
{
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_x3w_kernel<ASRC, BSRC, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES)); once = true; }
    g.tiles_x = (g.N + 255) / 256; g.tiles_y = (g.M + 255) / 256; g.tiles_z = splits;
    const int tiles = g.tiles_x * g.tiles_y * g.tiles_z;
    const int blocks = std::min((tiles + 7) / 8 * 8, max_blocks);
    hipLaunchKernelGGL((gemm_x3w_kernel<ASRC, BSRC, EPI, TAG>), dim3(blocks), dim3(XW_THREADS), XW_LDS_BYTES, 0, g);
}

template <int ASRC, int BSRC, int EPI>
void launch_fw(GemmArgs g, int splits, int max_blocks = 256)
{
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_f32w_kernel<ASRC, BSRC, EPI, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FW_LDS_BYTES)); once = true; }
    g.tiles_x = (g.N + 255) / 256; g.tiles_y = (g.M + 255) / 256; g.tiles_z = splits;
    const int tiles = g.tiles_x * g.tiles_y * g.tiles_z;
    const int blocks = std::min((tiles + 7) / 8 * 8, max_blocks);
    hipLaunchKernelGGL((gemm_f32w_kernel<ASRC, BSRC, EPI, 0>), dim3(blocks), dim3(XW_THREADS), FW_LDS_BYTES, 0, g);
}

template <class K> float time_us(K launch, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1000.0f / iters;
}

int main(int argc, char** argv)
{
    const int H1 = argc > 2 ? std::atoi(argv[2]) : 512, H2 = argc > 3 ? std::atoi(argv[3]) : 256;
    const int big = argc > 1 ? std::atoi(argv[1]) : 0;
    {   // ---- correctness on a ragged row count: three row tiles per workgroup-walk and a partial last tile ----
        const int R = 256 * 20 + 37;  // 21 row tiles on 8 workgroups: every workgroup walks over several
        auto h1 = rnd((size_t)R * H1, 3.0f, 1, true), w2 = rnd((size_t)H1 * H2, 0.5f, 2), b2 = rnd(H2, 0.1f, 3);
        auto d2 = rnd((size_t)R * H2, 1.0f, 4);
        std::vector<float> w2t((size_t)H1 * H2);
        for (int k = 0; k < H1; ++k) for (int n = 0; n < H2; ++n) w2t[(size_t)n * H1 + k] = w2[(size_t)k * H2 + n];
        h1.resize((size_t)(R + 32) * H1, 0.0f); d2.resize((size_t)(R + 32) * H2, 0.0f);  // zero rows up to the next multiple of 32
        float *dh1 = upload(h1), *dw2 = upload(w2), *dw2t = upload(w2t), *db2 = upload(b2), *dd2 = upload(d2);
        float *dh2 = dev<float>((size_t)R * H2), *dh2w = dev<float>((size_t)R * H2);
        float *dd1 = dev<float>((size_t)R * H1), *dd1w = dev<float>((size_t)R * H1);
        uint32_t* w2t3 = split_dev(dw2t, (size_t)H1 * H2);  // [3][H2][H1]: B of fwd2 as [n][k]
        uint32_t* w23 = split_dev(dw2, (size_t)H1 * H2);    // [3][H1][H2]: B of delta1 as [n][k]
        // fwd2: h2 = tanh(h1 . w2 + b2)
        GemmArgs g{};
        g.A = dh1; g.lda = H1; g.B = dw2; g.ldb = H2; g.C = dh2; g.ldc = H2; g.M = R; g.N = H2; g.K = H1; g.k_chunk = H1; g.bias = db2; g.a_vec = 1; g.b_vec = 1;
        dim3 grid(H2 / 128, (R + 127) / 128, 1);
        hipLaunchKernelGGL((gemm_kernel<A_ROWS, B_KROWS, EPI_BIAS_TANH, 0, 128, 32, 0>), grid, dim3(256), 0, 0, g);
        GemmArgs gw = g; gw.C = dh2w; gw.B = (const float*)w2t3; gw.ldb = H1; gw.b_plane = (uint64_t)H1 * H2;
        launch_w<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_BIAS_TANH>(gw, 1, 8);
        CK(hipDeviceSynchronize());
        std::vector<double> ref((size_t)R * H2);
        for (int r = 0; r < R; ++r) for (int n = 0; n < H2; ++n) {
            double s = b2[n];
            for (int k = 0; k < H1; ++k) s += (double)h1[(size_t)r * H1 + k] * w2[(size_t)k * H2 + n];
            ref[(size_t)r * H2 + n] = std::tanh(s);
        }
        std::printf("fwd2   (tanh out)  max |err| vs f64: f32 kernel %.3e   bf16x3 kernel %.3e\n", maxerr(download(dh2, ref.size()), ref), maxerr(download(dh2w, ref.size()), ref));
        {
            float* dh2f = dev<float>((size_t)R * H2);
            GemmArgs gf = g; gf.C = dh2f; gf.B = dw2t; gf.ldb = H1;
            launch_fw<X3_ROWS_F32, X3_ROWS_F32, EPI_BIAS_TANH>(gf, 1, 8);
            CK(hipDeviceSynchronize());
            auto a1 = download(dh2, ref.size()), a2 = download(dh2f, ref.size());
            double d = 0; for (size_t i = 0; i < a1.size(); ++i) d = std::fmax(d, std::fabs((double)a1[i] - a2[i]));
            std::printf("fwd2   wide f32 kernel: max |err| vs f64 %.3e, max |difference| to the 128-tile f32 kernel %.3e\n", maxerr(a2, ref), d);
        }
        // delta1 = (d2 . w2^T) * [h1 > 0]
        g = GemmArgs{};
        g.A = dd2; g.lda = H2; g.B = dw2t; g.ldb = H1; g.C = dd1; g.ldc = H1; g.M = R; g.N = H1; g.K = H2; g.k_chunk = H2; g.aux = dh1; g.ldaux = H1; g.a_vec = 1; g.b_vec = 1;
        grid = dim3(H1 / 128, (R + 127) / 128, 1);
        hipLaunchKernelGGL((gemm_kernel<A_ROWS, B_KROWS, EPI_MUL_DRELU, 0, 128, 32, 0>), grid, dim3(256), 0, 0, g);
        gw = g; gw.C = dd1w; gw.B = (const float*)w23; gw.ldb = H2; gw.b_plane = (uint64_t)H1 * H2;
        launch_w<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_MUL_DRELU>(gw, 1, 16);
        CK(hipDeviceSynchronize());
        std::vector<double> ref1((size_t)R * H1);
        for (int r = 0; r < R; ++r) for (int n = 0; n < H1; ++n) {
            double s = 0;
            for (int k = 0; k < H2; ++k) s += (double)d2[(size_t)r * H2 + k] * w2[(size_t)n * H2 + k];
            ref1[(size_t)r * H1 + n] = h1[(size_t)r * H1 + n] > 0 ? s : 0.0;
        }
        std::printf("delta1 max |err| vs f64: f32 kernel %.3e   bf16x3 kernel %.3e\n", maxerr(download(dd1, ref1.size()), ref1), maxerr(download(dd1w, ref1.size()), ref1));
        // dw2 = h1^T . d2 in split-K slabs, second call adds
        const int kc = 288, splits = (R + kc - 1) / kc;
        float *slab = dev<float>((size_t)splits * H1 * H2), *slabw = dev<float>((size_t)splits * H1 * H2);
        g = GemmArgs{};
        g.A = dh1; g.lda = H1; g.B = dd2; g.ldb = H2; g.C = slab; g.ldc = H2; g.M = H1; g.N = H2; g.K = R; g.k_chunk = kc; g.a_vec = 1; g.b_vec = 1;
        grid = dim3(H2 / 128, H1 / 128, splits);
        gw = g; gw.C = slabw;
        for (int pass = 0; pass < 2; ++pass) {
            g.slab_add = gw.slab_add = pass;
            hipLaunchKernelGGL((gemm_kernel<A_KROWS, B_KROWS, EPI_SLAB, 0, 128, 32, 0>), grid, dim3(256), 0, 0, g);
            launch_w<X3_KROWS_F32, X3_KROWS_F32, EPI_SLAB>(gw, splits, 16);
        }
        CK(hipDeviceSynchronize());
        std::vector<double> ref2((size_t)H1 * H2, 0.0);
        for (int r = 0; r < R; ++r) for (int m = 0; m < H1; ++m) {
            const double a = h1[(size_t)r * H1 + m];
            if (a == 0) continue;
            for (int n = 0; n < H2; ++n) ref2[(size_t)m * H2 + n] += 2.0 * a * d2[(size_t)r * H2 + n];
        }
        auto fold = [&](const float* s) {
            auto h = download(s, (size_t)splits * H1 * H2);
            std::vector<float> out((size_t)H1 * H2);
            for (size_t i = 0; i < out.size(); ++i) { double t = 0; for (int z = 0; z < splits; ++z) t += h[(size_t)z * H1 * H2 + i]; out[i] = (float)t; }
            return out;
        };
        double m2 = 0; for (double v : ref2) m2 = std::fmax(m2, std::fabs(v));
        std::printf("dw2    (max |value| %.1f) max |err| vs f64: f32 kernel %.3e   bf16x3 kernel %.3e\n", m2, maxerr(fold(slab), ref2), maxerr(fold(slabw), ref2));
    }
    if (big > 0) {  // ---- timing at `big` rows ----
        const size_t R = (size_t)big;
        float *h1 = dev<float>((R + 32) * H1), *h2 = dev<float>(R * H2), *d2 = dev<float>((R + 32) * H2), *d1 = dev<float>(R * H1);
        CK(hipMemset(h1 + R * H1, 0, 32 * H1 * sizeof(float))); CK(hipMemset(d2 + R * H2, 0, 32 * H2 * sizeof(float)));
        auto hw = rnd((size_t)H1 * H2, 0.5f, 5);
        float *w2 = upload(hw), *bias = upload(rnd(H1, 0.1f, 6));
        {
            auto blk = rnd((size_t)1 << 22, 1.0f, 7);
            for (size_t off = 0; off < R * H1; off += blk.size()) CK(hipMemcpy(h1 + off, blk.data(), std::min(blk.size(), R * H1 - off) * sizeof(float), hipMemcpyHostToDevice));
            for (size_t off = 0; off < R * H2; off += blk.size()) CK(hipMemcpy(d2 + off, blk.data(), std::min(blk.size(), R * H2 - off) * sizeof(float), hipMemcpyHostToDevice));
        }
        uint32_t* w3 = split_dev(w2, (size_t)H1 * H2);
        int splits = 64; int kc = (int)((R + splits - 1) / splits); kc = (kc + 31) / 32 * 32; splits = (int)((R + kc - 1) / kc);
        int splits_w = 128; int kcw = (int)((R + splits_w - 1) / splits_w); kcw = (kcw + 31) / 32 * 32; splits_w = (int)((R + kcw - 1) / kcw);
        float* slab = dev<float>((size_t)splits_w * H1 * H2);
        const double flop = 2.0 * R * H1 * H2;
        GemmArgs f2{}; f2.A = h1; f2.lda = H1; f2.B = w2; f2.ldb = H2; f2.C = h2; f2.ldc = H2; f2.M = (int)R; f2.N = H2; f2.K = H1; f2.k_chunk = H1; f2.bias = bias; f2.a_vec = 1; f2.b_vec = 1;
        GemmArgs f2x = f2; f2x.B = (const float*)w3; f2x.ldb = H1; f2x.b_plane = (uint64_t)H1 * H2;
        GemmArgs dl{}; dl.A = d2; dl.lda = H2; dl.B = w2; dl.ldb = H1; dl.C = d1; dl.ldc = H1; dl.M = (int)R; dl.N = H1; dl.K = H2; dl.k_chunk = H2; dl.aux = h1; dl.ldaux = H1; dl.a_vec = 1; dl.b_vec = 1;
        GemmArgs dlx = dl; dlx.B = (const float*)w3; dlx.ldb = H2; dlx.b_plane = (uint64_t)H1 * H2;
        GemmArgs dw{}; dw.A = h1; dw.lda = H1; dw.B = d2; dw.ldb = H2; dw.C = slab; dw.ldc = H2; dw.M = H1; dw.N = H2; dw.K = (int)R; dw.k_chunk = kc; dw.a_vec = 1; dw.b_vec = 1;
        GemmArgs dww = dw; dww.k_chunk = kcw;
        const dim3 g2(H2 / 128, (unsigned)((R + 127) / 128), 1), g1(H1 / 128, (unsigned)((R + 127) / 128), 1), gw(H2 / 128, H1 / 128, splits);
        const float t_f2 = time_us([&] { hipLaunchKernelGGL((gemm_kernel<A_ROWS, B_KROWS, EPI_BIAS_TANH, 0, 128, 32, 0>), g2, dim3(256), 0, 0, f2); }, 5);
        const float t_dl = time_us([&] { hipLaunchKernelGGL((gemm_kernel<A_ROWS, B_KROWS, EPI_MUL_DRELU, 0, 128, 32, 0>), g1, dim3(256), 0, 0, dl); }, 5);
        const float t_dw = time_us([&] { hipLaunchKernelGGL((gemm_kernel<A_KROWS, B_KROWS, EPI_SLAB, 0, 128, 32, 0>), gw, dim3(256), 0, 0, dw); }, 5);
        const float t_f2w = time_us([&] { launch_w<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_BIAS_TANH>(f2x, 1); }, 5);
        const float t_dlw = time_us([&] { launch_w<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_MUL_DRELU>(dlx, 1); }, 5);
        const float t_dww = time_us([&] { launch_w<X3_KROWS_F32, X3_KROWS_F32, EPI_SLAB>(dww, splits_w); }, 5);
        {
            std::vector<float> hwt((size_t)H1 * H2);
            for (int k = 0; k < H1; ++k) for (int n = 0; n < H2; ++n) hwt[(size_t)n * H1 + k] = hw[(size_t)k * H2 + n];
            float* w2t = upload(hwt);
            GemmArgs f2f = f2; f2f.B = w2t; f2f.ldb = H1;          // [n = H2][k = H1]
            GemmArgs dlf = dl; dlf.B = w2; dlf.ldb = H2;            // [n = H1][k = H2]: w2 itself
            GemmArgs dwf = dww;
            const float t1 = time_us([&] { launch_fw<X3_ROWS_F32, X3_ROWS_F32, EPI_BIAS_TANH>(f2f, 1); }, 5);
            const float t2 = time_us([&] { launch_fw<X3_ROWS_F32, X3_ROWS_F32, EPI_MUL_DRELU>(dlf, 1); }, 5);
            const float t3 = time_us([&] { launch_fw<X3_KROWS_F32, X3_KROWS_F32, EPI_SLAB>(dwf, splits_w); }, 5);
            std::printf("  wide f32 kernel: fwd2 %8.1f (%.1f)  delta1 %8.1f (%.1f)  dw2 %8.1f (%.1f)\n", t1, flop / t1 * 1e-6, t2, flop / t2 * 1e-6, t3, flop / t3 * 1e-6);
        }
        std::printf("rows %zu, %d x %d   (us per launch, f32-equivalent TFLOP/s)\n", R, H1, H2);
        std::printf("  fwd2    f32 %8.1f (%.1f)   bf16x3 %8.1f (%.1f)\n", t_f2, flop / t_f2 * 1e-6, t_f2w, flop / t_f2w * 1e-6);
        std::printf("  delta1  f32 %8.1f (%.1f)   bf16x3 %8.1f (%.1f)\n", t_dl, flop / t_dl * 1e-6, t_dlw, flop / t_dlw * 1e-6);
        std::printf("  dw2     f32 %8.1f (%.1f)   bf16x3 %8.1f (%.1f)\n", t_dw, flop / t_dw * 1e-6, t_dww, flop / t_dww * 1e-6);
    }
    return 0;
}

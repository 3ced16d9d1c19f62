// SimpleNeuralNet (reference lib.rs:744-1060) on gfx950: batched forward, softmax /
// cross-entropy, backward, gradient accumulation and SGD update.
//
// All dense contractions go through one LDS-tiled GEMM kernel built on the exact-f32 matrix
// instruction v_mfma_f32_32x32x2_f32 (128x128 block tile, 4 wavefronts of 64x64, K tile 16,
// global->register prefetch of the next K tile while the current one is multiplied).  The
// activation / derivative is fused into the epilogue.  Weight gradients are contractions over
// the batch: they are split over the batch dimension into slabs and summed by a second
// kernel, so the result does not depend on scheduling order.
#include <mutex>
#include <new>

#include "internal.h"
#include "prof.h"
#include "streamz_hip.h"

namespace sz {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = BM + 4;  // LDS row stride in floats: keeps transposing stores <= 2-way
constexpr int kGemmThreads = 256;

enum AMode { A_ROWS = 0 /* A[m][k] */, A_KROWS = 1 /* A[k][m] */ };
enum BMode { B_KROWS = 0 /* B[k][n] */, B_ROWS = 1 /* B[n][k] */ };
enum Epi {
    EPI_BIAS_RELU = 0,  // C = relu(acc + bias[n])        (v > 0 ? v : 0, lib.rs:882)
    EPI_BIAS_TANH = 1,  // C = tanh(acc + bias[n])        (lib.rs:883)
    EPI_BIAS = 2,       // C = acc + bias[n]
    EPI_MUL_DTANH = 3,  // C = acc * (1 - aux^2)          (lib.rs:1034)
    EPI_MUL_DRELU = 4,  // C = acc * (aux > 0)            (lib.rs:1039-1040)
    EPI_SLAB = 5        // C[z] = acc                     (split-K partial)
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    int lda, ldb, ldc;
    const float* bias;
    const float* aux;
    int ldaux;
    int k_chunk;  // K range handled by one blockIdx.z
    int a_vec, b_vec;
};

struct Frag4 { float v[4]; };

__device__ __forceinline__ Frag4 load4(const float* base, int64_t row, int col, int ld, int row_max, int col_max,
                                       int vec)
{
    Frag4 r;
    if (row < row_max && vec && col + 3 < col_max) {
        const float4 q = *reinterpret_cast<const float4*>(base + row * ld + col);
        r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r.v[j] = (row < row_max && col + j < col_max) ? base[row * ld + col + j] : 0.0f;
    }
    return r;
}

// Tile whose global rows run along the tile dimension and whose columns run along k
// (A[m][k], B[n][k]): 128 rows x 16 k, stored transposed (k-major) into LDS.
__device__ __forceinline__ void fetch_rows(const float* base, int ld, int r0, int r_max, int k0, int k_max, int vec,
                                           Frag4 (&f)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        f[i] = load4(base, r0 + (idx >> 2), k0 + 4 * (idx & 3), ld, r_max, k_max, vec);
    }
}
__device__ __forceinline__ void stash_rows(float* S, const Frag4 (&f)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        const int row = idx >> 2, kq = idx & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) S[(4 * kq + j) * LDT + row] = f[i].v[j];
    }
}
// Tile whose global rows run along k (A[k][m], B[k][n]): 16 k x 128, copied straight.
__device__ __forceinline__ void fetch_krows(const float* base, int ld, int c0, int c_max, int k0, int k_max, int vec,
                                            Frag4 (&f)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        f[i] = load4(base, k0 + (idx >> 5), c0 + 4 * (idx & 31), ld, k_max, c_max, vec);
    }
}
__device__ __forceinline__ void stash_krows(float* S, const Frag4 (&f)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        float4 q;
        q.x = f[i].v[0]; q.y = f[i].v[1]; q.z = f[i].v[2]; q.w = f[i].v[3];
        *reinterpret_cast<float4*>(S + (idx >> 5) * LDT + 4 * (idx & 31)) = q;
    }
}

// TAG only separates instantiations, so that each call site is its own kernel in a profile.
template <int AM, int BMd, int EPI, int TAG>
__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) float As[BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    Frag4 fa[2], fb[2];
    auto fetch = [&](int k0) {
        if (AM == A_ROWS) fetch_rows(g.A, g.lda, m0, g.M, k0, k_end, g.a_vec, fa);
        else fetch_krows(g.A, g.lda, m0, g.M, k0, k_end, g.a_vec, fa);
        if (BMd == B_ROWS) fetch_rows(g.B, g.ldb, n0, g.N, k0, k_end, g.b_vec, fb);
        else fetch_krows(g.B, g.ldb, n0, g.N, k0, k_end, g.b_vec, fb);
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();
        if (AM == A_ROWS) stash_rows(As, fa); else stash_krows(As, fa);
        if (BMd == B_ROWS) stash_rows(Bs, fb); else stash_krows(Bs, fb);
        __syncthreads();
        if (k0 + BK < k_end) fetch(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float* ar = As + (kk + half) * LDT + wm * 64 + l31;
            const float* br = Bs + (kk + half) * LDT + wn * 64 + l31;
            const float a0 = ar[0], a1 = ar[32];
            const float b0 = br[0], b1 = br[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    float* C = g.C;
    if (EPI == EPI_SLAB) C += (size_t)blockIdx.z * (size_t)g.M * (size_t)g.ldc;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wn * 64 + tn * 32 + l31;
            if (col >= g.N) continue;
            float bias = 0.0f;
            if (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_TANH || EPI == EPI_BIAS) bias = g.bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= g.M) continue;
                float v = acc[tm][tn][r];
                if (EPI == EPI_BIAS_RELU) { v += bias; v = v > 0.0f ? v : 0.0f; }
                else if (EPI == EPI_BIAS_TANH) v = tanhf(v + bias);
                else if (EPI == EPI_BIAS) v += bias;
                else if (EPI == EPI_MUL_DTANH) { const float h = g.aux[(size_t)row * g.ldaux + col]; v *= (1.0f - h * h); }
                else if (EPI == EPI_MUL_DRELU) { const float h = g.aux[(size_t)row * g.ldaux + col]; v *= (h > 0.0f ? 1.0f : 0.0f); }
                C[(size_t)row * g.ldc + col] = v;
            }
        }
}

inline int vec_ok(const void* p, int ld) { return (((uintptr_t)p & 15) == 0 && (ld & 3) == 0) ? 1 : 0; }

enum GemmTag { T_FWD1 = 0, T_FWD2, T_FWD3, T_DELTA2, T_DELTA1, T_DW3, T_DW2, T_DW1 };
const char* const kGemmTagName[] = {"fwd1", "fwd2", "fwd3", "delta2", "delta1", "dw3", "dw2", "dw1"};

template <int AM, int BMd, int EPI, int TAG>
int launch_gemm(GemmArgs g, int splits, hipStream_t s)
{
    if (g.M <= 0 || g.N <= 0) return SZH_OK;
    g.a_vec = vec_ok(g.A, g.lda);
    g.b_vec = vec_ok(g.B, g.ldb);
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits);
    {
        ProfScope prof(kGemmTagName[TAG], s);
        hipLaunchKernelGGL((gemm_kernel<AM, BMd, EPI, TAG>), grid, dim3(kGemmThreads), 0, s, g);
    }
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

// ---- counter-based random numbers ----------------------------------------------------
__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// 24-bit uniform in [0, 1), like rand's gen::<f32>()
__host__ __device__ inline float uniform01(uint64_t seed, uint64_t a, uint64_t b)
{
    const uint64_t h = mix64(mix64(seed ^ (a * 0xD1342543DE82EF95ull)) + b);
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}

// ---- row preparation: gather by permutation, input dropout, all-zero flag -------------
// (lib.rs:604-609: clone, apply_dropout, skip rows whose 60 values are all exactly 0.0)
__global__ void prep_rows_kernel(const float* __restrict__ x, const uint32_t* __restrict__ perm,
                                 const uint8_t* __restrict__ keep, float dropout, uint64_t seed, uint64_t counter0,
                                 uint64_t rows, int in, int skip_zero, float* __restrict__ xd, uint8_t* __restrict__ live)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t src = perm ? perm[r] : r;
    const float* xi = x + src * in;
    float* xo = xd + r * in;
    bool nz = false;
    for (int j = 0; j < in; ++j) {
        float v = xi[j];
        bool drop = false;
        if (keep) drop = keep[r * in + j] == 0;
        else if (dropout > 0.0f) drop = uniform01(seed, counter0 + r, (uint64_t)j) < dropout;
        if (drop) v = 0.0f;
        nz = nz || (v != 0.0f);
        xo[j] = v;
    }
    live[r] = (nz || !skip_zero) ? 1 : 0;
}

// ---- softmax, loss, output delta -----------------------------------------------------
// One thread per row.  mode 0: logits -> probabilities (lib.rs:887-890).
// mode 1: also loss = -sum_j t_j ln(max(o_j, 1e-12)) (lib.rs:611-615) added to stats[0] and
// the row counted in stats[1] when live; the row becomes delta_out = o - t (lib.rs:1028),
// or zeros when the row is not live.
__global__ void softmax_kernel(float* __restrict__ P, uint64_t rows, int C, int mode, const float* __restrict__ target,
                               const int32_t* __restrict__ labels, const uint8_t* __restrict__ live,
                               double* __restrict__ stats)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float loss = 0.0f;
    float cnt = 0.0f;
    if (r < rows) {
        float* p = P + r * C;
        float mx = -INFINITY;
        for (int j = 0; j < C; ++j) mx = fmaxf(mx, p[j]);
        float sum = 0.0f;
        for (int j = 0; j < C; ++j) { const float e = expf(p[j] - mx); p[j] = e; sum += e; }
        if (mode == 0) {
            for (int j = 0; j < C; ++j) p[j] = p[j] / sum;
        } else {
            const bool alive = live ? live[r] != 0 : true;
            const int lab = labels ? labels[r] : -1;
            for (int j = 0; j < C; ++j) {
                const float o = p[j] / sum;
                const float t = labels ? (j == lab ? 1.0f : 0.0f) : target[j];
                if (t != 0.0f) loss -= t * logf(fmaxf(o, 1e-12f));
                p[j] = alive ? o - t : 0.0f;
            }
            if (alive) cnt = 1.0f; else loss = 0.0f;
        }
    }
    if (mode == 0) return;
    __shared__ double s_loss[256];
    __shared__ double s_cnt[256];
    s_loss[threadIdx.x] = (double)loss;
    s_cnt[threadIdx.x] = (double)cnt;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            s_loss[threadIdx.x] += s_loss[threadIdx.x + st];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_cnt[0] > 0.0) {
        atomicAdd(&stats[0], s_loss[0]);
        atomicAdd(&stats[1], s_cnt[0]);
    }
}

// Per-window vote of identify_speaker_list (lib.rs:1393-1401) on probability rows.
__global__ void vote_kernel(const float* __restrict__ P, uint64_t rows, int C, float threshold,
                            unsigned long long* __restrict__ counts)
{
    extern __shared__ unsigned int hist[];
    for (int j = threadIdx.x; j < C; j += blockDim.x) hist[j] = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows && C > 0) {
        const float* p = P + r * C;
        int best = 0;
        float bv = p[0];
        for (int j = 1; j < C; ++j)
            if (p[j] >= bv) { bv = p[j]; best = j; }  // the last maximum wins
        if (bv >= threshold) atomicAdd(&hist[best], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < C; j += blockDim.x)
        if (hist[j]) atomicAdd(&counts[j], (unsigned long long)hist[j]);
}

// ---- column sums (bias gradients), split over rows into slabs ------------------------
__global__ void colsum_kernel(const float* __restrict__ D, uint64_t rows, int cols, int ld, int rows_per_split,
                              float* __restrict__ slab)
{
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.y * rows_per_split;
    const uint64_t r1 = min(rows, r0 + (uint64_t)rows_per_split);
    float s = 0.0f;
    if (c < cols)
        for (uint64_t r = r0 + rg; r < r1; r += 4) s += D[r * ld + c];
    part[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < cols)
        slab[(size_t)blockIdx.y * cols + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) +
                                             (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// grad[i] += sum_z slab[z][i]
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int splits, uint64_t len, float* __restrict__ grad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    float s = 0.0f;
    for (int z = 0; z < splits; ++z) s += slab[(size_t)z * len + i];
    grad[i] += s;
}

// stats = {loss, live rows} since the last zero_grad, then {loss, live rows} running totals
__global__ void stats_fold_kernel(double* __restrict__ stats)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stats[2] += stats[0];
        stats[3] += stats[1];
        stats[0] = 0.0;
        stats[1] = 0.0;
    }
}

__global__ void stats_to_tail_kernel(const double* __restrict__ stats, float* __restrict__ tail)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        tail[0] = (float)stats[0];
        tail[1] = (float)stats[1];
    }
}

// params -= (lr / live_rows) * grad  (lib.rs:1047-1059); nothing happens for an empty batch
__global__ void apply_kernel(float* __restrict__ params, const float* __restrict__ grad, uint64_t n, float lr)
{
    const float live = grad[n + 1];
    if (!(live > 0.0f)) return;
    const float scale = lr / live;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) params[i] -= grad[i] * scale;
}

constexpr uint64_t kChunkRows = 32768;

int ensure_workspace(Net* n, uint64_t rows)
{
    Workspace& w = n->ws;
    if (rows > kChunkRows) rows = kChunkRows;
    if (w.rows >= rows && w.out == n->out) return SZH_OK;
    float** ptrs[] = {&w.xd, &w.h1, &w.h2, &w.p, &w.d2, &w.d1, &w.slab};
    for (float** p : ptrs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (w.live) { (void)hipFree(w.live); w.live = nullptr; }
    w.rows = 0;
    const uint64_t R = rows < 256 ? 256 : rows;
    const uint64_t maxw = std::max<uint64_t>({(uint64_t)n->in * n->h1, (uint64_t)n->h1 * n->h2,
                                             (uint64_t)n->h2 * n->out, (uint64_t)n->h1});
    w.slab_floats = 64 * maxw;
    SZ_HIP(hipMalloc((void**)&w.xd, R * n->in * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.h1, R * n->h1 * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.h2, R * n->h2 * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.p, R * (n->out ? n->out : 1) * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.d2, R * n->h2 * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.d1, R * n->h1 * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.slab, w.slab_floats * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.live, R));
    w.rows = R;
    w.out = n->out;
    return SZH_OK;
}

// hidden layers of `rows` rows (rows <= workspace capacity): h1, h2 in the workspace
int forward_hidden(Net* n, const float* x, uint64_t rows, bool relu2, hipStream_t s)
{
    Workspace& w = n->ws;
    const float* P = n->d_params;
    GemmArgs g{};
    g.A = x; g.lda = n->in; g.B = P + n->off_w1(); g.ldb = n->h1; g.C = w.h1; g.ldc = n->h1;
    g.M = (int)rows; g.N = n->h1; g.K = n->in; g.k_chunk = n->in; g.bias = P + n->off_b1();
    int rc = launch_gemm<A_ROWS, B_KROWS, EPI_BIAS_RELU, T_FWD1>(g, 1, s);
    if (rc) return rc;
    g = GemmArgs{};
    g.A = w.h1; g.lda = n->h1; g.B = P + n->off_w2(); g.ldb = n->h2; g.C = w.h2; g.ldc = n->h2;
    g.M = (int)rows; g.N = n->h2; g.K = n->h1; g.k_chunk = n->h1; g.bias = P + n->off_b2();
    return relu2 ? launch_gemm<A_ROWS, B_KROWS, EPI_BIAS_RELU, T_FWD2>(g, 1, s)
                 : launch_gemm<A_ROWS, B_KROWS, EPI_BIAS_TANH, T_FWD2>(g, 1, s);
}

int forward_logits(Net* n, uint64_t rows, float* out, hipStream_t s)
{
    Workspace& w = n->ws;
    const float* P = n->d_params;
    GemmArgs g{};
    g.A = w.h2; g.lda = n->h2; g.B = P + n->off_w3(); g.ldb = n->out; g.C = out; g.ldc = n->out;
    g.M = (int)rows; g.N = n->out; g.K = n->h2; g.k_chunk = n->h2; g.bias = P + n->off_b3();
    return launch_gemm<A_ROWS, B_KROWS, EPI_BIAS, T_FWD3>(g, 1, s);
}

template <int TAG>
int grad_gemm(Net* n, const float* A, int lda, const float* B, int ldb, int M, int N, uint64_t rows, float* grad,
              hipStream_t s)
{
    Workspace& w = n->ws;
    int splits = (int)((rows + 511) / 512);
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    int k_chunk = (int)((rows + splits - 1) / splits);
    k_chunk = (k_chunk + BK - 1) / BK * BK;
    splits = (int)((rows + k_chunk - 1) / k_chunk);
    GemmArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = w.slab; g.ldc = N;
    g.M = M; g.N = N; g.K = (int)rows; g.k_chunk = k_chunk;
    int rc = launch_gemm<A_KROWS, B_KROWS, EPI_SLAB, TAG>(g, splits, s);
    if (rc) return rc;
    const uint64_t len = (uint64_t)M * N;
    {
        ProfScope prof("reduce", s);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, w.slab, splits,
                           len, grad);
    }
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

int grad_bias(Net* n, const float* D, int cols, uint64_t rows, float* grad, hipStream_t s)
{
    Workspace& w = n->ws;
    int splits = (int)((rows + 511) / 512);
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    const int rps = (int)((rows + splits - 1) / splits);
    splits = (int)((rows + rps - 1) / rps);
    {
        ProfScope prof("bias", s);
        hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, splits), dim3(256), 0, s, D, rows, cols, cols, rps,
                           w.slab);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, w.slab, splits,
                           (uint64_t)cols, grad);
    }
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

}  // namespace

float init_uniform(uint64_t seed, uint64_t index) { return uniform01(seed, index, 0x5EEDull) - 0.5f; }

int net_create(uint32_t in, uint32_t h1, uint32_t h2, uint32_t out, const float* init, uint64_t seed, Net** net)
{
    if (in == 0 || h1 == 0 || h2 == 0) { set_error("layer sizes must be positive"); return SZH_EINVAL; }
    Net* n = new (std::nothrow) Net();
    if (!n) { set_error("out of host memory"); return SZH_ENOMEM; }
    n->in = in; n->h1 = h1; n->h2 = h2; n->out = out;
    n->file_lists.resize(out);
    const uint64_t np = n->param_count();
    hipError_t he = hipGetDevice(&n->device);
    if (he == hipSuccess) he = hipMalloc((void**)&n->d_params, (np ? np : 1) * sizeof(float));
    if (he == hipSuccess) he = hipMalloc((void**)&n->d_grad, (np + 2) * sizeof(float));
    if (he == hipSuccess) he = hipMalloc((void**)&n->d_stats, 4 * sizeof(double));
    if (he == hipSuccess) he = hipMemset(n->d_grad, 0, (np + 2) * sizeof(float));
    if (he == hipSuccess) he = hipMemset(n->d_stats, 0, 4 * sizeof(double));
    if (he != hipSuccess) { net_destroy(n); return hip_fail(he, "net allocation"); }
    int rc;
    if (init) {
        rc = net_set_params(n, init);
    } else {
        // weights ~ U(-0.5, 0.5), biases 0 (lib.rs:769-781)
        std::vector<float> p(np, 0.0f);
        uint64_t ctr = 0;
        for (uint64_t i = 0; i < (uint64_t)in * h1; ++i) p[n->off_w1() + i] = init_uniform(seed, ctr++);
        for (uint64_t i = 0; i < (uint64_t)h1 * h2; ++i) p[n->off_w2() + i] = init_uniform(seed, ctr++);
        for (uint64_t i = 0; i < (uint64_t)h2 * out; ++i) p[n->off_w3() + i] = init_uniform(seed, ctr++);
        rc = net_set_params(n, p.data());
    }
    if (rc) { net_destroy(n); return rc; }
    *net = n;
    return SZH_OK;
}

void net_destroy(Net* n)
{
    if (!n) return;
    Workspace& w = n->ws;
    float* ptrs[] = {w.xd, w.h1, w.h2, w.p, w.d2, w.d1, w.slab, n->d_params, n->d_grad};
    for (float* p : ptrs) if (p) (void)hipFree(p);
    if (w.live) (void)hipFree(w.live);
    if (n->d_stats) (void)hipFree(n->d_stats);
    delete n;
}

int net_set_params(Net* n, const float* flat)
{
    if (n->param_count() == 0) return SZH_OK;
    SZ_HIP(hipMemcpy(n->d_params, flat, n->param_count() * sizeof(float), hipMemcpyHostToDevice));
    return SZH_OK;
}
int net_get_params(Net* n, float* flat)
{
    if (n->param_count() == 0) return SZH_OK;
    SZ_HIP(hipDeviceSynchronize());
    SZ_HIP(hipMemcpy(flat, n->d_params, n->param_count() * sizeof(float), hipMemcpyDeviceToHost));
    return SZH_OK;
}

int net_add_output_class(Net* n, const float* col, uint64_t seed)
{
    std::vector<float> oldp(n->param_count());
    int rc = net_get_params(n, oldp.data());
    if (rc) return rc;
    const uint32_t oc = n->out, nc = n->out + 1;
    const uint64_t ow3 = n->off_w3(), ob3 = n->off_b3();
    std::vector<float> np_(ow3 + (uint64_t)n->h2 * nc + nc, 0.0f);
    std::copy(oldp.begin(), oldp.begin() + ow3, np_.begin());
    for (uint32_t r = 0; r < n->h2; ++r) {
        for (uint32_t c = 0; c < oc; ++c) np_[ow3 + (uint64_t)r * nc + c] = oldp[ow3 + (uint64_t)r * oc + c];
        np_[ow3 + (uint64_t)r * nc + oc] = col ? col[r] : init_uniform(seed, r);
    }
    const uint64_t nb3 = ow3 + (uint64_t)n->h2 * nc;
    for (uint32_t c = 0; c < oc; ++c) np_[nb3 + c] = oldp[ob3 + c];
    float *dp = nullptr, *dg = nullptr;
    SZ_HIP(hipMalloc((void**)&dp, np_.size() * sizeof(float)));
    hipError_t he = hipMalloc((void**)&dg, (np_.size() + 2) * sizeof(float));
    if (he != hipSuccess) { (void)hipFree(dp); return hip_fail(he, "grad realloc"); }
    (void)hipFree(n->d_params);
    (void)hipFree(n->d_grad);
    n->d_params = dp;
    n->d_grad = dg;
    n->out = nc;
    if (n->file_lists.size() < nc) n->file_lists.resize(nc);  // lib.rs:817-819
    SZ_HIP(hipMemcpy(dp, np_.data(), np_.size() * sizeof(float), hipMemcpyHostToDevice));
    SZ_HIP(hipMemset(dg, 0, (np_.size() + 2) * sizeof(float)));
    return SZH_OK;
}

int net_forward_dev(Net* n, const float* x_dev, uint64_t rows, float* out_dev, ForwardKind kind, hipStream_t s)
{
    if (rows == 0) return SZH_OK;
    int rc = ensure_workspace(n, rows);
    if (rc) return rc;
    const uint32_t ow = kind == kFwdProbs ? n->out : n->h2;
    for (uint64_t r0 = 0; r0 < rows; r0 += kChunkRows) {
        const uint64_t R = std::min<uint64_t>(kChunkRows, rows - r0);
        rc = forward_hidden(n, x_dev + r0 * n->in, R, kind == kFwdEmbedRelu, s);
        if (rc) return rc;
        float* o = out_dev + r0 * ow;
        if (kind == kFwdProbs) {
            if (n->out == 0) continue;
            rc = forward_logits(n, R, o, s);
            if (rc) return rc;
            hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, o, R, (int)n->out, 0,
                               (const float*)nullptr, (const int32_t*)nullptr, (const uint8_t*)nullptr,
                               (double*)nullptr);
            SZ_HIP(hipGetLastError());
        } else {
            SZ_HIP(hipMemcpyAsync(o, n->ws.h2, R * n->h2 * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    return SZH_OK;
}

int net_zero_grad(Net* n, hipStream_t s)
{
    SZ_HIP(hipMemsetAsync(n->d_grad, 0, (n->param_count() + 2) * sizeof(float), s));
    hipLaunchKernelGGL(stats_fold_kernel, dim3(1), dim3(64), 0, s, n->d_stats);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

int net_reset_totals(Net* n, hipStream_t s)
{
    SZ_HIP(hipMemsetAsync(n->d_stats, 0, 4 * sizeof(double), s));
    return SZH_OK;
}

int net_read_totals(Net* n, hipStream_t s, double* loss_sum, double* live_rows)
{
    double h[4];
    SZ_HIP(hipMemcpyAsync(h, n->d_stats, sizeof(h), hipMemcpyDeviceToHost, s));
    SZ_HIP(hipStreamSynchronize(s));
    *loss_sum = h[0] + h[2];
    *live_rows = h[1] + h[3];
    return SZH_OK;
}

int net_accumulate(Net* n, const AccumArgs& a, hipStream_t s)
{
    if (a.rows == 0 || n->out == 0) return SZH_OK;
    int rc = ensure_workspace(n, a.rows);
    if (rc) return rc;
    Workspace& w = n->ws;
    const float* P = n->d_params;
    float* G = n->d_grad;
    for (uint64_t r0 = 0; r0 < a.rows; r0 += kChunkRows) {
        const uint64_t R = std::min<uint64_t>(kChunkRows, a.rows - r0);
        // gather / dropout / live flags
        {
            ProfScope prof("prep", s);
            hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s,
                           a.perm_dev ? a.x_dev : a.x_dev + r0 * n->in, a.perm_dev ? a.perm_dev + r0 : nullptr,
                           a.keep_dev ? a.keep_dev + r0 * n->in : nullptr, a.dropout, a.seed, a.row_counter0 + r0, R,
                           (int)n->in, a.skip_zero_rows ? 1 : 0, w.xd, w.live);
        }
        SZ_HIP(hipGetLastError());
        rc = forward_hidden(n, w.xd, R, false, s);
        if (rc) return rc;
        rc = forward_logits(n, R, w.p, s);
        if (rc) return rc;
        {
            ProfScope prof("softmax", s);
            hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, w.p, R, (int)n->out,
                               1, a.target_dev, a.labels_dev ? a.labels_dev + r0 : nullptr, (const uint8_t*)w.live,
                               n->d_stats);
        }
        SZ_HIP(hipGetLastError());
        // delta2 = (delta3 . w3^T) * (1 - h2^2)
        GemmArgs g{};
        g.A = w.p; g.lda = n->out; g.B = P + n->off_w3(); g.ldb = n->out; g.C = w.d2; g.ldc = n->h2;
        g.M = (int)R; g.N = n->h2; g.K = n->out; g.k_chunk = n->out; g.aux = w.h2; g.ldaux = n->h2;
        rc = launch_gemm<A_ROWS, B_ROWS, EPI_MUL_DTANH, T_DELTA2>(g, 1, s);
        if (rc) return rc;
        // delta1 = (delta2 . w2^T) * [h1 > 0]
        g = GemmArgs{};
        g.A = w.d2; g.lda = n->h2; g.B = P + n->off_w2(); g.ldb = n->h2; g.C = w.d1; g.ldc = n->h1;
        g.M = (int)R; g.N = n->h1; g.K = n->h2; g.k_chunk = n->h2; g.aux = w.h1; g.ldaux = n->h1;
        rc = launch_gemm<A_ROWS, B_ROWS, EPI_MUL_DRELU, T_DELTA1>(g, 1, s);
        if (rc) return rc;
        // weight gradients: contractions over the rows
        rc = grad_gemm<T_DW3>(n, w.h2, n->h2, w.p, n->out, n->h2, n->out, R, G + n->off_w3(), s);
        if (rc) return rc;
        rc = grad_bias(n, w.p, n->out, R, G + n->off_b3(), s);
        if (rc) return rc;
        rc = grad_gemm<T_DW2>(n, w.h1, n->h1, w.d2, n->h2, n->h1, n->h2, R, G + n->off_w2(), s);
        if (rc) return rc;
        rc = grad_bias(n, w.d2, n->h2, R, G + n->off_b2(), s);
        if (rc) return rc;
        rc = grad_gemm<T_DW1>(n, w.xd, n->in, w.d1, n->h1, n->in, n->h1, R, G + n->off_w1(), s);
        if (rc) return rc;
        rc = grad_bias(n, w.d1, n->h1, R, G + n->off_b1(), s);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(stats_to_tail_kernel, dim3(1), dim3(64), 0, s, n->d_stats, G + n->param_count());
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

int net_apply(Net* n, float lr, hipStream_t s)
{
    const uint64_t np = n->param_count();
    if (np == 0) return SZH_OK;
    {
        ProfScope prof("apply", s);
        hipLaunchKernelGGL(apply_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, n->d_params, n->d_grad,
                           np, lr);
    }
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

int net_vote_dev(Net* n, const float* x_dev, uint64_t rows, float threshold, unsigned long long* counts_dev,
                 hipStream_t s)
{
    if (rows == 0 || n->out == 0) return SZH_OK;
    int rc = ensure_workspace(n, rows);
    if (rc) return rc;
    for (uint64_t r0 = 0; r0 < rows; r0 += kChunkRows) {
        const uint64_t R = std::min<uint64_t>(kChunkRows, rows - r0);
        rc = forward_hidden(n, x_dev + r0 * n->in, R, false, s);
        if (rc) return rc;
        rc = forward_logits(n, R, n->ws.p, s);
        if (rc) return rc;
        hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, n->ws.p, R, (int)n->out,
                           0, (const float*)nullptr, (const int32_t*)nullptr, (const uint8_t*)nullptr,
                           (double*)nullptr);
        SZ_HIP(hipGetLastError());
        hipLaunchKernelGGL(vote_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), n->out * sizeof(unsigned int), s,
                           n->ws.p, R, (int)n->out, threshold, counts_dev);
        SZ_HIP(hipGetLastError());
    }
    return SZH_OK;
}

}  // namespace sz

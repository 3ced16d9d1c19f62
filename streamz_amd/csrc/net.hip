// SimpleNeuralNet (reference lib.rs:744-1060) on gfx950: batched forward, softmax /
// cross-entropy, backward, gradient accumulation and SGD update.
//
// All dense contractions go through one LDS-tiled GEMM kernel built on the exact-f32 matrix
// instruction v_mfma_f32_32x32x2_f32 (128x128 block tile, 4 wavefronts of 64x64, K tile 16,
// global->register prefetch of the next K tile while the current one is multiplied).  The
// activation / derivative is fused into the epilogue.  Weight gradients are contractions over
// the batch: they are split over the batch dimension into slabs and summed by a second
// kernel, so the result does not depend on scheduling order.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <new>

#include "gemm_bf16x3.h"
#include "gemm_kernel.h"
#include "infer_kernel.h"
#include "internal.h"
#include "prof.h"
#include "rng.h"
#include "streamz_hip.h"

namespace sz {

namespace {

using namespace gemm;

inline int vec_ok(const void* p, int ld) { return (((uintptr_t)p & 15) == 0 && (ld & 3) == 0) ? 1 : 0; }

constexpr int kGemmKT = 32, kGemmDB = 0;  // K tile and LDS double-buffering of the shipped kernels
constexpr int BK = kGemmKT;

enum GemmTag { T_FWD1 = 0, T_FWD2, T_FWD3, T_DELTA2, T_DELTA1, T_DW3, T_DW2, T_DW1 };
const char* const kGemmTagName[] = {"fwd1", "fwd2", "fwd3", "delta2", "delta1", "dw3", "dw2", "dw1"};

template <int AM, int BMd, int EPI, int TAG>
int launch_gemm(GemmArgs g, int splits, hipStream_t s)
{
    if (g.M <= 0 || g.N <= 0) return SZH_OK;
    g.a_vec = vec_ok(g.A, g.lda);
    g.b_vec = vec_ok(g.B, g.ldb);
    ProfScope prof(kGemmTagName[TAG], s);
    // 64-row tiles when 128-row tiles would leave compute units without a workgroup
    const long tiles128 = (long)((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM) * splits;
    if (g.M <= 64 && BMd == B_KROWS && g.N % (2 * BN) == 0 && (long)(g.N / (2 * BN)) * splits >= 256) {
        // one 64-row tile of rows and a k-major B: 256-column tiles (dW1)
        constexpr int BNW = BMd == B_KROWS ? 2 * BN : BN;
        dim3 grid(g.N / (2 * BN), 1, splits);
        hipLaunchKernelGGL((gemm_kernel<AM, BMd, EPI, TAG, 64, kGemmKT, kGemmDB, BNW>), grid, dim3(kGemmThreads), 0, s, g);
    } else if (g.M <= 64 || (splits == 1 && tiles128 < 256)) {
        dim3 grid((g.N + BN - 1) / BN, (g.M + 63) / 64, splits);
        hipLaunchKernelGGL((gemm_kernel<AM, BMd, EPI, TAG, 64, kGemmKT, kGemmDB>), grid, dim3(kGemmThreads), 0, s, g);
    } else {
        dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits);
        hipLaunchKernelGGL((gemm_kernel<AM, BMd, EPI, TAG, 128, kGemmKT, kGemmDB>), grid, dim3(kGemmThreads), 0, s, g);
    }
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

int compute_units()
{
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int v = 0;
        cus[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8 ? v : 256;
    }
    return cus[dev];
}

// The three large products of a training chunk on the bf16 matrix cores with every f32 operand
// split exactly into three bf16 pieces (gemm_bf16x3.h).
template <int ASRC, int BSRC, int EPI, int TAG>
int launch_gemm_x3(GemmArgs g, int splits, hipStream_t s)
{
    if (g.M <= 0 || g.N <= 0) return SZH_OK;
    if (int rc = allow_dynamic_lds((const void*)gemm_x3w_kernel<ASRC, BSRC, EPI, TAG>, XW_LDS_BYTES)) return rc;
    ProfScope prof(kGemmTagName[TAG], s);
    g.tiles_x = (g.N + XW_TN - 1) / XW_TN; g.tiles_y = (g.M + XW_TM - 1) / XW_TM; g.tiles_z = splits;
    // one workgroup per compute unit (its LDS use allows no second one), a multiple of the 8 XCDs
    const int tiles = g.tiles_x * g.tiles_y * g.tiles_z;
    const int blocks = std::min((tiles + 7) / 8 * 8, compute_units() / 8 * 8);
    hipLaunchKernelGGL((gemm_x3w_kernel<ASRC, BSRC, EPI, TAG>), dim3(blocks), dim3(XW_THREADS), XW_LDS_BYTES, s, g);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

// The f32 products with a row-major first operand on 256 x 256 tiles walked by one workgroup per CU
// (gemm_f32w_kernel): B as [N][K].
template <int EPI, int TAG, int SRC = X3_ROWS_F32>
int launch_gemm_f32w(GemmArgs g, hipStream_t s, int splits = 1)
{
    if (g.M <= 0 || g.N <= 0) return SZH_OK;
    if (int rc = allow_dynamic_lds((const void*)gemm_f32w_kernel<SRC, SRC, EPI, TAG>, FW_LDS_BYTES)) return rc;
    ProfScope prof(kGemmTagName[TAG], s);
    g.tiles_x = (g.N + XW_TN - 1) / XW_TN; g.tiles_y = (g.M + XW_TM - 1) / XW_TM; g.tiles_z = splits;
    const int tiles = g.tiles_x * g.tiles_y * g.tiles_z;
    const int blocks = std::min((tiles + 7) / 8 * 8, compute_units() / 8 * 8);
    hipLaunchKernelGGL((gemm_f32w_kernel<SRC, SRC, EPI, TAG>), dim3(blocks), dim3(XW_THREADS), FW_LDS_BYTES, s, g);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

// whether the hidden-layer forward product and its delta take that kernel: whole 256-wide tiles in both
// hidden dimensions and at least two rounds of tiles per CU (below that the 128-wide tiles, of which
// three workgroups fit a CU, are faster)
inline bool use_wide_f32(const Net* n, uint64_t rows)
{
    return n->h1 % 256 == 0 && n->h2 % 256 == 0 && (rows / 256) * (std::min(n->h1, n->h2) / 256) >= 512;
}

// whether a call of `rows` rows takes the split-bf16 kernels: opted in, whole 256-wide tiles in both
// hidden dimensions, and enough rows that 256-row tiles are not mostly padding
inline bool use_x3(const Net* n, uint64_t rows)
{
    return n->gemm_mode == SZH_GEMM_BF16X3 && n->h1 % 256 == 0 && n->h2 % 256 == 0 && rows >= 1024;
}

// ... and the output layer's products too (they run as GEMMs only beyond 64 classes)
inline bool use_x3_out(const Net* n, uint64_t rows) { return use_x3(n, rows) && n->out % 256 == 0; }

// ---- row preparation: gather by permutation, input dropout, all-zero flag -------------
// (lib.rs:604-609: clone, apply_dropout, skip rows whose 60 values are all exactly 0.0)
// One workgroup prepares 64 rows, one element per thread and step, so that reads and writes are
// contiguous; the per-row "any nonzero" flags are gathered in LDS.
constexpr int kPrepRows = 64;
__global__ __launch_bounds__(256) void prep_rows_kernel(const float* __restrict__ x, const uint32_t* __restrict__ perm,
                                                        const uint8_t* __restrict__ keep, float dropout, uint64_t seed,
                                                        uint64_t counter0, uint64_t rows, int in, int skip_zero,
                                                        float* __restrict__ xd, int ldx, uint8_t* __restrict__ live)
{
    __shared__ unsigned int nz[kPrepRows];
    __shared__ uint64_t row_key[kPrepRows];  // the row's half of uniform01's hash (rng.h)
    const uint64_t r0 = (uint64_t)blockIdx.x * kPrepRows;
    const int nr = (int)min((uint64_t)kPrepRows, rows - r0);
    const bool draw = !keep && dropout > 0.0f;
    if (threadIdx.x < kPrepRows) {
        nz[threadIdx.x] = 0;
        if (draw) row_key[threadIdx.x] = mix64(seed ^ ((counter0 + r0 + threadIdx.x) * 0xD1342543DE82EF95ull));
    }
    __syncthreads();
    const int n = nr * in;
    // element e = (row rl, column j); a step of 256 elements moves both without a division
    const int step_r = 256 / in, step_j = 256 % in;
    int rl = (int)threadIdx.x / in, j = (int)threadIdx.x % in;
    for (int e = threadIdx.x; e < n; e += 256) {
        const uint64_t r = r0 + rl;
        const uint64_t src = perm ? perm[r] : r;
        float v = x[src * in + j];
        bool drop = false;
        if (keep) drop = keep[r * in + j] == 0;
        else if (draw)  // uniform01(seed, counter0 + r, j) < dropout
            drop = (float)(mix64(row_key[rl] + (uint64_t)j) >> 40) * (1.0f / 16777216.0f) < dropout;
        if (drop) v = 0.0f;
        if (v != 0.0f) nz[rl] = 1;  // benign race: every writer stores the same value
        xd[r * ldx + j] = v;
        rl += step_r; j += step_j;
        if (j >= in) { j -= in; ++rl; }
    }
    // column `in` is 1 (it multiplies the bias gradient row of dW1), the rest of the padding 0
    const int npad = ldx - in;
    for (int e = threadIdx.x; e < nr * npad; e += 256) {
        const int rl = e / npad, j = e - rl * npad;
        xd[(r0 + rl) * ldx + in + j] = j == 0 ? 1.0f : 0.0f;
    }
    __syncthreads();
    if ((int)threadIdx.x < nr) live[r0 + threadIdx.x] = (nz[threadIdx.x] || !skip_zero) ? 1 : 0;
}

// The same for rows of a multiple of four inputs that are padded to 64 floats: one thread per group of
// four columns, a row per 16 consecutive lanes, so that every access is 16 bytes wide, rows are written
// whole and the "any nonzero" flag of a row is 16 bits of a ballot.
__global__ __launch_bounds__(256) void prep_rows4_kernel(const float* __restrict__ x, const uint32_t* __restrict__ perm,
                                                         const uint8_t* __restrict__ keep, float dropout, uint64_t seed,
                                                         uint64_t counter0, uint64_t rows, int in, int skip_zero,
                                                         float* __restrict__ xd, uint8_t* __restrict__ live)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t r = t >> 4;
    const int j = 4 * (int)(t & 15);
    const bool in_row = r < rows;
    float4 v = make_float4(j == in ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);  // column `in` is 1, the rest of the padding 0
    if (in_row && j < in) {
        const uint64_t src = perm ? perm[r] : r;
        v = *reinterpret_cast<const float4*>(x + src * in + j);
        float* e = &v.x;
        if (keep) {
            const uint32_t k4 = *reinterpret_cast<const uint32_t*>(keep + r * in + j);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (((k4 >> (8 * q)) & 0xffu) == 0) e[q] = 0.0f;
        } else if (dropout > 0.0f) {
            const uint64_t key = mix64(seed ^ ((counter0 + r) * 0xD1342543DE82EF95ull));  // uniform01(seed, counter0 + r, j + q)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((float)(mix64(key + (uint64_t)(j + q)) >> 40) * (1.0f / 16777216.0f) < dropout) e[q] = 0.0f;
        }
    }
    const bool nonzero = in_row && j < in && (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f);
    const unsigned long long any = __ballot(nonzero);
    if (!in_row) return;
    *reinterpret_cast<float4*>(xd + r * 64 + j) = v;
    if (j == 0) live[r] = (((any >> (threadIdx.x & 48)) & 0xffffull) || !skip_zero) ? 1 : 0;
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

// The same for wide outputs: one wavefront per row, lane j takes classes j, j + 64, ... (coalesced),
// max and sum by wave reductions.  Rows are dealt to the waves of the grid round-robin.
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
constexpr int kSoftmaxWideMaxC = 1024;  // 16 classes per lane in registers
__global__ __launch_bounds__(256) void softmax_wide_kernel(float* __restrict__ P, uint64_t rows, int C, int mode,
                                                          const float* __restrict__ target,
                                                          const int32_t* __restrict__ labels,
                                                          const uint8_t* __restrict__ live, double* __restrict__ stats)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double loss_acc = 0.0, cnt_acc = 0.0;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave; r < rows; r += (uint64_t)gridDim.x * 4) {
        float* p = P + r * C;
        float v[kSoftmaxWideMaxC / 64];
        float mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < kSoftmaxWideMaxC / 64; ++q) {
            const int j = lane + 64 * q;
            v[q] = j < C ? p[j] : -INFINITY;
            mx = fmaxf(mx, v[q]);
        }
        mx = wave_max(mx);
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < kSoftmaxWideMaxC / 64; ++q) {
            v[q] = lane + 64 * q < C ? expf(v[q] - mx) : 0.0f;
            sum += v[q];
        }
        sum = wave_sum(sum);
        if (mode == 0) {
#pragma unroll
            for (int q = 0; q < kSoftmaxWideMaxC / 64; ++q)
                if (lane + 64 * q < C) p[lane + 64 * q] = v[q] / sum;
        } else {
            const bool alive = live ? live[r] != 0 : true;
            const int lab = labels ? labels[r] : -1;
            float loss = 0.0f;
#pragma unroll
            for (int q = 0; q < kSoftmaxWideMaxC / 64; ++q) {
                const int j = lane + 64 * q;
                if (j < C) {
                    const float o = v[q] / sum;
                    const float t = labels ? (j == lab ? 1.0f : 0.0f) : target[j];
                    if (t != 0.0f) loss -= t * logf(fmaxf(o, 1e-12f));
                    p[j] = alive ? o - t : 0.0f;
                }
            }
            loss = wave_sum(loss);
            if (alive) { loss_acc += (double)loss; cnt_acc += 1.0; }
        }
    }
    if (mode == 0) return;
    __shared__ double s_loss[4], s_cnt[4];
    if (lane == 0) { s_loss[wave] = loss_acc; s_cnt[wave] = cnt_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double c = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        if (c > 0.0) {
            atomicAdd(&stats[0], (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]));
            atomicAdd(&stats[1], c);
        }
    }
}

// softmax over the rows of P, by the kernel that suits the width
static int launch_softmax(float* P, uint64_t rows, int C, int mode, const float* target, const int32_t* labels,
                          const uint8_t* live, double* stats, hipStream_t s)
{
    if (C > 32 && C <= kSoftmaxWideMaxC) {
        const unsigned blocks = (unsigned)std::min<uint64_t>((rows + 3) / 4, 256 * 16);
        hipLaunchKernelGGL(softmax_wide_kernel, dim3(blocks), dim3(256), 0, s, P, rows, C, mode, target, labels, live, stats);
    } else {
        hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, P, rows, C, mode, target,
                           labels, live, stats);
    }
    SZ_HIP(hipGetLastError());
    return SZH_OK;
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

// ---- output head for small class counts ------------------------------------------------
// For C <= 32 and hidden2 <= 256 the three contractions around the output layer are far too
// thin for 128-wide matrix tiles, so one kernel does them with vector FMAs on 32-row tiles
// held in LDS: logits = h2.w3 + b3, softmax, loss, delta3 = o - t (lib.rs:1022-1028),
// dW3 += h2^T.delta3, db3 += sum delta3 (lib.rs:1029-1033), delta2 = (delta3.w3^T)(1 - h2^2)
// (lib.rs:1034), db2 += sum delta2 (lib.rs:1038).  delta2 goes to global memory; the gradient
// partials of each workgroup go to a slab that slab_reduce_kernel sums.
constexpr int kHeadRows = 32;
constexpr int kHeadThreads = 256;

struct HeadArgs {
    const float* h2;       // [rows][H2]
    const float* w3;       // [H2][C]
    const float* w3tab;    // head_mfma_kernel only: output of head_tables_kernel
    const float* b3;       // [C]
    const float* target;   // [C] shared target, or
    const int32_t* labels; // [rows]
    const uint8_t* live;   // [rows]
    float* d2;             // [rows][H2] out
    float* d3;             // head_wide_kernel only: [rows][C] out, delta3
    float* slab_w3;        // [blocks][H2*C + C]  (dW3 then db3, the layout of the gradient vector)
    float* slab_b2;        // [blocks][H2]
    double* stats;         // [blocks][2] loss and live-row partials
    uint64_t rows;
    int H2, C, Cp;         // Cp = C rounded up to a multiple of 4
    int tiles_per_block;
    int add;               // add to the slabs instead of overwriting them
};

template <int CP>
__global__ __launch_bounds__(kHeadThreads) void head_kernel(HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int H2 = a.H2, C = a.C;
    constexpr int Cp = CP, Cq = CP / 4;
    const int LDH = H2 + 4;
    float* H2s = hs;                       // [32][H2+4]   h2 tile, later delta2 in place
    float* W3s = H2s + kHeadRows * LDH;    // [H2][Cp]     w3, zero-padded columns
    float* W3T = W3s + H2 * Cp;            // [Cp][H2+4]   w3 transposed
    float* D3s = W3T + Cp * LDH;           // [32][Cp]     logits -> delta3
    const int tid = threadIdx.x;

    for (int i = tid; i < H2 * Cp; i += kHeadThreads) {
        const int k = i / Cp, c = i - k * Cp;
        const float v = c < C ? a.w3[(size_t)k * C + c] : 0.0f;
        W3s[i] = v;
        W3T[c * LDH + k] = v;
    }
    float acc_w3[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc_w3[c] = 0.0f;
    float acc_b3 = 0.0f, acc_b2 = 0.0f;
    double loss_sum = 0.0, live_sum = 0.0;

    for (int t = 0; t < a.tiles_per_block; ++t) {
        const uint64_t r0 = ((uint64_t)blockIdx.x * a.tiles_per_block + t) * kHeadRows;
        if (r0 >= a.rows) break;
        const int nr = (int)min((uint64_t)kHeadRows, a.rows - r0);
        __syncthreads();  // previous tile fully consumed; W3 tables written
        // h2 tile -> LDS (rows past the end are zero)
        for (int i = tid; i < kHeadRows * (H2 / 4); i += kHeadThreads) {
            const int r = i / (H2 / 4), q = i - r * (H2 / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nr) v = *reinterpret_cast<const float4*>(a.h2 + (r0 + r) * H2 + 4 * q);
            *reinterpret_cast<float4*>(H2s + r * LDH + 4 * q) = v;
        }
        __syncthreads();
        // logits: 4 classes of one row per work item
        for (int o = tid; o < kHeadRows * Cq; o += kHeadThreads) {
            const int r = o / Cq, cg = o - r * Cq;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float* hr = H2s + r * LDH;
            const float* wr = W3s + 4 * cg;
#pragma unroll 8
            for (int k = 0; k < H2; ++k) {
                const float h = hr[k];
                const float4 w = *reinterpret_cast<const float4*>(wr + k * Cp);
                s0 += h * w.x; s1 += h * w.y; s2 += h * w.z; s3 += h * w.w;
            }
            float* d = D3s + r * Cp + 4 * cg;
            d[0] = s0; d[1] = s1; d[2] = s2; d[3] = s3;
        }
        __syncthreads();
        // softmax, loss and delta3: 8 lanes per row, classes s, s+8, s+16, s+24 each
        {
            const int r = tid >> 3, sl = tid & 7;
            float* p = D3s + r * Cp;
            const bool row_ok = r < nr;
            float v[4];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = sl + 8 * i;
                v[i] = (row_ok && j < C) ? p[j] + a.b3[j] : -INFINITY;
                mx = fmaxf(mx, v[i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 4));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = sl + 8 * i;
                v[i] = (row_ok && j < C) ? expf(v[i] - mx) : 0.0f;
                sum += v[i];
            }
            sum += __shfl_xor(sum, 4);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 1);
            const bool alive = row_ok && (a.live ? a.live[r0 + r] != 0 : true);
            const int lab = (row_ok && a.labels) ? a.labels[r0 + r] : -1;
            float loss = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = sl + 8 * i;
                if (j < Cp) {
                    float d = 0.0f;
                    if (row_ok && j < C) {
                        const float o = v[i] / sum;
                        const float tj = a.labels ? (j == lab ? 1.0f : 0.0f) : a.target[j];
                        if (tj != 0.0f) loss -= tj * logf(fmaxf(o, 1e-12f));
                        d = alive ? o - tj : 0.0f;
                    }
                    p[j] = d;
                }
            }
            loss += __shfl_xor(loss, 4);
            loss += __shfl_xor(loss, 2);
            loss += __shfl_xor(loss, 1);
            if (sl == 0 && alive) { loss_sum += (double)loss; live_sum += 1.0; }
        }
        __syncthreads();
        // dW3[k][c] += sum_r h2[r][k] delta3[r][c]   (thread = k), db3 (threads 0..C-1)
        if (tid < H2) {
#pragma unroll 4
            for (int r = 0; r < kHeadRows; ++r) {
                const float h = H2s[r * LDH + tid];
                const float* d = D3s + r * Cp;
#pragma unroll
                for (int cg = 0; cg < Cq; ++cg) {
                    const float4 dv = *reinterpret_cast<const float4*>(d + 4 * cg);
                    acc_w3[4 * cg + 0] += h * dv.x; acc_w3[4 * cg + 1] += h * dv.y;
                    acc_w3[4 * cg + 2] += h * dv.z; acc_w3[4 * cg + 3] += h * dv.w;
                }
            }
        }
        if (tid < C) {
            float sb = 0.0f;
            for (int r = 0; r < kHeadRows; ++r) sb += D3s[r * Cp + tid];
            acc_b3 += sb;
        }
        __syncthreads();
        // delta2 in place: 4 hidden units of one row per work item
        for (int o = tid; o < kHeadRows * (H2 / 4); o += kHeadThreads) {
            const int r = o / (H2 / 4), jq = o - r * (H2 / 4);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float* d = D3s + r * Cp;
#pragma unroll
            for (int c = 0; c < CP; ++c) {  // padded classes carry delta3 = 0 and w3 = 0
                const float dc = d[c];
                const float4 w = *reinterpret_cast<const float4*>(W3T + c * LDH + 4 * jq);
                s0 += dc * w.x; s1 += dc * w.y; s2 += dc * w.z; s3 += dc * w.w;
            }
            float4* hp = reinterpret_cast<float4*>(H2s + r * LDH + 4 * jq);
            const float4 h = *hp;
            *hp = make_float4(s0 * (1.0f - h.x * h.x), s1 * (1.0f - h.y * h.y), s2 * (1.0f - h.z * h.z),
                              s3 * (1.0f - h.w * h.w));
        }
        __syncthreads();
        // db2 and the coalesced store of delta2
        if (tid < H2) {
            float sb = 0.0f;
            for (int r = 0; r < kHeadRows; ++r) sb += H2s[r * LDH + tid];
            acc_b2 += sb;
        }
        for (int i = tid; i < nr * (H2 / 4); i += kHeadThreads) {
            const int r = i / (H2 / 4), q = i - r * (H2 / 4);
            *reinterpret_cast<float4*>(a.d2 + (r0 + r) * H2 + 4 * q) = *reinterpret_cast<const float4*>(H2s + r * LDH + 4 * q);
        }
    }
    // gradient partials of this workgroup
    float* sw = a.slab_w3 + (size_t)blockIdx.x * ((size_t)H2 * C + C);
    if (tid < H2) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) sw[(size_t)tid * C + c] = (a.add ? sw[(size_t)tid * C + c] : 0.0f) + acc_w3[c];
        float* sb = a.slab_b2 + (size_t)blockIdx.x * H2 + tid;
        *sb = (a.add ? *sb : 0.0f) + acc_b2;
    }
    if (tid < C) sw[(size_t)H2 * C + tid] = (a.add ? sw[(size_t)H2 * C + tid] : 0.0f) + acc_b3;
    // loss / live-row statistics: one atomic pair per workgroup
    for (int off = 32; off > 0; off >>= 1) {
        loss_sum += __shfl_down(loss_sum, off);
        live_sum += __shfl_down(live_sum, off);
    }
    __shared__ double s_stat[8];  // 64 bytes: keeps the dynamic LDS base 16-byte aligned
    if ((tid & 63) == 0) { s_stat[tid >> 6] = loss_sum; s_stat[4 + (tid >> 6)] = live_sum; }
    __syncthreads();
    if (tid == 0) {
        const double ls = (s_stat[0] + s_stat[1]) + (s_stat[2] + s_stat[3]);
        const double lv = (s_stat[4] + s_stat[5]) + (s_stat[6] + s_stat[7]);
        // per-workgroup partials (thousands of same-address atomics at the end of the kernel
        // would queue up at one L2 channel)
        double* st = a.stats + 2 * (size_t)blockIdx.x;
        st[0] = (a.add ? st[0] : 0.0) + ls;
        st[1] = (a.add ? st[1] : 0.0) + lv;
    }
}

// ---- output head on the matrix cores, for C <= 16 and hidden2 = 16 KB (KB = 4, 8 or 16) -------
// One wave owns a tile of 16 rows; v_mfma_f32_16x16x4_f32 throughout.  With lane = (m, g),
// m = lane % 16, g = lane / 16, an A operand is A[m][4g+s], a B operand is B[4g+s][m] and a
// result register v is D[4g+v][m] (s, v = 0..3; the contraction index of step s is 4g+s on both
// sides, which is all the instruction needs).  Hence
//   logits[row 4g+v][class m]   = sum_k  h2[row m][k] . w3[k][class m]           (A = h2, B = w3)
//   delta3 = softmax - target, kept as D and reused directly as a B operand whose contraction
//            index is the row:  dW3[k][class m] += sum_rows h2t[k][row] . delta3[row][class m]
//   h2t (rows in the g part) and delta3^T come from multiplying by an identity operand, which
//            moves a 16x16 block from the A layout to the D layout exactly,
//   delta2^T[k][row m] = sum_c w3[k][c] . delta3^T[c][row m], times 1 - h2^2 (lib.rs:1034),
//            stored as 16-byte pieces of delta2[row m][k..k+3].
// w3 sits in LDS in both operand layouts; the gradient partials of the 4 waves are summed in a
// fixed order and go to the slab of this workgroup.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMHeadBlocks = 512;

// value of the same register in another lane of the row of 16, by a DPP pattern
template <int CTRL>
__device__ __forceinline__ float dpp_read(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// all-reduce over each row of 16 lanes: lanes ^1, ^2 (quad permutes), then the mirrored half rows
__device__ __forceinline__ float row16_max(float x)
{
    x = fmaxf(x, dpp_read<0xB1>(x));
    x = fmaxf(x, dpp_read<0x4E>(x));
    x = fmaxf(x, dpp_read<0x141>(x));
    return fmaxf(x, dpp_read<0x140>(x));
}
__device__ __forceinline__ float row16_sum(float x)
{
    x += dpp_read<0xB1>(x);
    x += dpp_read<0x4E>(x);
    x += dpp_read<0x141>(x);
    return x + dpp_read<0x140>(x);
}

// Column sums of a 16 x 16 block held as float4 o = block[row m][4 g + 0..3] (m = lane & 15): a
// butterfly over the four lanes of a quad.  With its xor-1 partner a lane trades two of the four
// columns, with its xor-2 partner one of the two left, and ends with the sum over the quad's four rows
// of column 4 g + quad_col(m), quad_col = {0, 2, 1, 3}[m & 3].
__device__ __forceinline__ int quad_col(int m) { return ((m & 1) << 1) | ((m & 2) >> 1); }
__device__ __forceinline__ float quad_colsum(const float4& o, bool odd, bool hi)
{
    const float k0 = odd ? o.z : o.x, k1 = odd ? o.w : o.y;
    const float a0 = k0 + dpp_read<0xB1>(odd ? o.x : o.z);
    const float a1 = k1 + dpp_read<0xB1>(odd ? o.y : o.w);
    return (hi ? a1 : a0) + dpp_read<0x4E>(hi ? a0 : a1);
}

#define SZ_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

// ---- inference head: logits, softmax, per-window vote in one pass ------------------------
// identify_speaker_list's loop body (lib.rs:1391-1401) for up to 64 classes: logits of 16 rows
// per wave on the 16x16x4 matrix instruction (w3 padded to NCB blocks of 16 classes in LDS, h2
// read once as 16-byte pieces), softmax over the classes of a row with max subtraction
// (lib.rs:887-890) across the 16 lanes that hold the row, the last maximum wins, and the
// thresholded vote goes to a histogram in LDS that is added to counts at the end.
constexpr int kVoteThreads = 256;
constexpr size_t kVoteLdsMax = 144 * 1024;
__host__ __device__ constexpr int vote_ldw(int ncb) { return 16 * ncb + 4; }  // row stride of w3 in LDS
// w3 [H2][ldw] floats, vote histogram [16 ncb] u32, probability sums [16 ncb] f64 (8-byte aligned:
// the float count before them is even)
constexpr size_t vote_lds_bytes(int h2, int ncb) { return ((size_t)h2 * vote_ldw(ncb) + 16 * ncb) * sizeof(float) + 16 * ncb * sizeof(double); }

template <int NCB>
__global__ __launch_bounds__(kVoteThreads) void vote_head_kernel(const float* __restrict__ h2, uint64_t rows, int H2,
                                                                const float* __restrict__ w3,
                                                                const float* __restrict__ b3, int C, float threshold,
                                                                unsigned long long* __restrict__ counts,
                                                                double* __restrict__ psums)
{
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    constexpr int LDW = vote_ldw(NCB);
    float* W = vlds;                                                   // [H2][LDW]
    unsigned int* hist = reinterpret_cast<unsigned int*>(vlds + (size_t)H2 * LDW);  // [16 * NCB]
    double* dsum = reinterpret_cast<double*>(hist + 16 * NCB);                      // [16 * NCB]
    for (int e = threadIdx.x; e < H2 * 16 * NCB; e += kVoteThreads) {
        const int k = e / (16 * NCB), c = e - k * (16 * NCB);
        W[k * LDW + c] = c < C ? w3[(size_t)k * C + c] : 0.0f;
    }
    if (threadIdx.x < 16 * NCB) { hist[threadIdx.x] = 0; dsum[threadIdx.x] = 0.0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    float bias[NCB];
    double pacc[NCB];  // this lane's share of the per-class probability sums (lib.rs:1292-1296)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { bias[cb] = 16 * cb + n < C ? b3[16 * cb + n] : 0.0f; pacc[cb] = 0.0; }
    const uint64_t groups = (rows + 15) / 16;
    for (uint64_t g = (uint64_t)blockIdx.x * (kVoteThreads / 64) + wave; g < groups; g += (uint64_t)gridDim.x * (kVoteThreads / 64)) {
        const uint64_t row0 = g * 16;
        // operand A: lane (n, q) holds h2[row0 + n][16 step + 4 q + 0..3]
        const uint64_t ar = row0 + n < rows ? row0 + n : rows - 1;
        const float4* src = reinterpret_cast<const float4*>(h2 + ar * H2) + q;
        f32x4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const int steps = H2 / 16;
#pragma unroll 4
        for (int st = 0; st < steps; ++st) {
            const float4 a = src[4 * st];
            const float* wr = W + (16 * st + 4 * q) * LDW + n;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                acc[cb] = SZ_MFMA16(a.x, wr[16 * cb], acc[cb]);
                acc[cb] = SZ_MFMA16(a.y, wr[LDW + 16 * cb], acc[cb]);
                acc[cb] = SZ_MFMA16(a.z, wr[2 * LDW + 16 * cb], acc[cb]);
                acc[cb] = SZ_MFMA16(a.w, wr[3 * LDW + 16 * cb], acc[cb]);
            }
        }
        // result: register j of lane (n, q) is logit[row0 + 4 q + j][16 cb + n]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[NCB];
            float mx = -INFINITY;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                v[cb] = 16 * cb + n < C ? acc[cb][j] + bias[cb] : -INFINITY;
                mx = fmaxf(mx, v[cb]);
            }
            mx = row16_max(mx);
            float sum = 0.0f;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                v[cb] = 16 * cb + n < C ? expf(v[cb] - mx) : 0.0f;
                sum += v[cb];
            }
            sum = row16_sum(sum);
            const bool row_ok = row0 + 4 * q + j < rows;
            float best = -1.0f, best_c = -1.0f;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float p = v[cb] / sum;
                if (16 * cb + n < C && p >= best) { best = p; best_c = (float)(16 * cb + n); }
                if (row_ok) pacc[cb] += (double)p;
            }
            const float top = row16_max(best);
            const float win = row16_max(best == top ? best_c : -1.0f);  // the last maximum wins
            if (n == j && row_ok && top >= threshold && win >= 0.0f) atomicAdd(&hist[(int)win], 1u);
        }
    }
    if (psums) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            if (16 * cb + n < C) atomicAdd(&dsum[16 * cb + n], pacc[cb]);
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        if (counts && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
        if (psums) atomicAdd(&psums[threadIdx.x], dsum[threadIdx.x]);
    }
}

// per-class sums of probability rows, for shapes vote_head_kernel does not take
__global__ void prob_sum_kernel(const float* __restrict__ P, uint64_t rows, int C, double* __restrict__ psums)
{
    const uint64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const uint64_t r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double acc = 0.0;
        for (uint64_t r = r0; r < r1; ++r) acc += (double)P[r * C + c];
        if (r1 > r0) atomicAdd(&psums[c], acc);
    }
}

template <int NCB>
int launch_vote_head(const float* h2, uint64_t rows, int H2, const float* w3, const float* b3, int C, float threshold,
                     unsigned long long* counts, double* psums, hipStream_t s)
{
    const size_t lds = vote_lds_bytes(H2, NCB);
    if (int rc = allow_dynamic_lds((const void*)vote_head_kernel<NCB>, kVoteLdsMax)) return rc;
    const uint64_t groups = (rows + 15) / 16;
    const unsigned blocks = (unsigned)std::min<uint64_t>((groups + 3) / 4, 2048);
    ProfScope prof("vote_head", s);
    hipLaunchKernelGGL((vote_head_kernel<NCB>), dim3(blocks), dim3(kVoteThreads), lds, s, h2, rows, H2, w3, b3, C, threshold, counts, psums);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

// w3 in the two MFMA operand layouts the head kernels multiply with, as float4 per lane, for
// every block cb of 16 classes:  tab[((2 cb) KB + j) 64 + lane]     = w3[16j + 4g + 0..3][16cb + m],
//                                tab[((2 cb + 1) KB + j) 64 + lane] = w3[16j + m][16cb + 4g + 0..3],
// zero beyond column C.
__global__ void head_tables_kernel(const float* __restrict__ w3, int KB, int C, int ncb, float* __restrict__ tab)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ncb * KB * 64) return;
    const int cb = e / (KB * 64), r = e - cb * KB * 64;
    const int j = r >> 6, mm = r & 15, gg = (r & 63) >> 4;
    float t[4], u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t[q] = 16 * cb + mm < C ? w3[(size_t)(16 * j + 4 * gg + q) * C + 16 * cb + mm] : 0.0f;
        u[q] = 16 * cb + 4 * gg + q < C ? w3[(size_t)(16 * j + mm) * C + 16 * cb + 4 * gg + q] : 0.0f;
    }
    reinterpret_cast<float4*>(tab)[(2 * cb) * KB * 64 + r] = make_float4(t[0], t[1], t[2], t[3]);
    reinterpret_cast<float4*>(tab)[(2 * cb + 1) * KB * 64 + r] = make_float4(u[0], u[1], u[2], u[3]);
}

template <int KB>
__global__ __launch_bounds__(256, 2) void head_mfma_kernel(HeadArgs a)
{
    constexpr int H2 = KB * 16;
    __shared__ __attribute__((aligned(16))) float tab[2 * KB * 64 * 4];
    float4* Wb = reinterpret_cast<float4*>(tab);  // [KB][lane]: w3[16j + 4g + s][m], s = x..w
    float4* Wa = Wb + KB * 64;                    // [KB][lane]: w3[16j + m][4g + i], i = x..w
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: tile indices stay in scalar registers
    const int C = a.C;
    const bool cls = m < C;
    const float b3m = cls ? a.b3[m] : 0.0f;
    const float tgt_m = (a.target && cls) ? a.target[m] : 0.0f;
    f32x4 ident;
#pragma unroll
    for (int q = 0; q < 4; ++q) ident[q] = (4 * g + q == m) ? 1.0f : 0.0f;
    f32x4 acc3[KB];
    float accb2[KB];  // db2[16 j + 4 g + quad_col(m)], rows of the lane's quad only
    const bool odd = (m & 1) != 0, hi = (m & 2) != 0;
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        accb2[j] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc3[j][q] = 0.0f;
    }
    float accb3 = 0.0f;
    double loss_sum = 0.0;
    float live_cnt = 0.0f;  // exact: a lane sees far fewer than 2^24 rows

    {   // both operand layouts of w3, prepared by head_tables_kernel
        const float4* src = reinterpret_cast<const float4*>(a.w3tab);
        float4* dst = reinterpret_cast<float4*>(tab);
#pragma unroll
        for (int e = tid; e < 2 * KB * 64; e += 256) dst[e] = src[e];
    }
    // The h2 tile of a wave's next 16 rows is requested while the current one is still being
    // worked on (each register is refilled right after its last use), so that the memory latency is paid
    // once per wave.
    const uint64_t tiles = (a.rows + 15) / 16;
    const uint64_t t_step = (uint64_t)gridDim.x * 4;
    uint64_t t = (uint64_t)blockIdx.x * 4 + wave;
    float4 Ha[KB];
    {
        // rows past the end read row 0 instead: their delta3 is zero, so any finite value will do
        const bool ok0 = t < tiles && t * 16 + m < a.rows;
        const float* h0 = a.h2 + (ok0 ? t * 16 + m : 0) * H2 + 4 * g;
#pragma unroll
        for (int j = 0; j < KB; ++j) Ha[j] = *reinterpret_cast<const float4*>(h0 + 16 * j);
    }
    // labels and live flags of rows 4g..4g+3 travel with the tile (row indices clamped, so the
    // loads are unconditional and issue together)
    uint32_t lab4 = 0xffffffffu;  // one byte per row (C <= 16)
    uint32_t live4 = 0x01010101u;
    auto fetch_row_info = [&](uint64_t tile) {
        const uint64_t last = a.rows - 1;
        const uint64_t rb = min(tile * 16 + 4 * g, last);
        if (a.labels) {
            lab4 = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) lab4 |= ((uint32_t)a.labels[min(rb + v, last)] & 0xffu) << (8 * v);
        }
        if (a.live) {
            live4 = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) live4 |= (uint32_t)a.live[min(rb + v, last)] << (8 * v);
        }
    };
    fetch_row_info(t);
    __syncthreads();
    for (; t < tiles; t += t_step) {
        const uint64_t r0 = t * 16, rm = r0 + m;
        const bool okm = rm < a.rows;
        const bool okn = t + t_step < tiles && (t + t_step) * 16 + m < a.rows;
        const float* hnext = a.h2 + (okn ? (t + t_step) * 16 + m : 0) * H2 + 4 * g;
        f32x4 L = {0.f, 0.f, 0.f, 0.f};
        // The scheduling barriers keep the compiler from hoisting every operand read of the
        // unrolled loops to the top, which would need far more registers than there are.
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const float4 wb = Wb[j * 64 + lane];
            L = SZ_MFMA16(Ha[j].x, wb.x, L);
            L = SZ_MFMA16(Ha[j].y, wb.y, L);
            L = SZ_MFMA16(Ha[j].z, wb.z, L);
            L = SZ_MFMA16(Ha[j].w, wb.w, L);
            __builtin_amdgcn_sched_barrier(0);
        }
        // softmax over the 16 lanes that share g; loss and delta3 (lib.rs:1022-1028)
        f32x4 d3;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint64_t rv = r0 + 4 * g + v;
            const bool okv = rv < a.rows;
            const float x = cls ? L[v] + b3m : -INFINITY;
            const float mx = row16_max(x);
            const float e = cls ? expf(x - mx) : 0.0f;
            const float sum = row16_sum(e);
            const float o = e / sum;
            const bool alive = okv && ((live4 >> (8 * v)) & 0xffu) != 0;
            const float tj = a.labels ? (m == (int)((lab4 >> (8 * v)) & 0xffu) ? 1.0f : 0.0f) : tgt_m;
            float d = 0.0f;
            if (alive && cls) {
                if (tj != 0.0f) loss_sum -= (double)(tj * logf(fmaxf(o, 1e-12f)));
                d = o - tj;
            }
            if (alive && m == 0) live_cnt += 1.0f;
            d3[v] = d;
            accb3 += d;
        }
        f32x4 d3t = {0.f, 0.f, 0.f, 0.f};  // delta3[row m][class 4g+v]
#pragma unroll
        for (int v = 0; v < 4; ++v) d3t = SZ_MFMA16(d3[v], ident[v], d3t);
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const float4 h = Ha[j];
            // refill the registers whose value died with the previous block
            if (j > 0) Ha[j - 1] = *reinterpret_cast<const float4*>(hnext + 16 * (j - 1));
            const float4 wa = Wa[j * 64 + lane];
            f32x4 hb = {0.f, 0.f, 0.f, 0.f};  // h2[row 4g+v][16j + m]
            hb = SZ_MFMA16(h.x, ident[0], hb);
            hb = SZ_MFMA16(h.y, ident[1], hb);
            hb = SZ_MFMA16(h.z, ident[2], hb);
            hb = SZ_MFMA16(h.w, ident[3], hb);
            f32x4 q = {0.f, 0.f, 0.f, 0.f};   // (delta3 . w3^T)[row m][16j + 4g + v]
            q = SZ_MFMA16(wa.x, d3t[0], q);
            q = SZ_MFMA16(wa.y, d3t[1], q);
            q = SZ_MFMA16(wa.z, d3t[2], q);
            q = SZ_MFMA16(wa.w, d3t[3], q);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc3[j] = SZ_MFMA16(hb[v], d3[v], acc3[j]);
            const float4 o = make_float4(q[0] * (1.0f - h.x * h.x), q[1] * (1.0f - h.y * h.y),
                                         q[2] * (1.0f - h.z * h.z), q[3] * (1.0f - h.w * h.w));
            accb2[j] += quad_colsum(o, odd, hi);  // db2 (lib.rs:1038)
            if (okm) *reinterpret_cast<float4*>(a.d2 + rm * H2 + 16 * j + 4 * g) = o;
            __builtin_amdgcn_sched_barrier(0);
        }
        Ha[KB - 1] = *reinterpret_cast<const float4*>(hnext + 16 * (KB - 1));
        fetch_row_info(t + t_step);
    }

    // slab (+)= sums: the slab's current contents are requested now, ahead of the reductions, so
    // that their latency is hidden
    const int len3 = H2 * C + C;
    float* sw = a.slab_w3 + (size_t)blockIdx.x * len3;
    float* sb = a.slab_b2 + (size_t)blockIdx.x * H2;
    constexpr int kPer = (H2 * 16 + 16 + 255) / 256;  // len3 <= 16 (H2 + 1)
    float prev[kPer + 1];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int i = tid + 256 * q;
        prev[q] = (a.add && i < len3) ? sw[i] : 0.0f;
    }
    prev[kPer] = (a.add && tid < H2) ? sb[tid] : 0.0f;
    double st_prev[2] = {0.0, 0.0};
    if (a.add && tid == 0) { st_prev[0] = a.stats[2 * (size_t)blockIdx.x]; st_prev[1] = a.stats[2 * (size_t)blockIdx.x + 1]; }
    // sums over the rows held by different lane groups
#pragma unroll
    for (int j = 0; j < KB; ++j) {  // over the four quads of a row of 16 lanes
        accb2[j] += __shfl_xor(accb2[j], 4);
        accb2[j] += __shfl_xor(accb2[j], 8);
    }
    accb3 += __shfl_xor(accb3, 16);
    accb3 += __shfl_xor(accb3, 32);
    double live_sum = (double)live_cnt;
    for (int off = 32; off > 0; off >>= 1) {
        loss_sum += __shfl_down(loss_sum, off);
        live_sum += __shfl_down(live_sum, off);
    }
    __shared__ double s_stat[8];
    if (lane == 0) { s_stat[wave] = loss_sum; s_stat[4 + wave] = live_sum; }
    __syncthreads();  // every wave is done with the w3 tables; their LDS now holds the sums
    // Waves 0 and 1 store their dW3 partials lane-major (no bank conflicts) into the two halves
    // of the table space, waves 2 and 3 add theirs; the sum is ((w0 + w2) + (w1 + w3)).
    __shared__ float redb[4][H2 + 16];  // per wave: db2[H2], db3[16]
    float* RW = tab + (wave & 1) * (KB * 256);
    if (wave < 2) {
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) RW[(4 * j + v) * 64 + lane] = acc3[j][v];
    }
    if (m < 4) {
#pragma unroll
        for (int j = 0; j < KB; ++j) redb[wave][16 * j + 4 * g + quad_col(m)] = accb2[j];
    }
    if (g == 0) redb[wave][H2 + m] = accb3;
    __syncthreads();
    if (wave >= 2) {
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) RW[(4 * j + v) * 64 + lane] += acc3[j][v];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int i = tid + 256 * q;
        if (i < len3) {
            float sum;
            if (i < H2 * C) {  // dW3[k][c] sits at register (k/16, k%4) of lane 16 ((k%16)/4) + c
                const int k = i / C, c = i - k * C;
                const int at = (4 * (k >> 4) + (k & 3)) * 64 + 16 * ((k & 15) >> 2) + c;
                sum = tab[at] + tab[KB * 256 + at];
            } else {
                const int c = i - H2 * C;
                sum = (redb[0][H2 + c] + redb[2][H2 + c]) + (redb[1][H2 + c] + redb[3][H2 + c]);
            }
            sw[i] = prev[q] + sum;
        }
    }
    if (tid < H2) sb[tid] = prev[kPer] + ((redb[0][tid] + redb[2][tid]) + (redb[1][tid] + redb[3][tid]));

    if (tid == 0) {
        const double ls = (s_stat[0] + s_stat[1]) + (s_stat[2] + s_stat[3]);
        const double lv = (s_stat[4] + s_stat[5]) + (s_stat[6] + s_stat[7]);
        // per-workgroup partials (thousands of same-address atomics at the end of the kernel
        // would queue up at one L2 channel)
        double* st = a.stats + 2 * (size_t)blockIdx.x;
        st[0] = st_prev[0] + ls;
        st[1] = st_prev[1] + lv;
    }
}

// The same head for 17..64 classes (NCB blocks of 16): logits, softmax, loss, delta3, delta2, db2
// and db3 as in head_mfma_kernel, but dW3 = h2^T . delta3 is left to the wide GEMM kernel (a full
// dW3 would not fit a wave's registers): delta3 is written out, [rows][C].  WAVES waves share the
// w3 operand images in LDS.
template <int KB, int NCB, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void head_wide_kernel(HeadArgs a)
{
    constexpr int H2 = KB * 16, THREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float wtab[];  // 2 NCB KB 64 float4, reused for the sums
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C;
    float b3m[NCB], tgt_m[NCB];
    bool cls[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        cls[cb] = 16 * cb + m < C;
        b3m[cb] = cls[cb] ? a.b3[16 * cb + m] : 0.0f;
        tgt_m[cb] = (a.target && cls[cb]) ? a.target[16 * cb + m] : 0.0f;
    }
    f32x4 ident;
#pragma unroll
    for (int q = 0; q < 4; ++q) ident[q] = (4 * g + q == m) ? 1.0f : 0.0f;
    float accb2[KB], accb3[NCB];  // accb2: db2[16 j + 4 g + quad_col(m)], rows of the lane's quad only
    const bool odd = (m & 1) != 0, hi = (m & 2) != 0;
#pragma unroll
    for (int j = 0; j < KB; ++j) accb2[j] = 0.0f;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) accb3[cb] = 0.0f;
    double loss_sum = 0.0;
    float live_cnt = 0.0f;
    {
        const float4* src = reinterpret_cast<const float4*>(a.w3tab);
        float4* dst = reinterpret_cast<float4*>(wtab);
        for (int e = tid; e < 2 * NCB * KB * 64; e += THREADS) dst[e] = src[e];
    }
    const float4* Wt = reinterpret_cast<const float4*>(wtab);
    const uint64_t tiles = (a.rows + 15) / 16;
    const uint64_t t_step = (uint64_t)gridDim.x * WAVES;
    uint64_t t = (uint64_t)blockIdx.x * WAVES + wave;
    float4 Ha[KB];
    {
        const bool ok0 = t < tiles && t * 16 + m < a.rows;
        const float* h0 = a.h2 + (ok0 ? t * 16 + m : 0) * H2 + 4 * g;
#pragma unroll
        for (int j = 0; j < KB; ++j) Ha[j] = *reinterpret_cast<const float4*>(h0 + 16 * j);
    }
    uint32_t lab4 = 0xffffffffu, live4 = 0x01010101u;
    auto fetch_row_info = [&](uint64_t tile) {
        const uint64_t last = a.rows - 1;
        const uint64_t rb = min(tile * 16 + 4 * g, last);
        if (a.labels) {
            lab4 = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) lab4 |= ((uint32_t)a.labels[min(rb + v, last)] & 0xffu) << (8 * v);
        }
        if (a.live) {
            live4 = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) live4 |= (uint32_t)a.live[min(rb + v, last)] << (8 * v);
        }
    };
    fetch_row_info(t);
    __syncthreads();
    for (; t < tiles; t += t_step) {
        const uint64_t r0 = t * 16, rm = r0 + m;
        const bool okm = rm < a.rows;
        const bool okn = t + t_step < tiles && (t + t_step) * 16 + m < a.rows;
        const float* hnext = a.h2 + (okn ? (t + t_step) * 16 + m : 0) * H2 + 4 * g;
        f32x4 L[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) L[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KB; ++j) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float4 wb = Wt[((2 * cb) * KB + j) * 64 + lane];
                L[cb] = SZ_MFMA16(Ha[j].x, wb.x, L[cb]);
                L[cb] = SZ_MFMA16(Ha[j].y, wb.y, L[cb]);
                L[cb] = SZ_MFMA16(Ha[j].z, wb.z, L[cb]);
                L[cb] = SZ_MFMA16(Ha[j].w, wb.w, L[cb]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // softmax over the NCB x 16 classes of a row; loss and delta3 (lib.rs:1022-1028)
        f32x4 d3[NCB];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint64_t rv = r0 + 4 * g + v;
            const bool okv = rv < a.rows;
            float x[NCB], mx = -INFINITY;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { x[cb] = cls[cb] ? L[cb][v] + b3m[cb] : -INFINITY; mx = fmaxf(mx, x[cb]); }
            mx = row16_max(mx);
            float e[NCB], sum = 0.0f;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { e[cb] = cls[cb] ? expf(x[cb] - mx) : 0.0f; sum += e[cb]; }
            sum = row16_sum(sum);
            const bool alive = okv && ((live4 >> (8 * v)) & 0xffu) != 0;
            const int lab = (int)((lab4 >> (8 * v)) & 0xffu);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float o = e[cb] / sum;
                const float tj = a.labels ? (16 * cb + m == lab ? 1.0f : 0.0f) : tgt_m[cb];
                float d = 0.0f;
                if (alive && cls[cb]) {
                    if (tj != 0.0f) loss_sum -= (double)(tj * logf(fmaxf(o, 1e-12f)));
                    d = o - tj;
                }
                d3[cb][v] = d;
                accb3[cb] += d;
                if (okv && cls[cb]) a.d3[rv * C + 16 * cb + m] = d;
            }
            if (alive && m == 0) live_cnt += 1.0f;
        }
        f32x4 d3t[NCB];  // delta3[row m][class 16cb + 4g + v]
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            d3t[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < 4; ++v) d3t[cb] = SZ_MFMA16(d3[cb][v], ident[v], d3t[cb]);
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const float4 h = Ha[j];
            if (j > 0) Ha[j - 1] = *reinterpret_cast<const float4*>(hnext + 16 * (j - 1));
            f32x4 q = {0.f, 0.f, 0.f, 0.f};   // (delta3 . w3^T)[row m][16j + 4g + v]
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float4 wa = Wt[((2 * cb + 1) * KB + j) * 64 + lane];
                q = SZ_MFMA16(wa.x, d3t[cb][0], q);
                q = SZ_MFMA16(wa.y, d3t[cb][1], q);
                q = SZ_MFMA16(wa.z, d3t[cb][2], q);
                q = SZ_MFMA16(wa.w, d3t[cb][3], q);
            }
            const float4 o = make_float4(q[0] * (1.0f - h.x * h.x), q[1] * (1.0f - h.y * h.y),
                                         q[2] * (1.0f - h.z * h.z), q[3] * (1.0f - h.w * h.w));
            accb2[j] += quad_colsum(o, odd, hi);  // db2 (lib.rs:1038)
            if (okm) *reinterpret_cast<float4*>(a.d2 + rm * H2 + 16 * j + 4 * g) = o;
            __builtin_amdgcn_sched_barrier(0);
        }
        Ha[KB - 1] = *reinterpret_cast<const float4*>(hnext + 16 * (KB - 1));
        fetch_row_info(t + t_step);
    }

    // per-workgroup sums -> slabs (db3 in slab_w3 [blocks][C], db2 in slab_b2 [blocks][H2])
    float* sb3 = a.slab_w3 + (size_t)blockIdx.x * C;
    float* sb2 = a.slab_b2 + (size_t)blockIdx.x * H2;
    const float prev2 = (a.add && tid < H2) ? sb2[tid] : 0.0f;
    const float prev3 = (a.add && tid < C) ? sb3[tid] : 0.0f;
    double st_prev[2] = {0.0, 0.0};
    if (a.add && tid == 0) { st_prev[0] = a.stats[2 * (size_t)blockIdx.x]; st_prev[1] = a.stats[2 * (size_t)blockIdx.x + 1]; }
#pragma unroll
    for (int j = 0; j < KB; ++j) {  // over the four quads of a row of 16 lanes
        accb2[j] += __shfl_xor(accb2[j], 4);
        accb2[j] += __shfl_xor(accb2[j], 8);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        accb3[cb] += __shfl_xor(accb3[cb], 16);
        accb3[cb] += __shfl_xor(accb3[cb], 32);
    }
    double live_sum = (double)live_cnt;
    for (int off = 32; off > 0; off >>= 1) {
        loss_sum += __shfl_down(loss_sum, off);
        live_sum += __shfl_down(live_sum, off);
    }
    __shared__ double s_stat[2 * WAVES];
    if (lane == 0) { s_stat[wave] = loss_sum; s_stat[WAVES + wave] = live_sum; }
    __syncthreads();  // the w3 images are no longer needed; their LDS now holds the per-wave sums
    float* red = wtab;  // [WAVES][H2 + 16 NCB]
    constexpr int RW = H2 + 16 * NCB;
    if (m < 4) {
#pragma unroll
        for (int j = 0; j < KB; ++j) red[wave * RW + 16 * j + 4 * g + quad_col(m)] = accb2[j];
    }
    if (g == 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) red[wave * RW + H2 + 16 * cb + m] = accb3[cb];
    }
    __syncthreads();
    if (tid < H2) {
        float s2 = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s2 += red[w * RW + tid];
        sb2[tid] = prev2 + s2;
    }
    if (tid < C) {
        float s3 = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s3 += red[w * RW + H2 + tid];
        sb3[tid] = prev3 + s3;
    }
    if (tid == 0) {
        double ls = 0.0, lv = 0.0;
        for (int w = 0; w < WAVES; ++w) { ls += s_stat[w]; lv += s_stat[WAVES + w]; }
        double* st = a.stats + 2 * (size_t)blockIdx.x;
        st[0] = st_prev[0] + ls;
        st[1] = st_prev[1] + lv;
    }
}

// out[c][r] = in[r][c]; 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int rows, int cols,
                                                        float* __restrict__ out)
{
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) t[i][tx] = in[(size_t)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * rows + r0 + tx] = t[tx][i];
}

// out[c][0 .. ld) = in[0 .. rows)[c], zeros behind: the first layer's weights as the [N][K] operand of
// a product over the prepared rows' ld columns (whose columns from `rows` on hold 1, 0, 0, ...)
__global__ __launch_bounds__(256) void pad_transpose_kernel(const float* __restrict__ in, int rows, int cols, int ld,
                                                            float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cols * ld) return;
    const int c = i / ld, k = i % ld;
    out[i] = k < rows ? in[(size_t)k * cols + c] : 0.0f;
}

// ---- column sums (bias gradients), split over rows into slabs ------------------------
__global__ void colsum_kernel(const float* __restrict__ D, uint64_t rows, int cols, int ld, int rows_per_split,
                              int add, float* __restrict__ slab, uint64_t slab_stride)
{
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.y * rows_per_split;
    const uint64_t r1 = min(rows, r0 + (uint64_t)rows_per_split);
    float s = 0.0f;
    if (c < cols) {
        float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;  // four independent chains keep four loads in flight
        uint64_t r = r0 + rg;
        for (; r + 12 < r1; r += 16) {
            s += D[r * ld + c]; s1 += D[(r + 4) * ld + c]; s2 += D[(r + 8) * ld + c]; s3 += D[(r + 12) * ld + c];
        }
        for (; r < r1; r += 4) s += D[r * ld + c];
        s = (s + s1) + (s2 + s3);
    }
    part[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float* p = slab + (size_t)blockIdx.y * slab_stride + c;
        *p = (add ? *p : 0.0f) + ((part[0][threadIdx.x] + part[1][threadIdx.x]) +
                                  (part[2][threadIdx.x] + part[3][threadIdx.x]));
    }
}

// grad[i] += sum_z slab[z][i].  128 consecutive elements (32 lanes x 4) x 8 groups of slabs per
// workgroup; the groups' partial sums are combined in a fixed order, so the result is reproducible.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int splits, uint64_t len,
                                                          float* __restrict__ grad)
{
    __shared__ float part[8][128];
    const int c = threadIdx.x & 31, zg = threadIdx.x >> 5;
    const uint64_t i = (uint64_t)blockIdx.x * 128 + 4 * c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if ((len & 3) == 0 && i + 3 < len) {
#pragma unroll 4
        for (int z = zg; z < splits; z += 8) {
            const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)z * len + i);
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        }
    } else {
        for (int z = zg; z < splits; z += 8) {
            const float* p = slab + (size_t)z * len + i;
            if (i < len) s0 += p[0];
            if (i + 1 < len) s1 += p[1];
            if (i + 2 < len) s2 += p[2];
            if (i + 3 < len) s3 += p[3];
        }
    }
    part[zg][4 * c] = s0; part[zg][4 * c + 1] = s1; part[zg][4 * c + 2] = s2; part[zg][4 * c + 3] = s3;
    __syncthreads();
    if (threadIdx.x < 128) {
        const uint64_t o = (uint64_t)blockIdx.x * 128 + threadIdx.x;
        if (o < len) {
            float t = part[0][threadIdx.x];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += part[g][threadIdx.x];
            grad[o] += t;
        }
    }
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

// stats += the per-workgroup {loss, live rows} partials of the head kernels (fixed order), then
// the totals since the last zero_grad go to the two floats behind the gradient vector
__global__ __launch_bounds__(256) void stats_finish_kernel(const double* __restrict__ part, int n_part,
                                                           double* __restrict__ stats, float* __restrict__ tail)
{
    __shared__ double sl[256], sc[256];
    double l = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) { l += part[2 * i]; c += part[2 * i + 1]; }
    sl[threadIdx.x] = l; sc[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sl[threadIdx.x] += sl[threadIdx.x + off]; sc[threadIdx.x] += sc[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double ls = stats[0] + sl[0], lv = stats[1] + sc[0];
        stats[0] = ls; stats[1] = lv;
        tail[0] = (float)ls; tail[1] = (float)lv;
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

// Rows per launch.  Large enough that a launch is many rounds of workgroups per CU, so that the
// prologues and epilogues of some overlap the matrix work of others, and that the fixed cost of the
// ~7 launches per chunk and the partly filled last round of each are small (measured at 60-512-256:
// 49 152 rows, one round, is 15 % slower per row; 393 216 -> 1 572 864 rows gains 3 %, 1 322 800 ->
// 6 614 000 another 2 %).  The workspace holds every activation and delta of a chunk: it may take up to
// 48 GB of the 288 GB, and no more than half of what was free when the process first asked.
static const uint64_t kChunkRowsEnv = [] {
    // STREAMZ_CHUNK_ROWS overrides the rows per launch (a tuning knob; at least 256, multiple of 128)
    const char* e = std::getenv("STREAMZ_CHUNK_ROWS");
    const uint64_t v = e ? std::strtoull(e, nullptr, 10) : 0;
    return v >= 256 ? (v + 127) / 128 * 128 : (uint64_t)0;
}();
static uint64_t workspace_budget()
{
    static const uint64_t budget = [] {
        size_t free_b = 0, total_b = 0;
        uint64_t b = 48ull << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) b = std::min<uint64_t>(b, free_b / 2);
        return std::max<uint64_t>(b, 1ull << 30);
    }();
    return budget;
}
static uint64_t chunk_rows(const Net* n)
{
    if (kChunkRowsEnv) return kChunkRowsEnv;
    const uint64_t per_row = 4ull * (64 + n->in + 2ull * n->h1 + 2ull * n->h2 + n->out);
    const uint64_t r = workspace_budget() / per_row;
    return std::min<uint64_t>(std::max<uint64_t>(r, 131072), 8ull << 20) / 128 * 128;
}
// Most split-K slabs of a weight-gradient product: enough that tiles x splits is several
// workgroups per CU (dW1 has only 4 column tiles, dW2 has 8 tiles)
constexpr uint64_t kMaxSplits = 64, kMaxSplitsW1 = 256;
// dW2 on 256 x 256 tiles has a quarter of the tiles: more K ranges keep every CU busy
inline uint64_t w2_split_cap(const Net* n)
{
    const uint64_t tiles = (uint64_t)((n->h1 + 255) / 256) * ((n->h2 + 255) / 256);
    return tiles <= 4 ? 2 * kMaxSplits : kMaxSplits;
}

int ensure_workspace(Net* n, uint64_t rows)
{
    Workspace& w = n->ws;
    if (rows > chunk_rows(n)) rows = chunk_rows(n);
    if (w.rows >= rows && w.out == n->out) return SZH_OK;
    float** ptrs[] = {&w.xd, &w.h1, &w.h2, &w.p, &w.d2, &w.d1, &w.slab, &w.w2t};
    for (float** p : ptrs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (w.w2s) { (void)hipFree(w.w2s); w.w2s = nullptr; }
    if (w.w3s) { (void)hipFree(w.w3s); w.w3s = nullptr; }
    if (w.w3t) { (void)hipFree(w.w3t); w.w3t = nullptr; }
    if (w.w1t) { (void)hipFree(w.w1t); w.w1t = nullptr; }
    if (w.vote_tab) { (void)hipFree(w.vote_tab); w.vote_tab = nullptr; }
    if (w.live) { (void)hipFree(w.live); w.live = nullptr; }
    if (w.mask) { (void)hipFree(w.mask); w.mask = nullptr; }
    if (w.stat_slab) { (void)hipFree(w.stat_slab); w.stat_slab = nullptr; }
    w.rows = 0;
    const uint64_t R = rows < 256 ? 256 : rows;
    // split-K partials live in four regions so that they can accumulate over the chunks of a call
    const uint64_t head_blocks = 2048;  // upper bound of either head kernel's grid
    const uint64_t len3 = (uint64_t)(n->h2 + 1) * n->out;
    w.sl_w1 = 0;
    w.sl_w2 = w.sl_w1 + kMaxSplitsW1 * (uint64_t)(n->in + 1) * n->h1;
    w.sl_w3 = w.sl_w2 + w2_split_cap(n) * (uint64_t)n->h1 * n->h2;
    w.sl_b2 = w.sl_w3 + std::max<uint64_t>(kMaxSplits, head_blocks) * len3;
    w.sl_b3 = w.sl_b2 + std::max<uint64_t>(kMaxSplits, head_blocks) * n->h2;
    w.slab_floats = w.sl_b3 + head_blocks * 64;
    // prepared rows are padded to a whole tile of columns: [x | 1 | 0...]
    w.ldx = n->in + 1 <= 64 ? 64 : (int)((n->in + 1 + 127) / 128 * 128);
    SZ_HIP(hipMalloc((void**)&w.xd, R * w.ldx * sizeof(float)));
    // h1 and d2 carry 32 rows of slack: the split-bf16 weight-gradient kernel reads whole groups of
    // 32 rows and wants zeros behind the last one
    SZ_HIP(hipMalloc((void**)&w.h1, (R + 32) * n->h1 * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.h2, (R + 32) * n->h2 * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.p, (R + 32) * (n->out ? n->out : 1) * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.d2, (R + 32) * n->h2 * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.d1, R * n->h1 * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.slab, w.slab_floats * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.live, R));
    // mask words up to the next 256-row tile boundary: the wide kernels read them for whole tiles
    SZ_HIP(hipMalloc((void**)&w.mask, ((R + 255) / 256 * 8) * (uint64_t)n->h1 * sizeof(uint32_t)));
    // w2 and its transpose as bf16 planes, [2][3][hidden1 * hidden2]
    SZ_HIP(hipMalloc((void**)&w.w2s, 2 * 3 * ((uint64_t)n->h1 * n->h2 + 1) * sizeof(uint16_t)));
    SZ_HIP(hipMalloc((void**)&w.w3s, 2 * 3 * ((uint64_t)n->h2 * n->out + 2) * sizeof(uint16_t)));
    SZ_HIP(hipMalloc((void**)&w.w3t, ((uint64_t)n->h2 * n->out + 1) * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.w1t, (uint64_t)n->h1 * w.ldx * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.vote_tab, infer::table_floats((int)n->h1, (int)n->h2) * sizeof(float)));
    SZ_HIP(hipMalloc((void**)&w.stat_slab, 2 * head_blocks * sizeof(double)));
    SZ_HIP(hipMalloc((void**)&w.w2t, ((uint64_t)n->h1 * n->h2 + 8 * 64 * 4 * ((n->h2 + 15) / 16)) * sizeof(float)));
    w.rows = R;
    w.out = n->out;
    return SZH_OK;
}

// w2^T (and, for the backward pass, w2) as bf16 planes for the split-bf16 kernels; w.w2t must hold
// the transposed weights when transposed_ready
int prepare_w2_planes(Net* n, bool backward, bool transposed_ready, hipStream_t s)
{
    Workspace& w = n->ws;
    const float* P = n->d_params;
    const uint64_t count = (uint64_t)n->h1 * n->h2;
    ProfScope prof("w2split", s);
    if (!transposed_ready) {
        hipLaunchKernelGGL(transpose_kernel, dim3((n->h2 + 31) / 32, (n->h1 + 31) / 32), dim3(256), 0, s,
                           P + n->off_w2(), (int)n->h1, (int)n->h2, w.w2t);
    }
    const unsigned blocks = (unsigned)((count / 2 + 255) / 256);
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, s, w.w2t, count, w.w2s, count / 2);
    if (backward)
        hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, s, P + n->off_w2(), count, w.w2s + 3 * (count / 2),
                           count / 2);
    if (n->out % 256 == 0 && n->out > 0) {  // the output layer as [classes][hidden2] planes, and as it is for delta2
        const uint64_t c3 = (uint64_t)n->h2 * n->out;
        const unsigned b3 = (unsigned)((c3 / 2 + 255) / 256);
        hipLaunchKernelGGL(transpose_kernel, dim3((n->out + 31) / 32, (n->h2 + 31) / 32), dim3(256), 0, s,
                           P + n->off_w3(), (int)n->h2, (int)n->out, w.w3t);
        hipLaunchKernelGGL(split_planes_kernel, dim3(b3), dim3(256), 0, s, w.w3t, c3, w.w3s, c3 / 2);
        if (backward)
            hipLaunchKernelGGL(split_planes_kernel, dim3(b3), dim3(256), 0, s, P + n->off_w3(), c3, w.w3s + 3 * (c3 / 2), c3 / 2);
    }
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

// hidden layers of `rows` rows (rows <= workspace capacity): h1, h2 in the workspace.  With x3 the
// second layer runs on the split-bf16 kernel and prepare_w2_planes has been called.
int forward_hidden(Net* n, const float* x, uint64_t rows, bool relu2, hipStream_t s, uint32_t* relu_mask = nullptr,
                   int ldx = 0, bool x3 = false, bool wide = false, bool wide1 = false)
{
    Workspace& w = n->ws;
    const float* P = n->d_params;
    float* const h1 = w.h1 + w.row_base * n->h1;
    float* const h2 = w.h2 + w.row_base * n->h2;
    GemmArgs g{};
    g.A = x; g.lda = ldx ? ldx : (int)n->in; g.a_pad = ldx ? 1 : 0;
    g.B = P + n->off_w1(); g.ldb = n->h1; g.C = h1; g.ldc = n->h1;
    g.M = (int)rows; g.N = n->h1; g.K = n->in; g.k_chunk = n->in; g.bias = P + n->off_b1();
    g.mask = relu_mask;
    int rc;
    if (wide1 && relu_mask && ldx) {  // prepared rows: K = ldx, B as [hidden1][ldx]: w.w1t, which the caller has filled
        g.B = w.w1t; g.ldb = ldx; g.K = ldx; g.k_chunk = ldx;
        rc = launch_gemm_f32w<EPI_BIAS_RELU_MASK, T_FWD1>(g, s);
    } else {
        rc = relu_mask ? launch_gemm<A_ROWS, B_KROWS, EPI_BIAS_RELU_MASK, T_FWD1>(g, 1, s)
                       : launch_gemm<A_ROWS, B_KROWS, EPI_BIAS_RELU, T_FWD1>(g, 1, s);
    }
    if (rc) return rc;
    g = GemmArgs{};
    g.A = h1; g.lda = n->h1; g.B = P + n->off_w2(); g.ldb = n->h2; g.C = h2; g.ldc = n->h2;
    g.M = (int)rows; g.N = n->h2; g.K = n->h1; g.k_chunk = n->h1; g.bias = P + n->off_b2();
    if (x3) {  // B as [hidden2][hidden1] planes
        g.B = (const float*)w.w2s; g.ldb = n->h1; g.b_plane = (uint64_t)n->h1 * n->h2;
        return relu2 ? launch_gemm_x3<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_BIAS_RELU, T_FWD2>(g, 1, s)
                     : launch_gemm_x3<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_BIAS_TANH, T_FWD2>(g, 1, s);
    }
    if (wide) {  // B as [hidden2][hidden1]: w.w2t, which the caller has filled
        g.B = w.w2t; g.ldb = n->h1;
        return relu2 ? launch_gemm_f32w<EPI_BIAS_RELU, T_FWD2>(g, s) : launch_gemm_f32w<EPI_BIAS_TANH, T_FWD2>(g, s);
    }
    return relu2 ? launch_gemm<A_ROWS, B_KROWS, EPI_BIAS_RELU, T_FWD2>(g, 1, s)
                 : launch_gemm<A_ROWS, B_KROWS, EPI_BIAS_TANH, T_FWD2>(g, 1, s);
}

int forward_logits(Net* n, uint64_t rows, float* out, hipStream_t s, bool x3 = false, bool wide = false)
{
    Workspace& w = n->ws;
    const float* P = n->d_params;
    GemmArgs g{};
    g.A = w.h2 + w.row_base * n->h2; g.lda = n->h2; g.B = P + n->off_w3(); g.ldb = n->out; g.C = out; g.ldc = n->out;
    g.M = (int)rows; g.N = n->out; g.K = n->h2; g.k_chunk = n->h2; g.bias = P + n->off_b3();
    if (x3) {  // B as [classes][hidden2] planes
        g.B = (const float*)w.w3s; g.ldb = n->h2; g.b_plane = (uint64_t)n->h2 * n->out;
        return launch_gemm_x3<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_BIAS, T_FWD3>(g, 1, s);
    }
    if (wide) {  // B as [classes][hidden2]: w.w3t, which the caller has filled
        g.B = w.w3t; g.ldb = n->h2;
        return launch_gemm_f32w<EPI_BIAS, T_FWD3>(g, s);
    }
    return launch_gemm<A_ROWS, B_KROWS, EPI_BIAS, T_FWD3>(g, 1, s);
}

// How the rows of one net_accumulate call are split into K ranges of the weight-gradient
// products.  Fixed by the first (largest) chunk, so every chunk adds into the same slabs.
struct SplitPlan { int k_chunk; int splits; };
SplitPlan plan_splits(uint64_t rows0, int max_splits)
{
    int splits = (int)((rows0 + 511) / 512);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int k_chunk = (int)((rows0 + splits - 1) / splits);
    k_chunk = (k_chunk + BK - 1) / BK * BK;
    return {k_chunk, (int)((rows0 + k_chunk - 1) / k_chunk)};
}

// slab[z][M][N] (+)= A^T . B over the K range z of `rows` rows, A = [rows][M], B = [rows][N].
// With bias_row the contraction is done for M + 1 rows, the extra one against a column of ones,
// which is the bias gradient; it lands right behind the weight gradient, where the bias sits in
// the parameter vector.
template <int TAG>
int grad_gemm(const float* A, int lda, const float* B, int ldb, int M, int N, uint64_t rows, float* slab,
              const SplitPlan& sp, bool add, bool bias_row, hipStream_t s, bool a_padded = false)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = slab; g.ldc = N;
    g.M = bias_row ? M + 1 : M; g.N = N; g.K = (int)rows; g.k_chunk = sp.k_chunk;
    g.ones_row = bias_row ? M : -1;
    g.slab_add = add ? 1 : 0;
    g.a_pad = a_padded ? 1 : 0;
    return launch_gemm<A_KROWS, B_KROWS, EPI_SLAB, TAG>(g, (int)((rows + sp.k_chunk - 1) / sp.k_chunk), s);
}

// the same contraction on the split-bf16 kernel (M and N multiples of 256, no bias row)
template <int TAG>
int grad_gemm_x3(const float* A, int lda, const float* B, int ldb, int M, int N, uint64_t rows, float* slab,
                 const SplitPlan& sp, bool add, hipStream_t s, uint64_t slab_stride = 0)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = slab; g.ldc = N;
    g.M = M; g.N = N; g.K = (int)rows; g.k_chunk = sp.k_chunk;
    g.slab_add = add ? 1 : 0;
    g.slab_stride = slab_stride;
    return launch_gemm_x3<X3_KROWS_F32, X3_KROWS_F32, EPI_SLAB, TAG>(g, (int)((rows + sp.k_chunk - 1) / sp.k_chunk), s);
}

// ... and on the f32 walking kernel (same conditions)
template <int TAG>
int grad_gemm_f32w(const float* A, int lda, const float* B, int ldb, int M, int N, uint64_t rows, float* slab,
                   const SplitPlan& sp, bool add, hipStream_t s, uint64_t slab_stride = 0)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = slab; g.ldc = N;
    g.M = M; g.N = N; g.K = (int)rows; g.k_chunk = sp.k_chunk;
    g.slab_add = add ? 1 : 0;
    g.slab_stride = slab_stride;
    return launch_gemm_f32w<EPI_SLAB, TAG, X3_KROWS_F32>(g, s, (int)((rows + sp.k_chunk - 1) / sp.k_chunk));
}

int reduce_slabs(const float* slab, int splits, uint64_t len, float* grad, hipStream_t s)
{
    ProfScope prof("reduce", s);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((len + 127) / 128)), dim3(256), 0, s, slab, splits, len,
                       grad);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

// slab_stride: floats between the sums of two K ranges (0: cols)
int colsum_slabs(const float* D, int cols, uint64_t rows, float* slab, const SplitPlan& sp, bool add, hipStream_t s,
                 uint64_t slab_stride = 0)
{
    ProfScope prof("bias", s);
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, (unsigned)((rows + sp.k_chunk - 1) / sp.k_chunk)),
                       dim3(256), 0, s, D, rows, cols, cols, sp.k_chunk, add ? 1 : 0, slab, slab_stride ? slab_stride : (uint64_t)cols);
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
    if (const char* e = std::getenv("STREAMZ_GEMM")) n->gemm_mode = std::string(e) == "bf16x3" ? SZH_GEMM_BF16X3 : SZH_GEMM_F32;
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
    float* ptrs[] = {w.xd, w.h1, w.h2, w.p, w.d2, w.d1, w.slab, w.w2t, n->d_params, n->d_grad};
    for (float* p : ptrs) if (p) (void)hipFree(p);
    if (w.w2s) (void)hipFree(w.w2s);
    if (w.w3s) (void)hipFree(w.w3s);
    if (w.w3t) (void)hipFree(w.w3t);
    if (w.w1t) (void)hipFree(w.w1t);
    if (w.vote_tab) (void)hipFree(w.vote_tab);
    if (w.live) (void)hipFree(w.live);
    if (w.mask) (void)hipFree(w.mask);
    if (w.stat_slab) (void)hipFree(w.stat_slab);
    if (n->d_stats) (void)hipFree(n->d_stats);
    for (void* p : n->stage.dev) if (p) (void)hipFree(p);
    for (void* p : n->keep.buf) if (p) (void)hipFree(p);
    if (n->keep.compute) (void)hipStreamDestroy(n->keep.compute);
    if (n->stage.pinned) (void)hipHostFree(n->stage.pinned);
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
    const bool x3 = use_x3(n, rows);
    const bool wide = !x3 && use_wide_f32(n, rows);
    if (x3 && (rc = prepare_w2_planes(n, false, false, s))) return rc;
    if (wide) {
        hipLaunchKernelGGL(transpose_kernel, dim3((n->h2 + 31) / 32, (n->h1 + 31) / 32), dim3(256), 0, s,
                           n->d_params + n->off_w2(), (int)n->h1, (int)n->h2, n->ws.w2t);
        SZ_HIP(hipGetLastError());
    }
    for (uint64_t r0 = 0; r0 < rows; r0 += chunk_rows(n)) {
        const uint64_t R = std::min<uint64_t>(chunk_rows(n), rows - r0);
        rc = forward_hidden(n, x_dev + r0 * n->in, R, kind == kFwdEmbedRelu, s, nullptr, 0, x3, wide);
        if (rc) return rc;
        float* o = out_dev + r0 * ow;
        if (kind == kFwdProbs) {
            if (n->out == 0) continue;
            rc = forward_logits(n, R, o, s, x3 && use_x3_out(n, rows));
            if (rc) return rc;
            rc = launch_softmax(o, R, (int)n->out, 0, nullptr, nullptr, nullptr, nullptr, s);
            if (rc) return rc;
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

template <int KB, int NCB, int WAVES>
int launch_head_wide_one(const HeadArgs& h, int blocks, hipStream_t s)
{
    const size_t lds = (size_t)2 * NCB * KB * 64 * sizeof(float4);
    if (int rc = allow_dynamic_lds((const void*)head_wide_kernel<KB, NCB, WAVES>, lds)) return rc;
    hipLaunchKernelGGL((head_wide_kernel<KB, NCB, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, s, h);
    return SZH_OK;
}

template <int KB>
int launch_head_wide_kb(const HeadArgs& h, int ncb, int blocks, hipStream_t s)
{
    switch (ncb) {
    case 2: return launch_head_wide_one<KB, 2, 4>(h, blocks, s);
    case 3: return launch_head_wide_one<KB, 3, 8>(h, blocks, s);
    default: return launch_head_wide_one<KB, 4, 8>(h, blocks, s);
    }
}

int launch_head_wide(const HeadArgs& h, int kb, int ncb, int blocks, hipStream_t s)
{
    switch (kb) {
    case 16: return launch_head_wide_kb<16>(h, ncb, blocks, s);
    case 8: return launch_head_wide_kb<8>(h, ncb, blocks, s);
    default: return launch_head_wide_kb<4>(h, ncb, blocks, s);
    }
}

int net_accumulate(Net* n, const AccumArgs& a, hipStream_t s)
{
    if (a.rows == 0 || n->out == 0) return SZH_OK;
    int rc = ensure_workspace(n, a.rows);
    if (rc) return rc;
    Workspace& w = n->ws;
    const float* P = n->d_params;
    float* G = n->d_grad;
    {   // w2 transposed once per call (the weights are frozen while gradients accumulate), so that
        // delta1 = delta2 . w2^T reads its second operand k-major like the forward product does
        ProfScope prof("w2t", s);
        hipLaunchKernelGGL(transpose_kernel, dim3((n->h2 + 31) / 32, (n->h1 + 31) / 32), dim3(256), 0, s,
                           P + n->off_w2(), (int)n->h1, (int)n->h2, w.w2t);
    }
    SZ_HIP(hipGetLastError());
    const bool x3 = use_x3(n, a.rows);
    const bool wide1 = use_wide_f32(n, a.rows);  // the first layer's product (exact f32 in either mode)
    const bool wide = !x3 && wide1;
    if (x3 && (rc = prepare_w2_planes(n, true, true, s))) return rc;
    if (wide1) {
        ProfScope prof("w2t", s);
        hipLaunchKernelGGL(pad_transpose_kernel, dim3((n->h1 * w.ldx + 255) / 256), dim3(256), 0, s, P + n->off_w1(),
                           (int)n->in, (int)n->h1, w.ldx, w.w1t);
        SZ_HIP(hipGetLastError());
    }
    float* w3tab = w.w2t + (uint64_t)n->h1 * n->h2;  // behind the transposed w2 (a multiple of 4 floats in)
    // Every chunk adds its split-K partials into the same slabs (the first one overwrites them);
    // the slabs are folded into the gradient vector once, after the last chunk.
    // Chunks of equal size (a short last chunk would run its split-K products on a fraction of the
    // K ranges the first chunk fixed).
    const uint64_t n_chunks = (a.rows + chunk_rows(n) - 1) / chunk_rows(n);
    const uint64_t chunk = ((a.rows + n_chunks - 1) / n_chunks + 127) / 128 * 128;
    const uint64_t rows0 = std::min<uint64_t>(chunk, a.rows);
    const SplitPlan sp = plan_splits(rows0, (int)kMaxSplits);
    const SplitPlan sp1 = plan_splits(rows0, (int)kMaxSplitsW1);
    const SplitPlan sp2 = (x3 || wide) ? plan_splits(rows0, (int)w2_split_cap(n)) : sp;  // dW2
    // the output layer: matrix-core head with dW3 in registers (<= 16 classes), matrix-core head
    // plus a GEMM for dW3 (<= 64 classes), vector head (<= 32 classes, other hidden2), or GEMMs
    const bool h2_tiles = n->h2 == 64 || n->h2 == 128 || n->h2 == 256;
    const bool mfma_head = h2_tiles && n->out <= 16;
    const bool wide_head = h2_tiles && n->out > 16 && n->out <= 64;
    const bool valu_head = !mfma_head && !wide_head && n->out <= 32 && n->h2 <= 256 && (n->h2 % 4) == 0;
    const bool use_head = mfma_head || wide_head || valu_head;
    const int ncb = ((int)n->out + 15) / 16, wide_waves = ncb <= 2 ? 4 : 8;
    const uint64_t len3 = (uint64_t)n->h2 * n->out + n->out;
    int head_blocks = 0, head_tpb = 1;
    if (mfma_head || wide_head) {
        const int kb = (int)n->h2 / 16;
        hipLaunchKernelGGL(head_tables_kernel, dim3((ncb * kb * 64 + 255) / 256), dim3(256), 0, s, P + n->off_w3(), kb,
                           (int)n->out, ncb, w3tab);
        SZ_HIP(hipGetLastError());
        const uint64_t wave_tiles = (rows0 + 15) / 16;
        const int waves = mfma_head ? 4 : wide_waves;
        head_blocks = (int)std::min<uint64_t>(mfma_head || wide_waves == 4 ? kMHeadBlocks : kMHeadBlocks / 2,
                                              (wave_tiles + waves - 1) / waves);
    } else if (valu_head) {
        const int tiles = (int)((rows0 + kHeadRows - 1) / kHeadRows);
        head_blocks = tiles < 2048 ? tiles : 2048;
        head_tpb = (tiles + head_blocks - 1) / head_blocks;
        head_blocks = (tiles + head_tpb - 1) / head_tpb;
    }
    for (uint64_t r0 = 0; r0 < a.rows; r0 += chunk) {
        const uint64_t R = std::min<uint64_t>(chunk, a.rows - r0);
        const bool add = r0 != 0;
        // gather / dropout / live flags
        {
            ProfScope prof("prep", s);
            const float* xs = a.perm_dev ? a.x_dev : a.x_dev + r0 * n->in;
            const uint8_t* ks = a.keep_dev ? a.keep_dev + r0 * n->in : nullptr;
            if (n->in % 4 == 0 && w.ldx == 64 && ((uintptr_t)xs & 15) == 0 && ((uintptr_t)ks & 3) == 0)
                hipLaunchKernelGGL(prep_rows4_kernel, dim3((unsigned)((R * 16 + 255) / 256)), dim3(256), 0, s, xs,
                                   a.perm_dev ? a.perm_dev + r0 : nullptr, ks, a.dropout, a.seed, a.row_counter0 + r0, R,
                                   (int)n->in, a.skip_zero_rows ? 1 : 0, w.xd, w.live);
            else
                hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)((R + kPrepRows - 1) / kPrepRows)), dim3(256), 0, s, xs,
                                   a.perm_dev ? a.perm_dev + r0 : nullptr, ks, a.dropout, a.seed, a.row_counter0 + r0, R,
                                   (int)n->in, a.skip_zero_rows ? 1 : 0, w.xd, w.ldx, w.live);
        }
        SZ_HIP(hipGetLastError());
        // the ReLU derivative travels as one bit per unit
        uint32_t* relu_mask = w.mask;
        rc = forward_hidden(n, w.xd, R, false, s, relu_mask, w.ldx, x3, wide, wide1);
        if (rc) return rc;
        if (use_head) {
            HeadArgs h{};
            h.h2 = w.h2; h.w3 = P + n->off_w3(); h.b3 = P + n->off_b3();
            h.target = a.target_dev; h.labels = a.labels_dev ? a.labels_dev + r0 : nullptr; h.live = w.live;
            h.w3tab = w3tab; h.d2 = w.d2; h.stats = w.stat_slab; h.rows = R; h.H2 = (int)n->h2; h.C = (int)n->out;
            h.Cp = n->out <= 4 ? 4 : (n->out <= 8 ? 8 : (n->out <= 16 ? 16 : 32));
            h.tiles_per_block = head_tpb;
            h.add = add ? 1 : 0;
            h.slab_w3 = w.slab + w.sl_w3;
            h.slab_b2 = w.slab + w.sl_b2;
            h.d3 = w.p;
            if (wide_head) h.slab_w3 = w.slab + w.sl_b3;  // only db3 comes from this kernel
            {
            ProfScope prof("head", s);
            if (mfma_head) {
                // the grid is the same for every chunk, so that each workgroup keeps its slab
                if (n->h2 == 256) hipLaunchKernelGGL(head_mfma_kernel<16>, dim3(head_blocks), dim3(256), 0, s, h);
                else if (n->h2 == 128) hipLaunchKernelGGL(head_mfma_kernel<8>, dim3(head_blocks), dim3(256), 0, s, h);
                else hipLaunchKernelGGL(head_mfma_kernel<4>, dim3(head_blocks), dim3(256), 0, s, h);
            } else if (wide_head) {
                rc = launch_head_wide(h, (int)n->h2 / 16, ncb, head_blocks, s);
                if (rc) return rc;
            } else {
                const int blocks = (int)(((R + kHeadRows - 1) / kHeadRows + head_tpb - 1) / head_tpb);
                const int ldh = (int)n->h2 + 4;
                const size_t lds = sizeof(float) * ((size_t)kHeadRows * ldh + (size_t)n->h2 * h.Cp +
                                                    (size_t)h.Cp * ldh + (size_t)kHeadRows * h.Cp);
                const void* fn = h.Cp == 4 ? (const void*)head_kernel<4> : h.Cp == 8 ? (const void*)head_kernel<8>
                               : h.Cp == 16 ? (const void*)head_kernel<16> : (const void*)head_kernel<32>;
                rc = allow_dynamic_lds(fn, 112 * 1024);
                if (rc) return rc;
                if (h.Cp == 4) hipLaunchKernelGGL(head_kernel<4>, dim3(blocks), dim3(kHeadThreads), lds, s, h);
                else if (h.Cp == 8) hipLaunchKernelGGL(head_kernel<8>, dim3(blocks), dim3(kHeadThreads), lds, s, h);
                else if (h.Cp == 16) hipLaunchKernelGGL(head_kernel<16>, dim3(blocks), dim3(kHeadThreads), lds, s, h);
                else hipLaunchKernelGGL(head_kernel<32>, dim3(blocks), dim3(kHeadThreads), lds, s, h);
            }
            }
            SZ_HIP(hipGetLastError());
            if (wide_head) {  // dW3 = h2^T . delta3 on the wide GEMM kernel
                rc = grad_gemm<T_DW3>(w.h2, n->h2, w.p, n->out, n->h2, n->out, R, w.slab + w.sl_w3, sp1, add, false, s);
                if (rc) return rc;
            }
        } else {
            const bool x3o = use_x3_out(n, a.rows);
            const bool wide_o = wide && n->out % 256 == 0;  // the output layer's three products on the walking kernel too
            if (wide_o && r0 == 0) {
                hipLaunchKernelGGL(transpose_kernel, dim3((n->out + 31) / 32, (n->h2 + 31) / 32), dim3(256), 0, s,
                                   P + n->off_w3(), (int)n->h2, (int)n->out, w.w3t);
                SZ_HIP(hipGetLastError());
            }
            rc = forward_logits(n, R, w.p, s, x3o, wide_o);
            if (rc) return rc;
            {
                ProfScope prof("softmax", s);
                rc = launch_softmax(w.p, R, (int)n->out, 1, a.target_dev, a.labels_dev ? a.labels_dev + r0 : nullptr,
                                    (const uint8_t*)w.live, n->d_stats, s);
            }
            if (rc) return rc;
            // delta2 = (delta3 . w3^T) * (1 - h2^2)
            GemmArgs g{};
            g.A = w.p; g.lda = n->out; g.B = P + n->off_w3(); g.ldb = n->out; g.C = w.d2; g.ldc = n->h2;
            g.M = (int)R; g.N = n->h2; g.K = n->out; g.k_chunk = n->out; g.aux = w.h2; g.ldaux = n->h2;
            if (x3o) {  // B as [hidden2][classes] planes: w3 itself
                g.B = (const float*)(w.w3s + 3 * ((uint64_t)n->h2 * n->out / 2)); g.ldb = n->out;
                g.b_plane = (uint64_t)n->h2 * n->out;
                rc = launch_gemm_x3<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_MUL_DTANH, T_DELTA2>(g, 1, s);
                if (rc) return rc;
                if (R % 32) {  // zero rows up to the next multiple of 32 for the k-major reads of dW3
                    const uint64_t pad = 32 - R % 32;
                    SZ_HIP(hipMemsetAsync(w.h2 + R * n->h2, 0, pad * n->h2 * sizeof(float), s));
                    SZ_HIP(hipMemsetAsync(w.p + R * n->out, 0, pad * n->out * sizeof(float), s));
                }
                // [dW3 | db3] per K range: the matrix part from the GEMM, the bias row from column sums
                rc = grad_gemm_x3<T_DW3>(w.h2, n->h2, w.p, n->out, n->h2, n->out, R, w.slab + w.sl_w3, sp, add, s, len3);
                if (rc) return rc;
                rc = colsum_slabs(w.p, n->out, R, w.slab + w.sl_w3 + (uint64_t)n->h2 * n->out, sp, add, s, len3);
                if (rc) return rc;
            } else if (wide_o) {  // B = w3 itself, [hidden2][classes]
                rc = launch_gemm_f32w<EPI_MUL_DTANH, T_DELTA2>(g, s);
                if (rc) return rc;
                if (R % 32) {  // zero rows up to the next multiple of 32 for the k-major reads of dW3
                    const uint64_t pad = 32 - R % 32;
                    SZ_HIP(hipMemsetAsync(w.h2 + R * n->h2, 0, pad * n->h2 * sizeof(float), s));
                    SZ_HIP(hipMemsetAsync(w.p + R * n->out, 0, pad * n->out * sizeof(float), s));
                }
                rc = grad_gemm_f32w<T_DW3>(w.h2, n->h2, w.p, n->out, n->h2, n->out, R, w.slab + w.sl_w3, sp, add, s, len3);
                if (rc) return rc;
                rc = colsum_slabs(w.p, n->out, R, w.slab + w.sl_w3 + (uint64_t)n->h2 * n->out, sp, add, s, len3);
                if (rc) return rc;
            } else {
                rc = launch_gemm<A_ROWS, B_ROWS, EPI_MUL_DTANH, T_DELTA2>(g, 1, s);
                if (rc) return rc;
                rc = grad_gemm<T_DW3>(w.h2, n->h2, w.p, n->out, n->h2, n->out, R, w.slab + w.sl_w3, sp, add, true, s);
                if (rc) return rc;
            }
            rc = colsum_slabs(w.d2, n->h2, R, w.slab + w.sl_b2, sp, add, s);
            if (rc) return rc;
        }
        // delta1 = (delta2 . w2^T) * [h1 > 0]
        GemmArgs g{};
        g.A = w.d2; g.lda = n->h2; g.B = w.w2t; g.ldb = n->h1; g.C = w.d1; g.ldc = n->h1;
        g.M = (int)R; g.N = n->h1; g.K = n->h2; g.k_chunk = n->h2; g.aux = w.h1; g.ldaux = n->h1;
        g.mask = relu_mask;
        if (x3) {  // B as [hidden1][hidden2] planes: w2 itself
            g.B = (const float*)(w.w2s + 3 * ((uint64_t)n->h1 * n->h2 / 2)); g.ldb = n->h2;
            g.b_plane = (uint64_t)n->h1 * n->h2;
            rc = launch_gemm_x3<X3_ROWS_F32, X3_ROWS_SPLIT, EPI_MUL_MASK, T_DELTA1>(g, 1, s);
        } else if (wide) {  // B as [hidden1][hidden2]: w2 itself
            g.B = P + n->off_w2(); g.ldb = n->h2;
            rc = launch_gemm_f32w<EPI_MUL_MASK, T_DELTA1>(g, s);
        } else {
            rc = relu_mask ? launch_gemm<A_ROWS, B_KROWS, EPI_MUL_MASK, T_DELTA1>(g, 1, s)
                           : launch_gemm<A_ROWS, B_KROWS, EPI_MUL_DRELU, T_DELTA1>(g, 1, s);
        }
        if (rc) return rc;
        // weight gradients: contractions over the rows ([w1|b1] in one launch via the ones row)
        if ((x3 || wide) && R % 32) {  // zero rows up to the next multiple of 32
            const uint64_t pad = 32 - R % 32;
            SZ_HIP(hipMemsetAsync(w.h1 + R * n->h1, 0, pad * n->h1 * sizeof(float), s));
            SZ_HIP(hipMemsetAsync(w.d2 + R * n->h2, 0, pad * n->h2 * sizeof(float), s));
        }
        rc = x3 ? grad_gemm_x3<T_DW2>(w.h1, n->h1, w.d2, n->h2, n->h1, n->h2, R, w.slab + w.sl_w2, sp2, add, s)
           : wide ? grad_gemm_f32w<T_DW2>(w.h1, n->h1, w.d2, n->h2, n->h1, n->h2, R, w.slab + w.sl_w2, sp2, add, s)
                  : grad_gemm<T_DW2>(w.h1, n->h1, w.d2, n->h2, n->h1, n->h2, R, w.slab + w.sl_w2, sp2, add, false, s);
        if (rc) return rc;
        rc = grad_gemm<T_DW1>(w.xd, w.ldx, w.d1, n->h1, n->in + 1, n->h1, R, w.slab + w.sl_w1, sp1, add, false, s, true);
        if (rc) return rc;
    }
    if (wide_head) {
        rc = reduce_slabs(w.slab + w.sl_w3, sp1.splits, (uint64_t)n->h2 * n->out, G + n->off_w3(), s);
        if (rc) return rc;
        rc = reduce_slabs(w.slab + w.sl_b3, head_blocks, n->out, G + n->off_b3(), s);
        if (rc) return rc;
        rc = reduce_slabs(w.slab + w.sl_b2, head_blocks, n->h2, G + n->off_b2(), s);
        if (rc) return rc;
    } else {
        const int tail_slabs = use_head ? head_blocks : sp.splits;
        rc = reduce_slabs(w.slab + w.sl_w3, tail_slabs, len3, G + n->off_w3(), s);
        if (rc) return rc;
        rc = reduce_slabs(w.slab + w.sl_b2, tail_slabs, n->h2, G + n->off_b2(), s);
        if (rc) return rc;
    }
    rc = reduce_slabs(w.slab + w.sl_w2, sp2.splits, (uint64_t)n->h1 * n->h2, G + n->off_w2(), s);
    if (rc) return rc;
    rc = reduce_slabs(w.slab + w.sl_w1, sp1.splits, (uint64_t)(n->in + 1) * n->h1, G + n->off_w1(), s);
    if (rc) return rc;
    hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)w.stat_slab, use_head ? head_blocks : 0,
                       n->d_stats, G + n->param_count());
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

// ---- pooled embeddings --------------------------------------------------------------------
// Per-column order statistic of E[rows][D] by radix selection, 8 bits per pass: a workgroup owns
// 32 columns and keeps a 256-bin histogram per column in LDS; after each pass one thread per
// column picks the bin that holds rank k and narrows the key prefix.  Floats are compared through
// the usual order-preserving map to unsigned keys.
constexpr int kMedCols = 32;
constexpr int kMedLd = 257;  // histogram row stride: columns that hit the same bin use different banks

__device__ __forceinline__ uint32_t float_key(float v)
{
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// out[which][col] = the element of rank k[which] (0-based, ascending) of column col
__global__ __launch_bounds__(256) void column_select_kernel(const float* __restrict__ E, uint64_t rows, int D, uint64_t k0,
                                                           uint64_t k1, float* __restrict__ out)
{
    __shared__ unsigned int hist[kMedCols * kMedLd];
    __shared__ uint32_t prefix[kMedCols];
    __shared__ unsigned long long rank[kMedCols];
    const int c = threadIdx.x & (kMedCols - 1), phase = threadIdx.x / kMedCols;
    const int col = blockIdx.x * kMedCols + c;
    for (int which = 0; which < 2; ++which) {
        if (threadIdx.x < kMedCols) { prefix[threadIdx.x] = 0; rank[threadIdx.x] = which ? k1 : k0; }
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int e = threadIdx.x; e < kMedCols * kMedLd; e += 256) hist[e] = 0;
            __syncthreads();
            if (col < D) {
                const uint32_t want = prefix[c];
                for (uint64_t r = phase; r < rows; r += 256 / kMedCols) {
                    const uint32_t key = float_key(E[r * D + col]);
                    // keys that agree with the prefix chosen so far in the bits above this byte
                    if (shift == 24 || (key >> (shift + 8)) == (want >> (shift + 8)))
                        atomicAdd(&hist[c * kMedLd + ((key >> shift) & 255u)], 1u);
                }
            }
            __syncthreads();
            if (threadIdx.x < kMedCols) {
                unsigned long long k = rank[threadIdx.x];
                int bin = 0;
                for (; bin < 255; ++bin) {
                    const unsigned int h = hist[threadIdx.x * kMedLd + bin];
                    if (k < h) break;
                    k -= h;
                }
                rank[threadIdx.x] = k;
                prefix[threadIdx.x] |= (uint32_t)bin << shift;
            }
            __syncthreads();
        }
        if (threadIdx.x < kMedCols && blockIdx.x * kMedCols + threadIdx.x < D)
            out[(size_t)which * D + blockIdx.x * kMedCols + threadIdx.x] = key_float(prefix[threadIdx.x]);
        __syncthreads();
    }
}

// Pooled embedding of `rows` feature rows, un-normalised, to the host: mode 0 = mean of
// forward_embedding (lib.rs:1451-1466), 1 = per-dimension median of forward_embedding
// (lib.rs:1482-1492), 2 = per-dimension median of embed (lib.rs:1433-1443).
int net_pool_host(Net* n, const float* x_dev, uint64_t rows, int mode, float* out)
{
    const uint32_t D = n->h2;
    for (uint32_t i = 0; i < D; ++i) out[i] = 0.0f;
    if (rows == 0 || D == 0) return SZH_OK;
    const ForwardKind kind = mode == 2 ? kFwdEmbedTanh : kFwdEmbedRelu;
    float* E = nullptr;
    double* d_aux = nullptr;
    SZ_HIP(hipMalloc((void**)&E, rows * D * sizeof(float)));
    hipError_t he = hipMalloc((void**)&d_aux, 2 * (size_t)D * sizeof(double));
    int rc = he == hipSuccess ? SZH_OK : hip_fail(he, "pooling buffers");
    if (rc == SZH_OK) rc = net_forward_dev(n, x_dev, rows, E, kind, nullptr);
    if (rc == SZH_OK && mode == 0) {
        std::vector<double> sums(D);
        he = hipMemsetAsync(d_aux, 0, D * sizeof(double), nullptr);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(prob_sum_kernel, dim3((unsigned)std::min<uint64_t>((rows + 63) / 64, 1024)), dim3(256), 0, nullptr,
                               E, rows, (int)D, d_aux);
            he = hipMemcpy(sums.data(), d_aux, D * sizeof(double), hipMemcpyDeviceToHost);
        }
        if (he != hipSuccess) rc = hip_fail(he, "mean pooling");
        else for (uint32_t i = 0; i < D; ++i) out[i] = (float)sums[i] / (float)rows;
    } else if (rc == SZH_OK) {
        std::vector<float> sel(2 * (size_t)D);
        float* d_sel = reinterpret_cast<float*>(d_aux);
        const uint64_t mid = rows / 2;
        hipLaunchKernelGGL(column_select_kernel, dim3((D + kMedCols - 1) / kMedCols), dim3(256), 0, nullptr, E, rows, (int)D,
                           rows % 2 == 0 ? mid - 1 : mid, mid, d_sel);
        he = hipMemcpy(sel.data(), d_sel, 2 * (size_t)D * sizeof(float), hipMemcpyDeviceToHost);
        if (he != hipSuccess) rc = hip_fail(he, "median pooling");
        else for (uint32_t i = 0; i < D; ++i) out[i] = rows % 2 == 0 ? (sel[i] + sel[D + i]) / 2.0f : sel[D + i];
    }
    (void)hipFree(E);
    if (d_aux) (void)hipFree(d_aux);
    return rc;
}

int net_reserve_rows(Net* n, uint64_t rows) { return ensure_workspace(n, rows); }
uint64_t net_workspace_rows(const Net* n) { return n->ws.rows; }

int net_stage_dev(Net* n, int slot, size_t bytes, void** out)
{
    Staging& st = n->stage;
    if (st.dev_cap[slot] < bytes) {
        if (st.dev[slot]) { (void)hipFree(st.dev[slot]); st.dev[slot] = nullptr; st.dev_cap[slot] = 0; }
        const size_t cap = (std::max<size_t>(bytes, 4096) + 4095) / 4096 * 4096;
        SZ_HIP(hipMalloc(&st.dev[slot], cap));
        st.dev_cap[slot] = cap;
    }
    *out = st.dev[slot];
    return SZH_OK;
}

int net_keep_stream(Net* n, hipStream_t* compute)
{
    if (!n->keep.compute) SZ_HIP(hipStreamCreateWithFlags(&n->keep.compute, hipStreamNonBlocking));
    *compute = n->keep.compute;
    return SZH_OK;
}

int net_keep_buffer(Net* n, int which, size_t bytes, void** out)
{
    Net::IdentifyKeep& k = n->keep;
    if (k.cap[which] < bytes) {
        if (k.buf[which]) { (void)hipFree(k.buf[which]); k.buf[which] = nullptr; k.cap[which] = 0; }
        const size_t cap = (std::max<size_t>(bytes, 65536) + 65535) / 65536 * 65536;
        SZ_HIP(hipMalloc(&k.buf[which], cap));
        k.cap[which] = cap;
    }
    *out = k.buf[which];
    return SZH_OK;
}

int net_stage_pinned(Net* n, size_t bytes, void** out)
{
    Staging& st = n->stage;
    if (st.pinned_cap < bytes) {
        if (st.pinned) { (void)hipHostFree(st.pinned); st.pinned = nullptr; st.pinned_cap = 0; }
        const size_t cap = (std::max<size_t>(bytes, 65536) + 4095) / 4096 * 4096;
        SZ_HIP(hipHostMalloc(&st.pinned, cap, hipHostMallocDefault));
        st.pinned_cap = cap;
    }
    *out = st.pinned;
    return SZH_OK;
}

// ---- the vote of a batch of feature rows as one kernel (infer_kernel.h) -------------------------
static bool infer_fused_applies(const Net* n)
{
    static const bool off = std::getenv("STREAMZ_NO_FUSED_VOTE") != nullptr;  // (the per-layer path, for comparisons)
    return !off && n->in <= 60 && n->in % 4 == 0 && n->h1 % 128 == 0 && n->h1 <= 1024 && n->h2 % 32 == 0 && n->h2 <= 256 && n->out >= 1 &&
           n->out <= (uint32_t)infer::kMaxClasses;
}

// the weights in that kernel's fetch order (the workspace must exist)
bool net_vote_fused(const Net* n) { return infer_fused_applies(n); }

int net_vote_prepare(Net* n, hipStream_t s)
{
    if (!infer_fused_applies(n)) return SZH_OK;
    if (int rc = ensure_workspace(n, 256)) return rc;
    const float* P = n->d_params;
    ProfScope prof("w2t", s);
    const size_t total = infer::table_floats((int)n->h1, (int)n->h2);
    hipLaunchKernelGGL(infer::infer_tables_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P + n->off_w1(),
                       P + n->off_b1(), P + n->off_w2(), P + n->off_w3(), (int)n->in, (int)n->h1, (int)n->h2, (int)n->out, n->ws.vote_tab);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

template <int MT, int NCT>
static int launch_infer_vote_one(const infer::Args& a0, size_t lds, hipStream_t s)
{
    infer::Args a = a0;
    if (int rc = allow_dynamic_lds((const void*)infer::infer_vote_kernel<MT, NCT>, lds)) return rc;
    const uint64_t tiles = (a.rows + 32 * MT - 1) / (32 * MT);
    const unsigned blocks = (unsigned)std::min<uint64_t>(tiles, (uint64_t)compute_units());
    ProfScope prof("vote_fused", s);
    hipLaunchKernelGGL((infer::infer_vote_kernel<MT, NCT>), dim3(blocks), dim3(infer::kThreads), lds, s, a);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

static int launch_infer_vote(Net* n, const float* x_dev, uint64_t rows, float threshold, unsigned long long* counts_dev,
                             double* psums_dev, hipStream_t s)
{
    const float* P = n->d_params;
    infer::Args a{};
    a.x = x_dev; a.rows = rows; a.in = (int)n->in;
    a.tab = n->ws.vote_tab; a.b2 = P + n->off_b2(); a.b3 = P + n->off_b3();
    a.H1 = (int)n->h1; a.H2 = (int)n->h2; a.C = (int)n->out;
    a.threshold = threshold; a.counts = counts_dev; a.psums = psums_dev;
    const int nct = n->out <= 32 ? 1 : 2;
    // two blocks of 32 rows per workgroup (every weight fragment used twice) once that still leaves
    // every compute unit two tiles
    const int mt = (rows >= 128ull * compute_units() && infer::lds_bytes(2, a.H1, nct, a.in) <= 156 * 1024) ? 2 : 1;
    a.tail_off = (int)infer::act_floats(mt, a.H1, nct);
    const size_t lds = infer::lds_bytes(mt, a.H1, nct, a.in);
    if (mt == 2) return nct == 1 ? launch_infer_vote_one<2, 1>(a, lds, s) : launch_infer_vote_one<2, 2>(a, lds, s);
    return nct == 1 ? launch_infer_vote_one<1, 1>(a, lds, s) : launch_infer_vote_one<1, 2>(a, lds, s);
}

int net_vote_dev(Net* n, const float* x_dev, uint64_t rows, float threshold, unsigned long long* counts_dev,
                 double* psums_dev, hipStream_t s, bool tables_prepared)
{
    if (rows == 0 || n->out == 0) return SZH_OK;
    if (infer_fused_applies(n) && ((uintptr_t)x_dev & 15) == 0) {
        if (!tables_prepared)
            if (int rc = net_vote_prepare(n, s)) return rc;
        return launch_infer_vote(n, x_dev, rows, threshold, counts_dev, psums_dev, s);
    }
    const uint64_t base = n->ws.row_base;
    if (base != 0 && (rows > chunk_rows(n) || base + rows > n->ws.rows)) { set_error("vote: rows beyond the reserved workspace"); return SZH_EINVAL; }
    int rc = ensure_workspace(n, base + rows);
    if (rc) return rc;
    const float* h2 = n->ws.h2 + base * n->h2;
    float* p = n->ws.p + base * n->out;
    for (uint64_t r0 = 0; r0 < rows; r0 += chunk_rows(n)) {
        const uint64_t R = std::min<uint64_t>(chunk_rows(n), rows - r0);
        rc = forward_hidden(n, x_dev + r0 * n->in, R, false, s);
        if (rc) return rc;
        const int ncb = ((int)n->out + 15) / 16;
        if (ncb <= 4 && n->h2 % 16 == 0 && vote_lds_bytes((int)n->h2, ncb) <= kVoteLdsMax) {
            const float* P = n->d_params;
            const float *w3 = P + n->off_w3(), *b3 = P + n->off_b3();
            switch (ncb) {
            case 1: rc = launch_vote_head<1>(h2, R, (int)n->h2, w3, b3, (int)n->out, threshold, counts_dev, psums_dev, s); break;
            case 2: rc = launch_vote_head<2>(h2, R, (int)n->h2, w3, b3, (int)n->out, threshold, counts_dev, psums_dev, s); break;
            case 3: rc = launch_vote_head<3>(h2, R, (int)n->h2, w3, b3, (int)n->out, threshold, counts_dev, psums_dev, s); break;
            default: rc = launch_vote_head<4>(h2, R, (int)n->h2, w3, b3, (int)n->out, threshold, counts_dev, psums_dev, s); break;
            }
            if (rc) return rc;
            continue;
        }
        rc = forward_logits(n, R, p, s);
        if (rc) return rc;
        rc = launch_softmax(p, R, (int)n->out, 0, nullptr, nullptr, nullptr, nullptr, s);
        if (rc) return rc;
        if (counts_dev) {
            hipLaunchKernelGGL(vote_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), n->out * sizeof(unsigned int), s,
                               p, R, (int)n->out, threshold, counts_dev);
            SZ_HIP(hipGetLastError());
        }
        if (psums_dev) {
            hipLaunchKernelGGL(prob_sum_kernel, dim3((unsigned)std::min<uint64_t>((R + 63) / 64, 1024)), dim3(256), 0, s, p, R,
                               (int)n->out, psums_dev);
            SZ_HIP(hipGetLastError());
        }
    }
    return SZH_OK;
}

}  // namespace sz

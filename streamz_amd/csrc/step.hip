// The reference's own training schedule (lib.rs:599-622: shuffle, chunks of batch_size <= 8
// windows, one optimiser step per chunk) as ONE kernel launch per pretrain_from_features call.
//
// A step on 8 rows is ~7 MFLOP: far too little for a launch per layer, and the steps are
// strictly sequential.  So a small grid of workgroups stays resident for the whole call and
// splits the first hidden layer: workgroup b owns hidden-1 units [b*HS, (b+1)*HS), i.e. the
// columns of w1, the entries of b1 and the ROWS of w2 that belong to them, and keeps them in LDS
// from the first step to the last.  Per step:
//   1. every workgroup prepares the same <= 8 input rows (gather by permutation, dropout)
//   2. h1[:, own] = relu(x . w1[:, own] + b1[own])                       (own data only)
//   3. partial h2 = h1[:, own] . w2[own, :]  -> global slot of this workgroup
//   4. ONE grid-wide barrier
//   5. every workgroup sums the partials in the same order -> h2, then runs the output layer
//      redundantly (w3, b3, b2 are replicated and receive bit-identical updates everywhere):
//      softmax, loss, delta3, delta2                                      (lib.rs:1022-1034)
//   6. delta1[:, own] = delta2 . w2[own, :]^T, then the updates of everything the workgroup
//      owns or replicates, with the pre-update weights throughout        (lib.rs:1029-1059)
// The partial slots are double-buffered, so one barrier per step is enough.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include "internal.h"
#include "prof.h"
#include "rng.h"
#include "streamz_hip.h"

namespace sz {

namespace {

constexpr int kStepThreads = 256;
constexpr int kStepRows = 8;   // most rows of one optimiser step
constexpr int kStepHS = 32;    // hidden-1 units per workgroup
constexpr int kStepMaxC = 32;
constexpr int kStepLdW1 = kStepHS + 16;  // row stride of w1 in LDS: rows k and k+1 sit 16 banks apart

struct StepArgs {
    float* params;           // [w1|b1|w2|b2|w3|b3]
    int in, h1, h2, out;
    const float* win;        // [n_win][in]
    uint32_t n_win;
    const uint32_t* perm;    // [epochs][n_win]
    const uint8_t* keep;     // [epochs][n_win][in] explicit dropout masks, or null
    float dropout;
    uint64_t seed;
    const float* target;     // [out]
    uint32_t epochs, bs;
    float lr;
    float* part;             // [2][workgroups][8][h2]
    unsigned* counter;       // grid barrier: one arrival count per workgroup, zero at launch
    int* abort_flag;         // set when a barrier gives up (never in a healthy run)
    double* totals;          // += {loss, live rows}
};

// All workgroups arrive (barrier_arrive), then all leave (barrier_wait).  Workgroup b publishes the number of barriers it has
// reached in flags[b] (a plain release store: no read-modify-write that would queue up behind the
// other workgroups' at one address); one lane per workgroup then waits for that workgroup's flag.
__device__ void barrier_arrive(unsigned* flags, unsigned arrived)
{
    __threadfence();   // this thread's global writes are visible device-wide
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], arrived, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ bool barrier_wait(unsigned* flags, int n_blocks, unsigned arrived, int* abort_flag)
{
    __shared__ int ok;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    if ((int)threadIdx.x < n_blocks) {
        unsigned spins = 0;
        while (__hip_atomic_load(&flags[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < arrived) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
    }
    __syncthreads();
    __threadfence();   // later reads see what the other workgroups wrote before arriving
    return ok != 0;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// value of the same register in another lane of the row of 16 (DPP), for the softmax reductions
template <int CTRL>
__device__ __forceinline__ float dpp16(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// tanh from the hardware exponential and reciprocal; absolute error ~1e-7
__device__ __forceinline__ float step_tanh(float x)
{
    const float t = __expf(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// LDS layout, shared by the kernel and the host-side size computation (floats, every piece on a
// 16-byte boundary)
struct StepLds {
    int w1, b1, w2, b2, w3, b3, tgt, xs, h1, d1, h2, d2, d3, red, total;
    int ld2, ld3, ldd3, cw, k4;
    __host__ __device__ StepLds(int in, int h2n, int c)
    {
        cw = (c + 15) / 16 * 16;   // class columns, padded to whole 16-wide MFMA tiles
        ld2 = h2n + 1;             // odd strides: rows read by different lanes hit different banks
        ld3 = cw + 1;
        ldd3 = cw + 4;             // delta3 rows are read as float4
        k4 = (in + 3) & ~3;        // K of the first layer, padded to whole MFMA steps (rows >= in are zero)
        int at = 0;
        auto take = [&](int n) { const int p = at; at += (n + 3) & ~3; return p; };
        w1 = take(k4 * kStepLdW1); b1 = take(kStepHS); w2 = take(kStepHS * ld2); b2 = take(h2n);
        w3 = take(h2n * ld3); b3 = take(cw); tgt = take(cw); xs = take(kStepRows * in);
        h1 = take(kStepHS * kStepRows); d1 = take(kStepRows * kStepHS);
        h2 = take(16 * ld2); d2 = take(16 * ld2); d3 = take(kStepRows * ldd3);
        red = take(4 * kStepRows * std::max(std::max(h2n, kStepHS), cw));
        total = at;
    }
};

__global__ __launch_bounds__(kStepThreads) void step_kernel(StepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, g4 = lane >> 4;
    const int NB = gridDim.x, blk = blockIdx.x;
    const int IN = a.in, H2 = a.h2, C = a.out, HS = kStepHS;
    const StepLds L(IN, H2, C);
    const int LD2 = L.ld2, LD3 = L.ld3, LDD3 = L.ldd3, CW = L.cw, K4 = L.k4;
    constexpr int LDW1 = kStepLdW1;
    const int i0 = blk * HS;
    float* __restrict__ W1s = sm + L.w1;   // [K4][LDW1], rows >= IN zero
    float* __restrict__ b1s = sm + L.b1;   // [HS]
    float* __restrict__ W2s = sm + L.w2;   // [HS][LD2]
    float* __restrict__ b2s = sm + L.b2;   // [H2]
    float* __restrict__ W3s = sm + L.w3;   // [H2][LD3], columns >= C zero
    float* __restrict__ b3s = sm + L.b3;   // [CW]
    float* __restrict__ tgt = sm + L.tgt;  // [CW]
    float* __restrict__ xs = sm + L.xs;    // [8][IN]
    float* __restrict__ h1s = sm + L.h1;   // [HS][8]   (unit-major: 8 rows contiguous)
    float* __restrict__ d1s = sm + L.d1;   // [8][HS]
    float* __restrict__ h2s = sm + L.h2;   // [16][LD2], rows 8..15 stay zero (MFMA tiles are 16 rows)
    float* __restrict__ d2s = sm + L.d2;   // [16][LD2], rows 8..15 stay zero
    float* __restrict__ d3s = sm + L.d3;   // [8][LDD3]
    float* __restrict__ red = sm + L.red;  // per-wave partial results of the K-split products
    __shared__ int live_s[kStepRows];
    __shared__ int n_live_s;

    const float* P = a.params;
    const size_t off_b1 = (size_t)IN * a.h1, off_w2 = off_b1 + a.h1, off_b2 = off_w2 + (size_t)a.h1 * H2,
                 off_w3 = off_b2 + H2, off_b3 = off_w3 + (size_t)H2 * C;
    for (int e = tid; e < K4 * HS; e += kStepThreads) { const int j = e / HS, i = e - j * HS; W1s[j * LDW1 + i] = j < IN ? P[(size_t)j * a.h1 + i0 + i] : 0.0f; }
    for (int e = tid; e < HS; e += kStepThreads) b1s[e] = P[off_b1 + i0 + e];
    for (int e = tid; e < HS * H2; e += kStepThreads) { const int i = e / H2, k = e - i * H2; W2s[i * LD2 + k] = P[off_w2 + (size_t)(i0 + i) * H2 + k]; }
    for (int e = tid; e < H2; e += kStepThreads) b2s[e] = P[off_b2 + e];
    for (int e = tid; e < H2 * LD3; e += kStepThreads) { const int k = e / LD3, c = e - k * LD3; W3s[e] = c < C ? P[off_w3 + (size_t)k * C + c] : 0.0f; }
    for (int e = tid; e < CW; e += kStepThreads) { b3s[e] = e < C ? P[off_b3 + e] : 0.0f; tgt[e] = e < C ? a.target[e] : 0.0f; }
    for (int e = tid; e < 16 * LD2; e += kStepThreads) { h2s[e] = 0.0f; d2s[e] = 0.0f; }

    const uint32_t steps_per_epoch = (a.n_win + a.bs - 1) / a.bs;
    const uint64_t n_steps = (uint64_t)a.epochs * steps_per_epoch;
    double loss_acc = 0.0, live_acc = 0.0;

    // inputs of a step: element e = r * IN + j, two per thread (8 x 60 = 480 <= 512)
    auto load_x = [&](uint64_t step, float (&xv)[2]) {
        const uint32_t ep = (uint32_t)(step / steps_per_epoch);
        const uint32_t c0 = (uint32_t)(step - (uint64_t)ep * steps_per_epoch) * a.bs;
        const uint32_t rows = min(a.bs, a.n_win - c0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + q * kStepThreads;
            const int r = e / IN, j = e - r * IN;
            float v = 0.0f;
            if (r < (int)rows) {
                const uint64_t at = (uint64_t)ep * a.n_win + c0 + r;
                v = a.win[(size_t)a.perm[at] * IN + j];
                bool drop = false;
                if (a.keep) drop = a.keep[at * IN + j] == 0;
                else if (a.dropout > 0.0f) drop = uniform01(a.seed, at, (uint64_t)j) < a.dropout;
                if (drop) v = 0.0f;
            }
            xv[q] = v;
        }
    };
    float xv[2];
    if (n_steps > 0) load_x(0, xv);
    __syncthreads();

    const int kq = H2 / 4;  // K range of one wave in the K-split products
    bool healthy = true;
    for (uint64_t step = 0; step < n_steps && healthy; ++step) {
        // 1. rows of this step -> LDS; a row counts when something survived the dropout (lib.rs:607-609)
        if (tid < kStepRows) live_s[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + q * kStepThreads;
            if (e < kStepRows * IN) {
                xs[e] = xv[q];
                if (xv[q] != 0.0f) live_s[e / IN] = 1;  // benign race: every writer stores 1
            }
        }
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int r = 0; r < kStepRows; ++r) n += live_s[r];
            n_live_s = n;
        }
        // 2. own hidden-1 units on the matrix cores: rows x units, K = inputs; wave w takes units
        // [16 w, 16 w + 16).  Result register v of lane (unit m16, g4) is row 4 g4 + v.
        if (wave < HS / 16) {
            const int ib = wave * 16;
            // all fragments first (at most 16 steps: 64 inputs), then the chain of MFMAs
            float av[16], bv[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int k = 4 * st + g4;
                av[st] = (m16 < kStepRows && k < IN) ? xs[m16 * IN + k] : 0.0f;
                bv[st] = k < K4 ? W1s[k * LDW1 + ib + m16] : 0.0f;
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 16; st += 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bv[st], acc, 0, 0, 0);  // zero fragments past K4
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st + 1], bv[st + 1], acc2, 0, 0, 0);
            }
            acc += acc2;
            if (g4 < 2) {
                const float b = b1s[ib + m16];
                float4 o;
                o.x = fmaxf(acc[0] + b, 0.0f); o.y = fmaxf(acc[1] + b, 0.0f);
                o.z = fmaxf(acc[2] + b, 0.0f); o.w = fmaxf(acc[3] + b, 0.0f);
                *reinterpret_cast<float4*>(h1s + (ib + m16) * kStepRows + 4 * g4) = o;
            }
        }
        __syncthreads();
        // 3. partial second layer: thread = column k
        const unsigned buf = (unsigned)(step & 1);
        float* my_part = a.part + ((size_t)buf * NB + blk) * kStepRows * H2;
        {
            float af[kStepHS / 4];  // h1[row m16][unit 4 st + g4], shared by every column tile
#pragma unroll
            for (int st = 0; st < kStepHS / 4; ++st) af[st] = m16 < kStepRows ? h1s[(4 * st + g4) * kStepRows + m16] : 0.0f;
            const int nt = H2 / 16;  // column tiles; wave w takes tiles w, w + 4, w + 8, w + 12
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < kStepHS / 4; ++st)
#pragma unroll
                for (int u = 0; u < 4; ++u)  // a tile index past the end is clamped and its result dropped
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st], W2s[(4 * st + g4) * LD2 + 16 * min(wave + 4 * u, nt - 1) + m16], acc[u], 0, 0, 0);
            if (g4 < 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (wave + 4 * u < nt) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) my_part[(4 * g4 + v) * H2 + 16 * (wave + 4 * u) + m16] = acc[u][v];
                    }
            }
        }
        // 4. everybody's partials are out
        barrier_arrive(a.counter, (unsigned)(step + 1));
        // the next step's inputs (index arithmetic, dropout draws, loads) while the others arrive
        if (step + 1 < n_steps) load_x(step + 1, xv);
        healthy = barrier_wait(a.counter, NB, (unsigned)(step + 1), a.abort_flag);
        if (!healthy) break;
        // 5a. h2 = tanh(b2 + sum of the partials).  Wave w sums workgroups w, w+4, ... with 16-byte
        // loads (4 columns per lane), all issued before the first add; the four waves' sums are
        // then combined in a fixed order, the same in every workgroup.
        {
            const float* pb = a.part + (size_t)buf * NB * kStepRows * H2;
            const int c4 = 4 * lane;
            float4 s[kStepRows];
#pragma unroll
            for (int r = 0; r < kStepRows; ++r) s[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 < H2) {
                for (int b0 = wave; b0 < NB; b0 += 16) {
                    float4 v[4][kStepRows];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int r = 0; r < kStepRows; ++r)
                            v[q][r] = b0 + 4 * q < NB
                                          ? *reinterpret_cast<const float4*>(pb + ((size_t)(b0 + 4 * q) * kStepRows + r) * H2 + c4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int r = 0; r < kStepRows; ++r) {
                            s[r].x += v[q][r].x; s[r].y += v[q][r].y; s[r].z += v[q][r].z; s[r].w += v[q][r].w;
                        }
                }
#pragma unroll
                for (int r = 0; r < kStepRows; ++r)
                    *reinterpret_cast<float4*>(red + ((size_t)wave * kStepRows + r) * H2 + c4) = s[r];
            }
        }
        __syncthreads();
        if (tid < H2) {
#pragma unroll
            for (int r = 0; r < kStepRows; ++r) {
                const float t = (red[(0 * kStepRows + r) * H2 + tid] + red[(1 * kStepRows + r) * H2 + tid]) +
                                (red[(2 * kStepRows + r) * H2 + tid] + red[(3 * kStepRows + r) * H2 + tid]);
                h2s[r * LD2 + tid] = step_tanh(b2s[tid] + t);
            }
        }
        __syncthreads();
        // 5b. logits = h2 . w3 on the matrix cores, K split over the four waves
        for (int cb = 0; cb < CW; cb += 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int kb = wave * kq + g4;
#pragma unroll 4
            for (int st = 0; st < kq; st += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(h2s[m16 * LD2 + kb + st], W3s[(kb + st) * LD3 + cb + m16], acc, 0, 0, 0);
            if (g4 < 2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) red[((size_t)wave * kStepRows + 4 * g4 + v) * CW + cb + m16] = acc[v];
            }
        }
        __syncthreads();
        // softmax, loss with the pre-update weights, delta3 = o - t (lib.rs:611-615, 1028):
        // thread = (row, class); a row's classes sit in 16 (or 32) neighbouring lanes
        if (tid < kStepRows * CW) {
            const int r = tid / CW, c = tid - r * CW;
            const bool cls = c < C, alive = live_s[r] != 0;
            const float z = cls ? b3s[c] + ((red[(0 * kStepRows + r) * CW + c] + red[(1 * kStepRows + r) * CW + c]) +
                                            (red[(2 * kStepRows + r) * CW + c] + red[(3 * kStepRows + r) * CW + c]))
                                : -INFINITY;
            float mx = z;
            mx = fmaxf(mx, dpp16<0xB1>(mx)); mx = fmaxf(mx, dpp16<0x4E>(mx));
            mx = fmaxf(mx, dpp16<0x141>(mx)); mx = fmaxf(mx, dpp16<0x140>(mx));
            if (CW > 16) mx = fmaxf(mx, __shfl_xor(mx, 16));
            const float e = cls ? expf(z - mx) : 0.0f;
            float sum = e;
            sum += dpp16<0xB1>(sum); sum += dpp16<0x4E>(sum); sum += dpp16<0x141>(sum); sum += dpp16<0x140>(sum);
            if (CW > 16) sum += __shfl_xor(sum, 16);
            const float o = e / sum, t = tgt[c];
            if (cls && alive && t != 0.0f) loss_acc -= (double)(t * logf(fmaxf(o, 1e-12f)));
            if (c == 0 && alive) live_acc += 1.0;
            d3s[r * LDD3 + c] = (cls && alive) ? o - t : 0.0f;
        }
        __syncthreads();
        const int n_live = n_live_s;
        const float scale = n_live > 0 ? a.lr / (float)n_live : 0.0f;  // lib.rs:1047
        // delta2 = (delta3 . w3^T)(1 - h2^2), lib.rs:1034, on the matrix cores: rows x columns of
        // h2, K = classes; column tiles are dealt to the waves
        {
            float af[kStepMaxC / 4];  // delta3[row m16][class 4 st + g4]
#pragma unroll
            for (int st = 0; st < kStepMaxC / 4; ++st)  // zero past the classes: those steps add nothing
                af[st] = (4 * st < CW && m16 < kStepRows) ? d3s[m16 * LDD3 + 4 * st + g4] : 0.0f;
            const int nt = H2 / 16;
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < kStepMaxC / 4; ++st)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st], W3s[(16 * min(wave + 4 * u, nt - 1) + m16) * LD3 + min(4 * st + g4, CW - 1)], acc[u], 0, 0, 0);
            if (g4 < 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (wave + 4 * u < nt) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int at = (4 * g4 + v) * LD2 + 16 * (wave + 4 * u) + m16;
                            const float h = h2s[at];
                            d2s[at] = acc[u][v] * (1.0f - h * h);
                        }
                    }
            }
        }
        __syncthreads();
        if (n_live > 0) {
            // 6a. delta1 of the own units, delta2 . w2[own, :]^T with the old w2, on the matrix
            // cores: K split over the waves, two tiles of 16 units
            {
                static_assert(kStepHS == 32, "two tiles of 16 units");
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const int kb = wave * kq + g4;
#pragma unroll 4
                for (int st = 0; st < kq; st += 4) {
                    const float dv = d2s[m16 * LD2 + kb + st];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, W2s[m16 * LD2 + kb + st], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, W2s[(16 + m16) * LD2 + kb + st], acc1, 0, 0, 0);
                }
                if (g4 < 2) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        red[((size_t)wave * kStepRows + 4 * g4 + v) * HS + m16] = acc0[v];
                        red[((size_t)wave * kStepRows + 4 * g4 + v) * HS + 16 + m16] = acc1[v];
                    }
                }
            }
            // 6b. replicated output layer: dW3 = h2^T . delta3 on the matrix cores (rows of w3 x
            // classes, K = the 8 rows); register v of lane (class m16, g4) is row 16 t + 4 g4 + v
            for (int cb = 0; cb < CW; cb += 16) {
                const int nt = H2 / 16;
                float bf[kStepRows / 4];
#pragma unroll
                for (int st = 0; st < kStepRows / 4; ++st) bf[st] = d3s[(4 * st + g4) * LDD3 + cb + m16];
                f32x4 acc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < kStepRows / 4; ++st)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(h2s[(4 * st + g4) * LD2 + 16 * min(wave + 4 * u, nt - 1) + m16], bf[st], acc[u], 0, 0, 0);
                if (cb + m16 < C) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (wave + 4 * u < nt) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) W3s[(16 * (wave + 4 * u) + 4 * g4 + v) * LD3 + cb + m16] -= scale * acc[u][v];
                        }
                }
            }
            if (tid < H2) {
                float g2 = 0.0f;
#pragma unroll
                for (int r = 0; r < kStepRows; ++r) g2 += d2s[r * LD2 + tid];
                b2s[tid] -= scale * g2;
            }
            if (tid < C) {
                float g = 0.0f;
#pragma unroll
                for (int r = 0; r < kStepRows; ++r) g += d3s[r * LDD3 + tid];
                b3s[tid] -= scale * g;
            }
            __syncthreads();
            {   // delta1: combine the waves' parts, ReLU derivative (lib.rs:1039-1040)
                const int r = tid / HS, i = tid - r * HS;
                const float s = (red[(0 * kStepRows + r) * HS + i] + red[(1 * kStepRows + r) * HS + i]) +
                                (red[(2 * kStepRows + r) * HS + i] + red[(3 * kStepRows + r) * HS + i]);
                d1s[r * HS + i] = h1s[i * kStepRows + r] > 0.0f ? s : 0.0f;
            }
            // 6c. own rows of w2 -= scale * h1^T . delta2 on the matrix cores (units x columns, K = the
            // 8 rows); register v of lane (column m16, g4) is unit 16 ib + 4 g4 + v
            {
                float af[2][kStepRows / 4];  // h1[row 4 st + g4][unit 16 ib + m16]
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int st = 0; st < kStepRows / 4; ++st) af[ib][st] = h1s[(16 * ib + m16) * kStepRows + 4 * st + g4];
                for (int t = wave; t < H2 / 16; t += 4) {
                    const float b0 = d2s[g4 * LD2 + 16 * t + m16], b1 = d2s[(4 + g4) * LD2 + 16 * t + m16];
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ib][0], b0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ib][1], b1, acc, 0, 0, 0);
#pragma unroll
                        for (int v = 0; v < 4; ++v) W2s[(16 * ib + 4 * g4 + v) * LD2 + 16 * t + m16] -= scale * acc[v];
                    }
                }
            }
            __syncthreads();
            // own columns of w1 -= scale * x^T . delta1 (inputs x units, K = the 8 rows), own b1
            {
                float bf[2][kStepRows / 4];  // delta1[row 4 st + g4][unit 16 ib + m16]
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int st = 0; st < kStepRows / 4; ++st) bf[ib][st] = d1s[(4 * st + g4) * HS + 16 * ib + m16];
                for (int tj = wave; tj < (IN + 15) / 16; tj += 4) {
                    const int j = 16 * tj + m16;
                    const float a0 = j < IN ? xs[g4 * IN + j] : 0.0f, a1 = j < IN ? xs[(4 + g4) * IN + j] : 0.0f;
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bf[ib][0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf[ib][1], acc, 0, 0, 0);
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int jj = 16 * tj + 4 * g4 + v;
                            if (jj < IN) W1s[jj * LDW1 + 16 * ib + m16] -= scale * acc[v];
                        }
                    }
                }
            }
            if (tid < HS) {
                float g = 0.0f;
#pragma unroll
                for (int r = 0; r < kStepRows; ++r) g += d1s[r * HS + tid];
                b1s[tid] -= scale * g;
            }
        }
        __syncthreads();
    }

    // write the owned parameters back; workgroup 0 also writes the replicated ones and the totals
    float* Pw = a.params;
    for (int e = tid; e < IN * HS; e += kStepThreads) { const int j = e / HS, i = e - j * HS; Pw[(size_t)j * a.h1 + i0 + i] = W1s[j * LDW1 + i]; }
    for (int e = tid; e < HS; e += kStepThreads) Pw[off_b1 + i0 + e] = b1s[e];
    for (int e = tid; e < HS * H2; e += kStepThreads) { const int i = e / H2, k = e - i * H2; Pw[off_w2 + (size_t)(i0 + i) * H2 + k] = W2s[i * LD2 + k]; }
    if (blk == 0) {
        for (int e = tid; e < H2; e += kStepThreads) Pw[off_b2 + e] = b2s[e];
        for (int e = tid; e < H2 * C; e += kStepThreads) { const int k = e / C, c = e - k * C; Pw[off_w3 + e] = W3s[k * LD3 + c]; }
        for (int e = tid; e < C; e += kStepThreads) Pw[off_b3 + e] = b3s[e];
        if (loss_acc != 0.0) atomicAdd(&a.totals[0], loss_acc);
        if (live_acc != 0.0) atomicAdd(&a.totals[1], live_acc);
    }
}

size_t step_lds_bytes(int in, int h2, int c) { return (size_t)StepLds(in, h2, c).total * sizeof(float); }

}  // namespace

bool step_kernel_applies(const Net* n, uint32_t batch_size, uint64_t n_steps)
{
    return n_steps * (n->h1 / kStepHS + 1) < 0x7fffffffull && batch_size >= 1 && batch_size <= (uint32_t)kStepRows && n->in >= 1 && n->in <= 64 &&
           kStepRows * n->in <= 2 * kStepThreads && n->h1 % kStepHS == 0 && n->h1 / kStepHS >= 1 && n->h1 / kStepHS <= 64 &&
           n->h2 >= 16 && n->h2 % 16 == 0 && n->h2 <= (uint32_t)kStepThreads && n->out >= 1 && n->out <= (uint32_t)kStepMaxC &&
           kStepHS * kStepRows == kStepThreads;
}

int net_pretrain_steps(Net* n, const float* d_win, uint32_t n_win, const uint32_t* d_perm, const uint8_t* d_keep,
                       float dropout, uint64_t seed, const float* d_target, uint32_t epochs, uint32_t batch_size,
                       float lr, hipStream_t s)
{
    const int nb = (int)(n->h1 / kStepHS);
    struct Scratch { float* part = nullptr; unsigned* counter = nullptr; int* abort_flag = nullptr; };
    Scratch sc;
    const size_t part_bytes = 2ull * nb * kStepRows * n->h2 * sizeof(float);
    SZ_HIP(hipMalloc((void**)&sc.part, part_bytes + 64 + 64 * sizeof(unsigned)));
    sc.counter = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(sc.part) + part_bytes);
    sc.abort_flag = reinterpret_cast<int*>(sc.counter + 64);   // after the (at most 64) workgroup flags
    hipError_t e = hipMemsetAsync(sc.counter, 0, 64 + 64 * sizeof(unsigned), s);
    StepArgs a{};
    a.params = n->d_params;
    a.in = (int)n->in; a.h1 = (int)n->h1; a.h2 = (int)n->h2; a.out = (int)n->out;
    a.win = d_win; a.n_win = n_win; a.perm = d_perm; a.keep = d_keep; a.dropout = dropout; a.seed = seed;
    a.target = d_target; a.epochs = epochs; a.bs = batch_size; a.lr = lr;
    a.part = sc.part; a.counter = sc.counter; a.abort_flag = sc.abort_flag;
    a.totals = n->d_stats + 2;
    const size_t lds = step_lds_bytes(a.in, a.h2, a.out);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int abort_host = 0;
    if (e == hipSuccess) {
        ProfScope prof("steps", s);
        hipLaunchKernelGGL(step_kernel, dim3(nb), dim3(kStepThreads), lds, s, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&abort_host, sc.abort_flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(sc.part);
    if (e != hipSuccess) return hip_fail(e, "step kernel");
    if (abort_host) { set_error("training step kernel: grid barrier timed out"); return SZH_ENODEVICE; }
    return SZH_OK;
}

}  // namespace sz

// The reference's own training schedule (lib.rs:599-622: shuffle, chunks of batch_size <= 8
// windows, one optimiser step per chunk) as ONE kernel launch per pretrain_from_features call.
//
// A step on 8 rows is ~7 MFLOP: far too little for a launch per layer, and the steps are
// strictly sequential.  So a small grid of workgroups stays resident for the whole call and
// splits the first hidden layer: workgroup b owns hidden-1 units [32 b, 32 b + 32), i.e. the
// columns of w1, the entries of b1 and the ROWS of w2 that belong to them, and keeps them in LDS
// from the first step to the last.  Per step:
//   1. every workgroup prepares the same <= 8 input rows (gather by permutation, dropout)
//   2. h1[:, own] = relu(x . w1[:, own] + b1[own])                       (own data only)
//   3. partial h2 = h1[:, own] . w2[own, :]  -> this workgroup's slot in global memory
//   4. ONE exchange: every workgroup reads every slot
//   5. every workgroup sums the partials in the same order -> h2, then runs the output layer
//      redundantly (w3, b3, b2 are replicated and receive bit-identical updates everywhere):
//      softmax, loss, delta3, delta2                                      (lib.rs:1022-1034)
//   6. delta1[:, own] = delta2 . w2[own, :]^T, then the updates of everything the workgroup
//      owns or replicates, with the pre-update weights throughout        (lib.rs:1029-1059)
// The exchange costs no cache-wide fence.  When the workgroups fit one XCD (they share its L2) they
// ADD their partials into one array there with 64-bit fixed-point atomics whose top bits count the
// arrivals, so steps 3-5 are "add; read until complete" with no flag and no drain (see ONEX
// below).  Otherwise the slots are written through to memory (sc1 stores), every storing wave
// drains its stores, one lane raises the workgroup's flag (sc1), the readers poll the flags and
// read the slots with sc1 loads only, which bypass their CU's L1; slots are double-buffered, so
// one exchange per step is enough.  Every product runs on the 16x16x4 f32
// matrix instruction with the 8 rows padded to a 16-row tile (padding rows are kept zero in LDS,
// so no load is conditional).
//
// The kernel writes the trained parameters to a shadow vector; the host adopts it only if no
// workgroup gave up waiting (bounded by wall-clock time), so a call either trains completely or
// leaves the parameters untouched and the caller takes the per-layer path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "internal.h"
#include "prof.h"
#include "rng.h"
#include "streamz_hip.h"

namespace sz {

namespace {

constexpr int kStepThreads = 256;
constexpr int kStepRows = 8;    // most rows of one optimiser step
constexpr int kStepHS = 32;     // hidden-1 units per workgroup
constexpr int kStepMaxC = 32;
constexpr int kStepIn = 60;     // inputs (pretrain_from_features works on 60 features)
constexpr int kStepLdW1 = kStepHS + 16;  // row stride of w1 in LDS: rows k and k+1 sit 16 banks apart
constexpr int kStepLdH1 = kStepHS + 4;  // h1 is kept row-major [16 rows][units] (rows 8..15 zero); 36 = 4 mod 32
constexpr int kStepXs = 16 * kStepIn + 16;  // one buffer of input rows [16][60] + pad (floats)
constexpr int kStepMaxBlocks = 64;
constexpr unsigned long long kStepTimeoutTicks = 200000000ull;  // 2 s of the 100 MHz wall clock

struct StepArgs {
    const float* params;     // [w1|b1|w2|b2|w3|b3]
    float* params_out;       // shadow vector, written at the end of a healthy run
    int h1, h2, out;
    const float* win;        // [n_win][60]
    uint32_t n_win;
    const uint32_t* perm;    // [epochs][n_win]
    const uint8_t* keep;     // [epochs][n_win][60] explicit dropout masks, or null
    float dropout;
    uint64_t seed;
    uint64_t key_row0;       // dropout draws are keyed by (seed, key_row0 + position in perm, feature): the rows before this launch's first epoch
    const float* target;     // [out]
    uint32_t epochs, bs;
    float lr;
    float* part;             // [2][workgroups][h2][8] partial second-layer sums
    unsigned* flags;         // one arrival count per workgroup, zero at launch
    unsigned* ticket;        // one-XCD launch: roles handed out in arrival order (zero at launch)
    unsigned long long* acc; // one-XCD launch: [3][h2][8] fixed-point sums of the partial second layer (zero at launch)
    int nb;                  // workgroups that take part (the grid itself in the ordinary launch)
    int* abort_flag;         // set when a workgroup gives up waiting (never in a healthy run)
    int force_abort;         // test hook: the last workgroup reports a time-out before its first step
    double* totals;          // {loss, live rows} of this call (zero at launch)
    unsigned long long* prof;  // profiled build only: per-phase 100 MHz ticks of workgroup 0
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define SZ_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)
// The compiler sinks every LDS read to just above its use, which makes each matrix instruction
// wait a full LDS round trip.  Phases are therefore written as "all reads; SZ_SCHED_FENCE; all
// matrix instructions; SZ_SCHED_FENCE; epilogue", and the fence keeps the groups apart.
#define SZ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// value of the same register in another lane of the row of 16 (DPP), for the softmax reductions
template <int CTRL>
__device__ __forceinline__ float dpp16(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// tanh from the hardware exponential and reciprocal; absolute error ~1e-7; a NaN stays a NaN (the clamp
// alone would turn it into -1, where the reference's tanh, lib.rs:883, carries it on)
__device__ __forceinline__ float step_tanh(float x)
{
    const float t = __expf(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
    const float y = 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
    return x != x ? x : y;
}

// LDS layout, shared by the kernel and the host-side size computation (floats, every piece on a
// 16-byte boundary)
struct StepLds {
    int w1, b1, w2, b2, w3, b3, tgt, xs, h1, d1, h2, d2, d3, red, red2, total;
    int ld2, ld3, ldd3, cw;
    __host__ __device__ StepLds(int h2n, int cwn)
    {
        cw = cwn;                  // class columns, padded to whole 16-wide MFMA tiles
        ld2 = h2n + 4;             // = 4 mod 64: rows 4 g + j of a fragment and unit-major reads hit distinct banks
        ld3 = h2n + 4;             // w3 is kept transposed, [class][hidden2]
        ldd3 = cw + 4;
        int at = 0;
        auto take = [&](int n) { const int p = at; at += (n + 3) & ~3; return p; };
        w1 = take(kStepIn * kStepLdW1); b1 = take(kStepHS); w2 = take(kStepHS * ld2); b2 = take(h2n);
        w3 = take(cw * ld3); b3 = take(cw); tgt = take(cw); xs = take(2 * kStepXs);
        h1 = take(16 * kStepLdH1); d1 = take(kStepRows * kStepHS);
        h2 = take(16 * ld2); d2 = take(16 * ld2); d3 = take(4 * 16 * ldd3);
        red = take(4 * kStepRows * cw); red2 = take(4 * kStepRows * kStepHS);
        total = at;
    }
};

constexpr int kStepProfSlots = 16;

// H2T / CWT: hidden2 and the padded class count as compile-time constants (the shapes the
// reference's command line uses), or 0 for "read them from the arguments".
//
// Wave w owns the 16-column tiles [w TPW, (w + 1) TPW) of the second layer (TPW = tiles / 4, rounded
// up) in every phase that is indexed by those columns: the sums of the exchange, its share of the
// K-split products (logits, delta1), delta2, dW3 (the same rows of w3), b2 and dW2.  What one
// phase writes there, only the same wave reads in the next, so those phases follow each other
// without a workgroup barrier (a wavefront's LDS operations execute in program order).
//
// ONEX: every participating workgroup sits on the same XCD.  The grid is 8 x the participants
// (workgroups go to the 8 XCDs round-robin in launch order); workgroups that find themselves on
// another XCD leave at once, those on XCD 0 take their role (which hidden-1 units they own) from a
// ticket counter.  CUs of one XCD share one L2, where atomics without device scope execute: the
// sum of the partials is formed there by integer atomics instead of every workgroup reading every
// other's slot (16 KB read per workgroup and step instead of 128 KB at 16 workgroups).
template <int H2T, int CWT, bool PROF, bool ONEX>
__global__ __launch_bounds__(kStepThreads) void step_kernel(StepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int role_s;
    int role = (int)blockIdx.x;
    if (ONEX) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(xcc));  // XCC_ID
        if (threadIdx.x == 0) atomicAdd(a.flags + 40 + (xcc & 15u), 1u);   // arrivals per XCD (diagnostic, unused flag words)
        if (xcc != 0) return;
        if (threadIdx.x == 0) role_s = (int)atomicAdd(a.ticket, 1u);
        __syncthreads();
        role = role_s;
        if (role >= a.nb) return;
    }
    // PROF: wall-clock ticks spent up to each mark, summed over the steps (uniform, scalar registers)
    unsigned long long pacc[kStepProfSlots] = {};
    unsigned long long pt0 = 0;
#define STEP_MARK(i) do { if (PROF) { __syncthreads(); const unsigned long long t__ = wall_clock64(); pacc[i] += t__ - pt0; pt0 = t__; } } while (0)
    const int tid = threadIdx.x, lane = tid & 63, m16 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < kStepThreads / 64);
    const int NB = ONEX ? a.nb : (int)gridDim.x, blk = role;
    constexpr int IN = kStepIn, HS = kStepHS, LDW1 = kStepLdW1, LDH1 = kStepLdH1, TPWMAX = 4;
    const int H2 = H2T ? H2T : a.h2;
    const int C = a.out;
    const int CW = CWT ? CWT : (C + 15) / 16 * 16;
    const int NT = H2 / 16;              // 16-column tiles of the second layer
    const int TPW = (NT + 3) / 4;        // tiles per wave
    const int t0 = wave * TPW;           // this wave's first tile
    const int ntw = max(0, min(TPW, NT - t0));  // and how many it owns
    const StepLds L(H2, CW);
    const int LD2 = L.ld2, LD3 = L.ld3, LDD3 = L.ldd3;
    const int i0 = blk * HS;
    float* __restrict__ W1s = sm + L.w1;   // [60][LDW1]
    float* __restrict__ b1s = sm + L.b1;   // [HS]
    float* __restrict__ W2s = sm + L.w2;   // [HS][LD2]
    float* __restrict__ b2s = sm + L.b2;   // [H2]
    float* __restrict__ W3T = sm + L.w3;   // [CW][LD3] = w3 transposed, classes >= C zero
    float* __restrict__ b3s = sm + L.b3;   // [CW]
    float* __restrict__ tgt = sm + L.tgt;  // [CW]
    float* __restrict__ xsb = sm + L.xs;   // 2 x [16][60] (+ pad): this step's rows / the next step's; rows 8..15 stay zero
    float* __restrict__ h1s = sm + L.h1;   // [16][LDH1] row-major, rows 8..15 stay zero
    float* __restrict__ h2s = sm + L.h2;   // [16][LD2], rows 8..15 stay zero
    float* __restrict__ d2s = sm + L.d2;   // [16][LD2], rows 8..15 stay zero
    float* __restrict__ d3w = sm + L.d3 + wave * 16 * LDD3;  // this wave's copy of delta3 [16][LDD3], rows 8..15 zero
    float* __restrict__ red = sm + L.red;  // [4 waves][8][CW] K-split parts of the logits
    float* __restrict__ red2 = sm + L.red2;  // [4 waves][8][HS] K-split parts of delta1
    int* __restrict__ ctl = reinterpret_cast<int*>(sm + L.total);  // [2][8] live rows of this / the next step, [16] all flags seen

    const float* P = a.params;
    const size_t off_b1 = (size_t)IN * a.h1, off_w2 = off_b1 + a.h1, off_b2 = off_w2 + (size_t)a.h1 * H2,
                 off_w3 = off_b2 + H2, off_b3 = off_w3 + (size_t)H2 * C;
    for (int e = tid; e < IN * HS; e += kStepThreads) { const int j = e / HS, i = e - j * HS; W1s[j * LDW1 + i] = P[(size_t)j * a.h1 + i0 + i]; }
    for (int e = tid; e < HS; e += kStepThreads) b1s[e] = P[off_b1 + i0 + e];
    for (int e = tid; e < HS * H2; e += kStepThreads) { const int i = e / H2, k = e - i * H2; W2s[i * LD2 + k] = P[off_w2 + (size_t)(i0 + i) * H2 + k]; }
    for (int e = tid; e < H2; e += kStepThreads) b2s[e] = P[off_b2 + e];
    for (int e = tid; e < CW * H2; e += kStepThreads) { const int c = e / H2, k = e - c * H2; W3T[c * LD3 + k] = c < C ? P[off_w3 + (size_t)k * C + c] : 0.0f; }
    for (int e = tid; e < CW; e += kStepThreads) { b3s[e] = e < C ? P[off_b3 + e] : 0.0f; tgt[e] = e < C ? a.target[e] : 0.0f; }
    for (int e = tid; e < 16 * LD2; e += kStepThreads) { h2s[e] = 0.0f; d2s[e] = 0.0f; }
    for (int e = tid; e < 2 * kStepXs; e += kStepThreads) xsb[e] = 0.0f;
    for (int e = tid; e < 16 * LDH1; e += kStepThreads) h1s[e] = 0.0f;
    for (int e = tid; e < 4 * 16 * LDD3; e += kStepThreads) (sm + L.d3)[e] = 0.0f;
    if (tid < 17) ctl[tid] = tid == 16 ? 1 : 0;

    const uint32_t steps_per_epoch = (a.n_win + a.bs - 1) / a.bs;
    const uint64_t n_steps = (uint64_t)a.epochs * steps_per_epoch;
    double loss_acc = 0.0, live_acc = 0.0;

    // Inputs of a step: element e = r * 60 + j, two per thread (8 x 60 = 480 <= 512).  The row
    // indices (perm) are fetched two steps ahead and the values one step ahead, so that neither of
    // the two dependent loads is waited for; the address arithmetic and the dropout draws sit
    // behind the issue of the exchange's loads, whose latency covers them.
    constexpr int kXE = 2;
    int er[kXE], ej[kXE];
#pragma unroll
    for (int q = 0; q < kXE; ++q) {
        const int e = tid + q * kStepThreads;
        er[q] = e / IN; ej[q] = e - er[q] * IN;
    }
    struct Cursor { uint32_t ep, c0; };  // position of a step in the shuffled epochs
    auto advance = [&](Cursor& c) { c.c0 += a.bs; if (c.c0 >= a.n_win) { c.c0 = 0; c.ep += 1; } };
    auto rows_at = [&](const Cursor& c) -> int { return c.ep < a.epochs ? (int)min(a.bs, a.n_win - c.c0) : 0; };
    auto load_idx = [&](const Cursor& c, uint32_t (&idx)[kXE]) {
        const int rows = rows_at(c);
#pragma unroll
        for (int q = 0; q < kXE; ++q) idx[q] = er[q] < rows ? a.perm[(uint64_t)c.ep * a.n_win + c.c0 + er[q]] : 0xffffffffu;
    };
    auto load_val = [&](const Cursor& c, const uint32_t (&idx)[kXE], float (&xv)[kXE]) {
#pragma unroll
        for (int q = 0; q < kXE; ++q) {
            float v = 0.0f;
            if (idx[q] != 0xffffffffu) {
                const uint64_t at = (uint64_t)c.ep * a.n_win + c.c0 + er[q];
                v = a.win[(size_t)idx[q] * IN + ej[q]];
                bool drop = false;
                if (a.keep) drop = a.keep[at * IN + ej[q]] == 0;
                else if (a.dropout > 0.0f) drop = uniform01(a.seed, a.key_row0 + at, (uint64_t)ej[q]) < a.dropout;
                if (drop) v = 0.0f;
            }
            xv[q] = v;
        }
    };
    // rows of a step -> LDS buffer `b`; a row counts when something survived the dropout (lib.rs:607-609)
    auto stage_rows = [&](int b, const float (&xv)[kXE]) {
#pragma unroll
        for (int q = 0; q < kXE; ++q)
            if (er[q] < kStepRows) {
                xsb[b * kStepXs + er[q] * IN + ej[q]] = xv[q];
                if (xv[q] != 0.0f) ctl[8 * b + er[q]] = 1;  // benign race: every writer stores 1
            }
    };
    Cursor cur_val{0, 0}, cur_idx{0, 0};
    uint32_t idx[kXE];
    float xv[kXE];
    load_idx(cur_idx, idx);
    load_val(cur_val, idx, xv);     // values of step 0
    advance(cur_idx); advance(cur_val);
    load_idx(cur_idx, idx);         // indices of step 1
    advance(cur_idx);
    __syncthreads();
    stage_rows(0, xv);

    // the partial slots and the flags are only ever touched by sc1 (write-through / L1-bypassing) accesses
    const unsigned slot_bytes = kStepRows * (unsigned)H2 * 4u;
    __amdgpu_buffer_rsrc_t part_rs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, (int)(2u * (unsigned)NB * slot_bytes), 0x00020000);

    // one-XCD launch: fixed-point sums, three arrays used in turn
    constexpr float kFixClamp = 524287.0f;
    constexpr double kFixBias = 1572864.0;   // 1.5 x 2^20: doubles in [2^20, 2^21) are 2^-32 apart
    constexpr unsigned long long kFixMant = (1ull << 52) - 1, kFixOne = 1ull << 58;
    __amdgpu_buffer_rsrc_t acc_rs = __builtin_amdgcn_make_buffer_rsrc(a.acc, 0, ONEX ? 3 * H2 * kStepRows * 8 : 0, 0x00020000);
    int acc_cur = 0;

    // The output layer's share of a step's update: w3 (this wave's rows of it, i.e. its own columns of the
    // second layer), b2 of those columns and b3.  Nothing before the sums of the next step's exchange are
    // in needs them, so they are applied while that exchange is under way (and once after the last step),
    // from h2, delta2 and delta3 of the step they belong to, which stay in LDS until then.
    //   dW3 = h2^T . delta3 (rows of w3 x classes, K = the 8 rows): register v of lane (class m16, g4) is
    //   row 16 t + 4 g4 + v, 16 contiguous bytes of W3T.
    bool out_pending = false;
    float out_scale = 0.0f;
    auto output_updates = [&](const int (&tt)[TPWMAX]) {
        float d3f[kStepMaxC / 16][kStepRows / 4], hf[TPWMAX][kStepRows / 4];
        f32x4 w3o[kStepMaxC / 16][TPWMAX];
#pragma unroll
        for (int st = 0; st < kStepRows / 4; ++st) {
#pragma unroll
            for (int h = 0; h < kStepMaxC / 16; ++h) d3f[h][st] = 16 * h < CW ? d3w[(4 * st + g4) * LDD3 + 16 * h + m16] : 0.0f;
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u) hf[u][st] = h2s[(4 * st + g4) * LD2 + 16 * tt[u] + m16];
        }
#pragma unroll
        for (int u = 0; u < TPWMAX; ++u)
#pragma unroll
            for (int h = 0; h < kStepMaxC / 16; ++h)
                w3o[h][u] = 16 * h < CW ? *reinterpret_cast<const f32x4*>(W3T + (16 * h + m16) * LD3 + 16 * tt[u] + 4 * g4) : f32x4{0.f, 0.f, 0.f, 0.f};
        float g2 = 0.0f, g3 = 0.0f;
        if (lane < 16 * ntw) {
#pragma unroll
            for (int r = 0; r < kStepRows; ++r) g2 += d2s[r * LD2 + 16 * t0 + lane];
        }
        if (wave == 0 && lane < C) {
#pragma unroll
            for (int r = 0; r < kStepRows; ++r) g3 += d3w[r * LDD3 + lane];
        }
        SZ_SCHED_FENCE();
        f32x4 acc3[kStepMaxC / 16][TPWMAX];
#pragma unroll
        for (int h = 0; h < kStepMaxC / 16; ++h)
            if (16 * h < CW) {
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u) acc3[h][u] = SZ_MFMA16(hf[u][0], d3f[h][0], (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u) acc3[h][u] = SZ_MFMA16(hf[u][1], d3f[h][1], acc3[h][u]);
            }
        SZ_SCHED_FENCE();
#pragma unroll
        for (int h = 0; h < kStepMaxC / 16; ++h)
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u)
                if (16 * h < CW && 16 * h + m16 < C && u < ntw)
                    *reinterpret_cast<f32x4*>(W3T + (16 * h + m16) * LD3 + 16 * tt[u] + 4 * g4) = w3o[h][u] - out_scale * acc3[h][u];
        if (lane < 16 * ntw) b2s[16 * t0 + lane] -= out_scale * g2;   // b2 of the own columns
        if (wave == 0 && lane < C) b3s[lane] -= out_scale * g3;
        out_pending = false;
    };

    bool healthy = true;
    if (a.force_abort && blk == NB - 1) {
        if (tid == 0) __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        healthy = false;
    }
    __syncthreads();
    if (PROF) pt0 = wall_clock64();
    for (uint64_t step = 0; step < n_steps && healthy; ++step) {
        const int cb_ = (int)(step & 1);                 // LDS buffer of this step's rows / live flags
        const float* __restrict__ xs = xsb + cb_ * kStepXs;
        const unsigned buf = (unsigned)(step & 1);       // exchange slot of this step
        // tiles of this wave, clamped for the loads; tile u counts when u < ntw
        int tt[TPWMAX];
#pragma unroll
        for (int u = 0; u < TPWMAX; ++u) tt[u] = min(t0 + min(u, TPW - 1), NT - 1);
        STEP_MARK(0);
        // 1. own hidden-1 units: rows x units, K = inputs; wave w < 2 takes units [16 w, 16 w + 16).
        // Result register v of lane (unit m16, g4) is row 4 g4 + v.  Every wave also fetches its w2
        // fragments of the next phase here, ahead of the barrier.  The contraction index of MFMA
        // step st, lane group g4 in that phase is unit 16 (st / 4) + 4 g4 + st % 4: conflict-free
        // reads of W2s, and 16-byte reads of the row-major h1.
        float w2f[TPWMAX][HS / 4];
#pragma unroll
        for (int u = 0; u < TPWMAX; ++u)
#pragma unroll
            for (int st = 0; st < HS / 4; ++st) w2f[u][st] = W2s[(16 * (st >> 2) + 4 * g4 + (st & 3)) * LD2 + 16 * tt[u] + m16];
        if (wave < HS / 16) {
            const int ib = wave * 16;
            float av[IN / 4], bv[IN / 4];
#pragma unroll
            for (int st = 0; st < IN / 4; ++st) {
                av[st] = xs[m16 * IN + 4 * st + g4];
                bv[st] = W1s[(4 * st + g4) * LDW1 + ib + m16];
            }
            const float b = b1s[ib + m16];
            SZ_SCHED_FENCE();
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st + 1 < IN / 4; st += 2) {
                acc = SZ_MFMA16(av[st], bv[st], acc);
                acc2 = SZ_MFMA16(av[st + 1], bv[st + 1], acc2);
            }
            acc = SZ_MFMA16(av[IN / 4 - 1], bv[IN / 4 - 1], acc);
            acc += acc2;
            if (g4 < 2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) h1s[(4 * g4 + v) * LDH1 + ib + m16] = fmaxf(acc[v] + b, 0.0f);
            }
        }
        __syncthreads();
        STEP_MARK(1);   // h1
        // 2. partial second layer, own column tiles
        {
            f32x4 a4[HS / 16];
#pragma unroll
            for (int h = 0; h < HS / 16; ++h) a4[h] = *reinterpret_cast<const f32x4*>(h1s + m16 * LDH1 + 16 * h + 4 * g4);
            SZ_SCHED_FENCE();
            f32x4 acc[TPWMAX] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int st = 0; st < HS / 4; ++st)
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u) acc[u] = SZ_MFMA16(a4[st >> 2][st & 3], w2f[u][st], acc[u]);
            SZ_SCHED_FENCE();
            // slot layout [column][8 rows]: lane (column m16, g4 < 2) stores rows 4 g4 .. 4 g4 + 3 as 16 bytes
            if (ONEX) {
                // One XCD: the workgroups add their partials into one array in the shared L2 with 64-bit
                // integer atomics.  A partial x becomes the fixed-point number round(x 2^32) + 2^51, read off
                // the mantissa of the double x + 1.5 2^20 (exact for |x| < 2^19, beyond which x is clamped);
                // integer sums do not depend on the order of arrival, so every workgroup reads the same
                // bits whatever the timing.  Each addend also carries a 1 at bit 58: the top six bits of an
                // element count the workgroups that have added to it (at most 32 addends below 2^52 never
                // carry into them), so an element says itself when it is complete and no flag is needed.
                unsigned long long* accb = a.acc + (size_t)acc_cur * H2 * kStepRows;
                // A partial that the fixed-point format cannot hold (2^19 or more in magnitude, infinite, NaN)
                // must not be clamped silently: the reference's arithmetic carries such values on
                // (lib.rs:1016-1059).  The workgroup reports it (abort word 2) and the host repeats the call
                // with the exchange in f32.
                bool unfit = false;
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u)
                    if (g4 < 2 && u < ntw) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) unfit = unfit || !(fabsf(acc[u][v]) < kFixClamp);
                    }
                if (__builtin_amdgcn_ballot_w64(unfit) != 0 && lane == 0) {
                    __hip_atomic_store(a.abort_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ctl[16] = 0;   // looked at after the next workgroup barrier
                }
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u)
                    if (g4 < 2 && u < ntw) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const double d = (double)fminf(fmaxf(acc[u][v], -kFixClamp), kFixClamp) + kFixBias;
                            // element (tile, row 4 g4 + v, column m16) at ((tile 4 + v) 2 + g4) 16 + m16: one instruction covers 256 contiguous bytes
                            __hip_atomic_fetch_add(accb + ((tt[u] * 4 + v) * 2 + g4) * 16 + m16,
                                                   (__builtin_bit_cast(unsigned long long, d) & kFixMant) + kFixOne,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
            } else {
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u)
                    if (g4 < 2 && u < ntw)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[u]), part_rs,
                                                               (buf * NB + blk) * slot_bytes + ((16 * tt[u] + m16) * kStepRows + 4 * g4) * 4u, 0, 16);
            }
        }
        if (out_pending) output_updates(tt);   // the previous step's w3, b3, b2 updates, behind this step's stores
        STEP_MARK(2);   // partial h2 computed and stored
        if (!ONEX) {
            // 3. publish: every storing wave drains its write-through stores, then one lane raises the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(&a.flags[blk], (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ctl[16] = 1;
            }
            STEP_MARK(3);   // drain + flag
            if (wave == 0) {   // wave 0 waits for every workgroup's flag
                if (lane < NB) {
                    const unsigned long long tw0 = wall_clock64();
                    while (__hip_atomic_load(&a.flags[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1)) {
                        __builtin_amdgcn_s_sleep(1);
                        const unsigned long long waited = wall_clock64() - tw0;
                        if (waited < 2000ull) continue;   // the abort word is looked at after 20 us of waiting
                        if (waited > kStepTimeoutTicks || __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ctl[16] = 0;
                            break;
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the slot loads below the poll
            __syncthreads();
            healthy = ctl[16] != 0;
            if (!healthy) break;
        }
        STEP_MARK(4);   // flags seen
        // 4. h2 = tanh(b2 + sum of the partials) for the own columns, workgroups in index order (the
        // same in every workgroup).  A slot is 2 H2 pieces of 16 bytes: piece 2 c + h holds rows
        // 4 h .. 4 h + 3 of column c; in pass p a lane takes piece (lane & 31) of tile 2 p + lane / 32.
        // All loads are sc1 and issued before the first add.
        if (ONEX) {
            // The sums of this step: a lane takes elements 2 lane and 2 lane + 1 of each of its tiles (128 a
            // tile): one row, two columns, with sc1 loads (they never hit the CU's L1).  Each wave reads its own
            // tiles until every element shows all NB arrivals.  Loads of an element that is still being added
            // to slow those additions down; the previous step's output-layer updates, which every wave has just
            // done, keep the first look 0.6 us behind the atomics, by when most of them have landed.
            auto gave_up = [&](unsigned long long since) -> bool {
                const unsigned long long waited = wall_clock64() - since;
                if (waited < 2000ull) return false;   // the abort word is looked at after 20 us of waiting
                const int seen = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (waited > kStepTimeoutTicks || seen) {
                    if (!seen) __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a reported reason stays)
                    ctl[16] = 0;   // looked at after the next workgroup barrier
                    return true;
                }
                return false;
            };
            const unsigned long long tw0 = wall_clock64();
            u32x4 q[TPWMAX];
            auto attempt = [&]() {
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u)
                    q[u] = __builtin_amdgcn_raw_buffer_load_b128(acc_rs, (unsigned)((acc_cur * H2 * kStepRows + tt[u] * 128 + 2 * lane) * 8), 0, 16);
            };
            auto complete = [&]() -> bool {
                bool full = true;
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u)
                    if (u < ntw) full = full && (q[u][1] >> 26) == (unsigned)NB && (q[u][3] >> 26) == (unsigned)NB;
                return __builtin_amdgcn_ballot_w64(!full) == 0;
            };
            attempt();
            // behind those loads: the next step's values and the indices of the one after
            SZ_SCHED_FENCE();
            load_val(cur_val, idx, xv);
            load_idx(cur_idx, idx);
            advance(cur_val);
            advance(cur_idx);
            SZ_SCHED_FENCE();
            while (!complete() && !gave_up(tw0)) {
                __builtin_amdgcn_s_sleep(1);
                attempt();
            }
            // Complete sums mean that every workgroup has begun this step, so none reads the array of the
            // previous step any more: this workgroup clears its share of it for the step after the next
            // (the stores have landed by the barrier at the end of this step, which wave 1 reaches after
            // waiting for them, and every workgroup's atomics of that later step follow such a barrier).
            if (wave == 1) {
                const int freed = acc_cur == 0 ? 2 : acc_cur - 1;
                for (int c = blk; c < H2 * kStepRows * 8 / 1024; c += NB)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, acc_rs, (unsigned)(freed * H2 * kStepRows * 8 + c * 1024 + lane * 16), 0, 0);
            }
            const long long bias_sum = (long long)NB << 51;   // every addend is offset by 2^51
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u)
                if (u < ntw) {
                    const int col = 16 * tt[u] + ((2 * lane) & 15), row = 4 * ((lane >> 3) & 1) + (lane >> 4);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned long long bits = (unsigned long long)q[u][2 * i] | ((unsigned long long)q[u][2 * i + 1] << 32);
                        const float sum = (float)((double)((long long)(bits & (kFixOne - 1)) - bias_sum) * (1.0 / 4294967296.0));
                        h2s[row * LD2 + col + i] = step_tanh(b2s[col + i] + sum);
                    }
                }
            acc_cur = acc_cur == 2 ? 0 : acc_cur + 1;
        } else
        {
            f32x4 s[TPWMAX / 2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            unsigned poff[TPWMAX / 2];
#pragma unroll
            for (int p = 0; p < TPWMAX / 2; ++p) poff[p] = (unsigned)(32 * tt[2 * p + (lane >> 5)] + (lane & 31)) * 16u;
            for (int b0 = 0; b0 < NB; b0 += 16) {
                u32x4 v[TPWMAX / 2][16];
#pragma unroll
                for (int p = 0; p < TPWMAX / 2; ++p)
#pragma unroll
                    for (int b = 0; b < 16; ++b)
                        if (2 * p < TPW)
                            v[p][b] = __builtin_amdgcn_raw_buffer_load_b128(part_rs, (buf * NB + min(b0 + b, NB - 1)) * slot_bytes + poff[p], 0, 16);
                if (b0 == 0) {
                    // while those loads are in flight: the next step's values and the indices of the one after
                    SZ_SCHED_FENCE();
                    load_val(cur_val, idx, xv);
                    load_idx(cur_idx, idx);
                    advance(cur_val);
                    advance(cur_idx);
                    SZ_SCHED_FENCE();
                }
#pragma unroll
                for (int p = 0; p < TPWMAX / 2; ++p)
#pragma unroll
                    for (int b = 0; b < 16; ++b)
                        if (2 * p < TPW && b0 + b < NB) s[p] += __builtin_bit_cast(f32x4, v[p][b]);
            }
#pragma unroll
            for (int p = 0; p < TPWMAX / 2; ++p)
                if (2 * p + (lane >> 5) < ntw) {
                    const int col = 16 * tt[2 * p + (lane >> 5)] + ((lane & 31) >> 1), r0 = 4 * (lane & 1);
                    const float bb = b2s[col];
#pragma unroll
                    for (int v = 0; v < 4; ++v) h2s[(r0 + v) * LD2 + col] = step_tanh(bb + s[p][v]);
                }
        }
        STEP_MARK(5);   // partials summed, tanh
        // 5. logits = h2 . w3, K split over the waves by their own columns.  The contraction index of
        // step q, lane group g4 within tile t is column 16 t + 4 g4 + q, so that a lane's four values
        // are 16 contiguous bytes of h2 (row-major) and of w3 (transposed).
        {
            f32x4 af[TPWMAX];
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u) af[u] = *reinterpret_cast<const f32x4*>(h2s + m16 * LD2 + 16 * tt[u] + 4 * g4);
            for (int cb = 0; cb < CW; cb += 16) {
                f32x4 bf[TPWMAX];
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u) bf[u] = *reinterpret_cast<const f32x4*>(W3T + (cb + m16) * LD3 + 16 * tt[u] + 4 * g4);
                SZ_SCHED_FENCE();
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u)
                    if (u < ntw) {
                        acc = SZ_MFMA16(af[u][0], bf[u][0], acc);
                        acc2 = SZ_MFMA16(af[u][1], bf[u][1], acc2);
                        acc = SZ_MFMA16(af[u][2], bf[u][2], acc);
                        acc2 = SZ_MFMA16(af[u][3], bf[u][3], acc2);
                    }
                SZ_SCHED_FENCE();
                acc += acc2;
                if (g4 < 2) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) red[((size_t)wave * kStepRows + 4 * g4 + v) * CW + cb + m16] = acc[v];
                }
            }
        }
        if (wave == 0 && lane < kStepRows) ctl[8 * (cb_ ^ 1) + lane] = 0;  // live flags of the next step: cleared before they are set in 8.
        // operands of 7. that do not depend on the softmax, fetched ahead of the barrier: w3 for
        // delta2 (class 16 (st / 4) + 4 g4 + st % 4 in step st: conflict-free), h2 for its
        // epilogue, and this wave's share of w2 for delta1 (16 bytes a lane, as in 5.)
        float w3f[TPWMAX][kStepMaxC / 4], hv[TPWMAX][4];
        f32x4 w2a[TPWMAX], w2b[TPWMAX];
#pragma unroll
        for (int u = 0; u < TPWMAX; ++u) {
#pragma unroll
            for (int st = 0; st < kStepMaxC / 4; ++st)
                w3f[u][st] = 4 * st < CW ? W3T[(16 * (st >> 2) + 4 * g4 + (st & 3)) * LD3 + 16 * tt[u] + m16] : 0.0f;
#pragma unroll
            for (int v = 0; v < 4; ++v) hv[u][v] = h2s[(4 * (g4 & 1) + v) * LD2 + 16 * tt[u] + m16];
            w2a[u] = *reinterpret_cast<const f32x4*>(W2s + m16 * LD2 + 16 * tt[u] + 4 * g4);
            w2b[u] = *reinterpret_cast<const f32x4*>(W2s + (16 + m16) * LD2 + 16 * tt[u] + 4 * g4);
        }
        __syncthreads();
        if (ONEX) {   // a wave that gave up waiting has said so before this barrier
            healthy = ctl[16] != 0;
            if (!healthy) break;
        }
        STEP_MARK(6);   // logits
        // 6. every wave for itself: softmax, loss with the pre-update weights, delta3 = o - t
        // (lib.rs:611-615, 1028) -> its own copy of delta3.  Lane = (row, class); a row's classes sit
        // in 16 (or 32) neighbouring lanes.
        int n_live = 0;
#pragma unroll
        for (int r = 0; r < kStepRows; ++r) n_live += ctl[8 * cb_ + r];
        const float scale = n_live > 0 ? a.lr / (float)n_live : 0.0f;  // lib.rs:1047
        for (int e0 = 0; e0 < kStepRows * CW; e0 += 64) {
            const int e = e0 + lane;
            const int r = e / CW, c = e - r * CW;
            const bool cls = c < C, alive = ctl[8 * cb_ + r] != 0;
            const float z = cls ? b3s[c] + ((red[(0 * kStepRows + r) * CW + c] + red[(1 * kStepRows + r) * CW + c]) +
                                            (red[(2 * kStepRows + r) * CW + c] + red[(3 * kStepRows + r) * CW + c]))
                                : -INFINITY;
            float mx = z;
            mx = fmaxf(mx, dpp16<0xB1>(mx)); mx = fmaxf(mx, dpp16<0x4E>(mx));
            mx = fmaxf(mx, dpp16<0x141>(mx)); mx = fmaxf(mx, dpp16<0x140>(mx));
            if (CW > 16) mx = fmaxf(mx, __shfl_xor(mx, 16));
            const float ex = cls ? __expf(z - mx) : 0.0f;
            float sum = ex;
            sum += dpp16<0xB1>(sum); sum += dpp16<0x4E>(sum); sum += dpp16<0x141>(sum); sum += dpp16<0x140>(sum);
            if (CW > 16) sum += __shfl_xor(sum, 16);
            const float o = ex * __builtin_amdgcn_rcpf(sum), t = tgt[c];
            if (wave == 0) {
                if (cls && alive && t != 0.0f) loss_acc -= (double)(t * logf(fmaxf(o, 1e-12f)));
                if (c == 0 && alive) live_acc += 1.0;
            }
            d3w[r * LDD3 + c] = (cls && alive) ? o - t : 0.0f;
        }
        STEP_MARK(7);   // softmax, delta3
        // 7. still every wave for itself, on its own columns.  delta2 = (delta3 . w3^T)(1 - h2^2),
        // lib.rs:1034: rows x columns of h2, K = classes.
        {
            f32x4 a4[kStepMaxC / 16];  // delta3[row m16][class 16 h + 4 g4 + 0..3]
#pragma unroll
            for (int h = 0; h < kStepMaxC / 16; ++h)
                a4[h] = 16 * h < CW ? *reinterpret_cast<const f32x4*>(d3w + m16 * LDD3 + 16 * h + 4 * g4) : f32x4{0.f, 0.f, 0.f, 0.f};
            SZ_SCHED_FENCE();
            f32x4 acc[TPWMAX] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int st = 0; st < kStepMaxC / 4; ++st)
                if (4 * st < CW) {
#pragma unroll
                    for (int u = 0; u < TPWMAX; ++u) acc[u] = SZ_MFMA16(a4[st >> 2][st & 3], w3f[u][st], acc[u]);
                }
            SZ_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u)
                if (g4 < 2 && u < ntw) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) d2s[(4 * g4 + v) * LD2 + 16 * tt[u] + m16] = acc[u][v] * (1.0f - hv[u][v] * hv[u][v]);
                }
        }
        STEP_MARK(8);   // delta2
        if (n_live > 0) {
            // Reads of everything the rest of 7. needs, then its matrix instructions back to back:
            //   delta1 of the own units, delta2 . w2[own, :]^T with the old w2: this wave's share of K
            //     (its own columns), two tiles of 16 units;
            //   dW2^T = delta2^T . h1 for the own columns (columns x units, K = the 8 rows): register v
            //     of lane (unit m16, g4) is column 16 t + 4 g4 + v, 16 contiguous bytes of a W2s row.
            static_assert(kStepHS == 32, "two tiles of 16 units");
            f32x4 dv[TPWMAX];
            float h1f[2][kStepRows / 4], d2f[TPWMAX][kStepRows / 4];
            f32x4 w2o[TPWMAX][2];
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u) dv[u] = *reinterpret_cast<const f32x4*>(d2s + m16 * LD2 + 16 * tt[u] + 4 * g4);
#pragma unroll
            for (int st = 0; st < kStepRows / 4; ++st) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) h1f[ib][st] = h1s[(4 * st + g4) * LDH1 + 16 * ib + m16];
#pragma unroll
                for (int u = 0; u < TPWMAX; ++u) d2f[u][st] = d2s[(4 * st + g4) * LD2 + 16 * tt[u] + m16];
            }
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) w2o[u][ib] = *reinterpret_cast<const f32x4*>(W2s + (16 * ib + m16) * LD2 + 16 * tt[u] + 4 * g4);
            }
            SZ_SCHED_FENCE();
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u)
                if (u < ntw) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc0 = SZ_MFMA16(dv[u][q], w2a[u][q], acc0);
                        acc1 = SZ_MFMA16(dv[u][q], w2b[u][q], acc1);
                    }
                }
            f32x4 acc2[TPWMAX][2];
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc2[u][ib] = SZ_MFMA16(d2f[u][0], h1f[ib][0], (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc2[u][ib] = SZ_MFMA16(d2f[u][1], h1f[ib][1], acc2[u][ib]);
            SZ_SCHED_FENCE();
            if (g4 < 2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    red2[((size_t)wave * kStepRows + 4 * g4 + v) * HS + m16] = acc0[v];
                    red2[((size_t)wave * kStepRows + 4 * g4 + v) * HS + 16 + m16] = acc1[v];
                }
            }
#pragma unroll
            for (int u = 0; u < TPWMAX; ++u)
                if (u < ntw) {
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
                        *reinterpret_cast<f32x4*>(W2s + (16 * ib + m16) * LD2 + 16 * tt[u] + 4 * g4) = w2o[u][ib] - scale * acc2[u][ib];
                }
            // the updates of w3, b3 and b2 wait until the next step's exchange is under way (output_updates)
            out_pending = true;
            out_scale = scale;
            __syncthreads();
            STEP_MARK(11);  // w2 update
            // 8. delta1[row 4 st + g4][unit 16 ib + m16], the operand of dW1: every lane sums the four
            // waves' parts of its elements (the same order everywhere) and applies the ReLU derivative
            // (lib.rs:1039-1040).  Own columns of w1 -= scale * x^T . delta1 (inputs x units, K = the 8
            // rows); wave w takes inputs [16 w, 16 w + 16) (inputs past 59 read the next row's start:
            // finite, dropped).  b1 -= scale * column sums of delta1.
            {
                float bf[2][kStepRows / 4], wv[2][4];
                float p[2][kStepRows / 4][4], hm[2][kStepRows / 4];
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int st = 0; st < kStepRows / 4; ++st) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) p[ib][st][w] = red2[((size_t)w * kStepRows + 4 * st + g4) * HS + 16 * ib + m16];
                        hm[ib][st] = h1s[(4 * st + g4) * LDH1 + 16 * ib + m16];
                    }
                const int j = 16 * wave + m16;
                const float a0 = xs[g4 * IN + j], a1 = xs[(4 + g4) * IN + j];
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int v = 0; v < 4; ++v) wv[ib][v] = W1s[min(16 * wave + 4 * g4 + v, IN - 1) * LDW1 + 16 * ib + m16];
                SZ_SCHED_FENCE();
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int st = 0; st < kStepRows / 4; ++st) {
                        const float sum = (p[ib][st][0] + p[ib][st][1]) + (p[ib][st][2] + p[ib][st][3]);
                        bf[ib][st] = sum * (hm[ib][st] > 0.0f ? 1.0f : 0.0f);  // a product, as lib.rs:1039-1040: 0 x NaN stays NaN
                    }
                f32x4 acc[2];
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc[ib] = SZ_MFMA16(a0, bf[ib][0], (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc[ib] = SZ_MFMA16(a1, bf[ib][1], acc[ib]);
                if (wave == 0) {   // b1: rows g4 and 4 + g4 of a unit sit in this lane, the other rows 16 lanes apart
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        float g = bf[ib][0] + bf[ib][1];
                        g += __shfl_xor(g, 16);
                        g += __shfl_xor(g, 32);
                        if (g4 == 0) b1s[16 * ib + m16] -= scale * g;
                    }
                }
                SZ_SCHED_FENCE();
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int jj = 16 * wave + 4 * g4 + v;
                        if (jj < IN) W1s[jj * LDW1 + 16 * ib + m16] = wv[ib][v] - scale * acc[ib][v];
                    }
            }
        }
        // the next step's rows (fetched during the exchange) -> the other LDS buffer
        stage_rows(cb_ ^ 1, xv);
        STEP_MARK(12);  // delta1 combine, w1, b1 updates, next rows staged
        if (ONEX && wave == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (PROF && blk == 0 && tid == 0)
        for (int i = 0; i < kStepProfSlots; ++i) a.prof[i] = pacc[i];
#undef STEP_MARK
    if (!healthy) return;  // the host discards the shadow vector
    if (out_pending) {   // the last step's w3, b3, b2 updates
        int tt[TPWMAX];
#pragma unroll
        for (int u = 0; u < TPWMAX; ++u) tt[u] = min(t0 + min(u, TPW - 1), NT - 1);
        output_updates(tt);
    }
    __syncthreads();
    // the owned parameters go to the shadow vector; workgroup 0 also writes the replicated ones and the totals
    float* Pw = a.params_out;
    for (int e = tid; e < IN * HS; e += kStepThreads) { const int j = e / HS, i = e - j * HS; Pw[(size_t)j * a.h1 + i0 + i] = W1s[j * LDW1 + i]; }
    for (int e = tid; e < HS; e += kStepThreads) Pw[off_b1 + i0 + e] = b1s[e];
    for (int e = tid; e < HS * H2; e += kStepThreads) { const int i = e / H2, k = e - i * H2; Pw[off_w2 + (size_t)(i0 + i) * H2 + k] = W2s[i * LD2 + k]; }
    if (blk == 0) {
        for (int e = tid; e < H2; e += kStepThreads) Pw[off_b2 + e] = b2s[e];
        for (int e = tid; e < H2 * C; e += kStepThreads) { const int k = e / C, c = e - k * C; Pw[off_w3 + e] = W3T[c * LD3 + k]; }
        for (int e = tid; e < C; e += kStepThreads) Pw[off_b3 + e] = b3s[e];
        // per-lane sums of wave 0 -> one add per lane that saw a loss term
        if (loss_acc != 0.0) atomicAdd(&a.totals[0], loss_acc);
        if (live_acc != 0.0) atomicAdd(&a.totals[1], live_acc);
    }
}

int padded_classes(uint32_t out) { return ((int)out + 15) / 16 * 16; }
// the float arrays, then 17 ints of per-step state
size_t step_lds_bytes(int h2, int cw) { return (size_t)StepLds(h2, cw).total * sizeof(float) + 128; }

typedef void (*StepFn)(StepArgs);
template <bool PROF, bool ONEX>
StepFn pick_step_kernel(int h2, int cw)
{
    if (h2 == 256 && cw == 16) return step_kernel<256, 16, PROF, ONEX>;
    if (h2 == 256 && cw == 32) return step_kernel<256, 32, PROF, ONEX>;
    if (h2 == 128 && cw == 16) return step_kernel<128, 16, PROF, ONEX>;
    return step_kernel<0, 0, PROF, ONEX>;
}

}  // namespace

bool step_kernel_applies(const Net* n, uint32_t batch_size, uint64_t n_steps)
{
    return n_steps < 0x7fffffffull && batch_size >= 1 && batch_size <= (uint32_t)kStepRows && n->in == (uint32_t)kStepIn &&
           n->h1 % kStepHS == 0 && n->h1 / kStepHS >= 1 && n->h1 / kStepHS <= (uint32_t)kStepMaxBlocks && n->h2 >= 16 &&
           n->h2 % 16 == 0 && n->h2 <= 256 && n->out >= 1 && n->out <= (uint32_t)kStepMaxC && kStepHS * kStepRows == kStepThreads;
}

int net_pretrain_steps(Net* n, const float* d_win, uint32_t n_win, const uint32_t* d_perm, const uint8_t* d_keep,
                       float dropout, uint64_t seed, uint64_t key_row0, const float* d_target, uint32_t epochs,
                       uint32_t batch_size, float lr, hipStream_t s)
{
    // STREAMZ_STEP_FORCE_ABORT (test hook): 1 = a workgroup reports a time-out, the caller falls back;
    // 2 = the same, reported as an error instead
    static const int env_mode = [] { const char* v = std::getenv("STREAMZ_STEP_FORCE_ABORT"); return v ? std::atoi(v) : 0; }();
    // STREAMZ_STEP_PROF=1: the instrumented instantiation, per-phase times of workgroup 0 to stderr
    static const bool prof_on = [] { const char* v = std::getenv("STREAMZ_STEP_PROF"); return v && v[0] == '1'; }();
    // STREAMZ_STEP_ONE_XCD=0: never place the workgroups on one XCD
    const bool one_xcd_on = [] { const char* v = std::getenv("STREAMZ_STEP_ONE_XCD"); return !(v && v[0] == '0'); }();
    const int nb = (int)(n->h1 / kStepHS);
    const int cw = padded_classes(n->out);
    const size_t lds = step_lds_bytes((int)n->h2, cw);
    int dev = 0, cus = 0;
    SZ_HIP(hipGetDevice(&dev));
    SZ_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));

    // one allocation: [partial slots | shadow parameters | flags + abort word (zeroed)]
    const size_t part_bytes = (2ull * nb * kStepRows * n->h2 * sizeof(float) + 255) / 256 * 256;
    const size_t shadow_bytes = (n->param_count() * sizeof(float) + 255) / 256 * 256;
    // 64 flags, at byte 256 the abort word and the ticket counter, at byte 384 the call's {loss, live rows},
    // from byte 512 the three fixed-point sum arrays of the one-XCD launch
    const size_t flag_bytes = 512 + 3ull * n->h2 * kStepRows * sizeof(unsigned long long);
    char* base = nullptr;
    SZ_HIP(hipMalloc((void**)&base, part_bytes + shadow_bytes + flag_bytes));
    struct Free { char* p; ~Free() { (void)hipFree(p); } } guard{base};
    StepArgs a{};
    a.params = n->d_params;
    a.params_out = reinterpret_cast<float*>(base + part_bytes);
    a.h1 = (int)n->h1; a.h2 = (int)n->h2; a.out = (int)n->out;
    a.win = d_win; a.n_win = n_win; a.perm = d_perm; a.keep = d_keep; a.dropout = dropout; a.seed = seed; a.key_row0 = key_row0;
    a.target = d_target; a.epochs = epochs; a.bs = batch_size; a.lr = lr;
    a.part = reinterpret_cast<float*>(base);
    a.flags = reinterpret_cast<unsigned*>(base + part_bytes + shadow_bytes);
    a.abort_flag = reinterpret_cast<int*>(a.flags + kStepMaxBlocks);
    a.ticket = reinterpret_cast<unsigned*>(a.abort_flag + 1);
    a.nb = nb;
    a.acc = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.flags) + 512);
    a.force_abort = env_mode != 0 ? 1 : 0;
    a.totals = reinterpret_cast<double*>(reinterpret_cast<char*>(a.flags) + 384);
    unsigned long long* d_prof = nullptr;
    if (prof_on) { SZ_HIP(hipMalloc((void**)&d_prof, kStepProfSlots * sizeof(unsigned long long))); a.prof = d_prof; }
    struct FreeProf { unsigned long long* p; ~FreeProf() { if (p) (void)hipFree(p); } } guard2{d_prof};
    struct { int abort; int pad[31]; double totals[2]; } tail;  // bytes 256 .. 400 of the flag block
    static_assert(sizeof(tail) == 144, "abort word at 256, totals at 384");

    // First with all workgroups on one XCD (they must fit its CUs, one each), then spread over the chip.
    // A launch that did not run every step leaves the parameters alone, so the next one starts afresh.
    bool ran = false;
    // a process whose one-XCD launch once failed to assemble does not try (and wait for) it again
    static std::atomic<bool> one_xcd_failed{false};
    for (int onex = (one_xcd_on && !one_xcd_failed.load() && nb <= cus / 8) ? 1 : 0; onex >= 0 && !ran; --onex) {
        StepFn fn = prof_on ? (onex ? pick_step_kernel<true, true>((int)n->h2, cw) : pick_step_kernel<true, false>((int)n->h2, cw))
                            : (onex ? pick_step_kernel<false, true>((int)n->h2, cw) : pick_step_kernel<false, false>((int)n->h2, cw));
        SZ_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // every workgroup must be resident at once: the exchange waits for all of them
        int per_cu = 0;
        SZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, kStepThreads, lds));
        if ((long)per_cu * cus < nb) return SZH_STEP_NOT_RUN;
        SZ_HIP(hipMemsetAsync(a.flags, 0, flag_bytes, s));
        {
            ProfScope prof("steps", s);
            hipLaunchKernelGGL(fn, dim3(onex ? 8 * nb : nb), dim3(kStepThreads), lds, s, a);
        }
        SZ_HIP(hipGetLastError());
        SZ_HIP(hipMemcpyAsync(&tail, a.abort_flag, sizeof(tail), hipMemcpyDeviceToHost, s));
        SZ_HIP(hipStreamSynchronize(s));
        // a one-XCD launch also has to have handed out every role (no workgroup on XCD 0 means no abort either)
        ran = tail.abort == 0 && (!onex || tail.pad[0] >= nb);
        if (onex && std::getenv("STREAMZ_STEP_DEBUG")) {
            unsigned f[kStepMaxBlocks + 2] = {};
            (void)hipMemcpy(f, a.flags, sizeof(f), hipMemcpyDeviceToHost);
            std::fprintf(stderr, "one-XCD launch: abort %d, tickets %u, arrivals per XCD:", tail.abort, f[kStepMaxBlocks + 1]);
            for (int i = 0; i < 16; ++i) std::fprintf(stderr, " %u", f[40 + i]);
            std::fprintf(stderr, "\n");
        }
        if (!ran && a.force_abort) break;   // the test hook asks for the failure itself
        // abort word 2: a partial sum did not fit the one-XCD exchange's fixed-point format; the launch over
        // the chip, whose exchange is f32, takes this call (and says nothing about later calls)
        if (!ran && onex && tail.abort != 2) one_xcd_failed.store(true);
    }
    if (d_prof && ran) {
        unsigned long long h[kStepProfSlots] = {};
        if (hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            const uint64_t n_steps = (uint64_t)epochs * ((n_win + batch_size - 1) / batch_size);
            std::fprintf(stderr, "step_kernel phases, ns per step over %llu steps:", (unsigned long long)n_steps);
            for (int i = 0; i < kStepProfSlots; ++i) std::fprintf(stderr, " [%d] %.0f", i, 10.0 * (double)h[i] / (double)(n_steps ? n_steps : 1));
            std::fprintf(stderr, "\n");
        }
    }
    if (!ran) {
        // some workgroup gave up waiting: nothing was adopted, the parameters are as before the call
        if (env_mode == 2) { set_error("training step kernel: a workgroup timed out waiting for the others; parameters unchanged"); return SZH_ENODEVICE; }
        return SZH_STEP_NOT_RUN;
    }
    // adopt the shadow parameters and add the call's loss / live-row sums to the running totals
    SZ_HIP(hipMemcpyAsync(n->d_params, a.params_out, n->param_count() * sizeof(float), hipMemcpyDeviceToDevice, s));
    double run[2];
    SZ_HIP(hipMemcpyAsync(run, n->d_stats + 2, sizeof(run), hipMemcpyDeviceToHost, s));
    SZ_HIP(hipStreamSynchronize(s));
    run[0] += tail.totals[0];
    run[1] += tail.totals[1];
    SZ_HIP(hipMemcpyAsync(n->d_stats + 2, run, sizeof(run), hipMemcpyHostToDevice, s));
    SZ_HIP(hipStreamSynchronize(s));
    return SZH_OK;
}

}  // namespace sz

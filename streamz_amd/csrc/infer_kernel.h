// identify_speaker_list's loop body (lib.rs:1391-1401: forward of lib.rs:880-891, arg-max, threshold,
// vote) for a tile of rows as ONE kernel: the feature rows are the only thing read per window and the
// votes the only thing written; h1, h2 and the logits never leave the compute unit.
//
// Every layer is computed transposed, out^T = W^T . act^T, on v_mfma_f32_32x32x2_f32: the weights are
// the A operand (a lane holds one weight row), the activations of 32 feature rows the B operand (a lane
// holds one feature row).  A result register r of lane (m, half) then is out[row m][unit u(r)],
// u(r) = (r & 3) + 8 (r >> 2) + 4 half -- which is exactly what the next layer's B operand wants: lane
// (m, half) supplies act[m][k] for the k of its half of each step.  So a wave's layer-2 output feeds layer
// 3 straight from its accumulators, and only h1, which every wave needs all of, passes through LDS.
//
// A workgroup of 8 waves owns MT blocks of 32 rows:
//   0. the tile's rows come from an LDS buffer that was filled during the previous tile (one coalesced copy of
//      the tile's contiguous rows, requested before its first layer, stored after it), so that no wave waits
//      for HBM between tiles;
//   1. h1^T tiles (32 units x 32 rows, K = 64: the inputs, a one for the bias, zeros) are dealt to the
//      waves; ReLU; written to LDS as h1[row][unit] (16-byte pieces, conflict-free);
//   2. wave w computes h2^T for units 32 w .. 32 w + 31 over all of h1: its rows of w2^T come straight
//      from global memory (L2), 16 bytes per lane and four steps, two 32-unit trips in flight, from a copy
//      of the weights laid out in fetch order (a wave's load reads 1 KB of consecutive memory; with the
//      matrix as it lies, every lane of a load would touch a cache line of its own); one weight fragment
//      multiplies all MT row blocks, whose activations are read from LDS one step ahead;
//   3. partial logits^T over the wave's 32 units of h2 (w3 as the A operand), summed over the waves
//      through LDS; softmax over a row's classes by 16 lanes (DPP), the last maximum wins, thresholded
//      vote into an LDS histogram; per-class probability sums in f64.
// On this part f32 matrix instructions and vector instructions take turns on a SIMD even when they belong to
// different waves, so every vector instruction below is paid in matrix time: what can be a mask, an
// instruction modifier or a table layout is not a branch, a copy or an address computation.
// With MT = 1 a tile of 8192 rows is 256 workgroups, one per compute unit; each reads all weights once
// (672 KB from L2 at 60-512-256-16), which is what 32 rows per compute unit cost; MT = 2 halves that for
// calls with enough rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_kernel.h"

namespace sz {
namespace infer {

using gemm::f32x16;
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512, kWaves = 8;
constexpr int kMaxClasses = 64;

struct Args {
    const float* x;    // [rows][in] feature rows, 16-byte aligned
    uint64_t rows;
    int in;            // a multiple of 4, <= 60
    // The weights in the order the waves fetch them (infer_tables_kernel): every 16-byte load of a wave
    // reads 1 KB of consecutive memory, lane (u, half) = 32 half + u taking its own four values.
    // the three tables lie back to back at `tab`:
    //   w1f [H1/32][8 q][64 lanes][4]: w1p[32 kt + u][32 half + 4 q + i], w1p = [w1^T | b1 | 0] ([H1][64])
    //   w2f [H2/32][H1/32][4 q][64 lanes][4]: w2^T[32 jt + u][32 ko + 8 q + 4 half + i]
    //   w3f [2 ct][H2/32][4 q][64 lanes][4]: w3^T[32 ct + u][32 jt + 8 q + 4 half + i], zero behind class C
    const float* tab;
    const float* b2;
    const float* b3;
    int H1, H2, C;     // H1 a multiple of 128, H2 a multiple of 32 and <= 256, C <= 64
    float threshold;
    unsigned long long* counts;  // [C] votes, added to (optional)
    double* psums;               // [C] probability sums, added to (optional)
    int tail_off;                // floats from the LDS base to the histogram
};

// floats of LDS the activations / partial logits need, and the whole dynamic allocation
__host__ __device__ constexpr int part_ld(int nct) { return 32 * nct + 4; }
inline size_t act_floats(int mt, int h1, int nct)
{
    const size_t h = (size_t)mt * 32 * (h1 + 4), p = (size_t)kWaves * mt * 32 * part_ld(nct);
    return h > p ? h : p;
}
// ... plus the vote histogram, the probability sums and the staging buffer of the next tile's rows
inline size_t lds_bytes(int mt, int h1, int nct, int in)
{
    return act_floats(mt, h1, nct) * sizeof(float) + kMaxClasses * (sizeof(unsigned int) + sizeof(double)) + (size_t)mt * 32 * in * sizeof(float);
}

inline size_t table_floats(int h1, int h2) { return (size_t)h1 * 64 + (size_t)h2 * h1 + (size_t)kMaxClasses * h2; }

// the three tables, back to back at `out` (w1f, w2f, w3f); one thread per float
__global__ __launch_bounds__(256) void infer_tables_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ w3, int in, int H1,
                                                           int H2, int C, float* __restrict__ out)
{
    const size_t n1 = (size_t)H1 * 64, n2 = (size_t)H2 * H1, n3 = (size_t)kMaxClasses * H2;
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n1 + n2 + n3) return;
    const int i = (int)(e & 3), lane = (int)((e >> 2) & 63), u = lane & 31, half = lane >> 5;
    float v;
    if (e < n1) {
        const int q = (int)((e >> 8) & 7), kt = (int)(e >> 11);
        const int unit = 32 * kt + u, k = 32 * half + 4 * q + i;
        v = k < in ? w1[(size_t)k * H1 + unit] : (k == in ? b1[unit] : 0.0f);
    } else if (e < n1 + n2) {
        const size_t f = e - n1;
        const int KT = H1 >> 5;
        const int q = (int)((f >> 8) & 3);
        const size_t trip = f >> 10;  // jt * KT + ko
        const int ko = (int)(trip % KT), jt = (int)(trip / KT);
        v = w2[(size_t)(32 * ko + 8 * q + 4 * half + i) * H2 + 32 * jt + u];
    } else {
        const size_t f = e - n1 - n2;
        const int JT = H2 >> 5;
        const int q = (int)((f >> 8) & 3);
        const size_t blk = f >> 10;  // ct * JT + jt
        const int jt = (int)(blk % JT), ct = (int)(blk / JT);
        const int c = 32 * ct + u;
        v = c < C ? w3[(size_t)(32 * jt + 8 * q + 4 * half + i) * C + c] : 0.0f;
    }
    out[e] = v;
}

template <int CTRL>
__device__ __forceinline__ float dpp16(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float x)
{
    x = fmaxf(x, dpp16<0xB1>(x));
    x = fmaxf(x, dpp16<0x4E>(x));
    x = fmaxf(x, dpp16<0x141>(x));
    return fmaxf(x, dpp16<0x140>(x));
}
__device__ __forceinline__ float row16_sum(float x)
{
    x += dpp16<0xB1>(x);
    x += dpp16<0x4E>(x);
    x += dpp16<0x141>(x);
    return x + dpp16<0x140>(x);
}

// 16 bytes of the weight tables: a buffer load takes the tables' descriptor and a wave-uniform byte
// offset from scalar registers, the lane's own offset as 32 bits and the piece's as an immediate, where a
// plain load would hold a 64-bit address per load in vector registers
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v load_frag(__amdgpu_buffer_rsrc_t tab, unsigned uniform_bytes, unsigned lane_bytes)
{
    return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(tab, lane_bytes, uniform_bytes, 0));
}

#define SZ_MFMA32(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)
// tools/infer_bench.hip builds timing-only variants with parts of a tile left out (bit 1: the first
// layer's matrix instructions, 2: the second layer's, 4: output layer, softmax and vote, 8: the final global
// atomics, 16: every tile reads the first rows (no HBM latency), 32: no tanh, 64: no softmax arithmetic)
#ifndef SZ_INFER_ABLATE
#define SZ_INFER_ABLATE 0
#endif

// MT: blocks of 32 rows per workgroup; NCT: blocks of 32 classes
template <int MT, int NCT>
__global__ __launch_bounds__(kThreads) void infer_vote_kernel(Args a)
{
    extern __shared__ __attribute__((aligned(16))) float ilds[];
    const int H1 = a.H1, LD = H1 + 4;
    constexpr int PLD = part_ld(NCT);
    float* h1s = ilds;   // [MT * 32][LD]; later the partial logits [wave][MT][32 rows][PLD]
    unsigned int* hist = reinterpret_cast<unsigned int*>(ilds + a.tail_off);  // [64]
    double* dsum = reinterpret_cast<double*>(hist + kMaxClasses);             // [64] (8-byte aligned: tail_off is even)
    float* xs = reinterpret_cast<float*>(dsum + kMaxClasses);                 // [MT * 32][in]: the rows of the tile to come
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    if (tid < kMaxClasses) { hist[tid] = 0; dsum[tid] = 0.0; }
    const int KT = H1 >> 5, JT = a.H2 >> 5;
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.tab), 0, (int)(4u * (64u * H1 + (unsigned)a.H2 * H1 + 64u * a.H2)), 0x00020000);
    const unsigned w3_off = 4u * (64u * H1 + (unsigned)a.H2 * H1);
    // reduction role: 16 lanes per row, lane cl takes classes cl, cl + 16, ...
    const int rm = tid >> 4, cl = tid & 15;
    double pacc[2 * NCT];
#pragma unroll
    for (int u = 0; u < 2 * NCT; ++u) pacc[u] = 0.0;
    float b3v[2 * NCT];
#pragma unroll
    for (int u = 0; u < 2 * NCT; ++u) b3v[u] = cl + 16 * u < a.C ? a.b3[cl + 16 * u] : 0.0f;

    const uint64_t n_tiles = (a.rows + 32 * MT - 1) / (32 * MT);
    // A tile's rows are contiguous in memory, 8 * MT * in pieces of 16 bytes; piece e of the tile goes with thread e % 512.
    // Pieces behind the last row repeat the last piece (their rows' results are not counted).
    const uint32_t tile_pieces = 8u * MT * (uint32_t)a.in;
    const uint64_t last_piece = a.rows * (uint64_t)a.in / 4 - 1;
    f32x4v xn[MT];
    auto fetch_rows = [&](uint64_t t) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const uint32_t e = (uint32_t)tid + kThreads * j;
            uint64_t g = ((SZ_INFER_ABLATE & 16) ? 0 : t) * tile_pieces + e;
            g = g < last_piece ? g : last_piece;
            if (e < tile_pieces) xn[j] = reinterpret_cast<const f32x4v*>(a.x)[g];
        }
    };
    auto stage_rows = [&]() {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const uint32_t e = (uint32_t)tid + kThreads * j;
            if (e < tile_pieces) reinterpret_cast<f32x4v*>(xs)[e] = xn[j];
        }
    };
    if (blockIdx.x < n_tiles) { fetch_rows(blockIdx.x); stage_rows(); }
    __syncthreads();

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t row0 = tile * (32 * MT);
        // The tile's rows from the staging buffer; the weights of the first layer and of the second layer's
        // first trips are requested up front.
        f32x4v xr[MT][8];  // lane (m, half): x[m][32 half + 4 q + 0..3]
        // (in is a multiple of 4: a group of four lies inside the row, or starts with the bias's one, or is padding)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float* xp = xs + (32 * mt + l31) * a.in;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c0 = 32 * half + 4 * q;
                xr[mt][q] = *reinterpret_cast<const f32x4v*>(xp + (c0 < a.in ? c0 : a.in - 4));
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c0 = 32 * half + 4 * q;
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[mt][q][i] = c0 < a.in ? xr[mt][q][i] : 0.0f;
                xr[mt][q][0] = c0 == a.in ? 1.0f : xr[mt][q][0];
            }
        const uint64_t next_tile = tile + gridDim.x;
        if (next_tile < n_tiles) fetch_rows(next_tile);
        const unsigned boff = (unsigned)lane * 16u;
        // the wave's tiles of the first layer's weights, two in flight
        // (A wave without a tile or a unit block of its own requests the last one's weights and does not use
        // them: requests under a condition would make every one of these registers a value carried round the
        // tile loop, kept alive and copied.)
        f32x4v wa0[8], wa1[8];
        {
            const int ta = wave < KT ? wave : KT - 1, tb = wave + kWaves < KT ? wave + kWaves : KT - 1;
#pragma unroll
            for (int q = 0; q < 8; ++q) wa0[q] = load_frag(tab, 8192u * ta, boff + 1024u * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) wa1[q] = load_frag(tab, 8192u * tb, boff + 1024u * q);
        }
        const int jw = wave < JT ? wave : JT - 1;  // the wave's block of 32 second-layer units
        const unsigned w2_off = 256u * H1 + 4096u * (unsigned)(jw * KT);  // bytes to the wave's first trip of w2f; 4096 per trip
        constexpr int D = 2;  // trips of the second layer's weights in flight (a trip is 16 MT matrix instructions)
        f32x4v wb[D][4];
        constexpr int EARLY = MT == 1 ? D : 0;  // ... of which these are requested ahead of the first layer
#pragma unroll
        for (int d = 0; d < EARLY; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) wb[d][q] = load_frag(tab, w2_off + 4096u * d, boff + 1024u * q);
        // ---- 1. first layer: h1 = relu([x | 1 | 0] . w1p^T)
        auto layer1_tile = [&](int t1, const f32x4v (&w)[8]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (SZ_INFER_ABLATE & 1) acc[0] += w[q][i] * xr[mt][q][i];
                        else acc = SZ_MFMA32(w[q][i], xr[mt][q][i], acc);
                    }
                float* dst = h1s + (size_t)(32 * mt + l31) * LD + 32 * t1 + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4v o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = gemm::relu_max(acc[4 * q + i]);
                    *reinterpret_cast<f32x4v*>(dst + 8 * q) = o;
                }
            }
        };
        for (int t1 = wave; t1 < KT; t1 += 2 * kWaves) {
            layer1_tile(t1, wa0);
            if (t1 + 2 * kWaves < KT) {
#pragma unroll
                for (int q = 0; q < 8; ++q) wa0[q] = load_frag(tab, 8192u * (t1 + 2 * kWaves), boff + 1024u * q);
            }
            if (t1 + kWaves < KT) {
                layer1_tile(t1 + kWaves, wa1);
                if (t1 + 3 * kWaves < KT) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) wa1[q] = load_frag(tab, 8192u * (t1 + 3 * kWaves), boff + 1024u * q);
                }
            }
        }
#pragma unroll
        for (int d = EARLY; d < D; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) wb[d][q] = load_frag(tab, w2_off + 4096u * d, boff + 1024u * q);
        __syncthreads();
        // Every wave has taken this tile's rows out of the staging buffer: the next tile's go in.  They were
        // requested before the first layer (loads complete in order, so a request made here would be waited
        // for by the second layer's first weight fragments).
        if (next_tile < n_tiles) stage_rows();
        // ---- 2. second layer: the wave's 32 units of h2, over all of h1
        f32x16 h2[MT];
        f32x4v w3v[NCT][4];  // the output layer's weights for this wave's units
        if (wave < JT) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h2[mt][r] = 0.0f;
            const float* hp = h1s + (size_t)l31 * LD + 4 * half;
            // The activations of a (trip, row block) step are read from LDS one step ahead, while the matrix
            // instructions of the step before them are being issued: hv[s & 1] belongs to step s.
            f32x4v hv[2][4];
            auto read_h = [&](f32x4v (&dst)[4], int trip, int mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4v*>(hp + (size_t)(32 * mt) * LD + 32 * trip + 8 * q);
            };
            auto multiply = [&](const f32x4v (&w)[4], const f32x4v (&h)[4], f32x16& acc) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (SZ_INFER_ABLATE & 2) acc[0] += w[q][i] * h[q][i];
                        else acc = SZ_MFMA32(w[q][i], h[q][i], acc);
                    }
            };
            read_h(hv[0], 0, 0);
            for (int ko = 0; ko < KT; ko += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int trip = ko + d;
                    if (MT == 1) {
                        // (D is even, so a trip's buffer is d & 1)
                        if (trip + 1 < KT) read_h(hv[(d + 1) & 1], trip + 1, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        multiply(wb[d], hv[d & 1], h2[0]);
                    } else {
                        read_h(hv[1], trip, 1);
                        __builtin_amdgcn_sched_barrier(0);
                        multiply(wb[d], hv[0], h2[0]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (trip + 1 < KT) read_h(hv[0], trip + 1, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        multiply(wb[d], hv[1], h2[MT - 1]);
                    }
                    // the trip D further on, into the registers this one has just been multiplied from (asked
                    // for any earlier, the compiler copies the fragment it is still using)
                    if (trip + D < KT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) wb[d][q] = load_frag(tab, w2_off + 4096u * (trip + D), boff + 1024u * q);
                    }
                    // (keeps the compiler from moving the LDS reads of all trips to the top, which costs more
                    // registers than there are)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // (requested here, where the second layer's weight registers have become free; the tanh covers the wait)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    w3v[ct][q] = load_frag(tab, w3_off + 4096u * (unsigned)(ct * JT + jw), boff + 1024u * q);
            // bias and tanh (lib.rs:883); register 4 q + i is unit 32 wave + 8 q + 4 half + i
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4v bv = *reinterpret_cast<const f32x4v*>(a.b2 + 32 * wave + 8 * q + 4 * half);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        h2[mt][4 * q + i] = (SZ_INFER_ABLATE & 32) ? h2[mt][4 * q + i] + bv[i] : gemm::fast_tanh(h2[mt][4 * q + i] + bv[i]);
            }
        }
        __syncthreads();  // h1 has been read by everyone: its space takes the partial logits
        if (SZ_INFER_ABLATE & 4) {
            if (wave < JT && h2[0][0] == 123.456f) atomicAdd(&hist[0], 1u);
            continue;
        }
        // ---- 3. output layer: partial logits over the wave's units, then the sum over the waves
        if (wave < JT) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x16 p;
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[r] = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) p = SZ_MFMA32(w3v[ct][r >> 2][r & 3], h2[mt][r], p);
                    float* dst = h1s + (size_t)((wave * MT + mt) * 32 + l31) * PLD + 32 * ct + 4 * half;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4v o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = p[4 * q + i];
                        *reinterpret_cast<f32x4v*>(dst + 8 * q) = o;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (SZ_INFER_ABLATE & 64) continue;
            const uint64_t row = row0 + 32 * mt + rm;
            const bool row_ok = row < a.rows;
            float v[2 * NCT];
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < 2 * NCT; ++u) {
                const int c = cl + 16 * u;
                float s = b3v[u];
                float part[kWaves];  // (all requested before the first is added)
#pragma unroll
                for (int w = 0; w < kWaves; ++w) part[w] = h1s[(size_t)(((w < JT ? w : 0) * MT + mt) * 32 + rm) * PLD + c];
#pragma unroll
                for (int w = 0; w < kWaves; ++w) s += w < JT ? part[w] : 0.0f;
                v[u] = c < a.C ? s : -INFINITY;
                mx = fmaxf(mx, v[u]);
            }
            mx = row16_max(mx);
            float sum = 0.0f;
#pragma unroll
            for (int u = 0; u < 2 * NCT; ++u) {
                v[u] = cl + 16 * u < a.C ? expf(v[u] - mx) : 0.0f;
                sum += v[u];
            }
            sum = row16_sum(sum);
            float best = -1.0f, best_c = -1.0f;
#pragma unroll
            for (int u = 0; u < 2 * NCT; ++u) {
                const float p = v[u] / sum;
                if (cl + 16 * u < a.C && p >= best) { best = p; best_c = (float)(cl + 16 * u); }
                if (row_ok) pacc[u] += (double)p;
            }
            const float top = row16_max(best);
            const float win = row16_max(best == top ? best_c : -1.0f);  // the last maximum wins
            if (cl == 0 && row_ok && top >= a.threshold && win >= 0.0f) atomicAdd(&hist[(int)win], 1u);
        }
        __syncthreads();  // the partial logits have been read: the next tile's h1 may overwrite them
    }
    if (a.psums) {
#pragma unroll
        for (int u = 0; u < 2 * NCT; ++u)
            if (cl + 16 * u < a.C) atomicAdd(&dsum[cl + 16 * u], pacc[u]);
    }
    __syncthreads();
    if (tid < a.C && !(SZ_INFER_ABLATE & 8)) {
        if (a.counts && hist[tid]) atomicAdd(&a.counts[tid], (unsigned long long)hist[tid]);
        if (a.psums) atomicAdd(&a.psums[tid], dsum[tid]);
    }
}

}  // namespace infer
}  // namespace sz

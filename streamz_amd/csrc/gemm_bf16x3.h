// f32 GEMM on the bf16 matrix cores: every f32 operand value is split exactly into three bf16
// pieces (x = hi + mid + lo, 8 significand bits each, round-to-nearest), and a product a.b is the
// sum of the six piece products whose weight is at least 2^-16 of the leading one
//   hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi
// each of which is exact in f32; the accumulation is f32 inside v_mfma_f32_32x32x16_bf16.  The
// three dropped products are below 2^-24 of a.b, the rounding unit of an f32 product.  Six bf16
// MFMAs do the work of eight v_mfma_f32_32x32x2_f32 at one sixteenth of their cost each.
//
// Splitting happens once per element on the way into LDS; weights arrive pre-split from
// split_planes_kernel.  The accumulator layout is that of the f32 kernel (gemm_kernel.h), so the
// epilogues are shared with it.
//
// The file's second kernel, gemm_f32w_kernel, is the same 256 x 256 walking structure on
// v_mfma_f32_32x32x2_f32 with unsplit f32 operands: the exact-f32 kernel for products large enough
// to give every compute unit several such tiles.
#pragma once
#include "gemm_kernel.h"

namespace sz {
namespace gemm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // a plain vector: arrays of it stay in registers

enum X3Src {
    X3_ROWS_F32 = 0,   // f32 X[row][k]: the tile dimension runs over rows, k is contiguous
    X3_KROWS_F32 = 1,  // f32 X[k][col]: the tile dimension is contiguous, k runs over rows
    X3_ROWS_SPLIT = 2  // bf16 planes P[plane][row][k] made by split_planes_kernel
};

// two f32 values -> their three bf16 pieces, packed (first value in the low half)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo)
{
    f32x2 v = {x0, x1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    const bf16x2 m = __builtin_convertvector(r, bf16x2);
    mid = __builtin_bit_cast(uint32_t, m);
    f32x2 q = {r[0] - __uint_as_float(mid << 16), r[1] - __uint_as_float(mid & 0xffff0000u)};
    const bf16x2 l = __builtin_convertvector(q, bf16x2);
    lo = __builtin_bit_cast(uint32_t, l);
}

// dst[plane][i] = piece `plane` of src[i]; count is even and both arrays are 4-byte aligned
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, uint64_t count,
                                                           uint32_t* __restrict__ dst, uint64_t plane_words)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (2 * i >= count) return;
    const float2 v = *reinterpret_cast<const float2*>(src + 2 * i);
    uint32_t h, m, l;
    split2(v.x, v.y, h, m, l);
    dst[i] = h;
    dst[plane_words + i] = m;
    dst[2 * plane_words + i] = l;
}

// ---- 256 x 256 block tile ------------------------------------------------------------------
// Eight waves (4 x 2, each 64 rows x 128 columns = 2 x 4 MFMA tiles), one workgroup per compute
// unit.  Operands are fetched 32 k at a time into registers and enter LDS 16 k at a time: LDS holds
// two stages of [plane][256][16 k] per operand (row stride 24 bf16 = 48 bytes, conflict-free for the
// 16-byte fragment reads), so one barrier per 16-k step suffices: while the MFMAs of step j read
// stage j % 2, the other stage receives the k values of step j + 1, and once both halves of a
// register tile are in LDS the next 32 k are requested, two steps ahead of their use.
constexpr int XW_TM = 256, XW_TN = 256, XW_THREADS = 512;
constexpr int XW_LD = 24;                    // bf16 per LDS row
constexpr int XW_PLANE = 256 * XW_LD;        // bf16 per plane
constexpr int XW_OPER = 3 * XW_PLANE;        // one operand, one stage
constexpr int XW_STAGE = 2 * XW_OPER;        // A then B
constexpr size_t XW_LDS_BYTES = 2 * XW_STAGE * sizeof(__bf16);

template <int SRC> struct XWOperand;

template <> struct XWOperand<X3_ROWS_F32> {
    float4 f[4];  // f[2 j + h]: rows t / 4 + 128 j, k values 16 h + 4 (t % 4) ...
    __device__ __forceinline__ void load(const void* base, int ld, uint64_t, int r0, int r_max, int k0, int)
    {
        const float* b = (const float*)base + k0 + 4 * (threadIdx.x & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = min(r0 + (int)(threadIdx.x >> 2) + 128 * j, r_max - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) f[2 * j + h] = *reinterpret_cast<const float4*>(b + (size_t)row * ld + 16 * h);
        }
    }
    template <int H> __device__ __forceinline__ void stash(__bf16* S) const
    {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint2 h, m, l;
            split2(f[2 * j + H].x, f[2 * j + H].y, h.x, m.x, l.x);
            split2(f[2 * j + H].z, f[2 * j + H].w, h.y, m.y, l.y);
            __bf16* at = S + ((threadIdx.x >> 2) + 128 * j) * XW_LD + 4 * (threadIdx.x & 3);
            *reinterpret_cast<uint2*>(at) = h;
            *reinterpret_cast<uint2*>(at + XW_PLANE) = m;
            *reinterpret_cast<uint2*>(at + 2 * XW_PLANE) = l;
        }
    }
};

template <> struct XWOperand<X3_KROWS_F32> {
    float f[16];  // f[8 h + j]: column t % 256, k = 16 h + 8 (t / 256) + j
    // all 32 rows are read: the caller keeps zeros in the rows between the end of the data and
    // the next multiple of 32
    __device__ __forceinline__ void load(const void* base, int ld, uint64_t, int c0, int, int k0, int)
    {
        const float* b = (const float*)base + (size_t)(k0 + 8 * (int)(threadIdx.x >> 8)) * ld + c0 + (threadIdx.x & 255);
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = b[(size_t)(16 * (i >> 3) + (i & 7)) * ld];
    }
    template <int H> __device__ __forceinline__ void stash(__bf16* S) const
    {
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split2(f[8 * H + 2 * i], f[8 * H + 2 * i + 1], h[i], m[i], l[i]);
        __bf16* at = S + (threadIdx.x & 255) * XW_LD + 8 * (threadIdx.x >> 8);
        *reinterpret_cast<uint4*>(at) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(at + XW_PLANE) = make_uint4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<uint4*>(at + 2 * XW_PLANE) = make_uint4(l[0], l[1], l[2], l[3]);
    }
};

template <> struct XWOperand<X3_ROWS_SPLIT> {
    u32x4 f[6];  // f[2 p + h]: plane p, row t / 2, k values 16 h + 8 (t % 2) ...
    __device__ __forceinline__ void load(const void* base, int ld, uint64_t plane, int r0, int, int k0, int)
    {
        const __bf16* b = (const __bf16*)base + (size_t)(r0 + (threadIdx.x >> 1)) * ld + k0 + 8 * (threadIdx.x & 1);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) f[2 * p + h] = *reinterpret_cast<const u32x4*>(b + (size_t)p * plane + 16 * h);
    }
    template <int H> __device__ __forceinline__ void stash(__bf16* S) const
    {
        __bf16* at = S + (threadIdx.x >> 1) * XW_LD + 8 * (threadIdx.x & 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(at + p * XW_PLANE) = f[2 * p + H];
    }
};

// C[M][N] = epilogue(A . B) with the operands described by ASRC / BSRC:
//   X3_ROWS_F32    A = g.A [M][K] (lda)
//   X3_KROWS_F32   A = g.A [K][M] (lda)            B = g.B [K][N] (ldb)
//   X3_ROWS_SPLIT                                   B = g.B planes [3][N][K] (ldb = K, g.b_plane apart)
// Requirements checked by the host: N and, for a KROWS A, M are multiples of 256; the K range of
// every grid z index is a multiple of 32 long and KROWS operands hold zeros in the rows from K to the
// next multiple of 32;
// g.lda / g.ldb rows 16-byte aligned.
//
// The grid is one-dimensional and a workgroup walks over tiles  first, first + stride, ...  of the
// g.tiles_x * g.tiles_y * g.tiles_z tiles (x fastest).  The operand pipeline runs across tiles: the
// last trip of a tile fetches the first 32 k of the next one, so a tile's epilogue sits between two
// trips with the next operands already in LDS and in registers, and its stores drain under the
// next tile's first steps (with one workgroup per compute unit nothing else would hide them).
template <int ASRC, int BSRC, int EPI, int TAG>
__global__ __launch_bounds__(XW_THREADS) void gemm_x3w_kernel(GemmArgs g)
{
    constexpr int TN = 4, WN = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char xw_lds[];
    __bf16* L = reinterpret_cast<__bf16*>(xw_lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int a_off = (wm * 64 + l31) * XW_LD + 8 * half;
    const int b_off = XW_OPER + (wn * WN + l31) * XW_LD + 8 * half;

    // Tiles are numbered so that each XCD (workgroups go to the 8 XCDs round-robin in launch
    // order) works on one contiguous run of them: neighbours that share operand rows share one L2.
    const int n_tiles = g.tiles_x * g.tiles_y * g.tiles_z;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int run = (n_tiles + 7) >> 3;           // tiles per XCD
    const int run_end = min(n_tiles, (xcd + 1) * run);
    int tile = xcd * run + slot;
    if (tile >= run_end) return;
    struct Pos { int m0, n0, k_begin, k_end, bz; };
    auto place = [&](int t) {
        Pos p;
        const int bx = t % g.tiles_x, by = (t / g.tiles_x) % g.tiles_y;
        p.bz = t / (g.tiles_x * g.tiles_y);
        p.m0 = by * XW_TM; p.n0 = bx * XW_TN;
        p.k_begin = p.bz * g.k_chunk;
        p.k_end = min(g.K, p.k_begin + g.k_chunk);
        return p;
    };
    Pos cur = place(tile);

    XWOperand<ASRC> fa;
    XWOperand<BSRC> fb;
    fa.load(g.A, g.lda, 0, cur.m0, g.M, cur.k_begin, cur.k_end);
    fb.load(g.B, g.ldb, g.b_plane, cur.n0, g.N, cur.k_begin, cur.k_end);
    fa.template stash<0>(L);
    fb.template stash<0>(L + XW_OPER);
    __syncthreads();

    while (true) {
        float* C = g.C;
        if (EPI == EPI_SLAB) C += (size_t)cur.bz * (g.slab_stride ? (size_t)g.slab_stride : (size_t)g.M * (size_t)g.ldc);
        f32x16 acc[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        if (EPI == EPI_SLAB && g.slab_add) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col = cur.n0 + wn * WN + tn * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = cur.m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        acc[tm][tn][r] = C[(size_t)row * g.ldc + col];
                    }
                }
        }
        const int next = tile + per_xcd;
        const bool has_next = next < run_end;
        const Pos nxt = has_next ? place(next) : cur;

        bf16x8 a[2][3];
        auto load_a = [&](const __bf16* S) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[t][p] = *reinterpret_cast<const bf16x8*>(S + a_off + p * XW_PLANE + t * 32 * XW_LD);
        };
        auto multiply = [&](const __bf16* S, int tn0) {  // column tiles tn0, tn0 + 1
#pragma unroll
            for (int tn = tn0; tn < tn0 + 2; ++tn) {
                bf16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(S + b_off + p * XW_PLANE + tn * 32 * XW_LD);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    f32x16 c = acc[tm][tn];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[0], c, 0, 0, 0);
                    acc[tm][tn] = c;
                }
            }
        };
        // Two 16-k steps per trip.  On entry the registers hold the 32 k starting at k0 and their
        // first half is in stage 0.
        for (int k0 = cur.k_begin; k0 < cur.k_end; k0 += 32) {
            // what this trip fetches: the tile's next 32 k; on its last trip the first 32 k of the
            // next tile; on the workgroup's very last trip its own range again (never used)
            const bool last = k0 + 32 >= cur.k_end;
            const int fm0 = last ? nxt.m0 : cur.m0, fn0 = last ? nxt.n0 : cur.n0;
            const int fk = last ? (has_next ? nxt.k_begin : k0) : k0 + 32, fk_end = last ? nxt.k_end : cur.k_end;
            // step 0 reads stage 0; stage 1 receives the second half of the register tile, after
            // which the registers are free for the next fetch
            load_a(L);
            fa.template stash<1>(L + XW_STAGE);
            fb.template stash<1>(L + XW_STAGE + XW_OPER);
            fa.load(g.A, g.lda, 0, fm0, g.M, fk, fk_end);
            fb.load(g.B, g.ldb, g.b_plane, fn0, g.N, fk, fk_end);
            multiply(L, 0);
            multiply(L, 2);
            __syncthreads();
            // step 1 reads stage 1; stage 0 receives the first half of the fetched registers in
            // the step's second half: their loads were issued one step ago
            load_a(L + XW_STAGE);
            multiply(L + XW_STAGE, 0);
            __builtin_amdgcn_sched_barrier(0);
            fa.template stash<0>(L);
            fb.template stash<0>(L + XW_OPER);
            multiply(L + XW_STAGE, 2);
            __syncthreads();
        }
        finish_tile<EPI, XW_TM, TN, WN>(g, C, acc, cur.m0, cur.n0, wm, wn, l31, half);
        if (!has_next) break;
        tile = next;
        cur = nxt;
    }
}


// ---- the same 256 x 256 walking structure on v_mfma_f32_32x32x2_f32 (exact f32) ---------------
// LDS: two stages of [256][16 k] f32 per operand, row stride 20 floats (80 bytes, conflict-free
// 16-byte reads).  One MFMA adds two k values, that of lanes 0-31 and that of lanes 32-63: step s of a
// stage pairs k = s with k = s + 8, so that the eight values a lane needs from its row are contiguous.
typedef float v4f __attribute__((ext_vector_type(4)));  // a plain vector: arrays of it stay in registers
constexpr int FW_LD = 20;                    // floats per LDS row
constexpr int FW_OPER = 256 * FW_LD;         // one operand, one stage
constexpr int FW_STAGE = 2 * FW_OPER;
constexpr size_t FW_LDS_BYTES = 2 * FW_STAGE * sizeof(float);

template <int SRC> struct FWOperand;

template <> struct FWOperand<X3_ROWS_F32> {
    v4f f[4];  // f[2 j + h]: rows t / 4 + 128 j, k values 16 h + 4 (t % 4) ...
    __device__ __forceinline__ void load(const void* base, int ld, int r0, int r_max, int k0)
    {
        const float* b = (const float*)base + k0 + 4 * (threadIdx.x & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = min(r0 + (int)(threadIdx.x >> 2) + 128 * j, r_max - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) f[2 * j + h] = *reinterpret_cast<const v4f*>(b + (size_t)row * ld + 16 * h);
        }
    }
    template <int H> __device__ __forceinline__ void stash(float* S) const
    {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<v4f*>(S + ((threadIdx.x >> 2) + 128 * j) * FW_LD + 4 * (threadIdx.x & 3)) = f[2 * j + H];
    }
};

template <> struct FWOperand<X3_KROWS_F32> {
    float f[16];  // f[8 h + j]: column t % 256, k = 16 h + 8 (t / 256) + j
    __device__ __forceinline__ void load(const void* base, int ld, int c0, int, int k0)
    {
        const float* b = (const float*)base + (size_t)(k0 + 8 * (int)(threadIdx.x >> 8)) * ld + c0 + (threadIdx.x & 255);
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = b[(size_t)(16 * (i >> 3) + (i & 7)) * ld];
    }
    template <int H> __device__ __forceinline__ void stash(float* S) const
    {
        float* at = S + (threadIdx.x & 255) * FW_LD + 8 * (threadIdx.x >> 8);
        *reinterpret_cast<v4f*>(at) = v4f{f[8 * H], f[8 * H + 1], f[8 * H + 2], f[8 * H + 3]};
        *reinterpret_cast<v4f*>(at + 4) = v4f{f[8 * H + 4], f[8 * H + 5], f[8 * H + 6], f[8 * H + 7]};
    }
};

// A = g.A as for gemm_x3w_kernel; B = g.B [N][K] f32 (X3_ROWS_F32) or [K][N] (X3_KROWS_F32).
template <int ASRC, int BSRC, int EPI, int TAG>
__global__ __launch_bounds__(XW_THREADS) void gemm_f32w_kernel(GemmArgs g)
{
    constexpr int TN = 4, WN = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char xw_lds[];
    float* L = reinterpret_cast<float*>(xw_lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int a_off = (wm * 64 + l31) * FW_LD + 8 * half;
    const int b_off = FW_OPER + (wn * WN + l31) * FW_LD + 8 * half;

    const int n_tiles = g.tiles_x * g.tiles_y * g.tiles_z;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int run = (n_tiles + 7) >> 3;
    const int run_end = min(n_tiles, (xcd + 1) * run);
    int tile = xcd * run + slot;
    if (tile >= run_end) return;
    struct Pos { int m0, n0, k_begin, k_end, bz; };
    auto place = [&](int t) {
        Pos p;
        const int bx = t % g.tiles_x, by = (t / g.tiles_x) % g.tiles_y;
        p.bz = t / (g.tiles_x * g.tiles_y);
        p.m0 = by * XW_TM; p.n0 = bx * XW_TN;
        p.k_begin = p.bz * g.k_chunk;
        p.k_end = min(g.K, p.k_begin + g.k_chunk);
        return p;
    };
    Pos cur = place(tile);

    FWOperand<ASRC> fa;
    FWOperand<BSRC> fb;
    // EARLY (the split-K products, whose tiles are short): a tile's first 32 k are in the two stages
    // before its first trip starts, so that nothing fetched ahead of a tile's stores is waited for
    // behind them -- the memory counter covers stores as well as loads.  Measured: 10 % on those
    // products, and 2 % the other way on the long tiles of the row-major products, which keep the
    // second half in registers across the epilogue.
    constexpr bool EARLY = ASRC == X3_KROWS_F32;
    fa.load(g.A, g.lda, cur.m0, g.M, cur.k_begin);
    fb.load(g.B, g.ldb, cur.n0, g.N, cur.k_begin);
    fa.template stash<0>(L);
    fb.template stash<0>(L + FW_OPER);
    if (EARLY) {
        fa.template stash<1>(L + FW_STAGE);
        fb.template stash<1>(L + FW_STAGE + FW_OPER);
    }
    __syncthreads();

    while (true) {
        float* C = g.C;
        if (EPI == EPI_SLAB) C += (size_t)cur.bz * (g.slab_stride ? (size_t)g.slab_stride : (size_t)g.M * (size_t)g.ldc);
        f32x16 acc[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        if (EPI == EPI_SLAB && g.slab_add) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col = cur.n0 + wn * WN + tn * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = cur.m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        acc[tm][tn][r] = C[(size_t)row * g.ldc + col];
                    }
                }
        }
        const int next = tile + per_xcd;
        const bool has_next = next < run_end;
        const Pos nxt = has_next ? place(next) : cur;

        v4f a[2][2];
        auto load_a = [&](const float* S) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 2; ++q) a[t][q] = *reinterpret_cast<const v4f*>(S + a_off + t * 32 * FW_LD + 4 * q);
        };
        auto multiply = [&](const float* S, int tn0) {  // column tiles tn0, tn0 + 1
#pragma unroll
            for (int tn = tn0; tn < tn0 + 2; ++tn) {
                v4f b[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) b[q] = *reinterpret_cast<const v4f*>(S + b_off + tn * 32 * FW_LD + 4 * q);
                const float bv[8] = {b[0].x, b[0].y, b[0].z, b[0].w, b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const float av[8] = {a[tm][0].x, a[tm][0].y, a[tm][0].z, a[tm][0].w, a[tm][1].x, a[tm][1].y, a[tm][1].z, a[tm][1].w};
                    f32x16 c = acc[tm][tn];
#pragma unroll
                    for (int st = 0; st < 8; ++st) c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bv[st], c, 0, 0, 0);
                    acc[tm][tn] = c;
                }
            }
        };
        for (int k0 = cur.k_begin; k0 < cur.k_end; k0 += 32) {
            const bool last = k0 + 32 >= cur.k_end;
            const int fm0 = last ? nxt.m0 : cur.m0, fn0 = last ? nxt.n0 : cur.n0;
            const int fk = last ? (has_next ? nxt.k_begin : k0) : k0 + 32;
            load_a(L);
            if (!EARLY || k0 != cur.k_begin) {  // (EARLY: a tile's first trip finds stage 1 filled)
                fa.template stash<1>(L + FW_STAGE);
                fb.template stash<1>(L + FW_STAGE + FW_OPER);
            }
            fa.load(g.A, g.lda, fm0, g.M, fk);
            fb.load(g.B, g.ldb, fn0, g.N, fk);
            multiply(L, 0);
            multiply(L, 2);
            __syncthreads();
            load_a(L + FW_STAGE);
            multiply(L + FW_STAGE, 0);
            __builtin_amdgcn_sched_barrier(0);
            fa.template stash<0>(L);
            fb.template stash<0>(L + FW_OPER);
            multiply(L + FW_STAGE, 2);
            __syncthreads();
        }
        if (EARLY && has_next) {  // the next tile's k 16..31, ahead of this tile's stores
            fa.template stash<1>(L + FW_STAGE);
            fb.template stash<1>(L + FW_STAGE + FW_OPER);
        }
        finish_tile<EPI, XW_TM, TN, WN>(g, C, acc, cur.m0, cur.n0, wm, wn, l31, half);
        if (!has_next) break;
        tile = next;
        cur = nxt;
    }
}

}  // namespace gemm
}  // namespace sz

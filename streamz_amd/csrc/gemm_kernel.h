// LDS-tiled f32 GEMM on v_mfma_f32_32x32x2_f32 with fused epilogues; included by net.hip and by
// the tuning harness tools/gemm_bench.hip.
//
// C[M][N] = epilogue(sum_k A(m,k) B(k,n)).  Block tile TM x 128 (TM = 128: 4 waves as 2x2 of
// 64x64; TM = 64: 1x4 of 64x32), K tile KT.  One MFMA adds two k values: the one of lanes 0-31
// and the one of lanes 32-63.  With A row-major in memory (A[m][k]) step s of a K tile pairs
// k = s with k = s + KT/2, so that the KT/2 values a lane needs from its row of A are contiguous:
// the A tile stays row-major in LDS (row stride KT + 4 floats; 16-byte stores and loads, no bank
// conflicts).  Tiles whose rows run along k are kept k-major (row stride 132 floats), a fragment
// read being 32 consecutive floats per half-wave, and a step pairs k = 2 s with 2 s + 1.  The next K tile is fetched into registers
// while the current one is multiplied; with DB the LDS tiles are double-buffered and one
// barrier per K tile suffices.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sz {
namespace gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128;
constexpr int LDT = BM + 4;  // LDS row stride (floats) of k-major operands: 16-byte aligned rows
constexpr int LDR = BM + 1;  // LDS row stride of operands transposed on the way in: scalar stores <= 2-way
constexpr int kGemmThreads = 256;

enum AMode { A_ROWS = 0 /* A[m][k] */, A_KROWS = 1 /* A[k][m] */ };
enum BMode { B_KROWS = 0 /* B[k][n] */, B_ROWS = 1 /* B[n][k] */ };
enum Epi {
    EPI_BIAS_RELU = 0,  // C = relu(acc + bias[n])        (v > 0 ? v : 0, lib.rs:882)
    EPI_BIAS_TANH = 1,  // C = tanh(acc + bias[n])        (lib.rs:883)
    EPI_BIAS = 2,       // C = acc + bias[n]
    EPI_MUL_DTANH = 3,  // C = acc * (1 - aux^2)          (lib.rs:1034)
    EPI_MUL_DRELU = 4,  // C = acc * (aux > 0)            (lib.rs:1039-1040)
    EPI_SLAB = 5,       // C[z] (+)= acc                  (split-K partial)
    EPI_BIAS_RELU_MASK = 6,  // EPI_BIAS_RELU, and bit (m % 32) of mask[m / 32][n] = (C > 0)
    EPI_MUL_MASK = 7    // C = acc * bit of mask[m / 32][n]   (same product as EPI_MUL_DRELU)
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
    int k_chunk;   // K range handled by one grid z index
    int a_vec, b_vec;
    int ones_row = -1;  // A_KROWS only: logical row M-1 of A is all ones (bias-gradient row)
    int slab_add = 0;   // EPI_SLAB only: add to the slab instead of overwriting it
    int a_pad = 0;      // A's allocation is padded (finite values) up to the tile boundary past M resp. K,
                        // so edge tiles of A may be fetched without bounds checks
    uint32_t* mask = nullptr;  // EPI_*_MASK: [ceil(M / 32)][N] sign bits of the ReLU layer, one word per
                               // column and block of 32 rows (what one lane of an MFMA tile holds)
    uint64_t b_plane = 0;      // split-bf16 kernel: elements between the three planes of a pre-split B
    int tiles_x = 0, tiles_y = 0, tiles_z = 0;  // split-bf16 kernel: tiles along N, M and the K ranges
    uint64_t slab_stride = 0;  // split-bf16 kernel, EPI_SLAB: floats between the slabs of two K ranges (0: M * ldc)
};

struct Frag4 { float v[4]; };

// tanh from the hardware exponential and reciprocal (1 ulp each); absolute error ~1e-7.  No clamp and no
// special cases are needed: e^(2x) overflows to +inf (1 - 2/inf = 1) or underflows to 0 (1 - 2/1 = -1) exactly where
// the result has long rounded to +-1, and a NaN goes through both instructions as a NaN (the reference's tanh,
// lib.rs:883, carries it on too)
__device__ __forceinline__ float fast_tanh(float x)
{
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2^(x * 2 log2 e) = e^(2x)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// v > 0 ? v : 0 (lib.rs:882) as the one instruction it is: v_max_f32 returns the other operand for a NaN, and
// written in source the compiler first spends a second instruction on quieting one
__device__ __forceinline__ float relu_max(float v)
{
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}

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
// (A[m][k], B[n][k]): TR rows x 16 k, stored transposed (k-major) into LDS.
template <int TR, int KT>
__device__ __forceinline__ void fetch_rows(const float* base, int ld, int r0, int r_max, int k0, int k_max, int vec,
                                           Frag4 (&f)[TR * KT / 1024])
{
#pragma unroll
    for (int i = 0; i < TR * KT / 1024; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        f[i] = load4(base, r0 + idx / (KT / 4), k0 + 4 * (idx % (KT / 4)), ld, r_max, k_max, vec);
    }
}
// the same fragments kept row-major: S[row][k], row stride KT + 4
template <int TR, int KT>
__device__ __forceinline__ void stash_rows_rowmajor(float* S, const Frag4 (&f)[TR * KT / 1024])
{
#pragma unroll
    for (int i = 0; i < TR * KT / 1024; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        float4 q;
        q.x = f[i].v[0]; q.y = f[i].v[1]; q.z = f[i].v[2]; q.w = f[i].v[3];
        *reinterpret_cast<float4*>(S + (idx / (KT / 4)) * (KT + 4) + 4 * (idx % (KT / 4))) = q;
    }
}
// PAIR: as in stash_krows below
template <int TR, int KT, int PAIR = 0>
__device__ __forceinline__ void stash_rows(float* S, const Frag4 (&f)[TR * KT / 1024])
{
#pragma unroll
    for (int i = 0; i < TR * KT / 1024; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        const int row = idx / (KT / 4), kq = idx % (KT / 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * kq + j;
            S[(PAIR ? 2 * (k % (KT / 2)) + k / (KT / 2) : k) * LDR + row] = f[i].v[j];
        }
    }
}
// Tile whose global rows run along k (A[k][m], B[k][n]): 16 k x TR, copied straight.
template <int TR, int KT>
__device__ __forceinline__ void fetch_krows(const float* base, int ld, int c0, int c_max, int k0, int k_max, int vec,
                                            int ones_col, Frag4 (&f)[TR * KT / 1024])
{
#pragma unroll
    for (int i = 0; i < TR * KT / 1024; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        const int krow = k0 + idx / (TR / 4), col = c0 + 4 * (idx % (TR / 4));
        f[i] = load4(base, krow, col, ld, k_max, c_max, vec);
        if (ones_col >= 0 && krow < k_max) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (col + j == ones_col) f[i].v[j] = 1.0f;
        }
    }
}
// PAIR: k row k goes to LDS row 2 (k % (KT/2)) + k / (KT/2), so that rows k and k + KT/2, which
// one MFMA step multiplies, are neighbours.
template <int TR, int KT, int PAIR = 0, int LD = LDT>
__device__ __forceinline__ void stash_krows(float* S, const Frag4 (&f)[TR * KT / 1024])
{
#pragma unroll
    for (int i = 0; i < TR * KT / 1024; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        float4 q;
        q.x = f[i].v[0]; q.y = f[i].v[1]; q.z = f[i].v[2]; q.w = f[i].v[3];
        const int k = idx / (TR / 4);
        const int row = PAIR ? 2 * (k % (KT / 2)) + k / (KT / 2) : k;
        *reinterpret_cast<float4*>(S + row * LD + 4 * (idx % (TR / 4))) = q;
    }
}

// Unguarded variants for tiles that lie completely inside the operands with 16-byte aligned rows.
template <int TR, int KT>
__device__ __forceinline__ void fetch_rows_fast(const float* base, int ld, int r0, int k0, Frag4 (&f)[TR * KT / 1024])
{
#pragma unroll
    for (int i = 0; i < TR * KT / 1024; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        const float4 q = *reinterpret_cast<const float4*>(base + (int64_t)(r0 + idx / (KT / 4)) * ld + k0 + 4 * (idx % (KT / 4)));
        f[i].v[0] = q.x; f[i].v[1] = q.y; f[i].v[2] = q.z; f[i].v[3] = q.w;
    }
}
template <int TR, int KT>
__device__ __forceinline__ void fetch_krows_fast(const float* base, int ld, int c0, int k0, Frag4 (&f)[TR * KT / 1024])
{
#pragma unroll
    for (int i = 0; i < TR * KT / 1024; ++i) {
        const int idx = threadIdx.x + kGemmThreads * i;
        const float4 q = *reinterpret_cast<const float4*>(base + (int64_t)(k0 + idx / (TR / 4)) * ld + c0 + 4 * (idx % (TR / 4)));
        f[i].v[0] = q.x; f[i].v[1] = q.y; f[i].v[2] = q.z; f[i].v[3] = q.w;
    }
}

// Workgroups are handed to the 8 XCDs round-robin in launch order.  Tiles are renumbered so
// that each XCD works on one contiguous run of them (column tiles of a row tile, then row
// tiles, then K ranges): neighbours that share operand rows then share one L2.
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz)
{
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = lin & 7u, slot = lin >> 3, q = nb >> 3, r = nb & 7u;
    const unsigned t = xcd < r ? xcd * (q + 1) + slot : r * (q + 1) + (xcd - r) * q + slot;
    bx = (int)(t % gridDim.x);
    by = (int)((t / gridDim.x) % gridDim.y);
    bz = (int)(t / (gridDim.x * gridDim.y));
}

// Epilogue shared by the kernels whose accumulators have the 32x32 MFMA layout: a lane holds
// column `col` of a tile and the 16 rows (r & 3) + 8 (r >> 2) + 4 half.
// First pass: bias / activation / derivative applied in place (finish_values); second pass: the
// stores (store_tile).  finish_tile is the two in a row.
template <int EPI, int TM, int TN, int WN>
__device__ __forceinline__ void finish_values(const GemmArgs& g, f32x16 (&acc)[2][TN], int m0, int n0, int wm, int wn, int l31,
                                              int half)
{
    // One MFMA tile is 32 rows x 32 columns; a lane holds column `col` and the 16 rows
    // (r & 3) + 8 (r >> 2) + 4 half.  Its 16 sign bits and those of lane ^ 32 make the mask word
    // of (row block, col).  Everything the epilogue reads is requested before the first store
    // (the memory counter of this ISA covers loads and stores alike).
    float biasv[TN];
    uint32_t mwordv[2][TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int colc = min(n0 + wn * WN + tn * 32 + l31, g.N - 1);
        biasv[tn] = 0.0f;
        if (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_TANH || EPI == EPI_BIAS || EPI == EPI_BIAS_RELU_MASK)
            biasv[tn] = g.bias[colc];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            mwordv[tm][tn] = 0;
            if (EPI == EPI_MUL_MASK) mwordv[tm][tn] = g.mask[(size_t)((m0 + wm * 64 + tm * 32) >> 5) * g.N + colc];
        }
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * WN + tn * 32 + l31;
            if (col >= g.N) continue;
            const float bias = biasv[tn];
            const size_t mask_at = (size_t)((m0 + wm * 64 + tm * 32) >> 5) * g.N + col;
            uint32_t mword = mwordv[tm][tn];
            const uint32_t mhalf = mwordv[tm][tn] >> (4 * half);  // EPI_MUL_MASK: this lane's 16 rows are bits rofs of it
            if (EPI == EPI_BIAS_RELU_MASK) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc[tm][tn][r] + bias > 0.0f) mword |= 1u << ((r & 3) + 8 * (r >> 2) + 4 * half);
                mword |= (uint32_t)__shfl_xor((int)mword, 32);
                if (half == 0) g.mask[mask_at] = mword;
            }
            const int row_base = m0 + wm * 64 + tm * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rofs = (r & 3) + 8 * (r >> 2);
                const int row = min(row_base + rofs, g.M - 1);  // clamped: only used to read aux
                float v = acc[tm][tn][r];
                if (EPI == EPI_BIAS_RELU_MASK || EPI == EPI_BIAS_RELU) v = relu_max(v + bias);
                else if (EPI == EPI_BIAS_TANH) v = fast_tanh(v + bias);
                else if (EPI == EPI_BIAS) v += bias;
                else if (EPI == EPI_MUL_DTANH) { const float h = g.aux[(size_t)row * g.ldaux + col]; v *= (1.0f - h * h); }
                else if (EPI == EPI_MUL_DRELU) { const float h = g.aux[(size_t)row * g.ldaux + col]; v *= (h > 0.0f ? 1.0f : 0.0f); }
                else if (EPI == EPI_MUL_MASK) v *= (float)((mhalf >> rofs) & 1u);  // a product, as lib.rs:1039-1040: 0 x NaN stays NaN (bit field extract, convert, multiply)
                acc[tm][tn][r] = v;
            }
        }
}

// Every value was finished in place by finish_values, so everything the epilogue loads has arrived
// before the first store and the stores go out back to back.
template <int EPI, int TM, int TN, int WN>
__device__ __forceinline__ void store_tile(const GemmArgs& g, float* C, const f32x16 (&acc)[2][TN], int m0, int n0, int wm,
                                           int wn, int l31, int half)
{
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * WN + tn * 32 + l31;
            if (col >= g.N) continue;
            const int row_base = m0 + wm * 64 + tm * 32 + 4 * half;
            float* cp = C + (size_t)row_base * g.ldc + col;
            if (m0 + TM <= g.M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = acc[tm][tn][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rofs = (r & 3) + 8 * (r >> 2);
                    if (row_base + rofs < g.M) cp[(size_t)rofs * g.ldc] = acc[tm][tn][r];
                }
            }
        }
}

template <int EPI, int TM, int TN, int WN>
__device__ __forceinline__ void finish_tile(const GemmArgs& g, float* C, f32x16 (&acc)[2][TN], int m0, int n0, int wm, int wn,
                                            int l31, int half)
{
    finish_values<EPI, TM, TN, WN>(g, acc, m0, n0, wm, wn, l31, half);
    store_tile<EPI, TM, TN, WN>(g, C, acc, m0, n0, wm, wn, l31, half);
}

// TAG only separates instantiations, so that each call site is its own kernel in a profile.
// TM = rows of the block tile, KT = K tile (16 or 32), DB = double-buffered LDS tiles, BNT = columns
// of the block tile (128; 256 for a k-major B under a 64-row tile: twice the matrix work between the
// two barriers of a K tile).
template <int AM, int BMd, int EPI, int TAG, int TM, int KT, int DB, int BNT = BN>
__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(GemmArgs g)
{
    static_assert(BNT == BN || (BNT == 2 * BN && BMd == B_KROWS && TM == 64), "wide tiles: k-major B, 64 rows");
    constexpr int BK = KT;
    constexpr int WAVES_M = TM / 64, WAVES_N = 4 / WAVES_M;
    constexpr int WN = BNT / WAVES_N;  // columns per wave: 64 or 32
    constexpr int TN = WN / 32;       // 32-wide MFMA tiles per wave along N
    constexpr int LDTB = BNT + 4;     // row stride of a k-major B tile
    constexpr int LDA = AM == A_ROWS ? BK + 4 : LDT;  // A_ROWS: row-major tile [TM][BK + 4]
    constexpr int LDB = BMd == B_ROWS ? LDR : LDTB;
    constexpr int HS = BK / 2;                        // MFMA steps per K tile
    constexpr int A_TILE = AM == A_ROWS ? TM * LDA : BK * LDT;
    constexpr int B_TILE = BK * LDTB;
    __shared__ __attribute__((aligned(16))) float As_[(DB ? 2 : 1) * A_TILE];
    __shared__ __attribute__((aligned(16))) float Bs_[(DB ? 2 : 1) * B_TILE];
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    const int m0 = by * TM, n0 = bx * BNT;
    const int k_begin = bz * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;
    const int a_cols = g.ones_row >= 0 ? g.M - 1 : g.M;  // columns of A that exist in memory (A_KROWS)

    float* C = g.C;
    if (EPI == EPI_SLAB) C += (size_t)bz * (size_t)g.M * (size_t)g.ldc;
    // Accumulators start at zero, or, for a split-K slab that is being added to, at the slab's
    // current contents: those loads overlap the first operand fetch.
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
                const int col = n0 + wn * WN + tn * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (col < g.N && row < g.M) acc[tm][tn][r] = C[(size_t)row * g.ldc + col];
                }
            }
    }

    Frag4 fa[TM * BK / 1024], fb[BNT * BK / 1024];
    // interior tiles (the common case) take straight-line 16-byte loads; edge tiles are guarded
    const bool a_inside = g.a_vec && g.ones_row < 0 && (m0 + TM <= g.M || (AM == A_KROWS && g.a_pad));
    const bool b_inside = g.b_vec && n0 + BNT <= g.N;
    auto fetch = [&](int k0) {
        const bool k_inside = k0 + BK <= k_end;
        if (a_inside && (k_inside || (AM == A_ROWS && g.a_pad))) {
            if (AM == A_ROWS) fetch_rows_fast<TM, BK>(g.A, g.lda, m0, k0, fa);
            else fetch_krows_fast<TM, BK>(g.A, g.lda, m0, k0, fa);
        } else {
            if (AM == A_ROWS) fetch_rows<TM, BK>(g.A, g.lda, m0, g.M, k0, k_end, g.a_vec, fa);
            else fetch_krows<TM, BK>(g.A, g.lda, m0, a_cols, k0, k_end, g.a_vec, g.ones_row, fa);
        }
        if (b_inside && k_inside) {
            if (BMd == B_ROWS) fetch_rows_fast<BNT, BK>(g.B, g.ldb, n0, k0, fb);
            else fetch_krows_fast<BNT, BK>(g.B, g.ldb, n0, k0, fb);
        } else {
            if (BMd == B_ROWS) fetch_rows<BNT, BK>(g.B, g.ldb, n0, g.N, k0, k_end, g.b_vec, fb);
            else fetch_krows<BNT, BK>(g.B, g.ldb, n0, g.N, k0, k_end, g.b_vec, -1, fb);
        }
    };
    auto stash = [&](int buf) {
        float* As = As_ + buf * A_TILE;
        float* Bs = Bs_ + buf * B_TILE;
        if (AM == A_ROWS) stash_rows_rowmajor<TM, BK>(As, fa); else stash_krows<TM, BK>(As, fa);
        if (BMd == B_ROWS) stash_rows<BNT, BK, AM == A_ROWS>(Bs, fb); else stash_krows<BNT, BK, AM == A_ROWS, LDTB>(Bs, fb);
    };
    if (k_begin < k_end) fetch(k_begin);
    if (DB && k_begin < k_end) { stash(0); __syncthreads(); }
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        if (!DB) {
            __syncthreads();
            stash(0);
            __syncthreads();
        }
        if (k0 + BK < k_end) fetch(k0 + BK);
        const float* As = As_ + buf * A_TILE;
        const float* Bs = Bs_ + buf * B_TILE;
        // A_ROWS: step s multiplies k = s (lanes 0-31) and k = s + HS (lanes 32-63); otherwise
        // k = 2 s and 2 s + 1.  The fragments of the next step are read from LDS while the MFMAs
        // of this one run.
        constexpr int KS = 2;  // LDS rows between steps (the B tile of an A_ROWS product is stored paired)
        const float* br = Bs + half * LDB + wn * WN + l31;
        float fbn[2][TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) fbn[0][tn] = br[32 * tn];
        if (AM == A_ROWS) {
            const float* ar = As + (wm * 64 + l31) * LDA + HS * half;
            float4 a0 = *reinterpret_cast<const float4*>(ar), a1 = *reinterpret_cast<const float4*>(ar + 32 * LDA);
#pragma unroll
            for (int s4 = 0; s4 < HS; s4 += 4) {
                float4 n0 = a0, n1 = a1;
                if (s4 + 4 < HS) {
                    n0 = *reinterpret_cast<const float4*>(ar + s4 + 4);
                    n1 = *reinterpret_cast<const float4*>(ar + 32 * LDA + s4 + 4);
                }
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int st = s4 + j, cur = st & 1, nxt = cur ^ 1;
                    if (st + 1 < HS) {
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) fbn[nxt][tn] = br[KS * (st + 1) * LDB + 32 * tn];
                    }
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], fbn[cur][tn], acc[0][tn], 0, 0, 0);
                        acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], fbn[cur][tn], acc[1][tn], 0, 0, 0);
                    }
                    // keep the LDS reads of the next step between the MFMA groups instead of ahead of them
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);      // DS read
                }
                a0 = n0; a1 = n1;
            }
        } else {
            const float* ar = As + half * LDA + wm * 64 + l31;
            float fa0[2], fa1[2];
            fa0[0] = ar[0]; fa1[0] = ar[32];
#pragma unroll
            for (int st = 0; st < HS; ++st) {
                const int cur = st & 1, nxt = cur ^ 1;
                if (st + 1 < HS) {
                    fa0[nxt] = ar[KS * (st + 1) * LDA]; fa1[nxt] = ar[KS * (st + 1) * LDA + 32];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) fbn[nxt][tn] = br[KS * (st + 1) * LDB + 32 * tn];
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], fbn[cur][tn], acc[0][tn], 0, 0, 0);
                    acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], fbn[cur][tn], acc[1][tn], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);  // DS read
            }
        }
        if (DB) {
            if (k0 + BK < k_end) stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    finish_tile<EPI, TM, TN, WN>(g, C, acc, m0, n0, wm, wn, l31, half);
}


}  // namespace gemm
}  // namespace sz

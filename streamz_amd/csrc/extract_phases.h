// The five per-lane phases of the MFCC base-coefficient kernel (framing, 800-point real DFT
// as a 400-point complex FFT, power spectrum, mel filterbank, log, DCT-II) that restate
// reference lib.rs:292-315.  A group of 10 lanes works on one window; the phases talk to each
// other only through the window's scratch region R (LDS on the device, a plain array in the
// CPU lane emulator of the unit tests), so running every lane of a group through phase N
// before any lane starts phase N+1 is exactly what a wavefront does.
//
// Scratch region R, in floats (kRegionFloats per window).  Everything aliases: a lane reads all it
// needs from one layout into registers before any lane writes the next one (on the device the
// LDS operations of a wavefront execute in program order; the emulator copies the region).
//   [0, 400)    ten rows of Y[k1][n2] complex as two planes of 200 floats -- the even columns n2 = 2u (lane u's
//               first DFT) in [0, 200), the odd ones in [200, 400) -- row stride 20 floats, lane u at 2u: rows
//               k1 = 0..9 first, and, once every lane holds its row of those, rows 10..19   (phase 1 -> phase 2)
//   [3, 404)    power spectrum, bin k at pslot(k) = k + 3                   (phase 2 -> phase 3)
//   [0, 330)    mel partial sums, (fall, rise) pairs, lane q at 33 q (an odd stride, so that
//               the ten lanes hit different LDS banks)                      (phase 3 -> phase 4)
//   [332, 358)  log mel energies                                            (phase 4 -> phase 5)
// On the device consecutive windows' regions start kRegionStride = 404 floats apart.  The layout and that
// stride were chosen against LDS bank conflicts, which the counters put at 36 % of the LDS's busy cycles
// with the layout before this one (rows of 40 interleaved floats, bins in runs of 41, regions 420 apart):
// a store instruction of phase 1 now has its 16 lanes (one and a half windows) on 32 different banks,
// 404 = 12 x 32 + 20 continuing where the 20 floats of the window before end; the 16-byte row and bin reads
// of a lane group (lanes of three windows) fall on disjoint quadruples of the 64 banks.
#pragma once
#include <math.h>
#include <stdint.h>

#include "extract_tables.h"
#include "fftmath.h"

namespace sz {

constexpr int kRegionFloats = 404;
constexpr int kRegionStride = 404;
constexpr int kRowFloats = 20;     // one plane's row: ten lanes x one complex value
constexpr int kPlaneFloats = 200;  // ten rows
constexpr int kHalfRows = 10;      // rows of Y exchanged at a time
constexpr int kPartOff = 0;
constexpr int kLogOff = 332;
static_assert(kPartOff + kGroup * kPairStride <= kLogOff && kLogOff + kMels <= kRegionFloats, "region layout");
static_assert(2 * kPlaneFloats <= kRegionFloats && kHalfRows * kRowFloats == kPlaneFloats && pslot(kBins - 1) < kRegionFloats, "region layout");
static_assert(kRegionStride % 4 == 0 && (kPowerOff + 1) % 4 == 0, "16-byte reads of rows and bin runs");

struct alignas(8) pcm4 { int16_t s[4]; };
struct alignas(16) f4 { float v[4]; };

#if defined(__HIP_DEVICE_COMPILE__)
#define SZ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SZ_SCHED_FENCE() ((void)0)
#endif

SZ_HD float lo16(uint32_t w) { return (float)(int16_t)(w & 0xffffu); }
SZ_HD float hi16(uint32_t w) { return (float)(int16_t)(w >> 16); }

// Phase 1: load this lane's samples of the window, two 20-point DFTs over n1, twiddle; rows
// k1 < 10 of Y go to R, rows 10..19 stay in hi_a / hi_b until phase1_high stores them.
// win points at the first sample of the window and must be 8-byte aligned.  The two columns
// are transformed one after the other so that only one 20-point DFT is live in registers.
template <class TT>
SZ_HD void phase1(int u, const int16_t* win, float* R, const TT* T, v2f (&hi_a)[kHalfRows], v2f (&hi_b)[kHalfRows])
{
    uint32_t ra[20], rb[20];
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) {
        const pcm4 q = *reinterpret_cast<const pcm4*>(win + 40 * n1 + 4 * u);
        ra[n1] = (uint32_t)(uint16_t)q.s[0] | ((uint32_t)(uint16_t)q.s[1] << 16);
        rb[n1] = (uint32_t)(uint16_t)q.s[2] | ((uint32_t)(uint16_t)q.s[3] << 16);
    }
    {
        v2f a[20], A[20];
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) a[n1] = mk(lo16(ra[n1]), hi16(ra[n1]));
        dft20(a, A);
#pragma unroll
        for (int k1 = 0; k1 < 20; ++k1) {
            const v2f y = cmul(A[k1], mk(T->twa[k1][u][0], T->twa[k1][u][1]));
            if (k1 < kHalfRows) *reinterpret_cast<v2f*>(R + k1 * kRowFloats + 2 * u) = y;
            else hi_a[k1 - kHalfRows] = y;
        }
    }
    SZ_SCHED_FENCE();
    {
        v2f b[20], B[20];
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) b[n1] = mk(lo16(rb[n1]), hi16(rb[n1]));
        dft20(b, B);
#pragma unroll
        for (int k1 = 0; k1 < 20; ++k1) {
            const v2f y = cmul(B[k1], mk(T->twb[k1][u][0], T->twb[k1][u][1]));
            if (k1 < kHalfRows) *reinterpret_cast<v2f*>(R + kPlaneFloats + k1 * kRowFloats + 2 * u) = y;
            else hi_b[k1 - kHalfRows] = y;
        }
    }
}

// Rows 10..19 of Y replace rows 0..9 once every lane has read its row of those.
SZ_HD void phase1_high(int u, float* R, const v2f (&hi_a)[kHalfRows], const v2f (&hi_b)[kHalfRows])
{
#pragma unroll
    for (int k = 0; k < kHalfRows; ++k) {
        *reinterpret_cast<v2f*>(R + k * kRowFloats + 2 * u) = hi_a[k];
        *reinterpret_cast<v2f*>(R + kPlaneFloats + k * kRowFloats + 2 * u) = hi_b[k];
    }
}

// the row of the first half that lane u transforms in phase 2: row u (rows 0 and 10 for lane 0)
SZ_HD int first_row(int u) { return u; }
// its row of the second half, as an index into the stored rows 10..19
SZ_HD int second_row(int u) { return u == 0 ? 0 : kHalfRows - u; }

// row `row` of Y, columns n2 = 0..19 in order: the even columns from the first plane, the odd ones from the second
SZ_HD void read_row(const float* R, int row, v2f (&x)[20])
{
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const f4 e = *reinterpret_cast<const f4*>(R + row * kRowFloats + 4 * j);
        const f4 o = *reinterpret_cast<const f4*>(R + kPlaneFloats + row * kRowFloats + 4 * j);
        x[4 * j] = mk(e.v[0], e.v[1]);
        x[4 * j + 1] = mk(o.v[0], o.v[1]);
        x[4 * j + 2] = mk(e.v[2], e.v[3]);
        x[4 * j + 3] = mk(o.v[2], o.v[3]);
    }
}

// One split slot: from P = Z[k], Q = Z[400-k] (both already carrying the factor 1/2) to
// |X[k]|^2 and |X[400-k]|^2 with X the 800-point DFT of the real window.
SZ_HD void split_pair(v2f P, v2f Q, v2f w, float& pk, float& pn)
{
    const v2f A = add_conj(P, Q);
    const v2f C = cmul(w, sub_conj(P, Q));
    const v2f X = add_negi(A, C), Xn = sub_negi(A, C);
    pk = X[0] * X[0] + X[1] * X[1];
    pn = Xn[0] * Xn[0] + Xn[1] * Xn[1];
}

// Phase 2: two 20-point DFTs over n2 (x0 = this lane's row of the first half, read before rows
// 10..19 were stored; its row of the second half is read here), real-FFT split, power spectrum
// into R[0..410].
template <class TT>
SZ_HD void phase2(int u, float* R, const TT* T, const v2f (&x0)[20])
{
    const bool z = (u == 0);
    v2f x[20], ZA[20], ZB[20];
    dft20(x0, ZA);
    read_row(R, second_row(u), x);
    dft20(x, ZB);
#pragma unroll
    for (int s = 0; s < kSplitSlots; ++s) {
        v2f P, Q;
        if (s <= 10) {
            P = ZA[s];
            const v2f q0 = ZA[(20 - s) % 20], q1 = ZB[19 - s];
            Q = mk(z ? q0[0] : q1[0], z ? q0[1] : q1[1]);
        } else if (s <= 19) {
            P = mk(z ? ZB[s][0] : ZA[s][0], z ? ZB[s][1] : ZA[s][1]);
            Q = ZB[19 - s];
        } else {
            P = ZB[9];
            Q = ZB[10];
        }
        const v2f w = mk(T->sw[s][u][0], T->sw[s][u][1]);
        float pk, pn;
        split_pair(P, Q, w, pk, pn);
        if (s < 20 || z) {
            R[T->pk[s][u][0]] = pk;
            R[T->pk[s][u][1]] = pn;
        }
    }
}

// Phase 3: this lane's 40 bins of the mel sweep, fully unrolled.  phase3_load takes the lane's 40
// power values into registers; phase3_sweep, which overwrites the power spectrum with the partial
// sums, runs once every lane has done that.  The flush points come from a per-lane bit mask.
SZ_HD void phase3_load(int q, const float* R, float (&p)[kMelSteps])
{
#pragma unroll
    for (int j = 0; j < kMelSteps / 4; ++j) {
        const f4 v = *reinterpret_cast<const f4*>(R + kMelSteps * q + kPowerOff + 1 + 4 * j);  // pslot(40 q + 1 + 4 j ...)
#pragma unroll
        for (int i = 0; i < 4; ++i) p[4 * j + i] = v.v[i];
    }
}

template <class TT>
SZ_HD void phase3_sweep(int q, float* R, const TT* T, const float (&p)[kMelSteps])
{
    const uint64_t flush = T->mel_flush[q];
    float wf[kMelSteps], wr[kMelSteps];
#pragma unroll
    for (int st = 0; st < kMelSteps; ++st) {
        wf[st] = T->melw[st][q][0];
        wr[st] = T->melw[st][q][1];
    }
    float af = 0.0f, ar = 0.0f;
    float* out = R + kPartOff + q * kPairStride;
    if (q == 0) out[kZeroSlot] = 0.0f;  // what the unused recombination entries point at
#pragma unroll
    for (int st = 0; st < kMelSteps; ++st) {
        af += wf[st] * p[st];
        ar += wr[st] * p[st];
        if ((flush >> st) & 1) {
            out[0] = af;
            out[1] = ar;
            out += 2;
            af = 0.0f;
            ar = 0.0f;
        }
    }
}

// Phase 4: recombine the partial sums of filters q, q+10, q+20 and take the clamped log
// (lib.rs:309).  Every filter reads kMaxComb slots; the unused ones hold 0.
template <class TT>
SZ_HD void phase4(int q, float* R, const TT* T)
{
    int pos[3][kMaxComb];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int m = q + 10 * r < kMels ? q + 10 * r : 0;
#pragma unroll
        for (int i = 0; i < kMaxComb; ++i) pos[r][i] = T->comb_pos[m][i];
    }
    float v[3][kMaxComb];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < kMaxComb; ++i) v[r][i] = R[kPartOff + pos[r][i]];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int m = q + 10 * r;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < kMaxComb; ++i) s += v[r][i];
        if (m < kMels) R[kLogOff + m] = logf(fmaxf(s, 1e-12f));
    }
}

// Phase 5: DCT-II coefficients q and q+10 (lib.rs:312-315).
template <class TT>
SZ_HD void phase5(int q, const float* R, const TT* T, float& c_lo, float& c_hi)
{
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int n = 0; n < kMels; ++n) {
        const float e = R[kLogOff + n];
        s0 += e * T->dct2[n][q][0];
        s1 += e * T->dct2[n][q][1];
    }
    c_lo = s0;
    c_hi = s1;
}

}  // namespace sz

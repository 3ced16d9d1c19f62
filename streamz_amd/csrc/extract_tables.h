// Constant tables of the MFCC extractor, built once on the host (FeatureExtractor::new,
// reference lib.rs:239-258) and copied to the device as one blob.
//
// Geometry the tables encode.  One window (800 real samples) is transformed as a 400-point
// complex FFT of z[n] = x[2n] + i*x[2n+1], decomposed 400 = 20 x 20 and worked on by a group
// of 10 lanes ("u" = lane in group):
//   pass 1  lane u owns columns n2 = 2u, 2u+1 and does two 20-point DFTs over n1,
//           then multiplies by w400^(n2*k1) (tables twa/twb, which also carry the input
//           scale 1/32767 of lib.rs:167-169 and the 1/2 of the real-FFT split);
//   pass 2  lane u owns rows k1 = u and 20-u (lane 0: rows 0 and 10), two 20-point DFTs
//           over n2, giving Z[k1 + 20*k2];
//   split   21 slots per lane, each turning a pair (Z[k], Z[400-k]) into X[k], X[400-k]
//           with the twiddle w800^k (table sw) and storing |X|^2 at bins k and 400-k
//           (table kb gives k).
// Mel: lane (window, q) sweeps bins 40q+1 .. 40q+40; each bin feeds the falling edge of one
// triangle and the rising edge of the next (weights melw), partial sums are flushed when the
// bin crosses a triangle centre and recombined per filter through comb_pos.
#pragma once
#include <stdint.h>

namespace sz {

constexpr int kWindow = 800;      // lib.rs:26
constexpr int kHop = 400;         // lib.rs:288
constexpr int kBins = 401;        // lib.rs:299
constexpr int kMels = 26;         // lib.rs:27
constexpr int kMfcc = 20;         // lib.rs:28
constexpr int kFeat = 60;         // lib.rs:30-34
constexpr int kSampleRate = 44100;  // lib.rs:25

constexpr int kGroup = 10;        // lanes per window
constexpr int kWinPerWave = 6;    // windows per wavefront iteration
constexpr int kSplitSlots = 21;
constexpr int kMelSteps = 40;
constexpr int kMaxPairs = 16;     // partial (fall, rise) pairs per mel lane
constexpr int kPairStride = 2 * kMaxPairs + 1;  // floats between two lanes' partial arrays (odd: no bank clash)
constexpr int kMaxComb = 8;
constexpr int kZeroSlot = 2 * kMaxPairs;  // relative to lane 0's partial array: its padding float, kept 0

// Slot of power bin k (0..400) in a window's scratch region: bin 1, where lane 0's run of 40 bins starts, lies
// 16-byte aligned, so that a mel lane fetches its run with ten 16-byte reads (which, with the regions 404 floats
// apart, are free of bank conflicts; so are most of the scattered stores of the bins).
constexpr int kPowerOff = 3;
constexpr int pslot(int k) { return k + kPowerOff; }

struct MelStep {
    float wf;       // weight into the falling-edge accumulator
    float wr;       // weight into the rising-edge accumulator
    int32_t flush;  // non-zero: store both accumulators as the next partial pair, then clear
    int32_t pad;
};

struct ExtractTables {
    float twa[20][kGroup][2];
    float twb[20][kGroup][2];
    float sw[kSplitSlots][kGroup][2];
    int32_t kb[kSplitSlots][kGroup];      // bin k handled by (slot, lane); its partner is 400 - k
    int32_t pk[kSplitSlots][kGroup][2];   // where those two bins live in the region: pslot(k), pslot(400 - k)
    MelStep mel[kMelSteps][kGroup];
    float melw[kMelSteps][kGroup][2];     // (wf, wr) of mel[][], for one 8-byte load per step
    uint64_t mel_flush[kGroup];           // bit st set: mel[st][q].flush
    int32_t comb_n[32];
    // float index into a window's partial array; entries past comb_n point at a slot that holds 0
    alignas(16) int32_t comb_pos[32][kMaxComb];
    float dct2[kMels][kGroup][2];     // (D[q][n], D[q+10][n]),  D[c][n] = cos(pi*(2n+1)*c/52)
    float mel_dense[kMels][kBins];    // the full filterbank, kept for inspection / tests
};

// What the kernel reads of the tables, 13 KB: a workgroup keeps a copy in LDS (the per-lane table reads,
// ~150 per group of windows, then cost an LDS access instead of a trip through the vector L1, which the
// PCM stream keeps evicting).  Member names and shapes equal ExtractTables', so the phase functions take
// either.
struct KernelTables {
    float twa[20][kGroup][2];
    float twb[20][kGroup][2];
    float sw[kSplitSlots][kGroup][2];
    int32_t pk[kSplitSlots][kGroup][2];
    float melw[kMelSteps][kGroup][2];
    uint64_t mel_flush[kGroup];
    alignas(16) int32_t comb_pos[32][kMaxComb];
    float dct2[kMels][kGroup][2];
};
static_assert(sizeof(KernelTables) % 16 == 0, "copied in 16-byte pieces");

// Fills *t.  Returns 0, or a negative number if the filterbank does not have the
// "each bin feeds at most two adjacent triangles" structure the sweep relies on.
int build_extract_tables(ExtractTables* t);

// mel_filter::mel(44100, 800, Some(26), None, None, false, NormalizationFactor::One)
// (reference lib.rs:240-248): Slaney scale, triangles, area normalisation, f32.
// centres receives the 28 band edges in Hz.
void build_mel_filterbank(float (*w)[kBins], float* centres);

}  // namespace sz

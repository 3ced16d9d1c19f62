#include "extract_tables.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace sz {

namespace {
const double kPi = 3.14159265358979323846;

// Slaney mel scale in f32 (the reference instantiates mel::<f32>).
float hz_to_mel(float hz)
{
    const float f_sp = 200.0f / 3.0f;
    const float min_log_hz = 1000.0f;
    const float min_log_mel = min_log_hz / f_sp;
    const float logstep = std::log(6.4f) / 27.0f;
    return hz >= min_log_hz ? min_log_mel + std::log(hz / min_log_hz) / logstep : hz / f_sp;
}
float mel_to_hz(float mel)
{
    const float f_sp = 200.0f / 3.0f;
    const float min_log_hz = 1000.0f;
    const float min_log_mel = min_log_hz / f_sp;
    const float logstep = std::log(6.4f) / 27.0f;
    return mel >= min_log_mel ? min_log_hz * std::exp(logstep * (mel - min_log_mel)) : f_sp * mel;
}
float bin_hz(int k) { return ((float)kSampleRate / 2.0f) * (float)k / (float)(kBins - 1); }

void put(float* dst, double angle_turns, double scale)
{
    // e^{-2*pi*i*angle_turns} * scale
    dst[0] = (float)(std::cos(2.0 * kPi * angle_turns) * scale);
    dst[1] = (float)(-std::sin(2.0 * kPi * angle_turns) * scale);
}
}  // namespace

void build_mel_filterbank(float (*w)[kBins], float* centres)
{
    float edge[kMels + 2];
    const float lo = hz_to_mel(0.0f), hi = hz_to_mel((float)kSampleRate / 2.0f);
    for (int i = 0; i < kMels + 2; ++i)
        edge[i] = mel_to_hz(lo + (hi - lo) * (float)i / (float)(kMels + 1));
    for (int m = 0; m < kMels; ++m) {
        const float rise = edge[m + 1] - edge[m];
        const float fall = edge[m + 2] - edge[m + 1];
        const float area = 2.0f / (edge[m + 2] - edge[m]);
        for (int k = 0; k < kBins; ++k) {
            const float f = bin_hz(k);
            const float up = -(edge[m] - f) / rise;
            const float down = (edge[m + 2] - f) / fall;
            float v = up < down ? up : down;
            if (v < 0.0f) v = 0.0f;
            w[m][k] = v * area;
        }
    }
    if (centres) std::memcpy(centres, edge, sizeof(edge));
}

int build_extract_tables(ExtractTables* t)
{
    std::memset(t, 0, sizeof(*t));
    const double in_scale = 1.0 / (2.0 * 32767.0);

    // pass-1 twiddles
    for (int k1 = 0; k1 < 20; ++k1)
        for (int u = 0; u < kGroup; ++u) {
            put(t->twa[k1][u], (double)((2 * u) * k1 % 400) / 400.0, in_scale);
            put(t->twb[k1][u], (double)((2 * u + 1) * k1 % 400) / 400.0, in_scale);
        }

    // split slots
    for (int s = 0; s < kSplitSlots; ++s)
        for (int u = 0; u < kGroup; ++u) {
            int k;
            if (u == 0) k = s <= 10 ? 20 * s : (s <= 19 ? 10 + 20 * s : 190);
            else k = s <= 19 ? u + 20 * s : u;  // slot 20 is idle for u != 0
            t->kb[s][u] = k;
            t->pk[s][u][0] = pslot(k);
            t->pk[s][u][1] = pslot(400 - k);
            put(t->sw[s][u], (double)k / 800.0, 1.0);
        }

    // mel sweep
    float edge[kMels + 2];
    build_mel_filterbank(t->mel_dense, edge);
    std::vector<int> interval(kBins + 1, -1);
    for (int k = 1; k < kBins; ++k) {
        const float f = bin_hz(k);
        int i = -1;
        for (int e = 0; e < kMels + 2; ++e)
            if (edge[e] < f) i = e;
        if (i > kMels) i = kMels;  // f32 rounding can leave the Nyquist bin just above the last edge
        if (i < 0) return -1;
        interval[k] = i;
        // every weight outside (fall = i-1, rise = i) must be zero
        for (int m = 0; m < kMels; ++m)
            if (m != i && m != i - 1 && t->mel_dense[m][k] != 0.0f) return -2;
    }
    for (int m = 0; m < kMels; ++m)
        if (t->mel_dense[m][0] != 0.0f) return -3;

    std::vector<std::vector<int>> rise_pos(kMels), fall_pos(kMels);
    for (int q = 0; q < kGroup; ++q) {
        int pair = 0;
        for (int st = 0; st < kMelSteps; ++st) {
            const int k = kMelSteps * q + 1 + st;
            const int i = interval[k];
            MelStep& ms = t->mel[st][q];
            ms.wf = i >= 1 ? t->mel_dense[i - 1][k] : 0.0f;
            ms.wr = i < kMels ? t->mel_dense[i][k] : 0.0f;
            const bool last = st == kMelSteps - 1 || interval[k + 1] != i;
            ms.flush = last ? 1 : 0;
            t->melw[st][q][0] = ms.wf;
            t->melw[st][q][1] = ms.wr;
            if (last) t->mel_flush[q] |= 1ull << st;
            if (last) {
                if (pair >= kMaxPairs) return -4;
                const int base = q * kPairStride + pair * 2;
                if (i >= 1) fall_pos[i - 1].push_back(base + 0);
                if (i < kMels) rise_pos[i].push_back(base + 1);
                ++pair;
            }
        }
    }
    for (int m = 0; m < kMels; ++m) {
        int n = 0;
        for (int p : rise_pos[m]) { if (n >= kMaxComb) return -5; t->comb_pos[m][n++] = p; }
        for (int p : fall_pos[m]) { if (n >= kMaxComb) return -5; t->comb_pos[m][n++] = p; }
        t->comb_n[m] = n;
        for (; n < kMaxComb; ++n) t->comb_pos[m][n] = kZeroSlot;
    }

    // DCT-II without the factor 2 (rustdct), first 20 outputs
    for (int n = 0; n < kMels; ++n)
        for (int q = 0; q < kGroup; ++q) {
            t->dct2[n][q][0] = (float)std::cos(kPi * (2 * n + 1) * q / (2.0 * kMels));
            t->dct2[n][q][1] = (float)std::cos(kPi * (2 * n + 1) * (q + 10) / (2.0 * kMels));
        }
    return 0;
}

}  // namespace sz

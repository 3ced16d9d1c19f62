// CPU lane emulator for the extractor's per-lane phases (streamz_amd/csrc/extract_phases.h).
// Runs the ten lanes of one window group phase by phase, as a wavefront would, and compares
// the power spectrum, the log mel energies and the 20 MFCCs with a direct double-precision
// evaluation.  Prints "max_rel_power max_abs_logmel max_abs_mfcc"; exit code 0 on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "extract_phases.h"

using namespace sz;

int main(int argc, char** argv)
{
    static ExtractTables T;
    int rc = build_extract_tables(&T);
    if (rc != 0) { std::printf("table build failed %d\n", rc); return 2; }
    unsigned seed = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1u;
    int kind = argc > 2 ? std::atoi(argv[2]) : 0;
    std::srand(seed);
    alignas(16) int16_t pcm[kWindow];
    for (int i = 0; i < kWindow; ++i) {
        double v;
        if (kind == 0) v = 0.3 * ((std::rand() / (double)RAND_MAX) * 2 - 1);
        else if (kind == 1) v = 0.4 * std::sin(2 * M_PI * 440.0 * i / 44100.0) + 0.2 * std::sin(2 * M_PI * 3520.0 * i / 44100.0);
        else v = (std::rand() & 1) ? 1.0 : -1.0001;
        long q = std::lround(v * 32767.0);
        if (q > 32767) q = 32767;
        if (q < -32768) q = -32768;
        pcm[i] = (int16_t)q;
    }
    // A wavefront runs every lane through one instruction before the next; the region's layouts
    // alias, so each step below lets all ten lanes read before any of them writes.
    alignas(16) float R[kRegionFloats];
    static v2f hi_a[kGroup][kHalfRows], hi_b[kGroup][kHalfRows], x0[kGroup][20];
    for (int u = 0; u < kGroup; ++u) phase1(u, pcm, R, &T, hi_a[u], hi_b[u]);   // rows 0..9 of Y
    for (int u = 0; u < kGroup; ++u) read_row(R, first_row(u), x0[u]);
    for (int u = 0; u < kGroup; ++u) phase1_high(u, R, hi_a[u], hi_b[u]);       // rows 10..19 replace them
    // phase 2 reads its second row before any lane writes the aliased power spectrum: each lane
    // works on a private copy of Y and the writes are merged
    alignas(16) float Y[kRegionFloats];
    for (int i = 0; i < kRegionFloats; ++i) Y[i] = R[i];
    alignas(16) float Pm[kRegionFloats];
    for (int i = 0; i < kRegionFloats; ++i) Pm[i] = NAN;
    for (int u = 0; u < kGroup; ++u) {
        alignas(16) float tmp[kRegionFloats];
        for (int i = 0; i < kRegionFloats; ++i) tmp[i] = Y[i];
        phase2(u, tmp, &T, x0[u]);
        // collect the bins this lane owns from the tables
        for (int s = 0; s < kSplitSlots; ++s) {
            if (s == 20 && u != 0) continue;
            int k = T.kb[s][u];
            Pm[k] = tmp[pslot(k)];
            Pm[400 - k] = tmp[pslot(400 - k)];
        }
    }
    for (int i = 0; i < kBins; ++i) {
        if (std::isnan(Pm[i])) { std::printf("bin %d never written\n", i); return 3; }
    }
    for (int i = 0; i < kBins; ++i) R[pslot(i)] = Pm[i];
    // direct evaluation
    std::vector<double> pw(kBins);
    double pmax = 0;
    for (int k = 0; k < kBins; ++k) {
        double re = 0, im = 0;
        for (int n = 0; n < kWindow; ++n) {
            double x = (double)((float)pcm[n] / 32767.0f);
            double a = -2.0 * M_PI * (double)((long)n * k % kWindow) / kWindow;
            re += x * std::cos(a);
            im += x * std::sin(a);
        }
        pw[k] = re * re + im * im;
        if (pw[k] > pmax) pmax = pw[k];
    }
    double e_pow = 0;
    for (int k = 0; k < kBins; ++k) {
        double e = std::fabs((double)R[pslot(k)] - pw[k]) / pmax;
        if (e > e_pow) e_pow = e;
    }
    static float pw40[kGroup][kMelSteps];
    for (int q = 0; q < kGroup; ++q) phase3_load(q, R, pw40[q]);
    for (int q = 0; q < kGroup; ++q) phase3_sweep(q, R, &T, pw40[q]);   // partial sums overwrite the power spectrum
    for (int q = 0; q < kGroup; ++q) phase4(q, R, &T);
    double e_log = 0, e_mfcc = 0;
    std::vector<double> lm(kMels);
    for (int m = 0; m < kMels; ++m) {
        double s = 0;
        for (int k = 0; k < kBins; ++k) s += (double)T.mel_dense[m][k] * pw[k];
        lm[m] = std::log(s > 1e-12 ? s : 1e-12);
        double e = std::fabs(lm[m] - (double)R[kLogOff + m]);
        if (e > e_log) e_log = e;
    }
    for (int q = 0; q < kGroup; ++q) {
        float c0, c1;
        phase5(q, R, &T, c0, c1);
        for (int r = 0; r < 2; ++r) {
            int c = q + 10 * r;
            double s = 0;
            for (int n = 0; n < kMels; ++n) s += lm[n] * std::cos(M_PI * (2 * n + 1) * c / 52.0);
            double e = std::fabs(s - (double)(r ? c1 : c0));
            if (e > e_mfcc) e_mfcc = e;
        }
    }
    std::printf("%.3e %.3e %.3e\n", e_pow, e_log, e_mfcc);
    return 0;
}

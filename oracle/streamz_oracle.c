/*
 * streamz_oracle.c -- CPU restatement of the StreamZ MFCC + SimpleNeuralNet hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library may call, link or load this
 * file; it is the checker that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg compare the HIP path against.
 *
 * PARITY UNPINNED: the reference (Rust) cannot be built here and its own tests hold no
 * numeric vector for this path, so this restatement is pinned only by an independent
 * numpy/scipy formulation (tests/golden/make_golden.py), not by reference outputs.
 * The three third-party pieces whose source is not in the reference tree are restated
 * from their published definitions:
 *   - mel_filter 0.1.1  mel(sr, n_fft, n_mels, fmin, fmax, htk=false, norm=One):
 *     librosa-style Slaney filterbank with area ("norm=1") normalisation;
 *   - rustfft 6.4.0     unnormalised forward DFT;
 *   - rustdct 0.7.1     unnormalised DCT-II without the factor 2.
 *
 * Every function cites the reference lines it follows (paths under
 * /root/reference/streamz-rs/src/).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off so that a*b+c stays two roundings,
 * as in the Rust code, which never contracts to fma).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SO_WINDOW 800      /* lib.rs:26 WINDOW_SIZE */
#define SO_HOP 400         /* lib.rs:288 step = WINDOW_SIZE / 2 */
#define SO_NBINS 401       /* lib.rs:299 WINDOW_SIZE / 2 + 1 */
#define SO_NMELS 26        /* lib.rs:27 */
#define SO_MFCC 20         /* lib.rs:28 */
#define SO_FEAT 60         /* lib.rs:30-34 */
#define SO_SR 44100        /* lib.rs:25 */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------ */
/* window count: lib.rs:288-291, 317 (integer arithmetic only)                          */
uint64_t so_window_count(uint64_t n_samples)
{
    if (n_samples < SO_WINDOW) return 0;
    return (n_samples - SO_WINDOW) / SO_HOP + 1;
}

/* ------------------------------------------------------------------------------------ */
/* Slaney mel scale (librosa hz_to_mel / mel_to_hz with htk=false), evaluated in f32     */
/* because the reference instantiates mel::<f32> (lib.rs:240-248).                       */
static float hz_to_mel_f(float hz)
{
    const float f_sp = 200.0f / 3.0f;
    const float min_log_hz = 1000.0f;
    const float min_log_mel = min_log_hz / f_sp;
    const float logstep = logf(6.4f) / 27.0f;
    if (hz >= min_log_hz) return min_log_mel + logf(hz / min_log_hz) / logstep;
    return hz / f_sp;
}
static float mel_to_hz_f(float mel)
{
    const float f_sp = 200.0f / 3.0f;
    const float min_log_hz = 1000.0f;
    const float min_log_mel = min_log_hz / f_sp;
    const float logstep = logf(6.4f) / 27.0f;
    if (mel >= min_log_mel) return min_log_hz * expf(logstep * (mel - min_log_mel));
    return f_sp * mel;
}

/* mel filterbank [26][401]: lib.rs:240-248 -> mel_filter::mel(44100, 800, Some(26), None,
 * None, false, NormalizationFactor::One).  fmin = 0, fmax = sr/2, triangles
 * max(0, min(lower, upper)) over bin centre frequencies k*sr/n_fft, each row scaled by
 * 2 / (f[m+2] - f[m]). */
void so_mel_filterbank(float *w /* [26*401] */)
{
    float mel_f[SO_NMELS + 2];
    float fftfreqs[SO_NBINS];
    const float min_mel = hz_to_mel_f(0.0f);
    const float max_mel = hz_to_mel_f((float)SO_SR / 2.0f);
    for (int i = 0; i < SO_NMELS + 2; ++i) {
        float m = min_mel + (max_mel - min_mel) * (float)i / (float)(SO_NMELS + 1);
        mel_f[i] = mel_to_hz_f(m);
    }
    for (int k = 0; k < SO_NBINS; ++k)
        fftfreqs[k] = ((float)SO_SR / 2.0f) * (float)k / (float)(SO_NBINS - 1);
    for (int m = 0; m < SO_NMELS; ++m) {
        const float fd0 = mel_f[m + 1] - mel_f[m];
        const float fd1 = mel_f[m + 2] - mel_f[m + 1];
        const float enorm = 2.0f / (mel_f[m + 2] - mel_f[m]);
        for (int k = 0; k < SO_NBINS; ++k) {
            float lower = -(mel_f[m] - fftfreqs[k]) / fd0;
            float upper = (mel_f[m + 2] - fftfreqs[k]) / fd1;
            float v = lower < upper ? lower : upper;
            if (v < 0.0f) v = 0.0f;
            w[m * SO_NBINS + k] = v * enorm;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* forward DFT of one 800-sample frame -> power of bins 0..400 (lib.rs:292-301)          */

/* mode 0: direct evaluation of the definition with double accumulation (gold).         */
static void power_spectrum_f64(const float *x, float *p)
{
    static double ct[SO_WINDOW], st[SO_WINDOW];
    static int init = 0;
    if (!init) {
        for (int i = 0; i < SO_WINDOW; ++i) {
            ct[i] = cos(2.0 * M_PI * (double)i / SO_WINDOW);
            st[i] = sin(2.0 * M_PI * (double)i / SO_WINDOW);
        }
        init = 1;
    }
    for (int k = 0; k < SO_NBINS; ++k) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < SO_WINDOW; ++n) {
            re += (double)x[n] * ct[idx];
            im -= (double)x[n] * st[idx];
            idx += k;
            if (idx >= SO_WINDOW) idx -= SO_WINDOW;
        }
        /* the reference holds the spectrum in Complex<f32> and squares in f32 */
        float fr = (float)re, fi = (float)im;
        p[k] = fr * fr + fi * fi;
    }
}

/* mode 1: all-f32 mixed-radix (2,2,2,2,2,5,5) decimation-in-time FFT on 800 complex
 * points, the arithmetic class of rustfft's planner output for this length. */
typedef struct { float re, im; } cf32;

static cf32 g_tw800[SO_WINDOW];
static int g_tw_init = 0;
static void fft_rec(const cf32 *in, int stride, int n, cf32 *out)
{
    if (n == 1) { out[0] = in[0]; return; }
    int r = (n % 2 == 0) ? 2 : 5;
    int m = n / r;
    cf32 tmp[SO_WINDOW];
    for (int q = 0; q < r; ++q) fft_rec(in + q * stride, stride * r, m, tmp + q * m);
    const int tws = SO_WINDOW / n; /* twiddle stride into the 800-table */
    for (int k = 0; k < m; ++k) {
        cf32 t[5];
        for (int q = 0; q < r; ++q) {
            cf32 w = g_tw800[(q * k * tws) % SO_WINDOW];
            cf32 v = tmp[q * m + k];
            t[q].re = v.re * w.re - v.im * w.im;
            t[q].im = v.re * w.im + v.im * w.re;
        }
        for (int p = 0; p < r; ++p) {
            float sr = 0.0f, si = 0.0f;
            for (int q = 0; q < r; ++q) {
                cf32 w = g_tw800[((p * q) % r) * (SO_WINDOW / r)];
                sr += t[q].re * w.re - t[q].im * w.im;
                si += t[q].re * w.im + t[q].im * w.re;
            }
            out[k + p * m].re = sr;
            out[k + p * m].im = si;
        }
    }
}
static void power_spectrum_f32(const float *x, float *p)
{
    if (!g_tw_init) {
        for (int i = 0; i < SO_WINDOW; ++i) {
            g_tw800[i].re = (float)cos(2.0 * M_PI * (double)i / SO_WINDOW);
            g_tw800[i].im = (float)-sin(2.0 * M_PI * (double)i / SO_WINDOW);
        }
        g_tw_init = 1;
    }
    cf32 in[SO_WINDOW], out[SO_WINDOW];
    for (int i = 0; i < SO_WINDOW; ++i) { in[i].re = x[i]; in[i].im = 0.0f; }
    fft_rec(in, 1, SO_WINDOW, out);
    for (int k = 0; k < SO_NBINS; ++k) p[k] = out[k].re * out[k].re + out[k].im * out[k].im;
}

/* ------------------------------------------------------------------------------------ */
typedef struct {
    float mel[SO_NMELS * SO_NBINS];
    float dct[SO_MFCC * SO_NMELS]; /* cos(pi*(2n+1)*k/52), k<20 */
} so_extractor;

/* FeatureExtractor::new, lib.rs:239-258 */
so_extractor *so_extractor_create(void)
{
    so_extractor *e = (so_extractor *)malloc(sizeof(so_extractor));
    if (!e) return NULL;
    so_mel_filterbank(e->mel);
    for (int k = 0; k < SO_MFCC; ++k)
        for (int n = 0; n < SO_NMELS; ++n)
            e->dct[k * SO_NMELS + n] =
                (float)cos(M_PI * (double)(2 * n + 1) * (double)k / (2.0 * SO_NMELS));
    /* make the lazily built tables exist before any threaded use */
    float z[SO_WINDOW] = {0}, p[SO_NBINS];
    power_spectrum_f64(z, p);
    power_spectrum_f32(z, p);
    return e;
}
void so_extractor_destroy(so_extractor *e) { free(e); }
const float *so_extractor_mel(const so_extractor *e) { return e->mel; }

/* base MFCCs of one frame: lib.rs:292-315 */
static void frame_mfcc(const so_extractor *e, const int16_t *chunk, int fft_mode, float *coef)
{
    float x[SO_WINDOW], mags[SO_NBINS], en[SO_NMELS];
    for (int i = 0; i < SO_WINDOW; ++i) x[i] = (float)chunk[i] / 32767.0f; /* lib.rs:167-169 */
    if (fft_mode == 0) power_spectrum_f64(x, mags); else power_spectrum_f32(x, mags);
    for (int m = 0; m < SO_NMELS; ++m) {            /* lib.rs:303-310 */
        float sum = 0.0f;
        const float *w = e->mel + m * SO_NBINS;
        for (int j = 0; j < SO_NBINS; ++j) sum += w[j] * mags[j];
        en[m] = logf(sum > 1e-12f ? sum : 1e-12f);
    }
    for (int k = 0; k < SO_MFCC; ++k) {             /* lib.rs:312-315 */
        float s = 0.0f;
        for (int n = 0; n < SO_NMELS; ++n) s += en[n] * e->dct[k * SO_NMELS + n];
        coef[k] = s;
    }
}

/* add_deltas, lib.rs:212-228 */
static void add_deltas(const float *x, uint64_t n, float *d)
{
    for (uint64_t i = 0; i < n; ++i) {
        const float *prev = x + (i > 0 ? i - 1 : i) * SO_MFCC;
        const float *next = x + (i + 1 < n ? i + 1 : i) * SO_MFCC;
        for (int j = 0; j < SO_MFCC; ++j) d[i * SO_MFCC + j] = (next[j] - prev[j]) / 2.0f;
    }
}

/* Deltas, concat and per-frame z-score from base coefficients: lib.rs:321-342 */
void so_finish_features(const float *base, uint64_t n, float *out /* [n*60] */)
{
    if (n == 0) return;
    float *d1 = (float *)malloc(sizeof(float) * SO_MFCC * n);
    float *d2 = (float *)malloc(sizeof(float) * SO_MFCC * n);
    add_deltas(base, n, d1);
    add_deltas(d1, n, d2);
    for (uint64_t i = 0; i < n; ++i) {
        float *f = out + i * SO_FEAT;
        memcpy(f, base + i * SO_MFCC, sizeof(float) * SO_MFCC);
        memcpy(f + SO_MFCC, d1 + i * SO_MFCC, sizeof(float) * SO_MFCC);
        memcpy(f + 2 * SO_MFCC, d2 + i * SO_MFCC, sizeof(float) * SO_MFCC);
        float sum = 0.0f;
        for (int j = 0; j < SO_FEAT; ++j) sum += f[j];
        const float mean = sum / (float)SO_FEAT;
        float vs = 0.0f;
        for (int j = 0; j < SO_FEAT; ++j) { float dd = f[j] - mean; vs += dd * dd; }
        const float var = vs / (float)SO_FEAT;
        float sd = sqrtf(var);
        if (!(sd > 1e-6f)) sd = 1e-6f; /* f32::max(1e-6): NaN -> 1e-6 */
        for (int j = 0; j < SO_FEAT; ++j) f[j] = (f[j] - mean) / sd;
    }
    free(d1);
    free(d2);
}

/* FeatureExtractor::extract -> window_samples_with_plan, lib.rs:261-263, 279-345.
 * Returns the number of windows; writes n*60 floats.  base_out (optional) receives the
 * un-normalised 20 MFCCs per frame. */
uint64_t so_extract(const so_extractor *e, const int16_t *pcm, uint64_t n_samples, int fft_mode,
                    float *out, float *base_out)
{
    const uint64_t n = so_window_count(n_samples);
    if (n == 0) return 0;
    float *base = base_out ? base_out : (float *)malloc(sizeof(float) * SO_MFCC * n);
    for (uint64_t w = 0; w < n; ++w) frame_mfcc(e, pcm + w * SO_HOP, fft_mode, base + w * SO_MFCC);
    so_finish_features(base, n, out);
    if (!base_out) free(base);
    return n;
}

/* The file-parallel loop of main.rs:500-508 (rayon par_iter over files): OpenMP over files. */
void so_extract_files(const so_extractor *e, const int16_t *pcm, const uint64_t *offsets,
                      const uint64_t *lengths, uint64_t n_files, int fft_mode, float *out,
                      const uint64_t *out_offsets /* in windows */)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t f = 0; f < (int64_t)n_files; ++f)
        so_extract(e, pcm + offsets[f], lengths[f], fft_mode, out + out_offsets[f] * SO_FEAT, NULL);
}

/* ------------------------------------------------------------------------------------ */
/* SimpleNeuralNet, lib.rs:744-790.  Row-major: w1[in][h1], w2[h1][h2], w3[h2][out].     */
typedef struct {
    int in, h1, h2, out;
    float *w1, *b1, *w2, *b2, *w3, *b3;
} so_net;

so_net *so_net_create(int in, int h1, int h2, int out)
{
    so_net *n = (so_net *)calloc(1, sizeof(so_net));
    n->in = in; n->h1 = h1; n->h2 = h2; n->out = out;
    n->w1 = (float *)calloc((size_t)in * h1, sizeof(float));
    n->b1 = (float *)calloc((size_t)h1, sizeof(float));
    n->w2 = (float *)calloc((size_t)h1 * h2, sizeof(float));
    n->b2 = (float *)calloc((size_t)h2, sizeof(float));
    n->w3 = (float *)calloc((size_t)h2 * out, sizeof(float));
    n->b3 = (float *)calloc((size_t)out, sizeof(float));
    return n;
}
void so_net_destroy(so_net *n)
{
    if (!n) return;
    free(n->w1); free(n->b1); free(n->w2); free(n->b2); free(n->w3); free(n->b3); free(n);
}
float *so_net_param(so_net *n, int which)
{
    switch (which) {
    case 0: return n->w1; case 1: return n->b1; case 2: return n->w2;
    case 3: return n->b2; case 4: return n->w3; case 5: return n->b3;
    }
    return NULL;
}

/* hidden activations and softmax of one sample: lib.rs:880-891 / 1016-1026 */
static void fwd_sample(const so_net *n, const float *x, float *h1pre, float *h1, float *h2, float *o)
{
    for (int j = 0; j < n->h1; ++j) {
        float s = 0.0f;
        for (int i = 0; i < n->in; ++i) s += x[i] * n->w1[(size_t)i * n->h1 + j];
        s += n->b1[j];
        if (h1pre) h1pre[j] = s;
        h1[j] = s > 0.0f ? s : 0.0f;
    }
    for (int j = 0; j < n->h2; ++j) {
        float s = 0.0f;
        for (int i = 0; i < n->h1; ++i) s += h1[i] * n->w2[(size_t)i * n->h2 + j];
        h2[j] = tanhf(s + n->b2[j]);
    }
    if (!o) return;
    float mx = -INFINITY;
    for (int j = 0; j < n->out; ++j) {
        float s = 0.0f;
        for (int i = 0; i < n->h2; ++i) s += h2[i] * n->w3[(size_t)i * n->out + j];
        o[j] = s + n->b3[j];
        if (o[j] > mx) mx = o[j];
    }
    float sum = 0.0f;
    for (int j = 0; j < n->out; ++j) { o[j] = expf(o[j] - mx); sum += o[j]; }
    for (int j = 0; j < n->out; ++j) o[j] = o[j] / sum;
}

/* SimpleNeuralNet::forward, lib.rs:880-891: probabilities [B][out] */
void so_net_forward(const so_net *n, const float *x, uint64_t B, float *probs)
{
    float *h1 = (float *)malloc(sizeof(float) * n->h1);
    float *h2 = (float *)malloc(sizeof(float) * n->h2);
    for (uint64_t b = 0; b < B; ++b)
        fwd_sample(n, x + b * n->in, NULL, h1, h2, probs + b * n->out);
    free(h1); free(h2);
}

/* SimpleNeuralNet::embed, lib.rs:895-900: tanh hidden layer [B][h2] */
void so_net_embed(const so_net *n, const float *x, uint64_t B, float *emb)
{
    float *h1 = (float *)malloc(sizeof(float) * n->h1);
    for (uint64_t b = 0; b < B; ++b) fwd_sample(n, x + b * n->in, NULL, h1, emb + b * n->h2, NULL);
    free(h1);
}

/* SimpleNeuralNet::forward_embedding, lib.rs:1073-1079: ReLU on both hidden layers */
void so_net_forward_embedding(const so_net *n, const float *x, uint64_t B, float *emb)
{
    float *h1 = (float *)malloc(sizeof(float) * n->h1);
    for (uint64_t b = 0; b < B; ++b) {
        const float *xb = x + b * n->in;
        for (int j = 0; j < n->h1; ++j) {
            float s = 0.0f;
            for (int i = 0; i < n->in; ++i) s += n->w1[(size_t)i * n->h1 + j] * xb[i];
            s += n->b1[j];
            h1[j] = s > 0.0f ? s : 0.0f;
        }
        for (int j = 0; j < n->h2; ++j) {
            float s = 0.0f;
            for (int i = 0; i < n->h1; ++i) s += n->w2[(size_t)i * n->h2 + j] * h1[i];
            s += n->b2[j];
            emb[b * n->h2 + j] = s > 0.0f ? s : 0.0f;
        }
    }
    free(h1);
}

/* Gradient sums of a batch (weights frozen): lib.rs:1006-1045.  target is one shared
 * one-hot (or any) vector; labels, when non-NULL, gives a per-row class instead (used by
 * the multi-file accumulate mode).  g must be zeroed by the caller; layout
 * [w1|b1|w2|b2|w3|b3]. */
void so_net_grad(const so_net *n, const float *x, uint64_t B, const float *target,
                 const int32_t *labels, float *g)
{
    float *gw1 = g, *gb1 = gw1 + (size_t)n->in * n->h1, *gw2 = gb1 + n->h1;
    float *gb2 = gw2 + (size_t)n->h1 * n->h2, *gw3 = gb2 + n->h2, *gb3 = gw3 + (size_t)n->h2 * n->out;
    float *h1pre = (float *)malloc(sizeof(float) * n->h1);
    float *h1 = (float *)malloc(sizeof(float) * n->h1);
    float *h2 = (float *)malloc(sizeof(float) * n->h2);
    float *o = (float *)malloc(sizeof(float) * n->out);
    float *d2 = (float *)malloc(sizeof(float) * n->h2);
    float *d1 = (float *)malloc(sizeof(float) * n->h1);
    for (uint64_t b = 0; b < B; ++b) {
        const float *xb = x + b * n->in;
        fwd_sample(n, xb, h1pre, h1, h2, o);
        for (int j = 0; j < n->out; ++j) {          /* delta_out = out - t, lib.rs:1028 */
            float t = labels ? (labels[b] == j ? 1.0f : 0.0f) : target[j];
            o[j] = o[j] - t;
        }
        for (int i = 0; i < n->h2; ++i)             /* lib.rs:1029-1033 */
            for (int j = 0; j < n->out; ++j) gw3[(size_t)i * n->out + j] += h2[i] * o[j];
        for (int j = 0; j < n->out; ++j) gb3[j] += o[j];
        for (int i = 0; i < n->h2; ++i) {           /* lib.rs:1034 */
            float s = 0.0f;
            for (int j = 0; j < n->out; ++j) s += o[j] * n->w3[(size_t)i * n->out + j];
            d2[i] = s * (1.0f - h2[i] * h2[i]);
        }
        for (int i = 0; i < n->h1; ++i)             /* lib.rs:1035-1038 */
            for (int j = 0; j < n->h2; ++j) gw2[(size_t)i * n->h2 + j] += h1[i] * d2[j];
        for (int j = 0; j < n->h2; ++j) gb2[j] += d2[j];
        for (int i = 0; i < n->h1; ++i) {           /* lib.rs:1039-1040 */
            float s = 0.0f;
            for (int j = 0; j < n->h2; ++j) s += d2[j] * n->w2[(size_t)i * n->h2 + j];
            d1[i] = s * (h1pre[i] > 0.0f ? 1.0f : 0.0f);
        }
        for (int i = 0; i < n->in; ++i)             /* lib.rs:1041-1044 */
            for (int j = 0; j < n->h1; ++j) gw1[(size_t)i * n->h1 + j] += xb[i] * d1[j];
        for (int j = 0; j < n->h1; ++j) gb1[j] += d1[j];
    }
    free(h1pre); free(h1); free(h2); free(o); free(d2); free(d1);
}

size_t so_net_param_count(const so_net *n)
{
    return (size_t)n->in * n->h1 + n->h1 + (size_t)n->h1 * n->h2 + n->h2 + (size_t)n->h2 * n->out + n->out;
}

/* theta -= scale * g : lib.rs:1047-1059 */
void so_net_apply(so_net *n, const float *g, float scale)
{
    float *p[6] = {n->w1, n->b1, n->w2, n->b2, n->w3, n->b3};
    size_t cnt[6] = {(size_t)n->in * n->h1, (size_t)n->h1, (size_t)n->h1 * n->h2,
                     (size_t)n->h2, (size_t)n->h2 * n->out, (size_t)n->out};
    for (int t = 0; t < 6; ++t) {
        for (size_t i = 0; i < cnt[t]; ++i) p[t][i] -= g[i] * scale;
        g += cnt[t];
    }
}

/* SimpleNeuralNet::train_batch, lib.rs:1002-1060 */
void so_net_train_batch(so_net *n, const float *x, uint64_t B, const float *target, float lr)
{
    if (B == 0) return; /* lib.rs:1003-1005 */
    const size_t np = so_net_param_count(n);
    float *g = (float *)calloc(np, sizeof(float));
    so_net_grad(n, x, B, target, NULL, g);
    so_net_apply(n, g, lr / (float)B);
    free(g);
}

/* SimpleNeuralNet::train, lib.rs:954-999: the B = 1 case with scale = lr */
void so_net_train(so_net *n, const float *x, const float *target, float lr)
{
    so_net_train_batch(n, x, 1, target, lr);
}

/* ------------------------------------------------------------------------------------ */
/* xorshift generator used only when the caller injects no shuffle / dropout stream.    */
static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint32_t rng_u32(void)
{
    g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
    return (uint32_t)(g_rng >> 32);
}
void so_seed(uint64_t s) { g_rng = s ? s : 0x9E3779B97F4A7C15ull; }

/* pretrain_from_features, lib.rs:582-628.
 * perm : NULL, or [epochs][n_win] -- row order of each epoch after the shuffle (lib.rs:601).
 * keep : NULL, or [epochs][n_win][60] in shuffled order -- 0 means the element is zeroed
 *        by apply_dropout (lib.rs:119-129), non-zero means kept.
 * Returns the mean loss over all counted windows (0.0 if none). */
float so_pretrain_from_features(so_net *net, const float *windows, uint64_t n_win, uint32_t target_class,
                                uint32_t num_classes, uint32_t epochs, float lr, float dropout,
                                uint32_t batch_size, const uint32_t *perm, const uint8_t *keep)
{
    float *target = (float *)calloc(num_classes ? num_classes : 1, sizeof(float));
    if (target_class < num_classes) target[target_class] = 1.0f;
    const uint32_t bs = batch_size > 0 ? batch_size : 1; /* batch_size.max(1) */
    float total_loss = 0.0f;
    uint64_t count = 0;
    uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * (n_win ? n_win : 1));
    float *batch = (float *)malloc(sizeof(float) * SO_FEAT * bs);
    float *out = (float *)malloc(sizeof(float) * (net->out ? net->out : 1));
    for (uint32_t ep = 0; ep < epochs; ++ep) {
        if (perm) {
            memcpy(order, perm + (size_t)ep * n_win, sizeof(uint32_t) * n_win);
        } else {
            for (uint64_t i = 0; i < n_win; ++i) order[i] = (uint32_t)i;
            for (uint64_t i = n_win; i > 1; --i) { /* Fisher-Yates */
                uint64_t j = rng_u32() % i;
                uint32_t t = order[i - 1]; order[i - 1] = order[j]; order[j] = t;
            }
        }
        for (uint64_t c0 = 0; c0 < n_win; c0 += bs) {
            const uint64_t c1 = c0 + bs < n_win ? c0 + bs : n_win;
            uint64_t live = 0;
            for (uint64_t r = c0; r < c1; ++r) {
                float *f = batch + live * SO_FEAT;
                memcpy(f, windows + (size_t)order[r] * SO_FEAT, sizeof(float) * SO_FEAT);
                if (dropout > 0.0f) { /* apply_dropout, lib.rs:119-129 */
                    for (int j = 0; j < SO_FEAT; ++j) {
                        int drop;
                        if (keep) drop = keep[((size_t)ep * n_win + r) * SO_FEAT + j] == 0;
                        else drop = ((float)(rng_u32() >> 8) * (1.0f / 16777216.0f)) < dropout;
                        if (drop) f[j] = 0.0f;
                    }
                }
                int allzero = 1;                   /* lib.rs:607-609 */
                for (int j = 0; j < SO_FEAT; ++j) if (f[j] != 0.0f) { allzero = 0; break; }
                if (allzero) continue;
                so_net_forward(net, f, 1, out);    /* lib.rs:610-617 */
                float loss = 0.0f;
                for (uint32_t j = 0; j < num_classes && j < (uint32_t)net->out; ++j) {
                    float o = out[j] > 1e-12f ? out[j] : 1e-12f;
                    loss += target[j] * logf(o);
                }
                total_loss += -loss;
                count += 1;
                live += 1;
            }
            so_net_train_batch(net, batch, live, target, lr); /* lib.rs:620 */
        }
    }
    free(target); free(order); free(batch); free(out);
    return count > 0 ? total_loss / (float)count : 0.0f;
}

/* augment, lib.rs:103-116, with the random draws passed in: out[i] = clamp(in[(i + shift) % n] *
 * gain + noise[i] * 32767) truncated to i16 (Rust's `as i16` on an in-range float truncates). */
void so_augment(const int16_t *in, uint64_t n, float gain, uint64_t shift, const float *noise, int16_t *out)
{
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t idx = (i + shift) % n;
        float val = (float)in[idx] * gain + noise[i] * 32767.0f;
        if (val < -32768.0f) val = -32768.0f;
        if (val > 32767.0f) val = 32767.0f;
        out[i] = (int16_t)(int)val;
    }
}

/* identify_speaker_list, lib.rs:1383-1411.  ids receives the speaker indices with at
 * least one vote, ordered by vote count descending (stable, so ties keep ascending id).
 * The per-window argmax keeps the LAST maximum (Iterator::max_by). */
uint32_t so_identify_from_features(const so_net *net, const float *feats, uint64_t n_win, float threshold,
                                   uint32_t *ids, uint64_t *counts_out)
{
    const int C = net->out;
    uint64_t *counts = (uint64_t *)calloc(C ? C : 1, sizeof(uint64_t));
    float *out = (float *)malloc(sizeof(float) * (C ? C : 1));
    for (uint64_t w = 0; w < n_win; ++w) {
        so_net_forward(net, feats + w * SO_FEAT, 1, out);
        if (C == 0) continue;
        int best = 0;
        for (int j = 1; j < C; ++j) if (out[j] >= out[best]) best = j; /* last max wins */
        if (out[best] >= threshold) counts[best] += 1;
    }
    uint32_t n_ids = 0;
    for (int j = 0; j < C; ++j) if (counts[j] > 0) ids[n_ids++] = (uint32_t)j;
    for (uint32_t i = 1; i < n_ids; ++i) { /* stable insertion sort by count desc */
        uint32_t v = ids[i];
        uint32_t k = i;
        while (k > 0 && counts[ids[k - 1]] < counts[v]) { ids[k] = ids[k - 1]; --k; }
        ids[k] = v;
    }
    if (counts_out) memcpy(counts_out, counts, sizeof(uint64_t) * C);
    free(counts); free(out);
    return n_ids;
}

uint32_t so_identify_speaker_list(const so_net *net, const so_extractor *e, const int16_t *pcm,
                                  uint64_t n_samples, float threshold, int fft_mode, uint32_t *ids)
{
    const uint64_t n = so_window_count(n_samples);
    float *feats = (float *)malloc(sizeof(float) * SO_FEAT * (n ? n : 1));
    so_extract(e, pcm, n_samples, fft_mode, feats, NULL);
    uint32_t r = so_identify_from_features(net, feats, n, threshold, ids, NULL);
    free(feats);
    return r;
}

/* identify_speaker, lib.rs:1285-1303, on feature rows: per-class sums of the window
 * probabilities (f32, in window order), arg-max with the LAST maximum winning, 0 when the
 * net has no outputs.  sums_out (optional) receives the sums. */
uint32_t so_identify_speaker_feats(const so_net *net, const float *feats, uint64_t n_win, float *sums_out)
{
    const int C = net->out;
    float *sums = (float *)calloc(C ? C : 1, sizeof(float));
    float *out = (float *)malloc(sizeof(float) * (C ? C : 1));
    for (uint64_t w = 0; w < n_win; ++w) {
        so_net_forward(net, feats + w * SO_FEAT, 1, out);
        for (int j = 0; j < C; ++j) sums[j] += out[j];
    }
    int best = 0;
    for (int j = 1; j < C; ++j) if (sums[j] >= sums[best]) best = j;
    if (sums_out) memcpy(sums_out, sums, sizeof(float) * C);
    free(sums); free(out);
    return (uint32_t)best;
}

/* identify_speaker_with_threshold_feats, lib.rs:1346-1377 (and 1307-1343 after extraction):
 * -1 (None) for a net with at most one output or no windows; otherwise the arg-max of the
 * sums if sum / count >= threshold, else -1. */
int64_t so_identify_speaker_with_threshold_feats(const so_net *net, const float *feats, uint64_t n_win, float threshold)
{
    const int C = net->out;
    if (C <= 1) return -1;
    float *sums = (float *)calloc(C, sizeof(float));
    const uint32_t best = so_identify_speaker_feats(net, feats, n_win, sums);
    float count = 0.0f;
    for (uint64_t w = 0; w < n_win; ++w) count += 1.0f;
    int64_t r = -1;
    if (count != 0.0f && sums[best] / count >= threshold) r = (int64_t)best;
    free(sums);
    return r;
}

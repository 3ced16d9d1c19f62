// Internal declarations shared by the translation units of libstreamz_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "extract_tables.h"

namespace sz {

// ---- error plumbing ------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);  // records the message, returns SZH_ENODEVICE
#define SZ_HIP(expr)                                          \
    do {                                                      \
        hipError_t e__ = (expr);                              \
        if (e__ != hipSuccess) return ::sz::hip_fail(e__, #expr); \
    } while (0)

// Raises a kernel's dynamic-LDS limit to `bytes`, once per (kernel, device); callable from any
// thread.  Kernels that ask for more than 64 KB at launch need it.
int allow_dynamic_lds(const void* kernel, size_t bytes);

// ---- extraction ----------------------------------------------------------------------
// One unit of kernel work: up to kSegWindows consecutive windows of one file.
constexpr uint32_t kSegWindows = 48;
struct Segment {
    uint64_t pcm_off;         // first sample of the segment's first window in the PCM buffer
    uint64_t win_first;       // global index of the segment's first window
    uint64_t file_win_first;  // global index of the owning file's first window
    uint64_t out_first;       // finish kernel: output row of the segment's first window
    uint32_t n_win;           // windows in this segment (1..kSegWindows)
    uint32_t file_n_win;      // windows in the owning file
};

struct Extractor {
    ExtractTables host;
    KernelTables* dev = nullptr;  // the part of `host` the kernel reads
    int device = 0;
};

struct Plan {
    Extractor* ext = nullptr;
    std::vector<uint64_t> win_off;  // n_files + 1 prefix sums
    uint64_t total_windows = 0;
    uint64_t pcm_extent = 0;        // samples the plan may read: max(offset + length)
    uint32_t n_segs = 0;
    Segment* d_segs = nullptr;      // windows whose MFCCs are computed
    uint32_t n_fin = 0;
    Segment* d_fin = nullptr;       // windows that are finished and written (== d_segs unless a tile)
    float* d_base = nullptr;        // [base windows][20] un-normalised MFCCs
    // a plan that walks along a recording: samples added to every unit's position (device word, moved on
    // by `advance` at the end of every run), see plan_set_advance
    uint64_t* d_shift = nullptr;
    uint64_t advance = 0;
};

int extractor_create(Extractor** out);
void extractor_destroy(Extractor* e);
int plan_create(Extractor* e, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_files, Plan** out);
void plan_destroy(Plan* p);
int plan_run(Plan* p, const int16_t* pcm_dev, float* feats_dev, hipStream_t stream);
// A tile of a longer recording: MFCCs of lead + n_out + trail consecutive windows are computed
// (the first one starts at sample pcm_off) and rows lead .. lead + n_out - 1 are finished into
// feats rows 0 .. n_out - 1.  lead / trail are the neighbours the deltas need (2, or 0 at the
// recording's own ends, where add_deltas replicates the edge, lib.rs:215-220).
int plan_create_tile(Extractor* e, uint64_t pcm_off, uint32_t lead, uint32_t n_out, uint32_t trail, Plan** out);
// from now on every plan_run leaves the plan `samples` further into the PCM buffer for the next one (the
// move is part of the run's last kernel: no extra launch, and a captured run replays it)
int plan_set_advance(Plan* p, uint64_t samples);
// augment (lib.rs:103-116) of n device-resident samples; noise_dev NULL -> counter-based draws
// keyed by (seed, counter0 + i).
// looks up the current device's copy of the fixed array szh_extract_dev_noalloc works in
// (extract.hip); szh_init calls it so that the entry point's first call does no set-up
int noalloc_setup();
int augment_dev(const int16_t* in, uint64_t n, float gain, float noise_level, uint64_t shift, const float* noise_dev,
                uint64_t seed, uint64_t counter0, int16_t* out, hipStream_t stream);

// ---- network -------------------------------------------------------------------------
struct Workspace {
    uint64_t rows = 0;  // capacity in rows
    // first row the forward / vote path works in: callers that keep several batches in flight on
    // different streams (the tiled identify) give each its own range of rows
    uint64_t row_base = 0;
    int ldx = 0;        // row stride of xd
    uint32_t out = 0;   // output width it was sized for
    float *xd = nullptr, *h1 = nullptr, *h2 = nullptr, *p = nullptr, *d2 = nullptr, *d1 = nullptr, *slab = nullptr,
          *w2t = nullptr;
    uint8_t* live = nullptr;
    uint32_t* mask = nullptr;     // [rows / 32][hidden1] sign bits of the first layer
    uint32_t* w2s = nullptr;      // bf16 planes of w2^T and of w2 (split-bf16 kernels), two bf16 per word
    uint32_t* w3s = nullptr;      // the same for w3^T and w3 (GEMM output path only)
    float* w3t = nullptr;         // w3 transposed, the source of its planes
    float* w1t = nullptr;         // [hidden1][ldx]: w1 transposed, zeros behind its `in` columns
    float* vote_tab = nullptr;    // the three layers' weights in the fetch order of the fused vote kernel (infer_kernel.h)
    double* stat_slab = nullptr;  // per-workgroup {loss, live rows} partials of the head kernels
    uint64_t slab_floats = 0;
    // regions of `slab` (float offsets): split-K partials of dW1|db1, dW2, dW3|db3 and db2
    uint64_t sl_w1 = 0, sl_w2 = 0, sl_w3 = 0, sl_b2 = 0, sl_b3 = 0;
};

// Grow-only device (and pinned host) buffers a net keeps for its host-memory entry points, so that a
// per-window call (forward of one row, train_batch of eight) allocates nothing after the first.
struct Staging {
    static constexpr int kSlots = 4;
    void* dev[kSlots] = {};
    size_t dev_cap[kSlots] = {};
    void* pinned = nullptr;
    size_t pinned_cap = 0;
};

struct Net {
    uint32_t in = 0, h1 = 0, h2 = 0, out = 0;
    float* d_params = nullptr;  // [w1|b1|w2|b2|w3|b3]
    float* d_grad = nullptr;    // same layout + {loss_sum, live_rows}
    double* d_stats = nullptr;  // {loss, live rows} since zero_grad, then the same as running totals
    std::mutex mu;              // serialises host-API calls that share the workspace
    Workspace ws;
    Staging stage;              // guarded by mu
    // What identify_speaker* keeps between calls (guarded by mu): its stream, and the buffers of recordings
    // short enough to go through in one piece -- a call on a one-second clip otherwise spends most of its time
    // creating and destroying them.
    struct IdentifyKeep {
        hipStream_t compute = nullptr;
        void* buf[3] = {};       // features, two PCM staging buffers
        size_t cap[3] = {};
    } keep;
    int device = 0;
    int gemm_mode = 0;          // SZH_GEMM_*: arithmetic of the large batched products
    // metadata carried by model.npz (lib.rs:755-761)
    std::vector<std::vector<std::string>> file_lists;
    uint32_t sample_rate = 44100;
    uint16_t bits = 16;
    std::vector<float> emb_vectors, emb_mean, emb_std;
    uint32_t emb_dim = 0;
    // optional w4/b4 "encoding layer" columns are preserved through load/save untouched
    std::vector<std::vector<float>> w4_cols;
    std::vector<float> b4;

    uint64_t off_w1() const { return 0; }
    uint64_t off_b1() const { return (uint64_t)in * h1; }
    uint64_t off_w2() const { return off_b1() + h1; }
    uint64_t off_b2() const { return off_w2() + (uint64_t)h1 * h2; }
    uint64_t off_w3() const { return off_b2() + h2; }
    uint64_t off_b3() const { return off_w3() + (uint64_t)h2 * out; }
    uint64_t param_count() const { return off_b3() + out; }
};

int net_create(uint32_t in, uint32_t h1, uint32_t h2, uint32_t out, const float* init, uint64_t seed, Net** net);
void net_destroy(Net* n);
int net_set_params(Net* n, const float* flat);
int net_get_params(Net* n, float* flat);
int net_add_output_class(Net* n, const float* col, uint64_t seed);

enum ForwardKind { kFwdProbs = 0, kFwdEmbedTanh = 1, kFwdEmbedRelu = 2 };
// x_dev[n][in] -> out_dev ([n][out] probabilities or [n][h2] embeddings)
int net_forward_dev(Net* n, const float* x_dev, uint64_t rows, float* out_dev, ForwardKind kind, hipStream_t s);

struct AccumArgs {
    const float* x_dev = nullptr;       // [rows_src][in]
    uint64_t rows = 0;                  // rows of this call
    const uint32_t* perm_dev = nullptr; // optional: row r reads x_dev[perm[r]]
    const uint8_t* keep_dev = nullptr;  // optional: [rows][in] explicit dropout mask (0 = zero it)
    float dropout = 0.0f;               // used when keep_dev == nullptr
    uint64_t seed = 0;                  // counter-based generator key
    uint64_t row_counter0 = 0;          // counter of row 0 (keeps streams distinct across calls)
    const float* target_dev = nullptr;  // [out] shared target, or
    const int32_t* labels_dev = nullptr;// [rows] per-row class
    bool skip_zero_rows = false;        // lib.rs:607-609: rows whose inputs are all 0.0 do not count
};
int net_zero_grad(Net* n, hipStream_t s);
int net_accumulate(Net* n, const AccumArgs& a, hipStream_t s);
int net_apply(Net* n, float lr, hipStream_t s);
int net_reset_totals(Net* n, hipStream_t s);
// The reference's batch-of-8 schedule as one resident kernel (step.hip).  `applies` says whether
// the network and batch size fit it; d_perm / d_keep hold every epoch's shuffle / dropout mask.
bool step_kernel_applies(const Net* n, uint32_t batch_size, uint64_t n_steps);
// net_pretrain_steps returns this (not an error) when the resident kernel could not run to the end
// -- its workgroups would not all be resident, or one of them gave up waiting for the others --
// and the parameters are exactly as before the call: the caller takes the per-layer path.
constexpr int SZH_STEP_NOT_RUN = 1;
// Dropout draws are keyed by (seed, key_row0 + epoch * n_win + position, feature) with `epoch` counted
// from the launch's first one: the caller passes the rows of the epochs before it, so that the draws
// equal those of net_accumulate with row_counter0 = that position.
int net_pretrain_steps(Net* n, const float* d_win, uint32_t n_win, const uint32_t* d_perm, const uint8_t* d_keep,
                       float dropout, uint64_t seed, uint64_t key_row0, const float* d_target, uint32_t epochs,
                       uint32_t batch_size, float lr, hipStream_t s);
int net_read_totals(Net* n, hipStream_t s, double* loss_sum, double* live_rows);
// per-window votes (counts_dev[out], optional) and per-class probability sums (psums_dev[out],
// optional; the sums identify_speaker compares, lib.rs:1290-1297) of `rows` feature rows, added
// to what the arrays hold
// tables_prepared: net_vote_prepare ran on this net since its parameters last changed (callers that vote
// on many batches in a row, the tiled identify, prepare once)
int net_vote_dev(Net* n, const float* x_dev, uint64_t rows, float threshold, unsigned long long* counts_dev,
                 double* psums_dev, hipStream_t s, bool tables_prepared = false);
int net_vote_prepare(Net* n, hipStream_t s);
bool net_vote_fused(const Net* n);   // the one-kernel vote applies to this net's widths
// Pooled embedding (un-normalised) of device feature rows, written to host memory out[hidden2]:
// mode 0 mean of forward_embedding, 1 median of forward_embedding, 2 median of embed.
int net_pool_host(Net* n, const float* x_dev, uint64_t rows, int mode, float* out);
// sizes the workspace for calls of up to `rows` rows, so that later calls allocate nothing
int net_reserve_rows(Net* n, uint64_t rows);
// rows the workspace holds after net_reserve_rows (a request may be capped)
uint64_t net_workspace_rows(const Net* n);
// device buffer `slot` of at least `bytes` (grow-only, owned by the net; call with n->mu held)
int net_stage_dev(Net* n, int slot, size_t bytes, void** out);
// pinned host buffer of at least `bytes` (grow-only; small transfers go through it without a pageable-copy stall)
// Pageable host memory to device memory, or back, through a ring of pinned slots and helper threads (host_copy.cpp):
// start() takes the ring (pinned, owned by the caller, must outlive finish()), the direction, and starts the
// helpers.  To the device, copy() returns once every piece of the source has been handed to the stream (the last
// ones may still be in flight); to the host, once the destination holds the data.  finish() stops the helpers and
// waits until the ring is no longer in use.  One thread calls copy() at a time.
class HostStager {
public:
    HostStager();
    ~HostStager();
    HostStager(const HostStager&) = delete;
    HostStager& operator=(const HostStager&) = delete;
    // helper threads to use: STREAMZ_COPY_THREADS (0 = plain pageable copies, at most 8), 3 by default
    static int helpers_wanted();
    int start(void* pinned_ring, size_t ring_bytes, size_t piece_bytes, int helpers, bool to_host = false);
    bool started() const { return impl_ != nullptr; }
    int copy(void* dst, const void* src, size_t bytes, hipStream_t s);
    // several copies as one pipeline (the helpers run on into the next span while the first is still in flight)
    struct Span { void* dst; const void* src; size_t bytes; };
    int copy_many(const Span* spans, size_t n_spans, hipStream_t s);
    void finish();
private:
    struct Impl;
    Impl* impl_ = nullptr;
};

// the kept stream / a kept buffer of at least `bytes` (which: 0 features, 1 and 2 PCM); see Net::IdentifyKeep
int net_keep_stream(Net* n, hipStream_t* compute);
int net_keep_buffer(Net* n, int which, size_t bytes, void** out);
int net_stage_pinned(Net* n, size_t bytes, void** out);
// identify_speaker_list's vote counts for a long recording, tile by tile (stream.cpp).  pcm is a
// host or a device pointer; counts_dev[out] / psums_dev[out] (either may be null) are added to.
int identify_tiled(Net* n, Extractor* e, const int16_t* pcm, bool pcm_on_device, uint64_t n_samples, float threshold,
                   uint32_t tile_windows, unsigned long long* counts_dev, double* psums_dev);

// uniform in [-0.5, 0.5) from a counter-based generator; used for weight initialisation
float init_uniform(uint64_t seed, uint64_t index);

}  // namespace sz

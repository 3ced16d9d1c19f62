// extern "C" surface of libstreamz_hip.so (see include/streamz_hip.h) and the host-side
// drivers that the reference keeps in lib.rs next to the numeric code: the per-file epoch
// loop (lib.rs:582-665), identify_speaker_list (lib.rs:1383-1411), the feature cache
// (lib.rs:550-579) and model.npz persistence (lib.rs:1081-1281).
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>

#include "internal.h"
#include "rng.h"
#include "npy.h"
#include "streamz_hip.h"
#include "wav.h"

namespace sz {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return SZH_ENODEVICE;
}

int allow_dynamic_lds(const void* kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    int dev = 0;
    SZ_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = done[{kernel, dev}];
    if (have >= bytes) return SZH_OK;
    SZ_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return SZH_OK;
}

namespace {

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        SZ_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return SZH_OK;
    }
    template <class T> T* as() { return static_cast<T*>(p); }
};

int extract_files_host(Extractor* e, const int16_t* const* pcm, const uint64_t* n_samples, uint32_t n_files,
                       float* const* out)
{
    std::vector<uint64_t> off(n_files), len(n_files);
    uint64_t total = 0;
    for (uint32_t f = 0; f < n_files; ++f) {
        off[f] = total;
        len[f] = n_samples[f];
        total += (n_samples[f] + 7) / 8 * 8;
    }
    Plan* plan = nullptr;
    int rc = plan_create(e, off.data(), len.data(), n_files, &plan);
    if (rc) return rc;
    struct PlanGuard { Plan* p; ~PlanGuard() { plan_destroy(p); } } guard{plan};
    if (plan->total_windows == 0) return SZH_OK;
    DevBuf d_pcm, d_feat;
    rc = d_pcm.alloc(total * sizeof(int16_t));
    if (rc) return rc;
    rc = d_feat.alloc(plan->total_windows * kFeat * sizeof(float));
    if (rc) return rc;
    // From 32 MB of PCM on, the files go to the device through pinned pieces filled by helper threads, one
    // pipeline across all files (host_copy.cpp): about twice the rate of pageable copies.
    constexpr size_t kStagedBytes = 32u << 20, kRingBytes = 32u << 20, kPieceBytes = 4u << 20;
    struct Pinned { void* p = nullptr; ~Pinned() { if (p) (void)hipHostFree(p); } } ring;
    HostStager stager;   // (declared after the ring: it is finished before the ring is freed)
    if (total * sizeof(int16_t) >= kStagedBytes && HostStager::helpers_wanted() > 0) {
        SZ_HIP(hipHostMalloc(&ring.p, kRingBytes, hipHostMallocDefault));
        rc = stager.start(ring.p, kRingBytes, kPieceBytes, HostStager::helpers_wanted());
        if (rc) return rc;
    }
    if (stager.started()) {
        std::vector<HostStager::Span> spans;
        for (uint32_t f = 0; f < n_files; ++f)
            if (len[f] >= (uint64_t)kWindow) spans.push_back({d_pcm.as<int16_t>() + off[f], pcm[f], len[f] * sizeof(int16_t)});
        rc = stager.copy_many(spans.data(), spans.size(), nullptr);
        if (rc) return rc;
    } else {
        for (uint32_t f = 0; f < n_files; ++f)
            if (len[f] >= (uint64_t)kWindow)
                SZ_HIP(hipMemcpy(d_pcm.as<int16_t>() + off[f], pcm[f], len[f] * sizeof(int16_t), hipMemcpyHostToDevice));
    }
    stager.finish();
    rc = plan_run(plan, d_pcm.as<int16_t>(), d_feat.as<float>(), nullptr);
    if (rc) return rc;
    // the features come back the same way (a pageable copy into pages the host has not touched yet is slower still)
    HostStager back;
    if (ring.p) {
        rc = back.start(ring.p, kRingBytes, kPieceBytes, HostStager::helpers_wanted(), true);
        if (rc) return rc;
    } else {
        SZ_HIP(hipStreamSynchronize(nullptr));
    }
    std::vector<HostStager::Span> spans;
    for (uint32_t f = 0; f < n_files; ++f) {
        const uint64_t nw = plan->win_off[f + 1] - plan->win_off[f];
        if (!nw) continue;
        const float* src = d_feat.as<float>() + plan->win_off[f] * kFeat;
        if (ring.p) spans.push_back({out[f], src, nw * kFeat * sizeof(float)});
        else SZ_HIP(hipMemcpy(out[f], src, nw * kFeat * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (ring.p) {
        rc = back.copy_many(spans.data(), spans.size(), nullptr);
        if (rc) return rc;
    }
    return SZH_OK;
}

// Host rows -> the net's device staging buffer `slot`.  Small inputs travel through the net's pinned
// buffer as an asynchronous copy (a pageable hipMemcpy of a few hundred bytes costs more than the
// kernels of a one-row forward).
constexpr size_t kPinnedLimit = 1 << 20;
int stage_in(Net* n, int slot, const void* host, size_t bytes, size_t pinned_off, void** dev)
{
    int rc = net_stage_dev(n, slot, bytes, dev);
    if (rc) return rc;
    if (pinned_off + bytes <= kPinnedLimit) {
        void* pin = nullptr;
        rc = net_stage_pinned(n, kPinnedLimit, &pin);
        if (rc) return rc;
        std::memcpy(static_cast<char*>(pin) + pinned_off, host, bytes);
        SZ_HIP(hipMemcpyAsync(*dev, static_cast<char*>(pin) + pinned_off, bytes, hipMemcpyHostToDevice, nullptr));
    } else {
        SZ_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    }
    return SZH_OK;
}

int forward_host(Net* n, const float* x, uint64_t B, float* out, ForwardKind kind)
{
    if (B == 0) return SZH_OK;
    const uint32_t ow = kind == kFwdProbs ? n->out : n->h2;
    if (ow == 0) return SZH_OK;
    std::lock_guard<std::mutex> lock(n->mu);
    const size_t in_bytes = B * n->in * sizeof(float), out_bytes = B * ow * sizeof(float);
    void *dx = nullptr, *dout = nullptr;
    int rc = stage_in(n, 0, x, in_bytes, 0, &dx);
    if (rc) return rc;
    rc = net_stage_dev(n, 1, out_bytes, &dout);
    if (rc) return rc;
    rc = net_forward_dev(n, static_cast<const float*>(dx), B, static_cast<float*>(dout), kind, nullptr);
    if (rc) return rc;
    const size_t in_pin = (in_bytes + 255) / 256 * 256;
    if (in_pin + out_bytes <= kPinnedLimit) {  // result back through the pinned buffer, one synchronisation
        void* pin = nullptr;
        rc = net_stage_pinned(n, kPinnedLimit, &pin);
        if (rc) return rc;
        SZ_HIP(hipMemcpyAsync(static_cast<char*>(pin) + in_pin, dout, out_bytes, hipMemcpyDeviceToHost, nullptr));
        SZ_HIP(hipStreamSynchronize(nullptr));
        std::memcpy(out, static_cast<char*>(pin) + in_pin, out_bytes);
    } else {
        SZ_HIP(hipStreamSynchronize(nullptr));
        SZ_HIP(hipMemcpy(out, dout, out_bytes, hipMemcpyDeviceToHost));
    }
    return SZH_OK;
}

int train_batch_host(Net* n, const float* x, uint64_t B, const float* target, float lr)
{
    if (B == 0 || n->out == 0) return SZH_OK;  // lib.rs:1003-1005
    std::lock_guard<std::mutex> lock(n->mu);
    const size_t in_bytes = B * n->in * sizeof(float), t_bytes = n->out * sizeof(float);
    void *dx = nullptr, *dt = nullptr;
    int rc = stage_in(n, 0, x, in_bytes, 0, &dx);
    if (rc) return rc;
    rc = stage_in(n, 2, target, t_bytes, (in_bytes + 255) / 256 * 256, &dt);
    if (rc) return rc;
    rc = net_zero_grad(n, nullptr);
    if (rc) return rc;
    AccumArgs a;
    a.x_dev = static_cast<const float*>(dx);
    a.rows = B;
    a.target_dev = static_cast<const float*>(dt);
    // train_batch takes the rows as given: every row counts, also an all-zero one
    rc = net_accumulate(n, a, nullptr);
    if (rc) return rc;
    rc = net_apply(n, lr, nullptr);
    if (rc) return rc;
    SZ_HIP(hipStreamSynchronize(nullptr));
    return SZH_OK;
}

// Epochs [ep_first, ep_first + n_epochs) of the pretrain loop (lib.rs:599-622) over device-resident
// windows.  rng's perm / keep arrays are indexed by absolute epoch; `draw` carries the shuffle
// generator across calls.  Losses are added to the net's running totals.
int pretrain_epochs_dev(Net* n, const float* d_win, uint64_t n_win, const float* d_tgt, uint32_t ep_first,
                        uint32_t n_epochs, float lr, float dropout, uint64_t bs, const szh_rng* rng, uint64_t seed,
                        uint64_t& draw)
{
    if (n_win == 0 || n_epochs == 0) return SZH_OK;
    const bool have_keep = rng && rng->keep;
    std::vector<uint32_t> order(n_win);
    auto shuffle_epoch = [&](uint32_t ep) -> int {
        if (rng && rng->perm) {
            std::memcpy(order.data(), rng->perm + (size_t)ep * n_win, n_win * sizeof(uint32_t));
            for (uint64_t i = 0; i < n_win; ++i)
                if (order[i] >= n_win) { set_error("permutation entry out of range"); return SZH_EINVAL; }
        } else {
            for (uint64_t i = 0; i < n_win; ++i) order[i] = (uint32_t)i;
            for (uint64_t i = n_win; i > 1; --i) {  // Fisher-Yates, as SliceRandom::shuffle (lib.rs:601)
                const uint64_t j = mix64(seed ^ mix64(draw++)) % i;
                std::swap(order[i - 1], order[j]);
            }
        }
        return SZH_OK;
    };
    int rc;
    const uint64_t steps_per_epoch = (n_win + bs - 1) / bs;
    DevBuf d_perm, d_keep;
    // one epoch as launches per layer and step: the path of batch sizes above 8, and of a group of
    // epochs whose resident kernel did not run
    auto epoch_per_layer = [&](uint32_t ep, const uint32_t* host_order) -> int {
        if (!d_perm.p) { rc = d_perm.alloc(n_win * sizeof(uint32_t)); if (rc) return rc; }
        if (have_keep && !d_keep.p) { rc = d_keep.alloc(n_win * kFeat); if (rc) return rc; }
        SZ_HIP(hipMemcpyAsync(d_perm.p, host_order, n_win * sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));
        if (have_keep)
            SZ_HIP(hipMemcpyAsync(d_keep.p, rng->keep + (size_t)ep * n_win * kFeat, n_win * kFeat, hipMemcpyHostToDevice,
                                  nullptr));
        SZ_HIP(hipStreamSynchronize(nullptr));  // the caller reuses host_order
        for (uint64_t c0 = 0; c0 < n_win; c0 += bs) {
            AccumArgs a;
            a.x_dev = d_win;
            a.rows = std::min<uint64_t>(bs, n_win - c0);
            a.perm_dev = d_perm.as<uint32_t>() + c0;
            a.keep_dev = have_keep ? d_keep.as<uint8_t>() + c0 * kFeat : nullptr;
            a.dropout = dropout;
            a.seed = seed;
            a.row_counter0 = (uint64_t)ep * n_win + c0;
            a.target_dev = d_tgt;
            a.skip_zero_rows = true;
            rc = net_zero_grad(n, nullptr);
            if (rc) return rc;
            rc = net_accumulate(n, a, nullptr);
            if (rc) return rc;
            rc = net_apply(n, lr, nullptr);
            if (rc) return rc;
        }
        return net_zero_grad(n, nullptr);  // folds the last step's loss into the running totals
    };
    // Batches of up to 8 rows (the reference's CLI uses 8) run as one resident kernel per group of
    // epochs: all their shuffles (and injected masks) are uploaded first.
    const uint64_t group = std::max<uint64_t>(1, std::min<uint64_t>(n_epochs, (16ull << 20) / n_win));
    if (step_kernel_applies(n, (uint32_t)bs, group * steps_per_epoch)) {
        DevBuf d_perms, d_keeps;
        rc = d_perms.alloc(group * n_win * sizeof(uint32_t));
        if (rc) return rc;
        if (have_keep) { rc = d_keeps.alloc(group * n_win * kFeat); if (rc) return rc; }
        std::vector<uint32_t> orders(group * n_win);
        for (uint32_t e0 = 0; e0 < n_epochs; e0 += (uint32_t)group) {
            const uint32_t ne = (uint32_t)std::min<uint64_t>(group, n_epochs - e0);
            for (uint32_t e = 0; e < ne; ++e) {
                rc = shuffle_epoch(ep_first + e0 + e);
                if (rc) return rc;
                std::memcpy(orders.data() + (size_t)e * n_win, order.data(), n_win * sizeof(uint32_t));
            }
            SZ_HIP(hipMemcpy(d_perms.p, orders.data(), (size_t)ne * n_win * sizeof(uint32_t), hipMemcpyHostToDevice));
            if (have_keep)
                SZ_HIP(hipMemcpy(d_keeps.p, rng->keep + (size_t)(ep_first + e0) * n_win * kFeat, (size_t)ne * n_win * kFeat,
                                 hipMemcpyHostToDevice));
            // the dropout stream is keyed by (seed, absolute epoch * n_win + position, feature) on either path
            rc = net_pretrain_steps(n, d_win, (uint32_t)n_win, d_perms.as<uint32_t>(),
                                    have_keep ? d_keeps.as<uint8_t>() : nullptr, dropout, seed, (uint64_t)(ep_first + e0) * n_win, d_tgt,
                                    ne, (uint32_t)bs, lr, nullptr);
            if (rc == SZH_STEP_NOT_RUN) {  // parameters untouched: the same epochs, same shuffles, same dropout draws, layer by layer
                for (uint32_t e = 0; e < ne; ++e) {
                    rc = epoch_per_layer(ep_first + e0 + e, orders.data() + (size_t)e * n_win);
                    if (rc) return rc;
                }
            }
            if (rc) return rc;
        }
        return SZH_OK;
    }
    for (uint32_t ep = ep_first; ep < ep_first + n_epochs; ++ep) {
        rc = shuffle_epoch(ep);
        if (rc) return rc;
        rc = epoch_per_layer(ep, order.data());
        if (rc) return rc;
    }
    return SZH_OK;
}

// one-hot target of width num_classes (lib.rs:592-595), seen through the net's output width
int upload_target(Net* n, uint32_t target_class, uint32_t num_classes, DevBuf& d_tgt)
{
    std::vector<float> target(n->out, 0.0f);
    if (target_class < num_classes && target_class < n->out) target[target_class] = 1.0f;
    int rc = d_tgt.alloc(n->out * sizeof(float));
    if (rc) return rc;
    SZ_HIP(hipMemcpy(d_tgt.p, target.data(), n->out * sizeof(float), hipMemcpyHostToDevice));
    return SZH_OK;
}

int read_mean_loss(Net* n, float* mean_loss)
{
    double loss_sum = 0.0, live = 0.0;
    int rc = net_read_totals(n, nullptr, &loss_sum, &live);
    if (rc) return rc;
    *mean_loss = live > 0.0 ? (float)(loss_sum / live) : 0.0f;  // lib.rs:623-627
    return SZH_OK;
}

int pretrain_host(Net* n, const float* windows, uint64_t n_win, uint32_t target_class, uint32_t num_classes,
                  uint32_t epochs, float lr, float dropout, uint32_t batch_size, const szh_rng* rng, float* mean_loss)
{
    *mean_loss = 0.0f;
    if (n->in != (uint32_t)kFeat) { set_error("pretrain_from_features needs a net with 60 inputs"); return SZH_EINVAL; }
    if (n_win == 0 || epochs == 0 || n->out == 0) return SZH_OK;
    if (n_win > 0xffffffffull) { set_error("too many windows"); return SZH_EINVAL; }
    std::lock_guard<std::mutex> lock(n->mu);
    const uint64_t bs = batch_size > 0 ? batch_size : 1;  // batch_size.max(1), lib.rs:602
    DevBuf d_win, d_tgt;
    int rc = d_win.alloc(n_win * kFeat * sizeof(float));
    if (rc) return rc;
    rc = upload_target(n, target_class, num_classes, d_tgt);
    if (rc) return rc;
    SZ_HIP(hipMemcpy(d_win.p, windows, n_win * kFeat * sizeof(float), hipMemcpyHostToDevice));
    rc = net_reset_totals(n, nullptr);
    if (rc) return rc;
    uint64_t draw = 0;
    rc = pretrain_epochs_dev(n, d_win.as<float>(), n_win, d_tgt.as<float>(), 0, epochs, lr, dropout, bs, rng,
                             rng ? rng->seed : 0x5A17ull, draw);
    if (rc) return rc;
    return read_mean_loss(n, mean_loss);
}

// augment, lib.rs:103-116, on host buffers.  NULL noise: drawn from `seed`.
int augment_host(const int16_t* pcm, uint64_t n, float gain, float noise_level, uint64_t shift, const float* noise,
                 uint64_t seed, int16_t* out)
{
    if (n == 0) return SZH_OK;
    DevBuf d_in, d_out, d_noise;
    int rc = d_in.alloc(n * sizeof(int16_t));
    if (rc) return rc;
    rc = d_out.alloc(n * sizeof(int16_t));
    if (rc) return rc;
    SZ_HIP(hipMemcpy(d_in.p, pcm, n * sizeof(int16_t), hipMemcpyHostToDevice));
    if (noise) {
        rc = d_noise.alloc(n * sizeof(float));
        if (rc) return rc;
        SZ_HIP(hipMemcpy(d_noise.p, noise, n * sizeof(float), hipMemcpyHostToDevice));
    }
    rc = augment_dev(d_in.as<int16_t>(), n, gain, noise_level, shift, noise ? d_noise.as<float>() : nullptr, seed, 0,
                     d_out.as<int16_t>(), nullptr);
    if (rc) return rc;
    SZ_HIP(hipMemcpy(out, d_out.p, n * sizeof(int16_t), hipMemcpyDeviceToHost));
    return SZH_OK;
}

// pretrain_network, lib.rs:348-397: per epoch augment the raw samples, extract, shuffle, train.
int pretrain_network_host(Net* n, Extractor* e, const int16_t* pcm, uint64_t n_samples, uint32_t target_class,
                          uint32_t num_classes, uint32_t epochs, float lr, float dropout, uint32_t batch_size,
                          const szh_aug_draws* aug, const szh_rng* rng, float* mean_loss)
{
    *mean_loss = 0.0f;
    if (n->in != (uint32_t)kFeat) { set_error("pretrain_network needs a net with 60 inputs"); return SZH_EINVAL; }
    if (n_samples == 0 || epochs == 0 || n->out == 0) return SZH_OK;
    std::lock_guard<std::mutex> lock(n->mu);
    const uint64_t bs = batch_size > 0 ? batch_size : 1;
    const uint64_t seed = rng ? rng->seed : 0x5A17ull;
    const uint64_t off = 0;
    Plan* plan = nullptr;
    int rc = plan_create(e, &off, &n_samples, 1, &plan);
    if (rc) return rc;
    struct PlanGuard { Plan* p; ~PlanGuard() { plan_destroy(p); } } guard{plan};
    const uint64_t n_win = plan->total_windows;
    DevBuf d_in, d_aug, d_noise, d_feat, d_tgt;
    if ((rc = d_in.alloc(n_samples * sizeof(int16_t))) || (rc = d_aug.alloc((n_samples + 8) * sizeof(int16_t))) ||
        (rc = d_feat.alloc((n_win ? n_win : 1) * kFeat * sizeof(float))) || (rc = upload_target(n, target_class, num_classes, d_tgt)))
        return rc;
    if (aug && aug->noise) { rc = d_noise.alloc(n_samples * sizeof(float)); if (rc) return rc; }
    SZ_HIP(hipMemcpy(d_in.p, pcm, n_samples * sizeof(int16_t), hipMemcpyHostToDevice));
    rc = net_reset_totals(n, nullptr);
    if (rc) return rc;
    const uint64_t shift_range = std::min<uint64_t>(n_samples, (uint64_t)kWindow);  // lib.rs:107
    uint64_t draw = 0;
    for (uint32_t ep = 0; ep < epochs; ++ep) {
        // lib.rs:105-107: noise level in [0, 0.005), gain in [0.95, 1.05), shift in [0, min(len, 800))
        const float level = aug && aug->noise_level ? aug->noise_level[ep] : 0.005f * uniform01(seed, ep, 0xA061);
        const float gain = aug && aug->gain ? aug->gain[ep] : 0.95f + 0.1f * uniform01(seed, ep, 0xA062);
        const uint64_t shift = aug && aug->shift ? aug->shift[ep] % shift_range : mix64(seed ^ mix64(0xA063 + ep)) % shift_range;
        if (aug && aug->noise)
            SZ_HIP(hipMemcpy(d_noise.p, aug->noise + (size_t)ep * n_samples, n_samples * sizeof(float), hipMemcpyHostToDevice));
        rc = augment_dev(d_in.as<int16_t>(), n_samples, gain, level, shift, aug && aug->noise ? d_noise.as<float>() : nullptr,
                         mix64(seed + 0xA064), (uint64_t)ep * n_samples, d_aug.as<int16_t>(), nullptr);
        if (rc) return rc;
        if (n_win == 0) continue;
        rc = plan_run(plan, d_aug.as<int16_t>(), d_feat.as<float>(), nullptr);
        if (rc) return rc;
        rc = pretrain_epochs_dev(n, d_feat.as<float>(), n_win, d_tgt.as<float>(), ep, 1, lr, dropout, bs, rng, seed, draw);
        if (rc) return rc;
    }
    return read_mean_loss(n, mean_loss);
}

// order speakers by vote count, descending, ties by ascending id (stable sort, lib.rs:1403-1410)
uint32_t order_votes(const std::vector<unsigned long long>& counts, uint32_t* ids)
{
    std::vector<std::pair<uint32_t, unsigned long long>> pairs;
    for (uint32_t i = 0; i < counts.size(); ++i)
        if (counts[i] > 0) pairs.emplace_back(i, counts[i]);
    std::stable_sort(pairs.begin(), pairs.end(),
                     [](const std::pair<uint32_t, unsigned long long>& a,
                        const std::pair<uint32_t, unsigned long long>& b) { return a.second > b.second; });
    for (size_t i = 0; i < pairs.size(); ++i) ids[i] = pairs[i].first;
    return (uint32_t)pairs.size();
}

int vote_features_dev(Net* n, const float* d_feats, uint64_t n_win, float threshold, uint32_t* ids, uint32_t capacity,
                      uint32_t* n_ids, uint64_t* counts_out)
{
    if (capacity < n->out) { set_error("ids capacity %u < output size %u", capacity, n->out); return SZH_EINVAL; }
    std::vector<unsigned long long> counts(n->out, 0);
    if (n->out > 0 && n_win > 0) {
        void* d_cnt = nullptr;  // the net's staging buffer (the caller holds n->mu)
        int rc = net_stage_dev(n, 3, n->out * sizeof(unsigned long long), &d_cnt);
        if (rc) return rc;
        SZ_HIP(hipMemsetAsync(d_cnt, 0, n->out * sizeof(unsigned long long), nullptr));
        rc = net_vote_dev(n, d_feats, n_win, threshold, static_cast<unsigned long long*>(d_cnt), nullptr, nullptr);
        if (rc) return rc;
        SZ_HIP(hipStreamSynchronize(nullptr));
        SZ_HIP(hipMemcpy(counts.data(), d_cnt, n->out * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    *n_ids = order_votes(counts, ids);
    if (counts_out)
        for (uint32_t i = 0; i < n->out; ++i) counts_out[i] = counts[i];
    return SZH_OK;
}

std::string feature_cache_path(const std::string& path)
{
    // lib.rs:550-555: the directory is created as a side effect of asking for the path
    ::mkdir("feature_cache", 0777);
    std::string s = path;
    for (char& c : s)
        if (c == '/' || c == '\\') c = '_';
    return "feature_cache/" + s + ".npy";
}

bool file_exists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}

bool ends_with_ci(const std::string& s, const char* suf)
{
    const size_t n = std::strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
        if (std::tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
    return true;
}

int load_audio_mono(const std::string& path, std::vector<int16_t>& mono)
{
    if (!ends_with_ci(path, ".wav")) {
        set_error("Unsupported format: %s (only 16-bit PCM WAV is handled on this path)", path.c_str());
        return SZH_EFORMAT;
    }
    std::vector<int16_t> raw;
    WavInfo info;
    std::string err;
    if (!wav_read_raw(path, raw, info, err)) { set_error("%s", err.c_str()); return SZH_EIO; }
    if (info.sample_rate != (uint32_t)kSampleRate) {
        set_error("%s: sample rate %u; only 44100 Hz input is handled (resampling is not part of this path)",
                  path.c_str(), info.sample_rate);
        return SZH_EFORMAT;
    }
    mono = downmix_to_mono(raw, info.channels);
    return SZH_OK;
}

template <class T> const T* as(const NpyArray& a) { return reinterpret_cast<const T*>(a.data.data()); }

}  // namespace
}  // namespace sz

using namespace sz;

#define SZ_TRY try {
#define SZ_CATCH                                                           \
    }                                                                      \
    catch (const std::bad_alloc&) { set_error("out of host memory"); return SZH_ENOMEM; } \
    catch (const std::exception& ex) { set_error("internal error: %s", ex.what()); return SZH_EINVAL; }

extern "C" {

int szh_init(int device_ordinal)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) { set_error("no HIP device available"); return SZH_ENODEVICE; }
    if (device_ordinal < 0 || device_ordinal >= n) { set_error("device ordinal %d out of range (%d devices)", device_ordinal, n); return SZH_EINVAL; }
    SZ_HIP(hipSetDevice(device_ordinal));
    return noalloc_setup();
}

int szh_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

int szh_shutdown(void)
{
    // handles own all device memory; what is left is work still in flight on the device
    SZ_HIP(hipDeviceSynchronize());
    return SZH_OK;
}

const char* szh_last_error(void) { return g_err; }
void szh_free(void* p) { std::free(p); }

uint64_t szh_window_count(uint64_t n_samples)
{
    if (n_samples < (uint64_t)kWindow) return 0;
    return (n_samples - kWindow) / kHop + 1;
}

int szh_extractor_create(szh_extractor** ext)
{
    SZ_TRY
    if (!ext) { set_error("null argument"); return SZH_EINVAL; }
    Extractor* e = nullptr;
    const int rc = extractor_create(&e);
    if (rc) return rc;
    *ext = reinterpret_cast<szh_extractor*>(e);
    return SZH_OK;
    SZ_CATCH
}
void szh_extractor_destroy(szh_extractor* ext) { extractor_destroy(reinterpret_cast<Extractor*>(ext)); }

int szh_extractor_mel_filters(const szh_extractor* ext, float* out)
{
    if (!ext || !out) { set_error("null argument"); return SZH_EINVAL; }
    std::memcpy(out, reinterpret_cast<const Extractor*>(ext)->host.mel_dense, sizeof(float) * kMels * kBins);
    return SZH_OK;
}

int szh_extract(szh_extractor* ext, const int16_t* pcm, uint64_t n_samples, float* out, uint64_t capacity_windows,
                uint64_t* n_windows)
{
    SZ_TRY
    if (!ext || !n_windows || (n_samples && !pcm)) { set_error("null argument"); return SZH_EINVAL; }
    const uint64_t nw = szh_window_count(n_samples);
    *n_windows = nw;
    if (nw == 0) return SZH_OK;
    if (nw > capacity_windows || !out) { set_error("output holds %llu windows, %llu needed", (unsigned long long)capacity_windows, (unsigned long long)nw); return SZH_EINVAL; }
    return extract_files_host(reinterpret_cast<Extractor*>(ext), &pcm, &n_samples, 1, &out);
    SZ_CATCH
}

int szh_extract_batch(szh_extractor* ext, const int16_t* const* pcm, const uint64_t* n_samples, uint32_t n_files,
                      float* const* out)
{
    SZ_TRY
    if (!ext || (n_files && (!pcm || !n_samples || !out))) { set_error("null argument"); return SZH_EINVAL; }
    return extract_files_host(reinterpret_cast<Extractor*>(ext), pcm, n_samples, n_files, out);
    SZ_CATCH
}

int szh_plan_create(szh_extractor* ext, const uint64_t* offsets, const uint64_t* n_samples, uint32_t n_files,
                    szh_plan** plan)
{
    SZ_TRY
    if (!ext || !plan || (n_files && (!offsets || !n_samples))) { set_error("null argument"); return SZH_EINVAL; }
    Plan* p = nullptr;
    const int rc = plan_create(reinterpret_cast<Extractor*>(ext), offsets, n_samples, n_files, &p);
    if (rc) return rc;
    *plan = reinterpret_cast<szh_plan*>(p);
    return SZH_OK;
    SZ_CATCH
}
void szh_plan_destroy(szh_plan* plan) { plan_destroy(reinterpret_cast<Plan*>(plan)); }
uint64_t szh_plan_total_windows(const szh_plan* plan) { return plan ? reinterpret_cast<const Plan*>(plan)->total_windows : 0; }
int szh_plan_window_offsets(const szh_plan* plan, uint64_t* out)
{
    if (!plan || !out) { set_error("null argument"); return SZH_EINVAL; }
    const Plan* p = reinterpret_cast<const Plan*>(plan);
    std::memcpy(out, p->win_off.data(), p->win_off.size() * sizeof(uint64_t));
    return SZH_OK;
}
int szh_plan_run_dev(szh_plan* plan, const int16_t* pcm_dev, float* feats_dev, void* stream)
{
    if (!plan || !pcm_dev || !feats_dev) { set_error("null argument"); return SZH_EINVAL; }
    { int cur = -1; SZ_HIP(hipGetDevice(&cur)); if (cur != reinterpret_cast<Plan*>(plan)->ext->device) { set_error("plan belongs to device %d but the current device is %d", reinterpret_cast<Plan*>(plan)->ext->device, cur); return SZH_EINVAL; } }
    return plan_run(reinterpret_cast<Plan*>(plan), pcm_dev, feats_dev, static_cast<hipStream_t>(stream));
}

// ---- net ------------------------------------------------------------------------------

int szh_net_create(uint32_t input, uint32_t hidden1, uint32_t hidden2, uint32_t output, const float* init,
                   uint64_t seed, szh_net** net)
{
    SZ_TRY
    if (!net) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = nullptr;
    const int rc = net_create(input, hidden1, hidden2, output, init, seed, &n);
    if (rc) return rc;
    *net = reinterpret_cast<szh_net*>(n);
    return SZH_OK;
    SZ_CATCH
}
void szh_net_destroy(szh_net* net) { net_destroy(reinterpret_cast<Net*>(net)); }

int szh_net_clone(const szh_net* net, szh_net** out)
{
    SZ_TRY
    if (!net || !out) { set_error("null argument"); return SZH_EINVAL; }
    Net* s = const_cast<Net*>(reinterpret_cast<const Net*>(net));
    std::vector<float> p(s->param_count());
    int rc = net_get_params(s, p.data());
    if (rc) return rc;
    Net* n = nullptr;
    rc = net_create(s->in, s->h1, s->h2, s->out, p.data(), 0, &n);
    if (rc) return rc;
    n->file_lists = s->file_lists;
    n->sample_rate = s->sample_rate;
    n->bits = s->bits;
    n->gemm_mode = s->gemm_mode;
    n->emb_vectors = s->emb_vectors; n->emb_mean = s->emb_mean; n->emb_std = s->emb_std; n->emb_dim = s->emb_dim;
    n->w4_cols = s->w4_cols; n->b4 = s->b4;
    *out = reinterpret_cast<szh_net*>(n);
    return SZH_OK;
    SZ_CATCH
}

int szh_net_dims(const szh_net* net, uint32_t dims[4])
{
    if (!net || !dims) { set_error("null argument"); return SZH_EINVAL; }
    const Net* n = reinterpret_cast<const Net*>(net);
    dims[0] = n->in; dims[1] = n->h1; dims[2] = n->h2; dims[3] = n->out;
    return SZH_OK;
}
uint32_t szh_net_output_size(const szh_net* net) { return net ? reinterpret_cast<const Net*>(net)->out : 0; }
uint64_t szh_net_param_count(const szh_net* net) { return net ? reinterpret_cast<const Net*>(net)->param_count() : 0; }
int szh_net_get_params(szh_net* net, float* flat)
{
    if (!net || !flat) { set_error("null argument"); return SZH_EINVAL; }
    return net_get_params(reinterpret_cast<Net*>(net), flat);
}
int szh_net_set_params(szh_net* net, const float* flat)
{
    if (!net || !flat) { set_error("null argument"); return SZH_EINVAL; }
    return net_set_params(reinterpret_cast<Net*>(net), flat);
}
int szh_net_add_output_class(szh_net* net, const float* new_col, uint64_t seed)
{
    SZ_TRY
    if (!net) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    std::lock_guard<std::mutex> lock(n->mu);
    return net_add_output_class(n, new_col, seed);
    SZ_CATCH
}

int szh_net_forward(szh_net* net, const float* x, uint64_t B, float* probs)
{
    SZ_TRY
    if (!net || (B && (!x || !probs))) { set_error("null argument"); return SZH_EINVAL; }
    return forward_host(reinterpret_cast<Net*>(net), x, B, probs, kFwdProbs);
    SZ_CATCH
}
int szh_net_embed(szh_net* net, const float* x, uint64_t B, float* emb)
{
    SZ_TRY
    if (!net || (B && (!x || !emb))) { set_error("null argument"); return SZH_EINVAL; }
    return forward_host(reinterpret_cast<Net*>(net), x, B, emb, kFwdEmbedTanh);
    SZ_CATCH
}
int szh_net_forward_embedding(szh_net* net, const float* x, uint64_t B, float* emb)
{
    SZ_TRY
    if (!net || (B && (!x || !emb))) { set_error("null argument"); return SZH_EINVAL; }
    return forward_host(reinterpret_cast<Net*>(net), x, B, emb, kFwdEmbedRelu);
    SZ_CATCH
}
int szh_net_train(szh_net* net, const float* x, const float* target, float lr)
{
    SZ_TRY
    if (!net || !x || !target) { set_error("null argument"); return SZH_EINVAL; }
    return train_batch_host(reinterpret_cast<Net*>(net), x, 1, target, lr);
    SZ_CATCH
}
int szh_net_train_batch(szh_net* net, const float* x, uint64_t B, const float* target, float lr)
{
    SZ_TRY
    if (!net || (B && (!x || !target))) { set_error("null argument"); return SZH_EINVAL; }
    return train_batch_host(reinterpret_cast<Net*>(net), x, B, target, lr);
    SZ_CATCH
}

int szh_pretrain_from_features(szh_net* net, const float* windows, uint64_t n_windows, uint32_t target_class,
                               uint32_t num_classes, uint32_t epochs, float lr, float dropout, uint32_t batch_size,
                               const szh_rng* rng, float* mean_loss)
{
    SZ_TRY
    if (!net || !mean_loss || (n_windows && !windows)) { set_error("null argument"); return SZH_EINVAL; }
    return pretrain_host(reinterpret_cast<Net*>(net), windows, n_windows, target_class, num_classes, epochs, lr,
                         dropout, batch_size, rng, mean_loss);
    SZ_CATCH
}

int szh_augment(const int16_t* pcm, uint64_t n_samples, float gain, float noise_level, uint64_t shift, const float* noise,
                uint64_t seed, int16_t* out)
{
    SZ_TRY
    if (n_samples && (!pcm || !out)) { set_error("null argument"); return SZH_EINVAL; }
    if (n_samples && shift >= n_samples) { set_error("shift must be below the sample count"); return SZH_EINVAL; }
    return augment_host(pcm, n_samples, gain, noise_level, shift, noise, seed, out);
    SZ_CATCH
}

int szh_pretrain_network(szh_net* net, szh_extractor* ext, const int16_t* pcm, uint64_t n_samples, uint32_t target_class,
                         uint32_t num_classes, uint32_t epochs, float lr, float dropout, uint32_t batch_size,
                         const szh_aug_draws* aug, const szh_rng* rng, float* mean_loss)
{
    SZ_TRY
    if (!net || !ext || !mean_loss || (n_samples && !pcm)) { set_error("null argument"); return SZH_EINVAL; }
    return pretrain_network_host(reinterpret_cast<Net*>(net), reinterpret_cast<Extractor*>(ext), pcm, n_samples, target_class,
                                 num_classes, epochs, lr, dropout, batch_size, aug, rng, mean_loss);
    SZ_CATCH
}

// normalize, lib.rs:132-139
static void normalize_vec(float* v, uint32_t d)
{
    float ss = 0.0f;
    for (uint32_t i = 0; i < d; ++i) ss += v[i] * v[i];
    const float norm = std::sqrt(ss);
    if (norm > 1e-6f)
        for (uint32_t i = 0; i < d; ++i) v[i] /= norm;
}

int szh_net_pooled_embedding(szh_net* net, const float* feats, uint64_t n_windows, int mode, float* out)
{
    SZ_TRY
    if (!net || !out || (n_windows && !feats) || mode < 0 || mode > 2) { set_error("bad argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    std::lock_guard<std::mutex> lock(n->mu);
    DevBuf d_feats;
    if (n_windows) {
        const int rc = d_feats.alloc(n_windows * n->in * sizeof(float));
        if (rc) return rc;
        SZ_HIP(hipMemcpy(d_feats.p, feats, n_windows * n->in * sizeof(float), hipMemcpyHostToDevice));
    }
    const int rc = net_pool_host(n, d_feats.as<float>(), n_windows, mode, out);
    if (rc) return rc;
    normalize_vec(out, n->h2);
    return SZH_OK;
    SZ_CATCH
}

int szh_extract_embedding(szh_net* net, szh_extractor* ext, const int16_t* pcm, uint64_t n_samples, float* out)
{
    SZ_TRY
    if (!net || !ext || !out || (n_samples && !pcm)) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    Extractor* e = reinterpret_cast<Extractor*>(ext);
    if (n->in != (uint32_t)kFeat) { set_error("extract_embedding needs a net with 60 inputs"); return SZH_EINVAL; }
    std::lock_guard<std::mutex> lock(n->mu);
    const uint64_t nw = szh_window_count(n_samples);
    DevBuf d_pcm, d_feat;
    Plan* plan = nullptr;
    struct PlanGuard { Plan*& p; ~PlanGuard() { plan_destroy(p); } } guard{plan};
    if (nw) {
        const uint64_t off = 0;
        int rc = plan_create(e, &off, &n_samples, 1, &plan);
        if (rc == SZH_OK) rc = d_pcm.alloc(n_samples * sizeof(int16_t));
        if (rc == SZH_OK) rc = d_feat.alloc(nw * kFeat * sizeof(float));
        if (rc) return rc;
        SZ_HIP(hipMemcpy(d_pcm.p, pcm, n_samples * sizeof(int16_t), hipMemcpyHostToDevice));
        rc = plan_run(plan, d_pcm.as<int16_t>(), d_feat.as<float>(), nullptr);
        if (rc) return rc;
    }
    const int rc = net_pool_host(n, d_feat.as<float>(), nw, 2, out);
    if (rc) return rc;
    normalize_vec(out, n->h2);
    return SZH_OK;
    SZ_CATCH
}

int szh_train_from_feature_map(szh_net* net, const float* const* feats, const uint64_t* n_windows,
                               const uint32_t* classes, uint32_t n_files, uint32_t epochs, float lr, float dropout,
                               uint32_t batch_size, uint64_t seed, float* mean_loss)
{
    SZ_TRY
    if (!net || !mean_loss || (n_files && (!feats || !n_windows || !classes))) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    float total = 0.0f;
    uint32_t count = 0;
    for (uint32_t f = 0; f < n_files; ++f) {
        if (!feats[f]) continue;  // path not in the feature map (lib.rs:644)
        szh_rng rng{nullptr, nullptr, mix64(seed + f)};
        float loss = 0.0f;
        const int rc = pretrain_host(n, feats[f], n_windows[f], classes[f], n->out, epochs, lr, dropout, batch_size,
                                     &rng, &loss);
        if (rc) return rc;
        total += loss;
        count += 1;
    }
    *mean_loss = count ? total / (float)count : 0.0f;
    return SZH_OK;
    SZ_CATCH
}

int szh_identify_from_features(szh_net* net, const float* feats, uint64_t n_windows, float threshold, uint32_t* ids,
                               uint32_t capacity, uint32_t* n_ids, uint64_t* counts)
{
    SZ_TRY
    if (!net || !n_ids || (n_windows && !feats)) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    if (n->out && !ids) { set_error("null argument"); return SZH_EINVAL; }
    std::lock_guard<std::mutex> lock(n->mu);
    DevBuf d_feats;
    if (n_windows) {
        const int rc = d_feats.alloc(n_windows * n->in * sizeof(float));
        if (rc) return rc;
        SZ_HIP(hipMemcpy(d_feats.p, feats, n_windows * n->in * sizeof(float), hipMemcpyHostToDevice));
    }
    return vote_features_dev(n, d_feats.as<float>(), n_windows, threshold, ids, capacity, n_ids, counts);
    SZ_CATCH
}

// Windows per tile of a long recording; recordings of up to two tiles go through in one piece.
// Through the one-kernel vote a tile has no intermediates to keep in L2, and larger tiles amortise the launches
// (100 h of device-resident PCM, M windows/s: 8192: 254, 65536: 296, 262144: 305) and the per-tile hand-over
// between the copy and the kernels of a host recording (10 h, GB/s of PCM: 8192: 40, 65536: 50).
static uint32_t tile_windows_default(bool fused, bool resident)
{
    const char* v = std::getenv("STREAMZ_TILE_WINDOWS");
    const long long t = v ? std::atoll(v) : 0;
    return t >= 4 && t <= (1ll << 24) ? (uint32_t)t : !fused ? 8192u : resident ? 262144u : 65536u;
}

// Votes and / or probability sums of a recording (host or device PCM), read back to the host.
static int tally_pcm(Net* n, Extractor* e, const int16_t* pcm, bool on_device, uint64_t n_samples, float threshold,
                     uint32_t tile, std::vector<unsigned long long>* counts, std::vector<double>* sums)
{
    if (n->in != (uint32_t)kFeat) { set_error("identify_speaker* needs a net with 60 inputs"); return SZH_EINVAL; }
    std::lock_guard<std::mutex> lock(n->mu);
    if (counts) counts->assign(n->out, 0);
    if (sums) sums->assign(n->out, 0.0);
    if (n->out == 0 || szh_window_count(n_samples) == 0) return SZH_OK;
    // votes and probability sums side by side in the net's staging buffer (the mutex is held): one clear, one read
    void* d_tally = nullptr;
    const size_t cnt_bytes = n->out * sizeof(unsigned long long), sum_bytes = n->out * sizeof(double);
    int rc = net_stage_dev(n, 3, cnt_bytes + sum_bytes, &d_tally);
    if (rc) return rc;
    unsigned long long* d_cnt = static_cast<unsigned long long*>(d_tally);
    double* d_sum = reinterpret_cast<double*>(static_cast<char*>(d_tally) + cnt_bytes);
    // cleared on the stream the first tile's kernels run on (identify_tiled takes the same one from the net)
    hipStream_t compute = nullptr;
    rc = net_keep_stream(n, &compute);
    if (rc) return rc;
    SZ_HIP(hipMemsetAsync(d_tally, 0, cnt_bytes + sum_bytes, compute));
    rc = identify_tiled(n, e, pcm, on_device, n_samples, threshold, tile, counts ? d_cnt : nullptr, sums ? d_sum : nullptr);
    if (rc) return rc;
    std::vector<unsigned char> host(cnt_bytes + sum_bytes);
    SZ_HIP(hipMemcpy(host.data(), d_tally, host.size(), hipMemcpyDeviceToHost));
    if (counts) std::memcpy(counts->data(), host.data(), cnt_bytes);
    if (sums) std::memcpy(sums->data(), host.data() + cnt_bytes, sum_bytes);
    return SZH_OK;
}

static int identify_pcm(Net* n, Extractor* e, const int16_t* pcm, bool on_device, uint64_t n_samples, float threshold,
                        uint32_t tile, uint32_t* ids, uint32_t capacity, uint32_t* n_ids, uint64_t* counts_out)
{
    if (n->out && !ids) { set_error("null argument"); return SZH_EINVAL; }
    if (capacity < n->out) { set_error("ids capacity %u < output size %u", capacity, n->out); return SZH_EINVAL; }
    std::vector<unsigned long long> counts;
    const int rc = tally_pcm(n, e, pcm, on_device, n_samples, threshold, tile, &counts, nullptr);
    if (rc) return rc;
    *n_ids = order_votes(counts, ids);
    if (counts_out)
        for (uint32_t i = 0; i < n->out; ++i) counts_out[i] = counts[i];
    return SZH_OK;
}

// windows per tile for a host recording of nw windows: up to two tiles go through in one piece
static uint32_t host_tile(const Net* n, uint64_t nw)
{
    const uint32_t tile = tile_windows_default(net_vote_fused(n), false);
    return nw <= 2ull * tile ? (uint32_t)std::max<uint64_t>(nw, 4) : tile;
}

// The reference's decision on per-class probability sums (lib.rs:1298-1302, 1332-1342): the
// sums are compared as f32, the last maximum wins.  The reference adds the probabilities and
// counts the windows sequentially in f32 (its count stops growing at 2^24 windows, 42 h of
// audio); here both are exact in f64 on purpose -- a parallel sum cannot reproduce a sequential
// f32 rounding order, and the mean of a recording longer than that stays a mean.
static uint32_t best_of_sums(const std::vector<double>& sums, float* sums_out)
{
    uint32_t best = 0;
    for (uint32_t j = 0; j < sums.size(); ++j) {
        if ((float)sums[j] >= (float)sums[best]) best = j;
        if (sums_out) sums_out[j] = (float)sums[j];
    }
    return best;
}
static int64_t thresholded(const std::vector<double>& sums, uint64_t n_windows, float threshold, float* sums_out)
{
    const uint32_t best = best_of_sums(sums, sums_out);
    if (sums.size() <= 1 || n_windows == 0) return -1;  // lib.rs:1316-1318, 1329-1331
    return (float)(sums[best] / (double)n_windows) >= threshold ? (int64_t)best : -1;
}

int szh_identify_speaker_list(szh_net* net, szh_extractor* ext, const int16_t* pcm, uint64_t n_samples,
                              float threshold, uint32_t* ids, uint32_t capacity, uint32_t* n_ids)
{
    SZ_TRY
    if (!net || !ext || !n_ids || (n_samples && !pcm)) { set_error("null argument"); return SZH_EINVAL; }
    return identify_pcm(reinterpret_cast<Net*>(net), reinterpret_cast<Extractor*>(ext), pcm, false, n_samples, threshold,
                        host_tile(reinterpret_cast<Net*>(net), szh_window_count(n_samples)), ids, capacity, n_ids, nullptr);
    SZ_CATCH
}

int szh_identify_speaker(szh_net* net, szh_extractor* ext, const int16_t* pcm, uint64_t n_samples, uint32_t* id,
                         float* sums)
{
    SZ_TRY
    if (!net || !ext || !id || (n_samples && !pcm)) { set_error("null argument"); return SZH_EINVAL; }
    std::vector<double> s;
    const int rc = tally_pcm(reinterpret_cast<Net*>(net), reinterpret_cast<Extractor*>(ext), pcm, false, n_samples, 0.0f,
                             host_tile(reinterpret_cast<Net*>(net), szh_window_count(n_samples)), nullptr, &s);
    if (rc) return rc;
    *id = best_of_sums(s, sums);
    return SZH_OK;
    SZ_CATCH
}

int szh_identify_speaker_with_threshold(szh_net* net, szh_extractor* ext, const int16_t* pcm, uint64_t n_samples,
                                        float threshold, int64_t* id)
{
    SZ_TRY
    if (!net || !ext || !id || (n_samples && !pcm)) { set_error("null argument"); return SZH_EINVAL; }
    std::vector<double> s;
    const uint64_t nw = szh_window_count(n_samples);
    const int rc = tally_pcm(reinterpret_cast<Net*>(net), reinterpret_cast<Extractor*>(ext), pcm, false, n_samples, 0.0f,
                             host_tile(reinterpret_cast<Net*>(net), nw), nullptr, &s);
    if (rc) return rc;
    *id = thresholded(s, nw, threshold, nullptr);
    return SZH_OK;
    SZ_CATCH
}

int szh_identify_speaker_with_threshold_feats(szh_net* net, const float* feats, uint64_t n_windows, float threshold,
                                              int64_t* id, float* sums)
{
    SZ_TRY
    if (!net || !id || (n_windows && !feats)) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    std::vector<double> s(n->out, 0.0);
    if (n->out && n_windows) {
        std::lock_guard<std::mutex> lock(n->mu);
        DevBuf d_feats, d_sum;
        int rc = d_feats.alloc(n_windows * n->in * sizeof(float));
        if (rc == SZH_OK) rc = d_sum.alloc(n->out * sizeof(double));
        if (rc) return rc;
        SZ_HIP(hipMemcpy(d_feats.p, feats, n_windows * n->in * sizeof(float), hipMemcpyHostToDevice));
        SZ_HIP(hipMemset(d_sum.p, 0, n->out * sizeof(double)));
        rc = net_vote_dev(n, d_feats.as<float>(), n_windows, 0.0f, nullptr, d_sum.as<double>(), nullptr);
        if (rc) return rc;
        SZ_HIP(hipStreamSynchronize(nullptr));
        SZ_HIP(hipMemcpy(s.data(), d_sum.p, n->out * sizeof(double), hipMemcpyDeviceToHost));
    }
    *id = thresholded(s, n_windows, threshold, sums);
    return SZH_OK;
    SZ_CATCH
}

int szh_identify_speaker_list_dev(szh_net* net, szh_extractor* ext, const int16_t* pcm_dev, uint64_t n_samples,
                                  float threshold, uint32_t tile_windows, uint32_t* ids, uint32_t capacity,
                                  uint32_t* n_ids, uint64_t* counts)
{
    SZ_TRY
    if (!net || !ext || !n_ids || (n_samples && !pcm_dev)) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    return identify_pcm(n, reinterpret_cast<Extractor*>(ext), pcm_dev, true, n_samples, threshold,
                        tile_windows ? tile_windows : tile_windows_default(net_vote_fused(n), true), ids, capacity, n_ids, counts);
    SZ_CATCH
}

// ---- device-resident building blocks ----------------------------------------------------
// These calls enqueue work with pointers of the device the handle was created on: the calling
// thread must have that device current (szh_init), or the kernels would run elsewhere.
static int check_device(int handle_device)
{
    int cur = -1;
    SZ_HIP(hipGetDevice(&cur));
    if (cur != handle_device) {
        set_error("handle belongs to device %d but the calling thread's current device is %d (call szh_init first)", handle_device, cur);
        return SZH_EINVAL;
    }
    return SZH_OK;
}

uint64_t szh_net_grad_len(const szh_net* net) { return net ? reinterpret_cast<const Net*>(net)->param_count() + 2 : 0; }
float* szh_net_grad_dev(szh_net* net) { return net ? reinterpret_cast<Net*>(net)->d_grad : nullptr; }
int szh_net_set_gemm(szh_net* net, int mode)
{
    if (!net || (mode != SZH_GEMM_F32 && mode != SZH_GEMM_BF16X3)) { set_error("unknown GEMM mode"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    std::lock_guard<std::mutex> lock(n->mu);
    n->gemm_mode = mode;
    return SZH_OK;
}
int szh_net_get_gemm(const szh_net* net) { return net ? reinterpret_cast<const Net*>(net)->gemm_mode : SZH_GEMM_F32; }
int szh_net_zero_grad_dev(szh_net* net, void* stream)
{
    if (!net) { set_error("null argument"); return SZH_EINVAL; }
    if (int rc = check_device(reinterpret_cast<Net*>(net)->device)) return rc;
    return net_zero_grad(reinterpret_cast<Net*>(net), static_cast<hipStream_t>(stream));
}
int szh_net_accumulate_dev(szh_net* net, const float* feats_dev, const int32_t* labels_dev, uint64_t n, float dropout,
                           uint64_t seed, void* stream)
{
    SZ_TRY
    if (!net || (n && (!feats_dev || !labels_dev))) { set_error("null argument"); return SZH_EINVAL; }
    if (int rc = check_device(reinterpret_cast<Net*>(net)->device)) return rc;
    AccumArgs a;
    a.x_dev = feats_dev;
    a.rows = n;
    a.labels_dev = labels_dev;
    a.dropout = dropout;
    a.seed = seed;
    a.skip_zero_rows = true;
    return net_accumulate(reinterpret_cast<Net*>(net), a, static_cast<hipStream_t>(stream));
    SZ_CATCH
}
int szh_net_train_epochs_sharded(szh_net* net, szh_comm* comm, const float* feats, const int32_t* labels, uint64_t n,
                                 uint32_t epochs, float lr, float dropout, uint64_t seed, float* mean_loss)
{
    SZ_TRY
    if (!net || (n && (!feats || !labels))) { set_error("null argument"); return SZH_EINVAL; }
    Net* nn = reinterpret_cast<Net*>(net);
    if (int rc = check_device(nn->device)) return rc;
    std::lock_guard<std::mutex> lock(nn->mu);
    DevBuf d_x, d_l;
    if (int rc = d_x.alloc(n * kFeat * sizeof(float))) return rc;
    if (int rc = d_l.alloc(n * sizeof(int32_t))) return rc;
    if (n) {
        SZ_HIP(hipMemcpy(d_x.p, feats, n * kFeat * sizeof(float), hipMemcpyHostToDevice));
        SZ_HIP(hipMemcpy(d_l.p, labels, n * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    hipStream_t s = nullptr;
    float tail[2] = {0.0f, 0.0f};
    for (uint32_t ep = 0; ep < epochs; ++ep) {
        if (int rc = net_zero_grad(nn, s)) return rc;
        AccumArgs a;
        a.x_dev = d_x.as<float>();
        a.rows = n;
        a.labels_dev = d_l.as<int32_t>();
        a.dropout = dropout;
        a.seed = seed + ep;
        a.skip_zero_rows = true;
        if (int rc = net_accumulate(nn, a, s)) return rc;
        if (comm)
            if (int rc = szh_net_allreduce_grad_dev(net, comm, s)) return rc;
        if (ep + 1 == epochs) SZ_HIP(hipMemcpyAsync(tail, nn->d_grad + nn->param_count(), sizeof(tail), hipMemcpyDeviceToHost, s));
        if (int rc = net_apply(nn, lr, s)) return rc;
    }
    SZ_HIP(hipStreamSynchronize(s));
    if (mean_loss) *mean_loss = tail[1] > 0.0f ? tail[0] / tail[1] : 0.0f;
    return SZH_OK;
    SZ_CATCH
}
int szh_net_apply_grad_dev(szh_net* net, float lr, void* stream)
{
    if (!net) { set_error("null argument"); return SZH_EINVAL; }
    if (int rc = check_device(reinterpret_cast<Net*>(net)->device)) return rc;
    return net_apply(reinterpret_cast<Net*>(net), lr, static_cast<hipStream_t>(stream));
}
int szh_net_forward_dev(szh_net* net, const float* feats_dev, uint64_t n, float* probs_dev, void* stream)
{
    SZ_TRY
    if (!net || (n && (!feats_dev || !probs_dev))) { set_error("null argument"); return SZH_EINVAL; }
    if (int rc = check_device(reinterpret_cast<Net*>(net)->device)) return rc;
    return net_forward_dev(reinterpret_cast<Net*>(net), feats_dev, n, probs_dev, kFwdProbs, static_cast<hipStream_t>(stream));
    SZ_CATCH
}
int szh_net_vote_dev(szh_net* net, const float* feats_dev, uint64_t n, float threshold, unsigned long long* counts_dev,
                     void* stream)
{
    SZ_TRY
    if (!net || (n && (!feats_dev || !counts_dev))) { set_error("null argument"); return SZH_EINVAL; }
    if (int rc = check_device(reinterpret_cast<Net*>(net)->device)) return rc;
    return net_vote_dev(reinterpret_cast<Net*>(net), feats_dev, n, threshold, counts_dev, nullptr, static_cast<hipStream_t>(stream));
    SZ_CATCH
}

// ---- metadata ---------------------------------------------------------------------------

int szh_net_record_training_file(szh_net* net, uint32_t speaker, const char* path)
{
    SZ_TRY
    if (!net || !path) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    if (n->file_lists.size() <= speaker) n->file_lists.resize((size_t)speaker + 1);  // lib.rs:856-858
    auto& l = n->file_lists[speaker];
    if (std::find(l.begin(), l.end(), std::string(path)) == l.end()) l.emplace_back(path);
    return SZH_OK;
    SZ_CATCH
}
int64_t szh_net_file_list(const szh_net* net, uint32_t speaker, char* buf, uint64_t capacity)
{
    if (!net) { set_error("null argument"); return SZH_EINVAL; }
    const Net* n = reinterpret_cast<const Net*>(net);
    std::string joined;
    if (speaker < n->file_lists.size())
        for (size_t i = 0; i < n->file_lists[speaker].size(); ++i) {
            if (i) joined += "\n";
            joined += n->file_lists[speaker][i];
        }
    if (buf && capacity) {
        const size_t c = std::min<size_t>(joined.size(), (size_t)capacity - 1);
        std::memcpy(buf, joined.data(), c);
        buf[c] = 0;
    }
    return (int64_t)joined.size();
}
int szh_net_set_dataset_specs(szh_net* net, uint32_t sample_rate, uint16_t bits)
{
    if (!net) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    n->sample_rate = sample_rate;
    n->bits = bits;
    return SZH_OK;
}
int szh_net_set_embeddings(szh_net* net, const float* vectors, const float* mean_sims, const float* std_sims,
                           uint32_t cnt, uint32_t dim)
{
    SZ_TRY
    if (!net || (cnt && (!vectors || !mean_sims || !std_sims))) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    n->emb_dim = cnt ? dim : 0;
    n->emb_vectors.assign(vectors, vectors + (size_t)cnt * dim);
    n->emb_mean.assign(mean_sims, mean_sims + cnt);
    n->emb_std.assign(std_sims, std_sims + cnt);
    return SZH_OK;
    SZ_CATCH
}
int szh_net_get_embeddings(const szh_net* net, float* vectors, float* mean_sims, float* std_sims, uint32_t* cnt,
                           uint32_t* dim)
{
    if (!net || !cnt || !dim) { set_error("null argument"); return SZH_EINVAL; }
    const Net* n = reinterpret_cast<const Net*>(net);
    *cnt = (uint32_t)n->emb_mean.size();
    *dim = n->emb_dim;
    if (vectors) std::copy(n->emb_vectors.begin(), n->emb_vectors.end(), vectors);
    if (mean_sims) std::copy(n->emb_mean.begin(), n->emb_mean.end(), mean_sims);
    if (std_sims) std::copy(n->emb_std.begin(), n->emb_std.end(), std_sims);
    return SZH_OK;
}

// ---- persistence ------------------------------------------------------------------------

int szh_net_save(szh_net* net, const char* path)
{
    SZ_TRY
    if (!net || !path) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    std::vector<float> p(n->param_count());
    int rc = net_get_params(n, p.data());
    if (rc) return rc;
    NpzWriter z;
    z.add("w1", "<f4", {n->in, n->h1}, p.data() + n->off_w1(), (size_t)n->in * n->h1 * 4);
    z.add("b1", "<f4", {n->h1}, p.data() + n->off_b1(), (size_t)n->h1 * 4);
    z.add("w2", "<f4", {n->h1, n->h2}, p.data() + n->off_w2(), (size_t)n->h1 * n->h2 * 4);
    z.add("b2", "<f4", {n->h2}, p.data() + n->off_b2(), (size_t)n->h2 * 4);
    const int64_t sr = n->sample_rate, bits = n->bits, ns = n->out;
    z.add("sample_rate", "<i8", {1}, &sr, 8);
    z.add("bits", "<i8", {1}, &bits, 8);
    z.add("num_speakers", "<i8", {1}, &ns, 8);
    std::vector<float> col(n->h2);
    for (uint32_t c = 0; c < n->out; ++c) {  // one column and one bias per class, 1-based names
        for (uint32_t r = 0; r < n->h2; ++r) col[r] = p[n->off_w3() + (uint64_t)r * n->out + c];
        z.add("w3_" + std::to_string(c + 1), "<f4", {n->h2}, col.data(), (size_t)n->h2 * 4);
        z.add("b3_" + std::to_string(c + 1), "<f4", {1}, &p[n->off_b3() + c], 4);
    }
    for (size_t c = 0; c < n->b4.size(); ++c) {
        z.add("w4_" + std::to_string(c + 1), "<f4", {n->w4_cols[c].size()}, n->w4_cols[c].data(), n->w4_cols[c].size() * 4);
        z.add("b4_" + std::to_string(c + 1), "<f4", {1}, &n->b4[c], 4);
    }
    for (uint32_t c = 0; c < n->out; ++c) {  // lib.rs:1109-1113
        std::string joined;
        if (c < n->file_lists.size())
            for (size_t i = 0; i < n->file_lists[c].size(); ++i) {
                if (i) joined += "\n";
                joined += n->file_lists[c][i];
            }
        z.add("speaker_" + std::to_string(c) + "_files", "|u1", {joined.size()}, joined.data(), joined.size());
    }
    if (!n->emb_mean.empty()) {
        z.add("speaker_embeddings", "<f4", {n->emb_mean.size(), n->emb_dim}, n->emb_vectors.data(), n->emb_vectors.size() * 4);
        z.add("speaker_mean_sims", "<f4", {n->emb_mean.size()}, n->emb_mean.data(), n->emb_mean.size() * 4);
        z.add("speaker_std_sims", "<f4", {n->emb_std.size()}, n->emb_std.data(), n->emb_std.size() * 4);
    }
    std::string err;
    if (!z.finish(path, err)) { set_error("%s", err.c_str()); return SZH_EIO; }
    return SZH_OK;
    SZ_CATCH
}

int szh_net_load(const char* path, szh_net** net)
{
    SZ_TRY
    if (!net || !path) { set_error("null argument"); return SZH_EINVAL; }
    std::vector<std::pair<std::string, NpyArray>> members;
    std::string err;
    if (!npz_read(path, members, err)) { set_error("%s", err.c_str()); return SZH_EIO; }
    std::map<std::string, const NpyArray*> by;
    for (auto& m : members) by[m.first] = &m.second;
    auto need = [&](const char* name, const char* descr, size_t ndim) -> const NpyArray* {
        auto it = by.find(name);
        if (it == by.end()) { set_error("%s: member %s missing", path, name); return nullptr; }
        if (it->second->descr != descr || it->second->shape.size() != ndim || it->second->fortran_order) {
            set_error("%s: member %s has dtype %s / %zu dims, expected %s / %zu", path, name,
                      it->second->descr.c_str(), it->second->shape.size(), descr, ndim);
            return nullptr;
        }
        return it->second;
    };
    const NpyArray *w1 = need("w1", "<f4", 2), *b1 = need("b1", "<f4", 1), *w2 = need("w2", "<f4", 2),
                   *b2 = need("b2", "<f4", 1), *sr = need("sample_rate", "<i8", 1), *bits = need("bits", "<i8", 1);
    if (!w1 || !b1 || !w2 || !b2 || !sr || !bits) return SZH_EIO;
    const uint32_t in = (uint32_t)w1->shape[0], h1 = (uint32_t)w1->shape[1], h2 = (uint32_t)w2->shape[1];
    if (w1->shape[0] == 0 || w1->shape[0] > 0xffffffffull || w1->shape[1] == 0 || w1->shape[1] > 0xffffffffull ||
        w2->shape[1] == 0 || w2->shape[1] > 0xffffffffull || w2->shape[0] != h1 || b1->shape[0] != h1 || b2->shape[0] != h2 ||
        w1->data.size() != (size_t)in * h1 * 4 || b1->data.size() != (size_t)h1 * 4 || w2->data.size() != (size_t)h1 * h2 * 4 ||
        b2->data.size() != (size_t)h2 * 4 || sr->count() < 1 || bits->count() < 1) {
        set_error("%s: inconsistent layer shapes", path);
        return SZH_EIO;
    }
    // per-class columns w3_i / b3_i (1-based) until the first gap (lib.rs:1149-1166)
    std::vector<const NpyArray*> cols, biases;
    for (uint32_t i = 1;; ++i) {
        auto iw = by.find("w3_" + std::to_string(i)), ib = by.find("b3_" + std::to_string(i));
        if (iw == by.end() || ib == by.end()) break;
        if (iw->second->descr != "<f4" || iw->second->count() != h2 || ib->second->descr != "<f4" || ib->second->count() < 1) {
            set_error("%s: bad output column %u", path, i);
            return SZH_EIO;
        }
        cols.push_back(iw->second);
        biases.push_back(ib->second);
    }
    uint32_t stored = (uint32_t)cols.size();
    const NpyArray *w3m = nullptr, *b3m = nullptr;
    if (cols.empty() && by.count("w3")) {  // legacy monolithic layout (lib.rs:1199-1207)
        w3m = need("w3", "<f4", 2);
        b3m = need("b3", "<f4", 1);
        if (!w3m || !b3m || w3m->shape[0] != h2 || w3m->shape[1] < b3m->shape[0]) { if (w3m && b3m) set_error("%s: bad w3/b3", path); return SZH_EIO; }
        stored = (uint32_t)b3m->shape[0];
    }
    uint32_t outputs = stored;
    if (by.count("num_speakers")) {
        const NpyArray* nsp = need("num_speakers", "<i8", 1);
        if (!nsp || nsp->count() < 1) return SZH_EIO;
        const int64_t v = as<int64_t>(*nsp)[0];
        if (v < 0 || (uint64_t)v > stored) { set_error("%s: num_speakers %lld but %u stored columns", path, (long long)v, stored); return SZH_EIO; }
        outputs = (uint32_t)v;
    }
    Net tmp_layout;
    tmp_layout.in = in; tmp_layout.h1 = h1; tmp_layout.h2 = h2; tmp_layout.out = outputs;
    std::vector<float> p(tmp_layout.param_count(), 0.0f);
    std::memcpy(p.data() + tmp_layout.off_w1(), w1->data.data(), (size_t)in * h1 * 4);
    std::memcpy(p.data() + tmp_layout.off_b1(), b1->data.data(), (size_t)h1 * 4);
    std::memcpy(p.data() + tmp_layout.off_w2(), w2->data.data(), (size_t)h1 * h2 * 4);
    std::memcpy(p.data() + tmp_layout.off_b2(), b2->data.data(), (size_t)h2 * 4);
    for (uint32_t c = 0; c < outputs; ++c) {
        for (uint32_t r = 0; r < h2; ++r)
            p[tmp_layout.off_w3() + (uint64_t)r * outputs + c] =
                w3m ? as<float>(*w3m)[(uint64_t)r * w3m->shape[1] + c] : as<float>(*cols[c])[r];
        p[tmp_layout.off_b3() + c] = b3m ? as<float>(*b3m)[c] : as<float>(*biases[c])[0];
    }
    Net* n = nullptr;
    const int rc = net_create(in, h1, h2, outputs, p.data(), 0, &n);
    if (rc) return rc;
    n->sample_rate = (uint32_t)as<int64_t>(*sr)[0];
    n->bits = (uint16_t)as<int64_t>(*bits)[0];
    for (uint32_t i = 1;; ++i) {  // optional encoding layer, carried through unchanged
        auto iw = by.find("w4_" + std::to_string(i)), ib = by.find("b4_" + std::to_string(i));
        if (iw == by.end() || ib == by.end() || iw->second->descr != "<f4" || ib->second->descr != "<f4" || ib->second->count() < 1) break;
        n->w4_cols.emplace_back(as<float>(*iw->second), as<float>(*iw->second) + iw->second->count());
        n->b4.push_back(as<float>(*ib->second)[0]);
    }
    n->file_lists.assign(outputs, {});
    for (uint32_t c = 0; c < outputs; ++c) {
        auto it = by.find("speaker_" + std::to_string(c) + "_files");
        if (it == by.end() || it->second->descr != "|u1") continue;
        const std::string text(reinterpret_cast<const char*>(it->second->data.data()), it->second->data.size());
        size_t pos = 0;
        while (pos < text.size()) {  // str::lines
            size_t nl = text.find('\n', pos);
            if (nl == std::string::npos) nl = text.size();
            std::string line = text.substr(pos, nl - pos);
            if (!line.empty() && line.back() == '\r') line.pop_back();
            n->file_lists[c].push_back(line);
            pos = nl + 1;
        }
    }
    if (by.count("speaker_embeddings") && by.count("speaker_mean_sims") && by.count("speaker_std_sims")) {
        const NpyArray *e = need("speaker_embeddings", "<f4", 2), *m = need("speaker_mean_sims", "<f4", 1),
                       *s = need("speaker_std_sims", "<f4", 1);
        if (!e || !m || !s || m->shape[0] != e->shape[0] || s->shape[0] != e->shape[0]) { net_destroy(n); return SZH_EIO; }
        n->emb_dim = (uint32_t)e->shape[1];
        n->emb_vectors.assign(as<float>(*e), as<float>(*e) + e->count());
        n->emb_mean.assign(as<float>(*m), as<float>(*m) + m->count());
        n->emb_std.assign(as<float>(*s), as<float>(*s) + s->count());
    }
    *net = reinterpret_cast<szh_net*>(n);
    return SZH_OK;
    SZ_CATCH
}

int szh_wav_read(const char* path, int16_t** pcm, uint64_t* n_samples)
{
    SZ_TRY
    if (!path || !pcm || !n_samples) { set_error("null argument"); return SZH_EINVAL; }
    std::vector<int16_t> mono;
    const int rc = load_audio_mono(path, mono);
    if (rc) return rc;
    int16_t* out = static_cast<int16_t*>(std::malloc(mono.size() ? mono.size() * 2 : 2));
    if (!out) { set_error("out of host memory"); return SZH_ENOMEM; }
    if (!mono.empty()) std::memcpy(out, mono.data(), mono.size() * 2);
    *pcm = out;
    *n_samples = mono.size();
    return SZH_OK;
    SZ_CATCH
}

int szh_wav_info(const char* path, uint64_t* n_frames, uint32_t* sample_rate, uint16_t* channels)
{
    SZ_TRY
    if (!path || !n_frames) { set_error("null argument"); return SZH_EINVAL; }
    if (!ends_with_ci(path, ".wav")) { set_error("Unsupported format: %s (only 16-bit PCM WAV is handled on this path)", path); return SZH_EFORMAT; }
    WavInfo info;
    uint64_t frames = 0;
    std::string err;
    if (!wav_probe(path, info, frames, err)) { set_error("%s", err.c_str()); return SZH_EIO; }
    *n_frames = frames;
    if (sample_rate) *sample_rate = info.sample_rate;
    if (channels) *channels = info.channels;
    return SZH_OK;
    SZ_CATCH
}

int szh_shard_ranges(const uint64_t* n_windows, uint32_t n_files, uint32_t world, uint32_t* out)
{
    if (!out || world == 0 || (n_files && !n_windows)) { set_error("bad argument"); return SZH_EINVAL; }
    typedef unsigned __int128 u128;
    u128 total = 0;
    for (uint32_t f = 0; f < n_files; ++f) total += n_windows[f];
    out[0] = 0;
    uint32_t b = 0;   // boundary of the previous rank
    u128 prefix = 0;  // windows of files [0, b)
    for (uint32_t r = 1; r < world; ++r) {
        if (total == 0) { out[r] = (uint32_t)((uint64_t)n_files * r / world); continue; }  // nothing to balance on: by count
        const u128 target = total * r;  // compared with prefix * world
        // The prefix sums do not decrease, so the nearest one is the last below the target or the first at
        // or above it.  `low` is the first index (from b on) that has the last prefix sum below the target.
        uint32_t i = b, low = b;
        u128 p = prefix, p_low = prefix;
        bool have_low = false;
        while (p * world < target) {  // ends by i == n_files: the whole list's sum times world is >= target
            if (!have_low || p != p_low) { low = i; p_low = p; have_low = true; }
            p += n_windows[i];
            ++i;
        }
        if (have_low && target - p_low * world <= p * world - target) { i = low; p = p_low; }
        b = i;
        prefix = p;
        out[r] = b;
    }
    out[world] = n_files;
    return SZH_OK;
}

int szh_load_cached_features(const char* path, szh_extractor* ext, float** feats, uint64_t* n_windows)
{
    SZ_TRY
    if (!path || !ext || !feats || !n_windows) { set_error("null argument"); return SZH_EINVAL; }
    const std::string cache = feature_cache_path(path);
    std::string err;
    if (file_exists(cache)) {  // lib.rs:563-570
        NpyArray a;
        if (!npy_read_file(cache, a, err)) { set_error("%s", err.c_str()); return SZH_EIO; }
        // the reference's read_npy into Array2<f32> takes any (n, m); its consumers then fail on rows that
        // are not 60 wide.  Here the mismatch is reported at once instead of handing out a short buffer.
        if (a.descr != "<f4" || a.shape.size() != 2 || a.fortran_order || a.shape[1] != (uint64_t)kFeat ||
            a.data.size() != a.shape[0] * (uint64_t)kFeat * sizeof(float)) {
            set_error("%s: not a C-order (n, %d) float32 array", cache.c_str(), kFeat);
            return SZH_EFORMAT;
        }
        float* out = static_cast<float*>(std::malloc(a.data.size() ? a.data.size() : 4));
        if (!out) { set_error("out of host memory"); return SZH_ENOMEM; }
        std::memcpy(out, a.data.data(), a.data.size());
        *feats = out;
        *n_windows = a.shape[0];
        return SZH_OK;
    }
    std::vector<int16_t> mono;
    int rc = load_audio_mono(path, mono);
    if (rc) return rc;
    const uint64_t nw = szh_window_count(mono.size());
    float* out = static_cast<float*>(std::malloc(nw ? nw * kFeat * sizeof(float) : 4));
    if (!out) { set_error("out of host memory"); return SZH_ENOMEM; }
    uint64_t got = 0;
    rc = szh_extract(ext, mono.data(), mono.size(), out, nw, &got);
    if (rc) { std::free(out); return rc; }
    if (nw > 0) (void)npy_write_file(cache, "<f4", {nw, (uint64_t)kFeat}, out, nw * kFeat * sizeof(float), err);  // lib.rs:573-577
    *feats = out;
    *n_windows = nw;
    return SZH_OK;
    SZ_CATCH
}

}  // extern "C"

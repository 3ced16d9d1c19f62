// identify_speaker_list (lib.rs:1383-1411) over a recording that is too long to treat as one
// batch: the windows are processed in tiles of a fixed number of windows.  Every tile runs
// the same launch sequence (MFCCs of the tile plus two neighbour windows each side, deltas
// and z-score, three layers, softmax, vote), so the sequence is captured once into a HIP
// graph and replayed (the plan's move to the next tile is part of its last extraction kernel); the
// votes accumulate in device memory and are read once at the end.
//
// The deltas of a window need the MFCCs of windows t-2 .. t+2 (add_deltas applied twice,
// lib.rs:321-322), so a tile recomputes the MFCCs of two windows of each neighbour tile
// instead of keeping state between tiles.  Edge replication (lib.rs:215-220) happens only in
// the first and the last tile, which are launched directly.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "internal.h"
#include "prof.h"
#include "streamz_hip.h"

namespace sz {

namespace {

constexpr int kMaxLanes = 4;   // tiles of device-resident PCM kept in flight at once
// Replays queued per lane before the host waits for the oldest of them.  The queue only has to cover
// the host's launch jitter (a replay is tens of microseconds of GPU work, so 16 of them are about a
// millisecond); it is kept short on purpose: with the whole recording queued at once (4 843 replays for
// 100 h) processes that STARTED on the same GPU meanwhile crashed in their first kernel launch on this
// image -- seen in an earlier round, cause not established, so nothing here relies on a "safe" depth
// having been measured.
constexpr int kReplayDepth = 16;
// host recordings of at least kStagedCopyBytes are copied through kRingBytes of pinned memory in pieces of kPieceBytes
constexpr size_t kStagedCopyBytes = 32u << 20, kRingBytes = 32u << 20, kPieceBytes = 4u << 20;

struct Resources {
    hipStream_t compute = nullptr, copy = nullptr;
    // device-resident PCM: interior tiles go round-robin to `lanes` streams, each with its own plan,
    // feature rows, range of workspace rows and captured graph, so that a tile too small to fill the
    // chip overlaps with its neighbours
    hipStream_t lane_stream[kMaxLanes] = {};
    hipEvent_t lane_done[kMaxLanes][kReplayDepth] = {};   // ring of "replay finished" events per lane
    Plan* lane_plan[kMaxLanes] = {};
    hipGraph_t lane_graph[kMaxLanes] = {};
    hipGraphExec_t lane_exec[kMaxLanes] = {};
    Net* net = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t exec[2] = {nullptr, nullptr};
    Plan *first = nullptr, *mid = nullptr, *last = nullptr;
    void* feat = nullptr;
    void* pcm[2] = {nullptr, nullptr};
    bool kept_feat = false, kept_pcm = false;  // these belong to the net (Net::IdentifyKeep), as `compute` does
    HostStager stager;                         // long host recordings: copies staged through the net's pinned buffer
    ~Resources()
    {
        stager.finish();
        if (compute) (void)hipStreamSynchronize(compute);
        if (copy) (void)hipStreamSynchronize(copy);
        for (int i = 0; i < kMaxLanes; ++i) {
            if (lane_stream[i]) (void)hipStreamSynchronize(lane_stream[i]);
            for (int k = 0; k < kReplayDepth; ++k)
                if (lane_done[i][k]) (void)hipEventDestroy(lane_done[i][k]);
            if (lane_exec[i]) (void)hipGraphExecDestroy(lane_exec[i]);
            if (lane_graph[i]) (void)hipGraphDestroy(lane_graph[i]);
            plan_destroy(lane_plan[i]);
            if (lane_stream[i] && lane_stream[i] != compute) (void)hipStreamDestroy(lane_stream[i]);
        }
        if (net) net->ws.row_base = 0;
        for (int i = 0; i < 2; ++i) {
            if (exec[i]) (void)hipGraphExecDestroy(exec[i]);
            if (graph[i]) (void)hipGraphDestroy(graph[i]);
            if (ready[i]) (void)hipEventDestroy(ready[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
            if (pcm[i] && !kept_pcm) (void)hipFree(pcm[i]);
        }
        plan_destroy(first);
        plan_destroy(mid);
        plan_destroy(last);
        if (feat && !kept_feat) (void)hipFree(feat);
        if (copy) (void)hipStreamDestroy(copy);
    }
};

// samples covered by `windows` consecutive windows
uint64_t span(uint64_t windows) { return windows ? (windows - 1) * kHop + kWindow : 0; }

struct Tally {
    float threshold;
    unsigned long long* counts_dev;  // votes (optional)
    double* psums_dev;               // probability sums (optional)
};

int run_tile(Net* n, Plan* plan, const int16_t* pcm_dev, float* feat, uint64_t rows, const Tally& t, hipStream_t s)
{
    int rc = plan_run(plan, pcm_dev, feat, s);
    if (rc) return rc;
    return net_vote_dev(n, feat, rows, t.threshold, t.counts_dev, t.psums_dev, s, true);
}

// Captures run_tile on stream s.
int capture_tile(Resources& r, int slot, Net* n, const int16_t* pcm_dev, uint32_t rows, const Tally& tally)
{
    SZ_HIP(hipStreamBeginCapture(r.compute, hipStreamCaptureModeRelaxed));
    int rc = run_tile(n, r.mid, pcm_dev, static_cast<float*>(r.feat), rows, tally, r.compute);
    const hipError_t he = hipStreamEndCapture(r.compute, &r.graph[slot]);
    if (rc) return rc;
    if (he != hipSuccess) return hip_fail(he, "hipStreamEndCapture");
    SZ_HIP(hipGraphInstantiate(&r.exec[slot], r.graph[slot], nullptr, nullptr, 0));
    return SZH_OK;
}

}  // namespace

int identify_tiled(Net* n, Extractor* e, const int16_t* pcm, bool pcm_on_device, uint64_t n_samples, float threshold,
                   uint32_t tile_windows, unsigned long long* counts_dev, double* psums_dev)
{
    const Tally tally{threshold, counts_dev, psums_dev};
    const uint64_t nw = szh_window_count(n_samples);
    if (nw == 0 || n->out == 0) return SZH_OK;
    const uint64_t B = tile_windows;
    if (B < 4) { set_error("tile_windows must be at least 4"); return SZH_EINVAL; }
    if (pcm_on_device && ((uintptr_t)pcm & 15) != 0) { set_error("device PCM must be 16-byte aligned"); return SZH_EINVAL; }
    Resources r;
    if (pcm_on_device) SZ_HIP(hipDeviceSynchronize());  // the caller's work on other streams produced pcm (the tallies were cleared on the net's compute stream)
    // the net's own stream (every caller holds the net's mutex).  A second one, for the copies of a long host
    // recording, is made per call where it is needed: a stream that is kept holds on to one of the process's
    // four hardware queues, and the lanes below want one queue each.
    int rc = net_keep_stream(n, &r.compute);
    if (rc) return rc;
    const bool single = nw <= B + 2;
    // tile layout: [first: B] [mid: B]* [last: 1 .. B+2]
    const uint64_t n_mid = single ? 0 : (nw - B - 3) / B;
    const uint64_t last_first = B + n_mid * B;            // first window of the last tile
    const uint64_t last_rows = single ? 0 : nw - last_first;
    // rows of features and of network workspace: one range of B rows per tile in flight
    uint64_t max_lanes = 1;
    if (pcm_on_device && !single && n_mid >= 2 && !prof_enabled()) {
        const char* env_v = std::getenv("STREAMZ_TILE_LANES");
        const int env_lanes = env_v ? std::atoi(env_v) : 0;
        // three for small tiles: with four, the lanes' streams compete for the process's four hardware queues
        // (100 h in tiles of 8192 windows: 0.38 s with one lane, 0.24 with two, 0.21 with three, 0.25 with four);
        // two for tiles whose kernels fill the chip by themselves (65536 windows: 0.149 s with two, 0.152 with three)
        max_lanes = env_lanes > 0 ? (uint64_t)std::min(env_lanes, kMaxLanes) : B < 32768 ? 3 : B <= 262144 ? 2 : 1;
    }
    uint64_t feat_rows = single ? nw : std::max<uint64_t>(B * max_lanes, last_rows);
    // (the one-kernel vote keeps no activations in memory: its net needs no rows of workspace beyond the minimum)
    const bool fused = net_vote_fused(n);
    rc = net_reserve_rows(n, fused ? 256 : feat_rows);
    if (rc) return rc;
    if (!fused && max_lanes > 1 && net_workspace_rows(n) < B * max_lanes) {   // the reservation was capped
        max_lanes = std::max<uint64_t>(1, net_workspace_rows(n) / B);
        feat_rows = std::max<uint64_t>(B * max_lanes, last_rows);
    }
    // feature buffers of up to 32 MB and PCM staging buffers of up to 128 MB stay with the net between calls
    constexpr size_t kKeepLimit = 32u << 20, kKeepLimitPcm = 128u << 20;
    if (feat_rows * kFeat * sizeof(float) <= kKeepLimit) {
        rc = net_keep_buffer(n, 0, feat_rows * kFeat * sizeof(float), &r.feat);
        if (rc) return rc;
        r.kept_feat = true;
    } else {
        SZ_HIP(hipMalloc(&r.feat, feat_rows * kFeat * sizeof(float)));
    }
    float* feat = static_cast<float*>(r.feat);
    // the weight images the vote kernel reads, once for all tiles (the lanes start after r.compute has drained)
    rc = net_vote_prepare(n, r.compute);
    if (rc) return rc;

    if (pcm_on_device) {
        if (single) {
            rc = plan_create_tile(e, 0, 0, (uint32_t)nw, 0, &r.first);
            if (rc) return rc;
            rc = run_tile(n, r.first, pcm, feat, nw, tally, r.compute);
            if (rc) return rc;
            SZ_HIP(hipStreamSynchronize(r.compute));
            return SZH_OK;
        }
        rc = plan_create_tile(e, 0, 0, (uint32_t)B, 2, &r.first);
        if (rc) return rc;
        rc = run_tile(n, r.first, pcm, feat, B, tally, r.compute);
        if (rc) return rc;
        if (n_mid && prof_enabled()) {
            // event pairs inside a captured sequence would be replayed, not re-recorded
            rc = plan_create_tile(e, (B - 2) * kHop, 2, (uint32_t)B, 2, &r.mid);
            if (rc == SZH_OK) rc = plan_set_advance(r.mid, B * kHop);
            if (rc) return rc;
            for (uint64_t t = 0; t < n_mid && rc == SZH_OK; ++t) {
                rc = run_tile(n, r.mid, pcm, feat, B, tally, r.compute);
                if ((t & 63) == 63) SZ_HIP(hipStreamSynchronize(r.compute));   // a short queue here too
            }
            if (rc) return rc;
        } else if (n_mid) {
            // Lane l takes interior tiles l, l + lanes, ...: its plan starts at tile l and moves on by
            // `lanes` tiles per replay; its features and its workspace rows are the l-th range of B rows.
            const int lanes = (int)std::min<uint64_t>(std::min<uint64_t>(n_mid, (uint64_t)max_lanes), kMaxLanes);
            r.net = n;
            // the first tile (direct launches on r.compute) used the rows of lane 0
            SZ_HIP(hipStreamSynchronize(r.compute));
            for (int l = 0; l < lanes; ++l) {
                // The last lane runs on the net's own stream, the others on streams of this call: with the null
                // stream that makes four streams for three lanes, one per hardware queue of the process (HIP's
                // default is four).  A fifth stream is given a queue that another one already has, and which
                // one is not ours to choose: with two lanes on one queue, 190 instead of 254 M windows/s at
                // tiles of 8192 windows.
                if (l == lanes - 1) r.lane_stream[l] = r.compute;
                else SZ_HIP(hipStreamCreateWithFlags(&r.lane_stream[l], hipStreamNonBlocking));
                rc = plan_create_tile(e, (B - 2 + (uint64_t)l * B) * kHop, 2, (uint32_t)B, 2, &r.lane_plan[l]);
                if (rc == SZH_OK) rc = plan_set_advance(r.lane_plan[l], (uint64_t)lanes * B * kHop);
                if (rc) return rc;
                n->ws.row_base = (uint64_t)l * B;
                SZ_HIP(hipStreamBeginCapture(r.lane_stream[l], hipStreamCaptureModeRelaxed));
                rc = run_tile(n, r.lane_plan[l], pcm, feat + (uint64_t)l * B * kFeat, B, tally, r.lane_stream[l]);
                const hipError_t he = hipStreamEndCapture(r.lane_stream[l], &r.lane_graph[l]);
                n->ws.row_base = 0;
                if (rc) return rc;
                if (he != hipSuccess) return hip_fail(he, "hipStreamEndCapture");
                SZ_HIP(hipGraphInstantiate(&r.lane_exec[l], r.lane_graph[l], nullptr, nullptr, 0));
                for (int k = 0; k < kReplayDepth; ++k) SZ_HIP(hipEventCreateWithFlags(&r.lane_done[l][k], hipEventDisableTiming));
            }
            for (uint64_t t = 0; t < n_mid; ++t) {
                const int l = (int)(t % lanes);
                const int slot = (int)((t / lanes) % kReplayDepth);
                // at most kReplayDepth replays queued per lane: wait for the one that used this slot
                if (t / lanes >= (uint64_t)kReplayDepth) SZ_HIP(hipEventSynchronize(r.lane_done[l][slot]));
                SZ_HIP(hipGraphLaunch(r.lane_exec[l], r.lane_stream[l]));
                SZ_HIP(hipEventRecord(r.lane_done[l][slot], r.lane_stream[l]));
            }
            for (int l = 0; l < lanes; ++l) SZ_HIP(hipStreamSynchronize(r.lane_stream[l]));
        }
        rc = plan_create_tile(e, (last_first - 2) * kHop, 2, (uint32_t)last_rows, 0, &r.last);
        if (rc) return rc;
        rc = run_tile(n, r.last, pcm, feat, last_rows, tally, r.compute);
        if (rc) return rc;
        SZ_HIP(hipStreamSynchronize(r.compute));
        return SZH_OK;
    }

    // PCM in host memory: tiles are staged through two device buffers; the copy of tile j+1
    // overlaps the kernels of tile j.
    const uint64_t buf_windows = single ? nw : std::max<uint64_t>(B + 4, last_rows + 2);
    if (!single) SZ_HIP(hipStreamCreateWithFlags(&r.copy, hipStreamNonBlocking));
    // From 32 MB of PCM on, the copies go through a ring of pinned pieces filled by helper threads: twice the rate
    // of a pageable hipMemcpyAsync (host_copy.cpp).  Below that the threads' start-up would show.
    if (n_samples * sizeof(int16_t) >= kStagedCopyBytes) {
        const int helpers = HostStager::helpers_wanted();
        if (helpers > 0) {
            void* ring = nullptr;
            rc = net_stage_pinned(n, kRingBytes, &ring);
            if (rc == SZH_OK) rc = r.stager.start(ring, kRingBytes, kPieceBytes, helpers);
            if (rc) return rc;
        }
    }
    r.kept_pcm = span(buf_windows) * sizeof(int16_t) <= kKeepLimitPcm;
    for (int i = 0; i < 2; ++i) {
        if (r.kept_pcm) {
            if (single && i == 1) continue;   // one piece: one buffer
            rc = net_keep_buffer(n, 1 + i, span(buf_windows) * sizeof(int16_t), &r.pcm[i]);
            if (rc) return rc;
        } else {
            SZ_HIP(hipMalloc(&r.pcm[i], span(buf_windows) * sizeof(int16_t)));
        }
        SZ_HIP(hipEventCreateWithFlags(&r.ready[i], hipEventDisableTiming));
        SZ_HIP(hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming));
    }
    const uint64_t n_tiles = single ? 1 : n_mid + 2;
    if (single) {
        rc = plan_create_tile(e, 0, 0, (uint32_t)nw, 0, &r.first);
        if (rc) return rc;
    } else {
        rc = plan_create_tile(e, 0, 0, (uint32_t)B, 2, &r.first);
        if (rc == SZH_OK) rc = plan_create_tile(e, 0, 2, (uint32_t)B, 2, &r.mid);
        if (rc == SZH_OK) rc = plan_create_tile(e, 0, 2, (uint32_t)last_rows, 0, &r.last);
        if (rc) return rc;
    }
    for (uint64_t j = 0; j < n_tiles; ++j) {
        const int slot = (int)(j & 1);
        const bool is_first = j == 0, is_last = j + 1 == n_tiles;
        const uint64_t out_first = j * B;
        const uint64_t rows = single ? nw : is_last ? last_rows : B;
        const uint64_t lead = is_first ? 0 : 2, trail = is_last ? 0 : 2;
        const uint64_t s0 = (out_first - lead) * kHop;
        const uint64_t len = span(lead + rows + trail);
        int16_t* buf = static_cast<int16_t*>(r.pcm[slot]);
        if (j >= 2) {
            SZ_HIP(hipEventSynchronize(r.done[slot]));   // the host stays at most two tiles ahead of the GPU
            SZ_HIP(hipStreamWaitEvent(r.copy, r.done[slot], 0));
        }
        if (r.stager.started()) {
            rc = r.stager.copy(buf, pcm + s0, len * sizeof(int16_t), single ? r.compute : r.copy);
            if (rc) return rc;
            if (is_last) r.stager.finish();
        } else {   // (one piece has nothing to overlap with: its copy goes ahead of the kernels on their stream)
            SZ_HIP(hipMemcpyAsync(buf, pcm + s0, len * sizeof(int16_t), hipMemcpyHostToDevice, single ? r.compute : r.copy));
        }
        if (!single) {
            SZ_HIP(hipEventRecord(r.ready[slot], r.copy));
            SZ_HIP(hipStreamWaitEvent(r.compute, r.ready[slot], 0));
        }
        if (is_first || is_last || prof_enabled()) {
            Plan* plan = is_first ? r.first : is_last ? r.last : r.mid;
            rc = run_tile(n, plan, buf, feat, rows, tally, r.compute);
            if (rc) return rc;
        } else {
            if (!r.exec[slot]) {
                rc = capture_tile(r, slot, n, buf, (uint32_t)B, tally);
                if (rc) return rc;
            }
            SZ_HIP(hipGraphLaunch(r.exec[slot], r.compute));
        }
        SZ_HIP(hipEventRecord(r.done[slot], r.compute));
    }
    SZ_HIP(hipStreamSynchronize(r.compute));
    return SZH_OK;
}

}  // namespace sz

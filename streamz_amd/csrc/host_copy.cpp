// Pageable host memory to device memory, and back, at the rate of pinned DMA.
//
// hipMemcpyAsync from pageable memory moves about 28 GB/s on this platform (the runtime stages the data through its
// own pinned buffer on the calling thread), and into freshly allocated host pages far less (5 GB/s: the page faults
// are taken one after the other); between pinned memory and the device the same link moves 57 GB/s, and one host
// thread copies memory to memory at about 30 GB/s.  A long recording handed over as a host slice
// (identify_speaker_list, lib.rs:1383) or a batch of files (lib.rs:330-345) is bound by these copies, so the stager
// below does the staging itself with a few helper threads.  The data is cut into pieces; piece g (numbered across
// all copies of a stager's life, so that the pipeline keeps flowing from one copy into the next) uses slot
// g % slots of a pinned ring.
//   to the device:  helpers copy pieces into free slots; the calling thread sends every filled slot with
//                   hipMemcpyAsync and frees it for refilling once the event recorded behind the transfer completed.
//   to the host:    the calling thread asks for every piece with hipMemcpyAsync into a free slot; once the event
//                   behind it has completed, a helper copies the slot out to its destination, which frees the slot.
// Only the calling thread talks to HIP.  Transfers queued on one stream follow each other with a gap of about 15 us
// (a 4 MB piece takes 70), so every other piece goes to a second stream, which forks from the caller's at the start
// of a copy and joins it at the end.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "internal.h"
#include "streamz_hip.h"

namespace sz {

namespace {

// waits politely: a few thousand yields, then short sleeps
struct Backoff {
    int spins = 0;
    void pause()
    {
        if (++spins < 4096) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
};

}  // namespace

struct HostStager::Impl {
    bool to_host = false;
    char* ring = nullptr;
    size_t piece = 0;
    int slots = 0;
    std::vector<hipEvent_t> sent;          // per slot: recorded behind the slot's latest transfer
    std::vector<std::thread> helpers;
    hipStream_t second = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    // the copy in progress (written by the calling thread before `published` moves past its pieces): a list of
    // segments, each cut into pieces of its own (a piece never spans two segments)
    struct Seg {
        char* host;                        // source (to the device) or destination (to the host)
        char* dev;
        size_t bytes;
        uint64_t first;                    // number of the segment's first piece
    };
    std::vector<Seg> segs;
    struct Piece { char *host, *dev; size_t len; };
    Piece piece_of(uint64_t g) const
    {
        // the last segment whose first piece is <= g
        size_t lo = 0, hi = segs.size();
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (segs[mid].first <= g) lo = mid; else hi = mid;
        }
        const Seg& sg = segs[lo];
        const size_t off = (size_t)(g - sg.first) * piece;
        return {sg.host + off, sg.dev + off, std::min(piece, sg.bytes - off)};
    }
    std::atomic<uint64_t> published{0};    // pieces [0, published) exist
    // pieces [0, ready) are the helpers' to copy: their slots are free (to the device) / hold their data (to the host)
    std::atomic<uint64_t> ready{0};
    std::atomic<uint64_t> claim{0};        // next piece a helper takes
    std::vector<std::atomic<uint64_t>> copied;   // per slot: g + 1 once a helper is done with piece g
    std::atomic<bool> stop{false};
    uint64_t enqueued = 0;                 // pieces [0, enqueued) have been handed to a stream

    explicit Impl(int n_slots) : sent(n_slots, nullptr), copied(n_slots) { for (auto& c : copied) c.store(0); }

    void helper()
    {
        for (;;) {
            const uint64_t g = claim.fetch_add(1, std::memory_order_relaxed);
            Backoff b;
            while (g >= published.load(std::memory_order_acquire) || g >= ready.load(std::memory_order_acquire)) {
                if (stop.load(std::memory_order_acquire)) return;
                b.pause();
            }
            const Piece p = piece_of(g);
            char* slot = ring + (size_t)(g % slots) * piece;
            if (to_host) std::memcpy(p.host, slot, p.len);
            else std::memcpy(slot, p.host, p.len);
            copied[g % slots].store(g + 1, std::memory_order_release);
        }
    }

    bool helper_done(uint64_t g) const { return copied[g % slots].load(std::memory_order_acquire) >= g + 1; }

    // to the device: frees slots whose latest transfer has completed.  to the host: hands over slots whose data
    // has arrived.  Never blocks; anything but "done" or "not yet" from the event is an error.
    hipError_t advance_ready()
    {
        for (;;) {
            const uint64_t r = ready.load(std::memory_order_relaxed);
            const bool needs_event = to_host || r >= (uint64_t)slots;   // (the first use of a slot needs no wait)
            if (needs_event) {
                if ((to_host ? r : r - slots) >= enqueued) return hipSuccess;
                const hipError_t q = hipEventQuery(sent[r % slots]);
                if (q == hipErrorNotReady) return hipSuccess;
                if (q != hipSuccess) return q;
            }
            ready.store(r + 1, std::memory_order_release);
        }
    }
};

int HostStager::helpers_wanted()
{
    const char* v = std::getenv("STREAMZ_COPY_THREADS");
    return v ? std::max(0, std::min(std::atoi(v), 8)) : 3;
}

HostStager::HostStager() = default;

HostStager::~HostStager() { finish(); }

int HostStager::start(void* pinned_ring, size_t ring_bytes, size_t piece_bytes, int helpers, bool to_host)
{
    const int slots = piece_bytes ? (int)(ring_bytes / piece_bytes) : 0;
    if (impl_ || !pinned_ring || slots < 2 || helpers < 1) {
        set_error("host stager: ring of %zu bytes, pieces of %zu, %d helpers", ring_bytes, piece_bytes, helpers);
        return SZH_EINVAL;
    }
    impl_ = new Impl(slots);
    impl_->to_host = to_host;
    impl_->ring = static_cast<char*>(pinned_ring);
    impl_->piece = piece_bytes;
    impl_->slots = slots;
    for (int i = 0; i < slots; ++i) SZ_HIP(hipEventCreateWithFlags(&impl_->sent[i], hipEventDisableTiming));
    SZ_HIP(hipStreamCreateWithFlags(&impl_->second, hipStreamNonBlocking));
    SZ_HIP(hipEventCreateWithFlags(&impl_->fork, hipEventDisableTiming));
    SZ_HIP(hipEventCreateWithFlags(&impl_->join, hipEventDisableTiming));
    for (int i = 0; i < helpers; ++i) impl_->helpers.emplace_back([im = impl_] { im->helper(); });
    return SZH_OK;
}

int HostStager::copy(void* dst, const void* src, size_t bytes, hipStream_t s)
{
    const Span one{dst, src, bytes};
    return copy_many(&one, 1, s);
}

int HostStager::copy_many(const Span* spans, size_t n_spans, hipStream_t s)
{
    Impl& im = *impl_;
    const uint64_t first = im.enqueued;
    uint64_t end = first;
    im.segs.clear();
    for (size_t i = 0; i < n_spans; ++i) {
        if (spans[i].bytes == 0) continue;
        char* d = static_cast<char*>(spans[i].dst);
        char* sr = static_cast<char*>(const_cast<void*>(spans[i].src));
        im.segs.push_back({im.to_host ? d : sr, im.to_host ? sr : d, spans[i].bytes, end});
        end += (spans[i].bytes + im.piece - 1) / im.piece;
    }
    if (end == first) return SZH_OK;
    im.published.store(end, std::memory_order_release);
    const bool two = end - first > 1;
    if (two) {
        SZ_HIP(hipEventRecord(im.fork, s));
        SZ_HIP(hipStreamWaitEvent(im.second, im.fork, 0));
    }
    uint64_t finished = first;   // to the host: pieces [first, finished) are in their destination
    Backoff b;
    while (im.enqueued < end || (im.to_host && finished < end)) {
        bool progress = false;
        SZ_HIP(im.advance_ready());
        const uint64_t g = im.enqueued;
        // the next transfer: its slot must hold the filled piece (to the device) or be drained of the piece that
        // used it before (to the host)
        if (g < end && (im.to_host ? (g < (uint64_t)im.slots || im.helper_done(g - im.slots)) : im.helper_done(g))) {
            const hipStream_t sp = two && ((g - first) & 1) ? im.second : s;
            const Impl::Piece p = im.piece_of(g);
            char* slot = im.ring + (size_t)(g % im.slots) * im.piece;
            if (im.to_host) SZ_HIP(hipMemcpyAsync(slot, p.dev, p.len, hipMemcpyDeviceToHost, sp));
            else SZ_HIP(hipMemcpyAsync(p.dev, slot, p.len, hipMemcpyHostToDevice, sp));
            SZ_HIP(hipEventRecord(im.sent[g % im.slots], sp));
            im.enqueued = g + 1;
            progress = true;
        }
        while (im.to_host && finished < im.enqueued && im.helper_done(finished)) { ++finished; progress = true; }
        if (progress) b.spins = 0;
        else b.pause();
    }
    if (two) {
        SZ_HIP(hipEventRecord(im.join, im.second));
        SZ_HIP(hipStreamWaitEvent(s, im.join, 0));
    }
    return SZH_OK;
}

void HostStager::finish()
{
    if (!impl_) return;
    impl_->stop.store(true, std::memory_order_release);
    for (std::thread& t : impl_->helpers) t.join();
    // the ring belongs to the caller and outlives the stager: wait until the last transfers are through with it
    for (hipEvent_t e : impl_->sent)
        if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
    if (impl_->second) { (void)hipStreamSynchronize(impl_->second); (void)hipStreamDestroy(impl_->second); }
    if (impl_->fork) (void)hipEventDestroy(impl_->fork);
    if (impl_->join) (void)hipEventDestroy(impl_->join);
    delete impl_;
    impl_ = nullptr;
}

}  // namespace sz

// Gradient exchange between the ranks of a node: one process per GPU, RCCL over xGMI.
//
// The reference trains in one process (lib.rs:632-665); sharding an epoch over GPUs needs
// exactly one collective, the sum of the gradient vector [w1|b1|w2|b2|w3|b3|loss|live rows],
// done in place on the vector the accumulate kernels wrote.  librccl.so.1 (573 MB) is opened
// on the first szh_comm_* call only, so single-GPU users never map it; a process that already
// holds a copy (PyTorch bundles one under the same soname) gets that copy.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <random>
#include <thread>
#include <vector>

#include "internal.h"
#include "streamz_hip.h"

namespace sz {
namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.so) return SZH_OK;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) { set_error("librccl.so.1 not found: %s", dlerror()); return SZH_ENODEVICE; }
    Rccl r;
    r.so = so;
#define SZ_SYM(field, name)                                                                  \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(so, name));                          \
    if (!r.field) { set_error("librccl.so.1 lacks %s", name); dlclose(so); return SZH_ENODEVICE; }
    SZ_SYM(GetUniqueId, "ncclGetUniqueId")
    SZ_SYM(CommInitRank, "ncclCommInitRank")
    SZ_SYM(CommDestroy, "ncclCommDestroy")
    SZ_SYM(AllReduce, "ncclAllReduce")
    SZ_SYM(Broadcast, "ncclBroadcast")
    SZ_SYM(GetErrorString, "ncclGetErrorString")
#undef SZ_SYM
    g_rccl = r;
    return SZH_OK;
}

int rccl_fail(ncclResult_t r, const char* what)
{
    set_error("RCCL error %d (%s) in %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
    return SZH_ENODEVICE;
}
#define SZ_RCCL(expr)                                             \
    do {                                                          \
        ncclResult_t r__ = (expr);                                \
        if (r__ != ncclSuccess) return rccl_fail(r__, #expr);     \
    } while (0)

// ---- the host transport: sums formed in a shared-memory segment ---------------------------------
// For ranks that share a GPU (RCCL refuses to form a communicator of two ranks on one device), so that a
// multi-rank run can be rehearsed on a smaller machine.  The segment holds a header and one slot of
// kHostSlotFloats floats per rank; a reduction copies every rank's vector into its slot, and each rank adds
// the slots in rank order, so all ranks end with the same bits.
constexpr char kHostIdTag[8] = {'S', 'Z', 'H', 'H', 'O', 'S', 'T', '1'};
constexpr uint64_t kHostSlotFloats = 16ull << 20;
constexpr size_t kHostHeaderBytes = 4096;
constexpr uint32_t kHostMagic = 0x535A4831u;
constexpr int kHostTimeoutSeconds = 120;

struct HostHeader {
    std::atomic<uint32_t> magic;     // kHostMagic once rank 0 has initialised the header
    std::atomic<uint32_t> attached;  // ranks that have mapped the segment
    std::atomic<uint32_t> arrived;   // barrier: ranks that have arrived in this phase
    std::atomic<uint32_t> phase;     // barrier: generation
    std::atomic<uint32_t> failed;    // a rank gave up waiting: the others stop waiting too
    uint32_t world;
};
static_assert(sizeof(HostHeader) <= kHostHeaderBytes, "header fits its page");

template <class Pred>
bool wait_until(Pred done, std::atomic<uint32_t>* failed)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; !done(); ++spin) {
        if (failed && failed->load(std::memory_order_acquire)) return false;
        if (spin < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(100));
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kHostTimeoutSeconds)) {
            if (failed) failed->store(1, std::memory_order_release);
            return false;
        }
    }
    return true;
}

}  // namespace

struct Comm {
    ncclComm_t nccl = nullptr;  // null for a single rank created without an id: nothing to exchange
    int rank = 0, world = 1, device = 0;
    HostHeader* host = nullptr;  // the host transport's segment (then nccl is null)
    size_t host_bytes = 0;
    float* slot(int r) const { return reinterpret_cast<float*>(reinterpret_cast<char*>(host) + kHostHeaderBytes) + (size_t)r * kHostSlotFloats; }
};

namespace {

int host_barrier(Comm* c)
{
    HostHeader* h = c->host;
    const uint32_t gen = h->phase.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->phase.store(gen + 1, std::memory_order_release);
        return SZH_OK;
    }
    if (!wait_until([&] { return h->phase.load(std::memory_order_acquire) != gen; }, &h->failed)) {
        set_error("host communicator: rank %d waited more than %d s for the other ranks", c->rank, kHostTimeoutSeconds);
        return SZH_ENODEVICE;
    }
    return SZH_OK;
}

int host_attach(Comm* c, const uint8_t* id)
{
    char name[SZH_COMM_ID_BYTES - 8 + 1] = {};
    std::memcpy(name, id + 8, SZH_COMM_ID_BYTES - 8);
    const size_t bytes = kHostHeaderBytes + (size_t)c->world * kHostSlotFloats * sizeof(float);
    int fd = -1;
    if (c->rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        // sparse: only the pages a vector touches are ever backed by memory
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) { close(fd); shm_unlink(name); } set_error("host communicator: cannot create %s", name); return SZH_ENOMEM; }
    } else {
        const bool ok = wait_until([&] {
            if (fd < 0) fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            return fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes;
        }, nullptr);
        if (!ok) { if (fd >= 0) close(fd); set_error("host communicator: rank 0 never created %s", name); return SZH_ENODEVICE; }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (c->rank == 0) shm_unlink(name); set_error("host communicator: cannot map %s", name); return SZH_ENOMEM; }
    c->host = static_cast<HostHeader*>(p);
    c->host_bytes = bytes;
    HostHeader* h = c->host;
    if (c->rank == 0) {  // a new segment is all zeros
        h->world = (uint32_t)c->world;
        h->magic.store(kHostMagic, std::memory_order_release);
    } else if (!wait_until([&] { return h->magic.load(std::memory_order_acquire) == kHostMagic; }, nullptr) || h->world != (uint32_t)c->world) {
        set_error("host communicator: segment %s was not set up for %d ranks", name, c->world);
        return SZH_EINVAL;
    }
    h->attached.fetch_add(1, std::memory_order_acq_rel);
    const bool all = wait_until([&] { return h->attached.load(std::memory_order_acquire) >= (uint32_t)c->world; }, &h->failed);
    if (c->rank == 0) shm_unlink(name);  // every rank holds its mapping (or the wait failed): the name can go
    if (!all) { set_error("host communicator: only %u of %d ranks arrived", h->attached.load(), c->world); return SZH_ENODEVICE; }
    return SZH_OK;
}

// in place on `n` device floats: op 0 = sum over ranks, op 1 = copy of rank `root`
int host_exchange(Comm* c, float* dev, uint64_t n, int op, int root, hipStream_t stream)
{
    if (n > kHostSlotFloats) { set_error("host communicator: %llu floats exceed its limit of %llu", (unsigned long long)n, (unsigned long long)kHostSlotFloats); return SZH_EINVAL; }
    SZ_HIP(hipStreamSynchronize(stream));
    if (op == 0 || c->rank == root) SZ_HIP(hipMemcpy(c->slot(c->rank), dev, n * sizeof(float), hipMemcpyDeviceToHost));
    if (int rc = host_barrier(c)) return rc;
    if (op == 0) {
        std::vector<float> sum(c->slot(0), c->slot(0) + n);
        for (int r = 1; r < c->world; ++r) {
            const float* s = c->slot(r);
            for (uint64_t i = 0; i < n; ++i) sum[i] += s[i];
        }
        SZ_HIP(hipMemcpy(dev, sum.data(), n * sizeof(float), hipMemcpyHostToDevice));
    } else if (c->rank != root) {
        SZ_HIP(hipMemcpy(dev, c->slot(root), n * sizeof(float), hipMemcpyHostToDevice));
    }
    return host_barrier(c);  // nobody refills a slot while another rank still reads it
}

bool host_transport_requested()
{
    const char* e = std::getenv("STREAMZ_COMM");
    return e && std::strcmp(e, "host") == 0;
}

}  // namespace

}  // namespace sz

using namespace sz;

static_assert(SZH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size follows RCCL's ncclUniqueId");

extern "C" {

int szh_comm_unique_id(uint8_t id[SZH_COMM_ID_BYTES])
{
    if (!id) { set_error("null argument"); return SZH_EINVAL; }
    if (host_transport_requested()) {  // the id is the name of a shared-memory segment
        std::memset(id, 0, SZH_COMM_ID_BYTES);
        std::memcpy(id, kHostIdTag, 8);
        std::random_device rd;
        std::snprintf(reinterpret_cast<char*>(id) + 8, SZH_COMM_ID_BYTES - 8, "/szh-comm-%ld-%08x%08x", (long)getpid(), (unsigned)rd(), (unsigned)rd());
        return SZH_OK;
    }
    if (int rc = rccl_load()) return rc;
    ncclUniqueId u;
    SZ_RCCL(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, SZH_COMM_ID_BYTES);
    return SZH_OK;
}

int szh_comm_create(const uint8_t id[SZH_COMM_ID_BYTES], int rank, int world, szh_comm** comm)
{
    if (!comm || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) { set_error("bad argument"); return SZH_EINVAL; }
    Comm* c = new (std::nothrow) Comm();
    if (!c) { set_error("out of host memory"); return SZH_ENOMEM; }
    c->rank = rank;
    c->world = world;
    hipError_t he = hipGetDevice(&c->device);
    if (he != hipSuccess) { delete c; return hip_fail(he, "hipGetDevice"); }
    if (id && std::memcmp(id, kHostIdTag, 8) == 0) {
        if (world > 1)
            if (int rc = host_attach(c, id)) { szh_comm_destroy(reinterpret_cast<szh_comm*>(c)); return rc; }
    } else if (id) {
        if (int rc = rccl_load()) { delete c; return rc; }
        ncclUniqueId u;
        std::memcpy(u.internal, id, SZH_COMM_ID_BYTES);
        const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, u, rank);  // blocks until every rank has called
        if (r != ncclSuccess) { delete c; return rccl_fail(r, "ncclCommInitRank"); }
    }
    *comm = reinterpret_cast<szh_comm*>(c);
    return SZH_OK;
}

void szh_comm_destroy(szh_comm* comm)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return;
    if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
    if (c->host) (void)munmap(c->host, c->host_bytes);
    delete c;
}

int szh_comm_rank(const szh_comm* comm) { return comm ? reinterpret_cast<const Comm*>(comm)->rank : 0; }
int szh_comm_world(const szh_comm* comm) { return comm ? reinterpret_cast<const Comm*>(comm)->world : 1; }

int szh_net_allreduce_grad_dev(szh_net* net, szh_comm* comm, void* stream)
{
    if (!net || !comm) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (n->device != c->device) { set_error("net lives on device %d, communicator on device %d", n->device, c->device); return SZH_EINVAL; }
    if (c->host) return host_exchange(c, n->d_grad, n->param_count() + 2, 0, 0, static_cast<hipStream_t>(stream));
    if (!c->nccl) return SZH_OK;
    SZ_RCCL(g_rccl.AllReduce(n->d_grad, n->d_grad, n->param_count() + 2, ncclFloat32, ncclSum, c->nccl,
                             static_cast<hipStream_t>(stream)));
    return SZH_OK;
}

int szh_net_broadcast_params_dev(szh_net* net, szh_comm* comm, int root, void* stream)
{
    if (!net || !comm) { set_error("null argument"); return SZH_EINVAL; }
    Net* n = reinterpret_cast<Net*>(net);
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (root < 0 || root >= c->world) { set_error("root %d outside the %d ranks", root, c->world); return SZH_EINVAL; }
    if (n->device != c->device) { set_error("net lives on device %d, communicator on device %d", n->device, c->device); return SZH_EINVAL; }
    if (n->param_count() == 0) return SZH_OK;
    if (c->host) return host_exchange(c, n->d_params, n->param_count(), 1, root, static_cast<hipStream_t>(stream));
    if (!c->nccl) return SZH_OK;
    SZ_RCCL(g_rccl.Broadcast(n->d_params, n->d_params, n->param_count(), ncclFloat32, root, c->nccl,
                             static_cast<hipStream_t>(stream)));
    return SZH_OK;
}

}  // extern "C"

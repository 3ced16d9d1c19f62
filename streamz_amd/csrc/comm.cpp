// Gradient exchange between the ranks of a node: one process per GPU, RCCL over xGMI.
//
// The reference trains in one process (lib.rs:632-665); sharding an epoch over GPUs needs
// exactly one collective, the sum of the gradient vector [w1|b1|w2|b2|w3|b3|loss|live rows],
// done in place on the vector the accumulate kernels wrote.  librccl.so.1 (573 MB) is opened
// on the first szh_comm_* call only, so single-GPU users never map it; a process that already
// holds a copy (PyTorch bundles one under the same soname) gets that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

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

}  // namespace

struct Comm {
    ncclComm_t nccl = nullptr;  // null for a single rank created without an id: nothing to exchange
    int rank = 0, world = 1, device = 0;
};

}  // namespace sz

using namespace sz;

static_assert(SZH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size follows RCCL's ncclUniqueId");

extern "C" {

int szh_comm_unique_id(uint8_t id[SZH_COMM_ID_BYTES])
{
    if (!id) { set_error("null argument"); return SZH_EINVAL; }
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
    if (id) {
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
    if (!c->nccl || n->param_count() == 0) return SZH_OK;
    SZ_RCCL(g_rccl.Broadcast(n->d_params, n->d_params, n->param_count(), ncclFloat32, root, c->nccl,
                             static_cast<hipStream_t>(stream)));
    return SZH_OK;
}

}  // extern "C"

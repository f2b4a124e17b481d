// nbody_comm.hip — the built-in transports of the sharded step's two collectives (include/nbody.h, `nbody_comm`):
//   nbody_comm_rccl_*   RCCL (librccl loaded at run time): ncclAllGather in place + one grouped ncclSend/ncclRecv per step
//   nbody_comm_local_*  no collective library: rank THREADS of one process pull their peers' blocks with hipMemcpyPeerAsync
// The step that uses them (which launch waits for which collective) is nbody_shard.hip; a third implementation of the same
// two callbacks, over torch.distributed, lives in n-bodysimulation_amd/sharded.py.
#include "nbody_internal.hip.h"

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

extern "C" {

// ---- RCCL (librccl.so, loaded on first use) -------------------------------------------------------------------------

}  // extern "C"

namespace {

struct RcclId { char internal[128]; };  // ncclUniqueId
typedef void* RcclComm;                 // ncclComm_t
enum { kRcclFloat = 7 };                // ncclFloat32

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

RcclApi g_rccl;
std::mutex g_rccl_mu;  // ranks may be threads of one process (nbody_headless --ngpu): load the library once

int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return NBODY_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return nbody_fail(NBODY_ERR_HIP, "cannot load librccl.so: %s", dlerror());
    RcclApi api;
    api.handle = h;
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) ok = false; return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) return nbody_fail(NBODY_ERR_HIP, "librccl.so lacks an expected ncclXxx symbol");
    g_rccl = api;
    return NBODY_OK;
}

struct RcclUser {
    RcclComm comm = nullptr;
    int rank = 0, world = 1;
};

int rccl_all_gather(void* user, nbody_float4* d_x_full, int bodies_per_rank, void* hip_stream)
{
    RcclUser* u = static_cast<RcclUser*>(user);
    const size_t count = (size_t)bodies_per_rank * 4;  // floats per rank
    const float* send = reinterpret_cast<const float*>(d_x_full) + (size_t)u->rank * count;  // in place
    return g_rccl.AllGather(send, d_x_full, count, kRcclFloat, u->comm, static_cast<hipStream_t>(hip_stream));
}

int rccl_exchange(void* user, const nbody_shard_segment* send, int n_sends, const nbody_float4* d_jbuf,
                  const nbody_shard_segment* recv, int n_recvs, nbody_float4* d_rbuf, void* hip_stream)
{
    RcclUser* u = static_cast<RcclUser*>(user);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    int rc = g_rccl.GroupStart();
    for (int k = 0; k < n_sends && rc == 0; ++k)
        rc = g_rccl.Send(d_jbuf + send[k].offset, (size_t)send[k].count * 4, kRcclFloat, send[k].peer, u->comm, st);
    for (int k = 0; k < n_recvs && rc == 0; ++k)
        rc = g_rccl.Recv(d_rbuf + recv[k].offset, (size_t)recv[k].count * 4, kRcclFloat, recv[k].peer, u->comm, st);
    const int rc2 = g_rccl.GroupEnd();
    return rc ? rc : rc2;
}

}  // namespace

extern "C" {

int nbody_comm_rccl_unique_id(void* out_128_bytes)
{
    if (!out_128_bytes) return nbody_fail(NBODY_ERR_INVALID, "null out");
    if (int rc = rccl_load()) return rc;
    RcclId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return nbody_fail(NBODY_ERR_HIP, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(rc));
    std::memcpy(out_128_bytes, &id, sizeof id);
    return NBODY_OK;
}

int nbody_comm_rccl_create(nbody_comm* out, int rank, int world, const void* unique_id_128_bytes, int device)
{
    if (!out || !unique_id_128_bytes) return nbody_fail(NBODY_ERR_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return nbody_fail(NBODY_ERR_INVALID, "rank %d of %d", rank, world);
    if (int rc = rccl_load()) return rc;
    RcclUser* u = new (std::nothrow) RcclUser();
    if (!u) return nbody_fail(NBODY_ERR_NOMEM, "out of host memory");
    u->rank = rank;
    u->world = world;
    RcclId id;
    std::memcpy(&id, unique_id_128_bytes, sizeof id);
    if (device < 0 && hipGetDevice(&device) != hipSuccess) {
        delete u;
        return nbody_fail(NBODY_ERR_HIP, "no current HIP device");
    }
    DeviceScope scope(device);  // the communicator lives on the device that is current during ncclCommInitRank
    if (scope.err != hipSuccess) {
        delete u;
        return nbody_fail(NBODY_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.err));
    }
    const int rc = g_rccl.CommInitRank(&u->comm, world, id, rank);
    if (rc != 0) {
        delete u;
        return nbody_fail(NBODY_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(rc));
    }
    out->user = u;
    out->all_gather = rccl_all_gather;
    out->exchange = rccl_exchange;
    return NBODY_OK;
}

int nbody_comm_rccl_destroy(nbody_comm* comm)
{
    if (!comm || !comm->user) return NBODY_OK;
    RcclUser* u = static_cast<RcclUser*>(comm->user);
    if (u->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(u->comm);
    delete u;
    comm->user = nullptr;
    comm->all_gather = nullptr;
    comm->exchange = nullptr;
    return NBODY_OK;
}

}  // extern "C"

// ---- LOCAL transport: the ranks are threads of ONE process, each with its own device (or, for rehearsals, sharing one) ----------
//
// No RCCL: every rank PULLS what it needs with hipMemcpyPeerAsync (device to device over xGMI when peer access is enabled), ordered by
// events that the owning rank records and a host rendezvous of the rank threads per collective:
//   all-gather  each rank copies its advanced block into one of two staging buffers and records `ready`; after the rendezvous every rank
//               waits on its peers' `ready` events and pulls their staged blocks into its own position array. Two staging buffers
//               are enough: a rank's staging copy of step k+2 runs, by the chain of events, after every peer has finished pulling step k.
//   exchange    each rank publishes its J-side buffer and send table and records `cross`; after the rendezvous every rank pulls the
//               segments addressed to it from its peers' J-side buffers (which are next written after the following all-gather).
// A rank that fails aborts the group (nbody_comm_local_abort): peers waiting at a rendezvous return an error instead of hanging, and
// a rendezvous that nobody completes within the group's deadline does the same.
struct nbody_local_group {
    int world = 1;
    double deadline_s = 600.0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0, generation = 0;
    bool aborted = false;
    struct Rank {
        int device = -1;
        bool attached = false;
        float4* stage[2] = {nullptr, nullptr};
        size_t stage_bodies = 0;
        hipEvent_t ready[2] = {nullptr, nullptr};
        hipEvent_t cross = nullptr;
        const nbody_float4* jbuf = nullptr;
        nbody_shard_segment send[NBODY_MAX_RANKS];
        int n_sends = 0;
        unsigned long gathers = 0;
    } rank[NBODY_MAX_RANKS];

    bool rendezvous()   // false: the group was aborted (or nobody came within the deadline)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const int gen = generation;
        if (++waiting == world) {
            waiting = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        const bool came = cv.wait_for(lk, std::chrono::duration<double>(deadline_s), [&] { return generation != gen || aborted; });
        if (!came) { aborted = true; cv.notify_all(); }
        return generation != gen && !aborted;
    }
};

namespace {

struct LocalUser {
    nbody_local_group* g = nullptr;
    int rank = 0;
};

int local_all_gather(void* user, nbody_float4* d_x_full, int bodies_per_rank, void* hip_stream)
{
    LocalUser* u = static_cast<LocalUser*>(user);
    nbody_local_group* g = u->g;
    nbody_local_group::Rank& me = g->rank[u->rank];
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const size_t bytes = (size_t)bodies_per_rank * sizeof(float4);
    if (me.stage_bodies < (size_t)bodies_per_rank) {
        for (int b = 0; b < 2; ++b) {
            if (me.stage[b]) (void)hipFree(me.stage[b]);
            me.stage[b] = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&me.stage[b]), bytes ? bytes : 16) != hipSuccess) { nbody_comm_local_abort(g); return 1; }
        }
        me.stage_bodies = (size_t)bodies_per_rank;
    }
    const int b = (int)(me.gathers & 1);
    float4* const x = reinterpret_cast<float4*>(d_x_full);
    bool ok = hipMemcpyAsync(me.stage[b], x + (size_t)u->rank * bodies_per_rank, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess &&
              hipEventRecord(me.ready[b], st) == hipSuccess;
    if (!ok) nbody_comm_local_abort(g);
    if (!g->rendezvous()) return 1;          // everybody has staged its block and recorded its event
    for (int q = 0; q < g->world && ok; ++q) {
        if (q == u->rank) continue;
        const nbody_local_group::Rank& peer = g->rank[q];
        ok = hipStreamWaitEvent(st, peer.ready[b], 0) == hipSuccess &&
             hipMemcpyPeerAsync(x + (size_t)q * bodies_per_rank, me.device, peer.stage[b], peer.device, bytes, st) == hipSuccess;
    }
    ++me.gathers;
    if (!ok) { nbody_comm_local_abort(g); return 1; }
    return 0;
}

int local_exchange(void* user, const nbody_shard_segment* send, int n_sends, const nbody_float4* d_jbuf,
                   const nbody_shard_segment* recv, int n_recvs, nbody_float4* d_rbuf, void* hip_stream)
{
    LocalUser* u = static_cast<LocalUser*>(user);
    nbody_local_group* g = u->g;
    nbody_local_group::Rank& me = g->rank[u->rank];
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    me.jbuf = d_jbuf;
    me.n_sends = n_sends < NBODY_MAX_RANKS ? n_sends : NBODY_MAX_RANKS;
    for (int k = 0; k < me.n_sends; ++k) me.send[k] = send[k];
    bool ok = hipEventRecord(me.cross, st) == hipSuccess;   // (the stream already waits for this rank's cross launches)
    if (!ok) nbody_comm_local_abort(g);
    if (!g->rendezvous()) return 1;          // everybody has published its J-side buffer, its send table and its event
    for (int k = 0; k < n_recvs && ok; ++k) {
        const nbody_local_group::Rank& peer = g->rank[recv[k].peer];
        const nbody_shard_segment* src = nullptr;
        for (int m = 0; m < peer.n_sends; ++m)
            if (peer.send[m].peer == u->rank) src = &peer.send[m];
        if (!src || src->count != recv[k].count || src->body0 != recv[k].body0) { ok = false; break; }   // the two plans disagree
        ok = hipStreamWaitEvent(st, peer.cross, 0) == hipSuccess &&
             hipMemcpyPeerAsync(d_rbuf + recv[k].offset, me.device, peer.jbuf + src->offset, peer.device,
                                (size_t)recv[k].count * sizeof(float4), st) == hipSuccess;
    }
    if (!ok) { nbody_comm_local_abort(g); return 1; }
    return 0;
}

}  // namespace

extern "C" {

int nbody_comm_local_group_create(nbody_local_group** out, int world, double deadline_seconds)
{
    if (!out) return nbody_fail(NBODY_ERR_INVALID, "null out");
    *out = nullptr;
    if (world < 1 || world > NBODY_MAX_RANKS) return nbody_fail(NBODY_ERR_INVALID, "world=%d (1..%d)", world, NBODY_MAX_RANKS);
    nbody_local_group* g = new (std::nothrow) nbody_local_group();
    if (!g) return nbody_fail(NBODY_ERR_NOMEM, "out of host memory");
    g->world = world;
    if (deadline_seconds > 0) g->deadline_s = deadline_seconds;
    *out = g;
    return NBODY_OK;
}

int nbody_comm_local_abort(nbody_local_group* g)
{
    if (!g) return NBODY_OK;
    std::lock_guard<std::mutex> lk(g->mu);
    g->aborted = true;
    g->cv.notify_all();
    return NBODY_OK;
}

int nbody_comm_local_group_destroy(nbody_local_group* g)
{
    if (!g) return NBODY_OK;
    for (int r = 0; r < g->world; ++r)
        if (g->rank[r].attached) return nbody_fail(NBODY_ERR_INVALID, "rank %d of the local group still has its communicator", r);
    delete g;
    return NBODY_OK;
}

int nbody_comm_local_create(nbody_comm* out, nbody_local_group* g, int rank, int device)
{
    if (!out || !g) return nbody_fail(NBODY_ERR_INVALID, "null argument");
    if (rank < 0 || rank >= g->world) return nbody_fail(NBODY_ERR_INVALID, "rank %d of %d", rank, g->world);
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return nbody_fail(NBODY_ERR_HIP, "no current HIP device");
    DeviceScope scope(device);
    if (scope.err != hipSuccess) return nbody_fail(NBODY_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.err));
    nbody_local_group::Rank& me = g->rank[rank];
    if (me.attached) return nbody_fail(NBODY_ERR_INVALID, "rank %d of the local group already has a communicator", rank);
    LocalUser* u = new (std::nothrow) LocalUser();
    if (!u) return nbody_fail(NBODY_ERR_NOMEM, "out of host memory");
    u->g = g;
    u->rank = rank;
    me.device = device;
    for (hipEvent_t* e : {&me.ready[0], &me.ready[1], &me.cross})
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
            delete u;
            return nbody_fail(NBODY_ERR_HIP, "hipEventCreateWithFlags failed on device %d", device);
        }
    // direct device-to-device copies where the hardware offers them (xGMI); refused or repeated requests are not errors
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess)
        for (int d = 0; d < ndev; ++d) {
            int can = 0;
            if (d != device && hipDeviceCanAccessPeer(&can, device, d) == hipSuccess && can)
                if (hipDeviceEnablePeerAccess(d, 0) != hipSuccess) (void)hipGetLastError();
        }
    me.attached = true;
    out->user = u;
    out->all_gather = local_all_gather;
    out->exchange = local_exchange;
    return NBODY_OK;
}

int nbody_comm_local_destroy(nbody_comm* comm)
{
    if (!comm || !comm->user) return NBODY_OK;
    LocalUser* u = static_cast<LocalUser*>(comm->user);
    nbody_local_group::Rank& me = u->g->rank[u->rank];
    {
        DeviceScope scope(me.device);
        for (int b = 0; b < 2; ++b) {
            if (me.stage[b]) (void)hipFree(me.stage[b]);
            if (me.ready[b]) (void)hipEventDestroy(me.ready[b]);
            me.stage[b] = nullptr;
            me.ready[b] = nullptr;
        }
        if (me.cross) (void)hipEventDestroy(me.cross);
        me.cross = nullptr;
        me.stage_bodies = 0;
    }
    me.attached = false;
    delete u;
    comm->user = nullptr;
    comm->all_gather = nullptr;
    comm->exchange = nullptr;
    return NBODY_OK;
}

}  // extern "C"

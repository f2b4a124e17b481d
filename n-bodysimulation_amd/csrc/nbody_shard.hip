// nbody_shard.hip — one rank of the sharded (multi-GPU) step behind the C-ABI (include/nbody.h, "sharded step"),
// and the RCCL implementation of its two collectives. The reference runs on one device only
// (TestProject/kernel.cu:630, main.cpp:287); this is the build's own decomposition. The pair
// arithmetic is entirely nbody_accel_range / nbody_accel_wrapped / nbody_accel_cross / nbody_integrate_range of
// nbody_api.hip: this file only orders them on two streams.
#include "nbody.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

int nbody_fail(int code, const char* fmt, ...);  // nbody_api.hip

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return nbody_fail(NBODY_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                                 \
    } while (0)

namespace {

struct DeviceScope {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            changed = (err == hipSuccess);
        }
    }
    ~DeviceScope()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

// a[k] += b[k], individually rounded adds (the received partial sums are added in the plan's fixed order)
__global__ void __launch_bounds__(256) add_bodies(float4* a, const float4* b, int n)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    float4 p = a[k];
    const float4 q = b[k];
    p.x += q.x; p.y += q.y; p.z += q.z;
    a[k] = p;
}

}  // namespace

struct nbody_shard {
    nbody_ctx* ctx = nullptr;
    int device = 0;
    hipStream_t compute = nullptr;  // the context's launch stream
    hipStream_t comm = nullptr;     // own
    bool comm_high = false;         // measured default: normal priority
    nbody_comm cb{};
    bool have_comm = false;
    nbody_shard_plan_t plan{};
    float4 *x = nullptr, *v = nullptr, *a = nullptr, *jbuf = nullptr, *rbuf = nullptr;
    hipEvent_t ev_integrated = nullptr, ev_gathered = nullptr, ev_cross = nullptr, ev_exchanged = nullptr;
    bool fresh = true;          // positions consistent on every rank (just uploaded): the first step skips the all-gather
    bool gather_posted = false;
    // communication timing
    bool timing = false;
    struct StepEvents {
        hipEvent_t g0, g1, local_done, x0, x1, own_done;
        bool gathered, exchanged;
    };
    std::vector<StepEvents> timed;
};

namespace {

int new_event(hipEvent_t* e, bool timing)
{
    HIP_TRY(hipEventCreateWithFlags(e, timing ? hipEventDefault : hipEventDisableTiming));
    return NBODY_OK;
}

int check_shard(const nbody_shard* s)
{
    if (!s) return nbody_fail(NBODY_ERR_INVALID, "null shard");
    return NBODY_OK;
}

nbody_float4* nb(float4* p) { return reinterpret_cast<nbody_float4*>(p); }

// The communication stream. Normal priority by default. nbody_shard_set_comm_priority(.., 1) asks for the greatest priority the
// device offers, so that RCCL's few channel workgroups are placed ahead of the thousands of queued force workgroups they run
// beside — an option, not the default: with three or more processes on ONE GPU it was measured pathological (spinning
// high-priority RCCL kernels of different processes: 622 ms per step against 1.26, profiles/r04a_rehearsal_priority_probe.txt),
// and on a GPU per rank it has never been measured at all.
int make_comm_stream(nbody_shard* s, bool high)
{
    hipStream_t st = nullptr;
    hipError_t e = hipErrorUnknown;
    if (high) {
        int prio_least = 0, prio_greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_greatest);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            st = nullptr;
        }
    }
    if (!st) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) return nbody_fail(NBODY_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
    s->comm = st;
    s->comm_high = high;
    return NBODY_OK;
}

int ensure_comm(nbody_shard* s) { return s->comm ? NBODY_OK : make_comm_stream(s, s->comm_high); }

int phase_gather(nbody_shard* s)
{
    const nbody_shard_plan_t& p = s->plan;
    s->gather_posted = false;
    if (s->timing) {
        nbody_shard::StepEvents ev{};
        for (hipEvent_t* e : {&ev.g0, &ev.g1, &ev.local_done, &ev.x0, &ev.x1, &ev.own_done})
            if (int rc = new_event(e, true)) return rc;
        s->timed.push_back(ev);
    }
    if (p.world == 1 || s->fresh) return NBODY_OK;
    if (int rc = ensure_comm(s)) return rc;
    HIP_TRY(hipStreamWaitEvent(s->comm, s->ev_integrated, 0));  // the own block was advanced by the last integrate
    if (s->timing) HIP_TRY(hipEventRecord(s->timed.back().g0, s->comm));
    if (s->cb.all_gather(s->cb.user, nb(s->x), p.shard, static_cast<void*>(s->comm)) != 0)
        return nbody_fail(NBODY_ERR_HIP, "all-gather callback failed on rank %d", p.rank);
    if (s->timing) {
        HIP_TRY(hipEventRecord(s->timed.back().g1, s->comm));
        s->timed.back().gathered = true;
    }
    HIP_TRY(hipEventRecord(s->ev_gathered, s->comm));
    s->gather_posted = true;
    return NBODY_OK;
}

int phase_compute(nbody_shard* s)
{
    const nbody_shard_plan_t& p = s->plan;
    nbody_ctx* c = s->ctx;
    if (p.shard == 0) return NBODY_OK;
    if (p.schedule == NBODY_SCHEDULE_CANONICAL) {
        if (s->gather_posted) HIP_TRY(hipStreamWaitEvent(s->compute, s->ev_gathered, 0));
        return nbody_accel_range(c, nb(s->x), nb(s->a), p.i0, p.i1, 0, p.n_pad, 0);
    }
    if (p.schedule == NBODY_SCHEDULE_ONESIDED) {
        // own block against itself while the positions of the others are still on their way
        if (int rc = nbody_accel_range(c, nb(s->x), nb(s->a), p.i0, p.i1, p.i0, p.i1, 0)) return rc;
        if (s->timing && s->gather_posted) HIP_TRY(hipEventRecord(s->timed.back().local_done, s->compute));
        if (s->gather_posted) HIP_TRY(hipStreamWaitEvent(s->compute, s->ev_gathered, 0));
        if (p.world == 1) return NBODY_OK;
        // everybody else in one launch: sources i1, i1+1, ... wrapping round to i0-1
        return nbody_accel_wrapped(c, nb(s->x), p.n_pad, nb(s->a), p.i0, p.i1, p.i1 % p.n_pad, p.n_pad - p.shard, 1);
    }
    // symmetric schedule. The own block against itself is issued in two halves: the first hides the all-gather, the
    // second (phase_finish) hides the exchange. The sums land in `a` in a fixed order: cross launches, own block, received.
    HIP_TRY(hipMemsetAsync(s->a, 0, (size_t)p.shard * sizeof(float4), s->compute));
    if (int rc = nbody_accel_square_part(c, nb(s->x), nb(s->a), p.i0, p.i1, 1, 0, p.world > 1 ? 2 : 1)) return rc;
    if (s->timing && s->gather_posted) HIP_TRY(hipEventRecord(s->timed.back().local_done, s->compute));
    if (s->gather_posted) HIP_TRY(hipStreamWaitEvent(s->compute, s->ev_gathered, 0));
    if (p.world == 1) return NBODY_OK;
    for (int l = 0; l < p.n_launches; ++l) {
        const nbody_cross_launch& L = p.launch[l];
        if (int rc = nbody_accel_cross(c, nb(s->x), p.n_pad, nb(s->a + (L.i0 - p.i0)), L.i0, L.i1, 1, L.j0, L.count,
                                       nb(s->jbuf + L.jbuf_offset)))
            return rc;
    }
    HIP_TRY(hipEventRecord(s->ev_cross, s->compute));
    return NBODY_OK;
}

int phase_exchange(nbody_shard* s)
{
    const nbody_shard_plan_t& p = s->plan;
    if (p.world == 1 || p.schedule != NBODY_SCHEDULE_SYMMETRIC || p.shard == 0) return NBODY_OK;
    if (int rc = ensure_comm(s)) return rc;
    HIP_TRY(hipStreamWaitEvent(s->comm, s->ev_cross, 0));
    if (s->timing) HIP_TRY(hipEventRecord(s->timed.back().x0, s->comm));
    if (s->cb.exchange(s->cb.user, p.send, p.n_sends, nb(s->jbuf), p.recv, p.n_recvs, nb(s->rbuf), static_cast<void*>(s->comm)) != 0)
        return nbody_fail(NBODY_ERR_HIP, "exchange callback failed on rank %d", p.rank);
    if (s->timing) {
        HIP_TRY(hipEventRecord(s->timed.back().x1, s->comm));
        s->timed.back().exchanged = true;
    }
    HIP_TRY(hipEventRecord(s->ev_exchanged, s->comm));
    return NBODY_OK;
}

int phase_finish(nbody_shard* s)
{
    const nbody_shard_plan_t& p = s->plan;
    if (p.shard == 0) return NBODY_OK;
    if (p.schedule == NBODY_SCHEDULE_SYMMETRIC && p.world > 1) {  // second half of the own block (+ its slab sum), while the exchange runs
        if (int rc = nbody_accel_square_part(s->ctx, nb(s->x), nb(s->a), p.i0, p.i1, 1, 1, 2)) return rc;
        if (s->timing && !s->timed.empty() && s->timed.back().exchanged) HIP_TRY(hipEventRecord(s->timed.back().own_done, s->compute));
    }
    if (p.world > 1 && p.schedule == NBODY_SCHEDULE_SYMMETRIC) {
        HIP_TRY(hipStreamWaitEvent(s->compute, s->ev_exchanged, 0));
        for (int k = 0; k < p.n_recvs; ++k) {  // fixed order: nearest preceding rank first
            const nbody_shard_segment& r = p.recv[k];
            add_bodies<<<(r.count + 255) / 256, 256, 0, s->compute>>>(s->a + (r.body0 - p.i0), s->rbuf + r.offset, r.count);
        }
        HIP_TRY(hipGetLastError());
    }
    if (int rc = nbody_integrate_range(s->ctx, nb(s->x), nb(s->v), nb(s->a), p.i0, p.i1)) return rc;
    HIP_TRY(hipEventRecord(s->ev_integrated, s->compute));
    s->fresh = false;
    return NBODY_OK;
}

}  // namespace

extern "C" {

int nbody_shard_create(nbody_shard** out, nbody_ctx* ctx, int rank, int world, int n_total, const nbody_comm* comm)
{
    if (!out) return nbody_fail(NBODY_ERR_INVALID, "null out");
    *out = nullptr;
    int device = 0, kernel = 0;
    void* stream = nullptr;
    if (int rc = nbody_ctx_get(ctx, &device, &kernel, &stream)) return rc;
    if (world > 1 && (!comm || !comm->all_gather || !comm->exchange))
        return nbody_fail(NBODY_ERR_INVALID, "world=%d needs both communication callbacks", world);
    const int schedule = kernel == NBODY_KERNEL_STRICT ? NBODY_SCHEDULE_CANONICAL
                         : kernel == NBODY_KERNEL_ONESIDED ? NBODY_SCHEDULE_ONESIDED : NBODY_SCHEDULE_SYMMETRIC;
    nbody_shard* s = new (std::nothrow) nbody_shard();
    if (!s) return nbody_fail(NBODY_ERR_NOMEM, "out of host memory");
    if (nbody_shard_plan(rank, world, n_total, schedule, &s->plan) != NBODY_OK) {
        delete s;
        return nbody_fail(NBODY_ERR_INVALID, "bad shard geometry: rank %d of %d, %d bodies", rank, world, n_total);
    }
    s->ctx = ctx;
    s->device = device;
    s->compute = static_cast<hipStream_t>(stream);
    if (comm) { s->cb = *comm; s->have_comm = true; }
    DeviceScope scope(device);
    const nbody_shard_plan_t& p = s->plan;
    auto cleanup = [&](int rc) { nbody_shard_destroy(s); return rc; };
    auto alloc = [&](float4** ptr, size_t bodies) -> int {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(ptr), (bodies ? bodies : 1) * sizeof(float4)));
        HIP_TRY(hipMemset(*ptr, 0, (bodies ? bodies : 1) * sizeof(float4)));
        return NBODY_OK;
    };
    if (int rc = alloc(&s->x, p.n_pad)) return cleanup(rc);
    if (int rc = alloc(&s->v, p.shard)) return cleanup(rc);
    if (int rc = alloc(&s->a, p.shard)) return cleanup(rc);
    if (int rc = alloc(&s->jbuf, p.jbuf_bodies)) return cleanup(rc);
    if (int rc = alloc(&s->rbuf, p.rbuf_bodies)) return cleanup(rc);
    // (the communication stream is made on first use: nbody_shard_set_comm_priority may still choose its priority)
    for (hipEvent_t* ev : {&s->ev_integrated, &s->ev_gathered, &s->ev_cross, &s->ev_exchanged})
        if (int rc = new_event(ev, false)) return cleanup(rc);
    hipError_t e = hipDeviceSynchronize();  // the zero fills above ran on the null stream
    if (e != hipSuccess) return cleanup(nbody_fail(NBODY_ERR_HIP, "hipDeviceSynchronize failed: %s", hipGetErrorString(e)));
    // size the context's slab workspace once, for the largest launch of this rank
    if (p.shard > 0)
        if (int rc = nbody_ctx_reserve(ctx, p.shard)) return cleanup(rc);
    *out = s;
    return NBODY_OK;
}

int nbody_shard_destroy(nbody_shard* s)
{
    if (!s) return NBODY_OK;
    DeviceScope scope(s->device);
    if (s->compute) (void)hipStreamSynchronize(s->compute);
    if (s->comm) (void)hipStreamSynchronize(s->comm);
    for (float4* ptr : {s->x, s->v, s->a, s->jbuf, s->rbuf})
        if (ptr) (void)hipFree(ptr);
    for (hipEvent_t ev : {s->ev_integrated, s->ev_gathered, s->ev_cross, s->ev_exchanged})
        if (ev) (void)hipEventDestroy(ev);
    for (auto& t : s->timed)
        for (hipEvent_t ev : {t.g0, t.g1, t.local_done, t.x0, t.x1, t.own_done})
            if (ev) (void)hipEventDestroy(ev);
    if (s->comm) (void)hipStreamDestroy(s->comm);
    delete s;
    return NBODY_OK;
}

int nbody_shard_get_plan(nbody_shard* s, nbody_shard_plan_t* out)
{
    if (int rc = check_shard(s)) return rc;
    if (!out) return nbody_fail(NBODY_ERR_INVALID, "null out");
    *out = s->plan;
    return NBODY_OK;
}

int nbody_shard_buffers(nbody_shard* s, nbody_float4** d_x_full, nbody_float4** d_v_own, nbody_float4** d_a_own,
                        nbody_float4** d_jbuf, nbody_float4** d_rbuf)
{
    if (int rc = check_shard(s)) return rc;
    if (d_x_full) *d_x_full = nb(s->x);
    if (d_v_own) *d_v_own = nb(s->v);
    if (d_a_own) *d_a_own = nb(s->a);
    if (d_jbuf) *d_jbuf = nb(s->jbuf);
    if (d_rbuf) *d_rbuf = nb(s->rbuf);
    return NBODY_OK;
}

int nbody_shard_upload(nbody_shard* s, const nbody_float4* h_bodies)
{
    if (int rc = check_shard(s)) return rc;
    const nbody_shard_plan_t& p = s->plan;
    if (p.n_total > 0 && !h_bodies) return nbody_fail(NBODY_ERR_INVALID, "null host pointer");
    DeviceScope scope(s->device);
    HIP_TRY(hipStreamSynchronize(s->compute));
    if (s->comm) HIP_TRY(hipStreamSynchronize(s->comm));
    std::vector<nbody_float4> padded((size_t)p.n_pad);
    if (p.n_total > 0) std::memcpy(padded.data(), h_bodies, (size_t)p.n_total * sizeof(nbody_float4));
    for (int i = p.n_total; i < p.n_pad; ++i) {  // massless, on top of body 0: adds exactly +-0 to every sum
        padded[i] = p.n_total > 0 ? h_bodies[0] : nbody_float4{0, 0, 0, 0};
        padded[i].w = 0.0f;
    }
    if (p.n_pad > 0) HIP_TRY(hipMemcpy(s->x, padded.data(), (size_t)p.n_pad * sizeof(float4), hipMemcpyHostToDevice));
    if (p.shard > 0) {
        HIP_TRY(hipMemset(s->v, 0, (size_t)p.shard * sizeof(float4)));
        HIP_TRY(hipMemset(s->a, 0, (size_t)p.shard * sizeof(float4)));
    }
    HIP_TRY(hipDeviceSynchronize());
    s->fresh = true;
    return NBODY_OK;
}

int nbody_shard_upload_velocity(nbody_shard* s, const nbody_float4* h_velocity)
{
    if (int rc = check_shard(s)) return rc;
    const nbody_shard_plan_t& p = s->plan;
    if (p.n_total > 0 && !h_velocity) return nbody_fail(NBODY_ERR_INVALID, "null host pointer");
    DeviceScope scope(s->device);
    HIP_TRY(hipStreamSynchronize(s->compute));
    if (s->comm) HIP_TRY(hipStreamSynchronize(s->comm));
    if (p.shard == 0) return NBODY_OK;
    std::vector<nbody_float4> own((size_t)p.shard, nbody_float4{0, 0, 0, 0});  // padding bodies stay at rest
    for (int i = p.i0; i < p.i1 && i < p.n_total; ++i) own[(size_t)(i - p.i0)] = h_velocity[i];
    HIP_TRY(hipMemcpy(s->v, own.data(), (size_t)p.shard * sizeof(float4), hipMemcpyHostToDevice));
    return NBODY_OK;
}

int nbody_shard_download(nbody_shard* s, nbody_float4* h_x_own, nbody_float4* h_v_own, nbody_float4* h_a_own)
{
    if (int rc = check_shard(s)) return rc;
    const nbody_shard_plan_t& p = s->plan;
    DeviceScope scope(s->device);
    HIP_TRY(hipStreamSynchronize(s->compute));
    if (s->comm) HIP_TRY(hipStreamSynchronize(s->comm));
    const size_t bytes = (size_t)p.shard * sizeof(float4);
    if (bytes == 0) return NBODY_OK;
    if (h_x_own) HIP_TRY(hipMemcpy(h_x_own, s->x + p.i0, bytes, hipMemcpyDeviceToHost));
    if (h_v_own) HIP_TRY(hipMemcpy(h_v_own, s->v, bytes, hipMemcpyDeviceToHost));
    if (h_a_own) HIP_TRY(hipMemcpy(h_a_own, s->a, bytes, hipMemcpyDeviceToHost));
    return NBODY_OK;
}

int nbody_shard_step_phase(nbody_shard* s, int phase)
{
    if (int rc = check_shard(s)) return rc;
    DeviceScope scope(s->device);
    switch (phase) {
        case 0: return phase_gather(s);
        case 1: return phase_compute(s);
        case 2: return phase_exchange(s);
        case 3: return phase_finish(s);
        default: return nbody_fail(NBODY_ERR_INVALID, "phase must be 0..3 (got %d)", phase);
    }
}

int nbody_shard_step(nbody_shard* s, int steps)
{
    if (int rc = check_shard(s)) return rc;
    if (steps < 0) return nbody_fail(NBODY_ERR_INVALID, "steps=%d", steps);
    DeviceScope scope(s->device);
    for (int k = 0; k < steps; ++k) {
        if (int rc = phase_gather(s)) return rc;
        if (int rc = phase_compute(s)) return rc;
        if (int rc = phase_exchange(s)) return rc;
        if (int rc = phase_finish(s)) return rc;
    }
    return NBODY_OK;
}

int nbody_shard_sync(nbody_shard* s)
{
    if (int rc = check_shard(s)) return rc;
    DeviceScope scope(s->device);
    HIP_TRY(hipStreamSynchronize(s->compute));
    if (s->comm) HIP_TRY(hipStreamSynchronize(s->comm));
    return NBODY_OK;
}

int nbody_shard_comm_timing(nbody_shard* s, int enable)
{
    if (int rc = check_shard(s)) return rc;
    if (int rc = nbody_shard_sync(s)) return rc;
    for (auto& t : s->timed)
        for (hipEvent_t ev : {t.g0, t.g1, t.local_done, t.x0, t.x1, t.own_done})
            if (ev) (void)hipEventDestroy(ev);
    s->timed.clear();
    s->timing = enable != 0;
    return NBODY_OK;
}

int nbody_shard_set_comm_priority(nbody_shard* s, int high)
{
    if (int rc = check_shard(s)) return rc;
    if (int rc = nbody_shard_sync(s)) return rc;
    if ((high != 0) == s->comm_high) return NBODY_OK;
    s->comm_high = high != 0;
    if (!s->comm) return NBODY_OK;          // not made yet: the first use makes it with the new priority
    DeviceScope scope(s->device);
    hipStream_t old = s->comm;
    s->comm = nullptr;
    if (int rc = make_comm_stream(s, high != 0)) {
        s->comm = old;
        s->comm_high = !s->comm_high;
        return rc;
    }
    (void)hipStreamDestroy(old);
    return NBODY_OK;
}

int nbody_shard_comm_report(nbody_shard* s, int* steps, double* gather_ms, double* gather_exposed_ms, double* exchange_ms,
                            double* exchange_exposed_ms)
{
    if (int rc = check_shard(s)) return rc;
    if (int rc = nbody_shard_sync(s)) return rc;
    double g = 0, ge = 0, x = 0, xe = 0;
    int ng = 0, nx = 0;
    for (auto& t : s->timed) {
        float ms = 0;
        if (t.gathered) {
            HIP_TRY(hipEventElapsedTime(&ms, t.g0, t.g1));
            g += ms;
            // not hidden = from the end of the own-block pass to the end of the all-gather (0 when the gather ended first)
            if (hipEventElapsedTime(&ms, t.local_done, t.g1) == hipSuccess && ms > 0) ge += ms;
            ++ng;
        }
        if (t.exchanged) {
            HIP_TRY(hipEventElapsedTime(&ms, t.x0, t.x1));
            x += ms;
            // not hidden = from the end of the second half of the own block to the end of the exchange
            if (hipEventElapsedTime(&ms, t.own_done, t.x1) == hipSuccess && ms > 0) xe += ms;
            ++nx;
        }
    }
    if (steps) *steps = (int)s->timed.size();
    if (gather_ms) *gather_ms = ng ? g / ng : 0.0;
    if (gather_exposed_ms) *gather_exposed_ms = ng ? ge / ng : 0.0;
    if (exchange_ms) *exchange_ms = nx ? x / nx : 0.0;
    if (exchange_exposed_ms) *exchange_exposed_ms = nx ? xe / nx : 0.0;
    return NBODY_OK;
}

// ---- RCCL (librccl.so, loaded on first use) -------------------------------------------------------------------------

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

// nbody_shard.hip — one rank of the sharded (multi-GPU) step behind the C-ABI (include/nbody.h, "sharded step"); its two
// collectives are callbacks (the built-in transports: nbody_comm.hip). The reference runs on one device only
// (TestProject/kernel.cu:630, main.cpp:287); this is the build's own decomposition. The pair
// arithmetic is entirely nbody_accel_range / nbody_accel_wrapped / nbody_accel_cross / nbody_integrate_range of
// nbody_step.hip: this file only orders them on two streams.
#include "nbody_internal.hip.h"

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

namespace {

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
    // communication timing: a RING of kTimedRing step records whose events are reused; a record is folded into the running sums
    // (mean AND maximum over the steps: one late all-gather in twenty is invisible in a mean) before its events are recorded again
    bool timing = false;
    struct StepEvents {
        hipEvent_t g0, g1, local_done, x0, x1, own_done;
        bool gathered, exchanged, local_marked, own_marked, open;
    };
    std::vector<StepEvents> ring;
    size_t timed_steps = 0;     // steps begun with timing on since nbody_shard_comm_timing(.., 1)
    size_t cur = 0;             // ring slot of the step being issued
    nbody_comm_report_t agg{};  // what the folded records add up to (the *_ms fields hold SUMS until a report divides them)
};

namespace {

constexpr size_t kTimedRing = 64;

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

// Adds one finished step record to the running sums (waits for its events: a record is folded either when its ring slot comes up
// again, kTimedRing steps later, or by a report).
int fold_step(nbody_shard* s, nbody_shard::StepEvents& t)
{
    if (!t.open) return NBODY_OK;
    nbody_comm_report_t& a = s->agg;
    float ms = 0;
    if (t.gathered) {
        HIP_TRY(hipEventSynchronize(t.g1));
        HIP_TRY(hipEventElapsedTime(&ms, t.g0, t.g1));
        a.gather_ms += ms;
        if (ms > a.gather_ms_max) a.gather_ms_max = ms;
        // not hidden = from the end of the own-block pass to the end of the all-gather (0 when the gather ended first)
        double e = 0;
        if (t.local_marked) {
            HIP_TRY(hipEventSynchronize(t.local_done));
            if (hipEventElapsedTime(&ms, t.local_done, t.g1) == hipSuccess && ms > 0) e = ms;
            else (void)hipGetLastError();
        }
        a.gather_exposed_ms += e;
        if (e > a.gather_exposed_ms_max) a.gather_exposed_ms_max = e;
        ++a.gathers;
    }
    if (t.exchanged) {
        HIP_TRY(hipEventSynchronize(t.x1));
        HIP_TRY(hipEventElapsedTime(&ms, t.x0, t.x1));
        a.exchange_ms += ms;
        if (ms > a.exchange_ms_max) a.exchange_ms_max = ms;
        // not hidden = from the end of the second half of the own block to the end of the exchange
        double e = 0;
        if (t.own_marked) {
            HIP_TRY(hipEventSynchronize(t.own_done));
            if (hipEventElapsedTime(&ms, t.own_done, t.x1) == hipSuccess && ms > 0) e = ms;
            else (void)hipGetLastError();
        }
        a.exchange_exposed_ms += e;
        if (e > a.exchange_exposed_ms_max) a.exchange_exposed_ms_max = e;
        ++a.exchanges;
    }
    t.open = t.gathered = t.exchanged = t.local_marked = t.own_marked = false;
    return NBODY_OK;
}

// the ring slot of the step about to be issued: a new record while the ring grows, otherwise the oldest one, folded first
int begin_timed_step(nbody_shard* s)
{
    const size_t slot = s->timed_steps % kTimedRing;
    if (slot == s->ring.size()) {
        nbody_shard::StepEvents ev{};
        for (hipEvent_t* e : {&ev.g0, &ev.g1, &ev.local_done, &ev.x0, &ev.x1, &ev.own_done})
            if (int rc = new_event(e, true)) return rc;
        s->ring.push_back(ev);
    } else if (int rc = fold_step(s, s->ring[slot])) {
        return rc;
    }
    s->ring[slot].open = true;
    s->cur = slot;
    ++s->timed_steps;
    return NBODY_OK;
}

int phase_gather(nbody_shard* s)
{
    const nbody_shard_plan_t& p = s->plan;
    s->gather_posted = false;
    if (s->timing) if (int rc = begin_timed_step(s)) return rc;
    if (p.world == 1 || s->fresh) return NBODY_OK;
    if (int rc = ensure_comm(s)) return rc;
    HIP_TRY(hipStreamWaitEvent(s->comm, s->ev_integrated, 0));  // the own block was advanced by the last integrate
    if (s->timing) HIP_TRY(hipEventRecord(s->ring[s->cur].g0, s->comm));
    if (s->cb.all_gather(s->cb.user, nb(s->x), p.shard, static_cast<void*>(s->comm)) != 0)
        return nbody_fail(NBODY_ERR_HIP, "all-gather callback failed on rank %d", p.rank);
    if (s->timing) {
        HIP_TRY(hipEventRecord(s->ring[s->cur].g1, s->comm));
        s->ring[s->cur].gathered = true;
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
        if (s->timing && s->gather_posted) HIP_TRY(hipEventRecord(s->ring[s->cur].local_done, s->compute));
        if (s->gather_posted) HIP_TRY(hipStreamWaitEvent(s->compute, s->ev_gathered, 0));
        if (p.world == 1) return NBODY_OK;
        // everybody else in one launch: sources i1, i1+1, ... wrapping round to i0-1
        return nbody_accel_wrapped(c, nb(s->x), p.n_pad, nb(s->a), p.i0, p.i1, p.i1 % p.n_pad, p.n_pad - p.shard, 1);
    }
    // symmetric schedule. The own block against itself is issued in two halves: the first hides the all-gather, the
    // second (phase_finish) hides the exchange. The sums land in `a` in a fixed order: cross launches, own block, received.
    HIP_TRY(hipMemsetAsync(s->a, 0, (size_t)p.shard * sizeof(float4), s->compute));
    if (int rc = nbody_accel_square_part(c, nb(s->x), nb(s->a), p.i0, p.i1, 1, 0, p.world > 1 ? 2 : 1)) return rc;
    if (s->timing && s->gather_posted) { HIP_TRY(hipEventRecord(s->ring[s->cur].local_done, s->compute)); s->ring[s->cur].local_marked = true; }
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
    if (s->timing) HIP_TRY(hipEventRecord(s->ring[s->cur].x0, s->comm));
    if (s->cb.exchange(s->cb.user, p.send, p.n_sends, nb(s->jbuf), p.recv, p.n_recvs, nb(s->rbuf), static_cast<void*>(s->comm)) != 0)
        return nbody_fail(NBODY_ERR_HIP, "exchange callback failed on rank %d", p.rank);
    if (s->timing) {
        HIP_TRY(hipEventRecord(s->ring[s->cur].x1, s->comm));
        s->ring[s->cur].exchanged = true;
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
        if (s->timing && !s->ring.empty() && s->ring[s->cur].exchanged) { HIP_TRY(hipEventRecord(s->ring[s->cur].own_done, s->compute)); s->ring[s->cur].own_marked = true; }
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
    for (auto& t : s->ring)
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
    for (auto& t : s->ring) t.open = t.gathered = t.exchanged = t.local_marked = t.own_marked = false;   // (the events stay: reused)
    s->timed_steps = 0;
    s->cur = 0;
    s->agg = nbody_comm_report_t{};
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

int nbody_shard_comm_report_ex(nbody_shard* s, nbody_comm_report_t* out)
{
    if (int rc = check_shard(s)) return rc;
    if (!out) return nbody_fail(NBODY_ERR_INVALID, "null out");
    if (int rc = nbody_shard_sync(s)) return rc;
    DeviceScope scope(s->device);
    for (auto& t : s->ring)
        if (int rc = fold_step(s, t)) return rc;
    nbody_comm_report_t r = s->agg;   // sums so far; the report divides
    r.steps = (int)s->timed_steps;
    r.records_kept = (int)s->ring.size();
    if (r.gathers) { r.gather_ms /= r.gathers; r.gather_exposed_ms /= r.gathers; }
    if (r.exchanges) { r.exchange_ms /= r.exchanges; r.exchange_exposed_ms /= r.exchanges; }
    *out = r;
    return NBODY_OK;
}

int nbody_shard_comm_report(nbody_shard* s, int* steps, double* gather_ms, double* gather_exposed_ms, double* exchange_ms,
                            double* exchange_exposed_ms)
{
    nbody_comm_report_t r{};
    if (int rc = nbody_shard_comm_report_ex(s, &r)) return rc;
    if (steps) *steps = r.steps;
    if (gather_ms) *gather_ms = r.gather_ms;
    if (gather_exposed_ms) *gather_exposed_ms = r.gather_exposed_ms;
    if (exchange_ms) *exchange_ms = r.exchange_ms;
    if (exchange_exposed_ms) *exchange_exposed_ms = r.exchange_exposed_ms;
    return NBODY_OK;
}

}  // extern "C"

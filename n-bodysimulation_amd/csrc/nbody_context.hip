// nbody_context.hip — the part of include/nbody.h that is not stepping: error text, contexts and their knobs, the workspaces a
// context owns, event timing, the memory helpers of the drop-in boundary. No CPU compute path exists in this library on purpose.
#include "nbody_ctx.hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace nbi;

namespace {
thread_local char g_err[512] = "";
}

// records the calling thread's error message (also used by nbody_shard.hip); returns `code`
__attribute__((visibility("hidden"))) int nbody_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#pragma GCC visibility push(hidden)
namespace nbi {

// the fused step's spare position array
int ensure_xalt(nbody_ctx* c, int n)
{
    const size_t bytes = (size_t)n * sizeof(float4);
    if (bytes <= c->xalt_bytes) return NBODY_OK;
    if (c->xalt) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(c->xalt));
        c->xalt = nullptr;
        c->xalt_bytes = 0;
    }
    const hipError_t e = hipMalloc(&c->xalt, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->xalt = nullptr;
        return fail(NBODY_ERR_NOMEM, "cannot allocate the fused step's spare position array (%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    c->xalt_bytes = bytes;
    return NBODY_OK;
}

// the in-place fused step's counters and per-wave marks (zeroed once: every launch leaves them zero), and its host-mapped word
int ensure_fsync(nbody_ctx* c, size_t nwaves)
{
    if (!c->fhost) {
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->fhost), 64, hipHostMallocMapped | hipHostMallocCoherent);
        if (e == hipSuccess) {
            c->fhost[0] = 0;
            e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->fhost_dev), c->fhost, 0);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (c->fhost) (void)hipHostFree(c->fhost);
            c->fhost = c->fhost_dev = nullptr;
            return fail(NBODY_ERR_NOMEM, "cannot allocate the fused step's host-mapped word: %s", hipGetErrorString(e));
        }
    }
    if (c->fsync && nwaves <= c->fsync_waves) return NBODY_OK;
    if (c->fsync) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(c->fsync));
        c->fsync = nullptr;
        c->fsync_waves = 0;
    }
    const size_t waves = nwaves < 4096 ? 4096 : nwaves;
    const size_t bytes = nbk::kFusedSyncWords * sizeof(unsigned) + waves;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->fsync), bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->fsync = nullptr;
        return fail(NBODY_ERR_NOMEM, "cannot allocate the fused step's counters (%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    HIP_TRY(hipMemsetAsync(c->fsync, 0, bytes, c->stream));
    c->fsync_waves = waves;
    return NBODY_OK;
}

// the ticket kernel's per-block words (zeroed once: every launch leaves them zero) and its host-mapped error word
int ensure_tickets(nbody_ctx* c)
{
    if (!c->terr) {
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->terr), 64, hipHostMallocMapped | hipHostMallocCoherent);
        if (e == hipSuccess) {
            c->terr[0] = 0;
            e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->terr_dev), c->terr, 0);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (c->terr) (void)hipHostFree(c->terr);
            c->terr = c->terr_dev = nullptr;
            return fail(NBODY_ERR_NOMEM, "cannot allocate the ticket kernel's host-mapped word: %s", hipGetErrorString(e));
        }
    }
    if (c->tickets) return NBODY_OK;
    static_assert(nbk::kTicketWords == kSymMaxSlabs * nbk::kTicketMaxLanes, "one ticket per block and lane");
    const size_t bytes = (size_t)(nbk::kTicketWords + 1) * sizeof(unsigned);   // + the abort word
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->tickets), bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->tickets = nullptr;
        return fail(NBODY_ERR_NOMEM, "cannot allocate the ticket words: %s", hipGetErrorString(e));
    }
    HIP_TRY(hipMemsetAsync(c->tickets, 0, bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // once per context: the zeroes are there whichever stream the context launches on later
    return NBODY_OK;
}

// A ticket wait that timed out (never on a healthy run) leaves sums that are not to be trusted and tickets that are not zero: the
// next call that looks says so ONCE, after putting the tickets back (so that stepping can go on from whatever state the caller restores).
int ticket_error(nbody_ctx* c)
{
    if (!c->terr || !*static_cast<volatile unsigned*>(c->terr)) return NBODY_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    *static_cast<volatile unsigned*>(c->terr) = 0;
    if (c->tickets) HIP_TRY(hipMemsetAsync(c->tickets, 0, (size_t)(nbk::kTicketWords + 1) * sizeof(unsigned), c->stream));
    return fail(NBODY_ERR_HIP, "a workgroup of the in-place block-pair kernel waited more than 10 s for its turn to add (a predecessor did not finish): "
                               "the accelerations of that step are incomplete");
}

// Recomputes the effective workspace cap from what the device has free right now (the context's own workspaces count as
// available: they are released before a larger one is allocated). Needs the context's device to be current.
void refresh_ws_cap(nbody_ctx* c)
{
    size_t cap = kSymMaxWorkspace;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t avail = (free_b + c->slab_bytes + c->xslab_bytes) / 2;
        if (avail < cap) cap = avail;
    } else {
        (void)hipGetLastError();
    }
    if (c->ws_limit && c->ws_limit < cap) cap = c->ws_limit;
    if (c->ws_fail_above_limit) cap = kSymMaxWorkspace;   // test hook: let the shape choice ask for it, and the allocation fail
    c->ws_cap = cap;
}

// Grows a workspace to `bytes`. An allocation that fails is not an error of the step: the cap is lowered below the request and
// NBODY_ERR_NOMEM returned, so that the caller re-resolves its launch shape (a smaller symmetric footprint, finally the
// one-sided kernel's <= 64 slabs) — only when nothing smaller exists does the failure reach the user.
int grow_workspace(nbody_ctx* c, void** buf, size_t* have, size_t bytes)
{
    if (bytes <= *have) return NBODY_OK;
    if (*buf) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    hipError_t e = (c->ws_fail_above_limit && c->ws_limit && bytes > c->ws_limit) ? hipErrorOutOfMemory : hipMalloc(buf, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // the failed allocation must not surface at the next launch check
        *buf = nullptr;
        refresh_ws_cap(c);
        if (c->ws_cap >= bytes) c->ws_cap = bytes - 1;
        return fail(NBODY_ERR_NOMEM, "cannot allocate a workspace of %zu bytes: %s", bytes, hipGetErrorString(e));
    }
    *have = bytes;
    return NBODY_OK;
}

int ensure_xslabs(nbody_ctx* c, size_t bytes) { return grow_workspace(c, &c->xslabs, &c->xslab_bytes, bytes); }
int ensure_slabs(nbody_ctx* c, size_t bytes)
{
    c->ws_tag = 0;   // whoever asks for slabs overwrites what a balanced-run layout keeps cleared
    return grow_workspace(c, &c->slabs, &c->slab_bytes, bytes);
}

// The inbox workspace of a balanced-run layout: records of unit pieces that do not exist are never written and must read as zero,
// so the workspace is cleared once per layout (and again whenever anything else has used it in between).
int ensure_inbox(nbody_ctx* c, const BalShape& b)
{
    const nbk::BalLayout& y = b.y;
    unsigned long long tag = 0x9E3779B97F4A7C15ull;
    for (unsigned long long v : {(unsigned long long)y.bpl, (unsigned long long)y.ncht, (unsigned long long)y.L, (unsigned long long)y.smax,
                                 (unsigned long long)y.pmax, (unsigned long long)y.wv, (unsigned long long)y.nsteps})
        tag = (tag ^ v) * 0xBF58476D1CE4E5B9ull + 1;
    const void* before = c->slabs;
    const size_t had = c->slab_bytes;
    if (int rc = grow_workspace(c, &c->slabs, &c->slab_bytes, b.bytes)) { c->ws_tag = 0; return rc; }
    if (c->slabs != before || c->slab_bytes != had || c->ws_tag != tag) {
        HIP_TRY(hipMemsetAsync(c->slabs, 0, b.bytes, c->stream));
        c->ws_tag = tag;
    }
    return NBODY_OK;
}

// the clock stamps' device scratch and host-mapped per-launch records (allocated on first use)
static int ensure_cstamp(nbody_ctx* c)
{
    if (c->cdelta && c->cscratch) return NBODY_OK;
    const size_t bytes = kClockCapLaunches * sizeof(nbk::ClockDelta);
    if (!c->cscratch) {
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->cscratch), nbk::kClockBeginWgs * sizeof(nbk::ClockStamp));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c->cscratch = nullptr;
            return fail(NBODY_ERR_NOMEM, "cannot allocate the clock-stamp scratch: %s", hipGetErrorString(e));
        }
    }
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->cdelta), bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->cdelta_dev), c->cdelta, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (c->tickets) (void)hipFree(c->tickets);
    if (c->terr) (void)hipHostFree(c->terr);
    if (c->cdelta) (void)hipHostFree(c->cdelta);
        c->cdelta = c->cdelta_dev = nullptr;
        return fail(NBODY_ERR_NOMEM, "cannot allocate the clock-stamp records (%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    std::memset(c->cdelta, 0, bytes);
    return NBODY_OK;
}

// Called in front of and behind every timed force launch. Events alone (timing on), or — clock option — a clock stamp OUTSIDE
// the event pair on either side: clock_begin, event, <force launch>, event, clock_end. The events keep measuring the force kernel alone;
// the two stamps bracket it plus two launch boundaries (a few microseconds: 0.05 % of a 10-ms launch).
int time_mark(nbody_ctx* c)
{
    if (!c->timing) return NBODY_OK;
    const bool begin = (c->events_used & 1) == 0;
    if (begin && c->clock_stamps) {
        c->clock_pair_open = false;
        if (c->cdelta_used < kClockCapLaunches && ensure_cstamp(c) == NBODY_OK) {
            if (int rc = launch_clock_stamp(c, nullptr)) return rc;
            c->clock_pair_open = true;
        }
    }
    if (c->events_used == c->events.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->events.push_back(e);
    }
    HIP_TRY(hipEventRecord(c->events[c->events_used++], c->stream));
    if (!begin && c->clock_pair_open) {
        c->clock_pair_open = false;
        if (int rc = launch_clock_stamp(c, c->cdelta_dev + c->cdelta_used)) return rc;
        ++c->cdelta_used;
    }
    return NBODY_OK;
}

int check_ctx(const nbody_ctx* c)
{
    if (!c) return fail(NBODY_ERR_INVALID, "null context");
    return NBODY_OK;
}

// One default context per device, created on first use and released at process exit. A recursive
// mutex serialises the entry points that work on them (nbody_simulate, nbody_simulate_host_legacy).
std::recursive_mutex g_default_mu;

}  // namespace nbi
#pragma GCC visibility pop

namespace {

struct DefaultContexts {
    nbody_ctx* ctx[kMaxDevices] = {};
    ~DefaultContexts()
    {
        // at static destruction the HIP runtime may already be gone: release host state only
        for (nbody_ctx*& c : ctx) { delete c; c = nullptr; }
    }
} g_default;

}  // namespace

extern "C" {

const char* nbody_last_error(void) { return g_err; }

const char* nbody_version(void)
{
    return "nbody_hip 0.6 gfx950 fast=symmetric-dpp(w4,bpl10)+equal-mass-path+sums-in-place-beyond-the-workspace-cap(tickets,8-lanes)|symmetric-balanced-runs(8k-45k)|fused-step(<=8k;simulate:in-place,stores-drained+host-word)|onesided-lds-packed(bpl4,tile2048,u8) strict=ieee-seq f64=symmetric-dpp(w4,bpl6)+equal-mass-path|lds";
}

int nbody_device_count(int* count)
{
    if (!count) return fail(NBODY_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(NBODY_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return NBODY_OK;
}

int nbody_ctx_create(nbody_ctx** out, int device)
{
    if (!out) return fail(NBODY_ERR_INVALID, "null out");
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(NBODY_ERR_HIP, "no HIP device visible");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(NBODY_ERR_INVALID, "device %d out of range (%d devices)", device, ndev);
    DeviceScope guard(device);
    if (guard.err != hipSuccess) return fail(NBODY_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(guard.err));
    nbody_ctx* c = new (std::nothrow) nbody_ctx();
    if (!c) return fail(NBODY_ERR_NOMEM, "out of host memory");
    c->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        delete c;
        return fail(NBODY_ERR_HIP, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    }
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e = hipStreamCreate(&c->own_stream);
    if (e != hipSuccess) {
        delete c;
        return fail(NBODY_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    refresh_ws_cap(c);   // half of what the device has free now bounds one workspace (re-read whenever a workspace grows)
    *out = c;
    return NBODY_OK;
}

int nbody_ctx_destroy(nbody_ctx* c)
{
    if (!c) return NBODY_OK;
    DeviceScope guard(c->device);
    if (c->slabs) (void)hipFree(c->slabs);
    if (c->xslabs) (void)hipFree(c->xslabs);
    if (c->xalt) (void)hipFree(c->xalt);
    if (c->fsync) (void)hipFree(c->fsync);
    if (c->fhost) (void)hipHostFree(c->fhost);
    if (c->tickets) (void)hipFree(c->tickets);
    if (c->terr) (void)hipHostFree(c->terr);
    if (c->cdelta) (void)hipHostFree(c->cdelta);
    if (c->cscratch) (void)hipFree(c->cscratch);
    if (c->eqm) (void)hipFree(c->eqm);
    if (c->legacy_buf) (void)hipFree(c->legacy_buf);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return NBODY_OK;
}

int nbody_default_ctx(nbody_ctx** out)
{
    if (!out) return fail(NBODY_ERR_INVALID, "null out");
    // The reference never selects a device: its launch goes to the caller's CURRENT device (kernel.cu:630
    // only queries the properties of device 0). So the default context is the current device's.
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) return fail(NBODY_ERR_INVALID, "device %d not supported by the default context", device);
    std::lock_guard<std::recursive_mutex> lk(g_default_mu);
    if (!g_default.ctx[device]) {
        int rc = nbody_ctx_create(&g_default.ctx[device], device);
        if (rc != NBODY_OK) return rc;
    }
    *out = g_default.ctx[device];
    return NBODY_OK;
}

int nbody_ctx_set_params(nbody_ctx* c, float dt, float eps2)
{
    if (int rc = check_ctx(c)) return rc;
    if (!(eps2 > 0.0f) || !std::isfinite(eps2)) return fail(NBODY_ERR_INVALID, "eps2 must be finite and > 0 (got %g)", eps2);
    if (!std::isfinite(dt)) return fail(NBODY_ERR_INVALID, "dt must be finite (got %g)", dt);
    c->dt = dt;
    c->eps2 = eps2;
    return NBODY_OK;
}

int nbody_ctx_set_kernel(nbody_ctx* c, int kernel, int tile, int bodies_per_lane, int jsplit)
{
    if (int rc = check_ctx(c)) return rc;
    if (kernel != NBODY_KERNEL_FAST && kernel != NBODY_KERNEL_STRICT && kernel != NBODY_KERNEL_ONESIDED &&
        kernel != NBODY_KERNEL_SYMMETRIC)
        return fail(NBODY_ERR_CONFIG, "unknown kernel %d", kernel);
    if (tile != 0 && tile != 256 && tile != 512 && tile != 1024 && tile != 2048)
        return fail(NBODY_ERR_CONFIG, "tile must be 0 (auto), 256, 512, 1024 or 2048 (got %d)", tile);
    if (bodies_per_lane != 0 && bodies_per_lane != 1 && bodies_per_lane != 2 && bodies_per_lane != 4)
        return fail(NBODY_ERR_CONFIG, "bodies_per_lane must be 0 (auto), 1, 2 or 4 (got %d)", bodies_per_lane);
    if (tile == 2048 && bodies_per_lane != 4 && bodies_per_lane != 0)
        return fail(NBODY_ERR_CONFIG, "tile 2048 is built for bodies_per_lane 4 only");
    if (jsplit < 0 || jsplit > kMaxSplit) return fail(NBODY_ERR_CONFIG, "jsplit must be in [0,%d] (got %d)", kMaxSplit, jsplit);
    c->kernel = kernel;
    c->tile = tile;
    c->bpl = bodies_per_lane;
    c->jsplit = jsplit;
    return NBODY_OK;
}

int nbody_ctx_set_symmetric_shape(nbody_ctx* c, int waves, int bodies_per_lane)
{
    if (int rc = check_ctx(c)) return rc;
    bool ok = (waves == 0 && bodies_per_lane == 0);
    for (int k = 0; k < kSymCands && !ok; ++k)
        ok = (waves == 0 || waves == kSymCand[k][0]) && (bodies_per_lane == 0 || bodies_per_lane == kSymCand[k][1]);
    if (waves == 4 && bodies_per_lane == 6) ok = true;  // fp64 only
    if (!ok)
        return fail(NBODY_ERR_CONFIG, "symmetric kernel is built for (waves, bodies_per_lane) in {(4,10),(4,8),(2,10),(2,8),(1,10),(1,8),(2,4),(1,4),(1,2)} "
                    "(fp64: (4,6),(4,8),(2,4),(1,2)); got (%d,%d)", waves, bodies_per_lane);
    c->sym_waves = waves;
    c->sym_bpl = bodies_per_lane;
    return NBODY_OK;
}

int nbody_ctx_set_symmetric_runs(nbody_ctx* c, int mode)
{
    if (int rc = check_ctx(c)) return rc;
    if (mode < -1 || mode > 2) return fail(NBODY_ERR_CONFIG, "runs mode must be -1 (auto), 0 (never), 1 (unit runs always) or 2 (balanced runs always)");
    c->sym_runs = mode;
    return NBODY_OK;
}

int nbody_ctx_set_fused(nbody_ctx* c, int mode)
{
    if (int rc = check_ctx(c)) return rc;
    if (mode < -1 || mode > 1) return fail(NBODY_ERR_CONFIG, "fused mode must be -1 (auto), 0 (never) or 1 (always)");
    c->fused = mode;
    return NBODY_OK;
}

int nbody_ctx_set_fused_inplace(nbody_ctx* c, int mode)
{
    if (int rc = check_ctx(c)) return rc;
    if (mode < -1 || mode > 2)
        return fail(NBODY_ERR_CONFIG, "fused in-place mode must be -1 (auto), 0 (never), 1 (every fused step) or 2 (every step, fall-back path forced)");
    c->fused_inplace = mode;
    return NBODY_OK;
}

int nbody_ctx_fused_inplace_stats(nbody_ctx* c, unsigned long long* out_fallback_waves)
{
    if (int rc = check_ctx(c)) return rc;
    if (!out_fallback_waves) return fail(NBODY_ERR_INVALID, "null out");
    *out_fallback_waves = 0;
    if (!c->fsync) return NBODY_OK;
    ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned total = 0;
    HIP_TRY(hipMemcpy(&total, c->fsync + nbk::kFusedFallbacksTotal, sizeof total, hipMemcpyDeviceToHost));
    *out_fallback_waves = total;
    return NBODY_OK;
}

int nbody_ctx_set_equal_mass(nbody_ctx* c, int mode)
{
    if (int rc = check_ctx(c)) return rc;
    if (mode < -1 || mode > 1) return fail(NBODY_ERR_CONFIG, "equal-mass mode must be -1 (auto: launches of 32768 bodies or more), 0 (never) or 1 (launches of 4096 bodies or more)");
    c->eq_mode = mode;
    return NBODY_OK;
}

int nbody_ctx_equal_mass_verdict(nbody_ctx* c, int* scanned, int* uniform, float* mass)
{
    if (int rc = check_ctx(c)) return rc;
    ON_DEVICE(c);
    if (scanned) *scanned = 0;
    if (uniform) *uniform = 0;
    if (mass) *mass = 0.0f;
    if (!c->eqm || c->eq_gen == 0) return NBODY_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    nbk::MassInfo h[2];
    HIP_TRY(hipMemcpy(h, c->eqm, sizeof h, hipMemcpyDeviceToHost));
    // the most recent scan (either slot) carries the context's current generation unless it found the bodies uniform: a slot
    // whose stamp is the current generation is the latest scan and says "not uniform"
    const bool bad = h[0].bad_gen == c->eq_gen || h[1].bad_gen == c->eq_gen;
    if (scanned) *scanned = 1;
    if (uniform) *uniform = bad ? 0 : 1;
    if (mass) *mass = (float)h[c->eq_last_slot].m0;
    return NBODY_OK;
}

int nbody_ctx_set_workspace_limit(nbody_ctx* c, size_t bytes, int fail_above)
{
    if (int rc = check_ctx(c)) return rc;
    ON_DEVICE(c);
    c->ws_limit = bytes;
    c->ws_fail_above_limit = bytes != 0 && fail_above != 0;
    refresh_ws_cap(c);
    return NBODY_OK;
}

int nbody_ctx_set_inplace_sums(nbody_ctx* c, int mode)
{
    if (int rc = check_ctx(c)) return rc;
    if (mode < -1 || mode > 2) return fail(NBODY_ERR_CONFIG, "in-place sums mode must be -1 (auto), 0 (never), 1 (always) or 2 (always + the stall test hook)");
    c->inplace_sums = mode == 2 ? 1 : mode;
    c->ticket_test_stall = mode == 2;
    return NBODY_OK;
}

int nbody_ctx_set_stream(nbody_ctx* c, void* hip_stream)
{
    if (int rc = check_ctx(c)) return rc;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    if (st != c->stream) c->ws_tag = 0;   // the inbox clear was ordered on the old stream: the next balanced launch clears again on the new one
    c->stream = st;
    return NBODY_OK;
}

int nbody_ctx_reserve(nbody_ctx* c, int n_targets)
{
    if (int rc = check_ctx(c)) return rc;
    if (n_targets < 0) return fail(NBODY_ERR_INVALID, "n_targets < 0");
    ON_DEVICE(c);
    refresh_ws_cap(c);
    if (c->eq_mode != 0 && !c->eqm && n_targets >= (c->eq_mode == 1 ? kEqMinBodies : kEqAutoMinBodies) / 2) {   // the verdict slots of the equal-mass scan, ahead of the first step
        if (hipMalloc(reinterpret_cast<void**>(&c->eqm), 2 * sizeof(nbk::MassInfo)) == hipSuccess) HIP_TRY(hipMemset(c->eqm, 0, 2 * sizeof(nbk::MassInfo)));
        else { (void)hipGetLastError(); c->eqm = nullptr; }
    }
    {
        FusedShape fs{};
        if (fused_wanted(c, n_targets, &fs)) {   // whole steps of this size run the fused kernel; the workspace below
            (void)ensure_xalt(c, n_targets);     // still serves nbody_accel_range on such a block
            if (c->fused_inplace != 0) (void)ensure_fsync(c, (size_t)fs.grid * fs.wv);
        }
        // the device code of this library is loaded by the runtime on first use (milliseconds): now, not inside the first timed step
        load_device_code();
    }
    for (int attempt = 0;; ++attempt) {
        BalShape by{};
        if (bal_wanted(c, n_targets, &by)) {
            const int rc = ensure_inbox(c, by);
            if (rc == NBODY_ERR_NOMEM && attempt < 16) continue;
            return rc;
        }
        SymShape y{};
        RunShape ry{};
        size_t slabs = (size_t)resolve_shape(c, n_targets, n_targets).jsplit;
        bool symmetric = false;
        if (sym_wanted(c, n_targets, &y)) { symmetric = true; if ((size_t)y.nb > slabs) slabs = (size_t)y.nb; }
        if (run_wanted(c, n_targets, &ry)) { symmetric = true; if ((size_t)ry.max_slabs > slabs) slabs = (size_t)ry.max_slabs; }
        const int rc = ensure_slabs(c, slabs * (size_t)n_targets * sizeof(float4));
        // out of memory for a symmetric footprint: the cap has been lowered, the next resolution needs less
        if (rc == NBODY_ERR_NOMEM && symmetric && attempt < 16) continue;
        return rc;
    }
}

// Device-free view of the launch-shape logic (host tests; a context needs a GPU, this does not).

int nbody_ctx_set_graph(nbody_ctx* c, int mode)
{
    if (int rc = check_ctx(c)) return rc;
    if (mode < -1 || mode > 1) return fail(NBODY_ERR_CONFIG, "graph mode must be -1 (auto), 0 (off) or 1 (on)");
    c->use_graph = mode;
    return NBODY_OK;
}

int nbody_ctx_timing(nbody_ctx* c, int enable)
{
    if (int rc = check_ctx(c)) return rc;
    if (enable < 0 || enable > 2) return fail(NBODY_ERR_CONFIG, "timing mode must be 0 (off), 1 (events) or 2 (events + clock stamps)");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->timing = enable != 0;
    c->clock_stamps = enable == 2;
    c->clock_pair_open = false;
    c->events_used = 0;
    if (c->cdelta && c->cdelta_used) std::memset(c->cdelta, 0, c->cdelta_used * sizeof(nbk::ClockDelta));
    c->cdelta_used = 0;
    return NBODY_OK;
}

int nbody_ctx_timing_read(nbody_ctx* c, double* force_ms, int* launches)
{
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double total = 0.0;
    for (size_t k = 0; k + 1 < c->events_used; k += 2) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->events[k], c->events[k + 1]));
        total += ms;
    }
    if (force_ms) *force_ms = total;
    if (launches) *launches = (int)(c->events_used / 2);
    c->events_used = 0;
    return NBODY_OK;
}

int nbody_ctx_clock_read(nbody_ctx* c, nbody_clock_report* out)
{
    if (int rc = check_ctx(c)) return rc;
    if (!out) return fail(NBODY_ERR_INVALID, "null out");
    *out = nbody_clock_report{};
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t launches = c->cdelta_used;
    double cyc[8] = {0}, tk[8] = {0};
    long seen[8] = {0};
    double sum_c = 0, sum_t = 0, lmin = 0, lmax = 0;
    long complete = 0, missing = 0;
    for (size_t l = 0; l < launches; ++l) {
        const nbk::ClockDelta& d = c->cdelta[l];
        double lc = 0, lt = 0;
        int ln = 0;
        for (int x = 0; x < 8; ++x) {
            if (!d.dcycles[x] || !d.dticks[x]) { ++missing; continue; }   // no CU of this XCD was seen by both stamps of this launch
            cyc[x] += (double)d.dcycles[x]; tk[x] += (double)d.dticks[x]; ++seen[x];
            lc += (double)d.dcycles[x]; lt += (double)d.dticks[x]; ++ln;
        }
        if (!ln) continue;
        lc /= ln; lt /= ln;                                              // this launch: mean over the XCDs that answered
        sum_c += lc; sum_t += lt; ++complete;
        if (lmin == 0 || lc < lmin) lmin = lc;
        if (lc > lmax) lmax = lc;
    }
    out->launches = (int)complete;
    out->unpaired = (int)missing;
    if (complete) {
        out->cycles_per_launch = sum_c / (double)complete;
        out->ticks_per_launch = sum_t / (double)complete;
        out->sclk_mhz = sum_c / sum_t * 100.0;
        out->cycles_per_launch_min = lmin;
        out->cycles_per_launch_max = lmax;
        for (int x = 0; x < 8; ++x) {
            if (!seen[x]) continue;
            ++out->xcds;
            const double f = cyc[x] / tk[x] * 100.0;
            if (out->sclk_mhz_min_xcd == 0 || f < out->sclk_mhz_min_xcd) out->sclk_mhz_min_xcd = f;
            if (f > out->sclk_mhz_max_xcd) out->sclk_mhz_max_xcd = f;
        }
    }
    if (c->cdelta && launches) std::memset(c->cdelta, 0, launches * sizeof(nbk::ClockDelta));
    c->cdelta_used = 0;
    c->clock_pair_open = false;
    return NBODY_OK;
}

int nbody_ctx_sync(nbody_ctx* c)
{
    if (int rc = check_ctx(c)) return rc;
    // (a host_signal launch + a spin here, as nbody_simulate does, measured 1 us per call at best — 34.7 -> 33.8 us for a step + sync at
    //  N = 8192 — and nothing for long queues: not adopted, the plain synchronisation stays)
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ticket_error(c);
}

int nbody_ctx_get(nbody_ctx* c, int* device, int* kernel, void** hip_stream)
{
    if (int rc = check_ctx(c)) return rc;
    if (device) *device = c->device;
    if (kernel) *kernel = c->kernel;
    if (hip_stream) *hip_stream = static_cast<void*>(c->stream);
    return NBODY_OK;
}

int nbody_print_device_prop(void)
{
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    printf("== Device Properties ==\n");
    printf("Name: %s\n", prop.name);
    printf("Total global memory: %llu\n", (unsigned long long)prop.totalGlobalMem);
    printf("Total shared memory: %llu\n", (unsigned long long)prop.multiProcessorCount * (unsigned long long)prop.maxSharedMemoryPerMultiProcessor);
    printf("Multiprocessors count: %d\n", prop.multiProcessorCount);
    printf("Shared memory per multiprocessor: %llu\n", (unsigned long long)prop.maxSharedMemoryPerMultiProcessor);
    printf("Shared memory per block: %llu\n", (unsigned long long)prop.sharedMemPerBlock);
    printf("Registers per block: %d\n", prop.regsPerBlock);
    printf("Registers per multiprocessor: %d\n", prop.regsPerMultiprocessor);
    printf("Max (parallel) blocks per multiprocessor: %d\n", prop.maxBlocksPerMultiProcessor);
    printf("Max (parallel) threads per multiprocessor: %d\n", prop.maxThreadsPerMultiProcessor);
    printf("Max grid size: (%d, %d, %d)\n", prop.maxGridSize[0], prop.maxGridSize[1], prop.maxGridSize[2]);
    printf("Max threads per block: %d\n", prop.maxThreadsPerBlock);
    printf("Warp size: %d\n", prop.warpSize);
    printf("\n");
    fflush(stdout);
    return NBODY_OK;
}

// ---- memory helpers ---------------------------------------------------------------------

int nbody_malloc_device(void** d_ptr, size_t bytes)
{
    if (!d_ptr) return fail(NBODY_ERR_INVALID, "null d_ptr");
    *d_ptr = nullptr;
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return NBODY_OK;
}

int nbody_free_device(void* d_ptr)
{
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return NBODY_OK;
}

int nbody_malloc_host(void** h_ptr, size_t bytes)
{
    if (!h_ptr) return fail(NBODY_ERR_INVALID, "null h_ptr");
    *h_ptr = nullptr;
    HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return NBODY_OK;
}

int nbody_free_host(void* h_ptr)
{
    if (h_ptr) HIP_TRY(hipHostFree(h_ptr));
    return NBODY_OK;
}

int nbody_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes)
{
    if (bytes && (!d_dst || !h_src)) return fail(NBODY_ERR_INVALID, "null pointer");
    HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return NBODY_OK;
}

int nbody_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes)
{
    if (bytes && (!h_dst || !d_src)) return fail(NBODY_ERR_INVALID, "null pointer");
    HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return NBODY_OK;
}

int nbody_device_synchronize(void)
{
    HIP_TRY(hipDeviceSynchronize());
    return NBODY_OK;
}

}  // extern "C"

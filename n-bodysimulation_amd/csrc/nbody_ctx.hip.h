// nbody_ctx.hip.h — what the translation units of libnbody_hip.so share (not part of the C-ABI): the context object, the launch
// shapes of the decompositions and the host-side functions that cross a translation unit:
//   nbody_context.hip   error text, contexts and their knobs, workspaces, timing, memory helpers
//   nbody_plan.hip      which decomposition and which block shape for (targets, sources): pure host logic + the nbody_plan_* queries
//   nbody_step.hip      the launchers and the stepping entry points (accel_*, step, simulate, step_f64): the one unit with device code
//   nbody_autotune.hip  measuring the decompositions on the device at hand (opt-in)
//   nbody_shard.hip / nbody_comm.hip   the multi-GPU step and its transports
#pragma once
#include "nbody_internal.hip.h"

#include <map>
#include <mutex>
#include <vector>

#include "nbody_layout.hip.h"

static_assert(sizeof(nbody_float4) == sizeof(float4) && alignof(float4) == 16, "float4 layout");
static_assert(sizeof(nbody_double4) == sizeof(double4), "double4 layout");

#define fail nbody_fail

// Makes the context's device current for the scope of an entry point and puts the caller's device back afterwards
// (the reference never calls cudaSetDevice: a caller's current device must survive our calls).
#define ON_DEVICE(c)                                                                                     \
    DeviceScope device_guard_((c)->device);                                                              \
    if (device_guard_.err != hipSuccess)                                                                 \
        return fail(NBODY_ERR_HIP, "cannot select device %d: %s", (c)->device, hipGetErrorString(device_guard_.err))

constexpr int kMaxSplit = 64;
constexpr int kGraphMaxN = 16384;  // below this a step is a few tens of microseconds: launch-bound
constexpr int kGraphChunk = 32;    // steps per graph launch
constexpr size_t kClockCapLaunches = 4096;   // timed launches whose clock stamps are kept between two reads (further ones go unstamped)

struct nbody_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    float dt = NBODY_DEFAULT_DT;
    float eps2 = NBODY_DEFAULT_EPS2;
    int kernel = NBODY_KERNEL_FAST;
    int tile = 0;    // 0 = auto
    int bpl = 0;     // 0 = auto
    int jsplit = 0;  // 0 = auto
    int sym_waves = 0;  // symmetric kernel: waves per workgroup (0 = auto)
    int sym_bpl = 0;    // symmetric kernel: stationary bodies per lane (0 = auto)
    int sym_runs = -1;  // run-based variant of the symmetric kernel: -1 where the cost estimate prefers it, 0 never, 1 always
    int num_cu = 256;
    void* slabs = nullptr;     // workspace: jsplit slabs of n_targets float4 (or double4)
    size_t slab_bytes = 0;
    size_t ws_limit = 0;       // caller's cap on ONE workspace in bytes (nbody_ctx_set_workspace_limit); 0 = automatic
    size_t ws_cap = (size_t)96 << 30;  // effective cap the shape choice honours: min(96 GiB, ws_limit, half of the device memory that
                               // was free), lowered further whenever an allocation fails (the next choice then needs less)
    void* xalt = nullptr;      // the fused small-N step's second position array (positions alternate between it and the caller's)
    size_t xalt_bytes = 0;
    int fused = -1;            // fused small-N step: -1 where measurements prefer it (FAST, n <= kFusedMaxAuto), 0 never, 1 whenever FAST
    // the fused step IN PLACE (nbk::step_fused<.., INPLACE>): counters + per-wave marks in device memory, a host-mapped word the
    // launch writes when everything is visible (nbody_simulate spins on it instead of paying a stream synchronisation)
    int fused_inplace = -1;    // -1 auto: the single step of nbody_simulate (no copy-back launch, host-mapped completion word); 0 never (two arrays + copy-back);
                               //  1 every fused step; 2 every fused step AND every wave forced down the fall-back path (test hook)
    unsigned* fsync = nullptr;         // nbk::kFusedSyncWords counters, then one byte per wave
    size_t fsync_waves = 0;
    unsigned long long* fhost = nullptr;      // host-mapped 64-bit word: low half = done value, high half = fall-back waves of that launch
    unsigned long long* fhost_dev = nullptr;  // its device address
    unsigned fdone_seq = 0;            // value the last armed launch writes
    bool fdone_armed = false;          // the last fused launch writes fhost[0] = fdone_seq at its end
    bool want_host_done = false;       // nbody_simulate: arm the next in-place launch
    // nbody_simulate near a built-in switch-over size: the decompositions measured once per size on this device (nbody.h)
    struct Tuned { int choice; int fused, sym_runs, sym_bpl, sym_waves; double us_builtin, us_best; };
    std::map<int, Tuned> tuned;
    unsigned long long ws_tag = 0;     // which balanced-run layout the `slabs` workspace is cleared for (0 = none: any other user of it)
    bool ws_fail_above_limit = false;  // test hook: allocations above ws_limit are attempted and FAIL (out of memory) instead of
                               // being avoided by the shape choice
    nbk::MassInfo* eqm = nullptr;  // two verdict slots of nbk::mass_scan: [0] square launches / whole steps, [1] cross launches
    unsigned int eq_gen = 0;   // generation of the last scan
    int eq_last_slot = 0;      // slot the last scan wrote
    int eq_mode = -1;          // equal-mass path of the symmetric kernels where the device-side scan finds one common mass: -1 launches of
                               // kEqAutoMinBodies bodies or more, 1 of kEqMinBodies or more, 0 never
    // block pairs with the sums added IN PLACE (nbk::force_sym_ticket: no slab workspace)
    int inplace_sums = -1;     // -1 auto: where the slab workspace of the block-pair kernel does not fit the cap (instead of the one-sided kernel);
                               //  0 never; 1 wherever FAST / SYMMETRIC would run unit runs or block pairs
    bool ticket_test_stall = false;  // TEST hook (nbody_ctx_set_inplace_sums(ctx, 2)), one shot: the next in-place launch finds block 0's first ticket held
                                     // and may wait 2 ms — exercises the abort / error path without a defect to provoke it
    unsigned* tickets = nullptr;     // device: one word per (block, lane), zero between launches, + the abort word
    unsigned* terr = nullptr;        // host-mapped: a ticket wait timed out (the launch's sums are not to be trusted)
    unsigned* terr_dev = nullptr;
    void* xslabs = nullptr;    // workspace of nbody_accel_cross (its own, so that a square evaluation issued in parts
    size_t xslab_bytes = 0;    // around cross launches keeps its partial sums)
    bool legacy_eps = false;     // strict kernel evaluates `+ EPS2` as the older snapshot does
    void* legacy_buf = nullptr;  // device staging of the host-pointer adapter
    size_t legacy_bytes = 0;
    bool timing = false;
    std::vector<hipEvent_t> events;  // start/stop pairs around force launches
    size_t events_used = 0;
    // nbody_ctx_timing(.., 2): every timed launch is also bracketed by nbk::clock_begin / clock_end (shader cycles + 100-MHz ticks per XCD)
    bool clock_stamps = false;
    bool clock_pair_open = false;          // clock_begin of the current launch was issued (its clock_end follows)
    nbk::ClockStamp* cscratch = nullptr;   // device: kClockBeginWgs records, rewritten by every clock_begin
    nbk::ClockDelta* cdelta = nullptr;     // host-mapped: one record per timed launch, kClockCapLaunches of them, zero = not written
    nbk::ClockDelta* cdelta_dev = nullptr;
    size_t cdelta_used = 0;                // launches stamped since the last nbody_ctx_clock_read
    // hipGraph of `graph_chunk` (force, integrate) pairs for launch-bound small systems, cached
    // for one set of arguments
    int use_graph = 0;   // -1 auto (n <= kGraphMaxN), 0 never (default: measured neutral, see nbody.h), 1 always
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    struct GraphKey {
        const void *x, *a, *v, *slabs;
        int n, bpl, tile, jsplit, kernel, chunk;
        float dt, eps2;
        hipStream_t stream;
        bool operator==(const GraphKey& o) const
        {
            return x == o.x && a == o.a && v == o.v && slabs == o.slabs && n == o.n && bpl == o.bpl && tile == o.tile &&
                   jsplit == o.jsplit && kernel == o.kernel && chunk == o.chunk && dt == o.dt && eps2 == o.eps2 && stream == o.stream;
        }
    } graph_key{};
};

#pragma GCC visibility push(hidden)
namespace nbi {

struct Shape {
    int bpl, tile, jsplit, blocks_x;
};

// The symmetric kernel's decomposition: blocks of B = 64*waves*bpl bodies, one workgroup per block
// pair (I <= J), one slab per block.
struct SymShape {
    int waves, bpl, block, nb, grid;
};

constexpr int kSymMinAuto = 12288;  // FAST switches to the symmetric kernel from this many bodies
constexpr int kRunsMaxAuto = 160000; // the run-based variant is chosen automatically up to this many bodies
constexpr int kSymMaxSlabs = 2048;
constexpr size_t kSymMaxWorkspace = (size_t)96 << 30;  // one slab per block: beyond 96 GiB of partial sums (or beyond half of the
                                                       // free device memory, nbody_ctx::ws_cap) the one-sided kernel takes over

// The (waves, bodies per lane) request that applies to the fp32 kernels: (4,6) is an fp64-only shape and counts as "auto" here.
inline void fp32_shape_request(const nbody_ctx* c, int* waves, int* bpl)
{
    const bool f64_only = c->sym_waves == 4 && c->sym_bpl == 6;
    *waves = f64_only ? 0 : c->sym_waves;
    *bpl = f64_only ? 0 : c->sym_bpl;
}

// (waves, bodies per lane) instantiated by nbody_step.hip (launch_sym_untimed), largest block first
const int kSymCand[][2] = {{4, 10}, {4, 8}, {2, 10}, {2, 8}, {1, 10}, {1, 8}, {2, 4}, {1, 4}, {1, 2}};
constexpr int kSymCands = 9;
// The run-based variant (nbk::force_sym_run): independent waves, units of one 64-body chunk, L units per worker.
struct RunShape {
    int bpl, nbi, nworkers, max_slabs;
    long nunits;
    nbk::RunLayout layout;
};

// The FUSED small-N step (nbk::step_fused): a wave owns T targets, its lanes split the sources, the sum never leaves the wave and
// the integrate happens in the same launch. One workgroup per CU measured best (tools/balbench.hip, profiles/r03_fused_*.txt):
// T = 2 and N / (2 * 64 * CUs) waves per workgroup up to 16, T = 4 beyond.
struct FusedShape {
    int T, wv, tile, grid;
};

constexpr int kFusedMaxAuto = 8192;    // FAST: the fused step up to this many bodies — one workgroup of up to 16 waves per CU with two
                                       // targets per wave; beyond, four targets per wave lose to the balanced runs (9216: 34.3 vs 29.4 us)

// The BALANCED-run variant (nbk::force_sym_bal): workers of equal step counts, per-chunk inboxes, streaming reducer.
struct BalShape {
    nbk::BalLayout y;
    size_t bytes;   // inbox workspace
};

constexpr int kBalMinAuto = 6144;    // FAST: balanced runs from this many bodies (whole steps up to kFusedMaxAuto go to the fused step first;
                                     // a square block of nbody_accel_range below this: the one-sided kernel) ...
constexpr int kBalMaxAuto = 45056;   // ... up to this many (above: unit runs / block pairs). Measured: profiles/r03_balbench_*.txt
constexpr int kBalWavesPerSimd = 2, kBalWavesPerGroup = 4, kBalReduceWaves = 8;

// equal-mass path: from how many bodies a launch is scanned (nbody_step.hip: eq_scan)
constexpr int kEqMinBodies = 4096;       // mode 1: below this the scan launch costs more than the path saves even over many steps
constexpr int kEqAutoMinBodies = 32768;  // mode -1 (default): the scan is a dependent launch of its own (about 3 us per call). A caller that
                                         // steps one step per call — the reference's own loop, main.cpp:146-156 — pays it every step: 1.5 % of a
                                         // 190-us step at 32768 bodies, but 4-8 % at 9216 ... 16384, whatever the masses are. The automatic
                                         // mode therefore leaves smaller launches alone; mode 1 is for callers who know better.

// One default context per device, created on first use and released at process exit. A recursive
// mutex serialises the entry points that work on them (nbody_simulate, nbody_simulate_host_legacy).
constexpr int kMaxDevices = 64;
extern std::recursive_mutex g_default_mu;   // serialises the entry points that work on the default contexts

// nbody_step.hip
int launch_clock_stamp(nbody_ctx* c, nbk::ClockDelta* d_out);   // d_out == nullptr: nbk::clock_begin, else nbk::clock_end into *d_out
void load_device_code();   // makes the runtime load this library's device code now (milliseconds), not inside a first timed step

// nbody_autotune.hip
bool simulate_knobs_default(const nbody_ctx* c);
int simulate_prepare_locked(nbody_ctx* c, const nbody_float4* d_bodies, int n);

// nbody_plan.hip
Shape resolve_shape(const nbody_ctx* c, int n_targets, int n_sources);
int sym_waves_per_simd(int bpl);
double sym_cost(int W, int bpl, long tasks, double slab_bytes, int num_cu);
bool sym_resolve(const nbody_ctx* c, int n, SymShape* out, bool ignore_cap = false);
bool ticket_wanted(const nbody_ctx* c, int n, SymShape* out);
int ticket_lanes(const nbody_ctx* c, int n, int nb);
bool sym_resolve_cross(const nbody_ctx* c, int ni, int nj, SymShape* out, int* nbj, bool* hopeless = nullptr);
bool run_resolve(const nbody_ctx* c, int n, RunShape* out, double* cost_out);
bool run_wanted(const nbody_ctx* c, int n, RunShape* out);
bool fused_resolve(const nbody_ctx* c, int n, FusedShape* out);
bool fused_wanted(const nbody_ctx* c, int n, FusedShape* out);
bool bal_resolve(const nbody_ctx* c, int n, BalShape* out);
bool bal_wanted(const nbody_ctx* c, int n, BalShape* out);
bool sym_wanted(const nbody_ctx* c, int n, SymShape* out);
bool f64_sym_shape(const nbody_ctx* c, int n, int* W, int* BPL, int* nb_out);

// nbody_context.hip
int check_ctx(const nbody_ctx* c);
int time_mark(nbody_ctx* c);
int ensure_xalt(nbody_ctx* c, int n);
int ensure_fsync(nbody_ctx* c, size_t nwaves);
int ensure_tickets(nbody_ctx* c);
int ticket_error(nbody_ctx* c);   // NBODY_OK, or the error a timed-out ticket wait of an earlier launch left behind (and the tickets reset)
void refresh_ws_cap(nbody_ctx* c);
int grow_workspace(nbody_ctx* c, void** buf, size_t* have, size_t bytes);
int ensure_xslabs(nbody_ctx* c, size_t bytes);
int ensure_slabs(nbody_ctx* c, size_t bytes);
int ensure_inbox(nbody_ctx* c, const BalShape& b);

}  // namespace nbi
#pragma GCC visibility pop

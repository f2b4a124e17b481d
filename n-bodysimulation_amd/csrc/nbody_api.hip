// nbody_api.hip — implementation of include/nbody.h (libnbody_hip.so) for gfx950.
//
// Host launchers for the kernels in nbody_kernels.hip.h plus the small amount of host logic the
// reference keeps in simulate() (TestProject/kernel.cu:628-645): launch-shape selection and
// error reporting. No CPU compute path exists in this file on purpose.
#include "nbody.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "nbody_kernels.hip.h"

static_assert(sizeof(nbody_float4) == sizeof(float4) && alignof(float4) == 16, "float4 layout");
static_assert(sizeof(nbody_double4) == sizeof(double4), "double4 layout");

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

namespace {

#define fail nbody_fail

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(NBODY_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

// Makes `device` current for the scope of an entry point and puts the caller's device back afterwards
// (the reference never calls cudaSetDevice: a caller's current device must survive our calls).
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            changed = (err == hipSuccess);
        }
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define ON_DEVICE(c)                                                                                     \
    DeviceGuard device_guard_((c)->device);                                                              \
    if (device_guard_.err != hipSuccess)                                                                 \
        return fail(NBODY_ERR_HIP, "cannot select device %d: %s", (c)->device, hipGetErrorString(device_guard_.err))

constexpr int kMaxSplit = 64;
constexpr int kGraphMaxN = 16384;  // below this a step is a few tens of microseconds: launch-bound
constexpr int kGraphChunk = 32;    // steps per graph launch

}  // namespace

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
    void* xslabs = nullptr;    // workspace of nbody_accel_cross (its own, so that a square evaluation issued in parts
    size_t xslab_bytes = 0;    // around cross launches keeps its partial sums)
    bool legacy_eps = false;     // strict kernel evaluates `+ EPS2` as the older snapshot does
    void* legacy_buf = nullptr;  // device staging of the host-pointer adapter
    size_t legacy_bytes = 0;
    bool timing = false;
    std::vector<hipEvent_t> events;  // start/stop pairs around force launches
    size_t events_used = 0;
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

namespace {

using P2 = nbk::MathPacked<2>;
using P4 = nbk::MathPacked<4>;
using S1 = nbk::MathScalar<1>;

struct Shape {
    int bpl, tile, jsplit, blocks_x;
};

// Launch shape for (n_targets x n_sources). The reference hard-codes 32 threads/block and a
// 32-body tile (constants.h:11-12); here the block is 256 threads, each lane holds `bpl`
// targets, and the source range is cut into `jsplit` slabs (at most 64) so that the grid has many
// more workgroups than the chip has CUs: more, smaller workgroups smooth the tail of the launch
// (N=1048576 ran 268 ms/step with 4 slabs, 255 ms with 16).
Shape resolve_shape(const nbody_ctx* c, int n_targets, int n_sources)
{
    Shape s{};
    // (targets per lane, tile) candidates, largest first. Measured at N=262144 (bench.py, ms/step):
    // tile 1024: 15.98/15.82/15.76 at 8/16/32 slabs; tile 2048: 15.61/15.48/15.55. Smaller systems
    // take the first candidate whose grid can reach ~8 workgroups per CU (tools/kbench.hip sweeps at
    // N = 4096 ... 32768: the number of workgroups is the lever, a slab of a single tile is fine once
    // there are enough of them; N=8192, the reference's N_BODIES, ends at 1 target per lane, 256-body tile).
    static const int cand[][2] = {{4, 2048}, {4, 1024}, {4, 512}, {2, 512}, {2, 256}, {1, 256}};
    s.bpl = c->bpl;
    s.tile = c->tile;
    if (c->tile == 2048) s.bpl = 4;  // the 2048-body tile is only instantiated for 4 targets per lane
    if (!s.bpl || !s.tile) {
        int pick = 5;
        for (int k = 0; k < 6; ++k) {
            if ((c->bpl && cand[k][0] != c->bpl) || (c->tile && cand[k][1] != c->tile)) continue;
            pick = k;
            const long bx = (n_targets + nbk::kWG * cand[k][0] - 1) / (nbk::kWG * cand[k][0]);
            const long ntile = (n_sources + cand[k][1] - 1) / cand[k][1];
            const long js = ntile < 1 ? 1 : (ntile > kMaxSplit ? kMaxSplit : ntile);
            if (bx * js >= 8L * c->num_cu) break;
        }
        if (!s.bpl) s.bpl = cand[pick][0];
        if (!s.tile) s.tile = cand[pick][1];
        if (s.bpl != 4 && s.tile == 2048) s.tile = 1024;
    }
    s.blocks_x = (n_targets + nbk::kWG * s.bpl - 1) / (nbk::kWG * s.bpl);
    if (c->kernel == NBODY_KERNEL_STRICT) {
        s.bpl = 1;
        s.tile = 1024;
        s.jsplit = 1;
        s.blocks_x = (n_targets + nbk::kWG - 1) / nbk::kWG;
        return s;
    }
    if (c->jsplit) {
        s.jsplit = c->jsplit;
    } else {
        // ~16 workgroups per CU is enough when the number of target workgroups is a multiple of 8
        // (N=262144: 15.5 ms/step at 16, 32 or 64 slabs). Shapes with an odd count are slower and want
        // the finest split (92672 x 648704: 15.2 ms at 16 slabs, 14.2 ms at 64). Each slab costs
        // 16 B/body of HBM write + read, so no more of them than needed.
        const int want = (s.blocks_x % 8 == 0 ? 16 : 64) * c->num_cu;
        int js = 1;
        while (s.blocks_x * js < want && js < kMaxSplit) js *= 2;
        // never more slabs than tiles; small systems stop at 32 (every slab is one more 16 B/body read
        // in the integrate, which is no longer negligible next to a sub-100-us force kernel)
        const int ntile = (n_sources + s.tile - 1) / s.tile;
        if (js > ntile) js = ntile > 0 ? ntile : 1;
        if (n_targets < 65536 && js > 32) js = 32;
        s.jsplit = js;
    }
    if (s.jsplit < 1) s.jsplit = 1;
    if (s.jsplit > kMaxSplit) s.jsplit = kMaxSplit;
    return s;
}

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

// (waves, bodies per lane) instantiated below, largest block first
const int kSymCand[][2] = {{4, 10}, {4, 8}, {2, 10}, {2, 8}, {1, 10}, {1, 8}, {2, 4}, {1, 4}, {1, 2}};
constexpr int kSymCands = 9;

// Estimated time (shader cycles) of one launch of `tasks` equal block-pair tasks of shape (W waves, bpl bodies per lane) plus
// the cost of summing `slab_bytes` of partial sums afterwards. The kernel is VALU-bound with two or more waves on a SIMD, so a
// SIMD's time is the number of wave-tasks it hosts times the time of one alone: full rounds put `wps` waves on every SIMD
// (wps from the kernel's VGPR allocation), the last partial round ceil(rest * W / SIMDs). One wave-task = B steps of
// (41.33 * bpl + 22.6) cycles — 8 bpl/2 packed ops at 4.15, bpl v_rsq_f32 at 8.13, 10 per-step instructions at 2.26
// (tools/valu_mb.hip). Checked against tools/smalln_probe.py sweeps from 32768 to 1048576 bodies
// (profiles/r02_shape_probe_{mid,large}.jsonl): it ranks the shapes as measured at every size.
// Waves per SIMD the register allocation of force_sym<SymPacked<bpl>> / force_sym_run allows. This is a COMPILER OUTPUT
// (196 / 164 / 92 / 60 VGPRs with ROCm 7.2) that the cost estimates below rely on: tests/test_build_resources.py compiles
// this file with -Rpass-analysis=kernel-resource-usage and checks every shipped instantiation against this table.
int sym_waves_per_simd(int bpl) { return bpl >= 10 ? 2 : bpl >= 8 ? 3 : bpl >= 4 ? 5 : 8; }

double sym_cost(int W, int bpl, long tasks, double slab_bytes, int num_cu)
{
    const int wps = sym_waves_per_simd(bpl);                                // waves per SIMD the VGPR count allows
    const long simds = 4L * num_cu;
    const long slots = simds * wps / W;                                     // resident workgroups
    const long full = tasks / slots, rest = tasks - full * slots;
    const double deep = (double)full * wps + (double)((rest * W + simds - 1) / simds);
    const double step = 41.33 * bpl + 22.6;
    const double B = 64.0 * W * bpl;
    return B * step * deep + slab_bytes / 4.7e12 * 2.26e9;                   // slab sum at 4.7 TB/s, 2.26 GHz
}

bool sym_resolve(const nbody_ctx* c, int n, SymShape* out)
{
    int pick = -1;
    double best = 0.0;
    int rw, rb;
    fp32_shape_request(c, &rw, &rb);
    for (int k = 0; k < kSymCands; ++k) {
        if ((rw && kSymCand[k][0] != rw) || (rb && kSymCand[k][1] != rb)) continue;
        const long B = 64L * kSymCand[k][0] * kSymCand[k][1];
        const long nb = (n + B - 1) / B;
        if (nb < 2 && pick >= 0) continue;
        if (nb > kSymMaxSlabs || (size_t)nb * (size_t)n * sizeof(float4) > c->ws_cap) continue;
        const double cost = sym_cost(kSymCand[k][0], kSymCand[k][1], nb * (nb + 1) / 2, (double)nb * n * sizeof(float4), c->num_cu);
        if (pick < 0 || cost < best) { pick = k; best = cost; }
    }
    if (pick < 0) return false;
    SymShape y{};
    y.waves = kSymCand[pick][0];
    y.bpl = kSymCand[pick][1];
    y.block = 64 * y.waves * y.bpl;
    y.nb = (n + y.block - 1) / y.block;
    y.grid = y.nb * (y.nb - 1) / 2 + y.nb;
    if (y.nb < 2 || y.nb > kSymMaxSlabs) return false;
    if ((size_t)y.nb * (size_t)n * sizeof(float4) > c->ws_cap) return false;
    *out = y;
    return true;
}

// one range of n bodies starting at absolute index i0 against itself
void sym_square_params(nbk::SymParams* sp, const float4* x, int i0, int n, const SymShape& y, float4* slabs, float eps2)
{
    *sp = nbk::SymParams{};
    sp->x = x;
    sp->slabs_i = slabs;
    sp->slabs_j = slabs;
    sp->ni = n; sp->nj = n;
    sp->i0 = i0; sp->j0 = i0;
    sp->wrap = 0;
    sp->nbi = y.nb; sp->nbj = y.nb;
    sp->stride_i = n; sp->stride_j = n;
    sp->rect = 0;
    sp->eps2 = eps2;
}

// Block shape for the symmetric evaluation of TWO disjoint ranges (ni x nj bodies): the cheapest by the same estimate
// (padding of both sides to whole blocks included through the task count).
// `hopeless` (optional): set when no candidate could ever apply to ni targets however short the source run is made (a shape
// request that matches nothing built, or too many target blocks) — as opposed to a workspace over the cap, which fewer sources cure.
bool sym_resolve_cross(const nbody_ctx* c, int ni, int nj, SymShape* out, int* nbj, bool* hopeless = nullptr)
{
    int pick = -1;
    double best = 0.0;
    int rw, rb;
    fp32_shape_request(c, &rw, &rb);
    if (hopeless) *hopeless = true;
    for (int k = 0; k < kSymCands; ++k) {
        if ((rw && kSymCand[k][0] != rw) || (rb && kSymCand[k][1] != rb)) continue;
        const long B = 64L * kSymCand[k][0] * kSymCand[k][1];
        const long bi = (ni + B - 1) / B, bj = (nj + B - 1) / B;
        if (bi > kSymMaxSlabs) continue;
        if (hopeless) *hopeless = false;       // this shape takes the targets: a shorter source run may fit
        if (bj > 4 * kSymMaxSlabs) continue;
        if (((size_t)bj * (size_t)ni + (size_t)bi * (size_t)nj) * sizeof(float4) > c->ws_cap) continue;
        const double cost = sym_cost(kSymCand[k][0], kSymCand[k][1], bi * bj, ((double)bj * ni + (double)bi * nj) * sizeof(float4), c->num_cu);
        if (pick < 0 || cost < best) { pick = k; best = cost; }
    }
    if (pick < 0) return false;
    SymShape y{};
    y.waves = kSymCand[pick][0];
    y.bpl = kSymCand[pick][1];
    y.block = 64 * y.waves * y.bpl;
    y.nb = (ni + y.block - 1) / y.block;
    const int bj = (nj + y.block - 1) / y.block;
    y.grid = y.nb * bj;
    *out = y;
    *nbj = bj;
    return true;
}

// The run-based variant (nbk::force_sym_run): independent waves, units of one 64-body chunk, L units per worker.
struct RunShape {
    int bpl, nbi, nworkers, max_slabs;
    long nunits;
    nbk::RunLayout layout;
};

bool run_resolve(const nbody_ctx* c, int n, RunShape* out, double* cost_out)
{
    if (n < 128) return false;
    int pick = -1;
    double best = 0.0;
    RunShape cand[2];
    static const int bpls[2] = {10, 8};
    int rw, rb;
    fp32_shape_request(c, &rw, &rb);
    (void)rw;
    for (int k = 0; k < 2; ++k) {
        const int bpl = bpls[k];
        if (rb && rb != bpl) continue;
        // measured: 10 bodies per lane is the better run shape below 49152 bodies, 8 (three waves per SIMD) from there
        if (!rb && bpl != (n < 49152 ? 10 : 8)) continue;
        RunShape y{};
        y.bpl = bpl;
        y.layout.bi = 64 * bpl;
        y.layout.bpl = bpl;
        y.layout.ncht = (n + 63) / 64;
        y.nbi = (n + y.layout.bi - 1) / y.layout.bi;
        y.layout.L = 1;
        y.nunits = nbk::run_prefix(y.nbi, y.layout);
        const int wps = sym_waves_per_simd(bpl);
        const long simds = 4L * c->num_cu, slots = simds * wps;
        long L = (y.nunits + slots - 1) / slots;   // every worker resident at once
        if (L < 1) L = 1;
        y.layout.L = (int)L;
        y.nworkers = (int)((y.nunits + L - 1) / L);
        y.max_slabs = y.nbi + (int)((y.layout.ncht + L - 1) / L) + 2;
        if ((size_t)y.max_slabs * (size_t)n * sizeof(float4) > c->ws_cap) continue;
        const double unit = 64.0 * (41.33 * bpl + 22.6);
        const double deep = (double)((y.nworkers + simds - 1) / simds) * (double)L;          // units on the fullest SIMD
        const double slabs_avg = 0.5 * y.nbi + 0.5 * (double)y.layout.ncht / (double)L + 1.0;
        const double cost = deep * unit + 6000.0 * (double)((y.nworkers + simds - 1) / simds)  // + per-worker prologue
                            + slabs_avg * n * sizeof(float4) / 4.7e12 * 2.26e9;
        cand[k] = y;
        if (pick < 0 || cost < best) { pick = k; best = cost; }
    }
    if (pick < 0) return false;
    *out = cand[pick];
    if (cost_out) *cost_out = best;
    return true;
}

// Does a square problem of n bodies go to the run-based variant rather than to block pairs?
bool run_wanted(const nbody_ctx* c, int n, RunShape* out)
{
    {
        int rw, rb;
        fp32_shape_request(c, &rw, &rb);
        if (c->sym_runs == 0 || c->sym_runs == 2 || rw != 0) return false;
    }
    if (!(c->kernel == NBODY_KERNEL_SYMMETRIC || (c->kernel == NBODY_KERNEL_FAST && n >= kSymMinAuto))) return false;
    double rc = 0.0;
    if (!run_resolve(c, n, out, &rc)) return false;
    if (c->sym_runs == 1) return true;
    // Measured (tools/smalln_probe.py, profiles/r02_runs_probe.jsonl, same box): runs beat the best block shape by 3-9 %
    // from 16384 to 131072 bodies and lose 1 % at 262144 (twice the slabs, single waves); below 12288 the one-sided
    // kernel wins. The cost estimates of the two decompositions agree with that ordering only inside this range.
    if (n > kRunsMaxAuto) return false;
    SymShape y{};         // the block-pair choice it competes with
    if (!sym_resolve(c, n, &y)) return true;
    const double bc = sym_cost(y.waves, y.bpl, (long)y.nb * (y.nb + 1) / 2, (double)y.nb * n * sizeof(float4), c->num_cu);
    return rc < 1.03 * bc;
}

// The FUSED small-N step (nbk::step_fused): a wave owns T targets, its lanes split the sources, the sum never leaves the wave and
// the integrate happens in the same launch. One workgroup per CU measured best (tools/balbench.hip, profiles/r03_fused_*.txt):
// T = 2 and N / (2 * 64 * CUs) waves per workgroup up to 16, T = 4 beyond.
struct FusedShape {
    int T, wv, tile, grid;
};

constexpr int kFusedMaxAuto = 8192;    // FAST: the fused step up to this many bodies — one workgroup of up to 16 waves per CU with two
                                       // targets per wave; beyond, four targets per wave lose to the balanced runs (9216: 34.3 vs 29.4 us)

bool fused_resolve(const nbody_ctx* c, int n, FusedShape* out)
{
    if (n < 1) return false;
    FusedShape f{};
    f.T = 2;
    long waves = ((long)n + f.T - 1) / f.T;
    long per = (waves + c->num_cu - 1) / c->num_cu;
    if (per > 16) {
        f.T = 4;
        waves = ((long)n + f.T - 1) / f.T;
        per = (waves + c->num_cu - 1) / c->num_cu;
    }
    int wv = (int)((per + 1) / 2 * 2);   // even, 2 .. 16 (built: 2, 4, 6, 8, 10, 12, 14, 16)
    if (wv < 2) wv = 2;
    if (wv > 16) wv = 16;
    f.wv = wv;
    static const int lpt[9] = {0, 16, 8, 6, 4, 4, 3, 3, 2};   // loads per thread per tile for wv = 2k: tiles of 2048 ... 2688 bodies
    f.tile = 64 * wv * lpt[wv / 2];
    f.grid = (int)((waves + wv - 1) / wv);
    *out = f;
    return true;
}

bool fused_wanted(const nbody_ctx* c, int n, FusedShape* out)
{
    if (c->kernel != NBODY_KERNEL_FAST || c->fused == 0) return false;
    if (c->fused < 0 && (n > kFusedMaxAuto || c->sym_runs == 2)) return false;
    int rw, rb;
    fp32_shape_request(c, &rw, &rb);
    if (c->fused < 0 && (rw || rb || c->tile || c->bpl || c->jsplit)) return false;   // an explicit shape request addresses the other kernels
    return fused_resolve(c, n, out);
}

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

template <int T, bool INPLACE>
int launch_fused_t(const FusedShape& f, const nbk::FusedParams& p, hipStream_t st)
{
    // <targets per wave, waves per workgroup, tile, unroll of the per-lane source loop>: the unroll measured better per shape
    // (profiles/r03_smalln_probe.jsonl; N = 8192: 21.07 us at 4, 20.36 at 8; N = 4096: 7.99 at 4, 8.72 at 8)
    switch (f.wv) {
        case 2: nbk::step_fused<T, 2, 2048, 8, 1, INPLACE><<<f.grid, 128, 0, st>>>(p); break;
        case 4: nbk::step_fused<T, 4, 2048, 8, 1, INPLACE><<<f.grid, 256, 0, st>>>(p); break;
        case 6: nbk::step_fused<T, 6, 2304, 4, 1, INPLACE><<<f.grid, 384, 0, st>>>(p); break;
        case 8: nbk::step_fused<T, 8, 2048, 4, 1, INPLACE><<<f.grid, 512, 0, st>>>(p); break;
        case 10: nbk::step_fused<T, 10, 2560, 8, 1, INPLACE><<<f.grid, 640, 0, st>>>(p); break;
        case 12: nbk::step_fused<T, 12, 2304, 4, 1, INPLACE><<<f.grid, 768, 0, st>>>(p); break;
        case 14: nbk::step_fused<T, 14, 2688, 8, 1, INPLACE><<<f.grid, 896, 0, st>>>(p); break;
        case 16: nbk::step_fused<T, 16, 2048, 8, 1, INPLACE><<<f.grid, 1024, 0, st>>>(p); break;
        default: return 1;
    }
    return 0;
}

// The BALANCED-run variant (nbk::force_sym_bal): workers of equal step counts, per-chunk inboxes, streaming reducer.
struct BalShape {
    nbk::BalLayout y;
    size_t bytes;   // inbox workspace
};

constexpr int kBalMinAuto = 6144;    // FAST: balanced runs from this many bodies (whole steps up to kFusedMaxAuto go to the fused step first;
                                     // a square block of nbody_accel_range below this: the one-sided kernel) ...
constexpr int kBalMaxAuto = 45056;   // ... up to this many (above: unit runs / block pairs). Measured: profiles/r03_balbench_*.txt
constexpr int kBalWavesPerSimd = 2, kBalWavesPerGroup = 4, kBalReduceWaves = 8;

bool bal_resolve(const nbody_ctx* c, int n, BalShape* out)
{
    int rw, rb;
    fp32_shape_request(c, &rw, &rb);
    if (rw != 0) return false;                       // a waves-per-workgroup request means block pairs
    int bpl = rb;
    if (bpl == 0) bpl = n < 10240 ? 4 : (n < 20480 ? 8 : 10);   // measured best per size (tools/balbench.hip)
    if (bpl != 2 && bpl != 4 && bpl != 8 && bpl != 10) return false;
    BalShape b{};
    if (!nbk::bal_plan(n, bpl, 4 * c->num_cu * kBalWavesPerSimd, kBalWavesPerGroup, &b.y)) return false;
    if (b.y.pmax > 5) return false;                  // the reducer holds at most five pieces of a unit
    b.bytes = (size_t)b.y.ncht * (size_t)b.y.smax * 64 * sizeof(float4);
    if (b.bytes > c->ws_cap) return false;
    *out = b;
    return true;
}

// Does a square problem of n bodies go to the balanced-run variant?
bool bal_wanted(const nbody_ctx* c, int n, BalShape* out)
{
    if (c->sym_runs == 0 || c->sym_runs == 1) return false;
    if (c->sym_runs == 2) return (c->kernel == NBODY_KERNEL_FAST || c->kernel == NBODY_KERNEL_SYMMETRIC) && bal_resolve(c, n, out);
    if (c->kernel != NBODY_KERNEL_FAST || n < kBalMinAuto || n > kBalMaxAuto) return false;
    return bal_resolve(c, n, out);
}

// Does a square problem of n bodies (targets == sources) go to the symmetric kernel?
bool sym_wanted(const nbody_ctx* c, int n, SymShape* out)
{
    if (c->kernel == NBODY_KERNEL_SYMMETRIC) return sym_resolve(c, n, out);
    if (c->kernel == NBODY_KERNEL_FAST && n >= kSymMinAuto) return sym_resolve(c, n, out);
    return false;
}

// The fp64 step's symmetric shape: the rotation kernel in double (FAST from kSymMinAuto bodies, or SYMMETRIC). Shapes (waves,
// bodies per lane): (4,6) measured best at N=262144 (26.4 ms/step; (4,8) 27.1 with 256 VGPR + 23 AGPR and one wave per
// SIMD, (4,4) 27.2, (2,6) 26.8); (4,8) stays selectable through nbody_ctx_set_symmetric_shape. false = one-sided kernel.
bool f64_sym_shape(const nbody_ctx* c, int n, int* W, int* BPL, int* nb_out)
{
    static const int cand[][2] = {{4, 6}, {2, 4}, {1, 2}, {4, 8}};
    if (!(c->kernel == NBODY_KERNEL_SYMMETRIC || (c->kernel == NBODY_KERNEL_FAST && n >= kSymMinAuto))) return false;
    int pick = -1;
    for (int k = 0; k < 4; ++k) {
        if (k == 3 && !(c->sym_waves == 4 && c->sym_bpl == 8)) continue;  // only on request
        if ((c->sym_waves && cand[k][0] != c->sym_waves) || (c->sym_bpl && cand[k][1] != c->sym_bpl)) continue;
        pick = k;
        if ((long)n >= 128L * 64 * cand[k][0] * cand[k][1]) break;
    }
    if (pick < 0) return false;
    const int B = 64 * cand[pick][0] * cand[pick][1];
    const int nb = (n + B - 1) / B;
    if (nb < 2 || nb > kSymMaxSlabs) return false;
    if ((size_t)nb * (size_t)n * sizeof(double4) > c->ws_cap) return false;
    *W = cand[pick][0];
    *BPL = cand[pick][1];
    *nb_out = nb;
    return true;
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

// Equal-mass path: scans the bodies of the coming launch(es) on the stream (x[i0 .. i0+ni) and, when nj > 0, the run of nj bodies
// from j0, wrapping at `wrap`) and hands out the verdict slot and this scan's generation. No host round trip: the force kernel
// reads the verdict itself. *q stays nullptr when the path is switched off (or the verdict slots cannot be allocated).
constexpr int kEqMinBodies = 4096;       // mode 1: below this the scan launch costs more than the path saves even over many steps
constexpr int kEqAutoMinBodies = 32768;  // mode -1 (default): the scan is a dependent launch of its own (about 3 us per call). A caller that
                                         // steps one step per call — the reference's own loop, main.cpp:146-156 — pays it every step: 1.5 % of a
                                         // 190-us step at 32768 bodies, but 4-8 % at 9216 ... 16384, whatever the masses are. The automatic
                                         // mode therefore leaves smaller launches alone; mode 1 is for callers who know better.

template <class V4>
int eq_scan(nbody_ctx* c, int slot, const V4* x, int i0, int ni, int j0, int nj, int wrap, const nbk::MassInfo** q, unsigned int* gen)
{
    *q = nullptr;
    *gen = 0;
    if (c->eq_mode == 0 || ni <= 0 || ni + nj < (c->eq_mode == 1 ? kEqMinBodies : kEqAutoMinBodies)) return NBODY_OK;
    if (!c->eqm) {
        if (hipMalloc(reinterpret_cast<void**>(&c->eqm), 2 * sizeof(nbk::MassInfo)) != hipSuccess) {
            (void)hipGetLastError();
            c->eqm = nullptr;
            return NBODY_OK;   // not an error of the step: the general path runs
        }
        HIP_TRY(hipMemset(c->eqm, 0, 2 * sizeof(nbk::MassInfo)));
    }
    if (++c->eq_gen == 0) ++c->eq_gen;   // 0 is what a fresh slot holds
    nbk::MassScanParamsT<V4> mp{};
    mp.x = x;
    mp.i0 = i0; mp.ni = ni; mp.j0 = j0; mp.nj = nj; mp.wrap = wrap;
    mp.out = c->eqm + slot;
    mp.gen = c->eq_gen;
    int blocks = (ni + nj + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    nbk::mass_scan<V4><<<blocks, 256, 0, c->stream>>>(mp);
    HIP_TRY(hipGetLastError());
    c->eq_last_slot = slot;
    *q = c->eqm + slot;
    *gen = c->eq_gen;
    return NBODY_OK;
}

template <class M, int TILE>
void launch_lds(const nbk::ForceParams& p, dim3 grid, hipStream_t st)
{
    nbk::force_lds<M, TILE, 8, 1><<<grid, nbk::kWG, 0, st>>>(p);
}

int time_mark(nbody_ctx* c)
{
    if (!c->timing) return NBODY_OK;
    if (c->events_used == c->events.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->events.push_back(e);
    }
    HIP_TRY(hipEventRecord(c->events[c->events_used++], c->stream));
    return NBODY_OK;
}

int launch_force_untimed(nbody_ctx* c, const Shape& s, const nbk::ForceParams& p);

int launch_force(nbody_ctx* c, const Shape& s, const nbk::ForceParams& p)
{
    if (int rc = time_mark(c)) return rc;
    if (int rc = launch_force_untimed(c, s, p)) return rc;
    return time_mark(c);
}

int launch_force_untimed(nbody_ctx* c, const Shape& s, const nbk::ForceParams& p)
{
    if (p.i1 <= p.i0) return NBODY_OK;
    if (c->kernel == NBODY_KERNEL_STRICT) {
        if (c->legacy_eps) nbk::force_strict<1024, true><<<dim3(s.blocks_x), nbk::kWG, 0, c->stream>>>(p);
        else nbk::force_strict<1024, false><<<dim3(s.blocks_x), nbk::kWG, 0, c->stream>>>(p);
        HIP_TRY(hipGetLastError());
        return NBODY_OK;
    }
    const dim3 grid(s.blocks_x, s.jsplit);
    const int key = s.bpl * 10000 + s.tile;
    switch (key) {
        case 1 * 10000 + 256: nbk::force_lds<S1, 256, 8, 1><<<grid, nbk::kWG, 0, c->stream>>>(p); break;
        case 1 * 10000 + 512: nbk::force_lds<S1, 512, 8, 1><<<grid, nbk::kWG, 0, c->stream>>>(p); break;
        case 1 * 10000 + 1024: nbk::force_lds<S1, 1024, 8, 1><<<grid, nbk::kWG, 0, c->stream>>>(p); break;
        case 2 * 10000 + 256: launch_lds<P2, 256>(p, grid, c->stream); break;
        case 2 * 10000 + 512: launch_lds<P2, 512>(p, grid, c->stream); break;
        case 2 * 10000 + 1024: launch_lds<P2, 1024>(p, grid, c->stream); break;
        case 4 * 10000 + 256: launch_lds<P4, 256>(p, grid, c->stream); break;
        case 4 * 10000 + 512: launch_lds<P4, 512>(p, grid, c->stream); break;
        case 4 * 10000 + 1024: launch_lds<P4, 1024>(p, grid, c->stream); break;
        case 4 * 10000 + 2048: launch_lds<P4, 2048>(p, grid, c->stream); break;
        default:
            return fail(NBODY_ERR_CONFIG, "no force kernel for bodies_per_lane=%d tile=%d", s.bpl, s.tile);
    }
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

// launches tasks [p.task0, p.task0 + ntasks) of the shape's task list (ntasks < 0: all of them from p.task0)
int launch_sym_untimed(nbody_ctx* c, const SymShape& y0, const nbk::SymParams& p, int ntasks = -1)
{
    using nbk::SymPacked;
    SymShape y = y0;
    y.grid = ntasks >= 0 ? ntasks : y0.grid - p.task0;
    if (y.grid <= 0) return NBODY_OK;
    const int key = y.waves * 100 + y.bpl;
    // one range against itself, no wrap-around: the square-only build of the same kernel (the default large-N shape; measured
    // 1.5-2.7 % faster than the general one at N = 262144, profiles/r03_symbench_rows_262144.txt). Only for 10 bodies per lane: with
    // the equal-mass path compiled in, the square build for 8 needs 178 VGPRs (two waves per SIMD instead of three); the general
    // kernel keeps 164.
    const bool square = !p.rect && !p.wrap && p.i0 == p.j0 && p.ni == p.nj && p.slabs_i == p.slabs_j;
    if (square && key == 410) {
        nbk::force_sym_square<SymPacked<10>, 4><<<y.grid, 256, 0, c->stream>>>(p);
        HIP_TRY(hipGetLastError());
        return NBODY_OK;
    }
    // two disjoint ranges (nbody_accel_cross): the rectangular-only build, 1-3 % ahead of the general one on the equal-mass path
    // (163 VGPRs, three waves per SIMD; profiles/r03_symbench_rect_*.txt), equal on the general path
    if (p.rect && key == 410) {
        nbk::force_sym_rect<SymPacked<10>, 4><<<y.grid, 256, 0, c->stream>>>(p);
        HIP_TRY(hipGetLastError());
        return NBODY_OK;
    }
    switch (key) {
        case 410: nbk::force_sym<SymPacked<10>, 4><<<y.grid, 256, 0, c->stream>>>(p); break;
        case 408: nbk::force_sym<SymPacked<8>, 4><<<y.grid, 256, 0, c->stream>>>(p); break;
        case 210: nbk::force_sym<SymPacked<10>, 2><<<y.grid, 128, 0, c->stream>>>(p); break;
        case 208: nbk::force_sym<SymPacked<8>, 2><<<y.grid, 128, 0, c->stream>>>(p); break;
        case 110: nbk::force_sym<SymPacked<10>, 1><<<y.grid, 64, 0, c->stream>>>(p); break;
        case 108: nbk::force_sym<SymPacked<8>, 1><<<y.grid, 64, 0, c->stream>>>(p); break;
        case 204: nbk::force_sym<SymPacked<4>, 2><<<y.grid, 128, 0, c->stream>>>(p); break;
        case 104: nbk::force_sym<SymPacked<4>, 1><<<y.grid, 64, 0, c->stream>>>(p); break;
        case 102: nbk::force_sym<SymPacked<2>, 1><<<y.grid, 64, 0, c->stream>>>(p); break;
        default: return fail(NBODY_ERR_CONFIG, "no symmetric kernel for waves=%d bodies_per_lane=%d", y.waves, y.bpl);
    }
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

int launch_run(nbody_ctx* c, const RunShape& y, const nbk::RunParams& p)
{
    if (int rc = time_mark(c)) return rc;
    if (y.bpl == 10) nbk::force_sym_run<nbk::SymPacked<10>><<<y.nworkers, 64, 0, c->stream>>>(p);
    else nbk::force_sym_run<nbk::SymPacked<8>><<<y.nworkers, 64, 0, c->stream>>>(p);
    HIP_TRY(hipGetLastError());
    return time_mark(c);
}

template <class M>
void launch_bal_t(const nbk::BalParams& p, hipStream_t st)
{
    const int groups = (p.y.nworkers + kBalWavesPerGroup - 1) / kBalWavesPerGroup;
    nbk::force_sym_bal<M, kBalWavesPerGroup><<<groups, 64 * kBalWavesPerGroup, 0, st>>>(p);
}

int launch_bal(nbody_ctx* c, const nbk::BalParams& p, bool timed)
{
    if (timed) if (int rc = time_mark(c)) return rc;
    switch (p.y.bpl) {
        case 2: launch_bal_t<nbk::SymPacked<2>>(p, c->stream); break;
        case 4: launch_bal_t<nbk::SymPacked<4>>(p, c->stream); break;
        case 8: launch_bal_t<nbk::SymPacked<8>>(p, c->stream); break;
        case 10: launch_bal_t<nbk::SymPacked<10>>(p, c->stream); break;
        default: return fail(NBODY_ERR_CONFIG, "no balanced-run kernel for bodies_per_lane=%d", p.y.bpl);
    }
    HIP_TRY(hipGetLastError());
    if (timed) return time_mark(c);
    return NBODY_OK;
}

int launch_bal_reduce(nbody_ctx* c, const nbk::BalReduceParams& r)
{
    nbk::bal_reduce<kBalReduceWaves><<<r.y.ncht, 64 * kBalReduceWaves, 0, c->stream>>>(r);
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

int launch_fused(nbody_ctx* c, const FusedShape& f, const nbk::FusedParams& p, bool timed)
{
    if (timed) if (int rc = time_mark(c)) return rc;
    const int bad = p.sync ? (f.T == 2 ? launch_fused_t<2, true>(f, p, c->stream) : launch_fused_t<4, true>(f, p, c->stream))
                           : (f.T == 2 ? launch_fused_t<2, false>(f, p, c->stream) : launch_fused_t<4, false>(f, p, c->stream));
    if (bad) return fail(NBODY_ERR_CONFIG, "no fused step kernel for T=%d waves=%d", f.T, f.wv);
    HIP_TRY(hipGetLastError());
    if (timed) return time_mark(c);
    return NBODY_OK;
}

void run_params(nbk::RunParams* rp, const float4* x, int n, const RunShape& y, float4* slabs, float eps2)
{
    *rp = nbk::RunParams{};
    rp->x = x;
    rp->slabs = slabs;
    rp->n = n;
    rp->stride = n;
    rp->nbi = y.nbi;
    rp->nunits = y.nunits;
    rp->r = y.layout;
    rp->eps2 = eps2;
}

int launch_sym(nbody_ctx* c, const SymShape& y, const nbk::SymParams& p, int ntasks = -1)
{
    if (int rc = time_mark(c)) return rc;
    if (int rc = launch_sym_untimed(c, y, p, ntasks)) return rc;
    return time_mark(c);
}

int check_ctx(const nbody_ctx* c)
{
    if (!c) return fail(NBODY_ERR_INVALID, "null context");
    return NBODY_OK;
}

// One default context per device, created on first use and released at process exit. A recursive
// mutex serialises the entry points that work on them (nbody_simulate, nbody_simulate_host_legacy).
constexpr int kMaxDevices = 64;
std::recursive_mutex g_default_mu;
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
    return "nbody_hip 0.4 gfx950 fast=symmetric-dpp(w4,bpl10)+equal-mass-path|symmetric-balanced-runs(8k-45k)|fused-step(<=8k;simulate:in-place+host-word)|onesided-lds-packed(bpl4,tile2048,u8) strict=ieee-seq f64=symmetric-dpp(w4,bpl6)+equal-mass-path|lds";
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
    DeviceGuard guard(device);
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
    DeviceGuard guard(c->device);
    if (c->slabs) (void)hipFree(c->slabs);
    if (c->xslabs) (void)hipFree(c->xslabs);
    if (c->xalt) (void)hipFree(c->xalt);
    if (c->fsync) (void)hipFree(c->fsync);
    if (c->fhost) (void)hipHostFree(c->fhost);
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
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&nbk::copy_bodies)) != hipSuccess) (void)hipGetLastError();
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
int nbody_plan(int n_targets, int n_sources, int kernel, int tile, int bodies_per_lane, int jsplit, int num_cu,
               int* out_bodies_per_lane, int* out_tile, int* out_jsplit, int* out_blocks_x)
{
    if (n_targets < 0 || n_sources < 0 || num_cu <= 0) return fail(NBODY_ERR_INVALID, "bad plan arguments");
    nbody_ctx tmp;
    tmp.kernel = kernel;
    tmp.tile = tile;
    tmp.bpl = bodies_per_lane;
    tmp.jsplit = jsplit;
    tmp.num_cu = num_cu;
    const Shape s = resolve_shape(&tmp, n_targets, n_sources);
    if (out_bodies_per_lane) *out_bodies_per_lane = s.bpl;
    if (out_tile) *out_tile = s.tile;
    if (out_jsplit) *out_jsplit = s.jsplit;
    if (out_blocks_x) *out_blocks_x = s.blocks_x;
    return NBODY_OK;
}

// Device-free view of the symmetric kernel's shape choice.
int nbody_plan_symmetric(int n, int num_cu, int waves, int bodies_per_lane, int* out_waves, int* out_bodies_per_lane,
                         int* out_blocks, int* out_workgroups)
{
    if (n < 0 || num_cu <= 0) return fail(NBODY_ERR_INVALID, "bad plan arguments");
    nbody_ctx tmp;
    tmp.kernel = NBODY_KERNEL_SYMMETRIC;
    tmp.sym_waves = waves;
    tmp.sym_bpl = bodies_per_lane;
    tmp.num_cu = num_cu;
    SymShape y{};
    if (!sym_resolve(&tmp, n, &y)) return fail(NBODY_ERR_CONFIG, "no symmetric shape for %d bodies (waves=%d, bodies_per_lane=%d)", n, waves, bodies_per_lane);
    if (out_waves) *out_waves = y.waves;
    if (out_bodies_per_lane) *out_bodies_per_lane = y.bpl;
    if (out_blocks) *out_blocks = y.nb;
    if (out_workgroups) *out_workgroups = y.grid;
    return NBODY_OK;
}

// Device-free view of the fused step's launch shape (host tests).
int nbody_plan_fused(int n, int num_cu, int* out_targets_per_wave, int* out_waves, int* out_tile, int* out_workgroups)
{
    if (n < 1 || num_cu <= 0) return fail(NBODY_ERR_INVALID, "bad plan arguments");
    nbody_ctx tmp;
    tmp.num_cu = num_cu;
    FusedShape f{};
    if (!fused_resolve(&tmp, n, &f)) return fail(NBODY_ERR_CONFIG, "no fused shape for %d bodies", n);
    if (out_targets_per_wave) *out_targets_per_wave = f.T;
    if (out_waves) *out_waves = f.wv;
    if (out_tile) *out_tile = f.tile;
    if (out_workgroups) *out_workgroups = f.grid;
    return NBODY_OK;
}

int nbody_plan_symmetric_occupancy(int bodies_per_lane)
{
    if (bodies_per_lane < 2 || bodies_per_lane > 16) return 0;
    return sym_waves_per_simd(bodies_per_lane);
}

int nbody_ctx_launch_info(nbody_ctx* c, int n_targets, int n_sources, int* jsplit, int* blocks, int* lds_bytes)
{
    if (int rc = check_ctx(c)) return rc;
    if (n_targets < 0 || n_sources < 0) return fail(NBODY_ERR_INVALID, "negative size");
    SymShape y{};
    RunShape ry{};
    BalShape by{};
    if (n_targets == n_sources && bal_wanted(c, n_targets, &by)) {   // (nbody_accel_range on a square block; a whole step of a small system is fused: nbody_ctx_step_info)
        if (jsplit) *jsplit = by.y.smax;
        if (blocks) *blocks = (by.y.nworkers + kBalWavesPerGroup - 1) / kBalWavesPerGroup;
        if (lds_bytes) *lds_bytes = kBalWavesPerGroup * 64 * by.y.bpl * (int)sizeof(float4);
        return NBODY_OK;
    }
    if (n_targets == n_sources && run_wanted(c, n_targets, &ry)) {
        if (jsplit) *jsplit = ry.max_slabs;
        if (blocks) *blocks = ry.nworkers;
        if (lds_bytes) *lds_bytes = 0;
        return NBODY_OK;
    }
    if (n_targets == n_sources && sym_wanted(c, n_targets, &y)) {
        if (jsplit) *jsplit = y.nb;
        if (blocks) *blocks = y.grid;
        if (lds_bytes) *lds_bytes = y.block * (int)sizeof(float4);
        return NBODY_OK;
    }
    const Shape s = resolve_shape(c, n_targets, n_sources);
    if (jsplit) *jsplit = s.jsplit;
    if (blocks) *blocks = s.blocks_x * s.jsplit;
    if (lds_bytes) *lds_bytes = (c->kernel == NBODY_KERNEL_STRICT ? 1 : 2) * s.tile * (int)sizeof(float4);
    return NBODY_OK;
}

int nbody_ctx_step_info(nbody_ctx* c, int n, int* symmetric, int* block_bodies, int* slabs, int* workgroups,
                        double* evaluated_pairs)
{
    if (int rc = check_ctx(c)) return rc;
    if (n < 0) return fail(NBODY_ERR_INVALID, "negative size");
    SymShape y{};
    RunShape ry{};
    BalShape by{};
    FusedShape fs{};
    if (fused_wanted(c, n, &fs)) {
        if (symmetric) *symmetric = -1;  // one-sided arithmetic, force and integrate fused in one launch
        if (block_bodies) *block_bodies = fs.T * fs.wv;
        if (slabs) *slabs = 0;
        if (workgroups) *workgroups = fs.grid;
        if (evaluated_pairs) *evaluated_pairs = (double)n * (double)n;
        return NBODY_OK;
    }
    if (bal_wanted(c, n, &by)) {
        if (symmetric) *symmetric = 3;  // symmetric, in balanced runs of rotation steps
        if (block_bodies) *block_bodies = 64 * by.y.bpl;
        if (slabs) *slabs = by.y.smax;
        if (workgroups) *workgroups = (by.y.nworkers + kBalWavesPerGroup - 1) / kBalWavesPerGroup;
        if (evaluated_pairs) *evaluated_pairs = (double)by.y.nsteps * 64.0 * by.y.bpl;
        return NBODY_OK;
    }
    if (run_wanted(c, n, &ry)) {
        if (symmetric) *symmetric = 2;  // symmetric, in runs of chunk units
        if (block_bodies) *block_bodies = ry.layout.bi;
        if (slabs) *slabs = ry.max_slabs;
        if (workgroups) *workgroups = ry.nworkers;
        if (evaluated_pairs) *evaluated_pairs = (double)ry.nunits * ry.layout.bi * 64.0;
        return NBODY_OK;
    }
    if (sym_wanted(c, n, &y)) {
        if (symmetric) *symmetric = 1;
        if (block_bodies) *block_bodies = y.block;
        if (slabs) *slabs = y.nb;
        if (workgroups) *workgroups = y.grid;
        // block pairs I < J once (padded to whole blocks), diagonal blocks both ways
        if (evaluated_pairs) *evaluated_pairs = ((double)y.nb * (y.nb - 1) / 2 + y.nb) * (double)y.block * (double)y.block;
        return NBODY_OK;
    }
    const Shape s = resolve_shape(c, n, n);
    if (symmetric) *symmetric = 0;
    if (block_bodies) *block_bodies = nbk::kWG * s.bpl;
    if (slabs) *slabs = s.jsplit;
    if (workgroups) *workgroups = s.blocks_x * s.jsplit;
    if (evaluated_pairs) *evaluated_pairs = (double)n * (double)n;
    return NBODY_OK;
}

int nbody_ctx_step_info_f64(nbody_ctx* c, int n, int* symmetric, int* block_bodies, int* slabs, int* workgroups, double* evaluated_pairs)
{
    if (int rc = check_ctx(c)) return rc;
    if (n < 0) return fail(NBODY_ERR_INVALID, "negative size");
    int W = 0, BPL = 0, nb = 0;
    if (f64_sym_shape(c, n, &W, &BPL, &nb)) {
        const double B = 64.0 * W * BPL;
        if (symmetric) *symmetric = 1;
        if (block_bodies) *block_bodies = (int)B;
        if (slabs) *slabs = nb;
        if (workgroups) *workgroups = nb * (nb - 1) / 2 + nb;
        if (evaluated_pairs) *evaluated_pairs = ((double)nb * (nb - 1) / 2 + nb) * B * B;
        return NBODY_OK;
    }
    const int blocks_x = (n + nbk::kWG * 2 - 1) / (nbk::kWG * 2);
    int js = c->jsplit;
    if (!js) {
        js = 1;
        while (blocks_x * js < 8 * c->num_cu && js < kMaxSplit) js *= 2;
        const int ntile = (n + 511) / 512;
        while (js > 1 && ntile / js < 2) js /= 2;
    }
    if (symmetric) *symmetric = 0;
    if (block_bodies) *block_bodies = nbk::kWG * 2;
    if (slabs) *slabs = js;
    if (workgroups) *workgroups = blocks_x * js;
    if (evaluated_pairs) *evaluated_pairs = (double)n * (double)n;
    return NBODY_OK;
}

int nbody_ctx_square_info(nbody_ctx* c, int n, int nparts, int* symmetric, int* block_bodies, int* slabs, int* workgroups,
                          double* evaluated_pairs)
{
    if (int rc = check_ctx(c)) return rc;
    if (n < 0 || nparts < 1) return fail(NBODY_ERR_INVALID, "bad square info arguments");
    // one part: the whole-step logic (runs where the cost estimate prefers them); several parts: only a block-pair task
    // list splits, so nbody_accel_square_part launches the block-pair kernel wherever the symmetric kernel applies
    SymShape y{};
    if (nparts == 1 || !sym_wanted(c, n, &y)) return nbody_ctx_step_info(c, n, symmetric, block_bodies, slabs, workgroups, evaluated_pairs);
    if (symmetric) *symmetric = 1;
    if (block_bodies) *block_bodies = y.block;
    if (slabs) *slabs = y.nb;
    if (workgroups) *workgroups = y.grid;
    if (evaluated_pairs) *evaluated_pairs = ((double)y.nb * (y.nb - 1) / 2 + y.nb) * (double)y.block * (double)y.block;
    return NBODY_OK;
}

namespace {

// targets [i0,i1) x sources j0 .. j0+count-1 (indices taken modulo `wrap` when wrap > 0)
int accel_impl(nbody_ctx* c, const nbody_float4* d_bodies, nbody_float4* d_acc_out, int i0, int i1, int j0, int j1,
               int wrap, int accumulate)
{
    const int nt = i1 - i0;
    ON_DEVICE(c);
    // the symmetric decompositions need nb (or max_slabs) slabs of nt bodies: when that allocation fails the cap is lowered and
    // the shape resolved again (a smaller footprint, finally the one-sided kernel)
    for (int attempt = 0; attempt < 16; ++attempt) {
        BalShape by{};
        if (i0 == j0 && i1 == j1 && !wrap && bal_wanted(c, nt, &by)) {
            const int rc = ensure_inbox(c, by);
            if (rc == NBODY_ERR_NOMEM) continue;
            if (rc) return rc;
            nbk::BalParams bp{};
            bp.x = reinterpret_cast<const float4*>(d_bodies) + i0;
            bp.inbox = static_cast<float4*>(c->slabs);
            bp.n = nt;
            bp.y = by.y;
            bp.eps2 = c->eps2;
            if (int rc2 = eq_scan(c, 0, reinterpret_cast<const float4*>(d_bodies), i0, nt, 0, 0, 0, &bp.eqm, &bp.eq_gen)) return rc2;
            if (int rc2 = launch_bal(c, bp, true)) return rc2;
            nbk::BalReduceParams rp{};
            rp.inbox = static_cast<const float4*>(c->slabs);
            rp.y = by.y;
            rp.n = nt;
            rp.a = reinterpret_cast<float4*>(d_acc_out);
            rp.mode = 1;
            rp.accumulate = accumulate ? 1 : 0;
            return launch_bal_reduce(c, rp);
        }
        RunShape ry{};
        SymShape y{};
        size_t need = 0;
        if (i0 == j0 && i1 == j1 && !wrap && run_wanted(c, nt, &ry)) need = (size_t)ry.max_slabs * nt * sizeof(float4);
        else if (i0 == j0 && i1 == j1 && sym_wanted(c, nt, &y)) need = (size_t)y.nb * nt * sizeof(float4);
        if (!need) break;
        const int rc = ensure_slabs(c, need);
        if (rc == NBODY_OK) break;
        if (rc != NBODY_ERR_NOMEM) return rc;
    }
    RunShape ry{};
    if (i0 == j0 && i1 == j1 && !wrap && run_wanted(c, nt, &ry)) {
        // a square block in runs of chunk units (16k ... 128k bodies)
        if (int rc = ensure_slabs(c, (size_t)ry.max_slabs * nt * sizeof(float4))) return rc;
        nbk::RunParams rp{};
        run_params(&rp, reinterpret_cast<const float4*>(d_bodies) + i0, nt, ry, static_cast<float4*>(c->slabs), c->eps2);
        if (int rc = eq_scan(c, 0, reinterpret_cast<const float4*>(d_bodies), i0, nt, 0, 0, 0, &rp.eqm, &rp.eq_gen)) return rc;
        if (int rc = launch_run(c, ry, rp)) return rc;
        nbk::ReduceParams r{};
        r.out = reinterpret_cast<float4*>(d_acc_out);
        r.slabs = static_cast<const float4*>(c->slabs);
        r.nslab = 1;
        r.slab_stride = nt;
        r.n = nt;
        r.accumulate = accumulate ? 1 : 0;
        r.run = ry.layout;
        nbk::reduce_slabs<<<(nt + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(r);
        HIP_TRY(hipGetLastError());
        return NBODY_OK;
    }
    SymShape y{};
    if (i0 == j0 && i1 == j1 && sym_wanted(c, nt, &y)) {
        // a square block (targets == sources): every unordered pair once
        if (int rc = ensure_slabs(c, (size_t)y.nb * nt * sizeof(float4))) return rc;
        nbk::SymParams sp{};
        sym_square_params(&sp, reinterpret_cast<const float4*>(d_bodies), i0, nt, y, static_cast<float4*>(c->slabs), c->eps2);
        if (int rc = eq_scan(c, 0, sp.x, i0, nt, 0, 0, 0, &sp.eqm, &sp.eq_gen)) return rc;
        if (int rc = launch_sym(c, y, sp)) return rc;
        nbk::ReduceParams r{};
        r.out = reinterpret_cast<float4*>(d_acc_out);
        r.slabs = static_cast<const float4*>(c->slabs);
        r.nslab = y.nb;
        r.slab_stride = nt;
        r.n = nt;
        r.accumulate = accumulate ? 1 : 0;
        nbk::reduce_slabs<<<(nt + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(r);
        HIP_TRY(hipGetLastError());
        return NBODY_OK;
    }
    const Shape s = resolve_shape(c, nt, j1 - j0);
    nbk::ForceParams p{};
    p.x = reinterpret_cast<const float4*>(d_bodies);
    p.i0 = i0; p.i1 = i1; p.j0 = j0; p.j1 = j1;
    p.eps2 = c->eps2;
    p.wrap = wrap;
    if (s.jsplit == 1) {
        p.out = reinterpret_cast<float4*>(d_acc_out);
        p.slab_stride = 0;
        p.accumulate = accumulate ? 1 : 0;
        if (j1 == j0 && !accumulate) {
            HIP_TRY(hipMemsetAsync(d_acc_out, 0, (size_t)nt * sizeof(float4), c->stream));
            return NBODY_OK;
        }
        return launch_force(c, s, p);
    }
    if (int rc = ensure_slabs(c, (size_t)s.jsplit * nt * sizeof(float4))) return rc;
    p.out = static_cast<float4*>(c->slabs);
    p.slab_stride = nt;
    p.accumulate = 0;
    if (int rc = launch_force(c, s, p)) return rc;
    nbk::ReduceParams r{};
    r.out = reinterpret_cast<float4*>(d_acc_out);
    r.slabs = static_cast<const float4*>(c->slabs);
    r.nslab = s.jsplit;
    r.slab_stride = nt;
    r.n = nt;
    r.accumulate = accumulate ? 1 : 0;
    nbk::reduce_slabs<<<(nt + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(r);
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

}  // namespace

int nbody_accel_range(nbody_ctx* c, const nbody_float4* d_bodies, nbody_float4* d_acc_out, int i0, int i1,
                      int j0, int j1, int accumulate)
{
    if (int rc = check_ctx(c)) return rc;
    if (i0 < 0 || i1 < i0 || j0 < 0 || j1 < j0) return fail(NBODY_ERR_INVALID, "bad range i[%d,%d) j[%d,%d)", i0, i1, j0, j1);
    if (i1 == i0) return NBODY_OK;
    if (!d_bodies || !d_acc_out) return fail(NBODY_ERR_INVALID, "null device pointer");
    return accel_impl(c, d_bodies, d_acc_out, i0, i1, j0, j1, 0, accumulate);
}

int nbody_accel_square_part(nbody_ctx* c, const nbody_float4* d_bodies, nbody_float4* d_acc_out, int i0, int i1, int accumulate,
                            int part, int nparts)
{
    if (int rc = check_ctx(c)) return rc;
    if (i0 < 0 || i1 < i0 || nparts < 1 || part < 0 || part >= nparts)
        return fail(NBODY_ERR_INVALID, "bad square part: i[%d,%d) part %d of %d", i0, i1, part, nparts);
    if (i1 == i0) return NBODY_OK;
    if (!d_bodies || !d_acc_out) return fail(NBODY_ERR_INVALID, "null device pointer");
    const int nt = i1 - i0;
    SymShape y{};
    ON_DEVICE(c);
    if (part == 0 && nparts > 1) {   // settle the shape (and its workspace) once, before the first part is issued
        for (int attempt = 0; attempt < 16 && sym_wanted(c, nt, &y); ++attempt) {
            const int rc = ensure_slabs(c, (size_t)y.nb * nt * sizeof(float4));
            if (rc == NBODY_OK) break;
            if (rc != NBODY_ERR_NOMEM) return rc;
        }
    }
    // One part, or no block-pair launch for this size / kernel id: the first part is the whole evaluation. (With several
    // parts the block-pair decomposition is used even where runs would be a few per cent faster: only a task list splits.)
    if (nparts == 1 || !sym_wanted(c, nt, &y))
        return part == 0 ? accel_impl(c, d_bodies, d_acc_out, i0, i1, i0, i1, 0, accumulate) : NBODY_OK;
    if (int rc = ensure_slabs(c, (size_t)y.nb * nt * sizeof(float4))) return rc;
    nbk::SymParams sp{};
    sym_square_params(&sp, reinterpret_cast<const float4*>(d_bodies), i0, nt, y, static_cast<float4*>(c->slabs), c->eps2);
    if (int rc = eq_scan(c, 0, sp.x, i0, nt, 0, 0, 0, &sp.eqm, &sp.eq_gen)) return rc;   // every part asks again (a few microseconds; other launches may lie between the parts)
    const long t0 = (long)y.grid * part / nparts, t1 = (long)y.grid * (part + 1) / nparts;
    sp.task0 = (int)t0;
    if (int rc = launch_sym(c, y, sp, (int)(t1 - t0))) return rc;
    if (part != nparts - 1) return NBODY_OK;
    nbk::ReduceParams r{};   // every part has been issued on this stream by now: add the slabs in index order
    r.out = reinterpret_cast<float4*>(d_acc_out);
    r.slabs = static_cast<const float4*>(c->slabs);
    r.nslab = y.nb;
    r.slab_stride = nt;
    r.n = nt;
    r.accumulate = accumulate ? 1 : 0;
    nbk::reduce_slabs<<<(nt + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(r);
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

int nbody_accel_wrapped(nbody_ctx* c, const nbody_float4* d_bodies, int n_total, nbody_float4* d_acc_out, int i0, int i1,
                        int j0, int count, int accumulate)
{
    if (int rc = check_ctx(c)) return rc;
    if (n_total <= 0 || i0 < 0 || i1 < i0 || i1 > n_total || j0 < 0 || j0 >= n_total || count < 0 || count > n_total)
        return fail(NBODY_ERR_INVALID, "bad wrapped range: n=%d i[%d,%d) j0=%d count=%d", n_total, i0, i1, j0, count);
    if (i1 == i0) return NBODY_OK;
    if (!d_bodies || !d_acc_out) return fail(NBODY_ERR_INVALID, "null device pointer");
    return accel_impl(c, d_bodies, d_acc_out, i0, i1, j0, j0 + count, n_total, accumulate);
}

int nbody_accel_cross(nbody_ctx* c, const nbody_float4* d_bodies, int n_total, nbody_float4* d_acc_i, int i0, int i1,
                      int accumulate_i, int j0, int count, nbody_float4* d_acc_j_out)
{
    if (int rc = check_ctx(c)) return rc;
    if (n_total <= 0 || i0 < 0 || i1 < i0 || i1 > n_total || j0 < 0 || j0 >= n_total || count < 0 || count > n_total - (i1 - i0))
        return fail(NBODY_ERR_INVALID, "bad cross range: n=%d i[%d,%d) j0=%d count=%d", n_total, i0, i1, j0, count);
    // the source run j0 .. j0+count-1 (mod n_total) must not meet the targets: a shared body would be paired with itself
    {
        const long a0 = j0, a1 = (long)j0 + count;          // [a0,a1) possibly beyond n_total
        const bool hit = (i0 < a1 && a0 < i1) || (a1 > n_total && i0 < a1 - n_total);
        if (hit && count > 0 && i1 > i0) return fail(NBODY_ERR_INVALID, "cross ranges overlap: i[%d,%d) j0=%d count=%d (n=%d)", i0, i1, j0, count, n_total);
    }
    if (c->kernel == NBODY_KERNEL_STRICT)
        return fail(NBODY_ERR_CONFIG, "nbody_accel_cross is a FAST-arithmetic entry: the strict kernel keeps one sequential sum per target");
    const int ni = i1 - i0;
    if (ni == 0) return NBODY_OK;
    if (!d_bodies || !d_acc_i || (count > 0 && !d_acc_j_out)) return fail(NBODY_ERR_INVALID, "null device pointer");
    ON_DEVICE(c);
    if (count == 0) {
        if (!accumulate_i) HIP_TRY(hipMemsetAsync(d_acc_i, 0, (size_t)ni * sizeof(float4), c->stream));
        return NBODY_OK;
    }
    // The workspace is nbj I-side slabs of ni bodies + nbi J-side slabs of the run. When it exceeds the cap (or cannot be
    // allocated) the source run is cut into pieces that are evaluated one after the other, the I-side sums accumulating:
    // same pair arithmetic, a smaller footprint per launch.
    // The block shape is resolved ONCE, for a whole piece, and the workspace allocated for it before the first launch: the short
    // last piece reuses the shape with fewer source blocks, so nothing can fail once sums have started to accumulate.
    int pieces = 1;
    SymShape y{};
    int nbj_full = 0;
    for (;; pieces *= 2) {
        const int per = (count + pieces - 1) / pieces;
        bool hopeless = false;
        if (sym_resolve_cross(c, ni, per, &y, &nbj_full, &hopeless)) {
            const int rc = ensure_xslabs(c, ((size_t)nbj_full * ni + (size_t)y.nb * per) * sizeof(float4));
            if (rc == NBODY_OK) break;
            if (rc != NBODY_ERR_NOMEM) return rc;
        } else if (hopeless) {   // not a question of workspace: no built shape takes these targets (explicit shape request, block-count limit)
            return fail(NBODY_ERR_CONFIG, "no symmetric kernel shape for %d targets x %d sources (shape request %dx%d)", ni, count, c->sym_waves, c->sym_bpl);
        }
        if (per <= 64) return fail(NBODY_ERR_NOMEM, "no workspace for the symmetric evaluation of %d x %d bodies even in pieces of %d sources", ni, count, per);
    }
    const int per = (count + pieces - 1) / pieces;
    for (int q = 0, done = 0; done < count; ++q, done += per) {
        const int cnt = count - done < per ? count - done : per;
        const int nbj = (cnt + y.block - 1) / y.block;       // <= nbj_full: the footprint below fits what was allocated
        y.grid = y.nb * nbj;
        const size_t islabs = (size_t)nbj * ni;               // then y.nb J-side slabs of cnt bodies
        nbk::SymParams sp{};
        sp.x = reinterpret_cast<const float4*>(d_bodies);
        sp.slabs_i = static_cast<float4*>(c->xslabs);
        sp.slabs_j = static_cast<float4*>(c->xslabs) + islabs;
        sp.ni = ni; sp.nj = cnt;
        sp.i0 = i0; sp.j0 = (int)(((long)j0 + done) % n_total);
        sp.wrap = n_total;
        sp.nbi = y.nb; sp.nbj = nbj;
        sp.stride_i = ni; sp.stride_j = cnt;
        sp.rect = 1;
        sp.eps2 = c->eps2;
        if (int rc = eq_scan(c, 1, sp.x, i0, ni, sp.j0, cnt, n_total, &sp.eqm, &sp.eq_gen)) return rc;
        if (int rc = launch_sym(c, y, sp)) return rc;
        nbk::ReduceParams r{};
        r.out = reinterpret_cast<float4*>(d_acc_i);
        r.slabs = sp.slabs_i;
        r.nslab = nbj;
        r.slab_stride = ni;
        r.n = ni;
        r.accumulate = (accumulate_i || q > 0) ? 1 : 0;
        nbk::reduce_slabs<<<(ni + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(r);
        HIP_TRY(hipGetLastError());
        r.out = reinterpret_cast<float4*>(d_acc_j_out) + done;
        r.slabs = sp.slabs_j;
        r.nslab = y.nb;
        r.slab_stride = cnt;
        r.n = cnt;
        r.accumulate = 0;
        nbk::reduce_slabs<<<(cnt + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(r);
        HIP_TRY(hipGetLastError());
    }
    return NBODY_OK;
}

int nbody_integrate_range(nbody_ctx* c, nbody_float4* d_bodies, nbody_float4* d_velocity, const nbody_float4* d_acc,
                          int i0, int i1)
{
    if (int rc = check_ctx(c)) return rc;
    if (i0 < 0 || i1 < i0) return fail(NBODY_ERR_INVALID, "bad range [%d,%d)", i0, i1);
    const int n = i1 - i0;
    if (n == 0) return NBODY_OK;
    if (!d_bodies || !d_velocity || !d_acc) return fail(NBODY_ERR_INVALID, "null device pointer");
    ON_DEVICE(c);
    nbk::IntegrateParams q{};
    q.x = reinterpret_cast<float4*>(d_bodies) + i0;
    q.v = reinterpret_cast<float4*>(d_velocity);
    q.a = const_cast<float4*>(reinterpret_cast<const float4*>(d_acc));
    q.slabs = nullptr;
    q.nslab = 0;
    q.slab_stride = 0;
    q.n = n;
    q.dt = c->dt;
    nbk::integrate<<<(n + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(q);
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

int nbody_step(nbody_ctx* c, nbody_float4* d_bodies, nbody_float4* d_accelerations, nbody_float4* d_velocity, int n,
               int steps)
{
    if (int rc = check_ctx(c)) return rc;
    if (n < 0 || steps < 0) return fail(NBODY_ERR_INVALID, "n=%d steps=%d", n, steps);
    if (n == 0 || steps == 0) return NBODY_OK;  // an empty system is a no-op, whatever the pointers
    if (!d_bodies || !d_accelerations || !d_velocity) return fail(NBODY_ERR_INVALID, "null device pointer");
    ON_DEVICE(c);
    FusedShape fs{};
    if (fused_wanted(c, n, &fs) && ensure_xalt(c, n) == NBODY_OK) {   // (no spare array to be had: the two-kernel paths below)
        // small systems: one launch per step (force + integrate). Two-array kernel: positions alternate between the caller's array
        // and a spare one, and an odd number of steps ends with a copy-back launch. In-place kernel (nbk::step_fused<.., INPLACE>): the
        // caller's array is read and written by the same launch (used by nbody_simulate, see below).
        float4* const xa = reinterpret_cast<float4*>(d_bodies);
        float4* const xb = static_cast<float4*>(c->xalt);
        // (instrumented runs keep the plain kernel: one event pair per launch; the in-place counter packs workgroups and fall-back waves
        //  into 16 bits each, far beyond any size the fused step is meant for)
        const int mode = (c->timing || (long)fs.grid * fs.wv > 65535) ? 0 : c->fused_inplace;
        // default (-1): in place only where it pays — the odd last step of a call whose caller will wait on the launch's host-mapped
        // word (nbody_simulate). Queued, the two-array kernel is 3.4 us per step faster (the in-place launch ends with a chain of
        // round trips: look at the counter, stores through the L2, count out), and a copy-back launch costs 2.1 (profiles/r04_sync_probe_*.txt).
        const bool auto_inplace = mode < 0 && c->want_host_done && (steps & 1);
        const bool any_inplace = mode >= 1 || auto_inplace;
        if (any_inplace) if (int rc = ensure_fsync(c, (size_t)fs.grid * fs.wv)) return rc;
        nbk::FusedParams fp{};
        fp.v = reinterpret_cast<float4*>(d_velocity);
        fp.a = reinterpret_cast<float4*>(d_accelerations);
        fp.n = n;
        fp.dt = c->dt;
        fp.eps2 = c->eps2;
        c->fdone_armed = false;
        int parity = 0;   // 0: the current positions are in the caller's array
        for (int k = 0; k < steps; ++k) {
            const bool inplace = mode >= 1 || (auto_inplace && k == steps - 1 && parity == 0);
            if (inplace && parity == 0) {
                fp.xin = xa;
                fp.xout = xb;
                fp.sync = c->fsync;
                fp.fb = reinterpret_cast<unsigned char*>(c->fsync + nbk::kFusedSyncWords);
                fp.force_fallback = mode == 2;
                fp.host_word = nullptr;
                if (k == steps - 1 && c->want_host_done) {   // the caller (nbody_simulate) will spin on the host-mapped word
                    fp.host_word = c->fhost_dev;
                    fp.done_value = ++c->fdone_seq;
                    c->fdone_armed = true;
                }
            } else {
                fp.xin = parity ? xb : xa;
                fp.xout = parity ? xa : xb;
                fp.sync = nullptr;
                fp.fb = nullptr;
                fp.host_word = nullptr;
                parity ^= 1;
            }
            if (int rc = launch_fused(c, fs, fp, c->timing)) return rc;
        }
        if (parity) {   // the result belongs in the caller's array
            nbk::copy_bodies<<<(n + 255) / 256, 256, 0, c->stream>>>(xa, xb, n);
            HIP_TRY(hipGetLastError());
        }
        return NBODY_OK;
    }
    SymShape y{};
    RunShape ry{};
    BalShape by{};
    bool bal = false, runs = false, sym = false;
    for (int attempt = 0;; ++attempt) {   // a symmetric footprint that cannot be allocated lowers the cap: resolve again
        bal = bal_wanted(c, n, &by);
        if (bal) {
            const int rc = ensure_inbox(c, by);
            if (rc == NBODY_OK) break;
            if (rc != NBODY_ERR_NOMEM || attempt >= 16) return rc;
            continue;
        }
        runs = run_wanted(c, n, &ry);
        sym = !runs && sym_wanted(c, n, &y);
        if (!runs && !sym) break;
        const int rc = ensure_slabs(c, (runs ? (size_t)ry.max_slabs : (size_t)y.nb) * n * sizeof(float4));
        if (rc == NBODY_OK) break;
        if (rc != NBODY_ERR_NOMEM || attempt >= 16) return rc;
    }
    const Shape s = resolve_shape(c, n, n);
    nbk::ForceParams p{};
    nbk::SymParams sp{};
    nbk::RunParams rp{};
    nbk::IntegrateParams q{};
    q.x = reinterpret_cast<float4*>(d_bodies);
    q.v = reinterpret_cast<float4*>(d_velocity);
    q.a = reinterpret_cast<float4*>(d_accelerations);
    q.n = n;
    q.dt = c->dt;
    nbk::BalParams bp{};
    nbk::BalReduceParams brp{};
    if (bal) {
        bp.x = reinterpret_cast<const float4*>(d_bodies);
        bp.inbox = static_cast<float4*>(c->slabs);
        bp.n = n;
        bp.y = by.y;
        bp.eps2 = c->eps2;
        brp.inbox = static_cast<const float4*>(c->slabs);
        brp.y = by.y;
        brp.n = n;
        brp.x = q.x; brp.v = q.v; brp.a = q.a;
        brp.dt = c->dt;
        brp.mode = 0;
    } else if (runs) {
        if (int rc = ensure_slabs(c, (size_t)ry.max_slabs * n * sizeof(float4))) return rc;
        run_params(&rp, reinterpret_cast<const float4*>(d_bodies), n, ry, static_cast<float4*>(c->slabs), c->eps2);
        q.slabs = static_cast<const float4*>(c->slabs);
        q.nslab = 1;
        q.slab_stride = n;
        q.run = ry.layout;
    } else if (sym) {
        if (int rc = ensure_slabs(c, (size_t)y.nb * n * sizeof(float4))) return rc;
        sym_square_params(&sp, reinterpret_cast<const float4*>(d_bodies), 0, n, y, static_cast<float4*>(c->slabs), c->eps2);
        q.slabs = static_cast<const float4*>(c->slabs);
        q.nslab = y.nb;
        q.slab_stride = n;
    } else {
        p.x = reinterpret_cast<const float4*>(d_bodies);
        p.i0 = 0; p.i1 = n; p.j0 = 0; p.j1 = n;
        p.eps2 = c->eps2;
        p.accumulate = 0;
        if (s.jsplit == 1) {
            p.out = q.a;
            p.slab_stride = 0;
            q.slabs = nullptr;
            q.nslab = 0;
            q.slab_stride = 0;
        } else {
            if (int rc = ensure_slabs(c, (size_t)s.jsplit * n * sizeof(float4))) return rc;
            p.out = static_cast<float4*>(c->slabs);
            p.slab_stride = n;
            q.slabs = static_cast<const float4*>(c->slabs);
            q.nslab = s.jsplit;
            q.slab_stride = n;
        }
    }
    const bool graphable = !c->timing && (c->use_graph == 1 || (c->use_graph < 0 && n <= kGraphMaxN));
    // equal-mass path of the symmetric kernels: one scan per call and one more every kEqRescanSteps steps of a long call (the
    // integrate carries the masses through unchanged; the scan's coordinate bound of 1e15 leaves a factor of 1000 before a padding
    // lane at 1e18 could contribute anything but an exact zero, and no body crosses that in a thousand steps). A context with graph
    // replay switched ON never takes the path — whatever the number of steps of the call, so that run(k) and k x run(1) give the same
    // bits (the generation number would be frozen into a captured graph). Same size rule as nbody_accel_range, so that a step and
    // the accel + integrate pair it is made of keep giving the same bits.
    constexpr int kEqRescanSteps = 1024;
    const bool eq_path = (bal || runs || sym) && !graphable;
    auto scan_masses = [&]() -> int {
        const nbk::MassInfo* q = nullptr;
        unsigned int gen = 0;
        if (int rc = eq_scan(c, 0, reinterpret_cast<const float4*>(d_bodies), 0, n, 0, 0, 0, &q, &gen)) return rc;
        bp.eqm = q; bp.eq_gen = gen;
        rp.eqm = q; rp.eq_gen = gen;
        sp.eqm = q; sp.eq_gen = gen;
        return NBODY_OK;
    };
    if (eq_path) if (int rc = scan_masses()) return rc;
    const int iblocks = (n + nbk::kWG - 1) / nbk::kWG;
    // one step = one force launch + one integrate launch, both checked
    auto enqueue_step = [&](bool timed) -> int {
        int rc;
        if (bal) {
            if (int rb = launch_bal(c, bp, timed)) return rb;
            return launch_bal_reduce(c, brp);   // the inbox sum and the integrate in one kernel
        }
        if (runs) rc = launch_run(c, ry, rp);
        else if (sym) rc = timed ? launch_sym(c, y, sp) : launch_sym_untimed(c, y, sp);
        else rc = timed ? launch_force(c, s, p) : launch_force_untimed(c, s, p);
        if (rc != NBODY_OK) return rc;
        nbk::integrate<<<iblocks, nbk::kWG, 0, c->stream>>>(q);
        HIP_TRY(hipGetLastError());
        return NBODY_OK;
    };
    int k = 0;
    if (graphable && steps >= kGraphChunk) {
        // Launch-bound regime: replay a captured chain of kGraphChunk steps instead of 2*kGraphChunk
        // host launches. The kernels and their order are exactly those of the loop below.
        const nbody_ctx::GraphKey key{d_bodies, q.a, q.v, c->slabs, n, bal ? by.y.bpl : runs ? ry.bpl : sym ? y.bpl : s.bpl,
                                      bal ? -2 : runs ? -1 : sym ? y.waves : s.tile, bal ? by.y.L : runs ? ry.layout.L : sym ? y.nb : s.jsplit,
                                      c->kernel, kGraphChunk, c->dt, c->eps2, c->stream};
        if (!c->graph_exec || !(key == c->graph_key)) {
            if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
            if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
            HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            int rc = NBODY_OK;
            for (int g = 0; g < kGraphChunk && rc == NBODY_OK; ++g) rc = enqueue_step(false);
            hipGraph_t captured = nullptr;
            const hipError_t ce = hipStreamEndCapture(c->stream, &captured);  // always end the capture
            if (rc != NBODY_OK || ce != hipSuccess) {
                if (captured) (void)hipGraphDestroy(captured);  // a partial graph is never kept
                if (rc != NBODY_OK) return rc;
                return fail(NBODY_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
            }
            c->graph = captured;
            const hipError_t ie = hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0);
            if (ie != hipSuccess) {
                (void)hipGraphDestroy(c->graph);
                c->graph = nullptr;
                c->graph_exec = nullptr;
                return fail(NBODY_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
            }
            c->graph_key = key;
        }
        for (; k + kGraphChunk <= steps; k += kGraphChunk) HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
    }
    for (; k < steps; ++k) {
        if (eq_path && k > 0 && k % kEqRescanSteps == 0) if (int rc = scan_masses()) return rc;
        if (int rc = enqueue_step(true)) return rc;
    }
    return NBODY_OK;
}

// The rule by which a timing measurement may override the built-in decomposition (pure host logic: tests/test_abi.py). 1 = override.
//  (a) the built-in choice, timed first and last, agrees with itself within 10 % (else the machine is not quiet);
//  (b) the challenger's best time beats the built-in's best by more than `margin`;
//  (c) when confirmation trials are given: EVERY challenger trial beats EVERY built-in trial by more than `margin`.
extern "C" int nbody_autotune_decide(double builtin_first_us, double builtin_last_us, double challenger_us, const double* confirm_builtin_us,
                                     const double* confirm_challenger_us, int n_confirm, double margin)
{
    if (!(builtin_first_us > 0.0) || !(builtin_last_us > 0.0) || !(challenger_us > 0.0) || !(margin > 0.0) || n_confirm < 0) return 0;
    const double lo = builtin_first_us < builtin_last_us ? builtin_first_us : builtin_last_us;
    const double hi = builtin_first_us < builtin_last_us ? builtin_last_us : builtin_first_us;
    if (hi > 1.10 * lo) return 0;
    if (!(challenger_us < lo * (1.0 - margin))) return 0;
    if (n_confirm > 0) {
        if (!confirm_builtin_us || !confirm_challenger_us) return 0;
        double ch_max = 0.0, bi_min = 1e300;
        for (int k = 0; k < n_confirm; ++k) {
            if (!(confirm_builtin_us[k] > 0.0) || !(confirm_challenger_us[k] > 0.0)) return 0;
            if (confirm_builtin_us[k] < bi_min) bi_min = confirm_builtin_us[k];
            if (confirm_challenger_us[k] > ch_max) ch_max = confirm_challenger_us[k];
        }
        if (!(ch_max < bi_min * (1.0 - margin))) return 0;
    }
    return 1;
}

namespace {

struct TuneKnobs { int fused, sym_runs, sym_bpl, sym_waves; };

// Times whole steps of n bodies (scratch copies, dt = 0) under every decomposition that applies and returns the fastest. With
// `keep_builtin_within` > 0 the context's CURRENT knobs are measured first as candidate 0 and kept unless another candidate is
// faster by more than that fraction (so that timing noise cannot flip a choice between two runs of the same program).
int tune_measure(nbody_ctx* c, const nbody_float4* d_bodies, int n, int steps_per_trial, double keep_builtin_within, int* out_choice,
                 TuneKnobs* out_knobs, double* out_us_best, double* out_us_builtin)
{
    const size_t bytes = (size_t)n * sizeof(float4);
    float4 *xs = nullptr, *vs = nullptr, *as = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&] {
        if (xs) (void)hipFree(xs);
        if (vs) (void)hipFree(vs);
        if (as) (void)hipFree(as);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    if (hipMalloc(reinterpret_cast<void**>(&xs), bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&vs), bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&as), bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        cleanup();
        return fail(NBODY_ERR_NOMEM, "autotune: cannot allocate scratch state for %d bodies", n);
    }
    const TuneKnobs saved{c->fused, c->sym_runs, c->sym_bpl, c->sym_waves};
    const float saved_dt = c->dt;
    const bool saved_timing = c->timing;
    const int saved_inplace = c->fused_inplace;
    c->dt = 0.0f;          // the trial steps leave the scratch positions where they are
    c->timing = false;
    c->fused_inplace = 0;  // queued trial steps: the two-array kernel, as nbody_step runs them
    // choice id: 0 the knobs as they were, 1 fused step, 2x balanced runs with x bodies per lane (24, 28, 210), 3 unit runs, 4 block pairs / two-kernel one-sided
    struct Cand { int id; TuneKnobs k; };
    // (the built-in choice is timed FIRST and LAST and its better time counts: the first trial of a series runs on a colder chip,
    // and a candidate that is the same kernel as the built-in one must not "win" by that)
    const Cand cands[] = {{0, saved}, {1, {1, -1, 0, 0}}, {24, {0, 2, 4, 0}}, {28, {0, 2, 8, 0}}, {210, {0, 2, 10, 0}}, {3, {0, 1, 0, 0}}, {4, {0, 0, 0, 0}}, {0, saved}};
    int best = -1;
    double best_us = 0.0, builtin_us = 0.0, builtin_first = 0.0, builtin_last = 0.0;
    TuneKnobs best_k = saved;
    int rc = NBODY_OK;
    auto one_trial = [&]() -> double {   // microseconds per step of `steps_per_trial` queued steps with the context's current knobs; < 0: failed
        (void)hipEventRecord(e0, c->stream);
        if (nbody_step(c, reinterpret_cast<nbody_float4*>(xs), reinterpret_cast<nbody_float4*>(as), reinterpret_cast<nbody_float4*>(vs), n, steps_per_trial) != NBODY_OK) return -1.0;
        (void)hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess) return -1.0;
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return (double)ms * 1e3 / steps_per_trial;
    };
    for (const Cand& cd : cands) {
        if (cd.id == 0 && keep_builtin_within <= 0.0) continue;
        if (cd.id == 1 && n > 65536) continue;                      // the one-sided fused step cannot win there; do not spend seconds on it
        c->fused = cd.k.fused; c->sym_runs = cd.k.sym_runs; c->sym_bpl = cd.k.sym_bpl; c->sym_waves = cd.k.sym_waves;
        int kind = 0;
        if (nbody_ctx_step_info(c, n, &kind, nullptr, nullptr, nullptr, nullptr) != NBODY_OK) continue;
        if (cd.id != 0) {
            const int want = cd.id == 1 ? -1 : cd.id >= 24 ? 3 : cd.id == 3 ? 2 : kind;   // the decomposition the knobs were meant to select
            if (kind != want || (cd.id == 4 && kind != 0 && kind != 1)) continue;          // does not apply at this size
        }
        if (hipMemcpyAsync(xs, d_bodies, bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
            hipMemsetAsync(vs, 0, bytes, c->stream) != hipSuccess) { rc = fail(NBODY_ERR_HIP, "autotune: scratch setup failed"); break; }
        rc = nbody_step(c, reinterpret_cast<nbody_float4*>(xs), reinterpret_cast<nbody_float4*>(as), reinterpret_cast<nbody_float4*>(vs), n, 4);   // warm-up, workspace
        if (rc != NBODY_OK) { rc = NBODY_OK; continue; }           // this decomposition cannot run here (workspace): skip it
        double us = 1e30;
        for (int rep = 0; rep < 3 && rc == NBODY_OK; ++rep) {
            const double t = one_trial();
            if (t < 0.0) { rc = fail(NBODY_ERR_HIP, "autotune: trial failed"); break; }
            if (t < us) us = t;
        }
        if (rc != NBODY_OK) break;
        if (cd.id == 0) {
            if (builtin_us == 0.0) builtin_first = us;
            builtin_last = us;
            if (builtin_us == 0.0 || us < builtin_us) builtin_us = us;
            continue;
        }
        if (best < 0 || us < best_us) { best = cd.id; best_us = us; best_k = cd.k; }
    }
    if (rc == NBODY_OK && keep_builtin_within > 0.0 && builtin_us > 0.0) {
        // The built-in choice is only overridden by a CLEAR and REPEATABLE win — what this call decides also decides the low-order bits
        // of every later result, and a busy GPU (another process, another stream) makes single timings worthless:
        //  (a) the built-in choice, timed first and last, must agree with itself within 10 % (else the machine is not quiet: keep it);
        //  (b) the challenger must be faster by more than the margin;
        //  (c) and again in a confirmation round: three alternating trials each, EVERY challenger trial faster than EVERY built-in trial by the margin.
        bool override_it = best > 0 && nbody_autotune_decide(builtin_first, builtin_last, best_us, nullptr, nullptr, 0, keep_builtin_within) == 1;
        if (override_it) {
            double tb[3] = {0, 0, 0}, tc[3] = {0, 0, 0};
            for (int round = 0; round < 3 && override_it; ++round) {
                c->fused = saved.fused; c->sym_runs = saved.sym_runs; c->sym_bpl = saved.sym_bpl; c->sym_waves = saved.sym_waves;
                tb[round] = one_trial();
                c->fused = best_k.fused; c->sym_runs = best_k.sym_runs; c->sym_bpl = best_k.sym_bpl; c->sym_waves = best_k.sym_waves;
                tc[round] = one_trial();
                if (tb[round] < 0.0 || tc[round] < 0.0) override_it = false;
            }
            if (override_it) override_it = nbody_autotune_decide(builtin_first, builtin_last, best_us, tb, tc, 3, keep_builtin_within) == 1;
        }
        if (!override_it) { best = 0; best_us = builtin_us; best_k = saved; }
    }
    c->dt = saved_dt;
    c->timing = saved_timing;
    c->fused_inplace = saved_inplace;
    c->fused = saved.fused; c->sym_runs = saved.sym_runs; c->sym_bpl = saved.sym_bpl; c->sym_waves = saved.sym_waves;
    (void)hipStreamSynchronize(c->stream);
    cleanup();
    if (rc != NBODY_OK) return rc;
    if (best < 0) return fail(NBODY_ERR_CONFIG, "autotune: no decomposition ran for %d bodies", n);
    *out_choice = best;
    *out_knobs = best_k;
    *out_us_best = best_us;
    if (out_us_builtin) *out_us_builtin = builtin_us;
    return NBODY_OK;
}


// Is n within a quarter of one of the built-in switch-over sizes (measured on one pool of MI355X boxes with one compiler)?
bool near_switch_over(int n)
{
    for (const int s : {kFusedMaxAuto, kBalMaxAuto, kRunsMaxAuto})
        if ((double)n >= 0.75 * s && (double)n <= 1.25 * s) return true;
    return false;
}

// Measuring inside nbody_simulate() is OPT-IN (NBODY_AUTOTUNE=1 in the environment, looked at on each eligible call so that a host
// program may set it after loading the library): a caller that never asked for tuning — the reference's loop, main.cpp:146-156 —
// gets the built-in decomposition, hence the same low-order bits on every machine, and a first call that costs no measurement.
// profiles/r04_autotune_probe.jsonl: the built-in choice was kept at all 13 sizes measured.
bool autotune_enabled()
{
    const char* e = std::getenv("NBODY_AUTOTUNE");
    return e && *e && *e != '0';
}

}  // namespace

// Measures the decompositions that apply to whole steps of n bodies on THIS device and leaves the context's knobs (fused step,
// runs mode, bodies per lane) on the fastest: the switch-over sizes compiled into the library were measured on one pool of
// MI355X boxes with one compiler; a different chip or ROCm release may move them.
int nbody_ctx_autotune(nbody_ctx* c, const nbody_float4* d_bodies, int n, int steps_per_trial, int* out_choice, double* out_us_per_step)
{
    if (int rc = check_ctx(c)) return rc;
    if (n < 1 || steps_per_trial < 1 || !d_bodies) return fail(NBODY_ERR_INVALID, "bad autotune arguments");
    if (c->kernel != NBODY_KERNEL_FAST) return fail(NBODY_ERR_CONFIG, "autotune chooses among the FAST kernel's decompositions");
    ON_DEVICE(c);
    int best = 0;
    TuneKnobs k{};
    double us = 0.0;
    if (int rc = tune_measure(c, d_bodies, n, steps_per_trial, 0.0, &best, &k, &us, nullptr)) return rc;
    c->fused = k.fused; c->sym_runs = k.sym_runs; c->sym_bpl = k.sym_bpl; c->sym_waves = k.sym_waves;
    if (out_choice) *out_choice = best;
    if (out_us_per_step) *out_us_per_step = us;
    return NBODY_OK;
}

// Pins the decomposition nbody_simulate() uses for n bodies on this context, as a measurement would have: choice 0 = the built-in
// one, an id of nbody_ctx_autotune = that decomposition, -1 = forget n (the next eligible call measures again).
int nbody_ctx_set_autotuned(nbody_ctx* c, int n, int choice)
{
    if (int rc = check_ctx(c)) return rc;
    if (n < 1) return fail(NBODY_ERR_INVALID, "n=%d", n);
    if (choice == -1) { c->tuned.erase(n); return NBODY_OK; }
    nbody_ctx::Tuned t{choice, -1, -1, 0, 0, 0.0, 0.0};
    switch (choice) {
        case 0: break;
        case 1: t.fused = 1; break;
        case 24: t.fused = 0; t.sym_runs = 2; t.sym_bpl = 4; break;
        case 28: t.fused = 0; t.sym_runs = 2; t.sym_bpl = 8; break;
        case 210: t.fused = 0; t.sym_runs = 2; t.sym_bpl = 10; break;
        case 3: t.fused = 0; t.sym_runs = 1; break;
        case 4: t.fused = 0; t.sym_runs = 0; break;
        default: return fail(NBODY_ERR_CONFIG, "unknown decomposition id %d (0, 1, 24, 28, 210, 3, 4)", choice);
    }
    c->tuned[n] = t;
    return NBODY_OK;
}

// What nbody_simulate() found when it measured whole steps of n bodies on this context (see nbody.h). choice 0 = the built-in
// decomposition was kept; -1 = this size has not been measured (NBODY_AUTOTUNE not set, not near a switch-over, explicit knobs, or no call yet).
int nbody_ctx_autotuned(nbody_ctx* c, int n, int* out_choice, double* out_us_builtin, double* out_us_best)
{
    if (int rc = check_ctx(c)) return rc;
    const auto it = c->tuned.find(n);
    if (out_choice) *out_choice = it == c->tuned.end() ? -1 : it->second.choice;
    if (out_us_builtin) *out_us_builtin = it == c->tuned.end() ? 0.0 : it->second.us_builtin;
    if (out_us_best) *out_us_best = it == c->tuned.end() ? 0.0 : it->second.us_best;
    return NBODY_OK;
}

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
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->timing = enable != 0;
    c->events_used = 0;
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

namespace {

// Waits (bounded: 2 ms) for the host-mapped word to take the value the last armed launch writes. true: seen — everything queued on the
// stream before that launch's last store is complete and in memory; false: not armed, or not seen in time (the caller synchronises).
bool wait_host_word(nbody_ctx* c)
{
    if (!c->fdone_armed || !c->fhost) return false;
    c->fdone_armed = false;
    volatile unsigned long long* const w = c->fhost;
    const unsigned want = c->fdone_seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if ((unsigned)*w == want) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return true;
        }
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) return false;
    }
}

// Are all three arrays ordinary device allocations (hipMalloc)? Looked up on every call that asks — a pointer value can come back
// as another kind of memory after a free — which costs well under a microsecond on calls of 60 us and more (N > 8192).
bool arrays_are_device_memory(const void* x, const void* a, const void* v)
{
    for (const void* p : {x, a, v}) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();   // not known to the runtime (pageable host memory under HMM, ...): not device memory
            return false;
        }
        if (at.type != hipMemoryTypeDevice || at.isManaged) return false;
    }
    return true;
}

// One host_signal launch behind whatever is queued; then wait_host_word() can stand in for a stream synchronisation.
void arm_host_signal(nbody_ctx* c)
{
    if (ensure_fsync(c, 1) != NBODY_OK) return;
    nbk::host_signal<<<1, 64, 0, c->stream>>>(c->fhost_dev, (unsigned long long)++c->fdone_seq);
    if (hipGetLastError() == hipSuccess) c->fdone_armed = true;
    else --c->fdone_seq;
}

}  // namespace

int nbody_ctx_sync(nbody_ctx* c)
{
    if (int rc = check_ctx(c)) return rc;
    // (a host_signal launch + a spin here, as nbody_simulate does, measured 1 us per call at best — 34.7 -> 33.8 us for a step + sync at
    //  N = 8192 — and nothing for long queues: not adopted, the plain synchronisation stays)
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NBODY_OK;
}

int nbody_ctx_get(nbody_ctx* c, int* device, int* kernel, void** hip_stream)
{
    if (int rc = check_ctx(c)) return rc;
    if (device) *device = c->device;
    if (kernel) *kernel = c->kernel;
    if (hip_stream) *hip_stream = static_cast<void*>(c->stream);
    return NBODY_OK;
}

namespace {

bool simulate_knobs_default(const nbody_ctx* c)
{
    return c->kernel == NBODY_KERNEL_FAST && c->fused == -1 && c->sym_runs == -1 && c->sym_bpl == 0 && c->sym_waves == 0 &&
           c->tile == 0 && c->bpl == 0 && c->jsplit == 0 && c->use_graph == 0 && !c->timing;
}

// With NBODY_AUTOTUNE=1: near a built-in switch-over size the decomposition is MEASURED once per size on this device (scratch copies
// of the caller's bodies, a few tens of milliseconds) instead of trusted: the sizes were measured on one pool of boxes with one
// compiler. The built-in choice is kept unless another one wins clearly and repeatably (nbody_autotune_decide). Any explicit knob
// switches this off. Without the variable nothing is measured: pinned choices (nbody_ctx_set_autotuned) still apply.
int simulate_prepare_locked(nbody_ctx* c, const nbody_float4* d_bodies, int n)
{
    if (!(simulate_knobs_default(c) && n > 0 && d_bodies && near_switch_over(n) && !c->tuned.count(n) && autotune_enabled())) return NBODY_OK;
    ON_DEVICE(c);
    FusedShape fs0{};
    const double est_us = 2.0 + (double)n * n / (fused_wanted(c, n, &fs0) ? 3.2e6 : 5.5e6);   // rough step time: a trial lasts about 10 ms
    int trial = (int)(10000.0 / est_us);
    trial = trial < 3 ? 3 : trial > 50 ? 50 : trial;
    int choice = 0;
    TuneKnobs k{};
    double us_best = 0.0, us_builtin = 0.0;
    if (tune_measure(c, d_bodies, n, trial, 0.03, &choice, &k, &us_best, &us_builtin) == NBODY_OK)
        c->tuned[n] = nbody_ctx::Tuned{choice, k.fused, k.sym_runs, k.sym_bpl, k.sym_waves, us_builtin, us_best};
    else
        c->tuned[n] = nbody_ctx::Tuned{0, c->fused, c->sym_runs, c->sym_bpl, c->sym_waves, 0.0, 0.0};   // measurement failed: the built-in choice, and do not try again
    return NBODY_OK;
}

}  // namespace

// Everything nbody_simulate() would otherwise do inside its FIRST call for n bodies: workspaces, the device code, and — near a
// switch-over size — the measurement of the decompositions on scratch copies of d_bodies (which are only read).
int nbody_simulate_prepare(const nbody_float4* d_bodies, int n)
{
    std::lock_guard<std::recursive_mutex> lk(g_default_mu);
    nbody_ctx* c = nullptr;
    if (int rc = nbody_default_ctx(&c)) return rc;
    if (n < 0) return fail(NBODY_ERR_INVALID, "n=%d", n);
    if (int rc = nbody_ctx_reserve(c, n)) return rc;
    return simulate_prepare_locked(c, d_bodies, n);
}

int nbody_simulate(nbody_float4* d_bodies, nbody_float4* d_accelerations, nbody_float4* d_velocity, int n)
{
    std::lock_guard<std::recursive_mutex> lk(g_default_mu);
    nbody_ctx* c = nullptr;
    if (int rc = nbody_default_ctx(&c)) return rc;
    if (int rc = simulate_prepare_locked(c, d_bodies, n)) return rc;
    const bool knobs_default = simulate_knobs_default(c);
    const auto tuned = knobs_default ? c->tuned.find(n) : c->tuned.end();
    const bool apply = tuned != c->tuned.end() && tuned->second.choice > 0;
    if (apply) { c->fused = tuned->second.fused; c->sym_runs = tuned->second.sym_runs; c->sym_bpl = tuned->second.sym_bpl; c->sym_waves = tuned->second.sym_waves; }
    c->fdone_armed = false;
    c->want_host_done = true;    // a fused in-place step ends by writing a host-mapped word once all its results are visible
    const int rc = nbody_step(c, d_bodies, d_accelerations, d_velocity, n, 1);
    c->want_host_done = false;
    if (apply) { c->fused = -1; c->sym_runs = -1; c->sym_bpl = 0; c->sym_waves = 0; }
    if (rc) return rc;
    if (!c->fdone_armed && !c->timing && n > 0 && c->fused_inplace != 0) {   // (mode 0 = round 3's behaviour, the A/B: a plain stream synchronisation)
        // the other paths (balanced runs, unit runs, block pairs: two or three launches per step): one tiny launch behind them writes
        // the same host-mapped word — a launch boundary (1.5-2 us) instead of the 4 us a stream synchronisation costs over a spin.
        // What makes the step's ordinary stores visible before that word is the release between two kernels of one stream: enough for
        // DEVICE memory read next through a HIP copy or a kernel (the reference's arrays: cudaMalloc, main.cpp:275-283), not promised
        // for host-mapped or managed arrays the CPU reads directly — those get the stream synchronisation and its system-scope release.
        // (The in-place fused step above needs no such distinction: its results are system-scope stores, drained before its word.)
        ON_DEVICE(c);
        if (arrays_are_device_memory(d_bodies, d_accelerations, d_velocity)) arm_host_signal(c);
    }
    // simulate() is synchronous (kernel.cu:644). Waiting for the launch's own word costs about 4 us less per call than
    // hipStreamSynchronize (profiles/r04_sync_probe_*.txt); the stream synchronisation stays as the backstop (and reports errors).
    if (wait_host_word(c)) return NBODY_OK;
    ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NBODY_OK;
}

// The older snapshot's boundary (Sim-Without-OpenGL-Integration/kernel.cuh:5, kernel.cu:85-125): HOST
// pointers, float3 velocity/acceleration, copy-in / launch / copy-out on every call, DT = 0.01 and
// EPS2 = 0.002 compiled in as double literals. Device staging is kept in the default context
// instead of being re-allocated (and leaked) per call as the original does (kernel.cu:94-96).
int nbody_simulate_host_legacy(nbody_float4* h_bodies, nbody_float3* h_accelerations, nbody_float3* h_velocity, int n)
{
    std::lock_guard<std::recursive_mutex> lk(g_default_mu);  // eps2 / legacy_eps of the shared context are switched below
    nbody_ctx* c = nullptr;
    if (int rc = nbody_default_ctx(&c)) return rc;
    if (n < 0) return fail(NBODY_ERR_INVALID, "n=%d", n);
    if (n == 0) return NBODY_OK;
    if (!h_bodies || !h_accelerations || !h_velocity) return fail(NBODY_ERR_INVALID, "null host pointer");
    ON_DEVICE(c);
    const size_t size4 = sizeof(float4) * (size_t)n, size3 = 3 * sizeof(float) * (size_t)n;
    const size_t need = size4 + 2 * size3 + 64;
    if (need > c->legacy_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->legacy_buf) HIP_TRY(hipFree(c->legacy_buf));
        c->legacy_buf = nullptr;
        c->legacy_bytes = 0;
        HIP_TRY(hipMalloc(&c->legacy_buf, need));
        c->legacy_bytes = need;
    }
    char* base = static_cast<char*>(c->legacy_buf);
    float4* d_bodies = reinterpret_cast<float4*>(base);
    float* d_vel = reinterpret_cast<float*>(base + size4);
    float* d_acc = reinterpret_cast<float*>(base + size4 + ((size3 + 15) / 16) * 16);
    HIP_TRY(hipMemcpyAsync(d_bodies, h_bodies, size4, hipMemcpyHostToDevice, c->stream));      // kernel.cu:99-101
    HIP_TRY(hipMemcpyAsync(d_vel, h_velocity, size3, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_acc, h_accelerations, size3, hipMemcpyHostToDevice, c->stream));

    const float saved_eps2 = c->eps2;
    const bool saved_legacy = c->legacy_eps;
    c->eps2 = 0.002f;
    c->legacy_eps = true;
    const Shape s = resolve_shape(c, n, n);
    int rc = ensure_slabs(c, (size_t)s.jsplit * n * sizeof(float4));
    if (rc == NBODY_OK) {
        nbk::ForceParams p{};
        p.x = d_bodies;
        p.out = static_cast<float4*>(c->slabs);
        p.i0 = 0; p.i1 = n; p.j0 = 0; p.j1 = n;
        p.slab_stride = n;
        p.accumulate = 0;
        p.eps2 = c->eps2;
        rc = launch_force(c, s, p);
    }
    c->eps2 = saved_eps2;
    c->legacy_eps = saved_legacy;
    if (rc != NBODY_OK) return rc;
    nbk::IntegrateLegacyParams q{};
    q.x = d_bodies;
    q.v3 = d_vel;
    q.a3 = d_acc;
    q.slabs = static_cast<const float4*>(c->slabs);
    q.nslab = s.jsplit;
    q.slab_stride = n;
    q.n = n;
    nbk::integrate_legacy<<<(n + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(q);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_bodies, d_bodies, size4, hipMemcpyDeviceToHost, c->stream));      // kernel.cu:115-124
    HIP_TRY(hipMemcpyAsync(h_velocity, d_vel, size3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NBODY_OK;
}

int nbody_step_f64(nbody_ctx* c, nbody_double4* d_bodies, nbody_double4* d_accelerations, nbody_double4* d_velocity,
                   int n, int steps, double dt, double eps2)
{
    if (int rc = check_ctx(c)) return rc;
    if (!d_bodies || !d_accelerations || !d_velocity) return fail(NBODY_ERR_INVALID, "null device pointer");
    if (n < 0 || steps < 0) return fail(NBODY_ERR_INVALID, "n=%d steps=%d", n, steps);
    if (!(eps2 > 0.0)) return fail(NBODY_ERR_INVALID, "eps2 must be > 0");
    if (n == 0 || steps == 0) return NBODY_OK;
    ON_DEVICE(c);
    nbk::IntegrateParamsF64 q{};
    q.x = reinterpret_cast<double4*>(d_bodies);
    q.v = reinterpret_cast<double4*>(d_velocity);
    q.a = reinterpret_cast<double4*>(d_accelerations);
    q.n = n;
    q.dt = dt;
    // the symmetric rotation kernel in double; a footprint that cannot be allocated lowers the cap and the choice is made again
    for (int attempt = 0; attempt < 4; ++attempt) {
        int W = 0, BPL = 0, nb = 0;
        if (!f64_sym_shape(c, n, &W, &BPL, &nb)) break;   // the one-sided kernel below
        const int arc = ensure_slabs(c, (size_t)nb * n * sizeof(double4));
        if (arc == NBODY_ERR_NOMEM) continue;
        if (arc) return arc;
        nbk::SymParamsF64 sp{};
        sp.x = reinterpret_cast<const double4*>(d_bodies);
        sp.slabs_i = static_cast<double4*>(c->slabs);
        sp.slabs_j = sp.slabs_i;
        sp.ni = n; sp.nj = n; sp.i0 = 0; sp.j0 = 0; sp.wrap = 0;
        sp.nbi = nb; sp.nbj = nb; sp.stride_i = n; sp.stride_j = n; sp.rect = 0;
        sp.eps2 = eps2;
        if (int rc = eq_scan(c, 0, sp.x, 0, n, 0, 0, 0, &sp.eqm, &sp.eq_gen)) return rc;   // the equal-mass path, in double
        q.slabs = static_cast<const double4*>(c->slabs);
        q.nslab = nb;
        q.slab_stride = n;
        const int grid = nb * (nb - 1) / 2 + nb;
        for (int k = 0; k < steps; ++k) {
            if (int rc = time_mark(c)) return rc;
            switch (W * 100 + BPL) {
                case 408: nbk::force_sym<nbk::SymF64<8>, 4><<<grid, 256, 0, c->stream>>>(sp); break;
                case 406: nbk::force_sym<nbk::SymF64<6>, 4><<<grid, 256, 0, c->stream>>>(sp); break;
                case 204: nbk::force_sym<nbk::SymF64<4>, 2><<<grid, 128, 0, c->stream>>>(sp); break;
                default: nbk::force_sym<nbk::SymF64<2>, 1><<<grid, 64, 0, c->stream>>>(sp); break;
            }
            HIP_TRY(hipGetLastError());
            if (int rc = time_mark(c)) return rc;
            nbk::integrate_f64<<<(n + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(q);
            HIP_TRY(hipGetLastError());
        }
        return NBODY_OK;
    }
    // one-sided LDS-tiled kernel
    constexpr int BPL = 2, TILE = 512;
    const int blocks_x = (n + nbk::kWG * BPL - 1) / (nbk::kWG * BPL);
    int js = c->jsplit;
    if (!js) {
        js = 1;
        while (blocks_x * js < 8 * c->num_cu && js < kMaxSplit) js *= 2;
        const int ntile = (n + TILE - 1) / TILE;
        while (js > 1 && ntile / js < 2) js /= 2;
    }
    if (int rc = ensure_slabs(c, (size_t)js * n * sizeof(double4))) return rc;
    nbk::ForceParamsF64 p{};
    p.x = reinterpret_cast<const double4*>(d_bodies);
    p.out = static_cast<double4*>(c->slabs);
    p.n = n;
    p.slab_stride = n;
    p.eps2 = eps2;
    q.slabs = static_cast<const double4*>(c->slabs);
    q.nslab = js;
    q.slab_stride = n;
    for (int k = 0; k < steps; ++k) {
        if (int rc = time_mark(c)) return rc;
        nbk::force_f64<BPL, TILE><<<dim3(blocks_x, js), nbk::kWG, 0, c->stream>>>(p);
        HIP_TRY(hipGetLastError());
        if (int rc = time_mark(c)) return rc;
        nbk::integrate_f64<<<(n + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(q);
        HIP_TRY(hipGetLastError());
    }
    return NBODY_OK;
}

// utils.cpp:50-68 — the same lines, from hipDeviceProp_t of the current device (the reference reads device 0's
// cudaDeviceProp; "Warp size" prints the 64-lane wavefront here)
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

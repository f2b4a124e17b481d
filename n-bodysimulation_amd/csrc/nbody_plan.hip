// nbody_plan.hip — which decomposition, which block shape: host logic only (no launch, no allocation), visible without a device
// through nbody_plan*, and through nbody_ctx_*_info for a context's current knobs.
#include "nbody_ctx.hip.h"

#include <cmath>

using namespace nbi;

#pragma GCC visibility push(hidden)
namespace nbi {

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

// Estimated time (shader cycles) of one launch of `tasks` equal block-pair tasks of shape (W waves, bpl bodies per lane) plus
// the cost of summing `slab_bytes` of partial sums afterwards. The kernel is VALU-bound with two or more waves on a SIMD, so a
// SIMD's time is the number of wave-tasks it hosts times the time of one alone: full rounds put `wps` waves on every SIMD
// (wps from the kernel's VGPR allocation), the last partial round ceil(rest * W / SIMDs). One wave-task = B steps of
// (41.33 * bpl + 22.6) cycles — 8 bpl/2 packed ops at 4.15, bpl v_rsq_f32 at 8.13, 10 per-step instructions at 2.26
// (tools/valu_mb.hip). Checked against tools/smalln_probe.py sweeps from 32768 to 1048576 bodies
// (profiles/r02_shape_probe_{mid,large}.jsonl): it ranks the shapes as measured at every size.
// Waves per SIMD the register allocation of force_sym<SymPacked<bpl>> / force_sym_run allows. This is a COMPILER OUTPUT
// (196 / 164 / 92 / 60 VGPRs with ROCm 7.2) that the cost estimates below rely on: tests/test_build_resources.py compiles
// nbody_step.hip with -Rpass-analysis=kernel-resource-usage and checks every shipped instantiation against this table.
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

bool sym_resolve(const nbody_ctx* c, int n, SymShape* out, bool ignore_cap)
{
    const size_t cap = ignore_cap ? ~(size_t)0 : c->ws_cap;   // (the ticket kernel keeps no slabs: same shapes, no footprint)
    int pick = -1;
    double best = 0.0;
    int rw, rb;
    fp32_shape_request(c, &rw, &rb);
    for (int k = 0; k < kSymCands; ++k) {
        if ((rw && kSymCand[k][0] != rw) || (rb && kSymCand[k][1] != rb)) continue;
        const long B = 64L * kSymCand[k][0] * kSymCand[k][1];
        const long nb = (n + B - 1) / B;
        if (nb < 2 && pick >= 0) continue;
        if (nb > kSymMaxSlabs || (size_t)nb * (size_t)n * sizeof(float4) > cap) continue;
        const double cost = sym_cost(kSymCand[k][0], kSymCand[k][1], nb * (nb + 1) / 2, ignore_cap ? 0.0 : (double)nb * n * sizeof(float4), c->num_cu);
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
    if ((size_t)y.nb * (size_t)n * sizeof(float4) > cap) return false;
    *out = y;
    return true;
}

// How many accumulation lanes the in-place kernel gets for n bodies in nb blocks: 8 where they fit the workspace cap (8 x 16 n bytes: a
// few per cent of the slab kernel's nb x 16 n), fewer under a tight cap, ONE — the acceleration array itself, no workspace — as the last
// resort; never more lanes than half the blocks (every lane of every block must receive a contribution).
int ticket_lanes(const nbody_ctx* c, int n, int nb)
{
    int lanes = nbk::kTicketMaxLanes;
    while (lanes > 1 && ((size_t)lanes * (size_t)n * sizeof(float4) > c->ws_cap || lanes * 2 > nb)) lanes /= 2;
    return lanes;
}

// Does a square problem of n bodies go to the block-pair kernel with the sums added IN PLACE (nbk::force_sym_ticket)? Only shapes built
// for it: four waves x ten bodies per lane (the default large-N block) and one wave x ten (finer tasks for mid sizes).
bool ticket_wanted(const nbody_ctx* c, int n, SymShape* out)
{
    if (c->inplace_sums == 0) return false;
    if (!(c->kernel == NBODY_KERNEL_SYMMETRIC || (c->kernel == NBODY_KERNEL_FAST && n >= kSymMinAuto))) return false;
    if (c->sym_runs == 2) return false;                          // balanced runs were asked for explicitly
    SymShape y{};
    if (c->inplace_sums < 0) {
        // auto: ONLY where the slab kernel would run but for its workspace — it resolves with the cap ignored and not with it. Anything
        // else that keeps block pairs away (a single block, a shape request nothing matches) keeps meaning what it meant: one-sided.
        if (sym_resolve(c, n, &y)) return false;
        if (!sym_resolve(c, n, &y, true)) return false;
    }
    // Two shapes are built: four waves x ten bodies per lane (2560-body blocks: a task of about a millisecond, its two additions a per
    // cent of it) where that still makes a few thousand tasks, one wave x ten (640-body blocks) below. (The slab kernel's cost estimate
    // does not know the per-task price of the additions and would always pick the finer shape.)
    nbody_ctx tmp = *c;
    tmp.sym_waves = n >= kRunsMaxAuto ? 4 : 1;
    tmp.sym_bpl = 10;
    {
        int rw, rb;
        fp32_shape_request(c, &rw, &rb);
        if ((rw == 4 || rw == 1) && (rb == 0 || rb == 10)) tmp.sym_waves = rw;   // an explicit request for a built shape is honoured
    }
    if (!sym_resolve(&tmp, n, &y, true)) return false;
    *out = y;
    return true;
}

// Block shape for the symmetric evaluation of TWO disjoint ranges (ni x nj bodies): the cheapest by the same estimate
// (padding of both sides to whole blocks included through the task count).
// `hopeless` (optional): set when no candidate could ever apply to ni targets however short the source run is made (a shape
// request that matches nothing built, or too many target blocks) — as opposed to a workspace over the cap, which fewer sources cure.
bool sym_resolve_cross(const nbody_ctx* c, int ni, int nj, SymShape* out, int* nbj, bool* hopeless)
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

}  // namespace nbi
#pragma GCC visibility pop

extern "C" {

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
    if (n_targets == n_sources && ticket_wanted(c, n_targets, &y) && ticket_lanes(c, n_targets, y.nb) > 1) {   // block sums added in place
        if (jsplit) *jsplit = ticket_lanes(c, n_targets, y.nb);
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
    if (!bal_wanted(c, n, &by) && c->inplace_sums == 1 && ticket_wanted(c, n, &y)) {
        if (symmetric) *symmetric = 4;  // block pairs, partial sums added in place (tickets): no slabs, a few accumulation lanes
        if (block_bodies) *block_bodies = y.block;
        if (slabs) *slabs = ticket_lanes(c, n, y.nb) > 1 ? ticket_lanes(c, n, y.nb) : 0;
        if (workgroups) *workgroups = y.grid;
        if (evaluated_pairs) *evaluated_pairs = ((double)y.nb * (y.nb - 1) / 2 + y.nb) * (double)y.block * (double)y.block;
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
    if (ticket_wanted(c, n, &y)) {   // (auto: the slab workspace does not fit the cap)
        if (symmetric) *symmetric = 4;
        if (block_bodies) *block_bodies = y.block;
        if (slabs) *slabs = ticket_lanes(c, n, y.nb) > 1 ? ticket_lanes(c, n, y.nb) : 0;
        if (workgroups) *workgroups = y.grid;
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

// Device-free view of the ticket kernel's task list (host tests): which block pair task `task` of an nb-block launch evaluates and
// which contribution numbers its two sums carry (the closed forms the kernel uses; nbk::force_sym_ticket).
int nbody_plan_ticket_task(int nb, int task, int* out_i, int* out_j, int* out_seq_i, int* out_seq_j)
{
    if (nb < 2 || task < 0 || task >= nb * (nb - 1) / 2 + nb) return fail(NBODY_ERR_INVALID, "bad ticket task: nb=%d task=%d", nb, task);
    const int npair = nb * (nb - 1) / 2;
    int I, J, si, sj = -1;
    if (task >= npair) {
        I = J = task - npair;
        si = nb - 1;
    } else {
        int r = 0;
        while (r < nb - 2 && (int)(((long)(r + 1) * (2L * nb - (r + 1) - 1)) / 2) <= task) ++r;   // row of the triangular list = anti-diagonal d - 1
        const int c0 = task - (int)(((long)r * (2L * nb - r - 1)) / 2), d = r + 1;
        I = c0;
        J = c0 + d;
        si = (d - 1) + (d < I ? d : I);
        sj = (d - 1 < nb - 1 - J ? d - 1 : nb - 1 - J) + (d - 1);
    }
    if (out_i) *out_i = I;
    if (out_j) *out_j = J;
    if (out_seq_i) *out_seq_i = si;
    if (out_seq_j) *out_seq_j = sj;
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

}  // extern "C"

// nbody_step.hip — the launchers of the kernels in nbody_kernels.hip.h and the stepping entry points of include/nbody.h:
// nbody_accel_*, nbody_integrate_range, nbody_step, nbody_simulate (the reference's simulate(), TestProject/kernel.cu:628-645),
// nbody_simulate_host_legacy, nbody_step_f64. The one translation unit of the single-GPU path that carries device code.
#include "nbody_ctx.hip.h"
#include "nbody_kernels.hip.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace nbi;

namespace {

using P2 = nbk::MathPacked<2>;
using P4 = nbk::MathPacked<4>;
using S1 = nbk::MathScalar<1>;

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

// Equal-mass path: scans the bodies of the coming launch(es) on the stream (x[i0 .. i0+ni) and, when nj > 0, the run of nj bodies
// from j0, wrapping at `wrap`) and hands out the verdict slot and this scan's generation. No host round trip: the force kernel
// reads the verdict itself. *q stays nullptr when the path is switched off (or the verdict slots cannot be allocated).

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

// block pairs with the sums added in place: the whole task list in ONE launch (a ticket waits only for EARLIER tasks of the same launch)
int launch_ticket(nbody_ctx* c, const SymShape& y, const nbk::SymParams& p, bool timed)
{
    if (timed) if (int rc = time_mark(c)) return rc;
    if (y.bpl == 10 && y.waves == 4) nbk::force_sym_ticket<nbk::SymPacked<10>, 4><<<y.grid, 256, 0, c->stream>>>(p);
    else if (y.bpl == 10 && y.waves == 1) nbk::force_sym_ticket<nbk::SymPacked<10>, 1><<<y.grid, 64, 0, c->stream>>>(p);
    else return fail(NBODY_ERR_CONFIG, "no in-place block-pair kernel for waves=%d bodies_per_lane=%d", y.waves, y.bpl);
    HIP_TRY(hipGetLastError());
    if (timed) return time_mark(c);
    return NBODY_OK;
}

void ticket_params(nbody_ctx* c, nbk::SymParams* sp, const float4* x, int i0, int n, const SymShape& y, float4* acc, int lanes)
{
    sym_square_params(sp, x, i0, n, y, nullptr, c->eps2);
    sp->acc = acc;
    sp->tickets = c->tickets;
    sp->err = c->terr_dev;
    sp->acc_lanes = lanes;
    sp->acc_stride = n;
    sp->acc_timeout_us = nbk::kTicketTimeoutUs;
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

}  // namespace

#pragma GCC visibility push(hidden)
namespace nbi {

int launch_clock_stamp(nbody_ctx* c, nbk::ClockDelta* d_out)
{
    if (!d_out) nbk::clock_begin<<<nbk::kClockBeginWgs, 64, 0, c->stream>>>(c->cscratch);
    else nbk::clock_end<<<nbk::kClockEndWgs, 64, 0, c->stream>>>(c->cscratch, d_out);
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

void load_device_code()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&nbk::copy_bodies)) != hipSuccess) (void)hipGetLastError();
}

}  // namespace nbi
#pragma GCC visibility pop

namespace {

// targets [i0,i1) x sources j0 .. j0+count-1 (indices taken modulo `wrap` when wrap > 0)
int accel_impl(nbody_ctx* c, const nbody_float4* d_bodies, nbody_float4* d_acc_out, int i0, int i1, int j0, int j1,
               int wrap, int accumulate)
{
    const int nt = i1 - i0;
    ON_DEVICE(c);
    // A square block with the block sums added IN PLACE (nbk::force_sym_ticket) into 2 ... 8 accumulation lanes in the workspace, then one
    // slab sum over the lanes: where the slab workspace of the symmetric kernels does not fit the cap, or where in-place sums were asked
    // for. (One lane would be the output array itself, which cannot ALSO hold what `accumulate` is to add to: a cap below two lanes
    // leaves *done false and the one-sided kernel runs.)
    auto in_place = [&](bool* done) -> int {
        *done = false;
        SymShape ty{};
        if (!(i0 == j0 && i1 == j1 && !wrap && ticket_wanted(c, nt, &ty))) return NBODY_OK;
        int lanes = ticket_lanes(c, nt, ty.nb);
        while (lanes > 1 && ensure_slabs(c, (size_t)lanes * nt * sizeof(float4)) != NBODY_OK) lanes /= 2;
        if (lanes <= 1) return NBODY_OK;
        if (int rc = ensure_tickets(c)) return rc;
        if (int rc = ticket_error(c)) return rc;
        nbk::SymParams sp{};
        ticket_params(c, &sp, reinterpret_cast<const float4*>(d_bodies), i0, nt, ty, static_cast<float4*>(c->slabs), lanes);
        if (int rc = eq_scan(c, 0, sp.x, i0, nt, 0, 0, 0, &sp.eqm, &sp.eq_gen)) return rc;
        if (int rc = launch_ticket(c, ty, sp, true)) return rc;
        nbk::ReduceParams r{};
        r.out = reinterpret_cast<float4*>(d_acc_out);
        r.slabs = static_cast<const float4*>(c->slabs);
        r.nslab = lanes;
        r.slab_stride = nt;
        r.n = nt;
        r.accumulate = accumulate ? 1 : 0;
        nbk::reduce_slabs<<<(nt + nbk::kWG - 1) / nbk::kWG, nbk::kWG, 0, c->stream>>>(r);
        HIP_TRY(hipGetLastError());
        *done = true;
        return NBODY_OK;
    };
    if (c->inplace_sums == 1) {   // asked for: ahead of unit runs / block pairs with slabs (balanced runs keep their sizes)
        BalShape by0{};
        if (!(i0 == j0 && i1 == j1 && !wrap && bal_wanted(c, nt, &by0))) {
            bool done = false;
            if (int rc = in_place(&done)) return rc;
            if (done) return NBODY_OK;
        }
    }
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
    {
        bool done = false;
        if (int rc = in_place(&done)) return rc;
        if (done) return NBODY_OK;
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

extern "C" {

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
    if (int rc = ticket_error(c)) return rc;
    SymShape y{};
    RunShape ry{};
    BalShape by{};
    bool bal = false, runs = false, sym = false, tick = false;
    for (int attempt = 0;; ++attempt) {   // a symmetric footprint that cannot be allocated lowers the cap: resolve again
        bal = bal_wanted(c, n, &by);
        if (bal) {
            const int rc = ensure_inbox(c, by);
            if (rc == NBODY_OK) break;
            if (rc != NBODY_ERR_NOMEM || attempt >= 16) return rc;
            continue;
        }
        tick = c->inplace_sums == 1 && ticket_wanted(c, n, &y);   // asked for: instead of unit runs / block pairs with slabs
        if (tick) break;
        runs = run_wanted(c, n, &ry);
        sym = !runs && sym_wanted(c, n, &y);
        if (!runs && !sym) {
            tick = ticket_wanted(c, n, &y);   // no slab workspace to be had for the symmetric kernels: the sums go in place
            break;
        }
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
    } else if (tick) {
        if (int rc = ensure_tickets(c)) return rc;
        int lanes = ticket_lanes(c, n, y.nb);
        while (lanes > 1 && ensure_slabs(c, (size_t)lanes * n * sizeof(float4)) != NBODY_OK) lanes /= 2;   // (an allocation that fails: fewer lanes)
        if (lanes > 1) {   // the lanes live in the workspace; the integrate adds them in index order
            ticket_params(c, &sp, reinterpret_cast<const float4*>(d_bodies), 0, n, y, static_cast<float4*>(c->slabs), lanes);
            q.slabs = static_cast<const float4*>(c->slabs);
            q.nslab = lanes;
            q.slab_stride = n;
        } else {           // one lane: the sums land in the acceleration array itself, no workspace at all
            ticket_params(c, &sp, reinterpret_cast<const float4*>(d_bodies), 0, n, y, q.a, 1);
            q.slabs = nullptr;
            q.nslab = 0;
            q.slab_stride = 0;
        }
        if (c->ticket_test_stall) {   // TEST hook, one shot: block 0's first ticket is held by nobody; a waiter gives up after 2 ms
            c->ticket_test_stall = false;
            HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->tickets), 0x7fffffff, 1, c->stream));
            sp.acc_timeout_us = 2000;
        }
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
    const bool eq_path = (bal || runs || sym || tick) && !graphable;
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
        else if (tick) rc = launch_ticket(c, y, sp, timed);
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
        const nbody_ctx::GraphKey key{d_bodies, q.a, q.v, c->slabs, n, bal ? by.y.bpl : runs ? ry.bpl : (sym || tick) ? y.bpl : s.bpl,
                                      bal ? -2 : runs ? -1 : tick ? -3 - y.waves : sym ? y.waves : s.tile, bal ? by.y.L : runs ? ry.layout.L : (sym || tick) ? y.nb : s.jsplit,
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

}  // extern "C"

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

// Are all three arrays ORDINARY device allocations — plain hipMalloc: device memory, not managed, no allocation flags (fine-grained,
// uncached and signal memory from hipExtMallocWithFlags carry flags and take the stream synchronisation, like host-mapped and managed
// arrays)? Looked up on every call that asks — a pointer value can come back as another kind of memory after a free — which costs
// well under a microsecond on calls of 60 us and more (N > 8192). The promise of nbody.h / INTEGRATION.md is made for the three kinds
// of memory the tests pin (hipMalloc, hipHostMalloc, hipMallocManaged); anything this test cannot classify is synchronised the slow way.
bool arrays_are_device_memory(const void* x, const void* a, const void* v)
{
    for (const void* p : {x, a, v}) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();   // not known to the runtime (pageable host memory under HMM, ...): not device memory
            return false;
        }
        if (at.type != hipMemoryTypeDevice || at.isManaged || at.allocationFlags != 0) return false;
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

extern "C" {

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
    // (either way the step is complete here; a ticket wait that timed out inside it — the in-place block-pair kernel, never on a healthy
    //  run — is this call's error, as a failed launch would be)
    if (wait_host_word(c)) return ticket_error(c);
    ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ticket_error(c);
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

}  // extern "C"

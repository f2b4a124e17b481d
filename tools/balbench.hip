// tools/balbench.hip — developer bench for the BALANCED-run symmetric kernel (nbk::force_sym_bal + nbk::bal_reduce) at small and
// mid N against the kernels the library ships for those sizes: correctness against the one-sided kernel and an fp64 CPU sum on
// sampled targets, force-alone / force+reduce times, and the time of a queued step loop (force + reduce-and-integrate).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I n-bodysimulation_amd/csrc tools/balbench.hip -o build/balbench
//   build/balbench N [steps]
// Not part of the product; results feed DESIGN.md and profiles/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "nbody_experiments.hip.h"   // the product header + the measured alternatives

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                              \
        }                                                                                         \
    } while (0)

static uint64_t g_s = 12345;
static double u01()
{
    g_s += 0x9E3779B97F4A7C15ull;
    uint64_t z = g_s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// us per call of f, queued `steps` times between two events, best of `reps`
static double us_per_call(const std::function<void()>& f, int steps, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int k = 0; k < 8; ++k) f();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < steps; ++k) f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e3 / steps);
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

template <class M, int WV>
static void launch_bal_wv(const nbk::BalParams& p) { nbk::force_sym_bal<M, WV><<<(p.y.nworkers + WV - 1) / WV, 64 * WV>>>(p); }
static int g_bal_wps3 = 0;
template <class M>
static void launch_bal(const nbk::BalParams& p)
{
    if (g_bal_wps3 && p.y.wv == 4) {   // the build for exactly three waves per SIMD
        nbk::force_sym_bal_wps<M, 4, 3><<<(p.y.nworkers + 3) / 4, 256>>>(p);
        return;
    }
    switch (p.y.wv) {
        case 1: launch_bal_wv<M, 1>(p); break;
        case 4: launch_bal_wv<M, 4>(p); break;
        default: launch_bal_wv<M, 8>(p); break;
    }
}

static void launch_reduce(const nbk::BalReduceParams& r, int P)
{
    switch (P) {
        case 1: nbk::bal_reduce<1><<<r.y.ncht, 64>>>(r); break;
        case 2: nbk::bal_reduce<2><<<r.y.ncht, 128>>>(r); break;
        case 4: nbk::bal_reduce<4><<<r.y.ncht, 256>>>(r); break;
        case 8: nbk::bal_reduce<8><<<r.y.ncht, 512>>>(r); break;
        default: nbk::bal_reduce<16><<<r.y.ncht, 1024>>>(r); break;
    }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    const int steps = argc > 2 ? atoi(argv[2]) : std::max(50, std::min(2000, (int)(4e11 / ((double)n * n))));
    const float eps2 = 0.002f;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int simds = 4 * prop.multiProcessorCount;

    std::vector<float4> hx(n);
    for (int i = 0; i < n; ++i) {
        double r;
        do { r = 1.0 / std::sqrt(std::pow(u01() * 0.999 + 1e-9, -2.0 / 3.0) - 1.0); } while (r > 50.0);
        const double ct = 2.0 * u01() - 1.0, st = std::sqrt(1.0 - ct * ct), ph = 6.283185307179586 * u01();
        hx[i] = make_float4((float)(r * st * std::cos(ph)), (float)(r * st * std::sin(ph)), (float)(r * ct), 1.0f / n);
    }
    float4 *dx, *dv, *da, *da_ref, *slabs;
    CK(hipMalloc(&dx, (size_t)n * 16));
    CK(hipMalloc(&dv, (size_t)n * 16));
    CK(hipMalloc(&da, (size_t)n * 16));
    CK(hipMalloc(&da_ref, (size_t)n * 16));
    CK(hipMemcpy(dx, hx.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    CK(hipMemset(dv, 0, (size_t)n * 16));
    CK(hipMalloc(&slabs, (size_t)64 * n * 16));

    // the shipped one-sided path at this size (what nbody_step launches below 12288 bodies): force_lds + integrate over jsplit slabs
    int bx, js, tile, obpl;
    if (n <= 10240) { obpl = 1; tile = 256; } else { obpl = 2; tile = 512; }
    bx = (n + 256 * obpl - 1) / (256 * obpl);
    js = 1;
    while (bx * js < 16 * prop.multiProcessorCount && js < 64) js *= 2;
    js = std::min(js, std::min(32, (n + tile - 1) / tile));
    nbk::ForceParams fp{};
    fp.x = dx; fp.out = slabs; fp.i0 = 0; fp.i1 = n; fp.j0 = 0; fp.j1 = n; fp.slab_stride = n; fp.accumulate = 0; fp.eps2 = eps2; fp.wrap = 0;
    auto one_sided_force = [&] {
        if (obpl == 1) nbk::force_lds<nbk::MathScalar<1>, 256, 8, 1><<<dim3(bx, js), 256>>>(fp);
        else nbk::force_lds<nbk::MathPacked<2>, 512, 8, 1><<<dim3(bx, js), 256>>>(fp);
    };
    nbk::IntegrateParams ip{};
    ip.x = dx; ip.v = dv; ip.a = da_ref; ip.slabs = slabs; ip.nslab = js; ip.slab_stride = n; ip.n = n; ip.dt = 0.0f;   // dt = 0: positions stay put
    auto one_sided_step = [&] {
        one_sided_force();
        nbk::integrate<<<(n + 255) / 256, 256>>>(ip);
    };
    one_sided_step();
    CK(hipDeviceSynchronize());
    std::vector<float4> a_ref(n), a_bal(n);
    CK(hipMemcpy(a_ref.data(), da_ref, (size_t)n * 16, hipMemcpyDeviceToHost));

    const int nsamp = 64;
    std::vector<int> samp(nsamp);
    std::vector<double> truth(3 * nsamp);
    double amax = 0;
    for (int s = 0; s < nsamp; ++s) {
        const int i = (int)((long)s * (n - 1) / (nsamp - 1));
        samp[s] = i;
        double ax = 0, ay = 0, az = 0;
        for (int j = 0; j < n; ++j) {
            const double rx = (double)hx[j].x - hx[i].x, ry = (double)hx[j].y - hx[i].y, rz = (double)hx[j].z - hx[i].z;
            const double d = rx * rx + ry * ry + rz * rz + (double)eps2;
            const double f = hx[j].w / (d * std::sqrt(d));
            ax += rx * f; ay += ry * f; az += rz * f;
        }
        truth[3 * s] = ax; truth[3 * s + 1] = ay; truth[3 * s + 2] = az;
        amax = std::max(amax, std::sqrt(ax * ax + ay * ay + az * az));
    }
    auto err_vs_truth = [&](const std::vector<float4>& a) {
        double e = 0;
        for (int s = 0; s < nsamp; ++s) {
            const float4 q = a[samp[s]];
            e = std::max({e, std::fabs(q.x - truth[3 * s]), std::fabs(q.y - truth[3 * s + 1]), std::fabs(q.z - truth[3 * s + 2])});
        }
        return e / amax;
    };
    const double pairs = (double)n * n;
    const double t_of = us_per_call(one_sided_force, steps), t_os = us_per_call(one_sided_step, steps);
    printf("N=%d steps=%d | shipped one-sided (bpl %d tile %d, %d x %d workgroups): force %.2f us, step %.2f us = %.3e pairs/s, vs fp64 truth %.2g\n", n,
           steps, obpl, tile, bx, js, t_of, t_os, pairs / t_os * 1e6, err_vs_truth(a_ref));

    double scale = 0;
    for (int i = 0; i < n; ++i) scale = std::max({scale, (double)std::fabs(a_ref[i].x), (double)std::fabs(a_ref[i].y), (double)std::fabs(a_ref[i].z)});

    // ---- the fused small-N step: one launch per step, positions alternate between two arrays
    {
        float4* dx2;
        CK(hipMalloc(&dx2, (size_t)n * 16));
        CK(hipMemcpy(dx2, dx, (size_t)n * 16, hipMemcpyDeviceToDevice));
        struct Var { const char* name; int T, WV, TILE; std::function<void(const nbk::FusedParams&)> launch; };
        std::vector<Var> vars;
#define FUSED(T_, WV_, TILE_) vars.push_back({"fused T" #T_ " wv" #WV_ " tile" #TILE_, T_, WV_, TILE_, [&](const nbk::FusedParams& q) { \
            nbk::step_fused<T_, WV_, TILE_><<<(n + T_ * WV_ - 1) / (T_ * WV_), 64 * WV_>>>(q); }})
        if (getenv("BALBENCH_FUSED_UNROLL")) {
#define FUSEDU(T_, WV_, TILE_, U_) vars.push_back({"fused T" #T_ " wv" #WV_ " tile" #TILE_ " unroll" #U_, T_, WV_, TILE_, [&](const nbk::FusedParams& q) { \
            nbk::step_fused<T_, WV_, TILE_, U_><<<(n + T_ * WV_ - 1) / (T_ * WV_), 64 * WV_>>>(q); }})
            FUSEDU(2, 16, 2048, 2); FUSEDU(2, 16, 2048, 4); FUSEDU(2, 16, 2048, 8); FUSEDU(2, 16, 2048, 16); FUSEDU(2, 16, 1024, 4); FUSEDU(2, 16, 1024, 8); FUSEDU(2, 16, 2048, 4);
        } else if (getenv("BALBENCH_FUSED_OCC")) {   // sizes above 8192: several workgroups per CU, registers capped so that they are co-resident
#define FUSEDM(T_, WV_, TILE_, U_, M_) vars.push_back({"fused T" #T_ " wv" #WV_ " tile" #TILE_ " unroll" #U_ " minw" #M_, T_, WV_, TILE_, [&](const nbk::FusedParams& q) { \
            nbk::step_fused<T_, WV_, TILE_, U_, M_><<<(n + T_ * WV_ - 1) / (T_ * WV_), 64 * WV_>>>(q); }})
            FUSEDM(2, 10, 1280, 4, 5); FUSEDM(2, 10, 1280, 8, 5); FUSEDM(2, 10, 640, 4, 5); FUSEDM(2, 10, 2560, 4, 5); FUSEDM(2, 10, 1920, 4, 5);
            FUSEDM(2, 8, 1024, 4, 6); FUSEDM(2, 12, 1536, 4, 6); FUSEDM(2, 12, 768, 4, 6); FUSEDM(2, 8, 2048, 4, 6); FUSEDM(2, 16, 2048, 4, 8); FUSEDM(2, 14, 1792, 4, 7);
        } else if (getenv("BALBENCH_FUSED_TWO_PER_CU")) {   // sizes above 8192: two workgroups per CU
            FUSED(2, 10, 1280); FUSED(2, 10, 2560); FUSED(2, 12, 1536); FUSED(2, 12, 2304); FUSED(2, 16, 2048); FUSED(2, 14, 1792); FUSED(2, 8, 2048); FUSED(4, 10, 2560);
            FUSED(4, 6, 2304); FUSED(4, 8, 2048);
        } else {
        FUSED(2, 4, 1024); FUSED(2, 4, 2048); FUSED(2, 8, 2048); FUSED(2, 8, 4096); FUSED(2, 6, 1536); FUSED(2, 6, 3072); FUSED(2, 12, 3072); FUSED(2, 16, 2048);
        FUSED(2, 16, 4096); FUSED(4, 4, 1024); FUSED(4, 4, 2048); FUSED(4, 8, 2048); FUSED(4, 8, 4096); FUSED(4, 6, 3072); FUSED(2, 10, 2560); FUSED(2, 5, 2560);
        }
        for (auto& v : vars) {
            nbk::FusedParams q{};
            q.xin = dx; q.xout = dx2; q.v = dv; q.a = da; q.n = n; q.dt = 0.0f; q.eps2 = eps2;
            CK(hipMemset(da, 0xff, (size_t)n * 16));
            v.launch(q);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(a_bal.data(), da, (size_t)n * 16, hipMemcpyDeviceToHost));
            double dmax = 0;
            long bad = 0;
            for (int i = 0; i < n; ++i) {
                const double e = std::max({std::fabs((double)a_ref[i].x - a_bal[i].x), std::fabs((double)a_ref[i].y - a_bal[i].y),
                                           std::fabs((double)a_ref[i].z - a_bal[i].z)});
                if (!(e <= 1e30)) ++bad; else dmax = std::max(dmax, e);
            }
            int flip = 0;
            auto step = [&] {   // ping-pong, as nbody_step does
                nbk::FusedParams r = q;
                r.xin = flip ? dx2 : dx; r.xout = flip ? dx : dx2;
                flip ^= 1;
                v.launch(r);
            };
            const double t = us_per_call(step, steps);
            printf("%-26s: %d waves | %.2f us/step = %.3e pairs/s (%.1f%% of 157.3 TF) | vs one-sided %.2g of max|a|, nonfinite %ld, vs truth %.2g\n", v.name,
                   (n + v.T - 1) / v.T, t, pairs / t * 1e6, 20 * pairs / t * 1e6 / 157.3e12 * 100, dmax / scale, bad, err_vs_truth(a_bal));
            fflush(stdout);
        }
        CK(hipFree(dx2));
    }
    if (getenv("BALBENCH_FUSED_ONLY")) return 0;

    const bool only3 = getenv("BALBENCH_WPS3") != nullptr;
    for (int bpl : {2, 4, 8, 10}) {
        for (int wps : {2, 3, 4}) {
          for (int wv : {1, 4, 8}) {
            if (only3) {   // bodies per lane 8 at three waves per SIMD (forced register budget) against two
                if (bpl != 8 || wv != 4 || wps > 3) continue;
                g_bal_wps3 = wps == 3;
            } else
            if ((bpl == 10 && wps > 2) || (bpl == 8 && wps > 2) || (bpl == 2 && wps < 4) || (wps == 3 && wv == 1)) continue;
            nbk::BalLayout y{};
            if (!nbk::bal_plan(n, bpl, simds * wps, wv, &y)) continue;
            float4* inbox;
            const size_t ib = (size_t)y.ncht * y.smax * 64 * 16;
            CK(hipMalloc(&inbox, ib));
            CK(hipMemset(inbox, 0, ib));      // the one-time clear: records of pieces that do not exist stay zero
            nbk::BalParams bp{};
            bp.x = dx; bp.inbox = inbox; bp.n = n; bp.y = y; bp.eps2 = eps2;
            auto force = [&] {
                switch (bpl) {
                    case 2: launch_bal<nbk::SymPacked<2>>(bp); break;
                    case 4: launch_bal<nbk::SymPacked<4>>(bp); break;
                    case 8: launch_bal<nbk::SymPacked<8>>(bp); break;
                    default: launch_bal<nbk::SymPacked<10>>(bp); break;
                }
            };
            nbk::BalReduceParams rp{};
            rp.inbox = inbox; rp.y = y; rp.n = n; rp.x = dx; rp.v = dv; rp.a = da; rp.dt = 0.0f;
            rp.mode = 1; rp.accumulate = 0;
            // correctness once
            force();
            launch_reduce(rp, 8);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(a_bal.data(), da, (size_t)n * 16, hipMemcpyDeviceToHost));
            double dmax = 0;
            long bad = 0;
            for (int i = 0; i < n; ++i) {
                const double e = std::max({std::fabs((double)a_ref[i].x - a_bal[i].x), std::fabs((double)a_ref[i].y - a_bal[i].y),
                                           std::fabs((double)a_ref[i].z - a_bal[i].z)});
                if (!(e <= 1e30)) ++bad; else dmax = std::max(dmax, e);
            }
            const double t_f = us_per_call(force, steps);
            printf("bal bpl %2d wps %d wv %d: workers %5d L %5d pmax %d smax %3d inbox %.1f MB | force %.2f us |", bpl, wps, wv, y.nworkers, y.L, y.pmax,
                   y.smax, ib / 1e6, t_f);
            rp.mode = 0;   // integrate (dt = 0)
            double best = 1e30;
            int bestP = 0;
            for (int P : {4, 8, 16}) {
                auto step = [&] { force(); launch_reduce(rp, P); };
                const double t = us_per_call(step, steps);
                printf(" P%d %.2f", P, t);
                if (t < best) { best = t; bestP = P; }
            }
            printf(" us/step | best P%d = %.3e pairs/s (%.1f%% of 157.3 TF) | vs one-sided %.2g of max|a|, nonfinite %ld, vs truth %.2g\n", bestP,
                   pairs / best * 1e6, 20 * pairs / best * 1e6 / 157.3e12 * 100, dmax / scale, bad, err_vs_truth(a_bal));
            fflush(stdout);
            CK(hipFree(inbox));
          }
        }
    }
    return 0;
}

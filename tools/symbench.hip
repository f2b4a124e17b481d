// tools/symbench.hip — developer bench for the symmetric (each unordered pair once) force kernel
// against the shipped one-sided LDS kernel: same bodies, same process, interleaved timing, and a
// check of the accelerations of the two against each other and against an fp64 CPU sum on sampled
// targets.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I n-bodysimulation_amd/csrc tools/symbench.hip -o build/symbench
//   build/symbench [N] [reps]
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

// what each lane sees through row_ror:S (prints the lane mapping once, for the record)
__global__ void dpp_probe(int* out)
{
    const int lane = threadIdx.x;
    out[lane] = (int)nbk::ror<1>((float)lane);
    out[64 + lane] = (int)nbk::ror<15>((float)lane);
    out[128 + lane] = (int)nbk::next_row((float)lane, ((lane + 16) & 63) << 2);
}

static float median_ms(const std::function<void()>& f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ts[ts.size() / 2];
}

// ---- measured alternative (round 3): ROW-ACCUMULATING tasks --------------------------------------------------------------------
// One workgroup keeps its I-block in registers across R consecutive J blocks: the I-side sums are written once per R block pairs
// (slab[J0], J0 = the group's first J block), the J-side sums once per J block as before. Partial-sum bytes per launch: about
// (1 + 1/R)/2 of the shipped kernel's. Task list: row I has ceil((nb-1-I)/R) groups, then the nb diagonal blocks; row_of[] holds
// the first task of every row (host-made). Same pair arithmetic and rotation scheme as nbk::force_sym (rect == 0 only).
struct RowsParams {
    const float4* x;
    float4* slabs;
    const int* row_first_task;  // nb entries + total
    int n, nb, stride, R, npair_tasks;
    float eps2;
};

template <class M, int W>
__global__ void __launch_bounds__(64 * W, 1) force_sym_rows(const RowsParams p)
{
    using namespace nbk;
    constexpr int BPL = M::BPL;
    constexpr int B = 64 * W * BPL;
    constexpr int NCH = B / 64;
    __shared__ float4 sh[B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int task = blockIdx.x;
    int I, J0, J1;
    const bool diag = task >= p.npair_tasks;
    if (diag) {
        I = J0 = task - p.npair_tasks;
        J1 = J0 + 1;
    } else {
        int lo = 0, hi = p.nb - 2;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (p.row_first_task[mid] <= task) lo = mid; else hi = mid - 1;
        }
        I = lo;
        J0 = I + 1 + (task - p.row_first_task[I]) * p.R;
        J1 = J0 + p.R < p.nb ? J0 + p.R : p.nb;
    }
    M t;
    t.set_eps2(p.eps2);
    const int ibase = I * B + w * (64 * BPL) + lane;
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        const int i = ibase + k * 64;
        t.set(k, i < p.n ? p.x[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    }
    const int rot = ((lane + 16) & 63) << 2;
    for (int J = J0; J < J1; ++J) {
        const int jbase = J * B + lane;
        auto fetch = [&](int c) {
            const int j = jbase + c * 64;
            return j < p.n ? p.x[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        };
        if (!diag) {
            if (J != J0) __syncthreads();   // the previous block's J-side sums have been stored
#pragma unroll
            for (int r = 0; r < BPL; ++r) sh[r * (64 * W) + tid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            __syncthreads();
        }
        int c = w * BPL;
        float4 nxt = fetch(c);
        for (int q = 0; q < NCH; ++q) {
            float4 bj = nxt;
            const int cn = (c + 1 == NCH) ? 0 : c + 1;
            if (q + 1 < NCH) nxt = fetch(cn);
            if (diag) {
                float4 aj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                for (int ph = 0; ph < 4; ++ph) {
                    sym_row_pass<false>(t, bj, aj);
                    bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot); bj.w = next_row(bj.w, rot);
                }
            } else {
                float4 aj = sh[c * 64 + lane];
                for (int ph = 0; ph < 4; ++ph) {
                    sym_row_pass<true>(t, bj, aj);
                    bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot); bj.w = next_row(bj.w, rot);
                    aj.x = next_row(aj.x, rot); aj.y = next_row(aj.y, rot); aj.z = next_row(aj.z, rot);
                }
                sh[c * 64 + lane] = aj;
                __syncthreads();
            }
            c = cn;
        }
        if (!diag) {
            float4* const out_j = p.slabs + (size_t)I * p.stride;
            for (int e = tid; e < B; e += 64 * W) {
                const int j = J * B + e;
                if (j < p.n) { float4 a = sh[e]; a.w = 0.0f; out_j[j] = a; }
            }
        }
    }
    float4* const out_i = p.slabs + (size_t)J0 * p.stride;   // diagonal: J0 == I
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        const int i = ibase + k * 64;
        if (i < p.n) out_i[i] = t.acc(k);
    }
}

// body i of block I: slabs 0 .. I (J-side sums of the rows above, the diagonal), then every R-th slab from I + 1 (one per group)
__global__ void __launch_bounds__(256) reduce_rows(float4* out, const float4* slabs, int n, int nb, int B, int R)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int I = i / B;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 8
    for (int s = 0; s <= I; ++s) {
        const float4 q = slabs[(size_t)s * n + i];
        a.x += q.x; a.y += q.y; a.z += q.z;
    }
#pragma unroll 8
    for (int s = I + 1; s < nb; s += R) {
        const float4 q = slabs[(size_t)s * n + i];
        a.x += q.x; a.y += q.y; a.z += q.z;
    }
    out[i] = a;
}

struct SymVariant {
    std::string name;
    int B;
    std::function<void(const nbk::SymParams&, int grid)> launch;
    bool eq = false;   // run nbk::mass_scan first and hand its verdict to the kernel (the equal-mass path when the bodies are uniform)
};

template <class M, int W, int MINW = 1>
static SymVariant sym_variant(const char* name)
{
    return {name, 64 * W * M::BPL, [](const nbk::SymParams& p, int grid) { nbk::force_sym<M, W, MINW><<<grid, 64 * W>>>(p); }};
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 262144;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const bool plummer = argc > 3 ? atoi(argv[3]) != 0 : true;
    const float eps2 = 0.002f;

    {
        int* d;
        CK(hipMalloc(&d, 192 * sizeof(int)));
        dpp_probe<<<1, 64>>>(d);
        int h[192];
        CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        printf("row_ror:1  lane0..17 reads:");
        for (int l = 0; l < 18; ++l) printf(" %d", h[l]);
        printf("\nrow_ror:15 lane0..17 reads:");
        for (int l = 0; l < 18; ++l) printf(" %d", h[64 + l]);
        printf("\nnext_row   lane0,16,32,48 reads: %d %d %d %d\n", h[128], h[128 + 16], h[128 + 32], h[128 + 48]);
        CK(hipFree(d));
    }

    std::vector<float4> hx(n);
    for (int i = 0; i < n; ++i) {
        if (plummer) {
            // Plummer sphere a = 1, total mass 1
            double r;
            do { r = 1.0 / std::sqrt(std::pow(u01() * 0.999 + 1e-9, -2.0 / 3.0) - 1.0); } while (r > 50.0);
            const double ct = 2.0 * u01() - 1.0, st = std::sqrt(1.0 - ct * ct), ph = 6.283185307179586 * u01();
            hx[i] = make_float4((float)(r * st * std::cos(ph)), (float)(r * st * std::sin(ph)), (float)(r * ct), 1.0f / n);
        } else {
            hx[i] = make_float4((float)((2 * u01() - 1) * 1e5), (float)((2 * u01() - 1) * 1e5), (float)((2 * u01() - 1) * 1e5),
                                (float)(1e5 + u01() * (1e9 - 1e5)));
        }
    }
    float4 *dx, *da_ref, *da_sym, *slabs;
    CK(hipMalloc(&dx, (size_t)n * 16));
    CK(hipMalloc(&da_ref, (size_t)n * 16));
    CK(hipMalloc(&da_sym, (size_t)n * 16));
    CK(hipMemcpy(dx, hx.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    const int max_slabs = 1024;
    CK(hipMalloc(&slabs, (size_t)max_slabs * n * 16));

    auto reduce = [&](float4* out, int nslab) {
        nbk::ReduceParams r{};
        r.out = out; r.slabs = slabs; r.nslab = nslab; r.slab_stride = n; r.n = n; r.accumulate = 0;
        nbk::reduce_slabs<<<(n + 255) / 256, 256>>>(r);
    };

    // shipped one-sided kernel
    const int js = 16;
    nbk::ForceParams fp{};
    fp.x = dx; fp.out = slabs; fp.i0 = 0; fp.i1 = n; fp.j0 = 0; fp.j1 = n; fp.slab_stride = n; fp.accumulate = 0; fp.eps2 = eps2; fp.wrap = 0;
    auto one_sided = [&] {
        nbk::force_lds<nbk::MathPacked<4>, 2048, 8, 1><<<dim3((n + 1023) / 1024, js), 256>>>(fp);
        reduce(da_ref, js);
    };
    one_sided();
    CK(hipDeviceSynchronize());
    std::vector<float4> a_ref(n), a_sym(n);
    CK(hipMemcpy(a_ref.data(), da_ref, (size_t)n * 16, hipMemcpyDeviceToHost));

    // fp64 truth on sampled targets
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
            e = std::max(e, std::fabs(q.x - truth[3 * s]));
            e = std::max(e, std::fabs(q.y - truth[3 * s + 1]));
            e = std::max(e, std::fabs(q.z - truth[3 * s + 2]));
        }
        return e / amax;
    };
    printf("N=%d %s  one-sided vs fp64 truth: %.3g of max|a|\n", n, plummer ? "plummer" : "cube", err_vs_truth(a_ref));

    if (getenv("SYMBENCH_RECT")) {   // A/B of the rectangular launch (two disjoint halves of the system, as nbody_accel_cross issues it)
        const int h = n / 2, B = 2560;
        const int nbi = (h + B - 1) / B, nbj = (n - h + B - 1) / B;
        float4* xs;
        CK(hipMalloc(&xs, ((size_t)nbj * h + (size_t)nbi * (n - h)) * 16));
        nbk::MassInfo* di;
        CK(hipMalloc(&di, sizeof(nbk::MassInfo)));
        CK(hipMemset(di, 0, sizeof(nbk::MassInfo)));
        nbk::MassScanParams mp{};
        mp.x = dx; mp.i0 = 0; mp.ni = n; mp.out = di; mp.gen = 1;
        nbk::mass_scan<float4><<<1024, 256>>>(mp);
        for (int eq = 0; eq < 2; ++eq) {
            nbk::SymParams sp{};
            sp.x = dx; sp.slabs_i = xs; sp.slabs_j = xs + (size_t)nbj * h; sp.ni = h; sp.nj = n - h; sp.i0 = 0; sp.j0 = h; sp.wrap = n;
            sp.nbi = nbi; sp.nbj = nbj; sp.stride_i = h; sp.stride_j = n - h; sp.rect = 1; sp.eps2 = eps2;
            if (eq) { sp.eqm = di; sp.eq_gen = 1; }
            const int grid = nbi * nbj;
            for (int rep = 0; rep < 2; ++rep) {
                const float tg = median_ms([&] { nbk::force_sym<nbk::SymPacked<10>, 4><<<grid, 256>>>(sp); }, reps);
                const float tr = median_ms([&] { nbk::force_sym_rect<nbk::SymPacked<10>, 4><<<grid, 256>>>(sp); }, reps);
                printf("rect %d x %d blocks (%d tasks), %s path: general kernel %.3f ms, rect-only kernel %.3f ms\n", nbi, nbj, grid,
                       eq ? "equal-mass" : "general", tg, tr);
            }
        }
        CK(hipDeviceSynchronize());
        return 0;
    }
    std::vector<SymVariant> vars;
    nbk::MassInfo* dinfo;
    CK(hipMalloc(&dinfo, sizeof(nbk::MassInfo)));
    CK(hipMemset(dinfo, 0, sizeof(nbk::MassInfo)));
    unsigned eq_gen = 0;
    if (getenv("SYMBENCH_EQ")) {   // A/B of the equal-mass path (same kernels, verdict pointer null or set), interleaved
        for (int rep = 0; rep < 2; ++rep) {
            vars.push_back({"sym bpl10 w4 SQUARE LOCAL decision, off", 2560, [](const nbk::SymParams& p, int grid) { nbk::force_sym_square_local<nbk::SymPacked<10>, 4><<<grid, 256>>>(p, 0); }});
            vars.push_back({"sym bpl10 w4 SQUARE LOCAL decision, on", 2560, [](const nbk::SymParams& p, int grid) { nbk::force_sym_square_local<nbk::SymPacked<10>, 4><<<grid, 256>>>(p, 1); }});
            vars.push_back({"sym bpl10 w4 SQUARE general path", 2560, [](const nbk::SymParams& p, int grid) { nbk::force_sym_square<nbk::SymPacked<10>, 4><<<grid, 256>>>(p); }});
            vars.push_back({"sym bpl10 w4 SQUARE equal-mass", 2560, [](const nbk::SymParams& p, int grid) { nbk::force_sym_square<nbk::SymPacked<10>, 4><<<grid, 256>>>(p); }, true});
            vars.push_back(sym_variant<nbk::SymPacked<10>, 4>("sym bpl10 w4 general kernel, general path"));
            { auto v = sym_variant<nbk::SymPacked<10>, 4>("sym bpl10 w4 general kernel, equal-mass"); v.eq = true; vars.push_back(v); }
            vars.push_back(sym_variant<nbk::SymPacked<8>, 4>("sym bpl8 w4 general path"));
            { auto v = sym_variant<nbk::SymPacked<8>, 4>("sym bpl8 w4 equal-mass"); v.eq = true; vars.push_back(v); }
        }
    } else {
    vars.push_back(sym_variant<nbk::SymPacked<10>, 4>("sym packed bpl10 w4 (B=2560)"));
    vars.push_back({"sym bpl10 w4 SQUARE-only kernel", 2560, [](const nbk::SymParams& p, int grid) { nbk::force_sym_square<nbk::SymPacked<10>, 4><<<grid, 256>>>(p); }});
    vars.push_back({"sym bpl10 w4 waves_per_eu(2,2)", 2560, [](const nbk::SymParams& p, int grid) { nbk::force_sym_wps<nbk::SymPacked<10>, 4, 2><<<grid, 256>>>(p); }});
    vars.push_back(sym_variant<nbk::SymPacked<10>, 4, 2>("sym packed bpl10 w4 minw2 (B=2560)"));
    vars.push_back({"sym bpl12 w4 waves_per_eu(2,2)", 3072, [](const nbk::SymParams& p, int grid) { nbk::force_sym_wps<nbk::SymPacked<12>, 4, 2><<<grid, 256>>>(p); }});
    vars.push_back({"sym bpl8 w4 waves_per_eu(3,3)", 2048, [](const nbk::SymParams& p, int grid) { nbk::force_sym_wps<nbk::SymPacked<8>, 4, 3><<<grid, 256>>>(p); }});
    vars.push_back(sym_variant<nbk::SymPacked<8>, 4>("sym packed bpl8 w4 (B=2048)"));
    vars.push_back({"sym bpl10 w4 SQUARE-only again", 2560, [](const nbk::SymParams& p, int grid) { nbk::force_sym_square<nbk::SymPacked<10>, 4><<<grid, 256>>>(p); }});
    vars.push_back(sym_variant<nbk::SymPacked<10>, 4>("sym packed bpl10 w4 (B=2560) again"));
    }
    const double pairs = (double)n * n;
    const float t_ref = median_ms(one_sided, reps);
    printf("%-34s %8.3f ms  %.3e pairs/s  %.1f%% of 157.3 TF\n", "one-sided lds packed bpl4 t2048", t_ref, pairs / t_ref * 1e3,
           20 * pairs / t_ref * 1e3 / 157.3e12 * 100);
    for (auto& v : vars) {
        const int nb = (n + v.B - 1) / v.B;
        if (nb > max_slabs || nb < 2) continue;
        nbk::SymParams sp{};
        sp.x = dx; sp.slabs_i = slabs; sp.slabs_j = slabs; sp.ni = n; sp.nj = n; sp.i0 = 0; sp.j0 = 0; sp.wrap = 0; sp.nbi = nb; sp.nbj = nb;
        sp.stride_i = n; sp.stride_j = n; sp.rect = 0; sp.eps2 = eps2;
        const int grid = nb * (nb - 1) / 2 + nb;
        if (v.eq) {
            nbk::MassScanParams mp{};
            mp.x = dx; mp.i0 = 0; mp.ni = n; mp.j0 = 0; mp.nj = 0; mp.wrap = 0; mp.out = dinfo; mp.gen = ++eq_gen;
            nbk::mass_scan<float4><<<std::min((n + 255) / 256, 1024), 256>>>(mp);
            CK(hipGetLastError());
            sp.eqm = dinfo; sp.eq_gen = eq_gen;
            nbk::MassInfo hi{};
            CK(hipMemcpy(&hi, dinfo, sizeof hi, hipMemcpyDeviceToHost));
            printf("  mass_scan: %s (m0 = %g)\n", hi.bad_gen != eq_gen ? "uniform" : "NOT uniform", hi.m0);
        }
        auto run = [&] {
            v.launch(sp, grid);
            reduce(da_sym, nb);
        };
        CK(hipMemset(slabs, 0xff, (size_t)nb * n * 16));  // NaN fill: an unwritten slab element shows up
        run();
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(a_sym.data(), da_sym, (size_t)n * 16, hipMemcpyDeviceToHost));
        double dmax = 0, scale = 0;
        for (int i = 0; i < n; ++i) {
            scale = std::max(scale, (double)std::fabs(a_ref[i].x));
            scale = std::max(scale, (double)std::fabs(a_ref[i].y));
            scale = std::max(scale, (double)std::fabs(a_ref[i].z));
        }
        long bad = 0;
        for (int i = 0; i < n; ++i) {
            const double e = std::max({std::fabs((double)a_ref[i].x - a_sym[i].x), std::fabs((double)a_ref[i].y - a_sym[i].y),
                                       std::fabs((double)a_ref[i].z - a_sym[i].z)});
            if (!(e <= 1e30)) ++bad;
            else dmax = std::max(dmax, e);
        }
        const float ms = median_ms(run, reps);
        const float ms_force = median_ms([&] { v.launch(sp, grid); }, reps);
        const float t2 = median_ms(one_sided, reps);
        printf("%-34s %8.3f ms (force alone %.3f)  %.3e pairs/s  %.1f%% of 157.3 TF | vs one-sided max diff %.3g of max|a|, nonfinite %ld, vs truth %.3g | one-sided again %.3f ms\n",
               v.name.c_str(), ms, ms_force, pairs / ms * 1e3, 20 * pairs / ms * 1e3 / 157.3e12 * 100, dmax / scale, bad, err_vs_truth(a_sym), t2);
        fflush(stdout);
    }
    // ---- what the last partial round costs: the square-only kernel on grids of whole and partial rounds of 512 resident workgroups
    {
        constexpr int W = 4, BPL = 10, B = 64 * W * BPL;
        const int nb = (n + B - 1) / B;
        if (nb >= 2 && nb <= max_slabs) {
            nbk::SymParams sp{};
            sp.x = dx; sp.slabs_i = slabs; sp.slabs_j = slabs; sp.ni = n; sp.nj = n; sp.i0 = 0; sp.j0 = 0; sp.wrap = 0; sp.nbi = nb; sp.nbj = nb;
            sp.stride_i = n; sp.stride_j = n; sp.rect = 0; sp.eps2 = eps2;
            const int full = nb * (nb - 1) / 2 + nb, npair = nb * (nb - 1) / 2;
            printf("tail probe (square-only kernel, B=%d, %d tasks of which the last %d are diagonal): grid -> ms\n", B, full, nb);
            for (int grid : {512, 1024, 2560, 4608, 5120, 5120 + 128, 5120 + 236, 5120 + 384, 5632, full, npair}) {
                if (grid > full) continue;
                const float ms = median_ms([&] { nbk::force_sym_square<nbk::SymPacked<BPL>, W><<<grid, 64 * W>>>(sp); }, reps);
                printf("  grid %5d (%.2f rounds): %.3f ms\n", grid, grid / 512.0, ms);
            }
        }
    }
    // ---- row-accumulating tasks (R consecutive J blocks per workgroup): bytes of partial sums and time against the shipped shape
    {
        constexpr int W = 4, BPL = 10, B = 64 * W * BPL;
        const int nb = (n + B - 1) / B;
        if (nb >= 2 && nb <= max_slabs) {
            nbk::SymParams sp{};
            sp.x = dx; sp.slabs_i = slabs; sp.slabs_j = slabs; sp.ni = n; sp.nj = n; sp.i0 = 0; sp.j0 = 0; sp.wrap = 0; sp.nbi = nb; sp.nbj = nb;
            sp.stride_i = n; sp.stride_j = n; sp.rect = 0; sp.eps2 = eps2;
            const int grid0 = nb * (nb - 1) / 2 + nb;
            auto shipped = [&] {
                nbk::force_sym<nbk::SymPacked<BPL>, W><<<grid0, 64 * W>>>(sp);
                reduce(da_ref, nb);
            };
            printf("row-accumulating tasks, B=%d, nb=%d: shipped (R=1) %d tasks, %.1f MB of partial sums written + read again\n", B, nb, grid0,
                   (double)nb * n * 16 / 1e6);
            for (int R : {1, 2, 3, 4, 8}) {
                std::vector<int> first(nb + 1, 0);
                int tasks = 0;
                for (int I = 0; I < nb; ++I) { first[I] = tasks; tasks += (nb - 1 - I + R - 1) / R; }
                first[nb] = tasks;
                int* d_first;
                CK(hipMalloc(&d_first, (nb + 1) * sizeof(int)));
                CK(hipMemcpy(d_first, first.data(), (nb + 1) * sizeof(int), hipMemcpyHostToDevice));
                RowsParams rp{};
                rp.x = dx; rp.slabs = slabs; rp.row_first_task = d_first; rp.n = n; rp.nb = nb; rp.stride = n; rp.R = R; rp.npair_tasks = tasks; rp.eps2 = eps2;
                const int grid = tasks + nb;
                auto rows = [&] {
                    force_sym_rows<nbk::SymPacked<BPL>, W><<<grid, 64 * W>>>(rp);
                    reduce_rows<<<(n + 255) / 256, 256>>>(da_sym, slabs, n, nb, B, R);
                };
                CK(hipMemset(slabs, 0xff, (size_t)nb * n * 16));
                rows();
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(a_sym.data(), da_sym, (size_t)n * 16, hipMemcpyDeviceToHost));
                double dmax = 0, scale = 0;
                long bad = 0;
                for (int i = 0; i < n; ++i) scale = std::max({scale, (double)std::fabs(a_ref[i].x), (double)std::fabs(a_ref[i].y), (double)std::fabs(a_ref[i].z)});
                for (int i = 0; i < n; ++i) {
                    const double e = std::max({std::fabs((double)a_ref[i].x - a_sym[i].x), std::fabs((double)a_ref[i].y - a_sym[i].y),
                                               std::fabs((double)a_ref[i].z - a_sym[i].z)});
                    if (!(e <= 1e30)) ++bad; else dmax = std::max(dmax, e);
                }
                // partial sums actually written: per body of block I: I + 1 + ceil((nb-1-I)/R) slabs
                double bytes = 0;
                for (int I = 0; I < nb; ++I) bytes += (double)std::min(B, n - I * B) * (I + 1 + (nb - 1 - I + R - 1) / R) * 16;
                const float ms_a = median_ms(shipped, reps), ms_r = median_ms(rows, reps), ms_a2 = median_ms(shipped, reps), ms_r2 = median_ms(rows, reps);
                const float ms_fa = median_ms([&] { nbk::force_sym<nbk::SymPacked<BPL>, W><<<grid0, 64 * W>>>(sp); }, reps);
                const float ms_fr = median_ms([&] { force_sym_rows<nbk::SymPacked<BPL>, W><<<grid, 64 * W>>>(rp); }, reps);
                printf("        force alone: rows %.3f ms, shipped %.3f ms\n", ms_fr, ms_fa);
                printf("  R=%d: %5d tasks, partial sums %.1f MB (%.0f%% of shipped) | rows %.3f / %.3f ms, shipped interleaved %.3f / %.3f ms | vs one-sided %.3g of max|a|, nonfinite %ld, vs truth %.3g\n",
                       R, grid, bytes / 1e6, bytes / ((double)nb * n * 16) * 100, ms_r, ms_r2, ms_a, ms_a2, dmax / scale, bad, err_vs_truth(a_sym));
                fflush(stdout);
                CK(hipFree(d_first));
            }
        }
    }
    return 0;
}

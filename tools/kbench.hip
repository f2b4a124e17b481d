// tools/kbench.hip — developer bench: VALU issue-rate microbenchmarks and a sweep of the
// force-kernel variants, all in one process (interleaved rounds, one device).
//   hipcc -O3 --offload-arch=gfx950 -I n-bodysimulation_amd/csrc tools/kbench.hip -o build/kbench
// Not part of the product; results feed DESIGN.md.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "nbody_experiments.hip.h"   // the product header + the measured alternatives

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// ---- microbenchmarks: 8 independent chains per lane, ITER iterations --------------------
__global__ void __launch_bounds__(256) mb_fma(float* out, int iters, float b, float c)
{
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-3f + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_fmaf(a[k], b, c);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) mb_pkfma(float* out, int iters, float b, float c)
{
    f2 a[8];
    f2 bb = {b, b * 1.0001f}, cc = {c, c * 0.999f};
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (f2){threadIdx.x * 1e-3f + k, threadIdx.x * 2e-3f + k};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_elementwise_fma(a[k], bb, cc);
    }
    f2 s = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

__global__ void __launch_bounds__(256) mb_rsq(float* out, int iters, float b, float c)
{
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-3f + k + 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_amdgcn_rsqf(a[k]);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = s + b + c;
}

// 1 rsq + NF fma per chain step: do the costs add, or does the transcendental overlap?
template <int NF>
__global__ void __launch_bounds__(256) mb_mix(float* out, int iters, float b, float c)
{
    float a[8], r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * 1e-3f + k + 1.0f; r[k] = a[k]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            r[k] = __builtin_amdgcn_rsqf(r[k]);
#pragma unroll
            for (int f = 0; f < NF; ++f) a[k] = __builtin_fmaf(a[k], b, c);
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + r[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the pair body on register-resident sources (no memory in the loop): the ALU ceiling
template <class M>
__global__ void __launch_bounds__(256) mb_pair(float* out, int iters, float b, float c)
{
    M t;
    t.set_eps2(0.002f);
#pragma unroll
    for (int k = 0; k < M::BPL; ++k)
        t.set(k, make_float4(threadIdx.x * 1e-2f + k, threadIdx.x * 2e-2f - k, k * 0.5f, 0.f));
    float4 bj = make_float4(b, c, b + c, 1.0f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            t.pair(bj);
            bj.x += c;  // so the compiler cannot hoist the pair
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < M::BPL; ++k) { float4 a = t.acc(k); s += a.x + a.y + a.z; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static int g_inner = 1;  // launches per timed sample (small kernels: amortise the launch latency)
static float time_ms(const std::function<void()>& f0, int reps = 5)
{
    const int inner = g_inner;
    auto f = [&] { for (int q = 0; q < inner; ++q) f0(); };
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
    return ts[ts.size() / 2] / inner;
}

struct Variant {
    std::string name;
    double bpl;
    std::function<void(const nbk::ForceParams&, int nslab)> launch;
};

template <class M, int TILE, int UNROLL, int MINW, int LAYOUT = 0, int WG = 256>
static Variant v_lds(const char* mname)
{
    char nm[64];
    snprintf(nm, sizeof nm, "lds%d %s bpl%d tile%-4d u%-2d wg%d", LAYOUT, mname, M::BPL, TILE, UNROLL, WG);
    return {nm, M::BPL * WG / 256, [](const nbk::ForceParams& p, int nslab) {
                dim3 g((p.i1 - p.i0 + WG * M::BPL - 1) / (WG * M::BPL), nslab);
                nbk::force_lds<M, TILE, UNROLL, MINW, LAYOUT, WG><<<g, WG>>>(p);
            }};
}
template <class M, int UNROLL, int MINW>
static Variant v_sgpr(const char* mname)
{
    char nm[64];
    snprintf(nm, sizeof nm, "sgpr %s bpl%d          u%-2d w%d", mname, M::BPL, UNROLL, MINW);
    return {nm, M::BPL, [](const nbk::ForceParams& p, int nslab) {
                dim3 g((p.i1 - p.i0 + 256 * M::BPL - 1) / (256 * M::BPL), nslab);
                nbk::force_sgpr<M, UNROLL, MINW><<<g, 256>>>(p);
            }};
}
using S1 = nbk::MathScalar<1>; using S2 = nbk::MathScalar<2>; using S4 = nbk::MathScalar<4>; using S8 = nbk::MathScalar<8>;
using P2 = nbk::MathPacked<2>; using P4 = nbk::MathPacked<4>; using P8 = nbk::MathPacked<8>;

int main(int argc, char** argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 65536;
    int do_mb = argc > 2 ? atoi(argv[2]) : 1;
    g_inner = n <= 32768 ? 50 : 1;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz  arch %s\n", prop.name, prop.multiProcessorCount,
           prop.clockRate, prop.gcnArchName);
    const int ncu = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;

    float* scratch;
    CK(hipMalloc(&scratch, sizeof(float) * 256 * ncu * 32));

    if (do_mb) {
        printf("\n== VALU issue microbenchmarks (8 chains/lane). cyc = SIMD cycles per wave-instruction at %.2f GHz\n", ghz);
        printf("%-10s %5s %10s %10s\n", "kernel", "w/SIMD", "ms", "cyc/instr");
        const int iters = 20000;
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = ncu * wps;  // 4 waves per block -> wps waves per SIMD
            auto rep = [&](const char* nm, std::function<void()> f, double instr_per_lane) {
                float ms = time_ms(f);
                // wave-instructions issued per SIMD = wps * instr_per_lane
                double cyc = ms * 1e-3 * ghz * 1e9 / (wps * instr_per_lane);
                printf("%-10s %5d %10.3f %10.2f\n", nm, wps, ms, cyc);
            };
            rep("fma", [&] { mb_fma<<<blocks, 256>>>(scratch, iters, 1.0001f, 0.5f); }, 8.0 * iters);
            rep("pk_fma", [&] { mb_pkfma<<<blocks, 256>>>(scratch, iters, 1.0001f, 0.5f); }, 8.0 * iters);
            rep("rsq", [&] { mb_rsq<<<blocks, 256>>>(scratch, iters, 1.0f, 0.5f); }, 8.0 * iters);
            rep("rsq+4fma", [&] { mb_mix<4><<<blocks, 256>>>(scratch, iters, 1.0001f, 0.5f); }, 8.0 * 5 * iters);
            rep("rsq+12fma", [&] { mb_mix<12><<<blocks, 256>>>(scratch, iters, 1.0001f, 0.5f); }, 8.0 * 13 * iters);
            rep("pair S1", [&] { mb_pair<S1><<<blocks, 256>>>(scratch, iters / 4, 0.3f, 0.01f); }, 8.0 * 1 * (iters / 4));
            rep("pair S2", [&] { mb_pair<S2><<<blocks, 256>>>(scratch, iters / 4, 0.3f, 0.01f); }, 8.0 * 2 * (iters / 4));
            rep("pair S4", [&] { mb_pair<S4><<<blocks, 256>>>(scratch, iters / 4, 0.3f, 0.01f); }, 8.0 * 4 * (iters / 4));
            rep("pair P2", [&] { mb_pair<P2><<<blocks, 256>>>(scratch, iters / 4, 0.3f, 0.01f); }, 8.0 * 2 * (iters / 4));
            rep("pair P4", [&] { mb_pair<P4><<<blocks, 256>>>(scratch, iters / 4, 0.3f, 0.01f); }, 8.0 * 4 * (iters / 4));
            rep("pair P8", [&] { mb_pair<P8><<<blocks, 256>>>(scratch, iters / 4, 0.3f, 0.01f); }, 8.0 * 8 * (iters / 4));
        }
        printf("(pair rows: cyc/instr column = SIMD cycles per PAIR-row, i.e. per 64 pairs)\n");
    }

    // ---- force-kernel variants -------------------------------------------------------
    std::vector<float4> hx(n);
    srand(12345);
    for (int i = 0; i < n; ++i) {
        auto u = [] { return (float)((double)rand() / RAND_MAX); };
        hx[i] = make_float4(u() * 2e5f - 1e5f, u() * 2e5f - 1e5f, u() * 2e5f - 1e5f, 1e5f + u() * (1e9f - 1e5f));
    }
    float4 *dx, *dout, *dref;
    const int max_slab = 64;
    CK(hipMalloc(&dx, sizeof(float4) * n));
    CK(hipMalloc(&dout, sizeof(float4) * (size_t)n * max_slab));
    CK(hipMalloc(&dref, sizeof(float4) * n));
    CK(hipMemcpy(dx, hx.data(), sizeof(float4) * n, hipMemcpyHostToDevice));

    nbk::ForceParams p{};
    p.x = dx; p.i0 = 0; p.i1 = n; p.j0 = 0; p.j1 = n; p.slab_stride = n; p.accumulate = 0; p.eps2 = 0.002f;

    // reference on the GPU: strict kernel
    p.out = dref;
    float ms_strict = time_ms([&] { nbk::force_strict<1024><<<(n + 255) / 256, 256>>>(p); }, 1);
    std::vector<float4> href(n), hout((size_t)n * max_slab);
    CK(hipMemcpy(href.data(), dref, sizeof(float4) * n, hipMemcpyDeviceToHost));
    printf("\n== force kernels, N=%d (%.3g pairs)\nstrict: %.3f ms  %.3g pairs/s\n", n, (double)n * n, ms_strict,
           (double)n * n / (ms_strict * 1e-3));

    std::vector<Variant> vs = {
        v_lds<S1, 256, 8, 1, 0, 256>("S"), v_lds<P2, 256, 8, 1, 0, 256>("P"), v_lds<P2, 512, 8, 1, 0, 256>("P"), v_lds<P4, 512, 8, 1, 0, 256>("P"),
        v_lds<P2, 256, 8, 1, 0, 128>("P"), v_lds<P2, 128, 8, 1, 0, 128>("P"), v_lds<P4, 256, 8, 1, 0, 128>("P"), v_lds<P4, 512, 8, 1, 0, 128>("P"),
        v_lds<P2, 256, 8, 1, 0, 64>("P"),  v_lds<P2, 128, 8, 1, 0, 64>("P"),  v_lds<P2, 64, 8, 1, 0, 64>("P"),   v_lds<P4, 256, 8, 1, 0, 64>("P"),
        v_lds<P4, 128, 8, 1, 0, 64>("P"),  v_lds<P4, 512, 8, 1, 0, 64>("P"),  v_lds<S1, 128, 8, 1, 0, 64>("S"),  v_lds<P8, 256, 8, 1, 0, 64>("P"),
    };
    p.out = dout;
    printf("%-34s %5s %9s %12s %8s %10s\n", "variant", "slabs", "ms", "pairs/s", "%peak20", "max rel err");
    double amax = 0;
    for (int i = 0; i < n; ++i) amax = std::max({amax, (double)fabsf(href[i].x), (double)fabsf(href[i].y), (double)fabsf(href[i].z)});
    for (auto& v : vs) {
        for (int nslab : {1, 2, 4, 8, 16, 32, 64}) {
            // keep roughly 2..16 waves per SIMD worth of work
            double waves = (double)n / (64.0 * v.bpl) * nslab;   // v.bpl here = targets per 256 lanes
            if (waves < 1024 * 1.0 || waves > 1024 * 40) continue;
            float ms = time_ms([&] { v.launch(p, nslab); }, 3);
            CK(hipMemcpy(hout.data(), dout, sizeof(float4) * (size_t)n * nslab, hipMemcpyDeviceToHost));
            double err = 0;
            for (int i = 0; i < n; ++i) {
                double sx = 0, sy = 0, sz = 0;
                for (int s = 0; s < nslab; ++s) {
                    sx += hout[(size_t)s * n + i].x; sy += hout[(size_t)s * n + i].y; sz += hout[(size_t)s * n + i].z;
                }
                err = std::max({err, fabs(sx - href[i].x), fabs(sy - href[i].y), fabs(sz - href[i].z)});
            }
            double pps = (double)n * n / (ms * 1e-3);
            printf("%-34s %5d %9.3f %12.4g %8.1f %10.2e\n", v.name.c_str(), nslab, ms, pps, pps * 20 / 157.3e12 * 100, err / amax);
        }
    }
    return 0;
}

// tools/mfma_mb.hip — does the f32 matrix pipe run beside packed-VALU + transcendental work of the
// same wave? Probe for the "MFMA as a broadcast-subtract engine" idea. Developer tool only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long t0, t1, r0, r1; };

// per iteration: NM mfma 32x32x2 (zero C), NPK packed ops, NR rsq
template <int NM, int NPK8, int NR8>
__global__ void __launch_bounds__(256) k_mix(float* out, Stamp* st, int iters, float b, float c)
{
    f2 p0 = {threadIdx.x * 1.f, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
    float q0 = threadIdx.x + 1.f, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3, q4 = q0 + 4, q5 = q0 + 5, q6 = q0 + 6, q7 = q0 + 7;
    f2 b2 = {b, b}, c2 = {c, c};
    f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float av = threadIdx.x * 0.5f, bv = (threadIdx.x & 31) * 0.25f;
    const f16v zero = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (NM >= 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, zero, 0, 0, 0);
        if (NM >= 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, zero, 0, 0, 0);
        if (NM >= 3) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, av, zero, 0, 0, 0);
        if (NM >= 4) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, bv, zero, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NPK8; ++g)
            asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                         "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(b2), "v"(c2));
#pragma unroll
        for (int g = 0; g < NR8; ++g)
            asm volatile("v_rsq_f32_e32 %0, %0\n v_rsq_f32_e32 %1, %1\n v_rsq_f32_e32 %2, %2\n v_rsq_f32_e32 %3, %3\n"
                         "v_rsq_f32_e32 %4, %4\n v_rsq_f32_e32 %5, %5\n v_rsq_f32_e32 %6, %6\n v_rsq_f32_e32 %7, %7\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
        // consume the matrix results so they stay live (cheap: one add per accumulator)
        if (NM >= 1) av += acc0[0] * 1e-30f;
        if (NM >= 2) av += acc1[5] * 1e-30f;
        if (NM >= 3) bv += acc2[9] * 1e-30f;
        if (NM >= 4) bv += acc3[15] * 1e-30f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { Stamp s = {t0, t1, r0, r1}; st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s; }
    f2 s = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7 + av + bv;
}

typedef void (*Fn)(float*, Stamp*, int, float, float);
struct T { const char* name; Fn fn; };

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float* out; Stamp* st;
    CK(hipMalloc(&out, sizeof(float) * 256 * ncu * 8));
    CK(hipMalloc(&st, sizeof(Stamp) * 4 * ncu * 8));
    std::vector<Stamp> h(4 * ncu * 8);
    std::vector<T> tests = {
        {"0 mfma + 72 pk + 16 rsq", k_mix<0, 9, 2>}, {"3 mfma + 72 pk + 16 rsq", k_mix<3, 9, 2>},
        {"4 mfma + 72 pk + 16 rsq", k_mix<4, 9, 2>}, {"4 mfma only", k_mix<4, 0, 0>},
        {"3 mfma only", k_mix<3, 0, 0>},             {"4 mfma + 72 pk", k_mix<4, 9, 0>},
        {"0 mfma + 72 pk", k_mix<0, 9, 0>},          {"0 mfma + 96 pk + 16 rsq", k_mix<0, 12, 2>},
    };
    printf("%-28s %6s %9s %8s %14s\n", "test", "w/SIMD", "ms", "MHz", "cyc/iteration");
    for (auto& t : tests)
        for (int wps : {1, 2, 4}) {
            const int blocks = ncu * wps, iters = 40000 / wps;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            t.fn<<<blocks, 256>>>(out, st, iters, 1.0001f, 0.5f);
            CK(hipEventRecord(e0));
            t.fn<<<blocks, 256>>>(out, st, iters, 1.0001f, 0.5f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), st, sizeof(Stamp) * 4 * blocks, hipMemcpyDeviceToHost));
            std::vector<double> mhz;
            for (int w = 0; w < 4 * blocks; ++w) { double dt = h[w].t1 - h[w].t0, dr = h[w].r1 - h[w].r0; if (dr > 0) mhz.push_back(dt / dr * 100.0); }
            std::sort(mhz.begin(), mhz.end());
            double f = mhz[mhz.size() / 2] * 1e6;
            printf("%-28s %6d %9.3f %8.0f %14.1f\n", t.name, wps, ms, f * 1e-6, ms * 1e-3 * f / ((double)iters * wps));
        }
    return 0;
}

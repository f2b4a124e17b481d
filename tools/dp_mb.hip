// tools/dp_mb.hip — fp64 VALU issue costs on gfx950 (developer probe; same method as valu_mb.hip)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)
struct Stamp { unsigned long long t0, t1, r0, r1; };
#define X8(T) T("%0") T("%1") T("%2") T("%3") T("%4") T("%5") T("%6") T("%7")
#define KERNEL(NAME, T)                                                                              \
    __global__ void __launch_bounds__(256) NAME(double* out, Stamp* st, int iters, double b, double c) \
    {                                                                                                \
        double a0 = threadIdx.x + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();  \
        for (int it = 0; it < iters; ++it)                                                           \
            asm volatile(X8(T) X8(T) X8(T) X8(T) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();  \
        if ((threadIdx.x & 63) == 0) { Stamp s = {t0, t1, r0, r1}; st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s; } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    }
#define T_FMA(r) "v_fma_f64 " r ", %8, %9, " r "\n"
#define T_MUL(r) "v_mul_f64 " r ", " r ", %8\n"
#define T_ADD(r) "v_add_f64 " r ", " r ", %8\n"
#define T_RSQ(r) "v_rsq_f64_e32 " r ", " r "\n"
#define T_RCP(r) "v_rcp_f64_e32 " r ", " r "\n"
#define T_SQRT(r) "v_sqrt_f64_e32 " r ", " r "\n"
KERNEL(k_fma, T_FMA) KERNEL(k_mul, T_MUL) KERNEL(k_add, T_ADD) KERNEL(k_rsq, T_RSQ) KERNEL(k_rcp, T_RCP) KERNEL(k_sqrt, T_SQRT)
// mixed sequences: eight doubles (%0-%7) and eight floats (%8-%15); T2(d, f) is one instance
#define Y8(T) T("%0", "%8") T("%1", "%9") T("%2", "%10") T("%3", "%11") T("%4", "%12") T("%5", "%13") T("%6", "%14") T("%7", "%15")
#define KERNEL2(NAME, T)                                                                             \
    __global__ void __launch_bounds__(256) NAME(double* out, Stamp* st, int iters, double b, double c) \
    {                                                                                                \
        double a0 = threadIdx.x + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float f0 = 1, f1 = 2, f2 = 3, f3 = 4, f4 = 5, f5 = 6, f6 = 7, f7 = 8;                         \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();  \
        for (int it = 0; it < iters; ++it)                                                           \
            asm volatile(Y8(T) Y8(T) Y8(T) Y8(T) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                         "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();  \
        if ((threadIdx.x & 63) == 0) { Stamp s = {t0, t1, r0, r1}; st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s; } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7; \
    }
#define T_CVT_DF(d, f) "v_cvt_f32_f64_e32 " f ", " d "\n"
#define T_CVT_FD(d, f) "v_cvt_f64_f32_e32 " d ", " f "\n"
#define T_RSQ32(d, f) "v_rsq_f32_e32 " f ", " f "\n"
#define T_SEED32(d, f) "v_cvt_f32_f64_e32 " f ", " d "\n v_rsq_f32_e32 " f ", " f "\n v_cvt_f64_f32_e32 " d ", " f "\n"
KERNEL2(k_cvt_df, T_CVT_DF) KERNEL2(k_cvt_fd, T_CVT_FD) KERNEL2(k_rsq32, T_RSQ32) KERNEL2(k_seed32, T_SEED32)
typedef void (*Fn)(double*, Stamp*, int, double, double);
int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    double* out; Stamp* st;
    CK(hipMalloc(&out, sizeof(double) * 256 * ncu * 8)); CK(hipMalloc(&st, sizeof(Stamp) * 4 * ncu * 8));
    std::vector<Stamp> h(4 * ncu * 8);
    struct { const char* n; Fn f; } tests[] = {{"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_rsq_f64", k_rsq}, {"v_rcp_f64", k_rcp}, {"v_sqrt_f64", k_sqrt},
                                             {"v_cvt_f32_f64", k_cvt_df}, {"v_cvt_f64_f32", k_cvt_fd}, {"v_rsq_f32", k_rsq32},
                                             {"cvt+rsq32+cvt (one seed)", k_seed32}};
    printf("%-26s %6s %9s %8s %10s\n", "test", "w/SIMD", "ms", "MHz", "cyc/instr");
    for (auto& t : tests)
        for (int wps : {2, 4, 8}) {
            const int blocks = ncu * wps, iters = 2000000 / (32 * wps);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            t.f<<<blocks, 256>>>(out, st, iters, 1.0000001, 0.5);
            CK(hipEventRecord(e0)); t.f<<<blocks, 256>>>(out, st, iters, 1.0000001, 0.5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), st, sizeof(Stamp) * 4 * blocks, hipMemcpyDeviceToHost));
            std::vector<double> mhz;
            for (int w = 0; w < 4 * blocks; ++w) { double dt = h[w].t1 - h[w].t0, dr = h[w].r1 - h[w].r0; if (dr > 0) mhz.push_back(dt / dr * 100.0); }
            std::sort(mhz.begin(), mhz.end());
            const double f = mhz[mhz.size() / 2] * 1e6;
            printf("%-26s %6d %9.3f %8.0f %10.2f\n", t.n, wps, ms, f * 1e-6, ms * 1e-3 * f / ((double)iters * 32 * wps));
        }
    return 0;
}

// tools/valu_mb.hip — instruction-level VALU issue-rate probes for gfx950, with the real
// shader clock measured in-kernel (s_memtime / s_memrealtime). Developer tool only.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_mb.hip -o build/valu_mb
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

struct Stamp {
    unsigned long long t0, t1, r0, r1;
};

#define STAMP_BEGIN                                                  \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();            \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define STAMP_END                                                                     \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                         \
    if ((threadIdx.x & 63) == 0) {                                                    \
        Stamp s = {t0, t1, r0, r1};                                                   \
        st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;                         \
    }

// 16 scalar accumulators
#define DECL16                                                                                    \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,      \
          a6 = a0 + 6, a7 = a0 + 7, a8 = a0 + 8, a9 = a0 + 9, a10 = a0 + 10, a11 = a0 + 11,       \
          a12 = a0 + 12, a13 = a0 + 13, a14 = a0 + 14, a15 = a0 + 15;
#define OPS16                                                                                   \
    "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8),   \
        "+v"(a9), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15)
#define SUM16 (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15)

// one instruction template applied to the 16 accumulators; %16 = b, %17 = c
#define X16(T)                                                                                  \
    T("%0") T("%1") T("%2") T("%3") T("%4") T("%5") T("%6") T("%7") T("%8") T("%9") T("%10")    \
        T("%11") T("%12") T("%13") T("%14") T("%15")

#define KERNEL_S16(NAME, T)                                                               \
    __global__ void __launch_bounds__(256) NAME(float* out, Stamp* st, int iters, float b, float c) \
    {                                                                                     \
        DECL16                                                                            \
        STAMP_BEGIN                                                                       \
        for (int it = 0; it < iters; ++it) {                                              \
            asm volatile(X16(T) X16(T) : OPS16 : "v"(b), "v"(c));                         \
        }                                                                                 \
        STAMP_END                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SUM16;                               \
    }

#define T_FMAC(r) "v_fmac_f32_e32 " r ", %16, %17\n"
#define T_FMA3(r) "v_fma_f32 " r ", " r ", %16, %17\n"
#define T_MUL(r) "v_mul_f32_e32 " r ", %16, " r "\n"
#define T_ADD(r) "v_add_f32_e32 " r ", %16, " r "\n"
#define T_SUB(r) "v_sub_f32_e32 " r ", %16, " r "\n"
#define T_RSQ(r) "v_rsq_f32_e32 " r ", " r "\n"
#define T_MOV(r) "v_mov_b32_e32 " r ", %16\n"

KERNEL_S16(k_fmac, T_FMAC)
KERNEL_S16(k_fma3, T_FMA3)
KERNEL_S16(k_mul, T_MUL)
KERNEL_S16(k_add, T_ADD)
KERNEL_S16(k_rsq, T_RSQ)
KERNEL_S16(k_mov, T_MOV)

// SGPR-operand forms
#define KERNEL_S16S(NAME, T)                                                              \
    __global__ void __launch_bounds__(256) NAME(float* out, Stamp* st, int iters, float b, float c) \
    {                                                                                     \
        DECL16                                                                            \
        STAMP_BEGIN                                                                       \
        for (int it = 0; it < iters; ++it) {                                              \
            asm volatile(X16(T) X16(T) : OPS16 : "s"(b), "v"(c));                         \
        }                                                                                 \
        STAMP_END                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SUM16;                               \
    }
#define T_SUBS(r) "v_sub_f32_e32 " r ", %16, " r "\n"
#define T_FMA3S(r) "v_fma_f32 " r ", " r ", " r ", %16\n"
KERNEL_S16S(k_sub_s, T_SUBS)
KERNEL_S16S(k_fma_s, T_FMA3S)

// 8 packed accumulators (64-bit pairs); %8 = b2, %9 = c2
#define DECL8P                                                                                   \
    f2 p0 = {threadIdx.x * 1.f, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, \
       p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
#define OPS8P "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
#define SUM8P ((p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7).x + (p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7).y)
#define X8(T) T("%0") T("%1") T("%2") T("%3") T("%4") T("%5") T("%6") T("%7")

#define KERNEL_P8(NAME, T)                                                                \
    __global__ void __launch_bounds__(256) NAME(float* out, Stamp* st, int iters, float b, float c) \
    {                                                                                     \
        DECL8P                                                                            \
        f2 b2 = {b, b}, c2 = {c, c};                                                      \
        STAMP_BEGIN                                                                       \
        for (int it = 0; it < iters; ++it) {                                              \
            asm volatile(X8(T) X8(T) X8(T) X8(T) : OPS8P : "v"(b2), "v"(c2));             \
        }                                                                                 \
        STAMP_END                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8P;                               \
    }
#define T_PKFMA(r) "v_pk_fma_f32 " r ", " r ", %8, %9\n"
#define T_PKFMAC(r) "v_pk_fma_f32 " r ", %8, %9, " r "\n"
#define T_PKMUL(r) "v_pk_mul_f32 " r ", " r ", %8\n"
#define T_PKADD(r) "v_pk_add_f32 " r ", " r ", %8\n"
#define T_PKADDSEL(r) "v_pk_add_f32 " r ", %8, " r " op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
KERNEL_P8(k_pkfma, T_PKFMA)
KERNEL_P8(k_pkfmac, T_PKFMAC)
KERNEL_P8(k_pkmul, T_PKMUL)
KERNEL_P8(k_pkadd, T_PKADD)
KERNEL_P8(k_pkaddsel, T_PKADDSEL)

// mixes: NR rsq on a0.. and NF other ops; generic via a raw string
#define KERNEL_RAW16(NAME, BODY)                                                          \
    __global__ void __launch_bounds__(256) NAME(float* out, Stamp* st, int iters, float b, float c) \
    {                                                                                     \
        DECL16                                                                            \
        STAMP_BEGIN                                                                       \
        for (int it = 0; it < iters; ++it) {                                              \
            asm volatile(BODY : OPS16 : "v"(b), "v"(c));                                  \
        }                                                                                 \
        STAMP_END                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SUM16;                               \
    }
// 1 rsq : 3 fmac  (4 groups -> 16 instr)
KERNEL_RAW16(k_r1f3, T_RSQ("%0") T_FMAC("%4") T_FMAC("%5") T_FMAC("%6") T_RSQ("%1") T_FMAC("%7") T_FMAC("%8")
                         T_FMAC("%9") T_RSQ("%2") T_FMAC("%10") T_FMAC("%11") T_FMAC("%12") T_RSQ("%3")
                             T_FMAC("%13") T_FMAC("%14") T_FMAC("%15"))
// 1 rsq : 7 fmac (2 groups -> 16 instr)
KERNEL_RAW16(k_r1f7, T_RSQ("%0") T_FMAC("%2") T_FMAC("%3") T_FMAC("%4") T_FMAC("%5") T_FMAC("%6") T_FMAC("%7")
                         T_FMAC("%8") T_RSQ("%1") T_FMAC("%9") T_FMAC("%10") T_FMAC("%11") T_FMAC("%12")
                             T_FMAC("%13") T_FMAC("%14") T_FMAC("%15"))
// 1 rsq : 1 fmac
KERNEL_RAW16(k_r1f1, T_RSQ("%0") T_FMAC("%8") T_RSQ("%1") T_FMAC("%9") T_RSQ("%2") T_FMAC("%10") T_RSQ("%3")
                         T_FMAC("%11") T_RSQ("%4") T_FMAC("%12") T_RSQ("%5") T_FMAC("%13") T_RSQ("%6")
                             T_FMAC("%14") T_RSQ("%7") T_FMAC("%15"))
// 1 rsq : 12 mixed (3 sub, 3 fma, 3 mul, 3 fmac) -> the scalar pair's shape, 13 instr
KERNEL_RAW16(k_pairS, T_SUB("%1") T_SUB("%2") T_SUB("%3") T_FMA3("%4") T_FMAC("%5") T_FMAC("%6") T_RSQ("%0")
                          T_MUL("%7") T_MUL("%8") T_MUL("%9") T_FMAC("%10") T_FMAC("%11") T_FMAC("%12"))

// packed pair shape: 2 rsq + 3 pk_add + 6 pk_fma + 3 pk_mul = 14 instr per TWO pairs
__global__ void __launch_bounds__(256) k_pairP(float* out, Stamp* st, int iters, float b, float c)
{
    DECL8P
    float q0 = threadIdx.x + 1.f, q1 = q0 + 1.f;
    f2 b2 = {b, b}, c2 = {c, c};
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        asm volatile(T_PKADD("%0") T_PKADD("%1") T_PKADD("%2") T_PKFMA("%3") T_PKFMAC("%4") T_PKFMAC("%5")
                         "v_rsq_f32_e32 %10, %10\n v_rsq_f32_e32 %11, %11\n" T_PKMUL("%6") T_PKMUL("%7")
                             T_PKMUL("%0") T_PKFMAC("%1") T_PKFMAC("%2") T_PKFMAC("%3")
                     : OPS8P, "+v"(b2), "+v"(c2), "+v"(q0), "+v"(q1));
    }
    STAMP_END
    out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8P + q0 + q1;
}

// LDS broadcast read beside VALU: 1 ds_read_b128 per NV fmac
template <int NV>
__global__ void __launch_bounds__(256) k_lds_fmac(float* out, Stamp* st, int iters, float b, float c)
{
    __shared__ float4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = make_float4(i, b, c, 1.f);
    __syncthreads();
    DECL16
    float4 acc = make_float4(0, 0, 0, 0);
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        const float4 v = sh[it & 1023];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;  // 4 VALU that consume the read
        if (NV >= 16) asm volatile(X16(T_FMAC) : OPS16 : "v"(b), "v"(c));
        if (NV >= 32) asm volatile(X16(T_FMAC) : OPS16 : "v"(b), "v"(c));
        if (NV >= 48) asm volatile(X16(T_FMAC) : OPS16 : "v"(b), "v"(c));
    }
    STAMP_END
    out[blockIdx.x * blockDim.x + threadIdx.x] = SUM16 + acc.x + acc.y + acc.z + acc.w;
}


// scalar pair with the three subtracts and one mul taking an SGPR source (the "sgpr" kernel's shape)
__global__ void __launch_bounds__(256) k_pairS_sgpr(float* out, Stamp* st, int iters, float b, float c)
{
    DECL16
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_sub_f32_e32 %1, %16, %1\n v_sub_f32_e32 %2, %16, %2\n v_sub_f32_e32 %3, %16, %3\n"
                     T_FMA3("%4") T_FMAC("%5") T_FMAC("%6") T_RSQ("%0")
                     T_MUL("%7") "v_mul_f32_e32 %8, %16, %8\n" T_MUL("%9") T_FMAC("%10") T_FMAC("%11") T_FMAC("%12")
                     : OPS16 : "s"(b), "v"(c));
    }
    STAMP_END
    out[blockIdx.x * blockDim.x + threadIdx.x] = SUM16;
}
// 1 rsq : 12 fmac
KERNEL_RAW16(k_r1f12, T_RSQ("%0") T_FMAC("%1") T_FMAC("%2") T_FMAC("%3") T_FMAC("%4") T_FMAC("%5") T_FMAC("%6")
                          T_FMAC("%7") T_FMAC("%8") T_FMAC("%9") T_FMAC("%10") T_FMAC("%11") T_FMAC("%12"))
// 2 rsq back to back then 12 fmac (two pairs' worth, scalar)
KERNEL_RAW16(k_r2f24, T_RSQ("%0") T_RSQ("%13") T_FMAC("%1") T_FMAC("%2") T_FMAC("%3") T_FMAC("%4") T_FMAC("%5") T_FMAC("%6")
                          T_FMAC("%7") T_FMAC("%8") T_FMAC("%9") T_FMAC("%10") T_FMAC("%11") T_FMAC("%12")
                          T_FMAC("%1") T_FMAC("%2") T_FMAC("%3") T_FMAC("%4") T_FMAC("%5") T_FMAC("%6")
                          T_FMAC("%7") T_FMAC("%8") T_FMAC("%9") T_FMAC("%10") T_FMAC("%11") T_FMAC("%12"))

// packed mixes: NP pk_fma per 2 rsq
template <int NP>
__global__ void __launch_bounds__(256) k_r2pk(float* out, Stamp* st, int iters, float b, float c)
{
    DECL8P
    float q0 = threadIdx.x + 1.f, q1 = q0 + 1.f;
    f2 b2 = {b, b}, c2 = {c, c};
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_rsq_f32_e32 %10, %10\n v_rsq_f32_e32 %11, %11\n" : OPS8P, "+v"(b2), "+v"(c2), "+v"(q0), "+v"(q1));
        if (NP >= 4) asm volatile(T_PKFMAC("%0") T_PKFMAC("%1") T_PKFMAC("%2") T_PKFMAC("%3") : OPS8P, "+v"(b2), "+v"(c2), "+v"(q0), "+v"(q1));
        if (NP >= 8) asm volatile(T_PKFMAC("%4") T_PKFMAC("%5") T_PKFMAC("%6") T_PKFMAC("%7") : OPS8P, "+v"(b2), "+v"(c2), "+v"(q0), "+v"(q1));
        if (NP >= 12) asm volatile(T_PKFMAC("%0") T_PKFMAC("%1") T_PKFMAC("%2") T_PKFMAC("%3") : OPS8P, "+v"(b2), "+v"(c2), "+v"(q0), "+v"(q1));
        if (NP >= 16) asm volatile(T_PKFMAC("%4") T_PKFMAC("%5") T_PKFMAC("%6") T_PKFMAC("%7") : OPS8P, "+v"(b2), "+v"(c2), "+v"(q0), "+v"(q1));
    }
    STAMP_END
    out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8P + q0 + q1;
}
// packed + scalar blend: per two pairs 6 pk + 12 scalar + 2 rsq (half the work packed)
__global__ void __launch_bounds__(256) k_blend(float* out, Stamp* st, int iters, float b, float c)
{
    DECL8P
    DECL16
    f2 b2 = {b, b}, c2 = {c, c};
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        asm volatile(T_PKFMAC("%0") T_PKFMAC("%1") T_PKFMAC("%2") T_PKFMAC("%3") T_PKFMAC("%4") T_PKFMAC("%5")
                     : OPS8P, "+v"(b2), "+v"(c2));
        asm volatile(T_RSQ("%0") T_RSQ("%13") T_FMAC("%1") T_FMAC("%2") T_FMAC("%3") T_FMAC("%4") T_FMAC("%5") T_FMAC("%6")
                     T_FMAC("%7") T_FMAC("%8") T_FMAC("%9") T_FMAC("%10") T_FMAC("%11") T_FMAC("%12")
                     : OPS16 : "v"(b), "v"(c));
    }
    STAMP_END
    out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8P + SUM16;
}

struct Test {
    const char* name;
    void (*fn)(float*, Stamp*, int, float, float);
    int instr_per_iter;  // wave-instructions per loop iteration (VALU only)
    double flop_per_instr;
};

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device CUs %d nominal clock %.0f MHz\n", ncu, prop.clockRate * 1e-3);
    float* out;
    Stamp* st;
    const int maxblocks = ncu * 8;
    CK(hipMalloc(&out, sizeof(float) * 256 * maxblocks));
    CK(hipMalloc(&st, sizeof(Stamp) * 4 * maxblocks));
    std::vector<Stamp> hst(4 * maxblocks);

    std::vector<Test> tests = {
        {"v_fmac_f32_e32", k_fmac, 32, 2},      {"v_fma_f32 (3 vgpr)", k_fma3, 32, 2},
        {"v_mul_f32_e32", k_mul, 32, 1},        {"v_add_f32_e32", k_add, 32, 1},
        {"v_sub_f32 sgpr", k_sub_s, 32, 1},     {"v_fma_f32 sgpr", k_fma_s, 32, 2},
        {"v_rsq_f32", k_rsq, 32, 1},
        {"v_pk_fma_f32", k_pkfma, 32, 4},       {"v_pk_fma_f32 (acc)", k_pkfmac, 32, 4},
        {"v_pk_mul_f32", k_pkmul, 32, 2},       {"v_pk_add_f32", k_pkadd, 32, 2},
        {"v_pk_add_f32 opsel", k_pkaddsel, 32, 2},
        {"1 rsq : 1 fmac", k_r1f1, 16, 0},      {"1 rsq : 3 fmac", k_r1f3, 16, 0},
        {"1 rsq : 7 fmac", k_r1f7, 16, 0},      {"1 rsq : 12 fmac", k_r1f12, 13, 0},
        {"2 rsq : 24 fmac", k_r2f24, 26, 0},
        {"2 rsq : 4 pk", k_r2pk<4>, 6, 0},      {"2 rsq : 8 pk", k_r2pk<8>, 10, 0},
        {"2 rsq : 12 pk", k_r2pk<12>, 14, 0},   {"2 rsq : 16 pk", k_r2pk<16>, 18, 0},
        {"2rsq:6pk:12fmac", k_blend, 20, 0},
        {"pair scalar (13)", k_pairS, 13, 0},   {"pair scalar sgpr (13)", k_pairS_sgpr, 13, 0},
        {"pair packed (14/2)", k_pairP, 14, 0},
        {"ds_read_b128 + 4+16 valu", k_lds_fmac<16>, 20, 0},
        {"ds_read_b128 + 4+32 valu", k_lds_fmac<32>, 36, 0},
    };

    printf("%-26s %6s %9s %9s %12s %14s\n", "test", "w/SIMD", "ms", "MHz", "cyc/instr", "cyc/iteration");
    for (auto& t : tests) {
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = ncu * wps;
            const int iters = 4000000 / (t.instr_per_iter * wps);  // ~4M wave-instr per SIMD
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            t.fn<<<blocks, 256>>>(out, st, iters, 1.0001f, 0.5f);
            CK(hipEventRecord(e0));
            t.fn<<<blocks, 256>>>(out, st, iters, 1.0001f, 0.5f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hst.data(), st, sizeof(Stamp) * 4 * blocks, hipMemcpyDeviceToHost));
            std::vector<double> mhz;
            for (int w = 0; w < 4 * blocks; ++w) {
                double dt = (double)(hst[w].t1 - hst[w].t0), dr = (double)(hst[w].r1 - hst[w].r0);
                if (dr > 0) mhz.push_back(dt / dr * 100.0);
            }
            std::sort(mhz.begin(), mhz.end());
            const double f = mhz[mhz.size() / 2] * 1e6;
            // wall-clock based: every SIMD executed wps * iters * instr wave-instructions
            const double c = ms * 1e-3 * f / ((double)iters * t.instr_per_iter * wps);
            printf("%-26s %6d %9.3f %9.0f %12.2f %14.1f\n", t.name, wps, ms, f * 1e-6, c, c * t.instr_per_iter);
            CK(hipEventDestroy(e0));
            CK(hipEventDestroy(e1));
        }
    }
    return 0;
}

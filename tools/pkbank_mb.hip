// tools/pkbank_mb.hip — does the VGPR BANK of a packed-fp32 instruction's operands change its issue cost on gfx950?
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pkbank_mb.hip -o build/pkbank_mb && build/pkbank_mb
// Why: two builds of the rotation kernel with the same instruction multiset but another register assignment differed by 2 % (round 5,
// profiles/r05_one_rotation_body.txt). Each variant below issues 64 independent v_pk_fma_f32 per loop iteration from 8 waves per SIMD,
// with the three source register PAIRS placed so that their first registers fall on chosen banks (register number mod 4), and reports
// the time per wave instruction relative to a plain v_fma_f32 loop of the same shape.
// Result (profiles/r05h_pkbank_mb.txt): 1.823-1.837 x v_fma_f32 for every placement - operand banks do not matter; the 2 % came from elsewhere.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// 16 destinations v[32:33] ... v[62:63]; sources A, B, C given as register numbers of the pair's first register
#define PK4(d, A, B, C)                                                                                             \
    "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #A ":" #A "+1], v[" #B ":" #B "+1], v[" #C ":" #C "+1]\n"
#define ROW(A, B, C)                                                                                                \
    PK4(32, A, B, C) PK4(34, A, B, C) PK4(36, A, B, C) PK4(38, A, B, C) PK4(40, A, B, C) PK4(42, A, B, C) PK4(44, A, B, C) PK4(46, A, B, C) \
    PK4(48, A, B, C) PK4(50, A, B, C) PK4(52, A, B, C) PK4(54, A, B, C) PK4(56, A, B, C) PK4(58, A, B, C) PK4(60, A, B, C) PK4(62, A, B, C)
#define CLOBBER "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
                "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"

template <int V>
__global__ void __launch_bounds__(256) pk_loop(float* out, int iters)
{
    // sources live in v[8:9] ... v[22:23]: initialise them all (values stay finite: a*b+c with small a)
    asm volatile("v_mov_b32 v8, 0x3a000000\nv_mov_b32 v9, 0x3a000000\nv_mov_b32 v10, 0x3a000000\nv_mov_b32 v11, 0x3a000000\n"
                 "v_mov_b32 v12, 0x3a000000\nv_mov_b32 v13, 0x3a000000\nv_mov_b32 v14, 0x3a000000\nv_mov_b32 v15, 0x3a000000\n"
                 "v_mov_b32 v16, 0x3a000000\nv_mov_b32 v17, 0x3a000000\nv_mov_b32 v18, 0x3a000000\nv_mov_b32 v19, 0x3a000000\n"
                 "v_mov_b32 v20, 0x3a000000\nv_mov_b32 v21, 0x3a000000\nv_mov_b32 v22, 0x3a000000\nv_mov_b32 v23, 0x3a000000\n"
                 ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
    for (int k = 0; k < iters; ++k) {
        // first registers of the pairs: banks (A, B, C) = register number mod 4
        if constexpr (V == 0) asm volatile(ROW(8, 12, 16) ROW(8, 12, 16) ROW(8, 12, 16) ROW(8, 12, 16) ::: CLOBBER);        // 0 0 0
        if constexpr (V == 1) asm volatile(ROW(8, 10, 16) ROW(8, 10, 16) ROW(8, 10, 16) ROW(8, 10, 16) ::: CLOBBER);        // 0 2 0
        if constexpr (V == 2) asm volatile(ROW(8, 10, 14) ROW(8, 10, 14) ROW(8, 10, 14) ROW(8, 10, 14) ::: CLOBBER);        // 0 2 2
        if constexpr (V == 3) asm volatile(ROW(8, 12, 14) ROW(8, 12, 14) ROW(8, 12, 14) ROW(8, 12, 14) ::: CLOBBER);        // 0 0 2
        if constexpr (V == 4) asm volatile(ROW(10, 14, 18) ROW(10, 14, 18) ROW(10, 14, 18) ROW(10, 14, 18) ::: CLOBBER);    // 2 2 2 (pairs must be even-aligned on gfx950)
        if constexpr (V == 5) asm volatile(ROW(8, 8, 8) ROW(8, 8, 8) ROW(8, 8, 8) ROW(8, 8, 8) ::: CLOBBER);                // the same pair three times
        if constexpr (V == 6) asm volatile(ROW(8, 10, 10) ROW(8, 10, 10) ROW(8, 10, 10) ROW(8, 10, 10) ::: CLOBBER);        // 0 2 2, B and C the same pair
    }
    float r;
    asm volatile("v_add_f32 %0, v32, v63" : "=v"(r) :: "memory");
    if (r == 123.456f) out[threadIdx.x] = r;
}

__global__ void __launch_bounds__(256) fma_loop(float* out, int iters)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    const float b = 1.0001f, c = 1e-3f;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <class F>
static double time_ms(F f)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[2];
}

int main()
{
    float* out;
    CK(hipMalloc(&out, 4096));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 20000;
    const int grid = cus * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    // reference: plain v_fma_f32 at 64 per iteration; its known cost (2.26 cycles per wave instruction at 8 waves per SIMD, tools/valu_mb.hip)
    // is not assumed here: the ratio to it is what is reported
    const double t_fma = time_ms([&] { fma_loop<<<grid, 256>>>(out, iters); });
    const double per_simd = 8.0 * 64.0 * iters;   // wave instructions per SIMD
    printf("v_fma_f32           %8.3f ms  (%.3f ns per wave instruction per SIMD)\n", t_fma, t_fma * 1e6 / per_simd);
    const char* names[] = {"banks 0 0 0", "banks 0 2 0", "banks 0 2 2", "banks 0 0 2", "banks 2 2 2", "same pair x3", "banks 0 2 2 (B == C)"};
    double t[7];
    t[0] = time_ms([&] { pk_loop<0><<<grid, 256>>>(out, iters); });
    t[1] = time_ms([&] { pk_loop<1><<<grid, 256>>>(out, iters); });
    t[2] = time_ms([&] { pk_loop<2><<<grid, 256>>>(out, iters); });
    t[3] = time_ms([&] { pk_loop<3><<<grid, 256>>>(out, iters); });
    t[4] = time_ms([&] { pk_loop<4><<<grid, 256>>>(out, iters); });
    t[5] = time_ms([&] { pk_loop<5><<<grid, 256>>>(out, iters); });
    t[6] = time_ms([&] { pk_loop<6><<<grid, 256>>>(out, iters); });
    for (int v = 0; v < 7; ++v)
        printf("v_pk_fma_f32 %-24s %8.3f ms  = %.3f x v_fma_f32  (%.3f ns per wave instruction per SIMD)\n", names[v], t[v], t[v] / t_fma, t[v] * 1e6 / per_simd);
    return 0;
}

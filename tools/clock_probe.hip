// tools/clock_probe.hip — what nbk::clock_stamp sees: which XCD each workgroup of a stamp launch lands on (idle GPU, and queued
// behind a kernel that fills every CU), whether s_memtime of different XCDs agree, what s_memrealtime does across XCDs, and the
// shader clock under a packed-FMA load from stamp pairs. Developer tool behind bench.py's roofline.clock fields.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I n-bodysimulation_amd/csrc -I include tools/clock_probe.hip -o build/clock_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nbody_kernels.hip.h"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                              \
        }                                                                                         \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// every CU busy with packed FMAs for `iters` x 64 instructions per wave
__global__ void __launch_bounds__(256) busy(float* out, int iters)
{
    f2 a = {(float)threadIdx.x, 1.0f}, b = {1.000001f, 0.999999f}, c = {1e-6f, -1e-6f};
    f2 a1 = a + 1.0f, a2 = a + 2.0f, a3 = a + 3.0f;
    if (blockIdx.x >= gridDim.x - 64) iters *= 3;   // an uneven tail: 64 workgroups keep a few CUs busy while the others idle
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            a = a * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a.x + a.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y;
}

__global__ void __launch_bounds__(64) stamp_n(nbk::ClockStamp* out)
{
    if (threadIdx.x != 0) return;
    nbk::ClockStamp s;
    s.cycles = __builtin_amdgcn_s_memtime();
    s.ticks = __builtin_amdgcn_s_memrealtime();
    s.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // the whole register
    s.hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    out[blockIdx.x] = s;
}

int main(int argc, char** argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 64;
    const bool verbose = argc > 2;
    nbk::ClockStamp *h = nullptr, *d = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&h), 4 * (size_t)wgs * sizeof(nbk::ClockStamp), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0));
    float* out = nullptr;
    CK(hipMalloc(&out, 4096 * 256 * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto show = [&](const char* what, const nbk::ClockStamp* r) {
        printf("%s\n", what);
        if (!verbose) return;
        for (int k = 0; k < wgs; ++k)
            printf("  wg %2d  xcc_reg 0x%08x (id %u)  hw_id 0x%08x (cu %u se %u)  cycles %llu  ticks %llu\n", k, r[k].xcc, r[k].xcc & 15u, r[k].hw_id,
                   (r[k].hw_id >> 8) & 15u, (r[k].hw_id >> 13) & 7u, r[k].cycles, r[k].ticks);
    };
    // idle GPU
    stamp_n<<<wgs, 64, 0, st>>>(d);
    CK(hipStreamSynchronize(st));
    show("idle GPU, one stamp launch:", h);
    // queued behind and in front of a kernel that fills the chip for ~20 ms
    for (int rep = 0; rep < 3; ++rep) {
        busy<<<4096, 256, 0, st>>>(out, 40000);   // warm
        stamp_n<<<wgs, 64, 0, st>>>(d);
        busy<<<4096, 256, 0, st>>>(out, 40000);
        stamp_n<<<wgs, 64, 0, st>>>(d + wgs);
        CK(hipStreamSynchronize(st));
        char name[96];
        snprintf(name, sizeof name, "rep %d: stamp in front of a chip-filling launch:", rep);
        show(name, h);
        snprintf(name, sizeof name, "rep %d: stamp behind it:", rep);
        show(name, h + wgs);
        // pair by CU (xcc, se, sh, cu): the counters are per CU. Per XCD: matched CUs, and the spread of their cycle counts
        // (all equal = the counter runs whether or not the CU has work; smaller on CUs that idled in the tail = gated)
        for (unsigned x = 0; x < 16; ++x) {
            int m = 0;
            double cmin = 0, cmax = 0, tsum = 0, csum = 0;
            for (int k = 0; k < wgs; ++k) {
                if ((h[k].xcc & 15u) != x) continue;
                const unsigned key = h[k].hw_id & 0xff00u;
                bool dup = false;
                for (int q = 0; q < k; ++q) if ((h[q].xcc & 15u) == x && (h[q].hw_id & 0xff00u) == key) dup = true;
                if (dup) continue;
                for (int e = 0; e < wgs; ++e) {
                    if ((h[wgs + e].xcc & 15u) != x || (h[wgs + e].hw_id & 0xff00u) != key) continue;
                    const double dc = (double)(h[wgs + e].cycles - h[k].cycles), dt = (double)(h[wgs + e].ticks - h[k].ticks);
                    if (!m || dc < cmin) cmin = dc;
                    if (!m || dc > cmax) cmax = dc;
                    csum += dc; tsum += dt; ++m;
                    break;
                }
            }
            if (m) printf("  xcd %u: %2d CUs matched  d_cycles min %.0f max %.0f (spread %.4f %%)  -> %.1f MHz (mean), %.1f MHz (max)  %.3f ms\n", x, m, cmin, cmax,
                          (cmax - cmin) / cmax * 100.0, csum / tsum * 100.0, cmax / (tsum / m) * 100.0, tsum / m * 1e-5);
        }
    }
    return 0;
}

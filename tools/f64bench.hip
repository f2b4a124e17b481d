// tools/f64bench.hip — A/B of the general and the square-only build of the double-precision symmetric kernel at N bodies.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I n-bodysimulation_amd/csrc tools/f64bench.hip -o build/f64bench && build/f64bench [N] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nbody_kernels.hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
template <class F> static float median_ms(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 262144, reps = argc > 2 ? atoi(argv[2]) : 5;
    constexpr int W = 4, BPL = 6, B = 64 * W * BPL;
    const int nb = (n + B - 1) / B, grid = nb * (nb - 1) / 2 + nb;
    std::vector<double4> hx(n);
    unsigned long long s = 88172645463325252ull;
    auto u = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; ++i) hx[i] = make_double4(2 * u() - 1, 2 * u() - 1, 2 * u() - 1, 1.0 / n);
    double4 *dx, *slabs, *o1, *o2;
    CK(hipMalloc(&dx, (size_t)n * 32)); CK(hipMalloc(&slabs, (size_t)nb * n * 32)); CK(hipMalloc(&o1, (size_t)n * 32)); CK(hipMalloc(&o2, (size_t)n * 32));
    CK(hipMemcpy(dx, hx.data(), (size_t)n * 32, hipMemcpyHostToDevice));
    nbk::SymParamsF64 sp{};
    sp.x = dx; sp.slabs_i = slabs; sp.slabs_j = slabs; sp.ni = n; sp.nj = n; sp.nbi = nb; sp.nbj = nb; sp.stride_i = n; sp.stride_j = n; sp.eps2 = 0.002;
    auto general = [&] { nbk::force_sym<nbk::SymF64<BPL>, W><<<grid, 64 * W>>>(sp); };
    auto square = [&] { nbk::force_sym_square<nbk::SymF64<BPL>, W><<<grid, 64 * W>>>(sp); };
    general(); CK(hipDeviceSynchronize());
    std::vector<double4> a((size_t)n), b((size_t)n);
    CK(hipMemcpy(a.data(), slabs, (size_t)n * 32, hipMemcpyDeviceToHost));     // slab 0 is enough for a bitwise comparison
    CK(hipMemset(slabs, 0, (size_t)n * 32));
    square(); CK(hipDeviceSynchronize());
    CK(hipMemcpy(b.data(), slabs, (size_t)n * 32, hipMemcpyDeviceToHost));
    long diff = 0;
    for (int i = 0; i < n; ++i) diff += (a[i].x != b[i].x) || (a[i].y != b[i].y) || (a[i].z != b[i].z);
    for (int r = 0; r < 3; ++r)
        printf("N=%d f64 (4,6): general %.3f ms, square-only %.3f ms (slab 0 differs in %ld bodies)\n", n, median_ms(general, reps), median_ms(square, reps), diff);
    return 0;
}

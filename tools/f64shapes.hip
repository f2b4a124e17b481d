// tools/f64shapes.hip — the double-precision symmetric kernel at other block shapes (bodies per lane x waves per workgroup), one process,
// interleaved: is there a shape beyond (4,6) worth building into the library?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I n-bodysimulation_amd/csrc tools/f64shapes.hip -o build/f64shapes && build/f64shapes [N] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "nbody_kernels.hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 262144, reps = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<double4> hx(n);
    unsigned long long s = 88172645463325252ull;
    auto u = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; ++i) hx[i] = make_double4(2 * u() - 1, 2 * u() - 1, 2 * u() - 1, (0.5 + u()) / n);   // unequal masses: the general path
    double4 *dx, *slabs;
    const int nb_max = (n + 511) / 512;
    CK(hipMalloc(&dx, (size_t)n * 32)); CK(hipMalloc(&slabs, (size_t)nb_max * n * 32));
    CK(hipMemcpy(dx, hx.data(), (size_t)n * 32, hipMemcpyHostToDevice));
    struct Var { std::string name; std::function<void()> run; };
    std::vector<Var> vars;
    auto add = [&](auto tag_bpl, auto tag_w) {
        constexpr int BPL = decltype(tag_bpl)::value, W = decltype(tag_w)::value, B = 64 * W * BPL;
        const int nb = (n + B - 1) / B, grid = nb * (nb - 1) / 2 + nb;
        nbk::SymParamsF64 sp{};
        sp.x = dx; sp.slabs_i = slabs; sp.slabs_j = slabs; sp.ni = n; sp.nj = n; sp.nbi = nb; sp.nbj = nb; sp.stride_i = n; sp.stride_j = n; sp.eps2 = 0.002;
        vars.push_back({"bpl " + std::to_string(BPL) + " waves " + std::to_string(W) + " (block " + std::to_string(B) + ", " + std::to_string(nb) + " slabs)",
                        [=] { nbk::force_sym<nbk::SymF64<BPL>, W><<<grid, 64 * W>>>(sp); }});
    };
    add(std::integral_constant<int, 6>{}, std::integral_constant<int, 4>{});
    add(std::integral_constant<int, 7>{}, std::integral_constant<int, 4>{});
    add(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
    add(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
    add(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
    add(std::integral_constant<int, 6>{}, std::integral_constant<int, 2>{});
    add(std::integral_constant<int, 7>{}, std::integral_constant<int, 2>{});
    add(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vars) { v.run(); CK(hipDeviceSynchronize()); }
    for (int round = 0; round < 3; ++round)
        for (auto& v : vars) {
            std::vector<float> ts;
            for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); v.run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
            std::sort(ts.begin(), ts.end());
            const double ms = ts[ts.size() / 2];
            printf("N=%d f64 general path  %-44s %8.3f ms  %5.1f %% of 78.6 TF\n", n, v.name.c_str(), ms, 20.0 * n * (double)n / (ms * 1e-3) / 78.6e12 * 100);
        }
    return 0;
}

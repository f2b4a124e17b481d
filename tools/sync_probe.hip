// tools/sync_probe.hip — where does a synchronous one-step call go? (the reference's literal loop, main.cpp:146-156)
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I include tools/sync_probe.hip -o build/sync_probe -L n-bodysimulation_amd -lnbody_hip -Wl,-rpath,$PWD/n-bodysimulation_amd
//   build/sync_probe [N] [calls]
// Prints microseconds per step for: queued steps; simulate() per step; two steps per synchronised call (no copy-back);
// single-step calls queued without a sync (the copy-back launch alone); and the HIP floor: an empty kernel + stream sync,
// two empty kernels + sync, an empty kernel + spin on a host-mapped word written by the kernel.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nbody.h"

#define OK(x) do { int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "%s failed: %d %s\n", #x, rc_, nbody_last_error()); std::exit(1); } } while (0)
#define HOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void empty_kernel() {}
__global__ void flag_kernel(volatile unsigned* flag, unsigned v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = v; }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 8192;
    const int calls = argc > 2 ? std::atoi(argv[2]) : 2000;
    std::vector<nbody_float4> h(n);
    OK(nbody_fill_seeded(h.data(), n, 0, 12345));
    void *dx, *dv, *da;
    const size_t bytes = sizeof(nbody_float4) * (size_t)n;
    OK(nbody_malloc_device(&dx, bytes)); OK(nbody_malloc_device(&dv, bytes)); OK(nbody_malloc_device(&da, bytes));
    OK(nbody_memcpy_h2d(dx, h.data(), bytes));
    HOK(hipMemset(dv, 0, bytes)); HOK(hipMemset(da, 0, bytes));
    nbody_ctx* ctx = nullptr;
    OK(nbody_default_ctx(&ctx));
    auto X = (nbody_float4*)dx; auto V = (nbody_float4*)dv; auto A = (nbody_float4*)da;
    OK(nbody_step(ctx, X, A, V, n, 50)); OK(nbody_ctx_sync(ctx));
    OK(nbody_simulate_prepare(X, n));   // simulate()'s one-off work (near a switch-over size: the measurement of the decompositions)
    { int choice = -1; double ub = 0, ubest = 0; OK(nbody_ctx_autotuned(ctx, n, &choice, &ub, &ubest));
      std::printf("simulate() at N=%d: decomposition measured: choice %d (0 = built-in kept, -1 = not measured), built-in %.2f us, best %.2f us per queued step\n", n, choice, ub, ubest); }
    auto report = [&](const char* what, double secs, int steps) { std::printf("N=%d  %-58s %8.2f us/step\n", n, what, secs / steps * 1e6); };
    const int modes[] = {0, -1, 1, 0, -1, 1};
    for (int mode : modes) {
        OK(nbody_ctx_set_fused_inplace(ctx, mode));
        std::printf("-- fused in-place mode %d (%s)\n", mode, mode == 0 ? "never: two arrays + copy-back" : mode < 0 ? "default: the odd last step of a call" : "every fused step");
        OK(nbody_step(ctx, X, A, V, n, 10)); OK(nbody_ctx_sync(ctx));
        double t0 = now();
        OK(nbody_step(ctx, X, A, V, n, calls)); OK(nbody_ctx_sync(ctx));
        report("queued: nbody_step(steps=K) + one sync", now() - t0, calls);
        t0 = now();
        for (int k = 0; k < calls; ++k) OK(nbody_simulate(X, A, V, n));
        report("simulate() per step (the reference's loop)", now() - t0, calls);
        t0 = now();
        for (int k = 0; k < calls; ++k) { OK(nbody_step(ctx, X, A, V, n, 1)); OK(nbody_ctx_sync(ctx)); }
        report("nbody_step(1) + nbody_ctx_sync per step", now() - t0, calls);
        t0 = now();
        for (int k = 0; k < calls / 2; ++k) { OK(nbody_step(ctx, X, A, V, n, 2)); OK(nbody_ctx_sync(ctx)); }
        report("nbody_step(2) + sync per call", now() - t0, calls / 2 * 2);
        t0 = now();
        for (int k = 0; k < calls; ++k) OK(nbody_step(ctx, X, A, V, n, 1));
        OK(nbody_ctx_sync(ctx));
        report("nbody_step(1) x K queued, one sync", now() - t0, calls);
        unsigned long long fb = 0;
        OK(nbody_ctx_fused_inplace_stats(ctx, &fb));
        std::printf("   waves that took the fall-back path so far: %llu\n", fb);
    }
    hipStream_t st;
    HOK(hipStreamCreate(&st));
    for (int k = 0; k < 100; ++k) { empty_kernel<<<1, 64, 0, st>>>(); HOK(hipStreamSynchronize(st)); }
    double t0 = now();
    for (int k = 0; k < calls; ++k) { empty_kernel<<<1, 64, 0, st>>>(); HOK(hipStreamSynchronize(st)); }
    report("HIP floor: empty kernel + hipStreamSynchronize", now() - t0, calls);
    t0 = now();
    for (int k = 0; k < calls; ++k) { empty_kernel<<<1, 64, 0, st>>>(); empty_kernel<<<1, 64, 0, st>>>(); HOK(hipStreamSynchronize(st)); }
    report("HIP floor: two empty kernels + hipStreamSynchronize", now() - t0, calls);
    t0 = now();
    for (int k = 0; k < calls; ++k) { empty_kernel<<<256, 1024, 0, st>>>(); HOK(hipStreamSynchronize(st)); }
    report("HIP floor: empty kernel of 256 x 1024 threads + sync", now() - t0, calls);
    unsigned* flag = nullptr;
    HOK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *flag = 0;
    unsigned* dflag = nullptr;
    HOK(hipHostGetDevicePointer((void**)&dflag, flag, 0));
    t0 = now();
    for (int k = 1; k <= calls; ++k) {
        flag_kernel<<<1, 64, 0, st>>>(dflag, (unsigned)k);
        while (*(volatile unsigned*)flag != (unsigned)k) {}
    }
    report("HIP floor: kernel writes a host-mapped word, host spins on it", now() - t0, calls);
    HOK(hipStreamSynchronize(st));
    t0 = now();
    for (int k = 0; k < calls; ++k) { empty_kernel<<<1, 64, 0, st>>>(); }
    HOK(hipStreamSynchronize(st));
    report("HIP floor: empty kernels queued back to back", now() - t0, calls);
    return 0;
}

// compare_harness.cpp — compareHostToDevice of the reference (TestProject/validation.cpp:55-103) as
// a CHECKER program over oracle/validation_checker.hpp (the reference's validation.h:3-5 signatures): it drives the
// PRODUCT through the C-ABI (simulate()) and the CPU restatement (CPU_compute == oracle_step_inplace == the reference's
// CPU_compute) in lock-step, copies the device arrays back and applies the reference's verify_still_bodies rule to
// positions, velocities and accelerations.
// TEST INFRASTRUCTURE: lives under oracle/, links both libraries; nothing in the product links it.
//
//   compare_host_device [--n N] [--steps K (reference: 1000)] [--init libc|ref|plummer] [--seed S] [--jacobi]
// prints one JSON line with the three offender counts; exit code 0 unless a call failed.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "nbody.h"

struct float4 { float x, y, z, w; };
struct float3 { float x, y, z; };
#include "compat/validation.h"       // include/compat: verify_still_bodies / verify_equality4 / verify_equality3
#include "validation_checker.hpp"    // oracle/: bodyInteractions_CPU, CPU_compute, compareHostToDevice (validation.h:3-5)

int main(int argc, char** argv)
{
    int n = 1024, steps = 10;
    bool jacobi = false;
    unsigned long long seed = 12345;
    std::string init = "libc";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--n" && i + 1 < argc) n = std::atoi(argv[++i]);
        else if (a == "--steps" && i + 1 < argc) steps = std::atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
        else if (a == "--init" && i + 1 < argc) init = argv[++i];
        else if (a == "--jacobi") jacobi = true;
        else { std::cerr << "unknown option " << a << std::endl; return 2; }
    }
    std::vector<float4> bodies(n), velocity(n), accelerations(n);
    if (init == "libc") fill_with_random4(bodies.data(), n);
    else if (nbody_fill_seeded((nbody_float4*)bodies.data(), n, init == "plummer" ? 1 : 0, seed) != NBODY_OK) return 2;
    fill_with_zeroes4(velocity.data(), n);
    fill_with_zeroes4(accelerations.data(), n);
    const size_t size4 = sizeof(float4) * (size_t)n;
    float4 *d_bodies, *d_vel, *d_accel;
    if (nbody_malloc_device((void**)&d_bodies, size4) || nbody_malloc_device((void**)&d_vel, size4) ||
        nbody_malloc_device((void**)&d_accel, size4)) {
        std::cerr << nbody_last_error() << std::endl;
        return EXIT_FAILURE;
    }
    nbody_memcpy_h2d(d_bodies, bodies.data(), size4);
    nbody_memcpy_h2d(d_vel, velocity.data(), size4);
    nbody_memcpy_h2d(d_accel, accelerations.data(), size4);
    int bad[3] = {0, 0, 0};
    const int rc = compareHostToDevice_counts(d_bodies, d_accel, d_vel, bodies.data(), accelerations.data(), velocity.data(), n, steps, jacobi, bad);
    std::printf("{\"n\": %d, \"steps\": %d, \"cpu_order\": \"%s\", \"bad_positions\": %d, \"bad_velocities\": %d, \"bad_accelerations\": %d}\n", n,
                steps, jacobi ? "jacobi" : "inplace", bad[0], bad[1], bad[2]);
    nbody_free_device(d_bodies);
    nbody_free_device(d_vel);
    nbody_free_device(d_accel);
    return rc;
}

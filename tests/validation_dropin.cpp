// validation_dropin.cpp — a validation.cpp-style caller: includes the reference's header NAMES ("validation.h", "utils.h",
// "constants.h", "kernel.cuh" from include/compat) plus the checker header (oracle/validation_checker.hpp) and calls all six
// functions TestProject/validation.h:3-8 declares, with the reference's signatures.
//
//   validation_dropin --cpu-only OUT   no device: bodyInteractions_CPU, CPU_compute (K=3 steps on N_BODIES seeded bodies,
//                                      state written to OUT as raw float4[N] x,v,a), verify_* on host arrays
//   validation_dropin                  also compareHostToDevice(six float4*) — N_BODIES bodies, NBODY_COMPARE_STEPS steps
// Built by the tests with -DN_BODIES=... (the reference compiles its size in, constants.h:13).
#include <cstdio>
#include <cstring>
#include <vector>

struct float4 { float x, y, z, w; };
struct float3 { float x, y, z; };
#include "constants.h"
#include "kernel.cuh"
#include "utils.h"
#include "validation.h"
#include "validation_checker.hpp"

int main(int argc, char** argv)
{
    const bool cpu_only = argc > 1 && std::strcmp(argv[1], "--cpu-only") == 0;
    const int N = N_BODIES;
    std::vector<float4> bodies(N), velocity(N), accelerations(N);
    if (nbody_fill_seeded((nbody_float4*)bodies.data(), N, 0, 4242) != NBODY_OK) return 2;
    fill_with_zeroes4(velocity.data(), N);
    fill_with_zeroes4(accelerations.data(), N);

    // validation.h:3
    float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = bodyInteractions_CPU(bodies[0], bodies[1], acc);
    std::printf("pair %a %a %a %a\n", acc.x, acc.y, acc.z, acc.w);

    if (cpu_only) {
        // validation.h:4
        for (int k = 0; k < 3; ++k) CPU_compute(bodies.data(), accelerations.data(), velocity.data(), N);
        if (argc > 2) {
            FILE* f = std::fopen(argv[2], "wb");
            if (!f) return 2;
            std::fwrite(bodies.data(), sizeof(float4), N, f);
            std::fwrite(velocity.data(), sizeof(float4), N, f);
            std::fwrite(accelerations.data(), sizeof(float4), N, f);
            std::fclose(f);
        }
        // validation.h:6-8 on host arrays
        std::vector<float4> copy(bodies);
        std::vector<float3> a3(N), b3(N);
        for (int i = 0; i < N; ++i) a3[i] = b3[i] = float3{bodies[i].x, bodies[i].y, bodies[i].z};
        copy[0].x += 1.0f;
        b3[1].y += 1.0f;
        const int e4 = verify_equality4(copy.data(), bodies.data(), N);
        const int e3 = verify_equality3(a3.data(), b3.data(), N);
        copy[0].x = bodies[0].x * 1.5f + 1.0f;
        const int sb = verify_still_bodies(copy.data(), bodies.data(), N);
        std::printf("counts %d %d %d\n", e4, e3, sb);
        return (e4 == 1 && e3 == 1 && sb == 1) ? 0 : 1;
    }

    // validation.h:5 — device arrays prepared as main.cpp:275-283 does
    const size_t size4 = sizeof(float4) * (size_t)N;
    float4 *d_bodies = nullptr, *d_vel = nullptr, *d_accel = nullptr;
    if (nbody_malloc_device((void**)&d_bodies, size4) || nbody_malloc_device((void**)&d_vel, size4) ||
        nbody_malloc_device((void**)&d_accel, size4)) {
        std::fprintf(stderr, "%s\n", nbody_last_error());
        return 2;
    }
    nbody_memcpy_h2d(d_bodies, bodies.data(), size4);
    nbody_memcpy_h2d(d_vel, velocity.data(), size4);
    nbody_memcpy_h2d(d_accel, accelerations.data(), size4);
    const int rc = compareHostToDevice(d_bodies, d_accel, d_vel, bodies.data(), accelerations.data(), velocity.data());
    std::printf("compareHostToDevice rc %d\n", rc);
    nbody_free_device(d_bodies);
    nbody_free_device(d_vel);
    nbody_free_device(d_accel);
    return rc;
}

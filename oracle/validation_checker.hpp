// validation_checker.hpp — the CHECKER half of the reference's "validation.h" (TestProject/validation.h:3-5) with the
// reference's exact signatures:
//
//     float4 bodyInteractions_CPU(float4 bi, float4 bj, float4 ai);                       validation.cpp:9-24
//     void   CPU_compute(float4* gX, float4* gA, float4* gV, int N);                      validation.cpp:28-52
//     int    compareHostToDevice(float4* d_bodies, float4* d_accel, float4* d_vel,        validation.cpp:55-103
//                                float4* bodies, float4* accelerations, float4* velocity);
//
// TEST INFRASTRUCTURE ONLY. The product (include/, libnbody_hip.so) has no CPU force path on purpose, so these three
// declarations are NOT in include/compat/validation.h; a validation.cpp-style caller that needs them includes this header
// as well and links oracle/liboracle.so next to libnbody_hip.so:
//
//     #include <hip/hip_runtime.h>        // or any 16-byte {x,y,z,w} float4 / 12-byte float3
//     #include "validation.h"             // include/compat: verify_equality4 / verify_equality3 / verify_still_bodies
//     #include "validation_checker.hpp"   // oracle/: the three above
//
// The arithmetic is oracle/nbody_oracle.c (oracle_pair, oracle_step_inplace), which tests/test_oracle.py pins bit for bit
// to the reference's own compiled validation.cpp. compareHostToDevice drives the PRODUCT through simulate() (C-ABI) in
// lock-step with CPU_compute, exactly as the reference does: N_BODIES bodies, 1000 steps (validation.cpp:59,65), DT / EPS2
// of constants.h. Like the reference's, the CPU loop is the literal sequential in-place order.
#pragma once

#include <cstdlib>
#include <iostream>
#include <vector>

#include "nbody_compat.hpp"   // simulate(), verify_still_bodies(), DT, EPS2, N_BODIES, nbody_memcpy_d2h ...
#include "nbody_oracle.h"

#ifndef NBODY_COMPARE_STEPS
#define NBODY_COMPARE_STEPS 1000   // validation.cpp:65
#endif

namespace validation_checker {
static_assert(sizeof(ofloat4) == 16, "ofloat4 is the reference's float4 record");
template <class F4>
inline ofloat4* as_o(F4* p)
{
    static_assert(sizeof(F4) == sizeof(ofloat4), "float4 must be 16 bytes {x,y,z,w}");
    return reinterpret_cast<ofloat4*>(p);
}
}  // namespace validation_checker

// validation.cpp:9-24 (validation.h:3)
inline float4 bodyInteractions_CPU(float4 bi, float4 bj, float4 ai)
{
    const ofloat4 r = oracle_pair(*validation_checker::as_o(&bi), *validation_checker::as_o(&bj), *validation_checker::as_o(&ai), EPS2);
    float4 out = ai;
    out.x = r.x; out.y = r.y; out.z = r.z; out.w = r.w;
    return out;
}

// validation.cpp:28-52 (validation.h:4): one step on host arrays, sequential, IN PLACE
inline void CPU_compute(float4* gX, float4* gA, float4* gV, int N)
{
    oracle_step_inplace(validation_checker::as_o(gX), validation_checker::as_o(gA), validation_checker::as_o(gV), N, DT, EPS2);
}

// The same harness with the sizes as parameters and the three offender counts returned (what the checker program and the
// tests use); `jacobi` steps the CPU side in Jacobi order instead (what a race-free GPU step computes).
inline int compareHostToDevice_counts(float4* d_bodies, float4* d_accel, float4* d_vel, float4* bodies, float4* accelerations,
                                      float4* velocity, int N, int steps, bool jacobi, int bad[3])
{
    float4 *dToH_bodies = nullptr, *dToH_velocity = nullptr, *dToH_accelerations = nullptr;
    const size_t size4 = sizeof(float4) * (size_t)N;
    if (nbody_malloc_host((void**)&dToH_bodies, size4) || nbody_malloc_host((void**)&dToH_velocity, size4) ||   // validation.cpp:61-63
        nbody_malloc_host((void**)&dToH_accelerations, size4)) {
        std::cerr << nbody_last_error() << std::endl;
        return EXIT_FAILURE;
    }
    for (int i = 0; i < steps; i++) {
        try {
            simulate(d_bodies, d_accel, d_vel, N);                                                              // validation.cpp:67
        } catch (const std::exception& e) {
            std::cerr << e.what() << std::endl;
            return EXIT_FAILURE;
        }
        if (jacobi) oracle_step_jacobi(validation_checker::as_o(bodies), validation_checker::as_o(accelerations),
                                       validation_checker::as_o(velocity), N, DT, EPS2);
        else CPU_compute(bodies, accelerations, velocity, N);                                                   // validation.cpp:74
    }
    if (nbody_device_synchronize() != NBODY_OK) return EXIT_FAILURE;                                            // validation.cpp:77
    nbody_memcpy_d2h(dToH_bodies, d_bodies, size4);                                                             // validation.cpp:79-81
    nbody_memcpy_d2h(dToH_velocity, d_vel, size4);
    nbody_memcpy_d2h(dToH_accelerations, d_accel, size4);
    std::printf("Starting verification...\n");
    bad[0] = verify_still_bodies(dToH_bodies, bodies, N);                                                       // validation.cpp:84-86
    bad[1] = verify_still_bodies(dToH_velocity, velocity, N);
    bad[2] = verify_still_bodies(dToH_accelerations, accelerations, N);
    std::printf("Verification complete\n\n");
    nbody_free_host(dToH_bodies);                                                                               // validation.cpp:98-100
    nbody_free_host(dToH_accelerations);
    nbody_free_host(dToH_velocity);
    return 0;
}

// validation.cpp:55-103 (validation.h:5): N_BODIES bodies, 1000 lock-step steps, the 1 % rule on x, v and a
inline int compareHostToDevice(float4* d_bodies, float4* d_accel, float4* d_vel, float4* bodies, float4* accelerations,
                               float4* velocity)
{
    int bad[3] = {0, 0, 0};
    return compareHostToDevice_counts(d_bodies, d_accel, d_vel, bodies, accelerations, velocity, N_BODIES, NBODY_COMPARE_STEPS,
                                      false, bad);
}

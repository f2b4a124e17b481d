// A caller written the way the reference's headless main() is (TestProject/main.cpp:231-368), using the
// reference's header names. Built by tests/test_gpu_driver.py with -Iinclude/compat: if this compiles
// and runs unchanged against libnbody_hip.so, the boundary is a drop-in at source level.
#include <cstdio>
#include <cstdlib>
#include <iostream>

struct float4 { float x, y, z, w; };   // what <cuda_runtime.h> / <hip/hip_runtime.h> would provide
struct float3 { float x, y, z; };

#include "constants.h"
#include "kernel.cuh"
#include "utils.h"
#include "validation.h"

static int simulationLoopNoVisual(float4* d_bodies, float4* d_accel, float4* d_vel, int steps)
{
    int counter = 0;
    printf("Starting the simulation...\n");
    while (counter < steps) {
        try {
            simulate(d_bodies, d_accel, d_vel, N_BODIES);
        } catch (const std::exception& e) {
            std::cerr << e.what() << std::endl;
            return EXIT_FAILURE;
        }
        counter++;
    }
    printf("Simulation complete\n");
    return 0;
}

int main(int argc, char** argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 3;
    const int size4 = sizeof(float4) * N_BODIES;
    float4 *bodies, *velocity, *accelerations, *check;
    nbody_malloc_host((void**)&bodies, size4);
    nbody_malloc_host((void**)&velocity, size4);
    nbody_malloc_host((void**)&accelerations, size4);
    nbody_malloc_host((void**)&check, size4);
    fill_with_random4(bodies, N_BODIES);
    fill_with_zeroes4(velocity, N_BODIES);
    fill_with_zeroes4(accelerations, N_BODIES);

    float4 *d_bodies, *d_velocity, *d_accelerations;
    nbody_malloc_device((void**)&d_velocity, size4);
    nbody_malloc_device((void**)&d_accelerations, size4);
    nbody_malloc_device((void**)&d_bodies, size4);
    nbody_memcpy_h2d(d_velocity, velocity, size4);
    nbody_memcpy_h2d(d_accelerations, accelerations, size4);
    nbody_memcpy_h2d(d_bodies, bodies, size4);

    if (simulationLoopNoVisual(d_bodies, d_accelerations, d_velocity, steps) != 0) return EXIT_FAILURE;

    // every other declaration of utils.h / validation.h (utils.h:3-10, validation.h:6-8), once each
    print_device_prop();
    float3 v3[4], w3[4];
    fill_with_zeroes3(v3, 4);
    fill_with_zeroes3(w3, 4);
    w3[2].y = random_float(5.0f, 6.0f);
    const int off3 = verify_equality3(v3, w3, 4);
    float4 copy[4];
    copy_vector_bodies(bodies, copy, 4);
    printf("helpers: off3=%d copy_ok=%d body1=", off3, (int)(copy[3].w == bodies[3].w && copy[0].x == bodies[0].x));
    print_float4(bodies[1]);
    printf(" v3=");
    print_float3(w3[0]);
    printf("\n");

    nbody_memcpy_d2h(check, d_bodies, size4);
    const int moved = verify_equality4(check, bodies, N_BODIES);   // bodies that moved by more than 0.01
    printf("N_BODIES=%d DT=%g EPS2=%g steps=%d moved=%d body0=%.9g %.9g %.9g %.9g\n", N_BODIES, (double)DT, (double)EPS2, steps, moved,
           check[0].x, check[0].y, check[0].z, check[0].w);
    nbody_free_device(d_bodies);
    nbody_free_device(d_velocity);
    nbody_free_device(d_accelerations);
    nbody_free_host(bodies);
    nbody_free_host(velocity);
    nbody_free_host(accelerations);
    nbody_free_host(check);
    return 0;
}

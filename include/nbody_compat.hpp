// nbody_compat.hpp — source-level drop-in for the reference's C++ step/validate API on top of the
// C-ABI (include/nbody.h). Include it where the reference includes "kernel.cuh", "utils.h",
// "validation.h" and "constants.h"; `float4` must already be defined by the includer
// (<hip/hip_runtime.h>, or any 16-byte {x,y,z,w} float struct).
//
//   reference                                              here
//   void simulate(float4*, float4*, float4*, int)          same signature; throws std::runtime_error
//     TestProject/kernel.cuh:2, kernel.cu:628-645            (the reference throws it at kernel.cu:633-641)
//   random_float, fill_with_zeroes3/4, fill_with_random4,  same signatures (float3 = any 12-byte {x,y,z} struct)
//   print_float4/3, print_device_prop, copy_vector_bodies
//     utils.h:3-10
//   verify_still_bodies / verify_equality4 / _equality3    same signatures; print one summary line and
//     validation.h:6-8                                     (new) return the number of offending bodies
//   DT EPS2 N_BODIES MAX_X.. MIN_W MAX_W   constants.h     same macro names, only if not yet defined
//
// bodyInteractions_CPU / CPU_compute / compareHostToDevice (validation.h:3-5) are the *checker*: the CPU
// step the GPU result is judged against. They live with the oracle (oracle/nbody_oracle.c,
// oracle/compare_harness.cpp) on purpose — the product has no CPU force path — and are the only
// declarations of the reference's utils.h / validation.h that this header does not provide.
#pragma once

#include <cstdio>
#include <stdexcept>

#include "nbody.h"

// constants.h:11-26 — compile-time defaults of the reference; the engine itself takes all of
// these at run time.
#ifndef N_BODIES
#define N_BODIES 8192
#endif
#ifndef MAX_X
#define MAX_X 100000.0f
#define MAX_Y 100000.0f
#define MAX_Z 100000.0f
#define MIN_W 100000.0f
#define MAX_W 1000000000.0f
#endif
#ifndef EPS2
#define EPS2 0.002f
#endif
#ifndef DT
#define DT 0.1f
#endif

namespace nbody_compat {
template <class F4>
inline nbody_float4* as_nb(F4* p)
{
    static_assert(sizeof(F4) == sizeof(nbody_float4), "float4 must be 16 bytes {x,y,z,w}");
    return reinterpret_cast<nbody_float4*>(p);
}
template <class F3>
inline nbody_float3* as_nb3(F3* p)
{
    static_assert(sizeof(F3) == sizeof(nbody_float3), "float3 must be 12 bytes {x,y,z}");
    return reinterpret_cast<nbody_float3*>(p);
}
inline void check(int rc)
{
    if (rc != NBODY_OK) throw std::runtime_error(nbody_last_error());
}
}  // namespace nbody_compat

// kernel.cuh:2
template <class F4>
inline void simulate(F4* d_bodies, F4* d_accelerations, F4* d_velocity, int N)
{
    nbody_compat::check(nbody_simulate(nbody_compat::as_nb(d_bodies), nbody_compat::as_nb(d_accelerations),
                                       nbody_compat::as_nb(d_velocity), N));
}

// utils.h:3-10
inline float random_float(float min, float max) { return nbody_random_float(min, max); }
template <class F3>
inline void fill_with_zeroes3(F3 v[], int N) { nbody_fill_with_zeroes3(nbody_compat::as_nb3(v), N); }
template <class F4>
inline void fill_with_random4(F4 v[], int N) { nbody_fill_with_random4(nbody_compat::as_nb(v), N); }
template <class F4>
inline void fill_with_zeroes4(F4 v[], int N) { nbody_fill_with_zeroes4(nbody_compat::as_nb(v), N); }
template <class F4>
inline void print_float4(F4 v) { std::printf("%f %f %f %f", v.x, v.y, v.z, v.w); }   // utils.cpp:40-42
template <class F3>
inline void print_float3(F3 v) { std::printf("%f %f %f", v.x, v.y, v.z); }            // utils.cpp:45-47
inline void print_device_prop() { nbody_compat::check(nbody_print_device_prop()); }  // utils.cpp:49-68
template <class F4>
inline void copy_vector_bodies(F4 in[], F4 out[], int N)                             // utils.cpp:70-74
{
    for (int i = 0; i < N; i++) out[i] = in[i];
}

// validation.h:6-8
template <class F4>
inline int verify_still_bodies(F4 v[], F4 x[], int N)
{
    const int bad = nbody_verify_still_bodies(nbody_compat::as_nb(v), nbody_compat::as_nb(x), N);
    std::printf("verify_still_bodies: %d of %d bodies outside 1 %%\n", bad, N);
    return bad;
}
template <class F4>
inline int verify_equality4(F4 v[], F4 x[], int N)
{
    const int bad = nbody_verify_equality4(nbody_compat::as_nb(v), nbody_compat::as_nb(x), N);
    std::printf("verify_equality4: %d of %d bodies differ by more than 0.01\n", bad, N);
    return bad;
}
template <class F3>
inline int verify_equality3(F3 v[], F3 x[], int N)
{
    const int bad = nbody_verify_equality3(nbody_compat::as_nb3(v), nbody_compat::as_nb3(x), N);
    std::printf("verify_equality3: %d of %d bodies differ by more than 0.01\n", bad, N);
    return bad;
}

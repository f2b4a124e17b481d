// Forwarding header for the reference's "validation.h" (TestProject/validation.h:6-8): verify_equality4,
// verify_equality3, verify_still_bodies. The other three declarations of that header
// (bodyInteractions_CPU, CPU_compute, compareHostToDevice, validation.h:3-5) ARE the CPU checker the GPU
// step is judged against; by design they are not in the product (no CPU force path): they live under
// oracle/ (nbody_oracle.c, compare_harness.cpp). See ../nbody_compat.hpp.
#pragma once
#include "../nbody_compat.hpp"

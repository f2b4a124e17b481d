// Forwarding header for the reference's "utils.h" (TestProject/utils.h:3-10): random_float,
// fill_with_zeroes3, fill_with_zeroes4, fill_with_random4, print_float4, print_float3,
// print_device_prop, copy_vector_bodies — every declaration. See ../nbody_compat.hpp.
#pragma once
#include "../nbody_compat.hpp"

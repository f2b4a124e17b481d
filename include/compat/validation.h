// Forwarding header for the reference's "validation.h" (verify_still_bodies, verify_equality4:
// TestProject/validation.h:6,8). CPU_compute / compareHostToDevice are the checker and live under
// oracle/ (oracle/compare_harness.cpp), not in the product. See ../nbody_compat.hpp.
#pragma once
#include "../nbody_compat.hpp"

// Forwarding header for the reference's "utils.h" (fill_with_random4, fill_with_zeroes4:
// TestProject/utils.h:5-6). See ../nbody_compat.hpp.
#pragma once
#include "../nbody_compat.hpp"

// Forwarding header: a translation unit written against the reference's "kernel.cuh"
// (TestProject/kernel.cuh:2, simulate()) picks up the MI355X engine when include/compat is on
// its include path. float4 must be defined before this point (e.g. <hip/hip_runtime.h>).
#pragma once
#include "../nbody_compat.hpp"

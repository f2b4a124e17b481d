// Forwarding header for the reference's "constants.h": N_BODIES, MAX_X/Y/Z, MIN_W, MAX_W, EPS2, DT
// come from ../nbody_compat.hpp (same names and values as TestProject/constants.h:13-26; the engine
// itself takes them at run time).
#pragma once
#include "../nbody_compat.hpp"

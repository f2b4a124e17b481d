// oracle/ref_dt001.cpp — TEST INFRASTRUCTURE. The reference's CPU path at the benchmark's time step.
//
// BASELINE.json configs[1], [2] and [4] run dt = 0.01; the reference's TestProject/constants.h:26 compiles DT 0.1f in
// (the 0.01 comes from its older snapshot, Sim-Without-OpenGL-Integration/constants.h:14-15). This translation unit
// compiles the reference's own validation.cpp WHERE IT LIES (found through -I$(REF); nothing is copied) with exactly one
// change: the DT macro. constants.h is `#pragma once`, so validation.cpp:6's own `#include "constants.h"` is a no-op
// after the first inclusion here and the redefinition below is what validation.cpp:43-49 sees. Every other macro
// (EPS2 0.002f, N_BODIES, ...) is the reference's. Built by oracle/Makefile into oracle/_ref/libref_cpu_dt001.so with
// the same flags and export map as _ref/libref_cpu.so; used by tests/golden/make_golden.py and tests/test_oracle.py.
#include "constants.h"   // the reference's (via -I$(REF))
#undef DT
#define DT 0.01f
#include "validation.cpp"   // the reference's (via -I$(REF))

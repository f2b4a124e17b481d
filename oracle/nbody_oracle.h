/*
 * nbody_oracle.h — CPU restatement of the reference's all-pairs step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (include/, the
 * n-bodysimulation_amd package, libnbody_hip.so) may include, link or call
 * this.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg.
 *
 * Parity pin: `oracle_step_inplace` is checked BIT-EXACT against the
 * reference's own CPU_compute (TestProject/validation.cpp:28-52) compiled
 * from the sources where they lie into oracle/_ref/libref_cpu.so (see
 * oracle/Makefile), and against the golden vectors that build produced
 * (tests/golden/, generator tests/golden/make_golden.py).
 */
#ifndef NBODY_ORACLE_H
#define NBODY_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Same 16-byte AoS record as the reference's float4 {x,y,z,w=mass}
 * (TestProject/main.cpp:232-241). */
typedef struct { float x, y, z, w; } ofloat4;
typedef struct { double x, y, z, w; } odouble4;

/* validation.cpp:9-24 */
ofloat4 oracle_pair(ofloat4 bi, ofloat4 bj, ofloat4 ai, float eps2);

/* validation.cpp:28-52, literal: sequential, IN PLACE (body i sees the
 * already-advanced positions of bodies j<i). */
void oracle_step_inplace(ofloat4 *X, ofloat4 *A, ofloat4 *V, int n, float dt, float eps2);

/* Same arithmetic, but every force is taken from the positions at the start
 * of the step (what a race-free GPU step computes). OpenMP-parallel over i. */
void oracle_step_jacobi(ofloat4 *X, ofloat4 *A, ofloat4 *V, int n, float dt, float eps2);

/* Jacobi step with pair terms evaluated and summed in double from the fp32
 * inputs, rounded to fp32 once: the "true" acceleration yardstick. */
void oracle_step_jacobi_f64acc(ofloat4 *X, ofloat4 *A, ofloat4 *V, int n, float dt, float eps2);

/* Accelerations only, targets [i0,i1) against sources [j0,j1) of X; j==i is
 * skipped.  A_out has (i1-i0) entries.  mode 0: fp32 sequential (as
 * oracle_pair); mode 1: fp64 accumulate. */
void oracle_accel_range(const ofloat4 *X, ofloat4 *A_out, int i0, int i1, int j0, int j1,
                        float eps2, int mode);

/* Integrate, validation.cpp:43-49: v += (0.5f*dt)*a ; x += dt*v (xyz only). */
void oracle_integrate(ofloat4 *X, ofloat4 *V, const ofloat4 *A, int n, float dt);

/* All-double state (the build's own fp64 variant; no reference analogue). */
void oracle_step_jacobi_f64(odouble4 *X, odouble4 *A, odouble4 *V, int n, double dt, double eps2);
void oracle_accel_range_f64(const odouble4 *X, odouble4 *A_out, int i0, int i1, int j0, int j1, double eps2);

/* One step of the OLDER snapshot (Sim-Without-OpenGL-Integration/kernel.cu:5-82): float3 velocity,
 * double-literal DT = 0.01 / EPS2 = 0.002, Jacobi order. V3 = N packed float3. */
void oracle_step_legacy(ofloat4 *X, float *V3, int n);

/* utils.cpp:6,30-37 (libc rand(), 4 draws per body x,y,z,w) and :19-27. */
void oracle_fill_with_random4(ofloat4 *v, int n);
void oracle_fill_with_zeroes4(ofloat4 *v, int n);

/* validation.cpp:143-164 / 106-122, returning the number of bodies the
 * reference would have printed "Problem at body" for. */
int oracle_verify_still_bodies(const ofloat4 *v, const ofloat4 *x, int n);
int oracle_verify_equality4(const ofloat4 *v, const ofloat4 *x, int n);

int oracle_max_threads(void);
void oracle_set_threads(int t);

#ifdef __cplusplus
}
#endif
#endif

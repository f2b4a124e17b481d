/*
 * nbody_oracle.c — CPU restatement of the reference's all-pairs N-body step.
 *
 * TEST INFRASTRUCTURE ONLY (see nbody_oracle.h).  The product never routes
 * through this file.
 *
 * Every function names the reference lines it restates
 * (paths relative to the reference's TestProject/ directory).
 *
 * Build flags matter: -ffp-contract=off (the reference's MSVC /O2 x64 build and
 * a plain `g++ -O2` x86-64 build never fuse mul+add; see oracle/Makefile) and
 * no -ffast-math, so every operation is an individually rounded IEEE fp32 op in
 * the order written.
 */
#include "nbody_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IB 16 /* targets per block in the blocked Jacobi kernels */

/* validation.cpp:9-24 — one softened pair. r = bj - bi; d = r.r + EPS2 summed
 * left to right; inv = 1/sqrtf(d*d*d); ai.xyz += r * (bj.w * inv). */
ofloat4 oracle_pair(ofloat4 bi, ofloat4 bj, ofloat4 ai, float eps2)
{
    float rx = bj.x - bi.x;
    float ry = bj.y - bi.y;
    float rz = bj.z - bi.z;
    float d = rx * rx + ry * ry + rz * rz + eps2;
    float denom = 1.0f / sqrtf(d * d * d);
    float s = bj.w * denom;
    ai.x += rx * s;
    ai.y += ry * s;
    ai.z += rz * s;
    return ai;
}

/* validation.cpp:43-49 for one body. (0.5f*dt) is formed first, as the
 * reference's `0.5f * DT * a` parses. */
static inline void integrate_one(ofloat4 *x, ofloat4 *v, ofloat4 a, float dt)
{
    float hdt = 0.5f * dt;
    v->x += hdt * a.x;
    v->y += hdt * a.y;
    v->z += hdt * a.z;
    x->x += dt * v->x;
    x->y += dt * v->y;
    x->z += dt * v->z;
}

/* validation.cpp:28-52, literal order: the integrate of body i happens inside
 * the i loop, so body i+1 reads body i's advanced position. Serial on purpose
 * (the reference build has no OpenMP: TestProject.vcxproj:64-86). */
void oracle_step_inplace(ofloat4 *X, ofloat4 *A, ofloat4 *V, int n, float dt, float eps2)
{
    for (int i = 0; i < n; ++i) {
        ofloat4 body = X[i];
        ofloat4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = 0; j < n; ++j)
            if (i != j)
                acc = oracle_pair(body, X[j], acc, eps2);
        A[i] = acc;
        integrate_one(&X[i], &V[i], acc, dt);
    }
}

/*
 * Blocked fp32 acceleration kernel: IB targets advance together through the
 * sources in index order. Each target's sum is still the plain sequential
 * j = j0..j1-1 sum of oracle_pair terms (j == i skipped), so the result is
 * bit-identical to the scalar loop; the block only lets the compiler put the
 * IB independent targets in SIMD lanes.
 */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("default", "avx2", "avx512f")))
#endif
static void accel_block_f32(const ofloat4 *X, int ibase, int nb, int j0, int j1, float eps2,
                            float *ax, float *ay, float *az)
{
    float xi[IB], yi[IB], zi[IB];
    int idx[IB];
    for (int t = 0; t < IB; ++t) {
        int i = ibase + (t < nb ? t : 0);
        xi[t] = X[i].x; yi[t] = X[i].y; zi[t] = X[i].z;
        idx[t] = (t < nb) ? i : -1;
        ax[t] = 0.0f; ay[t] = 0.0f; az[t] = 0.0f;
    }
    /* Sources outside [ibase, ibase+IB) can never be a target of this block, so
     * the j == i test is only needed in the middle segment. */
    const int m0 = ibase < j0 ? j0 : (ibase > j1 ? j1 : ibase);
    const int m1 = ibase + IB < m0 ? m0 : (ibase + IB > j1 ? j1 : ibase + IB);
    for (int seg = 0; seg < 3; ++seg) {
        const int ja = seg == 0 ? j0 : (seg == 1 ? m0 : m1);
        const int jb = seg == 0 ? m0 : (seg == 1 ? m1 : j1);
        if (seg != 1) {
            for (int j = ja; j < jb; ++j) {
                const float xj = X[j].x, yj = X[j].y, zj = X[j].z, mj = X[j].w;
#pragma omp simd
                for (int t = 0; t < IB; ++t) {
                    float rx = xj - xi[t];
                    float ry = yj - yi[t];
                    float rz = zj - zi[t];
                    float d = rx * rx + ry * ry + rz * rz + eps2;
                    float denom = 1.0f / sqrtf(d * d * d);
                    float s = mj * denom;
                    ax[t] = ax[t] + rx * s;
                    ay[t] = ay[t] + ry * s;
                    az[t] = az[t] + rz * s;
                }
            }
        } else {
            for (int j = ja; j < jb; ++j) {
                const float xj = X[j].x, yj = X[j].y, zj = X[j].z, mj = X[j].w;
                for (int t = 0; t < IB; ++t) {
                    if (idx[t] == j) continue;
                    float rx = xj - xi[t];
                    float ry = yj - yi[t];
                    float rz = zj - zi[t];
                    float d = rx * rx + ry * ry + rz * rz + eps2;
                    float denom = 1.0f / sqrtf(d * d * d);
                    float s = mj * denom;
                    ax[t] = ax[t] + rx * s;
                    ay[t] = ay[t] + ry * s;
                    az[t] = az[t] + rz * s;
                }
            }
        }
    }
}

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("default", "avx2", "avx512f")))
#endif
static void accel_block_f64(const ofloat4 *X, int ibase, int nb, int j0, int j1, float eps2,
                            double *ax, double *ay, double *az)
{
    double xi[IB], yi[IB], zi[IB];
    int idx[IB];
    const double e2 = (double)eps2;
    for (int t = 0; t < IB; ++t) {
        int i = ibase + (t < nb ? t : 0);
        xi[t] = X[i].x; yi[t] = X[i].y; zi[t] = X[i].z;
        idx[t] = (t < nb) ? i : -1;
        ax[t] = 0.0; ay[t] = 0.0; az[t] = 0.0;
    }
    for (int j = j0; j < j1; ++j) {
        const double xj = X[j].x, yj = X[j].y, zj = X[j].z, mj = X[j].w;
#pragma omp simd
        for (int t = 0; t < IB; ++t) {
            double rx = xj - xi[t];
            double ry = yj - yi[t];
            double rz = zj - zi[t];
            double d = rx * rx + ry * ry + rz * rz + e2;
            double s = mj / (d * sqrt(d));
            int skip = (idx[t] == j);
            ax[t] = skip ? ax[t] : ax[t] + rx * s;
            ay[t] = skip ? ay[t] : ay[t] + ry * s;
            az[t] = skip ? az[t] : az[t] + rz * s;
        }
    }
}

void oracle_accel_range(const ofloat4 *X, ofloat4 *A_out, int i0, int i1, int j0, int j1,
                        float eps2, int mode)
{
    const int nblk = (i1 - i0 + IB - 1) / IB;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < nblk; ++b) {
        const int ibase = i0 + b * IB;
        const int nb = (i1 - ibase < IB) ? (i1 - ibase) : IB;
        if (mode == 0) {
            float ax[IB], ay[IB], az[IB];
            accel_block_f32(X, ibase, nb, j0, j1, eps2, ax, ay, az);
            for (int t = 0; t < nb; ++t) {
                ofloat4 a = {ax[t], ay[t], az[t], 0.0f};
                A_out[ibase - i0 + t] = a;
            }
        } else {
            double ax[IB], ay[IB], az[IB];
            accel_block_f64(X, ibase, nb, j0, j1, eps2, ax, ay, az);
            for (int t = 0; t < nb; ++t) {
                ofloat4 a = {(float)ax[t], (float)ay[t], (float)az[t], 0.0f};
                A_out[ibase - i0 + t] = a;
            }
        }
    }
}

void oracle_integrate(ofloat4 *X, ofloat4 *V, const ofloat4 *A, int n, float dt)
{
    for (int i = 0; i < n; ++i)
        integrate_one(&X[i], &V[i], A[i], dt);
}

/* Jacobi: validation.cpp:28-52 with the force loop reading the start-of-step
 * positions for every body (forces first, then integrate everybody). */
void oracle_step_jacobi(ofloat4 *X, ofloat4 *A, ofloat4 *V, int n, float dt, float eps2)
{
    oracle_accel_range(X, A, 0, n, 0, n, eps2, 0);
    oracle_integrate(X, V, A, n, dt);
}

void oracle_step_jacobi_f64acc(ofloat4 *X, ofloat4 *A, ofloat4 *V, int n, float dt, float eps2)
{
    oracle_accel_range(X, A, 0, n, 0, n, eps2, 1);
    oracle_integrate(X, V, A, n, dt);
}

/* The build's own fp64 variant (BASELINE.json configs[4]); the reference has no
 * double path. Same formula, all-double state. */
void oracle_step_jacobi_f64(odouble4 *X, odouble4 *A, odouble4 *V, int n, double dt, double eps2)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const double xi = X[i].x, yi = X[i].y, zi = X[i].z;
        double ax = 0.0, ay = 0.0, az = 0.0;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            double rx = X[j].x - xi, ry = X[j].y - yi, rz = X[j].z - zi;
            double d = rx * rx + ry * ry + rz * rz + eps2;
            double s = X[j].w * (1.0 / sqrt(d * d * d));
            ax += rx * s; ay += ry * s; az += rz * s;
        }
        A[i].x = ax; A[i].y = ay; A[i].z = az; A[i].w = 0.0;
    }
    const double hdt = 0.5 * dt;
    for (int i = 0; i < n; ++i) {
        V[i].x += hdt * A[i].x; V[i].y += hdt * A[i].y; V[i].z += hdt * A[i].z;
        X[i].x += dt * V[i].x;  X[i].y += dt * V[i].y;  X[i].z += dt * V[i].z;
    }
}

/* Accelerations only, all-double: targets [i0,i1) against sources [j0,j1), j == i skipped. Same pair
 * expression as oracle_step_jacobi_f64 (the checker of the build's own fp64 kernel at sizes where a
 * whole CPU step would take minutes: sampled targets against all sources). */
void oracle_accel_range_f64(const odouble4 *X, odouble4 *A_out, int i0, int i1, int j0, int j1, double eps2)
{
#pragma omp parallel for schedule(static)
    for (int i = i0; i < i1; ++i) {
        const double xi = X[i].x, yi = X[i].y, zi = X[i].z;
        double ax = 0.0, ay = 0.0, az = 0.0;
        for (int j = j0; j < j1; ++j) {
            if (j == i) continue;
            double rx = X[j].x - xi, ry = X[j].y - yi, rz = X[j].z - zi;
            double d = rx * rx + ry * ry + rz * rz + eps2;
            double s = X[j].w * (1.0 / sqrt(d * d * d));
            ax += rx * s; ay += ry * s; az += rz * s;
        }
        A_out[i - i0].x = ax; A_out[i - i0].y = ay; A_out[i - i0].z = az; A_out[i - i0].w = 0.0;
    }
}

/* The OLDER snapshot's step (Sim-Without-OpenGL-Integration/kernel.cu:5-25,38-82 with its
 * constants.h:14-15, `EPS2 0.002` and `DT 0.01` as DOUBLE literals), in Jacobi order: the float sum
 * r.r is promoted to double for `+ EPS2`; `gV + 0.5 * DT * a` and `gX + DT * gV` are evaluated in
 * double and rounded to float once. The j == i pair is NOT skipped by that kernel (kernel.cu:30-34).
 * V3 is N packed float3. */
void oracle_step_legacy(ofloat4 *X, float *V3, int n)
{
    float *A = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        const ofloat4 bi = X[i];
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        for (int j = 0; j < n; ++j) {
            const float rx = X[j].x - bi.x, ry = X[j].y - bi.y, rz = X[j].z - bi.z;
            const float d = (float)((double)(rx * rx + ry * ry + rz * rz) + 0.002);
            const float denom = 1.0f / sqrtf(d * d * d);
            const float s = X[j].w * denom;
            ax += rx * s; ay += ry * s; az += rz * s;
        }
        A[3 * i] = ax; A[3 * i + 1] = ay; A[3 * i + 2] = az;
    }
    for (int i = 0; i < n; ++i) {
        V3[3 * i + 0] = (float)((double)V3[3 * i + 0] + 0.5 * 0.01 * (double)A[3 * i + 0]);
        V3[3 * i + 1] = (float)((double)V3[3 * i + 1] + 0.5 * 0.01 * (double)A[3 * i + 1]);
        V3[3 * i + 2] = (float)((double)V3[3 * i + 2] + 0.5 * 0.01 * (double)A[3 * i + 2]);
        X[i].x = (float)((double)X[i].x + 0.01 * (double)V3[3 * i + 0]);
        X[i].y = (float)((double)X[i].y + 0.01 * (double)V3[3 * i + 1]);
        X[i].z = (float)((double)X[i].z + 0.01 * (double)V3[3 * i + 2]);
    }
    free(A);
}

/* utils.cpp:6 */
static float oracle_random_float(float lo, float hi)
{
    return ((float)rand() / RAND_MAX) * (hi - lo) + lo;
}

/* utils.cpp:30-37 with constants.h:15-19 (MAX_X/Y/Z 1e5f, MIN_W 1e5f, MAX_W 1e9f). */
void oracle_fill_with_random4(ofloat4 *v, int n)
{
    for (int i = 0; i < n; ++i) {
        v[i].x = oracle_random_float(-100000.0f, 100000.0f);
        v[i].y = oracle_random_float(-100000.0f, 100000.0f);
        v[i].z = oracle_random_float(-100000.0f, 100000.0f);
        v[i].w = oracle_random_float(100000.0f, 1000000000.0f);
    }
}

/* utils.cpp:19-27 */
void oracle_fill_with_zeroes4(ofloat4 *v, int n)
{
    for (int i = 0; i < n; ++i) {
        v[i].x = 0.0f; v[i].y = 0.0f; v[i].z = 0.0f; v[i].w = 0.0f;
    }
}

/* validation.cpp:143-164: per component, |v-x| must be <= 1 % of the smaller
 * magnitude. Returns how many bodies the reference would have reported. */
int oracle_verify_still_bodies(const ofloat4 *v, const ofloat4 *x, int n)
{
    const float tol = 1.0 / 100;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        float tx = fminf(fabsf(v[i].x * tol), fabsf(x[i].x * tol));
        float ty = fminf(fabsf(v[i].y * tol), fabsf(x[i].y * tol));
        float tz = fminf(fabsf(v[i].z * tol), fabsf(x[i].z * tol));
        float dx = fabsf(v[i].x - x[i].x);
        float dy = fabsf(v[i].y - x[i].y);
        float dz = fabsf(v[i].z - x[i].z);
        if (dx > tx || dy > ty || dz > tz) ++bad;
    }
    return bad;
}

/* validation.cpp:106-122: absolute 0.01 on x,y,z,w. */
int oracle_verify_equality4(const ofloat4 *v, const ofloat4 *x, int n)
{
    const float tol = 0.01;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        if (fabsf(v[i].x - x[i].x) > tol || fabsf(v[i].y - x[i].y) > tol ||
            fabsf(v[i].z - x[i].z) > tol || fabsf(v[i].w - x[i].w) > tol)
            ++bad;
    }
    return bad;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

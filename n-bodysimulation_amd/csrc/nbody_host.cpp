// nbody_host.cpp — host-side helpers of include/nbody.h that carry the meaning of the
// reference's utils.h / validation.h entry points (no device code, no CPU force path).
#include "nbody.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>

namespace {

// utils.cpp:6
inline float random_float(float lo, float hi) { return ((float)rand() / RAND_MAX) * (hi - lo) + lo; }

// splitmix64: small, portable, seedable
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // uniform in [0,1) with 53 bits
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

}  // namespace

extern "C" {

// utils.cpp:30-37 with constants.h:15-19
void nbody_fill_with_random4(nbody_float4* v, int n)
{
    for (int i = 0; i < n; ++i) {
        v[i].x = random_float(-100000.0f, 100000.0f);
        v[i].y = random_float(-100000.0f, 100000.0f);
        v[i].z = random_float(-100000.0f, 100000.0f);
        v[i].w = random_float(100000.0f, 1000000000.0f);
    }
}

// utils.cpp:6
float nbody_random_float(float lo, float hi) { return random_float(lo, hi); }

// utils.cpp:9-16
void nbody_fill_with_zeroes3(nbody_float3* v, int n)
{
    for (int i = 0; i < n; ++i) {
        v[i].x = 0.0f;
        v[i].y = 0.0f;
        v[i].z = 0.0f;
    }
}

// utils.cpp:19-27
void nbody_fill_with_zeroes4(nbody_float4* v, int n)
{
    for (int i = 0; i < n; ++i) {
        v[i].x = 0.0f;
        v[i].y = 0.0f;
        v[i].z = 0.0f;
        v[i].w = 0.0f;
    }
}

int nbody_fill_seeded(nbody_float4* b, int n, int init, unsigned long long seed)
{
    if (!b || n < 0) return NBODY_ERR_INVALID;
    SplitMix64 g(seed);
    if (init == 0) {
        // the reference's distribution (utils.cpp:30-37), 4 draws per body in x,y,z,w order
        for (int i = 0; i < n; ++i) {
            b[i].x = (float)(g.uniform() * 2.0e5 - 1.0e5);
            b[i].y = (float)(g.uniform() * 2.0e5 - 1.0e5);
            b[i].z = (float)(g.uniform() * 2.0e5 - 1.0e5);
            b[i].w = (float)(1.0e5 + g.uniform() * (1.0e9 - 1.0e5));
        }
        return NBODY_OK;
    }
    if (init == 1) {
        // Plummer sphere, scale radius 1, total mass 1 (G = 1), positions only (cold start):
        // r = (u^(-2/3) - 1)^(-1/2), isotropic direction; 4 draws per body.
        const float m = n > 0 ? (float)(1.0 / n) : 0.0f;
        const double two_pi = 6.283185307179586476925286766559;
        for (int i = 0; i < n; ++i) {
            double u = g.uniform();
            if (u < 1e-10) u = 1e-10;
            if (u > 0.999) u = 0.999;  // r <= 38.7 scale radii
            const double r = 1.0 / std::sqrt(std::pow(u, -2.0 / 3.0) - 1.0);
            const double cz = 2.0 * g.uniform() - 1.0;
            const double ph = two_pi * g.uniform();
            (void)g.uniform();  // keep 4 draws per body like the reference's generator
            const double sz = std::sqrt(1.0 - cz * cz);
            b[i].x = (float)(r * sz * std::cos(ph));
            b[i].y = (float)(r * sz * std::sin(ph));
            b[i].z = (float)(r * cz);
            b[i].w = m;
        }
        return NBODY_OK;
    }
    return NBODY_ERR_INVALID;
}

// validation.cpp:143-164 as a count
int nbody_verify_still_bodies(const nbody_float4* v, const nbody_float4* x, int n)
{
    const float tol = 1.0 / 100;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float tx = std::fmin(std::fabs(v[i].x * tol), std::fabs(x[i].x * tol));
        const float ty = std::fmin(std::fabs(v[i].y * tol), std::fabs(x[i].y * tol));
        const float tz = std::fmin(std::fabs(v[i].z * tol), std::fabs(x[i].z * tol));
        const float dx = std::fabs(v[i].x - x[i].x);
        const float dy = std::fabs(v[i].y - x[i].y);
        const float dz = std::fabs(v[i].z - x[i].z);
        if (dx > tx || dy > ty || dz > tz) ++bad;
    }
    return bad;
}

// validation.cpp:125-140 as a count
int nbody_verify_equality3(const nbody_float3* v, const nbody_float3* x, int n)
{
    const float tol = 0.01;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        if (std::fabs(v[i].x - x[i].x) > tol || std::fabs(v[i].y - x[i].y) > tol || std::fabs(v[i].z - x[i].z) > tol) ++bad;
    }
    return bad;
}

// validation.cpp:106-122 as a count
int nbody_verify_equality4(const nbody_float4* v, const nbody_float4* x, int n)
{
    const float tol = 0.01;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        if (std::fabs(v[i].x - x[i].x) > tol || std::fabs(v[i].y - x[i].y) > tol ||
            std::fabs(v[i].z - x[i].z) > tol || std::fabs(v[i].w - x[i].w) > tol)
            ++bad;
    }
    return bad;
}

}  // extern "C"

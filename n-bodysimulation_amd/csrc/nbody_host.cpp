// nbody_host.cpp — host-side helpers of include/nbody.h that carry the meaning of the
// reference's utils.h / validation.h entry points (no device code, no CPU force path).
#include "nbody.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>

namespace {

// utils.cpp:6
inline float random_float(float lo, float hi) { return ((float)rand() / (float)RAND_MAX) * (hi - lo) + lo; }  // RAND_MAX converts to float exactly as the reference's implicit conversion does

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

// ---- sharded step: who evaluates which pairs, and who owes whom which sums (pure host logic) -------------

}  // extern "C"

namespace {

// the cross launches of rank q (symmetric schedule): every unordered pair of blocks {a, b}, a != b, is taken by
// exactly one rank — rank a for b = a+1 .. a+(G-1)/2 (mod G); for an even G the pair {q, q+G/2} is shared:
// the lower rank takes the lower half of the upper rank's block, the upper rank takes the rest with its own
// upper half as targets.
int cross_launches(int q, int G, int S, nbody_cross_launch out[2])
{
    const int n_pad = G * S;
    int n = 0, off = 0;
    auto add = [&](int i0, int i1, int j0, int count) {
        if (count <= 0 || i1 <= i0) return;
        out[n].i0 = i0; out[n].i1 = i1; out[n].j0 = j0 % n_pad; out[n].count = count; out[n].jbuf_offset = off;
        off += count;
        ++n;
    };
    const int own0 = q * S, own1 = own0 + S;
    if (G % 2 == 1) {
        add(own0, own1, own1, ((G - 1) / 2) * S);
    } else {
        const int full = G / 2 - 1;
        if (q < G / 2) {
            add(own0, own1, own1, full * S + S / 2);
        } else {
            add(own0, own1, own1, full * S);
            add(own0 + S / 2, own1, (q - G / 2) * S, S);
        }
    }
    return n;
}

int send_segments(int q, int G, int S, nbody_shard_segment* out)
{
    nbody_cross_launch L[2];
    const int nl = cross_launches(q, G, S, L);
    const int n_pad = G * S;
    int n = 0;
    for (int l = 0; l < nl; ++l) {
        int pos = 0;
        while (pos < L[l].count) {
            const int abs0 = (L[l].j0 + pos) % n_pad;
            const int owner = abs0 / S;
            int len = (owner + 1) * S - abs0;
            if (len > L[l].count - pos) len = L[l].count - pos;
            out[n].peer = owner;
            out[n].offset = L[l].jbuf_offset + pos;
            out[n].count = len;
            out[n].body0 = abs0;
            ++n;
            pos += len;
        }
    }
    return n;
}

}  // namespace

extern "C" {

int nbody_shard_plan(int rank, int world, int n_total, int schedule, nbody_shard_plan_t* p)
{
    if (!p || world < 1 || world > NBODY_MAX_RANKS || rank < 0 || rank >= world || n_total < 0) return NBODY_ERR_INVALID;
    if (schedule != NBODY_SCHEDULE_CANONICAL && schedule != NBODY_SCHEDULE_ONESIDED && schedule != NBODY_SCHEDULE_SYMMETRIC)
        return NBODY_ERR_INVALID;
    *p = nbody_shard_plan_t{};
    p->rank = rank; p->world = world; p->n_total = n_total; p->schedule = schedule;
    int S = (n_total + world - 1) / world;
    if (S & 1) ++S;  // even, so that the block half-way round an even ring can be split between its two ranks
    if ((long)S * world > 2147483647L) return NBODY_ERR_INVALID;
    p->shard = S;
    p->n_pad = S * world;
    p->i0 = rank * S;
    p->i1 = p->i0 + S;
    if (schedule != NBODY_SCHEDULE_SYMMETRIC || world == 1 || S == 0) return NBODY_OK;
    p->n_launches = cross_launches(rank, world, S, p->launch);
    for (int l = 0; l < p->n_launches; ++l) p->jbuf_bodies += p->launch[l].count;
    p->n_sends = send_segments(rank, world, S, p->send);
    // what arrives: every other rank's segments addressed to this one, nearest preceding rank first (the order in
    // which the sums are added — fixed, so the result does not depend on arrival times)
    nbody_shard_segment tmp[NBODY_MAX_RANKS];
    for (int d = 1; d < world; ++d) {
        const int q = (rank - d + world) % world;
        const int ns = send_segments(q, world, S, tmp);
        for (int k = 0; k < ns; ++k) {
            if (tmp[k].peer != rank) continue;
            nbody_shard_segment& r = p->recv[p->n_recvs++];
            r.peer = q;
            r.offset = p->rbuf_bodies;
            r.count = tmp[k].count;
            r.body0 = tmp[k].body0;
            p->rbuf_bodies += tmp[k].count;
        }
    }
    return NBODY_OK;
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

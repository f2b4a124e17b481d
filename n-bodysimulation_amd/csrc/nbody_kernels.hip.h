// nbody_kernels.hip.h — gfx950 (MI355X / CDNA4) device code for the all-pairs step.
//
// Replaces the reference's device side: bodyInteractions (TestProject/kernel.cu:9-29),
// tile_interaction (kernel.cu:55-65) and kernel (kernel.cu:80-130). It is a new design, not a
// translation:
//   * force accumulation and integration are SEPARATE kernels; positions are only written
//     by the integrate kernel after every force has been taken, so the step is Jacobi (the
//     reference kernel's in-place update races between blocks: kernel.cu:94-129);
//   * 256-thread workgroups (4 wave64), BPL target bodies per lane held in registers, the
//     source bodies streamed through a double-buffered LDS tile that is filled with
//     coalesced 16-byte loads and read back with broadcast ds_read_b128 (one LDS read
//     feeds BPL pairs per lane);
//   * the source range can be split over gridDim.y "slabs" so that small N still fills
//     256 CUs; slabs are summed in a fixed order by the integrate kernel (deterministic,
//     no float atomics).
//
// Two arithmetic flavours:
//   fast   — packed across two targets: per two pairs 3 v_pk_add (r), 3 v_pk_fma (d = r.r + eps2),
//            2 v_rsq_f32, 3 v_pk_mul (f = m * rsq^3), 3 v_pk_fma (a += r f);
//   strict — the reference's operation order with individually rounded IEEE ops
//            (1.0f / sqrtf(d*d*d), no contraction, j == i skipped as validation.cpp:35 does),
//            one target per lane, sources in index order: bit-identical to the CPU
//            restatement of validation.cpp:28-52 in Jacobi order.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "nbody_layout.hip.h"   // kWG, MassInfo, RunLayout, BalLayout, FusedSync: shared with the host-side shape logic

namespace nbk {


struct ForceParams {
    const float4* x;    // bodies {x,y,z,mass}; indexed absolutely over the whole system
    float4* out;        // slab 0 of the output: out[s * slab_stride + (i - i0)]
    int i0, i1;         // target bodies [i0, i1)
    int j0, j1;         // source bodies [j0, j1)
    int slab_stride;    // elements between slabs
    int accumulate;     // start each sum from the value already in `out`
    float eps2;
    int wrap;           // 0, or the system size: source index j >= wrap means body j - wrap (a source
                        // run that continues past the end of the array wraps around to body 0)
};

// ---------------------------------------------------------------------------------------
// fast flavour
// ---------------------------------------------------------------------------------------

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Scalar-lane maths: BPL independent targets per lane, one VALU op per target per step.
// (hipcc's SLP pass still pairs some of these into v_pk_* ops on its own.)
template <int BPL_>
struct MathScalar {
    static constexpr int BPL = BPL_;
    float x[BPL], y[BPL], z[BPL];
    float ax[BPL], ay[BPL], az[BPL];
    float e2;  // eps2 held in a VGPR: a VALU op with an SGPR source issues at half rate on gfx950

    __device__ __forceinline__ void set_eps2(const float eps2)
    {
        e2 = eps2;
        asm volatile("" : "+v"(e2));
    }
    __device__ __forceinline__ void set(int k, const float4 b)
    {
        x[k] = b.x; y[k] = b.y; z[k] = b.z;
        ax[k] = 0.0f; ay[k] = 0.0f; az[k] = 0.0f;
    }
    __device__ __forceinline__ float4 acc(int k) const { return make_float4(ax[k], ay[k], az[k], 0.0f); }

    __device__ __forceinline__ void pair4(const float sx, const float sy, const float sz, const float sm)
    {
        pair(make_float4(sx, sy, sz, sm));
    }

    __device__ __forceinline__ void pair(const float4 bj)
    {
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const float rx = bj.x - x[k];
            const float ry = bj.y - y[k];
            const float rz = bj.z - z[k];
            float d = __builtin_fmaf(rx, rx, e2);
            d = __builtin_fmaf(ry, ry, d);
            d = __builtin_fmaf(rz, rz, d);
            const float inv = __builtin_amdgcn_rsqf(d);  // v_rsq_f32, 1 ulp
            const float inv2 = inv * inv;
            const float mi = bj.w * inv;
            const float f = inv2 * mi;
            ax[k] = __builtin_fmaf(rx, f, ax[k]);
            ay[k] = __builtin_fmaf(ry, f, ay[k]);
            az[k] = __builtin_fmaf(rz, f, az[k]);
        }
    }
};

// Packed maths: targets 2k and 2k+1 share a 64-bit register pair per component, so the
// subtract / fma / mul steps are v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32 (two pairs per
// instruction, the source broadcast through op_sel); only v_rsq_f32 stays per pair.
// 12 packed ops + 2 rsq for two pairs instead of 24 + 2.
template <int BPL_>
struct MathPacked {
    static_assert(BPL_ % 2 == 0, "packed maths handles targets two at a time");
    static constexpr int BPL = BPL_;
    static constexpr int H = BPL / 2;
    f32x2 x[H], y[H], z[H];
    f32x2 ax[H], ay[H], az[H];
    f32x2 e2;  // {eps2, eps2} in a VGPR pair

    __device__ __forceinline__ void set_eps2(const float eps2)
    {
        e2 = (f32x2){eps2, eps2};
        asm volatile("" : "+v"(e2));
    }
    __device__ __forceinline__ void set(int k, const float4 b)
    {
        x[k >> 1][k & 1] = b.x; y[k >> 1][k & 1] = b.y; z[k >> 1][k & 1] = b.z;
        ax[k >> 1][k & 1] = 0.0f; ay[k >> 1][k & 1] = 0.0f; az[k >> 1][k & 1] = 0.0f;
    }
    __device__ __forceinline__ float4 acc(int k) const
    {
        return make_float4(ax[k >> 1][k & 1], ay[k >> 1][k & 1], az[k >> 1][k & 1], 0.0f);
    }

    __device__ __forceinline__ void pair(const float4 bj) { pair4(bj.x, bj.y, bj.z, bj.w); }

    __device__ __forceinline__ void pair4(const float sx, const float sy, const float sz, const float sm)
    {
        const f32x2 bx = {sx, sx}, by = {sy, sy}, bz = {sz, sz}, bm = {sm, sm};
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const f32x2 rx = bx - x[k];
            const f32x2 ry = by - y[k];
            const f32x2 rz = bz - z[k];
            f32x2 d = __builtin_elementwise_fma(rx, rx, e2);
            d = __builtin_elementwise_fma(ry, ry, d);
            d = __builtin_elementwise_fma(rz, rz, d);
            f32x2 inv;
            inv.x = __builtin_amdgcn_rsqf(d.x);
            inv.y = __builtin_amdgcn_rsqf(d.y);
            const f32x2 inv2 = inv * inv;
            const f32x2 mi = bm * inv;
            const f32x2 f = inv2 * mi;
            ax[k] = __builtin_elementwise_fma(rx, f, ax[k]);
            ay[k] = __builtin_elementwise_fma(ry, f, ay[k]);
            az[k] = __builtin_elementwise_fma(rz, f, az[k]);
        }
    }
};

// Source range of slab s when [j0,j1) is cut into `nslab` runs of whole tiles whose lengths differ
// by at most one tile (slab s gets tiles [s*ntile/nslab, (s+1)*ntile/nslab)).
__device__ __forceinline__ void slab_range(int j0, int j1, int tile, int nslab, int s, int& a, int& b)
{
    const long ntile = (j1 - j0 + tile - 1) / tile;
    const long ta = (long)s * ntile / nslab;
    const long tb = (long)(s + 1) * ntile / nslab;
    const long la = (long)j0 + ta * tile, lb = (long)j0 + tb * tile;
    a = la > j1 ? j1 : (int)la;
    b = lb > j1 ? j1 : (int)lb;
}

template <class M, int WG = kWG>
__device__ __forceinline__ void load_targets(const ForceParams& p, int ibase, M& t)
{
#pragma unroll
    for (int k = 0; k < M::BPL; ++k) {
        int i = ibase + k * WG + (int)threadIdx.x;
        if (i > p.i1 - 1) i = p.i1 - 1;  // clamp: the surplus lanes compute a copy, never store
        t.set(k, p.x[i]);
    }
}

template <class M, int WG = kWG>
__device__ __forceinline__ void store_targets(const ForceParams& p, int ibase, int slab, const M& t)
{
    float4* out = p.out + (size_t)slab * p.slab_stride;
#pragma unroll
    for (int k = 0; k < M::BPL; ++k) {
        const int i = ibase + k * WG + (int)threadIdx.x;
        if (i < p.i1) {
            float4 a = t.acc(k);
            if (p.accumulate) {
                const float4 o = out[i - p.i0];
                a.x += o.x; a.y += o.y; a.z += o.z;
            }
            out[i - p.i0] = a;
        }
    }
}

// LDS-tiled force kernel. grid = (ceil((i1-i0) / (256*BPL)), nslab).
// LAYOUT of a source in the LDS tile: 0 = {x,y,z,m} (one ds_read_b128), 1 = {x,y,m,z},
// 2 = two 8-byte halves {x,y} | {z,m} in separate arrays. Same arithmetic; they differ only in
// how many v_mov / s_nop hipcc adds around the op_sel broadcasts (measured in tools/kbench.hip).
// WG = threads per workgroup (256 by default; 64 or 128 give small systems more, shorter workgroups).
template <class M, int TILE, int UNROLL, int MINW, int LAYOUT = 0, int WG = kWG>
__global__ void __launch_bounds__(WG, MINW) force_lds(const ForceParams p)
{
    static_assert(TILE % WG == 0, "tile must be a multiple of the workgroup");
    constexpr int LPT = TILE / WG;  // float4 loads per thread per tile
    __shared__ float4 sh[2][TILE];
    float2* const sh_xy = reinterpret_cast<float2*>(&sh[0][0]);             // LAYOUT 2: [2][TILE] halves
    float2* const sh_zm = reinterpret_cast<float2*>(&sh[0][0]) + 2 * TILE;

    const int tid = threadIdx.x;
    const int ibase = p.i0 + blockIdx.x * (WG * M::BPL);
    M t;
    t.set_eps2(p.eps2);
    load_targets<M, WG>(p, ibase, t);

    int ja, jb;
    slab_range(p.j0, p.j1, TILE, gridDim.y, blockIdx.y, ja, jb);

    // A source past the end is replaced by a massless body: it adds exactly +-0.
    float4 pre[LPT];
    auto fetch = [&](int jt) {
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            const int j = jt + l * WG + tid;
            const int js = (p.wrap && j >= p.wrap) ? j - p.wrap : j;
            pre[l] = (j < jb) ? p.x[js] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };

    if (ja < jb) fetch(ja);
    int buf = 0;
    for (int jt = ja; jt < jb; jt += TILE, buf ^= 1) {
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            const int e = l * WG + tid;
            if (LAYOUT == 0) sh[buf][e] = pre[l];
            if (LAYOUT == 1) sh[buf][e] = make_float4(pre[l].x, pre[l].y, pre[l].w, pre[l].z);
            if (LAYOUT == 2) {
                sh_xy[buf * TILE + e] = make_float2(pre[l].x, pre[l].y);
                sh_zm[buf * TILE + e] = make_float2(pre[l].z, pre[l].w);
            }
        }
        __syncthreads();  // one barrier per tile: the other buffer is only rewritten after
                          // every wave has passed the NEXT barrier, i.e. finished this tile
        if (jt + TILE < jb) fetch(jt + TILE);
#pragma unroll UNROLL
        for (int jj = 0; jj < TILE; ++jj) {
            if (LAYOUT == 0) t.pair(sh[buf][jj]);
            if (LAYOUT == 1) {
                const float4 s = sh[buf][jj];
                t.pair4(s.x, s.y, s.w, s.z);
            }
            if (LAYOUT == 2) {
                f32x2 a = *reinterpret_cast<const f32x2*>(&sh_xy[buf * TILE + jj]);
                f32x2 b = *reinterpret_cast<const f32x2*>(&sh_zm[buf * TILE + jj]);
                asm volatile("" : "+v"(a), "+v"(b));  // keep each half in a 64-bit pair of its own
                t.pair4(a.x, a.y, b.x, b.y);
            }
        }
    }
    store_targets<M, WG>(p, ibase, blockIdx.y, t);
}

// ---------------------------------------------------------------------------------------
// symmetric flavour — every unordered pair of bodies evaluated ONCE (Newton's third law)
// ---------------------------------------------------------------------------------------
//
// The system is cut into blocks of B = 64*W*BPL bodies. One workgroup (W wave64) takes one block
// pair (I < J): wave w keeps 64*BPL bodies of block I in registers (BPL per lane) with their
// accumulators, and block J passes through the wave 64 bodies at a time, ONE BODY PER LANE, also in
// registers: {x,y,z,m} plus that body's own accumulator. Lane l meets the J-body of lane l+s of its
// 16-lane row through a DPP row rotation (row_ror:s, s = 0..15, folded into the consuming
// instruction or one v_mov_dpp per component); after 16 rotations the seven registers of the J-body
// move on by one row (ds_bpermute, 4 times per 64 bodies). Per evaluated pair: r = xj - xi, d = r.r +
// eps2, w = rsq(d)^3, a_i += (m_j w) r, t += (m_i w) r — 16 FMA-class ops + 1 v_rsq_f32 for TWO
// counted interactions (the one-sided kernel needs 12 + 1 for one) — and per J-body and step three
// v_add_dpp that carry t back to the body's home lane. No LDS and no barrier inside the pair loop.
// The J-side sums of the W waves meet in LDS: in round q wave w works on chunk (q + w) mod (B/64),
// so no two waves hold the same chunk in the same round; one barrier per round.
// Output: slab[J][i in I] (I-side sums, from registers) and slab[I][j in J] (J-side sums, from LDS);
// the diagonal tasks (I == I, one-sided arithmetic, same rotation scheme) write slab[I][i in I]. Every
// slab element is written exactly once per launch and the integrate kernel adds the nb slabs in
// index order: deterministic, no float atomics.

// EQUAL-MASS systems (a Plummer model, most cosmological and cluster initial conditions: m_i = M / N). When every body a launch
// touches has bit-for-bit the same mass m0, a_i = m0 * sum_j w_ij r_ij: the symmetric kernels then accumulate sum w r on both sides
// — no m_j w, no m_i w, no mass to rotate: 89 instead of 100 VALU instructions per rotation step — and apply m0 once per stored sum.
// mass_scan decides on the device, in stream order (no host round trip): it compares every body of the launch's ranges with the
// first one and stamps `bad_gen` with the launch's generation number when one differs (or a coordinate is beyond kEqMaxCoord or
// not finite, see the padding below); the force kernel takes the equal-mass path when the stamp is not its own generation.
// Padding bodies (past the end of a range) have no mass to switch them off on that path: they sit at (1e18, 1e18, 1e18) (fp64: 1e150),
// where w = rsq(d)^3 underflows to exactly 0 against every body within kEqMaxCoord of the origin.
constexpr float kEqMaxCoord = 1e15f;
// where a padding body sits on the equal-mass path: w = rsq(d)^3 underflows to exactly 0 from there (fp32: d = 3e36, w = 2e-55;
// fp64: d = 3e300, w = 2e-451) while r itself stays finite, so w * r is an exact zero
template <class S> __device__ __forceinline__ S eq_far();
template <> __device__ __forceinline__ float eq_far<float>() { return 1e18f; }
template <> __device__ __forceinline__ double eq_far<double>() { return 1e150; }

__device__ __forceinline__ bool same_bits(const float a, const float b) { return __builtin_bit_cast(unsigned int, a) == __builtin_bit_cast(unsigned int, b); }
__device__ __forceinline__ bool same_bits(const double a, const double b) { return __builtin_bit_cast(unsigned long long, a) == __builtin_bit_cast(unsigned long long, b); }

template <class V4>
struct MassScanParamsT {
    const V4* x;
    int i0, ni;      // first range
    int j0, nj;      // second range (nj = 0: none); indices at or beyond `wrap` continue at body 0 when wrap > 0
    int wrap;
    MassInfo* out;
    unsigned int gen;
};
using MassScanParams = MassScanParamsT<float4>;

template <class V4>
__global__ void __launch_bounds__(256) mass_scan(const MassScanParamsT<V4> p)
{
    using S = decltype(p.x->w);
    const S ref = p.x[p.i0].w;
    bool bad = false;
    const int total = p.ni + p.nj;
#pragma unroll 1
    for (int e = (int)blockIdx.x * 256 + (int)threadIdx.x; e < total; e += (int)gridDim.x * 256) {
        int idx;
        if (e < p.ni) idx = p.i0 + e;
        else {
            idx = p.j0 + (e - p.ni);
            if (p.wrap && idx >= p.wrap) idx -= p.wrap;
        }
        const V4 b = p.x[idx];
        bad |= !same_bits(b.w, ref);   // bit for bit (-0 and +0 differ, a NaN equals only itself)
        const S lim = (S)kEqMaxCoord;
        bad |= !((b.x <= lim && b.x >= -lim) && (b.y <= lim && b.y >= -lim) && (b.z <= lim && b.z >= -lim));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.out->m0 = (double)ref;
        bad |= !(ref <= (S)3.0e38 && ref >= (S)-3.0e38);   // a NaN or infinite common mass: the general path reproduces it term by term
    }
    if (bad) p.out->bad_gen = p.gen;   // every writer stores the same value
}

// the launch's verdict: wave-uniform (scalar loads from a kernel-argument pointer)
__device__ __forceinline__ bool eq_uniform(const MassInfo* q, const unsigned int gen, double* m0)
{
    if (!q) return false;
    const unsigned int bad = __builtin_amdgcn_readfirstlane((int)q->bad_gen);
    const long long b = __builtin_bit_cast(long long, q->m0);
    const int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    *m0 = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    return bad != gen;
}

template <class V4, class S>
struct SymParamsT {
    const V4* x;          // bodies {x,y,z,mass}, indexed absolutely
    V4* slabs_i;          // I-side partial sums: slab s at slabs_i + s*stride_i, element = index within the I range
    V4* slabs_j;          // J-side partial sums: slab s at slabs_j + s*stride_j, element = index within the J run
    int ni, nj;           // bodies of the I range / of the J run
    int i0, j0;           // absolute index of the first I / first J body
    int wrap;             // 0, or the array length: a J index at or beyond it continues at body 0
    int nbi, nbj;         // blocks of B bodies on each side
    int stride_i, stride_j;
    int task0;            // first task of this launch (a grid may cover a sub-range of the task list: task = task0 + blockIdx.x)
    int rect;             // 0: ONE range (I range == J run): block pairs I < J once + the diagonal blocks one-sided
                          //    (nbi == nbj == nb, slabs_i == slabs_j: slab J gets the I-side sums, slab I the J-side sums)
                          // 1: TWO disjoint ranges: every (I, J) block pair, symmetric; slab J of slabs_i, slab I of slabs_j
    S eps2;
    const MassInfo* eqm;  // mass_scan's verdict on this launch's ranges (nullptr: general path)
    unsigned int eq_gen;  // this launch's generation
    // force_sym_ticket only (sums added in place, no slabs): acc_lanes accumulation arrays of the range's bodies, acc_stride elements
    // apart (ONE lane = the acceleration array itself); one ticket word per (block, lane), zero between launches, the abort word
    // behind them; the host-mapped error word
    V4* acc;
    unsigned* tickets;
    unsigned* err;
    int acc_lanes, acc_stride;
    int acc_timeout_us;   // how long a wait may last before the launch is aborted and flagged (10 s; the test hook sets milliseconds)
};
using SymParams = SymParamsT<float4, float>;
using SymParamsF64 = SymParamsT<double4, double>;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(const float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// value of the lane S places away in the 16-lane row (row_ror); ror<S> and ror<(16-S)%16> are inverse
template <int S>
__device__ __forceinline__ float ror(const float v)
{
    if constexpr (S == 0) return v;
    else return dpp_mov<0x120 + S>(v);
}

template <int S>
__device__ __forceinline__ double ror(const double v)
{
    if constexpr (S == 0) return v;
    else {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x120 + S, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x120 + S, 0xf, 0xf, true);
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
}

template <class V4> __device__ __forceinline__ V4 zero4();
template <> __device__ __forceinline__ float4 zero4<float4>() { return make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
template <> __device__ __forceinline__ double4 zero4<double4>() { return make_double4(0.0, 0.0, 0.0, 0.0); }
// a body past the end of its range: massless at the origin (adds exactly +-0 to every real body); on the equal-mass path, where
// no mass switches it off, far away instead (w underflows to exactly 0)
template <class V4, bool EQ> __device__ __forceinline__ V4 pad4()
{
    V4 v = zero4<V4>();
    if (EQ) { v.x = eq_far<decltype(v.x)>(); v.y = v.x; v.z = v.x; }
    return v;
}

// Packed arithmetic for the rotation kernel: BPL stationary bodies per lane, two per register pair.
template <int BPL_>
struct SymPacked {
    static_assert(BPL_ % 2 == 0, "packed maths handles bodies two at a time");
    using S = float;
    using V4 = float4;
    static constexpr int BPL = BPL_;
    static constexpr int H = BPL / 2;
    f32x2 x[H], y[H], z[H], m[H];
    f32x2 ax[H], ay[H], az[H];
    f32x2 e2;

    __device__ __forceinline__ void set_eps2(const float eps2)
    {
        e2 = (f32x2){eps2, eps2};
        asm volatile("" : "+v"(e2));
    }
    __device__ __forceinline__ void set(int k, const float4 b)
    {
        x[k >> 1][k & 1] = b.x; y[k >> 1][k & 1] = b.y; z[k >> 1][k & 1] = b.z; m[k >> 1][k & 1] = b.w;
        ax[k >> 1][k & 1] = 0.0f; ay[k >> 1][k & 1] = 0.0f; az[k >> 1][k & 1] = 0.0f;
    }
    __device__ __forceinline__ float4 acc(int k) const
    {
        return make_float4(ax[k >> 1][k & 1], ay[k >> 1][k & 1], az[k >> 1][k & 1], 0.0f);
    }
    // the sums so far times s (the equal-mass path accumulates sum w r and applies the common mass once, at the end)
    __device__ __forceinline__ void scale(const float s)
    {
        const f32x2 v = {s, s};
#pragma unroll
        for (int k = 0; k < H; ++k) { ax[k] = ax[k] * v; ay[k] = ay[k] * v; az[k] = az[k] * v; }
    }
    __device__ __forceinline__ float mass0() const { return m[0][0]; }
    // do all of this lane's stationary bodies carry exactly the mass m0?
    __device__ __forceinline__ bool masses_are(const float m0) const
    {
        const unsigned int b = __builtin_bit_cast(unsigned int, m0);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < H; ++k)
            ok = ok && __builtin_bit_cast(unsigned int, (float)m[k][0]) == b && __builtin_bit_cast(unsigned int, (float)m[k][1]) == b;
        return ok;
    }
    // all BPL stationary bodies against the moving body s; t = sum_k (m_k w_k) r_k when SYM.
    // EQ: every body of the launch has the same mass (MassInfo below): a_i += w r, t += w r — the two mass multiplies per pair
    // (and the moving body's mass itself) drop out, the common mass is applied once per stored sum. 14 instead of 16 packed ops
    // per two pair evaluations.
    template <bool SYM, bool EQ = false>
    __device__ __forceinline__ void pairs(const float sx, const float sy, const float sz, const float sm, float& tx,
                                          float& ty, float& tz)
    {
        const f32x2 bx = {sx, sx}, by = {sy, sy}, bz = {sz, sz}, bm = {sm, sm};
        f32x2 ux = {0.0f, 0.0f}, uy = {0.0f, 0.0f}, uz = {0.0f, 0.0f};
        // written stage by stage across the H register pairs: dependent packed ops of one pair are
        // then H instructions apart (back-to-back dependent v_pk_* cost an s_nop each on gfx950)
        f32x2 rx[H], ry[H], rz[H], d[H], w[H], fi[H];
#pragma unroll
        for (int k = 0; k < H; ++k) { rx[k] = bx - x[k]; ry[k] = by - y[k]; rz[k] = bz - z[k]; }
#pragma unroll
        for (int k = 0; k < H; ++k) d[k] = __builtin_elementwise_fma(rx[k], rx[k], e2);
#pragma unroll
        for (int k = 0; k < H; ++k) d[k] = __builtin_elementwise_fma(ry[k], ry[k], d[k]);
#pragma unroll
        for (int k = 0; k < H; ++k) d[k] = __builtin_elementwise_fma(rz[k], rz[k], d[k]);
#pragma unroll
        for (int k = 0; k < H; ++k) { w[k].x = __builtin_amdgcn_rsqf(d[k].x); w[k].y = __builtin_amdgcn_rsqf(d[k].y); }
#pragma unroll
        for (int k = 0; k < H; ++k) d[k] = w[k] * w[k];
#pragma unroll
        for (int k = 0; k < H; ++k) w[k] = d[k] * w[k];
#pragma unroll
        for (int k = 0; k < H; ++k) fi[k] = EQ ? w[k] : bm * w[k];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            ax[k] = __builtin_elementwise_fma(rx[k], fi[k], ax[k]);
            ay[k] = __builtin_elementwise_fma(ry[k], fi[k], ay[k]);
            az[k] = __builtin_elementwise_fma(rz[k], fi[k], az[k]);
        }
        if (SYM) {
            if (!EQ) {
#pragma unroll
                for (int k = 0; k < H; ++k) fi[k] = m[k] * w[k];
            }
            ux = rx[0] * fi[0]; uy = ry[0] * fi[0]; uz = rz[0] * fi[0];
#pragma unroll
            for (int k = 1; k < H; ++k) {
                ux = __builtin_elementwise_fma(rx[k], fi[k], ux);
                uy = __builtin_elementwise_fma(ry[k], fi[k], uy);
                uz = __builtin_elementwise_fma(rz[k], fi[k], uz);
            }
        }
        tx = ux.x + ux.y; ty = uy.x + uy.y; tz = uz.x + uz.y;
    }
};

// Double-precision arithmetic for the rotation kernel (the build's own fp64 variant, BASELINE configs[4]).
// d^(-3/2): v_rsq_f64 seed (about 5e-8 relative, tools/rsq64_probe.hip) and ONE third-order step
// y <- y (1 + r/2 + 3 r^2/8), r = 1 - d y^2 (error 5/16 r^3 ~ 1e-22), then the cube: 7 ops, against 9 with two Newton
// steps. (Correcting the cube directly — w = q^3 (1 + 3r/2 + 15r^2/8), 6 ops — measured 2.8 % SLOWER: 27.9 vs 27.2 ms
// per step at N=262144, same box, alternating runs; two more values stay live per pair.)
template <int BPL_>
struct SymF64 {
    using S = double;
    using V4 = double4;
    static constexpr int BPL = BPL_;
    double x[BPL], y[BPL], z[BPL], m[BPL];
    double ax[BPL], ay[BPL], az[BPL];
    double e2;

    __device__ __forceinline__ void set_eps2(const double eps2) { e2 = eps2; }
    __device__ __forceinline__ void set(int k, const double4 b)
    {
        x[k] = b.x; y[k] = b.y; z[k] = b.z; m[k] = b.w;
        ax[k] = 0.0; ay[k] = 0.0; az[k] = 0.0;
    }
    __device__ __forceinline__ double4 acc(int k) const { return make_double4(ax[k], ay[k], az[k], 0.0); }
    __device__ __forceinline__ void scale(const double s)
    {
#pragma unroll
        for (int k = 0; k < BPL; ++k) { ax[k] *= s; ay[k] *= s; az[k] *= s; }
    }
    template <bool SYM, bool EQ = false>
    __device__ __forceinline__ void pairs(const double sx, const double sy, const double sz, const double sm, double& tx,
                                          double& ty, double& tz)
    {
        tx = 0.0; ty = 0.0; tz = 0.0;
        double rx[BPL], ry[BPL], rz[BPL], w[BPL];
#pragma unroll
        for (int k = 0; k < BPL; ++k) { rx[k] = sx - x[k]; ry[k] = sy - y[k]; rz[k] = sz - z[k]; }
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const double d = __builtin_fma(rz[k], rz[k], __builtin_fma(ry[k], ry[k], __builtin_fma(rx[k], rx[k], e2)));
            double q = __builtin_amdgcn_rsq(d);
            const double r = __builtin_fma(-d, q * q, 1.0);
            q = __builtin_fma(q * r, __builtin_fma(0.375, r, 0.5), q);
            w[k] = q * q * q;
        }
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const double fi = EQ ? w[k] : sm * w[k];
            ax[k] = __builtin_fma(rx[k], fi, ax[k]);
            ay[k] = __builtin_fma(ry[k], fi, ay[k]);
            az[k] = __builtin_fma(rz[k], fi, az[k]);
            if (SYM) {
                const double fj = EQ ? w[k] : m[k] * w[k];
                tx = __builtin_fma(rx[k], fj, tx);
                ty = __builtin_fma(ry[k], fj, ty);
                tz = __builtin_fma(rz[k], fj, tz);
            }
        }
    }
};

template <int S, bool SYM, bool EQ = false, class M>
__device__ __forceinline__ void sym_step(M& t, const typename M::V4& bj, typename M::V4& aj)
{
    typename M::S tx, ty, tz;
    if constexpr (EQ) t.template pairs<SYM, true>(ror<S>(bj.x), ror<S>(bj.y), ror<S>(bj.z), typename M::S(0), tx, ty, tz);   // no mass to rotate
    else t.template pairs<SYM, false>(ror<S>(bj.x), ror<S>(bj.y), ror<S>(bj.z), ror<S>(bj.w), tx, ty, tz);
    if (SYM) {  // a_j = -sum_i (m_i w) r, delivered to the moving body's lane
        aj.x -= ror<(16 - S) % 16>(tx);
        aj.y -= ror<(16 - S) % 16>(ty);
        aj.z -= ror<(16 - S) % 16>(tz);
    }
}

template <bool SYM, bool EQ = false, class M>
__device__ __forceinline__ void sym_row_pass(M& t, const typename M::V4& bj, typename M::V4& aj)
{
    sym_step<0, SYM, EQ>(t, bj, aj);  sym_step<1, SYM, EQ>(t, bj, aj);  sym_step<2, SYM, EQ>(t, bj, aj);  sym_step<3, SYM, EQ>(t, bj, aj);
    sym_step<4, SYM, EQ>(t, bj, aj);  sym_step<5, SYM, EQ>(t, bj, aj);  sym_step<6, SYM, EQ>(t, bj, aj);  sym_step<7, SYM, EQ>(t, bj, aj);
    sym_step<8, SYM, EQ>(t, bj, aj);  sym_step<9, SYM, EQ>(t, bj, aj);  sym_step<10, SYM, EQ>(t, bj, aj); sym_step<11, SYM, EQ>(t, bj, aj);
    sym_step<12, SYM, EQ>(t, bj, aj); sym_step<13, SYM, EQ>(t, bj, aj); sym_step<14, SYM, EQ>(t, bj, aj); sym_step<15, SYM, EQ>(t, bj, aj);
}

__device__ __forceinline__ float next_row(const float v, const int addr)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double next_row(const double v, const int addr)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)b);
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// first index of row I in the row-major list of block pairs (I < J): I*(2nb - I - 1)/2
__device__ __forceinline__ int sym_row_offset(int I, int nb) { return (int)(((long)I * (2L * nb - I - 1)) / 2); }

// ONE rotation body for the three launch geometries (block = 64*W threads; EQ: the equal-mass path, m0 = the common mass):
//   kSymGeneral  what the parameters say at run time: rect == 0 -> grid = nb*(nb-1)/2 pair tasks followed by nb diagonal tasks;
//                rect == 1 -> grid = nbi*nbj pair tasks; the J run may start anywhere (j0) and wrap around (wrap);
//   kSymSquare   the SQUARE case alone (one range against itself: rect == 0, no wrap, i0 == j0): same tasks, arithmetic and slab
//                layout, without the run-time rectangle / wrap-around handling;
//   kSymRect     the RECTANGULAR case alone (two disjoint ranges, rect == 1; the J run may wrap): what nbody_accel_cross launches —
//                seven eighths of a rank's pairs in an 8-GPU run — without the triangular task list and the diagonal tasks.
// The geometry is a COMPILE-TIME policy (`if constexpr`): each instantiation contains only its own control flow, and that is what the
// specialised kernels are for — the instruction schedule of the rotation pass that hipcc finds for the simpler control flow, not a
// different algorithm: square 2.7 % faster than general at N = 262144 (tools/symbench.hip, profiles/r03_symbench_rows.txt: 10.98 vs
// 11.29 ms per launch on one box), rect 1-3 % on a two-halves launch (profiles/r03_symbench_rect.txt). Rounds 3-4 kept three copies of
// this function; the fold keeps every kernel's instruction counts, registers and occupancy and measures within 0.5 % of the copies on
// the same box (profiles/r05_one_rotation_body.txt). hipcc's register assignment in the rotation pass is sensitive to how this function
// is WRITTEN, not only to what it computes: a first version (stores through lambdas, the equal-mass dispatch in a helper) had the same VALU
// count and ran 2 % slower on the general path. Re-measure (`tools/gpu_round.sh <tag> symab`, table by
// tools/symbench_ab_table.py) after touching it.
//   kSymTicket   the SQUARE case with the partial sums added IN PLACE (no slab workspace): see force_sym_ticket below.
enum SymCase : int { kSymGeneral = 0, kSymSquare = 1, kSymRect = 2, kSymTicket = 3 };

// ---- sums in place: the TICKET protocol of force_sym_ticket ----------------------------------------------------------------------
// The slab layout costs nb x 16 N bytes (one partial sum per body per block: 412 MiB at N = 262144, 6.4 GiB at 1 M, beyond any cap
// near 4 M). Here every task adds its two block sums straight into the acceleration array instead, and the ORDER in which the nb
// contributions of a block are added is fixed by a ticket per block, so the result is the same bits on every run:
//   * tasks are listed ANTI-DIAGONAL by anti-diagonal — d = 1 .. nb-1, within it (I, I + d) by I — and then the nb diagonal blocks; a
//     block meets at most two tasks per anti-diagonal, so the ~500 tasks in flight at any time touch a block about ten times, spread
//     over the whole round (the row-major list of the slab kernels would put a row's hundred tasks — all contributors to ONE block —
//     side by side);
//   * contribution number s of block K (counted along that list; closed forms below) goes to accumulation LANE s mod L (L = 1, 2, 4 or 8
//     arrays of N sums; the integrate adds the lanes in index order): the ~ten contributions a block receives per round of resident
//     tasks — which all finish within microseconds of each other — then queue up two deep instead of ten deep. It waits until
//     tickets[K][lane] == s / L, adds (a lane's first contribution stores), drains its stores (s_waitcnt vmcnt(0) in every wave, then
//     the workgroup barrier), and hands the ticket on; a lane's last contribution leaves its ticket at zero for the next launch. With
//     L = 1 the one lane is the acceleration array itself: no workspace at all;
//   * the sums move through AGENT-SCOPE accesses (buffer_load_dwordx3 / buffer_store_dwordx4 with the sc1 policy bit; the tickets
//     through agent-scope 32-bit atomics): the eight XCDs' L2s are not coherent for ordinary accesses, and an agent-scope load is
//     defined to see an agent-scope store that completed before it was issued.
// A waiter only ever waits for a task EARLIER in the list. Workgroups of a grid start in index order, so the earliest unfinished task
// is always resident and never waits: no deadlock; and should a wait exceed ten seconds all the same (a stopped predecessor), the
// waiter raises the host-mapped error word and goes on — a wrong, flagged result instead of a hung GPU (the host checks the word at
// its next synchronisation and returns an error).
constexpr int kTicketTimeoutUs = 10000000;   // 10 s: what a wait may last on a product launch

__device__ __forceinline__ float4 load_f4_agent(const float4* src)
{
    unsigned long long* const q = reinterpret_cast<unsigned long long*>(const_cast<float4*>(src));
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__builtin_bit_cast(float, (unsigned)lo), __builtin_bit_cast(float, (unsigned)(lo >> 32)),
                       __builtin_bit_cast(float, (unsigned)hi), __builtin_bit_cast(float, (unsigned)(hi >> 32)));
}

__device__ __forceinline__ void store_f4_agent(float4* dst, const float4 v)
{
    unsigned long long* const q = reinterpret_cast<unsigned long long*>(dst);
    const unsigned long long lo = (unsigned long long)__builtin_bit_cast(unsigned, v.x) | ((unsigned long long)__builtin_bit_cast(unsigned, v.y) << 32);
    const unsigned long long hi = (unsigned long long)__builtin_bit_cast(unsigned, v.z) | ((unsigned long long)__builtin_bit_cast(unsigned, v.w) << 32);
    __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same at 128 bits: one buffer_load/store_dwordx4 with the sc1 policy bit per float4 (HIP's scoped atomics stop at 64 bits; the raw
// buffer intrinsics take the cache policy as an operand and leave the wait counters to the compiler). Coherence is what is needed here,
// not single-copy atomicity of the 16 bytes: the ticket orders the accesses. `lane` = a buffer resource over one accumulation lane.
typedef unsigned int nbk_u4 __attribute__((ext_vector_type(4)));
typedef float nbk_f4 __attribute__((ext_vector_type(4)));
constexpr int kBufferPolicySc1 = 16;   // aux / cache-policy operand of the gfx940+ buffer intrinsics: bit 0 sc0, bit 1 nt, bit 4 sc1

__device__ __forceinline__ __amdgpu_buffer_rsrc_t acc_lane_rsrc(float4* const base, const int n)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, n * (int)sizeof(float4), 0x00020000);
}

// (The 16 bytes are re-typed as a WHOLE vector. `__builtin_bit_cast(float, q.y)` on an element of an ext_vector_type value is
//  mis-lowered by this clang — every element reads element 0 — which the ISA test caught as 22 one-dword loads where 22 wide ones belong.)
__device__ __forceinline__ float4 load_f4_agent(const __amdgpu_buffer_rsrc_t lane, const int i)
{
    const nbk_f4 q = __builtin_bit_cast(nbk_f4, __builtin_amdgcn_raw_buffer_load_b128(lane, i * (int)sizeof(float4), 0, kBufferPolicySc1));
    return make_float4(q.x, q.y, q.z, q.w);
}

__device__ __forceinline__ void store_f4_agent(const __amdgpu_buffer_rsrc_t lane, const int i, const float4 v)
{
    nbk_f4 q;
    q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(nbk_u4, q), lane, i * (int)sizeof(float4), 0, kBufferPolicySc1);
}

// ONE thread: returns once *tk == want. `abort` is the word behind the last ticket: set by the first waiter that gives up, it lets every
// later wait of the launch fall through at once (the launch ends within seconds instead of one time-out per waiter).
__device__ __forceinline__ void ticket_spin(unsigned* const tk, const unsigned want, unsigned* const abort, unsigned* const err, const int timeout_us)
{
    if (__hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        __builtin_amdgcn_s_sleep(2);
        if (__hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return;
        if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100ull * (unsigned long long)timeout_us) {   // (100-MHz ticks) never on a healthy run: flag it and go on
            __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}

// all threads of the workgroup: returns once the turn has come (thread 0 looks, the barrier tells the others)
__device__ __forceinline__ void ticket_wait(unsigned* const tk, const unsigned want, unsigned* const abort, unsigned* const err, const int timeout_us)
{
    if (threadIdx.x == 0) ticket_spin(tk, want, abort, err, timeout_us);
    __syncthreads();
}

// every wave has drained its stores, then ONE thread hands the ticket on
__device__ __forceinline__ void ticket_pass(unsigned* const tk, const unsigned next)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the barrier alone waits for no store (hipcc emits lgkmcnt(0) in front of s_barrier)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(tk, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the I-side sums of a task (registers) -> added into their lane of acc at block I; contribution number `seq` of the nb of that block.
// (Measured: awaiting both turns of a task first and adding both sums in one go — one round trip instead of two — is 0.3 % SLOWER at
// N = 262144, and 1.5 % slower on the equal-mass path, than the two additions one after the other: profiles/r06b_inplace_ab.txt.)
template <class M, bool EQ>
__device__ __forceinline__ void sym_add_i(M& t, const SymParams& p, const int ibase, const int I, const unsigned seq, const float m0)
{
    if (EQ) t.scale(m0);
    const unsigned L = (unsigned)p.acc_lanes, lane = seq & (L - 1u), turn = seq / L;   // L is a power of two
    unsigned* const tk = p.tickets + (unsigned)I * L + lane;
    const __amdgpu_buffer_rsrc_t acc = acc_lane_rsrc(p.acc + (size_t)lane * p.acc_stride, p.ni);
    ticket_wait(tk, turn, p.tickets + kTicketWords, p.err, p.acc_timeout_us);
#pragma unroll
    for (int k = 0; k < M::BPL; ++k) {
        const int i = ibase + k * 64;
        if (i < p.ni) {
            float4 v = t.acc(k);
            v.w = 0.0f;
            if (turn) {   // (a lane's first contribution stores: the lanes need no clearing)
                const float4 o = load_f4_agent(acc, i);
                v.x += o.x; v.y += o.y; v.z += o.z;
            }
            store_f4_agent(acc, i, v);
        }
    }
    ticket_pass(tk, seq + L >= (unsigned)p.nbi ? 0u : turn + 1u);   // the lane's last contribution of the nb: back to zero
}

// the J-side sums of a task (LDS) -> added into their lane of acc at block J
template <class M, int W, bool EQ>
__device__ __forceinline__ void sym_add_j(const float4* const sh, const SymParams& p, const int J, const unsigned seq, const float m0)
{
    constexpr int B = 64 * W * M::BPL;
    const unsigned L = (unsigned)p.acc_lanes, lane = seq & (L - 1u), turn = seq / L;
    unsigned* const tk = p.tickets + (unsigned)J * L + lane;
    const __amdgpu_buffer_rsrc_t acc = acc_lane_rsrc(p.acc + (size_t)lane * p.acc_stride, p.ni);
    ticket_wait(tk, turn, p.tickets + kTicketWords, p.err, p.acc_timeout_us);
#pragma unroll
    for (int e = threadIdx.x; e < B; e += 64 * W) {
        const int j = J * B + e;
        if (j < p.ni) {
            float4 v = sh[e];
            if (EQ) { v.x *= m0; v.y *= m0; v.z *= m0; }
            v.w = 0.0f;
            if (turn) {
                const float4 o = load_f4_agent(acc, j);
                v.x += o.x; v.y += o.y; v.z += o.z;
            }
            store_f4_agent(acc, j, v);
        }
    }
    ticket_pass(tk, seq + L >= (unsigned)p.nbi ? 0u : turn + 1u);
}

// the I-side sums of a task (registers of the stationary bodies) -> slab J of the I range
template <class M, bool EQ>
__device__ __forceinline__ void sym_store_i(M& t, const SymParamsT<typename M::V4, typename M::S>& p, const int ibase, const int J,
                                            const typename M::S m0)
{
    if (EQ) t.scale(m0);
    typename M::V4* const out_i = p.slabs_i + (size_t)J * p.stride_i;
#pragma unroll
    for (int k = 0; k < M::BPL; ++k) {
        const int i = ibase + k * 64;
        if (i < p.ni) out_i[i] = t.acc(k);
    }
}

// the J-side sums of a task (LDS) -> slab I of the J range
template <class M, int W, bool EQ>
__device__ __forceinline__ void sym_store_j(const typename M::V4* const sh, const SymParamsT<typename M::V4, typename M::S>& p, const int I,
                                            const int J, const int nj, const typename M::S m0)
{
    constexpr int B = 64 * W * M::BPL;
    typename M::V4* const out_j = p.slabs_j + (size_t)I * p.stride_j;
    for (int e = threadIdx.x; e < B; e += 64 * W) {
        const int j = J * B + e;
        if (j < nj) {
            typename M::V4 a = sh[e];
            if (EQ) { a.x *= m0; a.y *= m0; a.z *= m0; }
            a.w = 0;
            out_j[j] = a;
        }
    }
}

template <class M, int W, bool EQ, int CASE>
__device__ __forceinline__ void force_sym_tile(const SymParamsT<typename M::V4, typename M::S>& p, typename M::V4* const sh,
                                               const typename M::S m0)
{
    constexpr int BPL = M::BPL;
    constexpr int B = 64 * W * BPL;
    constexpr int NCH = B / 64;
    using V4 = typename M::V4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int task = p.task0 + (int)blockIdx.x;
    int I, J;
    bool diag = false;
    constexpr bool kSq = CASE == kSymSquare || CASE == kSymTicket;   // one range against itself, nothing else compiled in
    bool rect = CASE == kSymRect;
    if constexpr (CASE == kSymGeneral) rect = p.rect;
    if (rect) {
        I = task % p.nbi;
        J = task / p.nbi;
    } else {
        const int nb = p.nbi;
        const int npair = nb * (nb - 1) / 2;
        diag = task >= npair;
        if (diag) {
            I = J = task - npair;
        } else {
            const float q = 2.0f * nb - 1.0f;
            I = (int)((q - __builtin_sqrtf(q * q - 8.0f * (float)task)) * 0.5f);
            if (I < 0) I = 0;
            if (I > nb - 2) I = nb - 2;
            while (I < nb - 2 && sym_row_offset(I + 1, nb) <= task) ++I;
            while (I > 0 && sym_row_offset(I, nb) > task) --I;
            J = I + 1 + (task - sym_row_offset(I, nb));
            if constexpr (CASE == kSymTicket) {   // the same list read anti-diagonal by anti-diagonal: row r holds d = r + 1, entry c is (c, c + d)
                const int d = I + 1, c0 = J - I - 1;
                I = c0;
                J = c0 + d;
            }
        }
    }

    // a body past the end of its range adds exactly +-0 to every real body (pad4), and what it collects itself is never stored
    const V4* const xi = p.x + p.i0;
    const int nj = kSq ? p.ni : p.nj;
    M t;
    t.set_eps2(p.eps2);
    const int ibase = I * B + w * (64 * BPL) + lane;  // index within the I range
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        const int i = ibase + k * 64;
        // (the two spellings of one address are kept apart: hipcc schedules the rotation pass of the general kernel differently —
        //  35 more s_nop per SymPacked<10> instantiation — when its I-side loads are written through the pre-offset pointer)
        if constexpr (CASE == kSymGeneral) t.set(k, i < p.ni ? p.x[p.i0 + i] : pad4<V4, EQ>());
        else t.set(k, i < p.ni ? xi[i] : pad4<V4, EQ>());
    }
    const int rot = ((lane + 16) & 63) << 2;
    const int jbase = J * B + lane;  // index within the J run

    auto fetch = [&](int c) {
        const int j = jbase + c * 64;
        if constexpr (kSq) {
            return j < p.ni ? xi[j] : pad4<V4, EQ>();
        } else {
            int ja = p.j0 + j;
            if (p.wrap && ja >= p.wrap) ja -= p.wrap;
            return j < p.nj ? p.x[ja] : pad4<V4, EQ>();
        }
    };
    if (!diag) {
#pragma unroll
        for (int r = 0; r < BPL; ++r) sh[r * (64 * W) + tid] = zero4<V4>();  // B = BPL * 64*W
        __syncthreads();
    }
    int c = w * BPL;  // chunk of this wave in round 0 (distinct per wave, NCH = W*BPL chunks)
    V4 nxt = fetch(c);
    for (int q = 0; q < NCH; ++q) {
        V4 bj = nxt;
        const int cn = (c + 1 == NCH) ? 0 : c + 1;
        if (q + 1 < NCH) nxt = fetch(cn);
        if (diag) {
            V4 aj = zero4<V4>();
            for (int ph = 0; ph < 4; ++ph) {
                sym_row_pass<false, EQ>(t, bj, aj);
                bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot);
                if (!EQ) bj.w = next_row(bj.w, rot);
            }
        } else {
            V4 aj = sh[c * 64 + lane];
            for (int ph = 0; ph < 4; ++ph) {
                sym_row_pass<true, EQ>(t, bj, aj);
                bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot);
                if (!EQ) bj.w = next_row(bj.w, rot);
                aj.x = next_row(aj.x, rot); aj.y = next_row(aj.y, rot); aj.z = next_row(aj.z, rot);
            }
            sh[c * 64 + lane] = aj;
            __syncthreads();
        }
        c = cn;
    }
    // (the order of the two stores is part of what was measured per geometry: kept)
    if constexpr (CASE == kSymTicket) {
        // contribution numbers along the anti-diagonal list (tests/test_abi.py checks the closed forms against a walk of the list)
        const int nb = p.nbi, d = J - I;
        const unsigned seq_i = diag ? (unsigned)(nb - 1) : (unsigned)((d - 1) + (d < I ? d : I));
        sym_add_i<M, EQ>(t, p, ibase, I, seq_i, m0);
        if (!diag) sym_add_j<M, W, EQ>(sh, p, J, (unsigned)((d - 1 < nb - 1 - J ? d - 1 : nb - 1 - J) + (d - 1)), m0);
    } else if constexpr (CASE == kSymGeneral) {
        sym_store_i<M, EQ>(t, p, ibase, J, m0);
        if (!diag) sym_store_j<M, W, EQ>(sh, p, I, J, nj, m0);
    } else {
        if (!diag) sym_store_j<M, W, EQ>(sh, p, I, J, nj, m0);
        sym_store_i<M, EQ>(t, p, ibase, J, m0);
    }
}

// the equal-mass decision, once per workgroup (eq_uniform reads the verdict of nbk::mass_scan), then the tile
#define NBK_SYM_KERNEL_BODY(CASE)                                                                    \
    __shared__ typename M::V4 sh[64 * W * M::BPL];                                                   \
    double m0 = 0.0;                                                                                 \
    if (eq_uniform(p.eqm, p.eq_gen, &m0)) force_sym_tile<M, W, true, CASE>(p, sh, (typename M::S)m0); \
    else force_sym_tile<M, W, false, CASE>(p, sh, (typename M::S)m0)

template <class M, int W, int CASE>
__device__ __forceinline__ void force_sym_body(const SymParamsT<typename M::V4, typename M::S>& p)
{
    NBK_SYM_KERNEL_BODY(CASE);
}

template <class M, int W, int MINW = 1>
__global__ void __launch_bounds__(64 * W, MINW) force_sym(const SymParamsT<typename M::V4, typename M::S> p)
{
    force_sym_body<M, W, kSymGeneral>(p);
}

template <class M, int W>
__global__ void __launch_bounds__(64 * W, 1) force_sym_square(const SymParamsT<typename M::V4, typename M::S> p)
{
    NBK_SYM_KERNEL_BODY(kSymSquare);
}

template <class M, int W>
__global__ void __launch_bounds__(64 * W, 1) force_sym_rect(const SymParamsT<typename M::V4, typename M::S> p)
{
    NBK_SYM_KERNEL_BODY(kSymRect);
}

// the square case with the sums added in place (fp32 only): p.acc / p.tickets / p.err instead of slabs
template <class M, int W>
__global__ void __launch_bounds__(64 * W, 1) force_sym_ticket(const SymParams p)
{
    NBK_SYM_KERNEL_BODY(kSymTicket);
}

// ---------------------------------------------------------------------------------------
// symmetric flavour in RUNS — the same pair arithmetic for systems of 16k ... 128k bodies, where whole block pairs are
// too coarse a unit of work for 1024 SIMDs
// ---------------------------------------------------------------------------------------
//
// One wave is a worker of its own (no LDS, no barrier). The unit of work is (I-block of 64*BPL bodies, ONE chunk of 64
// J bodies): row I of the unit list holds the chunks from the block's own first chunk to the last chunk of the system
// (own chunks one-sided, later chunks symmetric), rows are laid end to end, and worker g takes units [g*L, (g+1)*L) — a run
// that may continue into the next row. Per run and row the worker loads the I-block once, streams its chunks through the
// register rotation, stores every chunk's J-side sums straight from registers (one float4 per lane) and the I-side sums
// once at the end. Slabs of body k (I-block K): index I' < K holds the J-side sums from row I', index K + s the I-side
// sums of the s-th worker that touched row K — indices 0 .. K + nseg(K) - 1, all written, added in index order.


struct RunParams {
    const float4* x;
    float4* slabs;
    int n, stride;
    int nbi;       // I-blocks
    long nunits;   // run_prefix(nbi)
    RunLayout r;
    float eps2;
    const MassInfo* eqm;   // mass_scan's verdict on the n bodies (nullptr: general path)
    unsigned int eq_gen;
};

template <class M, bool EQ>
__device__ __forceinline__ void force_sym_run_t(const RunParams& p, const float m0)
{
    constexpr int BPL = M::BPL;
    const int lane = threadIdx.x;
    const long g = blockIdx.x;
    long u = g * p.r.L;
    const long u1 = (u + p.r.L < p.nunits) ? u + p.r.L : p.nunits;
    if (u >= u1) return;
    // row of the first unit: the largest I with run_prefix(I) <= u
    int I;
    {
        int lo = 0, hi = p.nbi - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (run_prefix(mid, p.r) <= u) lo = mid; else hi = mid - 1;
        }
        I = lo;
    }
    const int rot = ((lane + 16) & 63) << 2;
    auto fetch = [&](int c) {
        const int j = c * 64 + lane;
        return j < p.n ? p.x[j] : pad4<float4, EQ>();
    };
    M t;
    t.set_eps2(p.eps2);
    for (; u < u1; ++I) {
        const long row0 = run_prefix(I, p.r), row1 = run_prefix(I + 1, p.r);
        const long seg1 = u1 < row1 ? u1 : row1;
        int c = I * BPL + (int)(u - row0);
        const int ibase = I * (64 * BPL) + lane;
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const int i = ibase + k * 64;
            t.set(k, i < p.n ? p.x[i] : pad4<float4, EQ>());
        }
        float4* const out_j = p.slabs + (size_t)I * p.stride;
        float4 nxt = fetch(c);
        for (; u < seg1; ++u, ++c) {
            float4 bj = nxt;
            if (u + 1 < seg1) nxt = fetch(c + 1);
            if (c < (I + 1) * BPL) {  // a chunk of the block itself: both orders of every pair occur, one side each
                float4 aj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                for (int ph = 0; ph < 4; ++ph) {
                    sym_row_pass<false, EQ>(t, bj, aj);
                    bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot);
                    if (!EQ) bj.w = next_row(bj.w, rot);
                }
            } else {
                float4 aj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                for (int ph = 0; ph < 4; ++ph) {
                    sym_row_pass<true, EQ>(t, bj, aj);
                    bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot);
                    if (!EQ) bj.w = next_row(bj.w, rot);
                    aj.x = next_row(aj.x, rot); aj.y = next_row(aj.y, rot); aj.z = next_row(aj.z, rot);
                }
                const int j = c * 64 + lane;  // back in the home lane after four row moves
                if (EQ) { aj.x *= m0; aj.y *= m0; aj.z *= m0; }
                aj.w = 0.0f;
                if (j < p.n) out_j[j] = aj;
            }
        }
        const int seg = (int)(g - row0 / p.r.L);  // this worker's ordinal among the workers of row I
        float4* const out_i = p.slabs + (size_t)(I + seg) * p.stride;
        if (EQ) t.scale(m0);
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const int i = ibase + k * 64;
            if (i < p.n) out_i[i] = t.acc(k);
        }
    }
}

template <class M>
__global__ void __launch_bounds__(64, M::BPL >= 10 ? 2 : 3) force_sym_run(const RunParams p)   // (the waves per SIMD the cost model counts on)
{
    double m0 = 0.0;
    if (eq_uniform(p.eqm, p.eq_gen, &m0)) force_sym_run_t<M, true>(p, (float)m0);
    else force_sym_run_t<M, false>(p, 0.0f);
}

// ---------------------------------------------------------------------------------------
// symmetric flavour in BALANCED runs — small and mid systems (a few thousand ... ~50k bodies)
// ---------------------------------------------------------------------------------------
//
// The same unit list as force_sym_run (row I = the chunks from the I-block's own first chunk to the end of the system, rows end to
// end), but cut at the granularity of ONE ROTATION STEP: a unit is 64 steps (4 row phases x 16 rotations), the list is T = 64 x units
// steps long, and worker g (one wave, no LDS, no barrier) takes steps [g*L, (g+1)*L), L = ceil(T / workers) — every worker the same
// number of steps, every SIMD the same number of resident workers. That is what N = 8192 (the reference's N_BODIES) needs: 2112
// units over 1024 SIMDs is 2-or-3 units per SIMD when units are indivisible (a third of the machine idle), 66 steps per worker when
// they are not.
//   * A unit may be split between workers anywhere: each worker loads the chunk with the lane permutation of the row phase it
//     starts in (lane l holds chunk body (l + 16*phase) & 63), runs its part of the 16 rotations of a phase through compile-time DPP
//     steps guarded by wave-uniform range tests, and stores what the J bodies have collected SO FAR.
//   * Partial sums go to per-chunk INBOXES: inbox[c] is a run of 1-KiB records (64 float4) — first `pmax` records for every row
//     above the chunk's own block (record `piece` = which of the workers that share the unit wrote it; pieces that do not exist are
//     never written and stay zero from the one-time clear of the workspace), then one record per worker that touched the chunk's
//     own row (I-side sums). Every record has exactly one writer; nothing is atomic.
//   * WV consecutive workers form one workgroup. They finish together (equal work), and those that end in the same row hold sums
//     for the SAME I-block: they add them up through LDS (in worker order) and the first of them writes ONE record — the workgroup's
//     tile becomes 64*bpl x WV*L pairs, near square, which is what keeps the partial-sum volume at N*sqrt(2*workgroups) instead of
//     N*sqrt(2*waves) (N = 8192, 4 bodies per lane: 4.7 MB instead of 12.8 MB per step). Records of the other workers of such a run
//     are never written and never read.
//   * bal_reduce streams a chunk's records (no memory lookups: which records exist follows from a few integer divisions), P waves
//     taking every P-th item, adds them in a fixed order and then over the waves — reproducible run to run — and integrates.


struct BalParams {
    const float4* x;
    float4* inbox;   // ncht * smax records of 64 float4, cleared once when the layout is set up
    int n;
    BalLayout y;
    float eps2;
    const MassInfo* eqm;   // mass_scan's verdict on the n bodies (nullptr: general path)
    unsigned int eq_gen;
};

template <int S, bool SYM, bool EQ = false, class M>
__device__ __forceinline__ void sym_step_if(M& t, const typename M::V4& bj, typename M::V4& aj, const int ta, const int tb)
{
    if (S >= ta && S < tb) sym_step<S, SYM, EQ>(t, bj, aj);  // ta, tb are wave-uniform: a scalar branch around the step
}

// rotations [ta, tb) of one row phase
template <bool SYM, bool EQ = false, class M>
__device__ __forceinline__ void sym_row_range(M& t, const typename M::V4& bj, typename M::V4& aj, const int ta, const int tb)
{
    if (ta == 0 && tb == 16) {   // a whole phase: straight-line code, the scheduler may overlap consecutive rotations
        sym_row_pass<SYM, EQ>(t, bj, aj);
        return;
    }
    sym_step_if<0, SYM, EQ>(t, bj, aj, ta, tb);  sym_step_if<1, SYM, EQ>(t, bj, aj, ta, tb);  sym_step_if<2, SYM, EQ>(t, bj, aj, ta, tb);
    sym_step_if<3, SYM, EQ>(t, bj, aj, ta, tb);  sym_step_if<4, SYM, EQ>(t, bj, aj, ta, tb);  sym_step_if<5, SYM, EQ>(t, bj, aj, ta, tb);
    sym_step_if<6, SYM, EQ>(t, bj, aj, ta, tb);  sym_step_if<7, SYM, EQ>(t, bj, aj, ta, tb);  sym_step_if<8, SYM, EQ>(t, bj, aj, ta, tb);
    sym_step_if<9, SYM, EQ>(t, bj, aj, ta, tb);  sym_step_if<10, SYM, EQ>(t, bj, aj, ta, tb); sym_step_if<11, SYM, EQ>(t, bj, aj, ta, tb);
    sym_step_if<12, SYM, EQ>(t, bj, aj, ta, tb); sym_step_if<13, SYM, EQ>(t, bj, aj, ta, tb); sym_step_if<14, SYM, EQ>(t, bj, aj, ta, tb);
    sym_step_if<15, SYM, EQ>(t, bj, aj, ta, tb);
}

template <class M, int WV, bool EQ>
__device__ __forceinline__ void force_sym_bal_body_t(const BalParams& p, float4 (*const sh)[64 * M::BPL], int* const sh_row, const float m0)
{
    constexpr int BPL = M::BPL;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x * WV + w;
    const BalLayout& y = p.y;
    int s = g < y.nworkers ? g * y.L : y.nsteps;
    const int s1 = (s + y.L < y.nsteps) ? s + y.L : y.nsteps;
    int I = s < s1 ? bal_row_of_unit(s >> 6, y) : 0;
    int my_last_row = -1;
    const int rot = ((lane + 16) & 63) << 2;
    // chunk c as seen in row phase ph: lane l holds body (l + 16*ph) & 63 of the chunk
    auto fetch = [&](int c, int ph) {
        const int j = c * 64 + ((lane + 16 * ph) & 63);
        return j < p.n ? p.x[j] : pad4<float4, EQ>();
    };
    M t;
    t.set_eps2(p.eps2);
    for (; s < s1; ++I) {
        const int row0 = bal_row_prefix(I, y), row1 = bal_row_prefix(I + 1, y);   // units of row I: [row0, row1)
        const int seg1 = (row1 << 6) < s1 ? (row1 << 6) : s1;
        const int ibase = I * (64 * BPL) + lane;
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const int i = ibase + k * 64;
            t.set(k, i < p.n ? p.x[i] : pad4<float4, EQ>());
        }
        float4 nxt = fetch(I * BPL + ((s >> 6) - row0), (s & 63) >> 4);
        while (s < seg1) {
            const int u = s >> 6;
            const int c = I * BPL + (u - row0);
            const int q0 = s & 63;
            const int uend = ((u + 1) << 6) < seg1 ? ((u + 1) << 6) : seg1;
            const int q1 = q0 + (uend - s);              // this worker does steps [q0, q1) of the unit, 0 <= q0 < q1 <= 64
            const int ph0 = q0 >> 4, ph1 = (q1 - 1) >> 4;
            float4 bj = nxt;
            if (uend < seg1) nxt = fetch(c + 1, 0);     // the next unit of this worker starts at its first step
            float4 aj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (c < (I + 1) * BPL) {  // a chunk of the block itself: both orders of every pair occur, one side each
                for (int ph = ph0; ph <= ph1; ++ph) {
                    const int ta = ph == ph0 ? (q0 & 15) : 0, tb = ph == ph1 ? ((q1 - 1) & 15) + 1 : 16;
                    sym_row_range<false, EQ>(t, bj, aj, ta, tb);
                    if (ph < ph1) {
                        bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot);
                        if (!EQ) bj.w = next_row(bj.w, rot);
                    }
                }
            } else {
                for (int ph = ph0; ph <= ph1; ++ph) {
                    const int ta = ph == ph0 ? (q0 & 15) : 0, tb = ph == ph1 ? ((q1 - 1) & 15) + 1 : 16;
                    sym_row_range<true, EQ>(t, bj, aj, ta, tb);
                    if (ph < ph1) {
                        bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot);
                        if (!EQ) bj.w = next_row(bj.w, rot);
                        aj.x = next_row(aj.x, rot); aj.y = next_row(aj.y, rot); aj.z = next_row(aj.z, rot);
                    }
                }
                // what the chunk's bodies collected from this worker's part of the unit: lane l holds body (l + 16*ph1) & 63;
                // record = row I's run of the chunk's inbox, piece = this worker's ordinal among the workers sharing the unit
                const int piece = g - (int)(((unsigned)u << 6) / (unsigned)y.L);
                float4* const out_j = p.inbox + ((size_t)c * y.smax + I * y.pmax + piece) * 64;
                if (EQ) { aj.x *= m0; aj.y *= m0; aj.z *= m0; }
                aj.w = 0.0f;
                out_j[(lane + 16 * ph1) & 63] = aj;
            }
            s = uend;
        }
        if (EQ) t.scale(m0);   // the row's I-side sums are complete (t.set starts the next row from zero)
        if (s < s1) {
            // the worker goes on into the next row: these I-side sums are complete, one record in the inbox of each of the block's
            // chunks (after the J-side runs of the rows above)
            unsigned gf, gl;
            bal_row_workers(I, y, &gf, &gl);
            const int rec = I * y.pmax + (g - (int)gf);
#pragma unroll
            for (int k = 0; k < BPL; ++k) {
                const int c = I * BPL + k;
                if (c < y.ncht) p.inbox[((size_t)c * y.smax + rec) * 64 + lane] = t.acc(k);
            }
        } else {
            my_last_row = I;   // the last row of this worker: its sums meet those of its workgroup neighbours in LDS
#pragma unroll
            for (int k = 0; k < BPL; ++k) sh[w][k * 64 + lane] = t.acc(k);
        }
    }
    if (lane == 0) sh_row[w] = my_last_row;
    __syncthreads();
    // a run of consecutive workers that end in the same row: the first adds the run up (worker order) and writes one record
    if (my_last_row < 0 || (w > 0 && sh_row[w - 1] == my_last_row)) return;
    int run = 1;
    while (w + run < WV && sh_row[w + run] == my_last_row) ++run;
    unsigned gf, gl;
    bal_row_workers(my_last_row, y, &gf, &gl);
    const int rec = my_last_row * y.pmax + (g - (int)gf);
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        float4 a = sh[w][k * 64 + lane];
        for (int q = 1; q < run; ++q) {
            const float4 b = sh[w + q][k * 64 + lane];
            a.x += b.x; a.y += b.y; a.z += b.z;
        }
        const int c = my_last_row * BPL + k;
        if (c < y.ncht) p.inbox[((size_t)c * y.smax + rec) * 64 + lane] = a;
    }
}

template <class M, int WV>
__device__ __forceinline__ void force_sym_bal_body(const BalParams& p)
{
    __shared__ float4 sh[WV][64 * M::BPL];
    __shared__ int sh_row[WV];
    double m0 = 0.0;
    if (eq_uniform(p.eqm, p.eq_gen, &m0)) force_sym_bal_body_t<M, WV, true>(p, sh, sh_row, (float)m0);
    else force_sym_bal_body_t<M, WV, false>(p, sh, sh_row, 0.0f);
}

template <class M, int WV>
__global__ void __launch_bounds__(64 * WV) force_sym_bal(const BalParams p)
{
    force_sym_bal_body<M, WV>(p);
}

// Sum of one 64-body chunk's inbox (one workgroup of 64*P threads per chunk; wave p takes records p, p + P, ...), then either the
// integrate (x, v, a in place) or the accelerations alone. Order: records ascending within a wave, then the P waves in order —
// fixed, so results are reproducible run to run.
struct BalReduceParams {
    const float4* inbox;
    BalLayout y;
    int n;
    // integrate (mode 0): x, v, a of the bodies; accelerations only (mode 1): a (+= when accumulate)
    float4* x;
    float4* v;
    float4* a;
    float dt;
    int mode;
    int accumulate;
};

template <int P>
__global__ void __launch_bounds__(64 * P) bal_reduce(const BalReduceParams p)
{
#pragma clang fp contract(off)
    __shared__ float4 sh[P][64];
    const BalLayout& y = p.y;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x;
    const int K = c / y.bpl;
    const unsigned L = (unsigned)y.L;
    const float4* const box = p.inbox + (size_t)c * y.smax * 64 + lane;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto add = [&](const float4 q) { acc.x += q.x; acc.y += q.y; acc.z += q.z; };
    // the wave that will integrate asks for its bodies' state now: the loads fly while the records are summed
    const int i = c * 64 + lane;
    float4 v0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), x0 = v0;
    if (w == 0 && i < p.n) {
        if (p.mode == 0) { v0 = p.v[i]; x0 = p.x[i]; }
        else if (p.accumulate) v0 = p.a[i];
    }
    // J side: rows above the chunk's own block; unit (I, c) was shared by workers S/L .. (S+63)/L, one record each
    for (int I = w; I < K; I += 2 * P) {   // two rows at a time: up to 2*pmax loads in flight
        const int I2 = I + P;
        const unsigned Sa = ((unsigned)(bal_row_prefix(I, y) + (c - I * y.bpl))) << 6;
        const int npa = (int)((Sa + 63u) / L - Sa / L) + 1;
        int npb = 0;
        if (I2 < K) {
            const unsigned Sb = ((unsigned)(bal_row_prefix(I2, y) + (c - I2 * y.bpl))) << 6;
            npb = (int)((Sb + 63u) / L - Sb / L) + 1;
        }
        float4 qa[5], qb[5];   // pmax <= 5 (L >= 16)
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            qa[e] = e < npa ? box[(size_t)(I * y.pmax + e) * 64] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            qb[e] = e < npb ? box[(size_t)(I2 * y.pmax + e) * 64] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int e = 0; e < 5; ++e) if (e < npa) add(qa[e]);
#pragma unroll
        for (int e = 0; e < 5; ++e) if (e < npb) add(qb[e]);
    }
    // I side: the workers of row K that wrote a record (first of the row, first of each workgroup, a last worker that goes on)
    {
        unsigned gf, gl;
        bal_row_workers(K, y, &gf, &gl);
        const bool gl_ends = bal_last_row((int)gl, y) == K;
        const float4* const ibox = box + (size_t)K * y.pmax * 64;
        const unsigned wv = (unsigned)y.wv;
        // item 0 = gf; items 1 .. m = the multiples of wv in (gf, gl]; one more when gl itself writes and is not a multiple
        const unsigned first_mult = (gf / wv + 1u) * wv;
        const int m = first_mult <= gl ? (int)((gl - first_mult) / wv) + 1 : 0;
        const bool extra = gl > gf && (gl % wv) != 0 && bal_writes_iside(gl, gf, gl, gl_ends, y);
        const int items = 1 + m + (extra ? 1 : 0);
        auto worker_of = [&](int e) -> unsigned { return e == 0 ? gf : (e <= m ? first_mult + (unsigned)(e - 1) * wv : gl); };
        int e = w;
        for (; e + 3 * P < items; e += 4 * P) {
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = ibox[(size_t)(worker_of(e + k * P) - gf) * 64];
#pragma unroll
            for (int k = 0; k < 4; ++k) add(q[k]);
        }
        for (; e < items; e += P) add(ibox[(size_t)(worker_of(e) - gf) * 64]);
    }
    sh[w][lane] = acc;
    __syncthreads();
    if (w != 0) return;
    float4 a = sh[0][lane];
#pragma unroll
    for (int q = 1; q < P; ++q) {
        const float4 b = sh[q][lane];
        a.x += b.x; a.y += b.y; a.z += b.z;
    }
    a.w = 0.0f;
    if (i >= p.n) return;
    if (p.mode == 1) {
        if (p.accumulate) { a.x += v0.x; a.y += v0.y; a.z += v0.z; }
        p.a[i] = a;
        return;
    }
    p.a[i] = a;
    float4 v = v0;
    float4 x = x0;
    const float hdt = 0.5f * p.dt;
    v.x += hdt * a.x; v.y += hdt * a.y; v.z += hdt * a.z;
    x.x += p.dt * v.x; x.y += p.dt * v.y; x.z += p.dt * v.z;
    p.v[i] = v;
    p.x[i] = x;
}

// ---------------------------------------------------------------------------------------
// fused small-N step — one launch per step, no partial sums in memory
// ---------------------------------------------------------------------------------------
//
// Below a few thousand bodies a step is not bound by arithmetic alone: every kernel boundary costs 1.5-2 us, every partial-sum
// workspace has to be written by one launch and read back by the next. This kernel turns the one-sided evaluation round: a WAVE
// owns T target bodies (held by every lane, packed two per register pair) and its 64 LANES split the sources — lane l takes
// sources l, l + 64, ... of each LDS tile — so the complete sum of a target never leaves the wave: four DPP row rotations and
// four v_readlane (fixed order: reproducible) end the force part, and lanes 0 .. T-1 integrate their target right there. N = 8192
// with T = 2 is 4096 waves, one workgroup of 16 per CU, four per SIMD. Positions are read by every wave for the whole launch, so
// the advanced positions go to a SECOND array (xout); the host alternates the two arrays from step to step (nbody_step keeps the
// spare one and copies back after an odd number of steps). 12 packed ops + 2 v_rsq_f32 per two pairs, the one-sided count (35
// cycles per 64 pairs with the LDS reads): 16 us of VALU work at N = 8192 in a 20.5 us step, against 25.7 us for force + partial
// sums + reduce in two launches.
// A float4 stored / loaded at SYSTEM scope (two 64-bit relaxed atomics: global_store_dwordx2 sc0 sc1 — through the L2 to memory)
__device__ __forceinline__ void store_f4_system(float4* dst, const float4 v)
{
    unsigned long long* const q = reinterpret_cast<unsigned long long*>(dst);
    const unsigned long long lo = (unsigned long long)__builtin_bit_cast(unsigned, v.x) | ((unsigned long long)__builtin_bit_cast(unsigned, v.y) << 32);
    const unsigned long long hi = (unsigned long long)__builtin_bit_cast(unsigned, v.z) | ((unsigned long long)__builtin_bit_cast(unsigned, v.w) << 32);
    __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ float4 load_f4_system(const float4* src)
{
    unsigned long long* const q = reinterpret_cast<unsigned long long*>(const_cast<float4*>(src));
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_float4(__builtin_bit_cast(float, (unsigned)lo), __builtin_bit_cast(float, (unsigned)(lo >> 32)),
                       __builtin_bit_cast(float, (unsigned)hi), __builtin_bit_cast(float, (unsigned)(hi >> 32)));
}

struct FusedParams {
    const float4* xin;   // positions at the start of the step
    float4* xout;        // advanced positions (another array: every wave reads xin until the end of the launch)
    float4* v;           // velocities, in place
    float4* a;           // accelerations (output)
    int n;
    float dt, eps2;
    // in-place variant only (step_fused<.., INPLACE = true>): xin == the caller's array, xout == the context's spare array
    unsigned* sync;                 // FusedSync words (device memory, zero between launches)
    unsigned char* fb;              // one byte per wave of the grid: 1 = this wave's advanced positions are in xout, not in place
    unsigned long long* host_word;  // host-mapped, ONE 64-bit store once everything (repair included) is in memory: low half = done_value, high half = fall-backs of this launch
    unsigned done_value;
    int force_fallback;             // test hook: every wave takes the fall-back path
};


// dst[i] = src[i]: puts the positions of an odd step back into the caller's array (a launch on the same stream costs less than a
// device-to-device hipMemcpyAsync of 128 KiB)
__global__ void __launch_bounds__(256) copy_bodies(float4* __restrict__ dst, const float4* __restrict__ src, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// The clock a timed force launch ran at, WITHOUT a stamp inside the measured kernel (MI355X_MICROARCH.md: per-segment stamps cost
// cycles): clock_begin in front of the launch and clock_end behind it, on its stream (nbi::time_mark with the clock option on).
// clock_begin: sixteen waves note their CU's shader-cycle counter and the 100-MHz counter in device scratch. clock_end: one wave
// per workgroup, enough workgroups to land on every CU; a wave that finds a begin record of its own CU writes the two differences
// to the launch's host-mapped record under its XCD (CUs of one XCD give the same difference: any writer will do). Cycles / ticks x
// 100 MHz = the shader clock that XCD held while the force kernel ran; the interval includes the two launch boundaries.
__device__ __forceinline__ ClockStamp read_clock()
{
    ClockStamp s;
    s.cycles = __builtin_amdgcn_s_memtime();
    s.ticks = __builtin_amdgcn_s_memrealtime();
    s.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // HW_REG_XCC_ID, bits 3:0
    s.hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID: cu_id 11:8, sh_id 12, se_id 15:13
    return s;
}

__global__ void __launch_bounds__(64) clock_begin(ClockStamp* scratch)
{
    if (threadIdx.x == 0) scratch[blockIdx.x] = read_clock();
}

__global__ void __launch_bounds__(64) clock_end(const ClockStamp* scratch, ClockDelta* out)
{
    if (threadIdx.x != 0) return;
    const ClockStamp e = read_clock();
    for (int k = 0; k < kClockBeginWgs; ++k) {
        const ClockStamp b = scratch[k];
        if (b.xcc != e.xcc || ((b.hw_id ^ e.hw_id) & 0xff00u) != 0) continue;   // another CU: its counter started elsewhere
        __hip_atomic_store(out->dcycles + e.xcc, e.cycles - b.cycles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(out->dticks + e.xcc, e.ticks - b.ticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
    }
}

// One 64-bit word to host-mapped memory: launched behind the last kernel of a synchronous call (nbody_simulate on the paths that are
// not the fused step), so that the host can wait on the word instead of a stream synchronisation. The launch boundary in front of it is
// what makes the step's results visible first: the runtime's release between two kernels of one stream writes the L2s back.
__global__ void __launch_bounds__(64) host_signal(unsigned long long* host_word, unsigned long long value)
{
    if (threadIdx.x == 0) __hip_atomic_store(host_word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// INPLACE = true: the same step with the advanced positions written back INTO the array they were read from — one launch, no spare
// array to alternate with, no copy-back launch after an odd number of steps (what a caller who steps once per synchronous call, like
// the reference's loop main.cpp:146-156, pays every call). In place is only safe once EVERY workgroup has read everything it will
// ever read; the launch is one workgroup per CU, normally all resident at once, but nothing guarantees that (another process or
// stream may hold CUs), so the protocol NEVER WAITS for a workgroup that has not started:
//   * a workgroup counts itself in `readers` (release) once its last source tile has landed in LDS — a quarter of its run time before
//     it ends;
//   * at its end a wave looks at `readers` (acquire; a few polls at most, bounded): all workgroups counted -> its positions go in place;
//     otherwise -> to the spare array, with a mark in `fb` (a workgroup that starts later still reads the OLD positions: correct);
//   * every workgroup counts itself in `finished` (release) after its stores; the one that completes the count knows that all reads
//     and all writes of the launch are done: it moves marked positions into place (rare), zeroes the counters for the next launch,
//     and writes the host-mapped word — the caller may spin on that instead of paying a stream synchronisation.
// Every path ends without waiting on another wave: no deadlock whatever shares the GPU. Same arithmetic and the same bits as the
// two-array kernel (tests: test_fused_step_inplace_*).
template <int T, int WV, int TILE, int UNROLL = 8, int MINW = 1, bool INPLACE = false>
__global__ void __launch_bounds__(64 * WV, MINW) step_fused(const FusedParams p)
{
    static_assert(T % 2 == 0 && TILE % (64 * WV) == 0, "packed targets, whole loads per thread");
    constexpr int LPT = TILE / (64 * WV);
    __shared__ float4 sh[2][TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = ((int)blockIdx.x * WV + w) * T;   // first target of this wave
    MathPacked<T> t;
    t.set_eps2(p.eps2);
#pragma unroll
    for (int k = 0; k < T; ++k) {
        int i = i0 + k;
        if (i > p.n - 1) i = p.n - 1;     // surplus targets compute a copy, never store
        t.set(k, p.xin[i]);
    }
    // the lanes that will integrate ask for their body's state now: the loads fly during the force loop instead of after it
    const int i = i0 + lane;
    const bool mine = lane < T && i < p.n;
    float4 x_own = make_float4(0.0f, 0.0f, 0.0f, 0.0f), v_own = x_own;
    if (mine) { x_own = p.xin[i]; v_own = p.v[i]; }
    float4 pre[LPT];
    auto fetch = [&](int jt) {
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            const int j = jt + l * (64 * WV) + tid;
            pre[l] = j < p.n ? p.xin[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // past the end: massless, adds exactly 0
        }
    };
    fetch(0);
    int buf = 0;
    for (int jt = 0; jt < p.n; jt += TILE, buf ^= 1) {
#pragma unroll
        for (int l = 0; l < LPT; ++l) sh[buf][l * (64 * WV) + tid] = pre[l];
        __syncthreads();   // one barrier per tile: the other buffer is rewritten only after every wave has passed the next one
        if (jt + TILE < p.n) fetch(jt + TILE);
        else if (INPLACE && tid == 0)
            // The last tile is in LDS, x_own and the targets long since in registers: this workgroup has READ all it ever will. Every
            // load of the workgroup has returned its data before the barrier above (the data went to LDS), so a relaxed atomic — agent
            // scope: one counter for all eight XCDs — is ordered behind them by construction; a release here would add an L2
            // write-back to the critical path for stores that do not exist yet.
            __hip_atomic_fetch_add(p.sync + kFusedReaders, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int left = p.n - jt;
        if (left >= TILE) {
#pragma unroll UNROLL
            for (int jj = 0; jj < TILE / 64; ++jj) t.pair(sh[buf][jj * 64 + lane]);   // this lane's sources of the tile
        } else {   // the last tile may be short: only its occupied 64-body rows
            const int rows = (left + 63) / 64;
#pragma unroll 2
            for (int jj = 0; jj < rows; ++jj) t.pair(sh[buf][jj * 64 + lane]);
        }
    }
    // the sums of the 64 lanes, in a fixed order (same bits every run): four DPP row rotations leave every lane with the total of
    // its 16-lane row, the four row totals are read as scalars and added in row order
    auto wave_sum = [](float v) {
        v += ror<8>(v);
        v += ror<4>(v);
        v += ror<2>(v);
        v += ror<1>(v);
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
        return ((r0 + r1) + r2) + r3;
    };
    // (in place: the look at `readers` is issued before the reduction, so its round trip to the L2 runs beside it)
    unsigned seen = 0;
    if (INPLACE && mine) seen = __hip_atomic_load(p.sync + kFusedReaders, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float4 tot[T];
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const float4 s = t.acc(k);
        tot[k] = make_float4(wave_sum(s.x), wave_sum(s.y), wave_sum(s.z), 0.0f);
    }
    if (!INPLACE && !mine) return;
    if (mine) {
        float4 acc = tot[0];
#pragma unroll
        for (int k = 1; k < T; ++k) if (lane == k) acc = tot[k];
        bool in_place = false;
        if (INPLACE) {   // have all workgroups of the launch read everything? (a few polls, then the fall-back: never an unbounded wait)
            // (relaxed: nothing is READ on the strength of this value — the store below is issued only after the load has returned)
            for (int poll = 0; poll < 16 && seen != gridDim.x; ++poll) {
                __builtin_amdgcn_s_sleep(8);
                seen = __hip_atomic_load(p.sync + kFusedReaders, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            in_place = seen == gridDim.x && !p.force_fallback;
        }
        acc.w = 0.0f;
        float4 x = x_own;
        float4 v = v_own;
        {
#pragma clang fp contract(off)
            const float hdt = 0.5f * p.dt;
            v.x += hdt * acc.x; v.y += hdt * acc.y; v.z += hdt * acc.z;
            x.x += p.dt * v.x; x.y += p.dt * v.y; x.z += p.dt * v.z;
        }
        if (!INPLACE) {
            p.a[i] = acc;
            p.v[i] = v;
            p.xout[i] = x;   // .w (mass) carried through
        } else {
            // In place, results are stored THROUGH the L2 (system scope): the workgroup that ends the launch tells the host so while the
            // kernel is still winding down, and eight XCDs' L2s are not coherent for ordinary stores; a write-back fence per workgroup
            // instead (buffer_wbl2) cost 8 us per step (profiles/r04_fused_inplace_wbl2.txt).
            store_f4_system(p.a + i, acc);
            store_f4_system(p.v + i, v);
            store_f4_system((in_place ? const_cast<float4*>(p.xin) : p.xout) + i, x);
            if (!in_place && lane == 0) {
                __hip_atomic_store(p.fb + ((int)blockIdx.x * WV + w), (unsigned char)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_fetch_add(p.sync + kFusedFinished, 1u << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!INPLACE) return;
    // Every workgroup counts itself out once its stores are complete; the workgroup that completes the count ends the launch. The
    // barrier alone does NOT wait for stores (the compiler puts s_waitcnt lgkmcnt(0) in front of s_barrier, nothing for vmcnt), so
    // EVERY wave drains its own first: on gfx950 stores and atomics without return count in vmcnt, and a system-scope (sc0 sc1)
    // store is acknowledged once memory has it. That orders, for each wave: results (and, on the fall-back path, xout, the fb mark and
    // the 1 << 16 add) -> this wait -> barrier -> the workgroup's count-out below. The "memory" clobber keeps the compiler from moving a
    // store across the wait; tests/test_isa_protocol.py checks the instruction in the shipped code object.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (w != 0) return;   // the first wave alone ends the workgroup's part (and, if it is the last workgroup's, the launch)
    unsigned before = 0;
    if (lane == 0) before = __hip_atomic_fetch_add(p.sync + kFusedFinished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    before = (unsigned)__builtin_amdgcn_readfirstlane((int)before);
    if ((before & 0xffffu) + 1 != gridDim.x) return;
    const unsigned nfb = before >> 16;
    if (nfb != 0) {   // rare: some waves could not write in place; all reads of the launch are over now, so their positions go home
        const int nwaves = (int)gridDim.x * WV;
        float4* const x = const_cast<float4*>(p.xin);
        for (int wq = lane; wq < nwaves; wq += 64) {
            if (!__hip_atomic_load(p.fb + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) continue;
            __hip_atomic_store(p.fb + wq, (unsigned char)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int k = 0; k < T; ++k) {
                const int q = wq * T + k;
                if (q < p.n) store_f4_system(x + q, load_f4_system(p.xout + q));
            }
        }
        __builtin_amdgcn_s_waitcnt(0);   // these stores complete before the host hears of the launch (one wave: no barrier needed)
    }
    if (lane == 0) {
        // every result of the launch is in memory by now (see above): tell the host first, tidy up afterwards
        if (p.host_word)
            __hip_atomic_store(p.host_word, (unsigned long long)p.done_value | ((unsigned long long)nfb << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.sync + kFusedReaders, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.sync + kFusedFinished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nfb) p.sync[kFusedFallbacksTotal] += nfb;
    }
}

// ---------------------------------------------------------------------------------------
// strict flavour — the reference's arithmetic, operation by operation
// ---------------------------------------------------------------------------------------

// LEGACY = the older snapshot's arithmetic (Sim-Without-OpenGL-Integration/kernel.cu:5-25 with its
// constants.h:14 `#define EPS2 0.002`, a double literal): the float sum r.r is promoted to double
// for the `+ EPS2`, then rounded back to float.
template <bool LEGACY>
__device__ __forceinline__ void pair_strict(const float4 bi, const float4 bj, float& ax, float& ay,
                                            float& az, const float eps2)
{
#pragma clang fp contract(off)
    // kernel.cu:9-29 == validation.cpp:9-24
    const float rx = bj.x - bi.x;
    const float ry = bj.y - bi.y;
    const float rz = bj.z - bi.z;
    float d;
    if (LEGACY) d = (float)((double)(rx * rx + ry * ry + rz * rz) + 0.002);
    else d = rx * rx + ry * ry + rz * rz + eps2;
    const float denom = 1.0f / __builtin_sqrtf(d * d * d);  // correctly rounded sqrt and divide
    const float s = bj.w * denom;
    ax += rx * s;
    ay += ry * s;
    az += rz * s;
}

// One target per lane, sources in index order through one LDS tile. grid = ceil((i1-i0)/256).
template <int TILE, bool LEGACY = false>
__global__ void __launch_bounds__(kWG) force_strict(const ForceParams p)
{
#pragma clang fp contract(off)
    constexpr int LPT = TILE / kWG;
    __shared__ float4 sh[TILE];
    const int tid = threadIdx.x;
    const int i = p.i0 + blockIdx.x * kWG + tid;
    const int ic = (i < p.i1) ? i : p.i1 - 1;
    const float4 bi = p.x[ic];
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    if (p.accumulate) {
        const float4 o = p.out[ic - p.i0];
        ax = o.x; ay = o.y; az = o.z;
    }
    for (int jt = p.j0; jt < p.j1; jt += TILE) {
        __syncthreads();
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            const int j = jt + l * kWG + tid;
            const int js = (p.wrap && j >= p.wrap) ? j - p.wrap : j;
            if (j < p.j1) sh[l * kWG + tid] = p.x[js];
        }
        __syncthreads();
        const int cnt = (p.j1 - jt < TILE) ? (p.j1 - jt) : TILE;
        for (int jj = 0; jj < cnt; ++jj) {
            // validation.cpp:35 skips j == i; the older GPU kernel does not (its term is an exact 0)
            const int j = jt + jj;
            const int js = (p.wrap && j >= p.wrap) ? j - p.wrap : j;
            if (LEGACY || js != ic) pair_strict<LEGACY>(bi, sh[jj], ax, ay, az, p.eps2);
        }
    }
    if (i < p.i1) p.out[i - p.i0] = make_float4(ax, ay, az, 0.0f);
}

// ---------------------------------------------------------------------------------------
// integrate — kernel.cu:116-129 == validation.cpp:40-49, plus the fixed-order slab sum
// ---------------------------------------------------------------------------------------

struct IntegrateParams {
    float4* x;             // positions of the OWN bodies (already offset to i0)
    float4* v;             // velocities of the own bodies
    float4* a;             // accelerations of the own bodies (output when nslab > 0)
    const float4* slabs;   // nslab partial-sum slabs, slab_stride apart (may alias a when nslab==1)
    int nslab;
    int slab_stride;
    int n;                 // own bodies
    float dt;
    RunLayout run;         // run.bi != 0: body i has run_slab_count(i) slabs instead of nslab
};

__global__ void __launch_bounds__(kWG) integrate(const IntegrateParams p)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * kWG + threadIdx.x;
    if (i >= p.n) return;
    float4 a;
    if (p.nslab > 0) {
        const int nslab = p.run.bi ? run_slab_count(i, p.run) : p.nslab;
        a = p.slabs[i];
        // the sum is taken in slab order whatever the unrolling: only the loads are batched
#pragma unroll 8
        for (int s = 1; s < nslab; ++s) {
            const float4 q = p.slabs[(size_t)s * p.slab_stride + i];
            a.x += q.x; a.y += q.y; a.z += q.z;
        }
        a.w = 0.0f;
        p.a[i] = a;
    } else {
        a = p.a[i];
    }
    float4 v = p.v[i];
    float4 x = p.x[i];
    const float hdt = 0.5f * p.dt;  // `0.5f * DT * a` parses as (0.5f*DT)*a
    v.x += hdt * a.x;
    v.y += hdt * a.y;
    v.z += hdt * a.z;
    x.x += p.dt * v.x;
    x.y += p.dt * v.y;
    x.z += p.dt * v.z;
    p.v[i] = v;
    p.x[i] = x;  // .w (mass) carried through untouched
}

// The older snapshot's integrate (Sim-Without-OpenGL-Integration/kernel.cu:68-80): float3 velocity,
// and `0.5 * DT` / `DT` with DT = 0.01 as DOUBLE literals, so each update is evaluated in double
// and rounded to float once; the acceleration array is zeroed afterwards (kernel.cu:78-80).
struct IntegrateLegacyParams {
    float4* x;
    float* v3;             // N packed float3
    float* a3;             // N packed float3 (left zeroed, as the old kernel leaves it)
    const float4* slabs;   // partial sums (nslab >= 1)
    int nslab;
    int slab_stride;
    int n;
};

__global__ void __launch_bounds__(kWG) integrate_legacy(const IntegrateLegacyParams p)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * kWG + threadIdx.x;
    if (i >= p.n) return;
    float4 a = p.slabs[i];
    for (int s = 1; s < p.nslab; ++s) {
        const float4 q = p.slabs[(size_t)s * p.slab_stride + i];
        a.x += q.x; a.y += q.y; a.z += q.z;
    }
    float vx = p.v3[3 * i + 0], vy = p.v3[3 * i + 1], vz = p.v3[3 * i + 2];
    float4 x = p.x[i];
    vx = (float)((double)vx + 0.5 * 0.01 * (double)a.x);
    vy = (float)((double)vy + 0.5 * 0.01 * (double)a.y);
    vz = (float)((double)vz + 0.5 * 0.01 * (double)a.z);
    x.x = (float)((double)x.x + 0.01 * (double)vx);
    x.y = (float)((double)x.y + 0.01 * (double)vy);
    x.z = (float)((double)x.z + 0.01 * (double)vz);
    p.v3[3 * i + 0] = vx; p.v3[3 * i + 1] = vy; p.v3[3 * i + 2] = vz;
    p.a3[3 * i + 0] = 0.0f; p.a3[3 * i + 1] = 0.0f; p.a3[3 * i + 2] = 0.0f;
    p.x[i] = x;
}

// out[i] = (accumulate ? out[i] : 0) + slabs[0][i] + slabs[1][i] + ... in slab order.
struct ReduceParams {
    float4* out;
    const float4* slabs;
    int nslab;
    int slab_stride;
    int n;
    int accumulate;
    RunLayout run;  // run.bi != 0: body i has run_slab_count(i) slabs instead of nslab
};

__global__ void __launch_bounds__(kWG) reduce_slabs(const ReduceParams p)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * kWG + threadIdx.x;
    if (i >= p.n) return;
    float4 a = p.accumulate ? p.out[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int nslab = p.run.bi ? run_slab_count(i, p.run) : p.nslab;
#pragma unroll 8
    for (int s = 0; s < nslab; ++s) {
        const float4 q = p.slabs[(size_t)s * p.slab_stride + i];
        a.x += q.x; a.y += q.y; a.z += q.z;
    }
    a.w = 0.0f;
    p.out[i] = a;
}

// ---------------------------------------------------------------------------------------
// fp64 variant (BASELINE.json configs[4]; the build's own, no reference analogue)
// ---------------------------------------------------------------------------------------

struct ForceParamsF64 {
    const double4* x;
    double4* out;
    int n;
    int slab_stride;
    double eps2;
};

// d^(-3/2) in double: v_rsq_f64 seed refined by two Newton steps, then cubed.
__device__ __forceinline__ double inv_cube_f64(const double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y * y * y;
}

// grid = (ceil(n / (256*BPL)), nslab); LDS tile of TILE double4 (32 B each).
template <int BPL, int TILE>
__global__ void __launch_bounds__(kWG) force_f64(const ForceParamsF64 p)
{
    constexpr int LPT = TILE / kWG;
    __shared__ double4 sh[TILE];
    const int tid = threadIdx.x;
    const int ibase = blockIdx.x * (kWG * BPL);
    double xi[BPL], yi[BPL], zi[BPL], ax[BPL], ay[BPL], az[BPL];
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        int i = ibase + k * kWG + tid;
        if (i > p.n - 1) i = p.n - 1;
        const double4 b = p.x[i];
        xi[k] = b.x; yi[k] = b.y; zi[k] = b.z;
        ax[k] = 0.0; ay[k] = 0.0; az[k] = 0.0;
    }
    int ja, jb;
    slab_range(0, p.n, TILE, gridDim.y, blockIdx.y, ja, jb);
    for (int jt = ja; jt < jb; jt += TILE) {
        __syncthreads();
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            const int j = jt + l * kWG + tid;
            sh[l * kWG + tid] = (j < jb) ? p.x[j] : make_double4(0.0, 0.0, 0.0, 0.0);
        }
        __syncthreads();
#pragma unroll 4
        for (int jj = 0; jj < TILE; ++jj) {
            const double4 bj = sh[jj];
#pragma unroll
            for (int k = 0; k < BPL; ++k) {
                const double rx = bj.x - xi[k], ry = bj.y - yi[k], rz = bj.z - zi[k];
                const double d = __builtin_fma(rz, rz, __builtin_fma(ry, ry, __builtin_fma(rx, rx, p.eps2)));
                const double f = bj.w * inv_cube_f64(d);
                ax[k] = __builtin_fma(rx, f, ax[k]);
                ay[k] = __builtin_fma(ry, f, ay[k]);
                az[k] = __builtin_fma(rz, f, az[k]);
            }
        }
    }
    double4* out = p.out + (size_t)blockIdx.y * p.slab_stride;
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        const int i = ibase + k * kWG + tid;
        if (i < p.n) out[i] = make_double4(ax[k], ay[k], az[k], 0.0);
    }
}

struct IntegrateParamsF64 {
    double4* x;
    double4* v;
    double4* a;
    const double4* slabs;
    int nslab;
    int slab_stride;
    int n;
    double dt;
};

__global__ void __launch_bounds__(kWG) integrate_f64(const IntegrateParamsF64 p)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * kWG + threadIdx.x;
    if (i >= p.n) return;
    double4 a = p.slabs[i];
#pragma unroll 8
    for (int s = 1; s < p.nslab; ++s) {
        const double4 q = p.slabs[(size_t)s * p.slab_stride + i];
        a.x += q.x; a.y += q.y; a.z += q.z;
    }
    a.w = 0.0;
    p.a[i] = a;
    double4 v = p.v[i];
    double4 x = p.x[i];
    const double hdt = 0.5 * p.dt;
    v.x += hdt * a.x; v.y += hdt * a.y; v.z += hdt * a.z;
    x.x += p.dt * v.x; x.y += p.dt * v.y; x.z += p.dt * v.z;
    p.v[i] = v;
    p.x[i] = x;
}

}  // namespace nbk

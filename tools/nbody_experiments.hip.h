// tools/nbody_experiments.hip.h — MEASURED ALTERNATIVES that the product does not ship (developer probes only).
//
// These kernels and arithmetic policies were written, measured against the shipped ones (tools/kbench.hip, tools/symbench.hip,
// tools/balbench.hip; results under profiles/ and in DESIGN.md 4) and NOT adopted. They build on the product's device header
// (same parameter blocks, helpers and kernel bodies) but are instantiated by nothing in libnbody_hip.so:
// tests/test_build_resources.py checks that every __global__ of the product header IS instantiated by the library, so
// experiments live here.
#pragma once
#include "../n-bodysimulation_amd/csrc/nbody_kernels.hip.h"

namespace nbk {

// Same arithmetic, sources read straight from global memory at a wave-uniform address: the
// compiler turns that into scalar loads (s_load_dwordx4..x16), so source bodies sit in SGPRs
// and cost neither LDS traffic nor barriers. Kept as a measured alternative to force_lds (tools/kbench.hip
// only: equal speed for packed maths, slower for scalar maths; it does not implement `wrap`).
template <class M, int UNROLL, int MINW>
__global__ void __launch_bounds__(kWG, MINW) force_sgpr(const ForceParams p)
{
    const int ibase = p.i0 + blockIdx.x * (kWG * M::BPL);
    M t;
    t.set_eps2(p.eps2);
    load_targets(p, ibase, t);
    int ja, jb;
    slab_range(p.j0, p.j1, UNROLL, gridDim.y, blockIdx.y, ja, jb);
    const float4* __restrict__ xs = p.x;
    int j = ja;
    for (; j + UNROLL <= jb; j += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) t.pair(xs[j + u]);
    }
    for (; j < jb; ++j) t.pair(xs[j]);
    store_targets(p, ibase, blockIdx.y, t);
}

// Scalar arithmetic for the rotation kernel: the row rotation can fold into v_sub_f32_dpp / v_mul_f32_dpp.
template <int BPL_>
struct SymScalar {
    using S = float;
    using V4 = float4;
    static constexpr int BPL = BPL_;
    float x[BPL], y[BPL], z[BPL], m[BPL];
    float ax[BPL], ay[BPL], az[BPL];
    float e2;

    __device__ __forceinline__ void set_eps2(const float eps2)
    {
        e2 = eps2;
        asm volatile("" : "+v"(e2));
    }
    __device__ __forceinline__ void set(int k, const float4 b)
    {
        x[k] = b.x; y[k] = b.y; z[k] = b.z; m[k] = b.w;
        ax[k] = 0.0f; ay[k] = 0.0f; az[k] = 0.0f;
    }
    __device__ __forceinline__ float4 acc(int k) const { return make_float4(ax[k], ay[k], az[k], 0.0f); }
    __device__ __forceinline__ void scale(const float s)
    {
#pragma unroll
        for (int k = 0; k < BPL; ++k) { ax[k] *= s; ay[k] *= s; az[k] *= s; }
    }
    template <bool SYM, bool EQ = false>
    __device__ __forceinline__ void pairs(const float sx, const float sy, const float sz, const float sm, float& tx,
                                          float& ty, float& tz)
    {
        tx = 0.0f; ty = 0.0f; tz = 0.0f;
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const float rx = sx - x[k];
            const float ry = sy - y[k];
            const float rz = sz - z[k];
            float d = __builtin_fmaf(rx, rx, e2);
            d = __builtin_fmaf(ry, ry, d);
            d = __builtin_fmaf(rz, rz, d);
            const float inv = __builtin_amdgcn_rsqf(d);
            const float w = inv * inv * inv;
            const float fi = EQ ? w : sm * w;
            ax[k] = __builtin_fmaf(rx, fi, ax[k]);
            ay[k] = __builtin_fmaf(ry, fi, ay[k]);
            az[k] = __builtin_fmaf(rz, fi, az[k]);
            if (SYM) {
                const float fj = EQ ? w : m[k] * w;
                if (k == 0) { tx = rx * fj; ty = ry * fj; tz = rz * fj; }
                else {
                    tx = __builtin_fmaf(rx, fj, tx);
                    ty = __builtin_fmaf(ry, fj, ty);
                    tz = __builtin_fmaf(rz, fj, tz);
                }
            }
        }
    }
};

// MEASURED ALTERNATIVE, not shipped (tools/symbench.hip with SYMBENCH_EQ=1, profiles/r03_symbench_local_decision_262144.txt): the
// equal-mass decision taken LOCALLY, per wave and per 64-body chunk, from the masses the wave holds and loads anyway — no scan launch,
// no verdict slot, no far-away padding. Equal to the scan-based kernel on equal masses (10.31-10.36 vs 10.24-10.33 ms per launch at
// N = 262144, one box) but 1 % SLOWER on unequal ones (11.36-11.38 vs 11.21-11.28: 242 VGPRs and another schedule of the rotation
// pass) — and unequal masses are what the reference's own initial conditions have. The scan costs 3 us per call; it stays. A wave whose stationary bodies all carry m0 (finite,
// of ordinary magnitude) accumulates its I-side sums in units of m0; a chunk that carries m0 too takes the equal-mass pass (its
// J-side sums are brought to true units when they meet the other waves' in LDS); any other chunk takes the general pass with its
// masses divided by m0. Padding bodies are massless as ever: their chunk (or their wave) simply is not uniform.
__device__ __forceinline__ bool wave_all(const bool ok) { return __builtin_amdgcn_ballot_w64(ok) == ~0ull; }

template <bool SYM, class M>
__device__ __forceinline__ void sym_chunk(M& t, typename M::V4 bj, typename M::V4& aj, const int rot, const bool eq)
{
    if (eq) {
        for (int ph = 0; ph < 4; ++ph) {
            sym_row_pass<SYM, true>(t, bj, aj);
            bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot);
            if (SYM) { aj.x = next_row(aj.x, rot); aj.y = next_row(aj.y, rot); aj.z = next_row(aj.z, rot); }
        }
    } else {
        for (int ph = 0; ph < 4; ++ph) {
            sym_row_pass<SYM, false>(t, bj, aj);
            bj.x = next_row(bj.x, rot); bj.y = next_row(bj.y, rot); bj.z = next_row(bj.z, rot); bj.w = next_row(bj.w, rot);
            if (SYM) { aj.x = next_row(aj.x, rot); aj.y = next_row(aj.y, rot); aj.z = next_row(aj.z, rot); }
        }
    }
}

template <class M, int W>
__global__ void __launch_bounds__(64 * W, 1) force_sym_square_local(const SymParamsT<typename M::V4, typename M::S> p, const int allow_eq)
{
    constexpr int BPL = M::BPL;
    constexpr int B = 64 * W * BPL;
    constexpr int NCH = B / 64;
    using V4 = typename M::V4;
    using S = typename M::S;
    __shared__ V4 sh[B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = p.nbi;
    const int task = p.task0 + (int)blockIdx.x;
    const int npair = nb * (nb - 1) / 2;
    const bool diag = task >= npair;
    int I, J;
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
    }
    const V4* const x = p.x + p.i0;
    const int n = p.ni;
    M t;
    t.set_eps2(p.eps2);
    const int ibase = I * B + w * (64 * BPL) + lane;
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        const int i = ibase + k * 64;
        t.set(k, i < n ? x[i] : zero4<V4>());
    }
    // the wave's own verdict on its stationary bodies
    S m0 = (S)0, inv_m0 = (S)0;
    bool eqI = false;
    if (allow_eq) {
        m0 = __builtin_bit_cast(S, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t.mass0())));
        const S am = m0 < 0 ? -m0 : m0;
        eqI = wave_all(t.masses_are(m0)) && am >= (S)1e-30 && am <= (S)1e30;
        inv_m0 = eqI ? (S)1 / m0 : (S)0;
    }
    const int rot = ((lane + 16) & 63) << 2;
    const int jbase = J * B + lane;
    auto fetch = [&](int c) {
        const int j = jbase + c * 64;
        return j < n ? x[j] : zero4<V4>();
    };
    if (!diag) {
#pragma unroll
        for (int r = 0; r < BPL; ++r) sh[r * (64 * W) + tid] = zero4<V4>();
        __syncthreads();
    }
    int c = w * BPL;
    V4 nxt = fetch(c);
    for (int q = 0; q < NCH; ++q) {
        V4 bj = nxt;
        const int cn = (c + 1 == NCH) ? 0 : c + 1;
        if (q + 1 < NCH) nxt = fetch(cn);
        const bool eqJ = eqI && wave_all(same_bits(bj.w, m0));
        if (eqI && !eqJ) bj.w *= inv_m0;                    // this chunk's masses in units of m0, like the sums they go into
        if (diag) {
            V4 aj = zero4<V4>();
            sym_chunk<false>(t, bj, aj, rot, eqJ);
        } else {
            V4 aj = eqJ ? zero4<V4>() : sh[c * 64 + lane];
            sym_chunk<true>(t, bj, aj, rot, eqJ);
            if (eqJ) {
                V4 o = sh[c * 64 + lane];
                o.x += aj.x * m0; o.y += aj.y * m0; o.z += aj.z * m0;
                aj = o;
            }
            sh[c * 64 + lane] = aj;
            __syncthreads();
        }
        c = cn;
    }
    if (!diag) {
        V4* const out_j = p.slabs_j + (size_t)I * p.stride_j;
        for (int e = tid; e < B; e += 64 * W) {
            const int j = J * B + e;
            if (j < n) { V4 a = sh[e]; a.w = 0; out_j[j] = a; }
        }
    }
    if (eqI) t.scale(m0);
    V4* const out_i = p.slabs_i + (size_t)J * p.stride_i;
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        const int i = ibase + k * 64;
        if (i < n) out_i[i] = t.acc(k);
    }
}

// The same kernel compiled for EXACTLY WPS waves per SIMD (amdgpu_waves_per_eu). Measured alternative, tools/symbench.hip only: it does
// not change the register allocation (200 VGPRs either way) and is within noise of the plain build (profiles/r03_symbench_rows_262144.txt).
template <class M, int W, int WPS>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
force_sym_wps(const SymParamsT<typename M::V4, typename M::S> p)
{
    force_sym_body<M, W, kSymGeneral>(p);
}

// The same kernel compiled for exactly WPS waves per SIMD (the register allocator then fits that occupancy). Measured alternative,
// tools/balbench.hip only: 8 bodies per lane at three waves per SIMD (168 VGPRs, 8 spilled dwords) is 1-5 % SLOWER than two waves
// at 12288 ... 32768 bodies (more workers, more records) — profiles/r03_balbench_wps3.txt.
template <class M, int WV, int WPS>
__global__ void __launch_bounds__(64 * WV) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) force_sym_bal_wps(const BalParams p)
{
    force_sym_bal_body<M, WV>(p);
}

}  // namespace nbk

// nbody_layout.hip.h — the host-and-device parts of the decompositions: plain structs and index arithmetic that both the kernels
// (nbody_kernels.hip.h) and the host-side shape logic (nbody_plan.hip, nbody_context.hip) use. No kernel lives here, so any
// translation unit may include it.
#pragma once
#include <hip/hip_runtime.h>

namespace nbk {

constexpr int kWG = 256;  // threads per workgroup (4 wave64)

// verdict slot of nbk::mass_scan (the equal-mass path of the symmetric kernels; see nbody_kernels.hip.h)
struct MassInfo {
    unsigned int bad_gen;   // generation of the last scan that found the bodies NOT uniform
    unsigned int pad_;
    double m0;              // mass of the first body of the last scan (a float mass converts exactly)
};

// ---- unit runs (nbk::force_sym_run): row I of the unit list = the chunks from block I's first chunk to the last chunk of the system
struct RunLayout {
    int bi;    // bodies per I-block (64*BPL); 0 = not a run layout: a fixed number of slabs
    int bpl;   // chunks per I-block
    int ncht;  // chunks of 64 bodies in the system
    int L;     // units per worker
};

__host__ __device__ inline long run_prefix(long I, const RunLayout& r) { return I * r.ncht - (long)r.bpl * I * (I - 1) / 2; }

// number of slabs that hold a partial sum of body k
__host__ __device__ inline int run_slab_count(int k, const RunLayout& r)
{
    const long K = k / r.bi;
    const long first = run_prefix(K, r) / r.L, last = (run_prefix(K + 1, r) - 1) / r.L;
    return (int)(K + (last - first + 1));
}

// ---- balanced runs (nbk::force_sym_bal + nbk::bal_reduce): workers of equal step counts, per-chunk inboxes
struct BalLayout {
    int bpl;       // stationary bodies per lane; an I-block is 64*bpl bodies
    int nbi;       // I-blocks (rows)
    int ncht;      // 64-body chunks in the system
    int L;         // rotation steps per worker
    int nworkers;
    int wv;        // workers (waves) per workgroup: consecutive workers that end in the same row combine their I-side sums
    int pmax;      // records a unit's J-side sums can be spread over (workers sharing one unit)
    int smax;      // records per inbox
    int nsteps;    // 64 * units (< 2^31: checked on the host)
};

// units before row I
__host__ __device__ inline int bal_row_prefix(int I, const BalLayout& y) { return I * y.ncht - y.bpl * (I * (I - 1) / 2); }

// row of unit u: the largest I with bal_row_prefix(I) <= u
__host__ __device__ inline int bal_row_of_unit(int u, const BalLayout& y)
{
    const float q = 2.0f * y.ncht + y.bpl;
    int I = (int)((q - sqrtf(fmaxf(q * q - 8.0f * y.bpl * (float)u, 0.0f))) / (2.0f * y.bpl));
    if (I < 0) I = 0;
    if (I > y.nbi - 1) I = y.nbi - 1;
    while (I < y.nbi - 1 && bal_row_prefix(I + 1, y) <= u) ++I;
    while (I > 0 && bal_row_prefix(I, y) > u) --I;
    return I;
}

// first and last worker with steps in row K
__host__ __device__ inline void bal_row_workers(int K, const BalLayout& y, unsigned* gf, unsigned* gl)
{
    const unsigned row0 = (unsigned)bal_row_prefix(K, y), row1 = (unsigned)bal_row_prefix(K + 1, y);
    *gf = (row0 << 6) / (unsigned)y.L;
    *gl = ((row1 << 6) - 1u) / (unsigned)y.L;
}

// last row worker g has steps in, and whether that worker exists at all
__host__ __device__ inline int bal_last_row(int g, const BalLayout& y)
{
    const long e = (long)(g + 1) * y.L;
    const int last = (int)((e < y.nsteps ? e : (long)y.nsteps) - 1);
    return bal_row_of_unit(last >> 6, y);
}

// Does worker g (gf <= g <= gl, the workers of row K) write its row-K sums itself? Not when it is a follower of a combined run:
// a worker that ENDS in row K, is not the first of its workgroup and whose predecessor also ends in row K (true for every
// predecessor >= gf: its range ends where g's begins, inside row K).
__host__ __device__ inline bool bal_writes_iside(unsigned g, unsigned gf, unsigned gl, bool gl_ends_in_row, const BalLayout& y)
{
    if (g == gf || (g % (unsigned)y.wv) == 0) return true;
    if (g < gl) return false;          // gf < g < gl: ends in row K, predecessor too
    return !gl_ends_in_row;            // g == gl: a follower only when its last row is K
}

// Host side: the layout for n bodies with `bpl` stationary bodies per lane and about `workers_target` workers (resident waves) in
// workgroups of `wv`. false when the decomposition does not apply (fewer than two chunks, step count beyond 2^31).
inline bool bal_plan(int n, int bpl, int workers_target, int wv, BalLayout* out)
{
    if (n < 128 || bpl < 1 || workers_target < 1 || wv < 1) return false;
    BalLayout y{};
    y.bpl = bpl;
    y.wv = wv;
    y.ncht = (n + 63) / 64;
    y.nbi = (n + 64 * bpl - 1) / (64 * bpl);
    const long units = (long)y.nbi * y.ncht - (long)bpl * ((long)y.nbi * (y.nbi - 1) / 2);
    if (units < 1 || units * 64 >= (1L << 31) - 64) return false;
    y.nsteps = (int)(units * 64);
    long L = (y.nsteps + (long)workers_target - 1) / workers_target;
    if (L < 16) L = 16;   // at least one row phase per worker
    y.L = (int)L;
    y.nworkers = (int)((y.nsteps + L - 1) / L);
    y.pmax = L >= 64 ? 2 : (int)(63 / L) + 2;
    int smax = 1;
    for (int K = 0; K < y.nbi; ++K) {
        unsigned gf, gl;
        bal_row_workers(K, y, &gf, &gl);
        const int s = K * y.pmax + (int)(gl - gf + 1);
        if (s > smax) smax = s;
    }
    y.smax = smax;
    *out = y;
    return true;
}

// ---- the fused small-N step in place (nbk::step_fused<.., INPLACE>)
// Device words of the in-place protocol.
// kFusedFinished counts workgroups in its low half and fall-back waves in its high half (a wave's mark precedes its workgroup's
// count, so the workgroup that completes the count reads both in the one value its atomic returns: one round trip, not two).
enum FusedSync : int { kFusedReaders = 0, kFusedFinished = 1, kFusedFallbacksTotal = 2, kFusedSyncWords = 4 };

// The clock stamps around a timed force launch (nbk::clock_begin / clock_end). s_memtime is a free-running counter of SHADER cycles
// (MI355X_MICROARCH.md: "tick = shader cycle") kept PER CU: the CUs of one XCD count at the same rate from different starting values
// (measured, tools/clock_probe.hip: same-XCD CUs agree to 1e-6 over a launch, tail idleness included; different XCDs run 1.5-2 % apart),
// so a difference is only meaningful between two readings on the SAME CU. s_memrealtime is the constant 100-MHz counter, the same on
// every XCD to within a fraction of a microsecond.
//   ClockStamp  what one wave read in front of the launch (device scratch; xcc = HW_REG_XCC_ID, hw_id = HW_REG_HW_ID: se / sh / cu)
//   ClockDelta  per timed launch, per XCD: shader cycles and 100-MHz ticks between the two stamps on one CU of that XCD (host-mapped;
//               0 = no CU of that XCD was seen by both stamps)
struct ClockStamp {
    unsigned long long cycles, ticks;
    unsigned xcc, hw_id;
};
struct ClockDelta {
    unsigned long long dcycles[8], dticks[8];
};
constexpr int kClockBeginWgs = 16;   // workgroups of the stamp in front: consecutive workgroup ids go to consecutive XCDs, two CUs each
constexpr int kClockEndWgs = 256;    // workgroups of the stamp behind: lands on every CU of an otherwise idle chip, finds its partner by CU

constexpr int kTicketMaxLanes = 8;    // nbk::force_sym_ticket: accumulation lanes per body at most
constexpr int kTicketWords = 2048 * kTicketMaxLanes;   // one ticket per (block, lane): nbi::kSymMaxSlabs blocks; the word behind them is the launch's abort flag

}  // namespace nbk

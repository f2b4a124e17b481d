/*
 * nbody.h — C-ABI of the MI355X (gfx950) all-pairs N-body engine, libnbody_hip.so.
 *
 * This is the drop-in boundary for the reference's GPU step. Reference paths below are
 * relative to the reference's TestProject/ directory.
 *
 *   reference                                         replaced by
 *   ------------------------------------------------  -----------------------------------
 *   void simulate(float4*,float4*,float4*,int)         nbody_simulate()         (kernel.cuh:2,
 *                                                                               kernel.cu:628-645)
 *   kernel<<<>>> + tile_interaction + bodyInteractions nbody_accel_range() +    (kernel.cu:9-29,
 *                                                      nbody_integrate_range()   55-65, 80-130)
 *   cudaMalloc/cudaMemcpy/cudaFree/cudaMallocHost/...  nbody_malloc_device() ...  (main.cpp:250-283,
 *     used by main() and compareHostToDevice()                                  352-366;
 *                                                                               validation.cpp:61-81)
 *   DT / EPS2 / THREADS_PER_BLOCK / TILE_WIDTH_FACTOR  nbody_ctx_set_params(),    (constants.h:11-12,
 *                                                      nbody_ctx_set_kernel()      25-26)
 *
 * Conventions
 *   - plain C, no C++ or torch types; every pointer named d_* is DEVICE memory, h_* host.
 *   - bodies are the reference's AoS float4 {x, y, z, w = mass}; velocities/accelerations are
 *     float4 with w carried as 0 (main.cpp:232-241, 271-283).
 *   - every entry point returns NBODY_OK (0) or an error code; nbody_last_error() gives the
 *     message of the calling thread's last failure. (The reference throws std::runtime_error
 *     from simulate(), kernel.cu:633-641; include/nbody_compat.hpp turns the code back into that
 *     exception for C++ callers.)
 *   - nothing here falls back to a CPU path: without a usable HIP device every compute entry
 *     fails with NBODY_ERR_HIP.
 *
 * Which symbols are what (tests/test_abi.py keeps this map equal to the declarations below)
 *   THE DROP-IN BOUNDARY — all a maintainer of the reference binds to put this library behind main.cpp / validation.cpp (20):
 *     the step      nbody_simulate  nbody_last_error
 *     memory        nbody_malloc_device  nbody_free_device  nbody_malloc_host  nbody_free_host  nbody_memcpy_h2d  nbody_memcpy_d2h
 *                   nbody_device_synchronize
 *     utils.h       nbody_fill_with_random4  nbody_fill_with_zeroes4  nbody_fill_with_zeroes3  nbody_random_float  nbody_print_device_prop
 *     validation.h  nbody_verify_still_bodies  nbody_verify_equality4  nbody_verify_equality3
 *     around it     nbody_simulate_prepare  nbody_default_ctx  nbody_simulate_host_legacy (the older snapshot's host-pointer boundary)
 *   EXTENSIONS — no reference counterpart; a caller of the boundary never needs them (67):
 *     contexts and knobs      nbody_ctx_create  nbody_ctx_destroy  nbody_ctx_set_params  nbody_ctx_set_kernel  nbody_ctx_set_symmetric_shape
 *                             nbody_ctx_set_symmetric_runs  nbody_ctx_set_fused  nbody_ctx_set_fused_inplace  nbody_ctx_fused_inplace_stats
 *                             nbody_ctx_set_equal_mass  nbody_ctx_equal_mass_verdict  nbody_ctx_set_workspace_limit  nbody_ctx_set_inplace_sums
 *                             nbody_ctx_set_stream
 *                             nbody_ctx_set_graph  nbody_ctx_reserve  nbody_ctx_sync  nbody_ctx_get  nbody_device_count
 *     queued stepping, pieces nbody_step  nbody_step_f64  nbody_accel_range  nbody_accel_square_part  nbody_accel_wrapped  nbody_accel_cross
 *                             nbody_integrate_range
 *     measuring the machine   nbody_ctx_autotune  nbody_ctx_autotuned  nbody_ctx_set_autotuned  nbody_autotune_decide  nbody_ctx_timing
 *                             nbody_ctx_timing_read  nbody_ctx_clock_read
 *     multi-GPU               nbody_shard_plan  nbody_shard_create  nbody_shard_destroy  nbody_shard_get_plan  nbody_shard_buffers
 *                             nbody_shard_upload  nbody_shard_upload_velocity  nbody_shard_download  nbody_shard_step  nbody_shard_step_phase
 *                             nbody_shard_sync  nbody_shard_comm_timing  nbody_shard_set_comm_priority  nbody_shard_comm_report
 *                             nbody_shard_comm_report_ex
 *                             nbody_comm_rccl_unique_id  nbody_comm_rccl_create  nbody_comm_rccl_destroy  nbody_comm_local_group_create
 *                             nbody_comm_local_group_destroy  nbody_comm_local_create  nbody_comm_local_destroy  nbody_comm_local_abort
 *     what would be launched  nbody_version  nbody_plan  nbody_plan_symmetric  nbody_plan_fused  nbody_plan_symmetric_occupancy
 *                             nbody_plan_ticket_task
 *                             nbody_ctx_launch_info  nbody_ctx_step_info  nbody_ctx_step_info_f64  nbody_ctx_square_info
 *     seeded initial data     nbody_fill_seeded
 */
#ifndef NBODY_H
#define NBODY_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbody_float4 {
    float x, y, z, w;
} nbody_float4;

typedef struct nbody_float3 {
    float x, y, z;
} nbody_float3;

typedef struct nbody_double4 {
    double x, y, z, w;
} nbody_double4;

typedef struct nbody_ctx nbody_ctx;

enum {
    NBODY_OK = 0,
    NBODY_ERR_INVALID = 1, /* bad argument (null pointer, n < 0, range outside [0,n) ...) */
    NBODY_ERR_HIP = 2,     /* a HIP runtime call failed / no device */
    NBODY_ERR_CONFIG = 3,  /* unsupported kernel configuration */
    NBODY_ERR_NOMEM = 4
};

/* Arithmetic flavour of the force kernel. */
enum {
    /* The d*d*d overflow corner (validation.cpp:16 == kernel.cu:20): the reference evaluates 1.0f/sqrtf(d*d*d), which  */
    /* overflows for r > ~2.6e6 and makes such a pair contribute EXACTLY 0. Only NBODY_KERNEL_STRICT reproduces that.    */
    /* FAST / ONESIDED / SYMMETRIC evaluate rsq(d)^3, which does not overflow there: they keep the physically correct    */
    /* term m/r^2 the reference drops (<= 1e9/(2.6e6)^2 = 1.5e-4 with the reference's masses, ~1e-7 of a typical max|a|); */
    /* results stay finite for any finite input (tests/test_gpu_parity.py::test_fast_kernels_at_the_overflow_corner).    */
    NBODY_KERNEL_FAST = 0,      /* packed fp32, v_rsq_f32, fma; tolerance-level parity. The library picks   */
                                /* the decomposition by size: whole steps up to 8192 bodies run the FUSED     */
                                /* one-launch step (one-sided arithmetic), from there — and square blocks of  */
                                /* nbody_accel_range from 6144 bodies — the SYMMETRIC kernel (balanced runs,  */
                                /* unit runs, block pairs); anything else the ONESIDED one                   */
    NBODY_KERNEL_STRICT = 1,    /* the reference's operation order with IEEE sqrt/div, no fma contraction,  */
                                /* j==i skipped: bit-identical to validation.cpp's arithmetic taken in      */
                                /* Jacobi order                                                             */
    NBODY_KERNEL_ONESIDED = 2,  /* FAST arithmetic, always the one-sided LDS-tiled kernel: every target     */
                                /* evaluates all N sources (N*N pair evaluations, as kernel.cu:55-65 does)  */
    NBODY_KERNEL_SYMMETRIC = 3  /* FAST arithmetic, every unordered pair evaluated once and applied to both */
                                /* bodies (N*N/2 evaluations for the same N*N interactions); square problems */
                                /* of at least two blocks, otherwise falls back to ONESIDED                 */
};

/* constants.h:25-26 */
#define NBODY_DEFAULT_EPS2 0.002f
#define NBODY_DEFAULT_DT 0.1f

/* ---- the reference boundary ------------------------------------------------------------ */

/* Drop-in for `void simulate(float4* d_bodies, float4* d_accelerations, float4* d_velocity,
 * int N)` (kernel.cuh:2; kernel.cu:628-645): one step on the arrays of the caller's CURRENT device, in place
 * (the reference never selects a device either; the caller's current device is left unchanged),
 * SYNCHRONOUS (returns after the device has finished, like the reference's
 * cudaDeviceSynchronize at kernel.cu:644): the three arrays are complete in memory when the call returns — tested for plain hipMalloc
 * arrays read next by a copy or a kernel on any stream, and for host-mapped (hipHostMalloc) and managed arrays read by the CPU; any
 * other kind of memory gets a full stream synchronisation. d_accelerations is pure output. Uses DT / EPS2 of
 * constants.h:25-26 and the FAST kernel unless the process-wide default context was
 * reconfigured through nbody_default_ctx(). N need not be a multiple of anything. */
int nbody_simulate(nbody_float4* d_bodies, nbody_float4* d_accelerations, nbody_float4* d_velocity,
                   int n);
/* Optional: everything nbody_simulate() would otherwise do inside its FIRST call for n bodies — workspaces, loading the device code
 * and, only when NBODY_AUTOTUNE=1 is set, near a switch-over size, the measurement of the decompositions (nbody_ctx_autotuned below)
 * on scratch copies of d_bodies, which are only read. A caller that times its step loop (nbody_headless does) calls this before the loop; the reference has no
 * counterpart (its first cudaLaunch pays the same kind of one-off cost inside the loop, main.cpp:146-156). */
int nbody_simulate_prepare(const nbody_float4* d_bodies, int n);

/* The OLDER boundary of the reference's snapshot (Sim-Without-OpenGL-Integration/kernel.cuh:5,
 * kernel.cu:85-125): `void simulate(float4* bodies, float3* accelerations, float3* velocity, int N)`
 * with HOST pointers — every call copies the three arrays in, runs one step and copies bodies and
 * velocity back; accelerations are not returned (the old kernel zeroes them, kernel.cu:78-80).
 * DT = 0.01 and EPS2 = 0.002 are that snapshot's double literals (its constants.h:14-15), so the
 * integrate is evaluated in double and rounded once per update. An adapter, not a fast path:
 * it crosses PCIe five times per step by construction. */
int nbody_simulate_host_legacy(nbody_float4* h_bodies, nbody_float3* h_accelerations,
                               nbody_float3* h_velocity, int n);

/* The context nbody_simulate() uses for the calling thread's current device (one per device, created on first use). */
int nbody_default_ctx(nbody_ctx** out);

/* ---- contexts ---------------------------------------------------------------------------- */

/* A context owns a device id, a stream and the slab workspace. `device` < 0 = current device. A context (and a shard built
 * on it) is used by one host thread at a time; different contexts are independent (nbody_last_error() is per thread). */
int nbody_ctx_create(nbody_ctx** out, int device);
int nbody_ctx_destroy(nbody_ctx* ctx);

/* dt / eps2 (defaults: constants.h:25-26). eps2 must be > 0: like the reference's GPU kernel
 * (kernel.cu:61-63) the FAST kernel evaluates the j == i pair, which is only an exact zero
 * when the softening keeps 1/sqrt(d^3) finite. */
int nbody_ctx_set_params(nbody_ctx* ctx, float dt, float eps2);

/* kernel: NBODY_KERNEL_*; tile: LDS tile in bodies (0 = default 2048 at 4 bodies per lane, else 1024; the reference's
 * THREADS_PER_BLOCK*TILE_WIDTH_FACTOR, constants.h:11-12, is 32); bodies_per_lane: register
 * blocking (0 = default 4); jsplit: source-range slabs per launch (0 = auto from N). */
int nbody_ctx_set_kernel(nbody_ctx* ctx, int kernel, int tile, int bodies_per_lane, int jsplit);

/* Shape of the symmetric kernel: wave64s per workgroup and stationary bodies per lane; a block is
 * 64*waves*bodies_per_lane bodies. Built: (4,10) (4,8) (2,10) (2,8) (1,10) (1,8) (2,4) (1,4) (1,2); 0 = auto (the
 * cheapest by the library's cost estimate, see nbody_plan_symmetric). The fp64 step takes (4,6) (4,8) (2,4) (1,2). */
int nbody_ctx_set_symmetric_shape(nbody_ctx* ctx, int waves, int bodies_per_lane);

/* The symmetric kernel has two more decompositions for small and mid-size systems, where whole block pairs are too coarse a
 * unit for 1024 SIMDs. UNIT RUNS (about 45k ... 160k bodies): independent waves take runs of (I-block, one 64-body chunk)
 * units. BALANCED RUNS (about 6k ... 45k bodies, the reference's N_BODIES = 8192 among them): the same unit list cut at the
 * granularity of one rotation step, so that every resident wave gets the same number of steps; partial sums go to per-chunk
 * inboxes that one kernel sums and integrates. mode -1 (default): each is used where measurements prefer it (never when a waves
 * count was set through nbody_ctx_set_symmetric_shape); 0: never; 1: unit runs always (bodies_per_lane 8 or 10); 2: balanced
 * runs always (bodies_per_lane 2, 4, 8 or 10; 0 = by size). Same pair arithmetic, same run-to-run reproducibility. */
int nbody_ctx_set_symmetric_runs(nbody_ctx* ctx, int mode);

/* Small systems (FAST kernel, up to 8192 bodies; mode -1, the default): nbody_step() runs ONE launch per step — a wave owns a few
 * target bodies, its 64 lanes split the sources, the sum never leaves the wave and the integrate happens in the same kernel; no
 * partial sums in memory, no second launch. Positions alternate between the caller's array and a spare one the context keeps (the
 * result is always left in the caller's array: after an odd number of steps one device copy puts it there). One-sided arithmetic,
 * same tolerances as the other fast kernels, bitwise reproducible run to run. mode 0: never (the LDS-tiled one-sided kernel +
 * integrate, as nbody_accel_range + nbody_integrate_range compute it: bit-identical to that pair); mode 1: whenever the kernel is FAST,
 * at any size. N = 8192, the reference's N_BODIES: 20.5 us per step against 25.7; N = 2048: 4.2 against 11.4. */
int nbody_ctx_set_fused(nbody_ctx* ctx, int mode);
/* The fused step IN PLACE: one launch reads and writes the caller's position array (a non-blocking protocol between the launch's
 * workgroups decides per wave whether everything has been read; waves that cannot know write to the context's spare array and the
 * last workgroup moves their positions home — same bits either way). mode -1 (default): nbody_simulate() — one synchronous step
 * per call, the reference's loop (main.cpp:146-156) — runs its step in place and waits on a host-mapped word the launch writes
 * when all its results are in memory, instead of a copy-back launch and a stream synchronisation (N = 8192: 30 us per call
 * against 33.5); queued steps keep the two-array kernel, which is faster when nobody waits (20.3 against 23.6 us per step).
 * Above 8192 bodies (other kernels) nbody_simulate() queues one tiny launch that writes the same word behind the step and waits
 * on it likewise (about 4 us per call less than a stream synchronisation).
 * 0: never (two arrays + copy-back and a plain stream synchronisation, the round-3 behaviour); 1: every fused step; 2: as 1 with
 * every wave forced down the fall-back path (tests). */
int nbody_ctx_set_fused_inplace(nbody_ctx* ctx, int mode);
/* Waves that took the fall-back path of the in-place step since the context was created (synchronises the stream). */
int nbody_ctx_fused_inplace_stats(nbody_ctx* ctx, unsigned long long* out_fallback_waves);

/* EQUAL-MASS systems (a Plummer model, most cluster and cosmological initial conditions: m_i = M / N). When every body a launch
 * touches has bit for bit the same mass m0, a_i = m0 * sum_j w_ij r_ij: the symmetric kernels then accumulate sum w r on both
 * sides of a pair and apply m0 once per stored partial sum — 14 instead of 16 packed operations per two pair evaluations and no
 * mass to rotate (N = 262144: 9.9 ms per launch against 10.9, 88 % against 80 % of the fp32 vector peak on the 20-FLOP convention).
 * The decision is made ON THE DEVICE, per launch, in stream order: a small scan kernel compares every body of the launch's ranges
 * with the first one (and checks that all coordinates are finite and within 1e15 of the origin: padding lanes sit at 1e18, where
 * their term underflows to exactly 0) and the force kernel reads its verdict — no host round trip, nothing to declare. Bodies
 * that are not uniform (the reference's own initial conditions: masses uniform in [MIN_W, MAX_W] = [1e5, 1e9], constants.h:18-19) take the general path, bit for bit
 * as before. Same tolerances on either path; the two differ by rounding (m0 * sum(w r) against sum((m0 w) r)), each is bitwise
 * reproducible run to run. The scan is a dependent launch of its own, about 3 us per nbody_step call (once per call: the integrate
 * carries the masses through unchanged) or accel launch; never under graph replay. mode -1 (default): launches of 32768 bodies or
 * more of the symmetric kernels (block pairs, unit runs, balanced runs, nbody_accel_cross; for the latter both ranges count) — a
 * caller that steps ONE step per call, as the reference's loop does, would otherwise pay the scan every step: 4-8 % of a step at
 * 9216 ... 16384 bodies, whatever the masses are; mode 1: launches of 4096 bodies or more (for callers who queue many steps per
 * call, or know their masses are equal: +4 ... 6 % from 8193 to 32767 bodies); 0: never. */
int nbody_ctx_set_equal_mass(nbody_ctx* ctx, int mode);

/* What the last scan found (synchronises the context's stream): *scanned 0 = no scan has run yet; *uniform 1 = the bodies of the
 * last scanned launch had one common mass (*mass), so that launch took the equal-mass path. Any out pointer may be NULL. */
int nbody_ctx_equal_mass_verdict(nbody_ctx* ctx, int* scanned, int* uniform, float* mass);

/* The symmetric kernels keep one slab of partial sums per block of bodies (nb x n x 16 B: 412 MiB at N = 262144, 6.4 GiB at
 * N = 1048576, growing as N^2/B). The launch-shape choice only uses a slab decomposition whose workspace fits a cap:
 * min(96 GiB, half of the device memory that is free, `bytes` if non-zero); beyond it — or when the allocation itself fails —
 * a whole step, and nbody_accel_range on a square block, keep the symmetric arithmetic and add the block sums IN PLACE instead
 * (nbody_ctx_set_inplace_sums below: a few accumulation lanes, or no workspace at all); only a cap below two lanes of
 * nbody_accel_range ends in the one-sided kernel (<= 64 slabs, about 30 % slower at large N); nbody_accel_cross cuts its source
 * run into pieces. Never an error.
 * bytes = 0: automatic. fail_above != 0 is a TEST hook: the shape choice ignores `bytes`, and every workspace allocation larger
 * than `bytes` fails as if the device were out of memory (exercises the fallback path without exhausting a 288 GB device). */
int nbody_ctx_set_workspace_limit(nbody_ctx* ctx, size_t bytes, int fail_above);

/* Block pairs with the partial sums added IN PLACE (nbk::force_sym_ticket): the same pair arithmetic and block shapes as the slab
 * kernel, but every task ADDS its two block sums into one of a few accumulation lanes per body (8 x 16 n bytes where the cap allows:
 * a few per cent of the slab footprint; fewer under a tight cap; with one lane the sums go straight into d_accelerations and there is
 * no workspace at all), in an order fixed per block and lane by a ticket (the same bits on every run), through agent-scope accesses;
 * the integrate adds the lanes in index order. The reference's
 * analogue is its shared-memory guard that rejects large N outright (kernel.cu:639-641); here N has no workspace cliff.
 * mode -1 (default): whole steps whose slab workspace does not fit the cap above; 1: every whole step FAST / SYMMETRIC would run as
 * unit runs or block pairs (from 12288 bodies; balanced runs and the fused small-N step keep their own sizes); 0: never (the older
 * fallback: one-sided kernel). The sums differ from the slab kernel's by rounding only (another, equally fixed order of the same
 * nb block sums per body). A workgroup that waited more than 10 s for its turn (never on a healthy run) aborts the launch's remaining waits
 * and raises an error that the next nbody_step / nbody_ctx_sync returns (the accelerations of that step are incomplete; the tickets are
 * reset, stepping can go on from restored state). mode 2 is a TEST hook: as 1, and the NEXT in-place launch finds one ticket held by nobody
 * and gives up after 2 ms — the abort path exercised without a defect to provoke it. */
int nbody_ctx_set_inplace_sums(nbody_ctx* ctx, int mode);

/* Launch on this HIP stream (a hipStream_t passed as void*; NULL = the context's own stream). */
int nbody_ctx_set_stream(nbody_ctx* ctx, void* hip_stream);

/* nbody_step() can replay a captured hipGraph of 32 (force, integrate) pairs instead of 64 host
 * launches. mode: 0 never (default), -1 auto (N <= 16384 and steps >= 32), 1 whenever steps >= 32.
 * Results are identical either way (same kernels, same order). Measured on MI355X it is neutral:
 * queued eager launches are not host-bound (17.4 vs 17.1 us/step at N=1024, 28.1 vs 27.2 at
 * N=8192); the floor at small N is the dependent-kernel boundary, not the host. */
int nbody_ctx_set_graph(nbody_ctx* ctx, int mode);

/* Which decomposition runs a whole step of n bodies fastest depends on the chip and on the compiler; the switch-over sizes built
 * into the library (fused step up to 8192 bodies, balanced runs to 45056, unit runs to 160000, block pairs above) were measured on
 * MI355X with ROCm 7.2. nbody_ctx_autotune measures instead: it times `steps_per_trial` steps (three repeats, best taken) of every
 * decomposition that applies to n bodies — on scratch copies of d_bodies with dt = 0, the caller's arrays are only read — and leaves
 * the context's knobs (nbody_ctx_set_fused, nbody_ctx_set_symmetric_runs, bodies per lane) on the fastest. FAST kernel only.
 * *out_choice: 1 fused step, 24 / 28 / 210 balanced runs with 4 / 8 / 10 bodies per lane, 3 unit runs, 4 what the library picks
 * without any run-based variant (block pairs, or the two-kernel one-sided path at small sizes). Synchronous; costs a few hundred
 * steps of the system. The knobs apply to every size the context is used with afterwards. */
int nbody_ctx_autotune(nbody_ctx* ctx, const nbody_float4* d_bodies, int n, int steps_per_trial, int* out_choice,
                       double* out_us_per_step);
/* nbody_simulate() — the default context — measures NOTHING by default: a caller that never asked for tuning (the reference's loop,
 * main.cpp:146-156) gets the built-in decomposition, the same low-order bits on every machine and a first call without trial steps.
 * With NBODY_AUTOTUNE=1 in the environment the first call with n within a quarter of a built-in switch-over size (8192, 45056,
 * 160000) measures the decompositions once on scratch copies of the caller's bodies (trials of about 10 ms each) and keeps the
 * built-in choice unless another one wins clearly and repeatably (nbody_autotune_decide); later calls with that n use what was found
 * (results are bit-identical to forcing that decomposition through the knobs). Setting any shape knob of the default context
 * switches the measurement off. nbody_ctx_autotuned reports what was found for n: choice as above, 0 = built-in kept, -1 = not
 * measured; the two timings in microseconds per step. */
int nbody_ctx_autotuned(nbody_ctx* ctx, int n, int* out_choice, double* out_us_builtin, double* out_us_best);
/* Pins what nbody_simulate() uses for n bodies instead of measuring (a caller that wants the same decomposition, hence the same
 * low-order bits, on every machine): choice 0 = built-in, an id of nbody_ctx_autotune = that decomposition, -1 = forget n. */
int nbody_ctx_set_autotuned(nbody_ctx* ctx, int n, int choice);
/* The decision rule itself (pure host logic, no device): may a measured challenger override the built-in decomposition? 1 = yes.
 * The built-in choice timed first and last must agree with itself within 10 % (a quiet machine), the challenger must beat it by
 * more than `margin` (0.03), and — when n_confirm > 0 — every confirmation trial of the challenger must beat every confirmation
 * trial of the built-in choice by that margin. */
int nbody_autotune_decide(double builtin_first_us, double builtin_last_us, double challenger_us, const double* confirm_builtin_us,
                          const double* confirm_challenger_us, int n_confirm, double margin);

/* Pre-size the slab workspace for up to n_targets bodies so later calls never allocate. */
int nbody_ctx_reserve(nbody_ctx* ctx, int n_targets);

/* ---- stepping ---------------------------------------------------------------------------- */

/* `steps` whole steps on [0,n): forces from the positions at the start of each step, then
 * v += (0.5f*dt)*a ; x += dt*v (validation.cpp:40-49 == kernel.cu:116-129). ASYNCHRONOUS on the
 * context's stream: no host synchronisation, the caller syncs (nbody_ctx_sync / its stream). */
int nbody_step(nbody_ctx* ctx, nbody_float4* d_bodies, nbody_float4* d_accelerations,
               nbody_float4* d_velocity, int n, int steps);

/* Accelerations of targets [i0,i1) from sources [j0,j1) of d_bodies (absolute indices into one
 * array of at least max(i1,j1) bodies). d_acc_out has i1-i0 entries. accumulate != 0 continues
 * the sums already in d_acc_out (STRICT: continues the sequential sum exactly). Asynchronous.
 * This is what a rank of the sharded multi-GPU step calls for its local and remote blocks. */
int nbody_accel_range(nbody_ctx* ctx, const nbody_float4* d_bodies, nbody_float4* d_acc_out, int i0,
                      int i1, int j0, int j1, int accumulate);

/* nbody_accel_range(i0, i1, i0, i1) — a block against itself — issued in `nparts` launches, so that a caller can put
 * other work of the same stream between them (the sharded step hides its exchange behind the second half). Call with
 * part = 0 .. nparts-1 in order on one stream; the LAST part adds the partial sums up and writes d_acc_out (with
 * accumulate != 0: continues the sums already there). Other launches of this context may run in between, also
 * nbody_accel_cross (it has a workspace of its own); another nbody_accel_range / nbody_step may not. Where the symmetric
 * kernel does not apply (kernel id, size) part 0 does the whole evaluation and the others nothing. */
int nbody_accel_square_part(nbody_ctx* ctx, const nbody_float4* d_bodies, nbody_float4* d_acc_out, int i0, int i1,
                            int accumulate, int part, int nparts);

/* The same with a source run that may wrap around the end of the array: sources j0, j0+1, ...,
 * j0+count-1, indices taken modulo n_total. One launch covers "every block except my own" for a rank
 * of the sharded step (j0 = end of the own block, count = n_total - block). */
int nbody_accel_wrapped(nbody_ctx* ctx, const nbody_float4* d_bodies, int n_total,
                        nbody_float4* d_acc_out, int i0, int i1, int j0, int count, int accumulate);

/* SYMMETRIC evaluation of two DISJOINT sets of bodies of one array of n_total: targets [i0,i1) against the
 * source run j0, j0+1, ..., j0+count-1 (indices modulo n_total; the run must not touch [i0,i1)). Every pair
 * (i, j) is evaluated once and applied to both bodies (Newton's third law): d_acc_i[i1-i0] receives (or, with
 * accumulate_i != 0, continues) the accelerations the sources exert on the targets, d_acc_j_out[count] is
 * OVERWRITTEN with the accelerations the targets exert on the sources, in run order. FAST arithmetic only.
 * This is what a rank of the sharded step calls for the blocks of other ranks it is responsible for: the
 * second output is what it sends back to their owners. Asynchronous. */
int nbody_accel_cross(nbody_ctx* ctx, const nbody_float4* d_bodies, int n_total, nbody_float4* d_acc_i, int i0,
                      int i1, int accumulate_i, int j0, int count, nbody_float4* d_acc_j_out);

/* Integrate bodies [i0,i1): d_bodies is the whole array (indexed absolutely), d_velocity and
 * d_acc hold the i1-i0 own entries. Asynchronous. */
int nbody_integrate_range(nbody_ctx* ctx, nbody_float4* d_bodies, nbody_float4* d_velocity,
                          const nbody_float4* d_acc, int i0, int i1);

int nbody_ctx_sync(nbody_ctx* ctx);

/* The context's device, kernel id (NBODY_KERNEL_*) and launch stream (hipStream_t as void*). Any out pointer may be NULL. */
int nbody_ctx_get(nbody_ctx* ctx, int* device, int* kernel, void** hip_stream);

/* ---- sharded step: one rank per GPU (no reference analogue: the reference drives device 0 only, ------------
 *      kernel.cu:630, main.cpp:287; this lifts that) ---------------------------------------------------------
 *
 * The bodies are cut into `world` contiguous blocks of `shard` bodies (n_total padded with massless bodies to
 * world*shard, shard even). Rank r owns block r: its velocities and accelerations never leave the rank; every rank
 * holds all positions. Per step, on rank r:
 *
 *   schedule SYMMETRIC (ctx kernel FAST or SYMMETRIC) — every unordered pair of bodies evaluated once in the machine:
 *     comm stream     all-gather of positions (in place, shard*16 B per rank)        | overlapped with
 *     compute stream  own block x own block, symmetric kernel, FIRST half of its tasks | each other
 *     compute stream  own block x the blocks r+1 .. r+(world-1)/2 (for an even world the block half-way round is
 *                     shared between its two ranks): nbody_accel_cross — the J-side sums belong to OTHER ranks
 *     comm stream     exchange: those J-side sums go to their owners, this rank's arrive   | overlapped with
 *     compute stream  own block x own block, SECOND half of its tasks, slab sum            | each other
 *     compute stream  add the received sums in a fixed order, integrate the own block
 *   schedule ONESIDED (ctx kernel ONESIDED): all-gather overlapped with own x own, then own x everybody else in ONE
 *     one-sided launch over a source run that wraps around the end of the array; no exchange.
 *   schedule CANONICAL (ctx kernel STRICT): all-gather, then own x [0, n) in index order — bit-identical to the
 *     single-device strict step, no overlap (the parity path).
 *
 * Communication goes through two callbacks (nbody_comm), so the same executor runs over RCCL from C
 * (nbody_comm_rccl_*), over torch.distributed from Python (n-bodysimulation_amd/sharded.py), or over plain device
 * copies when one process drives several "ranks" on one GPU in a test. */

enum { NBODY_SCHEDULE_CANONICAL = 0, NBODY_SCHEDULE_ONESIDED = 1, NBODY_SCHEDULE_SYMMETRIC = 2 };
#define NBODY_MAX_RANKS 64

typedef struct nbody_cross_launch {
    int i0, i1;     /* targets: a sub-range of the own block (absolute indices) */
    int j0, count;  /* source run j0, j0+1, ... (mod n_pad), `count` bodies of other ranks */
    int jbuf_offset; /* where this launch's J-side sums start in the rank's J buffer (bodies) */
} nbody_cross_launch;

typedef struct nbody_shard_segment {
    int peer;    /* the other rank */
    int offset;  /* first body of the segment in this rank's J buffer (sends) / receive buffer (recvs) */
    int count;   /* bodies */
    int body0;   /* absolute index of the body the first entry belongs to */
} nbody_shard_segment;

typedef struct nbody_shard_plan_t {
    int rank, world, n_total, schedule;
    int shard, n_pad, i0, i1;       /* own block [i0,i1) = [rank*shard, (rank+1)*shard) */
    int n_launches;                 /* symmetric schedule: 0..2 nbody_accel_cross launches */
    nbody_cross_launch launch[2];
    int jbuf_bodies;                /* J buffer: the launches' J-side sums back to back */
    int n_sends, n_recvs;
    nbody_shard_segment send[NBODY_MAX_RANKS];
    nbody_shard_segment recv[NBODY_MAX_RANKS]; /* in the order the received sums are added */
    int rbuf_bodies;                /* receive buffer */
} nbody_shard_plan_t;

/* Pure host logic (no device, no context): what rank `rank` of `world` does for n_total bodies under `schedule`. */
int nbody_shard_plan(int rank, int world, int n_total, int schedule, nbody_shard_plan_t* out);

/* The two collective operations of the step. Both ENQUEUE on hip_stream (a hipStream_t) and return 0, or non-zero
 * on failure; neither may block the host on the GPU. */
typedef struct nbody_comm {
    void* user;
    /* in-place all-gather over the ranks: rank r contributes the `bodies_per_rank` bodies at d_x_full + r*bodies_per_rank */
    int (*all_gather)(void* user, nbody_float4* d_x_full, int bodies_per_rank, void* hip_stream);
    /* one grouped exchange: send[k] = `count` bodies at d_jbuf + offset to `peer`; recv[k] = `count` bodies from `peer`
     * into d_rbuf + offset. Every rank calls it once per step, also with nothing to send or receive. */
    int (*exchange)(void* user, const nbody_shard_segment* send, int n_sends, const nbody_float4* d_jbuf,
                    const nbody_shard_segment* recv, int n_recvs, nbody_float4* d_rbuf, void* hip_stream);
} nbody_comm;

/* RCCL over xGMI from C, without a link-time dependency: librccl.so is loaded on first use. Rank 0 makes a unique
 * id (128 bytes), the caller distributes it to every rank by its own means, every rank creates its communicator. */
int nbody_comm_rccl_unique_id(void* out_128_bytes);
/* The communicator is created on `device` (< 0: the calling thread's current device). */
int nbody_comm_rccl_create(nbody_comm* out, int rank, int world, const void* unique_id_128_bytes, int device);
int nbody_comm_rccl_destroy(nbody_comm* comm);

/* The two collectives WITHOUT RCCL, for ranks that are threads of ONE process (nbody_headless --ngpu G --transport local): every rank
 * pulls its peers' blocks with hipMemcpyPeerAsync — straight over xGMI where peer access exists — ordered by events the owning rank
 * records and a host rendezvous of the rank threads per collective. One group per job, one communicator per rank; the ranks may
 * share a device (rehearsals on a one-GPU box). A rank that fails calls nbody_comm_local_abort so that its peers' collectives
 * return an error instead of waiting; a rendezvous that is not completed within deadline_seconds (<= 0: 600) does the same. */
typedef struct nbody_local_group nbody_local_group;
int nbody_comm_local_group_create(nbody_local_group** out, int world, double deadline_seconds);
int nbody_comm_local_group_destroy(nbody_local_group* group);   /* after every rank's nbody_comm_local_destroy */
int nbody_comm_local_create(nbody_comm* out, nbody_local_group* group, int rank, int device);
int nbody_comm_local_destroy(nbody_comm* comm);
int nbody_comm_local_abort(nbody_local_group* group);

typedef struct nbody_shard nbody_shard;

/* A rank of the sharded step. Owns the rank's device state on ctx's device: all positions (n_pad bodies), the own
 * block's velocities and accelerations, the J and receive buffers, a communication stream and the events between
 * it and ctx's launch stream. `comm` may be NULL when world == 1. The schedule follows ctx's kernel at creation. */
int nbody_shard_create(nbody_shard** out, nbody_ctx* ctx, int rank, int world, int n_total, const nbody_comm* comm);
int nbody_shard_destroy(nbody_shard* shard);
int nbody_shard_get_plan(nbody_shard* shard, nbody_shard_plan_t* out);
/* Device pointers of the rank's arrays (x: n_pad bodies, v/a: shard bodies, jbuf/rbuf as in the plan) for callers that
 * wrap them (torch) or fill them on the device. Any out pointer may be NULL. */
int nbody_shard_buffers(nbody_shard* shard, nbody_float4** d_x_full, nbody_float4** d_v_own, nbody_float4** d_a_own,
                        nbody_float4** d_jbuf, nbody_float4** d_rbuf);
/* Every rank passes the same n_total bodies (host memory): positions of all, zero velocity/acceleration; padding
 * bodies are massless and sit on body 0. Synchronous. */
int nbody_shard_upload(nbody_shard* shard, const nbody_float4* h_bodies);
/* Velocities of all n_total bodies (host memory; every rank passes the same array and keeps its own block) — to
 * continue a run: nbody_shard_upload() zeroes them. Synchronous. */
int nbody_shard_upload_velocity(nbody_shard* shard, const nbody_float4* h_velocity);
/* Own block (shard entries each, padding included) to host memory. Synchronous. */
int nbody_shard_download(nbody_shard* shard, nbody_float4* h_x_own, nbody_float4* h_v_own, nbody_float4* h_a_own);
/* `steps` whole steps, asynchronous (nbody_shard_sync waits for both streams). */
int nbody_shard_step(nbody_shard* shard, int steps);
/* One step in four parts, for a driver that holds several ranks in one thread (tests): 0 = start the all-gather,
 * 1 = force launches, 2 = start the exchange, 3 = add received sums + integrate. nbody_shard_step = 0,1,2,3 per step. */
int nbody_shard_step_phase(nbody_shard* shard, int phase);
int nbody_shard_sync(nbody_shard* shard);
/* Per-step communication timing (events on the two streams; a ring of 64 step records whose events are reused, each folded into
 * running sums before its slot comes up again — memory does not grow with the number of timed steps). Enabling (or disabling)
 * starts the statistics afresh. */
int nbody_shard_comm_timing(nbody_shard* shard, int enable);
/* Priority of the rank's communication stream: 0 = normal (the default), 1 = the greatest the device offers (RCCL's few
 * channel workgroups are then placed ahead of the queued force workgroups). Measured: with three or more processes
 * sharing ONE GPU, 1 is pathological (hundreds of ms per step: spinning high-priority RCCL kernels of different
 * processes); with a GPU per rank it is a knob to try (bench.py --comm native --comm-priority auto measures both).
 * The stream is made on first use; a later call synchronises both streams and replaces it. Call between steps. */
int nbody_shard_set_comm_priority(nbody_shard* shard, int high);
/* What the timed steps since nbody_shard_comm_timing(.., 1) add up to. Mean AND maximum over the steps — one late all-gather in
 * twenty is invisible in a mean: the all-gather's time and the part of it not hidden behind the first half of the own-block pass
 * (from the end of that pass to the end of the gather; 0 when the gather ended first); the exchange's time and the part not hidden
 * behind the second half. nbody_shard_comm_report is the older, means-only form; any of its out pointers may be NULL. */
typedef struct nbody_comm_report_t {
    int steps;           /* timed steps begun */
    int gathers, exchanges;   /* steps that carried an all-gather / an exchange (the first step after an upload has neither) */
    int records_kept;    /* step records alive (<= 64) */
    double gather_ms, gather_exposed_ms, exchange_ms, exchange_exposed_ms;                       /* means over those steps */
    double gather_ms_max, gather_exposed_ms_max, exchange_ms_max, exchange_exposed_ms_max;       /* maxima over those steps */
} nbody_comm_report_t;
int nbody_shard_comm_report_ex(nbody_shard* shard, nbody_comm_report_t* out);
int nbody_shard_comm_report(nbody_shard* shard, int* steps, double* gather_ms, double* gather_exposed_ms, double* exchange_ms,
                            double* exchange_exposed_ms);

/* ---- fp64 variant (the build's own; the reference has no double path) -------------------- */
int nbody_step_f64(nbody_ctx* ctx, nbody_double4* d_bodies, nbody_double4* d_accelerations,
                   nbody_double4* d_velocity, int n, int steps, double dt, double eps2);

/* ---- memory helpers (what main.cpp / validation.cpp use the CUDA runtime for) ------------ */
int nbody_device_count(int* count);
int nbody_malloc_device(void** d_ptr, size_t bytes);   /* cudaMalloc      main.cpp:278-279,353 */
int nbody_free_device(void* d_ptr);                    /* cudaFree        main.cpp:358-363     */
int nbody_malloc_host(void** h_ptr, size_t bytes);     /* cudaMallocHost  main.cpp:250-252     */
int nbody_free_host(void* h_ptr);                      /* cudaFreeHost    main.cpp:364-366     */
int nbody_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes); /* main.cpp:282-283,354   */
int nbody_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes); /* validation.cpp:79-81   */
int nbody_device_synchronize(void);                    /* cudaDeviceSynchronize validation.cpp:77 */

/* ---- host-side helpers with the reference's meaning (utils.h / validation.h) ------------- */
/* utils.cpp:30-37: x,y,z in U(-1e5,1e5), w in U(1e5,1e9), 4 libc rand() draws per body in
 * x,y,z,w order. */
void nbody_fill_with_random4(nbody_float4* h_v, int n);
/* utils.cpp:19-27 */
void nbody_fill_with_zeroes4(nbody_float4* h_v, int n);
/* utils.cpp:6 — ((float)rand() / RAND_MAX) * (max - min) + min, one libc rand() draw */
float nbody_random_float(float min, float max);
/* utils.cpp:9-16 */
void nbody_fill_with_zeroes3(nbody_float3* h_v, int n);
/* utils.cpp:50-68: the reference prints cudaDeviceProp fields of device 0; this prints the same lines
 * from hipDeviceProp_t of the CURRENT device (returns NBODY_ERR_HIP when there is none). */
int nbody_print_device_prop(void);
/* Portable, seeded generators for reproducible cross-machine inputs (libc rand() is not:
 * RAND_MAX differs between libcs). init: 0 = the reference's uniform cube and mass range,
 * 1 = Plummer sphere (a = 1, total mass 1, G = 1), cold start. */
int nbody_fill_seeded(nbody_float4* h_bodies, int n, int init, unsigned long long seed);
/* validation.cpp:143-164 and 106-122, returning the number of bodies the reference would have
 * printed "Problem at body" for instead of printing them. */
int nbody_verify_still_bodies(const nbody_float4* h_v, const nbody_float4* h_x, int n);
int nbody_verify_equality4(const nbody_float4* h_v, const nbody_float4* h_x, int n);
/* validation.cpp:125-140 (x, y, z only) */
int nbody_verify_equality3(const nbody_float3* h_v, const nbody_float3* h_x, int n);

/* ---- measurement ------------------------------------------------------------------------- */
/* With timing on (enable = 1), every force-kernel launch made through this context is bracketed by a pair of
 * hipEvents on the launch stream. nbody_ctx_timing_read() synchronises the stream, returns the
 * summed force-kernel time (ms) and the number of launches since the last read, and resets the
 * counters. (bench.py's roofline figure; costs two event records per launch.)
 * enable = 2 adds the CLOCK: outside the event pair, in front of and behind the force launch, one tiny launch each reads shader-cycle
 * counters (s_memtime, kept per CU) and the constant 100-MHz counter (s_memrealtime); the launch behind pairs the readings CU by CU and
 * writes, per XCD, the shader cycles and 100-MHz ticks that elapsed into host-mapped memory — no stamp executes inside the measured
 * kernel. nbody_ctx_clock_read() turns the records since the last read into: shader cycles that elapsed per force launch (mean over
 * the XCDs; the two launch boundaries, a few microseconds, included), 100-MHz ticks per launch, and their ratio x 100 MHz = the shader
 * clock held under THAT load — overall, and for the slowest / fastest XCD (the eight XCDs of one MI355X run 1.5-2 % apart). launches =
 * timed launches with at least one XCD paired (at most 4096 between two reads; later ones go unstamped), unpaired = (launch, XCD)
 * records in which no CU was seen by both stamps (0 expected on a GPU that runs nothing else). This is what separates "a slower box"
 * from "slower code" in a bench line. */
typedef struct nbody_clock_report {
    int launches, xcds, unpaired;
    double cycles_per_launch, cycles_per_launch_min, cycles_per_launch_max;
    double ticks_per_launch;        /* 100-MHz ticks: x 1e-5 = milliseconds */
    double sclk_mhz, sclk_mhz_min_xcd, sclk_mhz_max_xcd;
} nbody_clock_report;
int nbody_ctx_timing(nbody_ctx* ctx, int enable);
int nbody_ctx_timing_read(nbody_ctx* ctx, double* force_ms, int* launches);
int nbody_ctx_clock_read(nbody_ctx* ctx, nbody_clock_report* out);

/* ---- diagnostics ------------------------------------------------------------------------- */
const char* nbody_last_error(void);
/* e.g. "nbody_hip 0.1 gfx950 fast=lds-packed(bpl4,tile2048,u8) ..." */
const char* nbody_version(void);
/* What the context resolved for a problem of n targets x m sources: slabs per launch, grid
 * blocks, LDS bytes per block. Any out pointer may be NULL. */
int nbody_ctx_launch_info(nbody_ctx* ctx, int n_targets, int n_sources, int* jsplit, int* blocks,
                          int* lds_bytes);

/* What nbody_step() does for n bodies: symmetric = 1 when the symmetric kernel runs on block pairs, 2 when it runs in
 * runs of chunk units, 3 in balanced runs of rotation steps (slabs = records per inbox), 4 on block pairs with the sums added in
 * place (slabs = accumulation lanes, 0 when the sums go straight into the acceleration array: nbody_ctx_set_inplace_sums), 0 for the one-sided kernel, -1 for the fused
 * small-N step (one-sided arithmetic, force + integrate in one launch: slabs = 0, block_bodies = targets per workgroup); block_bodies = bodies
 * per block (symmetric) or per workgroup (one-sided); slabs = partial-sum slabs the integrate adds;
 * workgroups = grid size; evaluated_pairs = pair evaluations per step (n*n one-sided; about n*n/2 plus
 * the diagonal blocks symmetric — the interactions applied are n*n either way). Any out pointer may be NULL. */
int nbody_ctx_step_info(nbody_ctx* ctx, int n, int* symmetric, int* block_bodies, int* slabs, int* workgroups,
                        double* evaluated_pairs);

/* The same for the fp64 step (nbody_step_f64): symmetric = 1 for the double-precision rotation kernel, 0 for the one-sided one. */
int nbody_ctx_step_info_f64(nbody_ctx* ctx, int n, int* symmetric, int* block_bodies, int* slabs, int* workgroups,
                            double* evaluated_pairs);

/* What nbody_accel_square_part(.., nparts) launches for a block of n bodies against itself: with one part this is
 * nbody_ctx_step_info; with several parts only the block-pair task list splits, so the block-pair kernel runs wherever the
 * symmetric kernel applies (also at sizes where a whole step would use the run-based variant). What one rank's own-block
 * pass of the sharded step launches (nparts = 2 when world > 1). */
int nbody_ctx_square_info(nbody_ctx* ctx, int n, int nparts, int* symmetric, int* block_bodies, int* slabs, int* workgroups,
                          double* evaluated_pairs);

/* The same resolution without a context or a device (pure host logic, for tests and tooling):
 * given the user's kernel/tile/bodies_per_lane/jsplit choices (0 = auto) and a CU count, what the
 * launcher would pick. blocks_x = workgroups along the targets; the grid is blocks_x * jsplit. */
int nbody_plan(int n_targets, int n_sources, int kernel, int tile, int bodies_per_lane, int jsplit,
               int num_cu, int* out_bodies_per_lane, int* out_tile, int* out_jsplit, int* out_blocks_x);

/* The symmetric kernel's block-shape choice for n bodies without a context or a device: waves / bodies_per_lane = 0 leave
 * the choice to the library (cheapest by its cost estimate: per-step instruction costs, waves per SIMD, the last partial
 * round, the slab sum), non-zero values restrict it. blocks = slabs per body, workgroups = grid size. */
int nbody_plan_symmetric(int n, int num_cu, int waves, int bodies_per_lane, int* out_waves, int* out_bodies_per_lane,
                         int* out_blocks, int* out_workgroups);

/* The fused small-N step's launch shape for n bodies on a chip of num_cu CUs, without a context or a device: targets per wave (2,
 * or 4 when two would need more than 16 waves per workgroup), waves per workgroup (even, 2 ... 16), LDS tile in bodies, workgroups
 * (one per CU up to 8192 bodies on 256 CUs). */
int nbody_plan_fused(int n, int num_cu, int* out_targets_per_wave, int* out_waves, int* out_tile, int* out_workgroups);

/* Waves per SIMD the library's cost estimates assume for the fp32 symmetric kernels with this many stationary bodies per
 * lane (2 at 10, 3 at 8, 5 at 4, 8 at 2): a property of the compiled kernels' register allocation, checked against the
 * compiler's own resource report by tests/test_build_resources.py. 0 for an unsupported count. */
int nbody_plan_symmetric_occupancy(int bodies_per_lane);

/* The in-place block-pair kernel's task list without a device (host tests): task t of a launch over nb blocks evaluates the block
 * pair (i, j) — anti-diagonal by anti-diagonal, then the nb diagonal blocks — and its sums are contribution number seq_i of block i
 * and seq_j of block j (-1 for a diagonal task, which has one sum): every block receives contributions 0 .. nb-1 exactly once, and a
 * contribution's predecessor always belongs to an earlier task. */
int nbody_plan_ticket_task(int nb, int task, int* out_i, int* out_j, int* out_seq_i, int* out_seq_j);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_H */

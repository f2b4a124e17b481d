#!/usr/bin/env python3
"""bench.py — body-pair interactions/s of the all-pairs step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bodies B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over all bodies: the O(N^2) force accumulation followed by
the half-kick/drift integrate (TestProject/kernel.cu:80-130 in the reference).

Workload (BASELINE.json): N = 262144 bodies, fp32, 1 GPU (configs[2], the size the metric is quoted
on). With G > 1 ranks the bodies are block-partitioned and positions all-gathered once per step
(RCCL); the run is WEAK-scaled: N(G) = 262144 * sqrt(G) rounded to a multiple of 8192*G (376832,
524288, 720896 bodies at G = 2, 4, 8), so every GPU evaluates 6.9e10 (+-5 %) pairs per step. `--bodies` overrides (e.g. --bodies 1048576 for configs[3]).

Rank 0 prints ONE JSON line. `value` = N^2 * steps / wall time of the timed region (max over
ranks), inputs resident in HBM before the region starts. `roofline` is the force kernel's
algorithmic FLOP rate (20 FLOP/pair) over its own HIP-event time on its launch
stream, against the 157.3 TFLOP/s fp32 vector peak; `cpu_baseline` times the reference's own CPU
path (oracle/_ref, kind "reference") or the checker's restatement (kind "port") on a bounded
sample on the host cores — a reported baseline, not the product.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 20.0            # the agreed algorithmic count
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
BASE_N = 262144


def weak_n(gpus: int) -> int:
    if gpus == 1:
        return BASE_N
    # every rank's block a multiple of 8192 bodies (8 target workgroups): shapes with an odd number of
    # target workgroups run 2-4 % slower (tools/shape_probe.py), which would be a shape artefact, not scaling
    q = 8192 * gpus
    return int(round(BASE_N * math.sqrt(gpus) / q)) * q


def cpu_baseline(seconds_budget: float = 12.0):
    """Times the reference's CPU step (serial, as the reference builds it) on a bounded sample."""
    import numpy as np
    from oracle import oracle as O   # checker / baseline only — never the measured product
    import nbody_amd
    n = 16384
    x0 = nbody_amd.engine.seeded_bodies(n, 0, 12345)
    out = {}
    if O.have_ref():
        fn, kind = O.ref_step, "reference"
    else:
        fn, kind = (lambda x, a, v, steps=1: O.step_inplace(x, a, v, steps=steps)), "port"
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    fn(x, a, v, steps=1)            # warm caches
    t0 = time.perf_counter()
    done = 0
    while done < 64:
        fn(x, a, v, steps=1)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    out.update({"value": n * (n - 1) * done / dt, "unit": "pairs/s", "cores": 1, "kind": kind,
                "sample": f"{done} serial in-place steps of validation.cpp:28-52 at N={n} (reference init, DT=0.1, EPS2=0.002), {dt:.1f} s"})
    # the best honest CPU line: our Jacobi restatement, SIMD over targets, OpenMP over all cores
    thr = O.max_threads()
    O.set_threads(thr)
    O.accel_range(x0, 0, n)          # spin up the thread pool
    t0 = time.perf_counter()
    reps = 0
    while reps < 200:
        O.accel_range(x0, 0, n)
        reps += 1
        if time.perf_counter() - t0 > seconds_budget / 3:
            break
    dt2 = time.perf_counter() - t0
    out.update({"port_all_cores_value": n * (n - 1) * reps / dt2, "port_all_cores": thr,
                "host_cpus": os.cpu_count()})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bodies", dest="n", type=int, default=0, help="number of bodies (default: weak-scaled from 262144)")
    ap.add_argument("--dt", type=float, default=0.01)
    ap.add_argument("--eps2", type=float, default=0.002)
    ap.add_argument("--init", type=int, default=1, help="0 reference cube, 1 Plummer")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--bpl", type=int, default=0)
    ap.add_argument("--jsplit", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="f64 = the build's own double-precision variant "
                    "(BASELINE configs[4]; single GPU only)")
    ap.add_argument("--backend", default="nccl", help="collective backend for --gpus > 1: nccl (RCCL; default) or gloo "
                    "(rehearsal of the multi-rank path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import nbody_amd
    if not os.path.exists(nbody_amd._lib.LIB_PATH) and int(os.environ.get("RANK", "0")) == 0:
        import __graft_entry__            # un-built snapshot: compile the HIP library in-tree first
        __graft_entry__.build()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus > 1 launch through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank (use --backend gloo to rehearse)")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    n = args.n or weak_n(world)
    x0 = nbody_amd.engine.seeded_bodies(n, args.init, 12345)
    kopts = dict(tile=args.tile, bodies_per_lane=args.bpl, jsplit=args.jsplit)

    f64 = args.dtype == "f64"
    if f64 and world > 1:
        raise SystemExit("--dtype f64 is a single-GPU variant")
    if f64:
        import numpy as np

        class _F64Sim:   # same alloc/init/H2D sequence as engine.Simulation, double state
            def __init__(self):
                self.ctx = nbody_amd.engine.Context(device=dev.index, jsplit=args.jsplit)
                self.x = torch.from_numpy(x0.astype(np.float64)).to(dev)
                self.v = torch.zeros_like(self.x)
                self.a = torch.zeros_like(self.x)
                self.shard, self.n_pad = n, n

            def run(self, k, sync=False):
                self.ctx.step_f64(self.x, self.a, self.v, args.dt, args.eps2, k)

        sim = _F64Sim()
        ctx = sim.ctx
        run = lambda k: sim.run(k)
        sync = ctx.sync
        info = {"jsplit": args.jsplit or "auto", "kernel": "nbk::force_f64<2,512>"}
    elif world == 1:
        sim = nbody_amd.engine.Simulation(x0, dt=args.dt, eps2=args.eps2, device=dev.index, **kopts)
        ctx = sim.ctx
        run = lambda k: sim.run(k, sync=False)
        sync = ctx.sync
        info = ctx.launch_info(n, n)
    else:
        backend = nbody_amd.sharded.HipBackend(dev, args.dt, args.eps2, **kopts)
        sim = nbody_amd.sharded.ShardedSimulation(x0, dt=args.dt, eps2=args.eps2, backend=backend)
        ctx = sim.backend.ctx
        run = sim.step
        sync = sim.sync
        info = ctx.launch_info(sim.shard, sim.shard)   # the local pass; the remote passes differ in source count

    def barrier():
        sync()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    run(args.warmup)
    barrier()
    if world > 1:
        backend.comm_timing = True
    ctx.timing(True)
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    force_ms, launches = ctx.timing_read()
    ctx.timing(False)
    comm = backend.comm_report() if world > 1 else None

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs_total = float(n) * n * args.steps
    value = pairs_total / elapsed
    # roofline of the dominant kernel (force accumulation), from its own event time on this rank
    rank_pairs = float(sim.shard) * sim.n_pad * args.steps if world > 1 else pairs_total
    kernel_s = force_ms * 1e-3
    if f64:   # the fp64 launches are not event-bracketed: the step is >99.9 % force kernel, use the wall time
        kernel_s, launches, force_ms = elapsed, args.steps, elapsed * 1e3
    achieved = FLOP_PER_PAIR * rank_pairs / kernel_s / 1e12 if kernel_s > 0 else 0.0
    peak = 78.6 if f64 else FP32_VECTOR_PEAK_TFLOPS
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and n == BASE_N and not f64 and os.path.exists(tpath):   # the PMC passes were taken on this workload only
        try:
            traffic = json.load(open(tpath)).get("force_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    line = {
        "metric": "body_pair_interactions_per_s",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"all-pairs gravity step, N={n} bodies, {'fp64' if f64 else 'fp32'}, {'Plummer' if args.init == 1 else 'reference-cube'} init seed 12345, "
                        f"dt={args.dt}, eps2={args.eps2}",
            "n_bodies": n,
            "pairs_per_step": float(n) * n,
            "partition": "single GPU" if world == 1 else f"{world} contiguous blocks of {sim.shard} bodies, {args.backend} all-gather of positions per step",
            "kernel": nbody_amd.load().nbody_version().decode(),
            "launch": info,
            "gflops_at_20_flop_per_pair": value * FLOP_PER_PAIR / 1e9,
            **({"comm_rank0": comm} if comm else {}),
        },
        "roofline": {
            "bound": "valu",
            "achieved": achieved,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": achieved / peak,
            "traffic": traffic,
            "kernel": "nbk::force_f64" if f64 else "nbk::force_lds (fp32 packed)",
            "kernel_ms_avg": force_ms / max(launches, 1),
            "kernel_launches": launches,
            "flop_per_pair": FLOP_PER_PAIR,
            "note": ("fp64 vector-ALU bound; peak = 78.6 TFLOP/s fp64 vector" if f64 else
                     "fp32 vector-ALU bound (no MFMA, HBM traffic is O(N) per step); peak = 157.3 TFLOP/s fp32 vector = fp32 MFMA peak"),
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not f64:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

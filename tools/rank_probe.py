"""Developer probe: the per-step work of ONE rank of the sharded run, timed alone on one GPU with no-op collectives
(nothing is exchanged, so the numbers are meaningless as physics — only the launch sequence and its time are real).
Predicts the compute side of multi-GPU scaling from a single GPU.
usage: rank_probe.py [--bodies N] [--kernel fast|onesided] G [G ...]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import nbody_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=1048576)
ap.add_argument("--kernel", default="fast")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--sym-waves", type=int, default=0)
ap.add_argument("--sym-bpl", type=int, default=0)
ap.add_argument("--no-equal-mass", action="store_true", help="the general pair arithmetic (what unequal masses get: bench.py's headline path)")
ap.add_argument("worlds", type=int, nargs="*", default=[1, 2, 4, 8])
args = ap.parse_args()
L, lib = nbody_amd._lib, nbody_amd.load()
x0 = nbody_amd.engine.seeded_bodies(args.bodies, 1, 1)
kernel = {"fast": nbody_amd.KERNEL_FAST, "onesided": nbody_amd.KERNEL_ONESIDED}[args.kernel]
for G in args.worlds:
    rank = G // 2
    ctx = nbody_amd.engine.Context(dt=0.01, kernel=kernel)
    if args.sym_waves or args.sym_bpl:
        ctx.set_symmetric_shape(args.sym_waves, args.sym_bpl)
    if args.no_equal_mass:
        ctx.set_equal_mass(0)
    g = L.ALL_GATHER_FN(lambda *a: 0)
    e = L.EXCHANGE_FN(lambda *a: 0)
    comm = L.Comm(None, g, e)
    h = C.c_void_p()
    L.check(lib.nbody_shard_create(C.byref(h), ctx._h, rank, G, args.bodies, C.byref(comm)))
    L.check(lib.nbody_shard_upload(h, C.c_void_p(x0.ctypes.data)))
    plan = L.ShardPlan()
    L.check(lib.nbody_shard_get_plan(h, C.byref(plan)))
    L.check(lib.nbody_shard_step(h, 2))
    L.check(lib.nbody_shard_sync(h))
    ctx.timing(True)
    t = time.perf_counter()
    L.check(lib.nbody_shard_step(h, args.steps))
    L.check(lib.nbody_shard_sync(h))
    dt = (time.perf_counter() - t) / args.steps
    force_ms, launches = ctx.timing_read()
    verdict = ctx.equal_mass_verdict()
    n = args.bodies
    print(json.dumps({"G": G, "rank": rank, "n": n, "shard": plan.shard, "schedule": plan.schedule,
                      "ms_per_step": round(dt * 1e3, 3), "force_ms_per_step": round(force_ms / args.steps, 3),
                      "force_launches_per_step": launches // args.steps, "equal_mass": verdict,
                      "rank_interactions_per_s": float("%.4g" % (plan.shard * float(plan.n_pad) / dt)),
                      "job_pairs_per_s_if_all_ranks_equal": float("%.4g" % (float(n) * n / dt))}), flush=True)
    L.check(lib.nbody_shard_destroy(h))
    ctx.close()

"""Developer probe: us/step and interactions/s across N for the one-sided kernel (auto shape) and every block shape
of the symmetric kernel, queued steps (no host sync inside the timed loop). Feeds the FAST kernel's switch-over size
and the symmetric block-shape choice in nbody_plan.hip.  usage: smalln_probe.py [N ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd  # noqa: E402

SHAPES = [(1, 2), (1, 4), (2, 4), (1, 8), (2, 8), (2, 10), (2, 12), (4, 8), (4, 10)]
if os.environ.get("PROBE_SHAPES"):
    SHAPES = [tuple(int(t) for t in sh.split("x")) for sh in os.environ["PROBE_SHAPES"].split(",")]
sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 16384, 32768, 65536, 131072]
for n in sizes:
    x0 = nbody_amd.engine.seeded_bodies(n, 1, 1)
    steps = max(16, min(2000, int(2e11 / (float(n) * n))))
    row = {"n": n, "steps": steps}

    def timed(sim):
        sim.run(max(steps // 8, 4))
        sim.ctx.sync()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            sim.run(steps, sync=False)
            sim.ctx.sync()
            best = min(best, time.perf_counter() - t)
        return {"us_per_step": round(best / steps * 1e6, 2), "pairs_per_s": float("%.4g" % (float(n) * n * steps / best))}

    sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nbody_amd.KERNEL_ONESIDED)
    row["onesided"] = dict(timed(sim), launch=sim.ctx.launch_info(n, n))
    for (w, b) in SHAPES:
        if -(-n // (64 * w * b)) < 2:
            continue
        sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nbody_amd.KERNEL_SYMMETRIC)
        sim.ctx.set_symmetric_shape(w, b)
        sim.ctx.reserve(n)
        info = sim.ctx.step_info(n)
        if not info["symmetric"]:
            continue
        row[f"sym_w{w}_bpl{b}"] = dict(timed(sim), slabs=info["slabs"], workgroups=info["workgroups"])
    for b in (8, 10):
        sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nbody_amd.KERNEL_SYMMETRIC)
        sim.ctx.set_symmetric_shape(0, b)
        sim.ctx.set_symmetric_runs(1)
        sim.ctx.reserve(n)
        info = sim.ctx.step_info(n)
        row[f"runs_bpl{b}"] = dict(timed(sim), slabs=info["slabs"], workgroups=info["workgroups"])
    for b in (4, 8, 10):
        sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nbody_amd.KERNEL_SYMMETRIC)
        sim.ctx.set_symmetric_shape(0, b)
        sim.ctx.set_symmetric_runs(2)
        sim.ctx.reserve(n)
        info = sim.ctx.step_info(n)
        if info["balanced"]:
            row[f"balanced_bpl{b}"] = dict(timed(sim), records=info["slabs"], workgroups=info["workgroups"])
    sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nbody_amd.KERNEL_FAST)
    row["fast_auto"] = dict(timed(sim), info=sim.ctx.step_info(n))
    print(json.dumps(row), flush=True)

"""Developer probe: whole steps across N with the automatic choice, equal-mass path off and on, and — on the equal-mass path — the
decompositions the automatic choice competes with (balanced runs, unit runs, block pairs), to check that the switch-over sizes still
hold there.  usage: eq_curve.py [N ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [9216, 12288, 16384, 24576, 32768, 40960, 49152, 65536, 98304, 131072, 163840, 196608, 262144]
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
        return round(best / steps * 1e6, 2)

    def kind(info):
        return "fused" if info["fused"] else "balanced" if info["balanced"] else "runs" if info["runs"] else "blocks" if info["symmetric"] else "onesided"

    for eq in (0, 1):
        sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002)
        sim.ctx.set_equal_mass(eq)
        row["auto_eq_on" if eq else "auto_eq_off"] = {"us": timed(sim), "kind": kind(sim.ctx.step_info(n))}
    for name, runs in (("balanced", 2), ("runs", 1), ("blocks", 0)):
        sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nbody_amd.KERNEL_SYMMETRIC)
        sim.ctx.set_symmetric_runs(runs)
        sim.ctx.set_equal_mass(1)
        try:
            sim.ctx.reserve(n)
        except nbody_amd.NBodyError:
            continue
        info = sim.ctx.step_info(n)
        if kind(info) != name:
            continue
        row[name + "_eq_on"] = {"us": timed(sim), "block": info["block_bodies"]}
    row["pairs_per_s_auto_eq_on"] = float("%.4g" % (float(n) * n / row["auto_eq_on"]["us"] * 1e6))
    row["pairs_per_s_auto_eq_off"] = float("%.4g" % (float(n) * n / row["auto_eq_off"]["us"] * 1e6))
    print(json.dumps(row), flush=True)

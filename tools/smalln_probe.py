"""Developer probe: steps/s and pairs/s at small N with and without hipGraph replay."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd
for n in (1024, 4096, 8192, 16384, 32768, 65536):
    x0 = nbody_amd.engine.seeded_bodies(n, 1, 1)
    row = {"n": n}
    for mode in (0, 1):
        sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002)
        sim.ctx.set_graph(mode)
        steps = 640 if n <= 16384 else 128
        sim.run(64)
        t = time.perf_counter(); sim.run(steps); dt = time.perf_counter() - t
        row["graph" if mode else "eager"] = {"us_per_step": round(dt / steps * 1e6, 2), "pairs_per_s": float("%.4g" % (n * n * steps / dt))}
    row["launch"] = sim.ctx.launch_info(n, n)
    print(json.dumps(row))

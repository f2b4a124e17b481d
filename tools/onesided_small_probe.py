"""Developer probe: the one-sided kernel's launch shapes (tile, targets per lane, slabs) at N = 4096 ... 16384,
queued steps. Confirms that the automatic choice is the best of them (profiles/r02_onesided_small_probe.txt)."""
import sys, time, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd
for n in (4096, 8192, 16384):
    x0 = nbody_amd.engine.seeded_bodies(n, 1, 1)
    steps = 2000
    for (tile, bpl, js) in [(0,0,0),(256,1,16),(256,1,32),(256,2,16),(256,2,32),(256,2,64),(512,2,16),(256,4,32),(256,4,64),(512,4,16)]:
        try:
            sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nbody_amd.KERNEL_ONESIDED, tile=tile, bodies_per_lane=bpl, jsplit=js)
        except Exception as e:
            print(n, tile, bpl, js, "ERR", e); continue
        sim.run(200); sim.ctx.sync()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); sim.run(steps, sync=False); sim.ctx.sync(); best = min(best, time.perf_counter() - t)
        print(n, (tile, bpl, js), sim.ctx.launch_info(n, n), round(best / steps * 1e6, 2), "us", "%.2e" % (n * n * steps / best), flush=True)

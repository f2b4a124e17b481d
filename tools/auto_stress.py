"""Developer stress: 40 random sizes between 12288 and 200000 through nbody_step with the automatically chosen kernel
(run-based or block pairs): sampled targets against the fp64-accumulated CPU sums, momentum balance."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import nbody_amd as nb
from oracle import oracle
oracle.build()
rng = np.random.default_rng(3)
worst = 0.0
kinds = {}
for case in range(40):
    n = int(rng.integers(12288, 200000))
    x0 = nb.engine.seeded_bodies(n, case % 2, 500 + case)
    sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    info = sim.ctx.step_info(n)
    kinds[(info["symmetric"], info["runs"])] = kinds.get((info["symmetric"], info["runs"]), 0) + 1
    sim.run(1)
    x, v, a = sim.state()
    for i0 in (0, n // 2, n - 128):
        t = oracle.accel_range(x0, i0, i0 + 128, 0, n, eps2=0.002, f64acc=True)
        e = np.abs(a[i0:i0 + 128] - t)[:, :3].max() / np.abs(t[:, :3]).max()
        worst = max(worst, e)
        assert e <= 2e-5, (case, n, info, e)
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * a[:, :3]).sum(0)).max() / (m * np.abs(a[:, :3])).sum() < 1e-6, (case, n)
print("40 cases ok, worst", worst, kinds)

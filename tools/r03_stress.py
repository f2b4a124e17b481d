"""Developer stress (round 3): random sizes through nbody_step with the automatically chosen kernel — the fused one-launch step up to
8192 bodies, balanced runs to 45056, unit runs / block pairs above — and through the balanced runs forced at small sizes with every
bodies-per-lane shape: sampled targets against the fp64-accumulated CPU sums, momentum balance, several steps (odd and even counts).
    python tools/r03_stress.py [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import nbody_amd as nb  # noqa: E402
from oracle import oracle  # noqa: E402

oracle.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
worst, kinds = 0.0, {}


def check(sim, x0, n, tag):
    global worst
    sim.run(1)
    x, v, a = sim.state()
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(x)), tag
    for i0 in sorted({0, max(0, n // 2 - 64), max(0, n - 128)}):
        i1 = min(i0 + 128, n)
        t = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002, f64acc=True)
        e = np.abs(a[i0:i1] - t)[:, :3].max() / max(np.abs(t[:, :3]).max(), 1e-30)
        worst = max(worst, e)
        assert e <= 2e-5, (tag, n, i0, e)
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * a[:, :3]).sum(0)).max() / max((m * np.abs(a[:, :3])).sum(), 1e-300) < 1e-6, (tag, n)


cases = 0
for case in range(60):        # the automatic choice over the whole small / mid range
    lo, hi = [(1, 600), (600, 8193), (8193, 12000), (12000, 46000), (46000, 120000)][case % 5]
    n = int(rng.integers(lo, hi))
    x0 = nb.engine.seeded_bodies(n, case % 2, 900 + case)
    sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    info = sim.ctx.step_info(n)
    kind = "fused" if info["fused"] else "balanced" if info["balanced"] else "runs" if info["runs"] else "blocks" if info["symmetric"] else "onesided"
    kinds[kind] = kinds.get(kind, 0) + 1
    check(sim, x0, n, (kind, case))
    cases += 1
for case in range(40):        # balanced runs forced, every bodies-per-lane shape, sizes where units are shared by up to five workers
    bpl = (2, 4, 8, 10)[case % 4]
    n = int(rng.integers(128, 9000))
    x0 = nb.engine.seeded_bodies(n, case % 2, 1900 + case)
    sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
    sim.ctx.set_symmetric_shape(0, bpl)
    sim.ctx.set_symmetric_runs(2)
    info = sim.ctx.step_info(n)
    if not info["balanced"]:
        continue
    check(sim, x0, n, ("balanced-forced", bpl, case))
    cases += 1
for case in range(12):        # fused step: several steps in odd and even splits equal one call, bit for bit
    n = int(rng.integers(1, 8193))
    x0 = nb.engine.seeded_bodies(n, 1, 2900 + case)
    a_sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    a_sim.run(5)
    b_sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    for k in (2, 3):
        b_sim.run(k)
    for p, q in zip(a_sim.state(), b_sim.state()):
        assert np.array_equal(p, q), ("fused split", n)
    cases += 1
eq_taken, worst_pair = 0, 0.0
for case in range(48):        # equal-mass path: random sizes, every symmetric decomposition and block shape, shifted and rescaled systems
    lo, hi = [(4096, 12000), (12000, 46000), (46000, 170000), (170000, 400000)][case % 4]
    n = int(rng.integers(lo, hi))
    x0 = nb.engine.seeded_bodies(n, 1, 3900 + case)
    size = 10.0 ** rng.uniform(-3, 6)
    x0[:, :3] *= np.float32(size)                                          # size of the system
    x0[:, :3] += (rng.normal(0, 1, 3) * size * 10.0 ** rng.uniform(-1, 2)).astype(np.float32)   # where it sits (far inside the 1e15 bound)
    x0[:, 3] = np.float32(10.0 ** rng.uniform(-20, 12))                    # the common mass
    uniform = case % 6 != 5
    if not uniform:
        x0[int(rng.integers(0, n)), 3] *= np.float32(1.5)
    shape = [(0, 0), (4, 10), (4, 8), (2, 10), (1, 8), (2, 4)][case % 6] if case % 3 == 0 else (0, 0)
    runs_mode = [-1, 1, 2][case % 3] if shape == (0, 0) else 0
    out = {}
    for mode in (1, 0):
        sim = nb.engine.Simulation(x0, dt=0.0, eps2=float(np.float32(0.002 * size * size)),
                                   kernel=nb.KERNEL_SYMMETRIC if shape != (0, 0) or runs_mode > 0 else nb.KERNEL_FAST)
        if shape != (0, 0):
            sim.ctx.set_symmetric_shape(*shape)
        if runs_mode >= 0:
            try:
                sim.ctx.set_symmetric_runs(runs_mode)
            except nb.NBodyError:
                pass
        sim.ctx.set_equal_mass(mode)
        sim.run(1)
        out[mode] = sim.state()[2]
        if mode == 1:
            v = sim.ctx.equal_mass_verdict()
            info = sim.ctx.step_info(n)
            if info["symmetric"]:
                assert v["scanned"] and v["uniform"] == uniform, (case, n, v, info)
                eq_taken += int(uniform)
    scale = max(np.abs(out[0][:, :3]).max(), 1e-300)
    assert np.isfinite(out[1]).all(), (case, n)
    d = np.abs(out[1] - out[0])[:, :3].max() / scale
    worst_pair = max(worst_pair, d)
    assert d <= 1e-5, (case, n, d)      # two fp32 summation roundings of up to 4e5 terms each
    if not uniform:
        assert np.array_equal(out[1], out[0]), (case, n)
    cases += 1
print(f"{cases} cases ok, worst rel err {worst:.3g}, automatic choice: {kinds}; equal-mass path taken in {eq_taken} cases, "
      f"worst difference to the general path {worst_pair:.3g} of max|a|")

"""Developer stress of the in-place block-pair kernel (nbk::force_sym_ticket): random sizes from 1300 to 700000 bodies, both block
shapes (640- and 2560-body blocks), every number of accumulation lanes a workspace cap can leave (8, 4, 2, 1 = straight into the
acceleration array), general and equal-mass path, whole steps and nbody_accel_range on an offset square block. Each case: sampled
targets against the fp64-accumulated CPU sums, the momentum balance of the whole system, and a second run that must give the same bits.

    python tools/inplace_stress.py [cases] [seed]        (on an MI355X; profiles/r06k_inplace_stress.txt)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import nbody_amd as nb
from oracle import oracle

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 48
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
worst, worst_mom, kinds = 0.0, 0.0, {}
for case in range(cases):
    big = case % 6 == 5
    n = int(rng.integers(160000, 700000)) if big else int(rng.integers(1300, 160000))
    init = int(rng.integers(0, 2))                    # 0: the reference's cube and unequal masses; 1: Plummer, every body 1/N
    x0 = nb.engine.seeded_bodies(n, init, 900 + case)
    lanes_cap = (8, 4, 2, 1)[int(rng.integers(0, 4))]
    sims = []
    for rep in range(2):
        sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
        sim.ctx.set_inplace_sums(1)
        sim.ctx.set_equal_mass(-1 if init == 1 else 0)
        if lanes_cap < 8:
            sim.ctx.set_workspace_limit(max(1, (2 * lanes_cap - 1) * n * 16 - 1) if lanes_cap > 1 else 1)
        info = sim.ctx.step_info(n)
        assert info["ticket"], (case, n, info)
        want = 0 if lanes_cap == 1 else lanes_cap
        nblk = -(-n // info["block_bodies"])
        while want > 1 and want * 2 > nblk:           # never more lanes than half the blocks
            want //= 2
        assert info["slabs"] == (want if want > 1 else 0), (case, n, lanes_cap, info)
        sim.run(1)
        sims.append(sim.state())
    for p, q in zip(*sims):
        assert np.array_equal(p, q), (case, n, "not reproducible")
    a = sims[0][2]
    for i0 in (0, n // 2, n - 128):
        t = oracle.accel_range(x0, i0, i0 + 128, 0, n, eps2=0.002, f64acc=True)
        e = np.abs(a[i0:i0 + 128] - t)[:, :3].max() / np.abs(t[:, :3]).max()
        worst = max(worst, e)
        assert e <= 2e-5, (case, n, info, e)
    m = x0[:, 3:4].astype(np.float64)
    mom = np.abs((m * a[:, :3]).sum(0)).max() / (m * np.abs(a[:, :3])).sum()
    worst_mom = max(worst_mom, mom)
    assert mom < 1e-6, (case, n, mom)
    key = (info["block_bodies"], info["slabs"], "eq" if init == 1 else "gen")
    kinds[key] = kinds.get(key, 0) + 1
    if case % 8 == 0:
        print(f"case {case}: n={n} {key} ok", flush=True)

# nbody_accel_range on a square block that does not start at body 0, under caps that leave 8 / 4 / 2 lanes
for case in range(12):
    n = int(rng.integers(20000, 90000))
    i0 = int(rng.integers(1, 5000))
    nt = n - i0 - int(rng.integers(0, 3000))
    x0 = nb.engine.seeded_bodies(n, 0, 1200 + case)
    x = torch.from_numpy(x0).cuda()
    lanes = (8, 4, 2)[case % 3]
    ctx = nb.engine.Context(dt=0.01, eps2=0.002)
    ctx.set_workspace_limit((2 * lanes - 1) * nt * 16 - 1 if lanes < 8 else 8 * nt * 16)
    assert ctx.launch_info(nt, nt)["jsplit"] == lanes, (case, nt, lanes, ctx.launch_info(nt, nt))
    outs = []
    for rep in range(2):
        out = torch.full((nt, 4), -3.0, device="cuda")
        ctx.accel_range(x, out, i0, i0 + nt, i0, i0 + nt)
        ctx.sync()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    t = oracle.accel_range(x0, i0 + nt - 256, i0 + nt, i0, i0 + nt, eps2=0.002, f64acc=True)
    e = np.abs(outs[0][nt - 256:] - t)[:, :3].max() / np.abs(t[:, :3]).max()
    worst = max(worst, e)
    assert e <= 2e-5, (case, nt, e)
print(f"{cases} whole-step cases + 12 square-block cases ok; worst |da|/max|a| vs fp64-accumulated sums {worst:.3g}; worst momentum imbalance {worst_mom:.3g}; "
      f"(block bodies, lanes, path) x cases: {sorted(kinds.items())}")

"""Developer stress: 120 random sizes x every block shape of the symmetric kernel against the fp64-accumulated CPU sums
(tools/sym_stress.py [seed]; worst error seen 1.9e-6 of max|a|)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import nbody_amd as nb
from oracle import oracle
oracle.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
shapes = [(1, 2), (1, 4), (2, 4), (1, 8), (1, 10), (2, 8), (2, 10), (4, 8), (4, 10), (0, 0)]
worst = 0.0
for case in range(120):
    w, b = shapes[case % len(shapes)]
    blk = 64 * max(w, 1) * max(b, 2)
    n = int(rng.integers(2 * blk, 2 * blk + 6000))
    init = case % 2
    x0 = nb.engine.seeded_bodies(n, init, 1000 + case)
    ctx = nb.engine.Context(kernel=nb.KERNEL_SYMMETRIC)
    ctx.set_symmetric_shape(w, b)
    x = torch.from_numpy(x0).cuda()
    a = torch.zeros_like(x)
    ctx.accel_range(x, a, 0, n, 0, n)
    ctx.sync()
    truth = oracle.accel_range(x0, 0, n, eps2=0.002, f64acc=True)
    e = np.abs(a.cpu().numpy() - truth)[:, :3].max() / np.abs(truth[:, :3]).max()
    worst = max(worst, e)
    assert e <= 1e-5, (case, n, w, b, e)
print("120 cases ok, worst rel err", worst)

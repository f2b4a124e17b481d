"""Developer probe: the per-step work of ONE rank of the weak-scaled sharded run (no communication),
to predict scaling efficiency from a single GPU. usage: rank_probe.py G [G ...]"""
import sys, os, time, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd
def weak_n(g):
    q = 8192 * g
    return 262144 if g == 1 else int(round(262144 * math.sqrt(g) / q)) * q
for G in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    n = weak_n(G); S = n // G
    x = torch.from_numpy(nbody_amd.engine.seeded_bodies(n, 1, 1)).cuda()
    v = torch.zeros((S, 4), device="cuda"); a = torch.zeros((S, 4), device="cuda")
    ctx = nbody_amd.engine.Context(dt=0.01); ctx.reserve(S)
    r = G // 2; i0, i1 = r * S, (r + 1) * S
    def step():
        ctx.accel_range(x, a, i0, i1, i0, i1, False)
        if G > 1:
            ctx.accel_wrapped(x, a, i0, i1, i1 % n, n - S, True)
        ctx.integrate_range(x, v, a, i0, i1)
    for _ in range(3): step()
    ctx.sync(); t = time.perf_counter()
    K = 10
    for _ in range(K): step()
    ctx.sync(); dt = (time.perf_counter() - t) / K
    print(json.dumps({"G": G, "n": n, "shard": S, "ms_per_step": round(dt * 1e3, 3), "rank_pairs_per_s": float("%.4g" % (S * n / dt)),
                      "job_pairs_per_s_if_all_ranks_equal": float("%.4g" % (n * n / dt))}))

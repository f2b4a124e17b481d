"""Developer probe: time one accel_range launch (targets x sources) for several slab counts."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd
nt, ns = int(sys.argv[1]), int(sys.argv[2])
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = max(nt, ns)
x = torch.from_numpy(nbody_amd.engine.seeded_bodies(n, 1, 1)).cuda()
a = torch.zeros((nt, 4), device="cuda")
for js in [int(v) for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else "0,8,16,24,32,45,48,64".split(","))]:
    ctx = nbody_amd.engine.Context(tile=tile, jsplit=js)
    ctx.reserve(nt)
    for _ in range(2):
        ctx.accel_range(x, a, 0, nt, 0, ns)
    ctx.sync()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        ctx.accel_range(x, a, 0, nt, 0, ns)
    ctx.sync()
    dt = (time.perf_counter() - t) / reps
    info = ctx.launch_info(nt, ns)
    print(json.dumps({"nt": nt, "ns": ns, "jsplit": info["jsplit"], "blocks": info["blocks"], "ms": round(dt * 1e3, 3),
                      "pairs_per_s": float("%.4g" % (nt * ns / dt))}))

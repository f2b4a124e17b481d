"""Developer probe: fp64 step rate and fp32-vs-fp64 position agreement (BASELINE configs[4])."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x0 = nbody_amd.engine.seeded_bodies(n, 1, 12345)
ctx = nbody_amd.engine.Context()
x = torch.from_numpy(x0.astype(np.float64)).cuda(); v = torch.zeros_like(x); a = torch.zeros_like(x)
ctx.step_f64(x, a, v, 0.01, 0.002, 1); ctx.sync()
t = time.perf_counter(); ctx.step_f64(x, a, v, 0.01, 0.002, steps); ctx.sync(); dt = time.perf_counter() - t
sim = nbody_amd.engine.Simulation(x0, dt=0.01, eps2=0.002); sim.run(steps + 1)
x32, _, _ = sim.state()
d = np.abs(x32 - x.cpu().numpy())[:, :3].max()
print(json.dumps({"n": n, "steps": steps, "f64_pairs_per_s": n * n * steps / dt, "f64_ms_per_step": dt / steps * 1e3,
                  "f64_tflops_at_20": 20 * n * n * steps / dt / 1e12, "max_abs_dx_f32_vs_f64": float(d)}))

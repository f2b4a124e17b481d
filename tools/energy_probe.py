"""tools/energy_probe.py — KE + PE/2 (the invariant of the reference's update rule, kernel.cu:116-129) along a configs[1] run:
N=65536, dt=0.01, Plummer, cold start. Prints the terms every 100 steps (test thresholds of test_config2_full_run_* come from here)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import nbody_amd as nb
from test_gpu_parity import _energy_terms

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dt, eps2 = 0.01, 0.002
x0 = nb.engine.seeded_bodies(n, 1, 12345)
sim = nb.engine.Simulation(x0, dt=dt, eps2=eps2)
ke0, pe0 = _energy_terms(x0, np.zeros_like(x0), eps2)
h0 = ke0 + 0.5 * pe0
print(json.dumps({"step": 0, "ke": ke0, "pe": pe0, "h": h0, "mass": float(x0[:, 3].sum())}), flush=True)
for k in range(100, total + 1, 100):
    sim.run(100)
    x, v, a = sim.state()
    ke, pe = _energy_terms(x, v, eps2)
    m = x0[:, 3:4].astype(np.float64)
    p = np.abs((m * v[:, :3]).sum(0)).max() / (m * np.abs(v[:, :3])).sum()
    print(json.dumps({"step": k, "ke": ke, "pe": pe, "h": ke + 0.5 * pe, "rel_dh": (ke + 0.5 * pe - h0) / abs(h0), "rel_p": p}), flush=True)

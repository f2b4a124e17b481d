"""Developer probe: acceleration error of the fast kernel and of the fp32 sequential CPU sum, both
against the fp64-accumulated truth, on 1024 sampled targets, as N grows (Plummer and reference init)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd
from oracle import oracle as O
for init, name in ((1, "plummer"), (0, "reference-cube")):
    for n in (1024, 8192, 65536, 262144):
        x0 = nbody_amd.engine.seeded_bodies(n, init, 2024)
        ctx = nbody_amd.engine.Context()
        x = torch.from_numpy(x0).cuda(); a = torch.zeros_like(x)
        ctx.accel_range(x, a, 0, n, 0, n); ctx.sync()
        ag = a.cpu().numpy()
        idx = np.linspace(0, n - 1024, 4).astype(int)
        e_gpu, e_seq, scale = [], [], 0.0
        for i0 in idx:
            t = O.accel_range(x0, int(i0), int(i0) + 256, 0, n, eps2=0.002, f64acc=True)
            s = O.accel_range(x0, int(i0), int(i0) + 256, 0, n, eps2=0.002)
            scale = max(scale, np.abs(t[:, :3]).max())
            e_gpu.append(np.abs(ag[i0:i0 + 256] - t)[:, :3]); e_seq.append(np.abs(s - t)[:, :3])
        e_gpu, e_seq = np.concatenate(e_gpu), np.concatenate(e_seq)
        print(json.dumps({"init": name, "n": n, "gpu_fast_max_rel": float("%.3g" % (e_gpu.max() / scale)),
                          "gpu_fast_rms_rel": float("%.3g" % (np.sqrt((e_gpu ** 2).mean()) / scale)),
                          "cpu_fp32_seq_max_rel": float("%.3g" % (e_seq.max() / scale)),
                          "cpu_fp32_seq_rms_rel": float("%.3g" % (np.sqrt((e_seq ** 2).mean()) / scale))}))

"""What nbody_simulate()'s one-off measurement finds on the box at hand, for sizes around the built-in switch-over sizes
(8192, 45056, 160000): decomposition id (0 = built-in kept), built-in and best microseconds per queued step.
    python tools/autotune_probe.py [n ...]   -> one JSON line per size (profiles/r04_autotune_probe.jsonl)"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [6144, 7168, 8192, 9216, 10240, 36864, 40000, 45056, 49152, 56000, 131072, 160000, 196608]
lib = nbody_amd.load()
for n in sizes:
    x = torch.from_numpy(nbody_amd.engine.seeded_bodies(n, 0, 12345)).cuda()
    nbody_amd._lib.check(lib.nbody_simulate_prepare(C.c_void_p(x.data_ptr()), n))
    r = nbody_amd.engine.simulate_autotuned(n)
    info = nbody_amd.engine.Context().step_info(n)
    print(json.dumps({"n": n, **r, "builtin": {k: info[k] for k in ("fused", "balanced", "runs", "symmetric", "block_bodies")}}), flush=True)

"""Regenerates tests/golden/*.npz.  Run from the repo root IN THE BUILD CONTAINER (it needs
oracle/_ref/libref_cpu.so, i.e. the reference's own CPU path compiled by `make -C oracle`):

    python tests/golden/make_golden.py

Every file holds inputs and expected outputs only (data, no reference source):

  ref_cpu_n1024.npz     x0 = the reference's fill_with_random4 after srand(1) (utils.cpp:30-37, glibc
                        rand()), v0 = a0 = 0; x/v/a after K in {1,10,100} calls of the REFERENCE's
                        CPU_compute (validation.cpp:28-52; DT=0.1f, EPS2=0.002f compiled in).
  ref_cpu_n8192.npz     the reference's SHIPPED configuration (constants.h:13,25-26: N_BODIES 8192, DT 0.1f,
                        EPS2 0.002f, unseeded fill_with_random4): x/a after K = 1 and x/v/a after K = 10
                        calls of the reference's CPU_compute. x0 itself is not stored (128 KiB): it is
                        fill_with_random4 after srand(1), which the product's generator reproduces bit
                        for bit; x0_head holds its first 8 bodies as a check.
  ref_cpu_plummer_n1024_dt0.01.npz
                        the BENCHMARK's time step, held by the reference build: x0 = the seeded Plummer sphere
                        (seed 12345, the bodies of jacobi_plummer_n1024_dt0.01.npz), x/v/a after K in {1,10,100}
                        calls of the reference's CPU_compute compiled with DT 0.01f instead of constants.h:26's
                        0.1f (oracle/ref_dt001.cpp -> oracle/_ref/libref_cpu_dt001.so: validation.cpp compiled
                        where it lies, the DT macro the only change; EPS2 0.002f as shipped).
  ref_pairs.npz         256 random (bi,bj,ai) triples and the reference's bodyInteractions_CPU result.
  jacobi_*.npz          outputs of OUR oracle's Jacobi step (fp32 sequential) for seeded inputs, so
                        the GPU box can check the strict kernel bit-for-bit without re-running the
                        oracle, and detect drift of the oracle itself.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import nbody_amd  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle)"
    libc = ctypes.CDLL(None)

    # --- the reference itself ---------------------------------------------------------------
    n = 1024
    libc.srand(1)  # an unseeded process starts in this state (main.cpp never calls srand)
    x0 = O.ref_fill_with_random4(n)
    out = {"x0": x0}
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    done = 0
    for K in (1, 10, 100):
        O.ref_step(x, a, v, steps=K - done)
        done = K
        out[f"x_{K}"], out[f"v_{K}"], out[f"a_{K}"] = x.copy(), v.copy(), a.copy()
    np.savez_compressed(os.path.join(OUT, "ref_cpu_n1024.npz"), **out)

    # --- the reference's own shipped size ------------------------------------------------------
    n = 8192
    libc.srand(1)
    x0 = O.ref_fill_with_random4(n)
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    out = {"x0_head": x0[:8].copy()}
    O.ref_step(x, a, v, steps=1)
    out["x_1"], out["a_1"] = x.copy(), a.copy()
    O.ref_step(x, a, v, steps=9)
    out["x_10"], out["v_10"], out["a_10"] = x.copy(), v.copy(), a.copy()
    np.savez_compressed(os.path.join(OUT, "ref_cpu_n8192.npz"), **out)

    # --- the reference at the benchmark's dt (BASELINE configs[1], [2], [4]: dt = 0.01) -----------
    assert O.have_ref_dt001(), "build oracle/_ref/libref_cpu_dt001.so first (make -C oracle)"
    x0 = nbody_amd.engine.seeded_bodies(1024, 1, 12345)
    out = {"x0": x0, "dt": np.float32(0.01), "eps2": np.float32(0.002)}
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    done = 0
    for K in (1, 10, 100):
        O.ref_step_dt001(x, a, v, steps=K - done)
        done = K
        out[f"x_{K}"], out[f"v_{K}"], out[f"a_{K}"] = x.copy(), v.copy(), a.copy()
    np.savez_compressed(os.path.join(OUT, "ref_cpu_plummer_n1024_dt0.01.npz"), **out)

    rng = np.random.default_rng(2024)
    bi = (rng.uniform(-1e5, 1e5, (256, 4))).astype(np.float32)
    bj = (rng.uniform(-1e5, 1e5, (256, 4))).astype(np.float32)
    bj[:, 3] = rng.uniform(1e5, 1e9, 256).astype(np.float32)
    bj[:32, :3] = bi[:32, :3] + rng.uniform(-1, 1, (32, 3)).astype(np.float32)  # close pairs
    bj[32:36, :3] = bi[32:36, :3]                                              # coincident
    ai = rng.normal(0, 10, (256, 4)).astype(np.float32)
    res = np.stack([O.ref_pair(bi[k], bj[k], ai[k]) for k in range(256)])
    np.savez_compressed(os.path.join(OUT, "ref_pairs.npz"), bi=bi, bj=bj, ai=ai, out=res)

    # --- our oracle, Jacobi order, for the GPU-side checks -----------------------------------
    cases = {
        "jacobi_refinit_n1024_dt0.1": (nbody_amd.engine.seeded_bodies(1024, 0, 12345), 0.1, 0.002, (1, 10)),
        "jacobi_refinit_n1000_dt0.1": (nbody_amd.engine.seeded_bodies(1000, 0, 777), 0.1, 0.002, (1, 5)),
        "jacobi_plummer_n1024_dt0.01": (nbody_amd.engine.seeded_bodies(1024, 1, 12345), 0.01, 0.002, (1, 10, 100)),
    }
    for name, (x0, dt, eps2, Ks) in cases.items():
        out = {"x0": x0, "dt": np.float32(dt), "eps2": np.float32(eps2)}
        x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
        done = 0
        for K in Ks:
            O.step_jacobi(x, a, v, dt=dt, eps2=eps2, steps=K - done)
            done = K
            out[f"x_{K}"], out[f"v_{K}"], out[f"a_{K}"] = x.copy(), v.copy(), a.copy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()

"""The C++ host programs over the C-ABI: the headless driver (main.cpp's headless branch) and the
compareHostToDevice checker program (validation.cpp:55-103)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "n-bodysimulation_amd", "bin", "nbody_headless")
HARNESS = os.path.join(ROOT, "oracle", "compare_host_device")


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, **kw)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def _f4(path, n):
    return np.fromfile(path, np.float32).reshape(n, 4)


def test_headless_matches_python_host_layer_bitwise(nb, tmp_path):
    n, steps = 3000, 5
    out = _run([DRIVER, "--n", str(n), "--steps", str(steps), "--init", "ref", "--seed", "99", "--dump", str(tmp_path / "s")])
    assert "Starting the simulation..." in out and "Simulation complete" in out     # main.cpp:145,158
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n"] == n and line["steps"] == steps and line["pairs_per_s"] > 0
    sim = nb.engine.Simulation(nb.engine.seeded_bodies(n, 0, 99))                    # DT/EPS2 defaults of constants.h
    sim.run(steps)
    x, v, a = sim.state()
    assert np.array_equal(_f4(tmp_path / "s.x.f4", n), x)
    assert np.array_equal(_f4(tmp_path / "s.v.f4", n), v)
    assert np.array_equal(_f4(tmp_path / "s.a.f4", n), a)
    hdr = json.load(open(tmp_path / "s.json"))
    assert hdr["n"] == n and hdr["steps_done"] == steps


def test_headless_dump_load_resume_is_exact(tmp_path):
    n = 2048
    base = ["--n", str(n), "--init", "plummer", "--dt", "0.01", "--quiet"]
    _run([DRIVER, *base, "--steps", "5", "--dump", str(tmp_path / "full")])
    _run([DRIVER, *base, "--steps", "3", "--dump", str(tmp_path / "half")])
    _run([DRIVER, *base, "--steps", "2", "--load", str(tmp_path / "half"), "--dump", str(tmp_path / "resumed")])
    for ext in ("x", "v", "a"):
        assert np.array_equal(_f4(tmp_path / f"full.{ext}.f4", n), _f4(tmp_path / f"resumed.{ext}.f4", n)), ext
    assert json.load(open(tmp_path / "resumed.json"))["steps_done"] == 5


def test_headless_reference_loop_and_prompts(tmp_path):
    """--sync-each-step is the reference's loop (one synchronous simulate() per step); --interactive
    takes the reference's three stdin answers (main.cpp:163-228)."""
    n = 1500
    _run([DRIVER, "--n", str(n), "--steps", "4", "--init", "libc", "--quiet", "--dump", str(tmp_path / "q")])
    _run([DRIVER, "--n", str(n), "--steps", "4", "--init", "libc", "--quiet", "--sync-each-step", "--dump", str(tmp_path / "s")])
    assert np.array_equal(_f4(tmp_path / "q.x.f4", n), _f4(tmp_path / "s.x.f4", n))
    out = _run([DRIVER, "--n", str(n), "--init", "libc", "--interactive", "--dump", str(tmp_path / "i")], input="0\nn\n4\n")
    assert json.loads(out.strip().splitlines()[-1])["steps"] == 4
    assert np.array_equal(_f4(tmp_path / "q.x.f4", n), _f4(tmp_path / "i.x.f4", n))
    r = subprocess.run([DRIVER, "--interactive"], input="1\n", capture_output=True, text=True)
    assert r.returncode != 0 and "reduction" in r.stderr
    # body 0 of the libc init is the reference's
    x0 = _f4(tmp_path / "q.x.f4", n)
    assert x0[0, 3] == np.float32(798460160.0)


def test_headless_strict_kernel_matches_oracle(oracle, nb, tmp_path):
    n = 777
    _run([DRIVER, "--n", str(n), "--steps", "3", "--init", "ref", "--seed", "5", "--kernel", "strict", "--quiet", "--dump", str(tmp_path / "st")])
    x0 = nb.engine.seeded_bodies(n, 0, 5)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=3)
    assert np.array_equal(_f4(tmp_path / "st.x.f4", n), xo) and np.array_equal(_f4(tmp_path / "st.v.f4", n), vo)


def test_headless_sharded_path_over_rccl_and_f64(nb, oracle, tmp_path):
    """--shard: the C++ host drives nbody_shard_* with the RCCL communicator (threads = ranks; one here, the box has
    one GPU) and lands on the same bits as the plain run. --ngpu beyond the visible devices is refused.
    --precision f64: nbody_step_f64 from the C++ host, checked against the all-double checker."""
    import torch
    n = 20000
    base = ["--n", str(n), "--steps", "3", "--init", "plummer", "--dt", "0.01", "--seed", "4"]
    _run([DRIVER, *base, "--quiet", "--dump", str(tmp_path / "plain")])
    out = _run([DRIVER, *base, "--shard", "--dump", str(tmp_path / "shard")])
    assert json.loads(out.strip().splitlines()[-1])["ngpu"] == 1
    for ext in ("x", "v", "a"):
        assert np.array_equal(_f4(tmp_path / f"plain.{ext}.f4", n), _f4(tmp_path / f"shard.{ext}.f4", n)), ext
    # a sharded run resumes from a dump (positions AND velocities) exactly like the single-device one
    _run([DRIVER, *base[:2], "--steps", "2", "--init", "plummer", "--dt", "0.01", "--seed", "4", "--quiet", "--dump", str(tmp_path / "two")])
    _run([DRIVER, *base[:2], "--steps", "1", "--dt", "0.01", "--quiet", "--shard", "--load", str(tmp_path / "two"), "--dump", str(tmp_path / "three")])
    for ext in ("x", "v", "a"):
        assert np.array_equal(_f4(tmp_path / f"plain.{ext}.f4", n), _f4(tmp_path / f"three.{ext}.f4", n)), ext
    r = subprocess.run([DRIVER, "--n", "4096", "--ngpu", str(torch.cuda.device_count() + 1)], capture_output=True, text=True)
    assert r.returncode != 0 and "device(s) visible" in r.stderr
    n = 1500
    out = _run([DRIVER, "--n", str(n), "--steps", "3", "--init", "plummer", "--dt", "0.01", "--seed", "4", "--precision", "f64",
                "--dump", str(tmp_path / "d")])
    assert json.loads(out.strip().splitlines()[-1])["kernel"] == "f64"
    x64 = np.fromfile(tmp_path / "d.x.f8", np.float64).reshape(n, 4)
    xo = nb.engine.seeded_bodies(n, 1, 4).astype(np.float64)
    vo, ao = np.zeros_like(xo), np.zeros_like(xo)
    oracle.step_jacobi_f64(xo, ao, vo, dt=np.float32(0.01).item(), eps2=np.float32(0.002).item(), steps=3)
    assert np.abs(x64 - xo)[:, :3].max() <= 1e-13
    assert json.load(open(tmp_path / "d.json"))["dtype"] == "f64"


@pytest.mark.parametrize("ranks", [2, 3, 4, 8])
def test_headless_rank_threads_over_the_local_transport(nb, oracle, tmp_path, ranks):
    """nbody_headless --ngpu G with G > 1 — rank THREADS in one process, one context, one communicator and one shard each, the gates
    between them, the per-rank downloads and the merge — executed for real: --transport local moves the data with
    hipMemcpyPeerAsync pulls between the ranks (no RCCL, which refuses two ranks on one device) and --share-devices puts all ranks on
    this box's one GPU. STRICT kernel (canonical schedule): bit-identical to the single-device Jacobi oracle, padding bodies included
    (n is not a multiple of any rank count). FAST (symmetric schedule, J-side sums exchanged) and ONESIDED: within the fast tolerances
    of it, and bit-identical from one run to the next."""
    n, steps = 5003, 3
    base = ["--n", str(n), "--steps", str(steps), "--init", "plummer", "--dt", "0.01", "--seed", "21", "--quiet",
            "--ngpu", str(ranks), "--transport", "local", "--share-devices", "--timeout", "120"]
    x0 = nb.engine.seeded_bodies(n, 1, 21)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.01, eps2=0.002, steps=steps)
    _run([DRIVER, *base, "--kernel", "strict", "--dump", str(tmp_path / "st")])
    assert np.array_equal(_f4(tmp_path / "st.x.f4", n), xo) and np.array_equal(_f4(tmp_path / "st.v.f4", n), vo)
    assert np.array_equal(_f4(tmp_path / "st.a.f4", n), ao)
    for kernel in ("fast", "onesided"):
        _run([DRIVER, *base, "--kernel", kernel, "--dump", str(tmp_path / kernel)])
        x, a = _f4(tmp_path / f"{kernel}.x.f4", n), _f4(tmp_path / f"{kernel}.a.f4", n)
        assert np.abs(x - xo)[:, :3].max() <= 1e-6
        assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
    _run([DRIVER, *base, "--kernel", "fast", "--dump", str(tmp_path / "again")])
    for ext in ("x", "v", "a"):
        assert np.array_equal(_f4(tmp_path / f"fast.{ext}.f4", n), _f4(tmp_path / f"again.{ext}.f4", n)), ext


def test_headless_local_transport_at_the_block_sizes_a_node_runs(nb, tmp_path):
    """Four rank threads x 65536 bodies (the default block shapes of both the own-block pass and the cross launches), two steps, so
    that the all-gather of ADVANCED positions is exercised too: against the single-GPU run of the same system."""
    n = 262144
    base = ["--n", str(n), "--steps", "2", "--init", "plummer", "--dt", "0.01", "--seed", "5", "--quiet"]
    _run([DRIVER, *base, "--dump", str(tmp_path / "one")])
    out = _run([DRIVER, *base[:-1], "--ngpu", "4", "--transport", "local", "--share-devices", "--dump", str(tmp_path / "four")])
    assert json.loads(out.strip().splitlines()[-1])["ngpu"] == 4
    x1, a1 = _f4(tmp_path / "one.x.f4", n), _f4(tmp_path / "one.a.f4", n)
    x4, a4 = _f4(tmp_path / "four.x.f4", n), _f4(tmp_path / "four.a.f4", n)
    assert np.abs(x4 - x1)[:, :3].max() <= 1e-6
    assert np.abs(a4 - a1)[:, :3].max() / np.abs(a1[:, :3]).max() <= 2e-5
    # misuse is refused with a message
    r = subprocess.run([DRIVER, "--n", "4096", "--ngpu", "2", "--share-devices"], capture_output=True, text=True)
    assert r.returncode != 0 and "--transport local" in r.stderr


def test_in_place_step_with_the_gpu_shared_between_processes(tmp_path):
    """What the in-place fused step's protocol is for: it never waits for a workgroup that has not started, so nothing deadlocks when
    the 256 workgroups of a launch are NOT all resident — here three processes step the reference's loop (one synchronous simulate()
    per step, N = 8192: one workgroup per CU each) on the one GPU at the same time, and a fourth keeps it busy with queued steps of a
    large system. Every run must finish, and every result must be bit-identical to the same run with the GPU to itself — whether a
    wave wrote in place or through the spare array (`inplace_fallback_waves` says how many did)."""
    n, steps = 8192, 1500
    base = [DRIVER, "--n", str(n), "--steps", str(steps), "--init", "libc", "--sync-each-step"]
    alone = json.loads(_run([*base, "--dump", str(tmp_path / "alone")]).strip().splitlines()[-1])
    assert alone["steps"] == steps and alone["inplace_fallback_waves"] >= 0
    hog = subprocess.Popen([DRIVER, "--n", "131072", "--steps", "1500", "--init", "plummer", "--dt", "0.01", "--quiet"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    procs = [subprocess.Popen([*base, "--dump", str(tmp_path / f"shared{k}")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in range(3)]
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=240)
            assert p.returncode == 0, err
            outs.append(json.loads(out.strip().splitlines()[-1]))
        hog.communicate(timeout=240)
        assert hog.returncode == 0
    finally:
        for p in procs + [hog]:
            if p.poll() is None:
                p.kill()
    # nothing is measured by default (opt-in: NBODY_AUTOTUNE), so no timing under contention can pick another decomposition
    assert alone["autotuned_choice"] == -1 and [o["autotuned_choice"] for o in outs] == [-1, -1, -1], (alone, outs)
    for k in range(3):
        for ext in ("x", "v", "a"):
            assert np.array_equal(_f4(tmp_path / f"alone.{ext}.f4", n), _f4(tmp_path / f"shared{k}.{ext}.f4", n)), (k, ext)
    print("in-place fall-back waves: alone", alone["inplace_fallback_waves"], "shared", [o["inplace_fallback_waves"] for o in outs],
          "us/step alone %.1f shared %s" % (alone["seconds"] / steps * 1e6, ["%.1f" % (o["seconds"] / steps * 1e6) for o in outs]))


def test_headless_says_what_clock_it_ran_at():
    """nbody_headless --clock: the C++ driver's line read against the clock it was measured at (nbody_ctx_timing(ctx, 2) /
    nbody_ctx_clock_read through the C-ABI from a g++-built program): one record per force launch, cycles / sclk = the launch's duration."""
    out = _run([DRIVER, "--n", "65536", "--steps", "12", "--init", "plummer", "--dt", "0.01", "--clock"])
    lines = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 2 and "clock" in lines[0] and lines[1]["steps"] == 12
    c = lines[0]["clock"]
    assert c["launches"] == 12 and 800 < c["sclk_mhz_under_load"] < 2600
    assert c["sclk_mhz_slowest_xcd"] <= c["sclk_mhz_under_load"] <= c["sclk_mhz_fastest_xcd"]
    assert abs(c["kernel_cycles_per_launch"] / (c["sclk_mhz_under_load"] * 1e3) - c["ms_per_launch_by_device_clock"]) < 1e-4
    assert c["force_kernel_ms_per_launch"] <= c["ms_per_launch_by_device_clock"] * 1.002 < c["force_kernel_ms_per_launch"] + 0.06


def test_in_place_block_sums_with_the_gpu_shared_between_processes(tmp_path):
    """The ticket protocol of nbk::force_sym_ticket waits — for EARLIER tasks of the same launch only. With the GPU to itself every
    workgroup's predecessor is resident or done; here three processes run the in-place step at the same time on the one GPU while a
    fourth keeps it full, so a launch's workgroups start late, in bursts, or get their queue switched out as a whole. Every run must
    finish (no wait may time out: the driver would report the error) and every result must be bit-identical to the same run alone —
    the tickets fix the order of the additions, not the scheduling."""
    n, steps = 200000, 6                        # 79 blocks of 2560: 3160 block-pair tasks per launch
    base = [DRIVER, "--n", str(n), "--steps", str(steps), "--init", "plummer", "--dt", "0.01", "--no-equal-mass", "--inplace-sums", "on"]
    alone = json.loads(_run([*base, "--dump", str(tmp_path / "alone")]).strip().splitlines()[-1])
    assert alone["steps"] == steps
    slabs = json.loads(_run([DRIVER, "--n", str(n), "--steps", str(steps), "--init", "plummer", "--dt", "0.01", "--no-equal-mass", "--inplace-sums", "off",
                             "--dump", str(tmp_path / "slabs")]).strip().splitlines()[-1])
    assert slabs["steps"] == steps
    xa, xs = _f4(tmp_path / "alone.x.f4", n), _f4(tmp_path / "slabs.x.f4", n)
    assert not np.array_equal(_f4(tmp_path / "alone.a.f4", n), _f4(tmp_path / "slabs.a.f4", n))      # another (fixed) order of the same block sums ...
    assert np.abs(xa - xs)[:, :3].max() <= 1e-6                                                       # ... within rounding of the slab kernel
    hog = subprocess.Popen([DRIVER, "--n", "131072", "--steps", "400", "--init", "plummer", "--dt", "0.01", "--quiet"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    procs = [subprocess.Popen([*base, "--dump", str(tmp_path / f"shared{k}")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in range(3)]
    try:
        for p in procs:
            out, err = p.communicate(timeout=240)
            assert p.returncode == 0, err
            assert json.loads(out.strip().splitlines()[-1])["steps"] == steps
        hog.communicate(timeout=240)
        assert hog.returncode == 0
    finally:
        for p in procs + [hog]:
            if p.poll() is None:
                p.kill()
    for k in range(3):
        for ext in ("x", "v", "a"):
            assert np.array_equal(_f4(tmp_path / f"alone.{ext}.f4", n), _f4(tmp_path / f"shared{k}.{ext}.f4", n)), (k, ext)


def test_compare_host_to_device_program():
    """compareHostToDevice in the reference's own terms: lock-step GPU/CPU steps, then the 1 % rule on
    positions, velocities and accelerations."""
    out = _run([HARNESS, "--n", "1024", "--steps", "10"])
    assert "Starting verification..." in out and "Verification complete" in out      # validation.cpp:83,87
    r = json.loads(out.strip().splitlines()[-1])
    assert r["bad_positions"] == 0 and r["cpu_order"] == "inplace"
    out = _run([HARNESS, "--n", "1024", "--steps", "10", "--jacobi"])
    r = json.loads(out.strip().splitlines()[-1])
    assert r["bad_positions"] == 0 and r["bad_velocities"] <= 2 and r["bad_accelerations"] <= 2


def test_source_level_dropin_with_reference_header_names(nb, tmp_path):
    """tests/dropin_main.cpp is written against "constants.h", "kernel.cuh", "utils.h", "validation.h"
    like the reference's main.cpp; it compiles with -Iinclude/compat, links libnbody_hip.so and agrees
    with the Python host layer bit for bit (N_BODIES = 8192, libc-rand init, DT/EPS2 of constants.h)."""
    exe = str(tmp_path / "dropin_main")
    libdir = os.path.join(ROOT, "n-bodysimulation_amd")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include", "compat"), "-o", exe,
                        os.path.join(ROOT, "tests", "dropin_main.cpp"), "-L" + libdir, "-lnbody_hip", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # (simulate() measures nothing unless NBODY_AUTOTUNE is set: the reference's loop gets the built-in decomposition's bits on every box)
    out = _run([exe, "4"], env={k: v for k, v in os.environ.items() if k != "NBODY_AUTOTUNE"})
    assert "Starting the simulation..." in out and "Simulation complete" in out
    # the rest of utils.h / validation.h, called once each by the same translation unit
    assert "== Device Properties ==" in out and "Warp size: 64" in out
    assert "verify_equality3: 1 of 4 bodies differ" in out and "helpers: off3=1 copy_ok=1 body1=" in out
    last = out.strip().splitlines()[-1]
    assert "N_BODIES=8192" in last and "DT=0.1" in last
    import ctypes
    ctypes.CDLL(None).srand(1)
    sim = nb.engine.Simulation(nb.engine.libc_random_bodies(8192))
    sim.run(4)
    x, _, _ = sim.state()
    body0 = [float(t) for t in last.split("body0=")[1].split()]
    assert np.array_equal(np.array(body0, np.float32), x[0])

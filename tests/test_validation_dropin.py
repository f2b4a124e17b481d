"""The reference's validation.h, all six declarations, at source level.

include/compat/validation.h carries verify_equality4 / verify_equality3 / verify_still_bodies (validation.h:6-8);
oracle/validation_checker.hpp carries bodyInteractions_CPU / CPU_compute / compareHostToDevice (validation.h:3-5) — the CPU
checker, kept out of the product on purpose. tests/validation_dropin.cpp includes both under the reference's header names
and calls all six with the reference's signatures."""
import os
import subprocess

import numpy as np
import pytest

from conftest import same_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, n_bodies, extra=()):
    exe = str(tmp_path / "validation_dropin")
    libdir = os.path.join(ROOT, "n-bodysimulation_amd")
    odir = os.path.join(ROOT, "oracle")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", f"-DN_BODIES={n_bodies}", *extra,
                        "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"), "-I" + odir,
                        "-o", exe, os.path.join(ROOT, "tests", "validation_dropin.cpp"),
                        "-L" + libdir, "-lnbody_hip", "-L" + odir, "-loracle", "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_validation_h_checker_half_compiles_and_matches_the_oracle(nb, oracle, tmp_path):
    """No device: bodyInteractions_CPU and CPU_compute (reference signatures) are the pinned restatement — three in-place
    steps written by the C++ caller equal oracle.step_inplace bit for bit; the verify_* counts are the reference's rules."""
    n = 300
    exe = _build(tmp_path, n)
    out_file = str(tmp_path / "state.bin")
    r = subprocess.run([exe, "--cpu-only", out_file], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "counts 1 1 1" in r.stdout
    x0 = nb.engine.seeded_bodies(n, 0, 4242)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_inplace(xo, ao, vo, steps=3)                       # DT 0.1f / EPS2 0.002f: constants.h:25-26
    got = np.fromfile(out_file, np.float32).reshape(3, n, 4)
    assert same_bits(got[0], xo) and same_bits(got[1], vo) and same_bits(got[2], ao)
    pair = [float.fromhex(t) for t in r.stdout.split("pair ")[1].split()[:4]]
    want = oracle.accel_range(x0, 0, 1, 1, 2, eps2=0.002)[0]       # body 0 <- body 1, the single pair term
    assert np.array_equal(np.array(pair[:3], np.float32), want[:3])


@pytest.mark.gpu
def test_compare_host_to_device_with_the_reference_signature(nb, tmp_path):
    """compareHostToDevice(float4* x6): N_BODIES bodies and 1000 lock-step steps compiled in, as validation.cpp:59,65."""
    exe = _build(tmp_path, 512)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Starting verification..." in r.stdout and "Verification complete" in r.stdout and "compareHostToDevice rc 0" in r.stdout
    assert r.stdout.count("verify_still_bodies:") == 3
    # a short lock-step run stays inside the reference's own 1 % rule on positions
    exe = _build(tmp_path, 512, ["-DNBODY_COMPARE_STEPS=5"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "verify_still_bodies: 0 of 512 bodies outside 1 %" in r.stdout.split("Starting verification...")[1].splitlines()[1]

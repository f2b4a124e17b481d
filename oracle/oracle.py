"""ctypes front-end of the CPU checker (TEST INFRASTRUCTURE ONLY).

Two libraries, both built by ``make -C oracle``:

* ``liboracle.so``        — our C restatement (``nbody_oracle.c``), functions cite the
  reference lines they follow.
* ``_ref/libref_cpu.so``  — the reference's own ``CPU_compute`` & friends compiled from
  ``/root/reference/TestProject/{validation,utils}.cpp`` where they lie (no copy).  Exists
  only where it was built (this container; the prebuilt file travels with gpurun).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product (``n-bodysimulation_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_cpu.so")
REF_DT001_SO = os.path.join(_HERE, "_ref", "libref_cpu_dt001.so")   # the same sources with DT 0.01f (ref_dt001.cpp)

# constants.h:25-26
REF_EPS2 = np.float32(0.002)
REF_DT = np.float32(0.1)


class _F4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


def build(quiet: bool = True) -> None:
    """(Re)build the checker libraries; a no-op when they are up to date."""
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True)


_lib = None
_ref = None
_ref_dt001 = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        p = C.c_void_p
        L.oracle_step_inplace.argtypes = [p, p, p, C.c_int, C.c_float, C.c_float]
        L.oracle_step_jacobi.argtypes = [p, p, p, C.c_int, C.c_float, C.c_float]
        L.oracle_step_jacobi_f64acc.argtypes = [p, p, p, C.c_int, C.c_float, C.c_float]
        L.oracle_step_jacobi_f64.argtypes = [p, p, p, C.c_int, C.c_double, C.c_double]
        L.oracle_accel_range_f64.argtypes = [p, p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        L.oracle_step_legacy.argtypes = [p, p, C.c_int]
        L.oracle_accel_range.argtypes = [p, p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        L.oracle_integrate.argtypes = [p, p, p, C.c_int, C.c_float]
        L.oracle_fill_with_random4.argtypes = [p, C.c_int]
        L.oracle_fill_with_zeroes4.argtypes = [p, C.c_int]
        L.oracle_verify_still_bodies.argtypes = [p, p, C.c_int]
        L.oracle_verify_still_bodies.restype = C.c_int
        L.oracle_verify_equality4.argtypes = [p, p, C.c_int]
        L.oracle_verify_equality4.restype = C.c_int
        L.oracle_pair.argtypes = [_F4, _F4, _F4, C.c_float]
        L.oracle_pair.restype = _F4
        L.oracle_max_threads.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def have_ref_dt001() -> bool:
    return os.path.exists(REF_DT001_SO)


def _bind_ref(path: str) -> C.CDLL:
    if True:
        R = C.CDLL(path)
        p = C.c_void_p
        R.CPU_compute = R._Z11CPU_computeP6float4S0_S0_i
        R.CPU_compute.argtypes = [p, p, p, C.c_int]
        R.CPU_compute.restype = None
        R.bodyInteractions_CPU = R._Z20bodyInteractions_CPU6float4S_S_
        R.bodyInteractions_CPU.argtypes = [_F4, _F4, _F4]
        R.bodyInteractions_CPU.restype = _F4
        R.fill_with_random4 = R._Z17fill_with_random4P6float4i
        R.fill_with_random4.argtypes = [p, C.c_int]
        R.fill_with_random4.restype = None
        R.fill_with_zeroes4 = R._Z17fill_with_zeroes4P6float4i
        R.fill_with_zeroes4.argtypes = [p, C.c_int]
        R.fill_with_zeroes4.restype = None
        R.verify_still_bodies = R._Z19verify_still_bodiesP6float4S0_i
        R.verify_still_bodies.argtypes = [p, p, C.c_int]
        R.verify_still_bodies.restype = None
    return R


def ref() -> C.CDLL:
    """The reference's own CPU objects (C++ linkage, hence the mangled names)."""
    global _ref
    if _ref is None:
        _ref = _bind_ref(REF_SO)
    return _ref


def ref_dt001() -> C.CDLL:
    """The reference's CPU objects compiled with DT 0.01f (oracle/ref_dt001.cpp)."""
    global _ref_dt001
    if _ref_dt001 is None:
        _ref_dt001 = _bind_ref(REF_DT001_SO)
    return _ref_dt001


def _chk(a: np.ndarray, dtype=np.float32) -> np.ndarray:
    assert a.dtype == dtype and a.ndim == 2 and a.shape[1] == 4 and a.flags.c_contiguous, (
        a.dtype, a.shape)
    return a


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def step_inplace(X, A, V, dt=REF_DT, eps2=REF_EPS2, steps=1):
    """validation.cpp:28-52, literal (sequential, in place). Arrays are updated in place."""
    _chk(X), _chk(A), _chk(V)
    for _ in range(steps):
        lib().oracle_step_inplace(_ptr(X), _ptr(A), _ptr(V), len(X), float(dt), float(eps2))


def step_jacobi(X, A, V, dt=REF_DT, eps2=REF_EPS2, steps=1, f64acc=False):
    _chk(X), _chk(A), _chk(V)
    fn = lib().oracle_step_jacobi_f64acc if f64acc else lib().oracle_step_jacobi
    for _ in range(steps):
        fn(_ptr(X), _ptr(A), _ptr(V), len(X), float(dt), float(eps2))


def step_jacobi_f64(X, A, V, dt, eps2, steps=1):
    _chk(X, np.float64), _chk(A, np.float64), _chk(V, np.float64)
    for _ in range(steps):
        lib().oracle_step_jacobi_f64(_ptr(X), _ptr(A), _ptr(V), len(X), float(dt), float(eps2))


def step_legacy(X, V3, steps=1):
    """The older snapshot's step (float3 velocity, double-literal DT=0.01/EPS2=0.002), Jacobi order."""
    _chk(X)
    assert V3.dtype == np.float32 and V3.shape == (len(X), 3) and V3.flags.c_contiguous
    for _ in range(steps):
        lib().oracle_step_legacy(_ptr(X), _ptr(V3), len(X))


def accel_range(X, i0, i1, j0=0, j1=None, eps2=REF_EPS2, f64acc=False):
    """Accelerations of targets [i0,i1) from sources [j0,j1); returns (i1-i0,4) float32."""
    _chk(X)
    j1 = len(X) if j1 is None else j1
    out = np.zeros((i1 - i0, 4), np.float32)
    lib().oracle_accel_range(_ptr(X), _ptr(out), i0, i1, j0, j1, float(eps2), 1 if f64acc else 0)
    return out


def accel_range_f64(X, i0, i1, j0=0, j1=None, eps2=REF_EPS2):
    """All-double accelerations of targets [i0,i1) from sources [j0,j1); returns (i1-i0,4) float64."""
    _chk(X, np.float64)
    j1 = len(X) if j1 is None else j1
    out = np.zeros((i1 - i0, 4), np.float64)
    lib().oracle_accel_range_f64(_ptr(X), _ptr(out), i0, i1, j0, j1, float(eps2))
    return out


def integrate(X, V, A, dt=REF_DT):
    _chk(X), _chk(V), _chk(A)
    lib().oracle_integrate(_ptr(X), _ptr(V), _ptr(A), len(X), float(dt))


def pair(bi, bj, ai, eps2=REF_EPS2):
    r = lib().oracle_pair(_F4(*map(float, bi)), _F4(*map(float, bj)), _F4(*map(float, ai)), float(eps2))
    return np.array([r.x, r.y, r.z, r.w], np.float32)


def verify_still_bodies(v, x) -> int:
    _chk(v), _chk(x)
    return lib().oracle_verify_still_bodies(_ptr(v), _ptr(x), len(v))


def verify_equality4(v, x) -> int:
    _chk(v), _chk(x)
    return lib().oracle_verify_equality4(_ptr(v), _ptr(x), len(v))


def fill_with_random4_libc(n: int) -> np.ndarray:
    """utils.cpp:30-37 through this process's libc rand() state (unseeded on first use)."""
    out = np.zeros((n, 4), np.float32)
    lib().oracle_fill_with_random4(_ptr(out), n)
    return out


def max_threads() -> int:
    return lib().oracle_max_threads()


def set_threads(t: int) -> None:
    lib().oracle_set_threads(t)


# ---- the reference's own objects (only where oracle/_ref was built) -----------------------

def ref_step(X, A, V, steps=1):
    """Reference CPU_compute (validation.cpp:28-52) with its compiled-in DT=0.1f, EPS2=0.002f."""
    _chk(X), _chk(A), _chk(V)
    for _ in range(steps):
        ref().CPU_compute(_ptr(X), _ptr(A), _ptr(V), len(X))


def ref_step_dt001(X, A, V, steps=1):
    """Reference CPU_compute (validation.cpp:28-52) compiled with DT=0.01f (EPS2=0.002f as shipped)."""
    _chk(X), _chk(A), _chk(V)
    for _ in range(steps):
        ref_dt001().CPU_compute(_ptr(X), _ptr(A), _ptr(V), len(X))


def ref_pair(bi, bj, ai):
    r = ref().bodyInteractions_CPU(_F4(*map(float, bi)), _F4(*map(float, bj)), _F4(*map(float, ai)))
    return np.array([r.x, r.y, r.z, r.w], np.float32)


def ref_fill_with_random4(n: int) -> np.ndarray:
    out = np.zeros((n, 4), np.float32)
    ref().fill_with_random4(_ptr(out), n)
    return out

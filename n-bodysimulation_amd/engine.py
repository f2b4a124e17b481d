"""Host side of the all-pairs step: the Python mirror of the reference's headless driver.

The reference's sequence (TestProject/main.cpp:231-368, headless branch):
    allocate pinned host arrays -> fill_with_random4 / fill_with_zeroes4 -> cudaMalloc + H2D
    -> simulationLoopNoVisual: `steps` x simulate(d_bodies, d_accel, d_vel, N) -> free.
``Simulation`` reproduces that sequence on top of the C-ABI (``include/nbody.h``); device memory
and streams come from PyTorch-ROCm, which is plumbing only — every force/integrate runs in
``libnbody_hip.so``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import KERNEL_FAST, KERNEL_STRICT, check

INIT_REFERENCE = 0  # uniform cube +-1e5, masses U(1e5,1e9): utils.cpp:30-37, constants.h:15-19
INIT_PLUMMER = 1    # Plummer sphere a=1, M=1, cold


def seeded_bodies(n: int, init: int = INIT_REFERENCE, seed: int = 12345) -> np.ndarray:
    """Reproducible initial bodies as an (n,4) float32 array {x,y,z,mass} (host-side helper of
    the C library; runs without a GPU)."""
    out = np.zeros((n, 4), np.float32)
    check(_lib.load().nbody_fill_seeded(C.c_void_p(out.ctypes.data), n, init, seed))
    return out


def libc_random_bodies(n: int) -> np.ndarray:
    """utils.cpp:30-37 literally: libc rand(), unseeded unless the process seeded it."""
    out = np.zeros((n, 4), np.float32)
    _lib.load().nbody_fill_with_random4(C.c_void_p(out.ctypes.data), n)
    return out


def verify_still_bodies(v: np.ndarray, x: np.ndarray) -> int:
    """validation.cpp:143-164 as a count of offending bodies."""
    v = np.ascontiguousarray(v, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    return _lib.load().nbody_verify_still_bodies(C.c_void_p(v.ctypes.data), C.c_void_p(x.ctypes.data), len(v))


def verify_equality4(v: np.ndarray, x: np.ndarray) -> int:
    """validation.cpp:106-122 as a count."""
    v = np.ascontiguousarray(v, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    return _lib.load().nbody_verify_equality4(C.c_void_p(v.ctypes.data), C.c_void_p(x.ctypes.data), len(v))


def _dptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _check_f4(t: torch.Tensor, n: Optional[int] = None, dtype=torch.float32) -> None:
    if not (t.is_cuda and t.dtype == dtype and t.dim() == 2 and t.shape[1] == 4 and t.is_contiguous()):
        raise ValueError(f"expected a contiguous CUDA (n,4) {dtype} tensor, got {tuple(t.shape)} {t.dtype} {t.device}")
    if n is not None and t.shape[0] != n:
        raise ValueError(f"expected {n} bodies, got {t.shape[0]}")


class Context:
    """Owns an ``nbody_ctx``: device, stream, dt/eps2, kernel choice, slab workspace."""

    def __init__(self, device: Optional[int] = None, dt: float = _lib.DEFAULT_DT, eps2: float = _lib.DEFAULT_EPS2,
                 kernel: int = KERNEL_FAST, tile: int = 0, bodies_per_lane: int = 0, jsplit: int = 0,
                 stream: Optional[torch.cuda.Stream] = None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.device = int(device)
        check(self._lib.nbody_ctx_create(C.byref(self._h), self.device))
        self.set_params(dt, eps2)
        self.set_kernel(kernel, tile, bodies_per_lane, jsplit)
        self._stream = None
        if stream is not None:
            self.set_stream(stream)

    def set_params(self, dt: float, eps2: float) -> None:
        check(self._lib.nbody_ctx_set_params(self._h, dt, eps2))
        self.dt, self.eps2 = float(dt), float(eps2)

    def set_kernel(self, kernel: int = KERNEL_FAST, tile: int = 0, bodies_per_lane: int = 0, jsplit: int = 0) -> None:
        check(self._lib.nbody_ctx_set_kernel(self._h, kernel, tile, bodies_per_lane, jsplit))
        self.kernel = kernel

    def set_symmetric_shape(self, waves: int = 0, bodies_per_lane: int = 0) -> None:
        """Block shape of the symmetric kernel (64*waves*bodies_per_lane bodies per block); 0 = auto."""
        check(self._lib.nbody_ctx_set_symmetric_shape(self._h, waves, bodies_per_lane))

    def set_symmetric_runs(self, mode: int) -> None:
        """Run-based decompositions of the symmetric kernel: -1 where measurements prefer them, 0 never, 1 unit runs always,
        2 balanced (step-granular) runs always."""
        check(self._lib.nbody_ctx_set_symmetric_runs(self._h, mode))

    def autotune(self, x: torch.Tensor, steps_per_trial: int = 50) -> dict:
        """Times every decomposition that applies to whole steps of len(x) bodies on this device (scratch copies, dt = 0) and leaves
        the context on the fastest. Returns {"choice": id, "us_per_step": t} (ids: nbody_ctx_autotune in nbody.h)."""
        _check_f4(x)
        choice, us = C.c_int(), C.c_double()
        check(self._lib.nbody_ctx_autotune(self._h, _dptr(x), x.shape[0], steps_per_trial, C.byref(choice), C.byref(us)))
        return {"choice": choice.value, "us_per_step": us.value}

    def set_fused(self, mode: int) -> None:
        """The fused small-N step (force + integrate in one launch): -1 where measurements prefer it, 0 never, 1 always (FAST)."""
        check(self._lib.nbody_ctx_set_fused(self._h, mode))

    def set_fused_inplace(self, mode: int) -> None:
        """The fused step in place: -1 the odd last step of a call (default), 0 never (two arrays + copy-back), 1 every fused
        step, 2 every step with the fall-back path forced (tests)."""
        check(self._lib.nbody_ctx_set_fused_inplace(self._h, mode))

    def fused_inplace_fallbacks(self) -> int:
        """Waves that took the in-place step's fall-back path so far (synchronises)."""
        out = C.c_ulonglong()
        check(self._lib.nbody_ctx_fused_inplace_stats(self._h, C.byref(out)))
        return out.value

    def set_equal_mass(self, mode: int) -> None:
        """Equal-mass path of the symmetric kernels (decided on the device, per launch, by a scan of the masses): -1 launches of
        32768 bodies or more, 1 launches of 4096 bodies or more, 0 never."""
        check(self._lib.nbody_ctx_set_equal_mass(self._h, mode))

    def equal_mass_verdict(self) -> dict:
        """What the last device-side mass scan found: {"scanned": bool, "uniform": bool, "mass": m0}. Synchronises the stream."""
        sc, un, m = C.c_int(), C.c_int(), C.c_float()
        check(self._lib.nbody_ctx_equal_mass_verdict(self._h, C.byref(sc), C.byref(un), C.byref(m)))
        return {"scanned": bool(sc.value), "uniform": bool(un.value), "mass": m.value}

    def set_workspace_limit(self, nbytes: int = 0, fail_above: bool = False) -> None:
        """Cap on one partial-sum workspace (0 = automatic: min(96 GiB, half of the free device memory)). Shapes that need
        more are not chosen; the step falls back towards the one-sided kernel. fail_above=True is the test hook of nbody.h."""
        check(self._lib.nbody_ctx_set_workspace_limit(self._h, nbytes, 1 if fail_above else 0))

    def set_inplace_sums(self, mode: int = -1) -> None:
        """Block pairs with the partial sums added in place (no slab workspace): -1 where the slab workspace does not fit the cap,
        1 wherever unit runs / block pairs would run, 0 never."""
        check(self._lib.nbody_ctx_set_inplace_sums(self._h, mode))

    def set_stream(self, stream: Optional[torch.cuda.Stream]) -> None:
        self._stream = stream  # keep it alive
        check(self._lib.nbody_ctx_set_stream(self._h, C.c_void_p(stream.cuda_stream) if stream is not None else None))

    def set_graph(self, mode: int) -> None:
        """hipGraph replay of 32-step chains in nbody_step: -1 auto (small N), 0 off, 1 on."""
        check(self._lib.nbody_ctx_set_graph(self._h, mode))

    def reserve(self, n_targets: int) -> None:
        check(self._lib.nbody_ctx_reserve(self._h, n_targets))

    def launch_info(self, n_targets: int, n_sources: int) -> dict:
        js, bl, lds = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.nbody_ctx_launch_info(self._h, n_targets, n_sources, C.byref(js), C.byref(bl), C.byref(lds)))
        return {"jsplit": js.value, "blocks": bl.value, "lds_bytes": lds.value}

    def step_info(self, n: int) -> dict:
        """What a whole step of n bodies launches (symmetric or one-sided kernel, slabs, pair evaluations)."""
        return self._info(self._lib.nbody_ctx_step_info, n)

    def _info(self, fn, *args) -> dict:
        sym, blk, slabs, wgs, ev = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
        check(fn(self._h, *args, C.byref(sym), C.byref(blk), C.byref(slabs), C.byref(wgs), C.byref(ev)))
        return {"symmetric": sym.value > 0, "runs": sym.value == 2, "balanced": sym.value == 3, "fused": sym.value == -1, "ticket": sym.value == 4,
                "block_bodies": blk.value, "slabs": slabs.value, "workgroups": wgs.value, "evaluated_pairs": ev.value}

    def step_info_f64(self, n: int) -> dict:
        """What nbody_step_f64 launches for n bodies."""
        return self._info(self._lib.nbody_ctx_step_info_f64, n)

    def square_info(self, n: int, nparts: int = 1) -> dict:
        """What nbody_accel_square_part(.., nparts) launches for a block of n bodies against itself (one rank's own-block
        pass of the sharded step: nparts = 2 when world > 1)."""
        return self._info(self._lib.nbody_ctx_square_info, n, nparts)

    def step(self, x: torch.Tensor, a: torch.Tensor, v: torch.Tensor, steps: int = 1) -> None:
        n = x.shape[0]
        _check_f4(x), _check_f4(a, n), _check_f4(v, n)
        check(self._lib.nbody_step(self._h, _dptr(x), _dptr(a), _dptr(v), n, steps))

    def step_f64(self, x: torch.Tensor, a: torch.Tensor, v: torch.Tensor, dt: float, eps2: float, steps: int = 1) -> None:
        n = x.shape[0]
        _check_f4(x, None, torch.float64), _check_f4(a, n, torch.float64), _check_f4(v, n, torch.float64)
        check(self._lib.nbody_step_f64(self._h, _dptr(x), _dptr(a), _dptr(v), n, steps, dt, eps2))

    def accel_range(self, x: torch.Tensor, a_out: torch.Tensor, i0: int, i1: int, j0: int, j1: int,
                    accumulate: bool = False) -> None:
        _check_f4(x), _check_f4(a_out, i1 - i0)
        if i1 > x.shape[0] or j1 > x.shape[0]:
            raise ValueError("range beyond the body array")
        check(self._lib.nbody_accel_range(self._h, _dptr(x), _dptr(a_out), i0, i1, j0, j1, 1 if accumulate else 0))

    def accel_wrapped(self, x: torch.Tensor, a_out: torch.Tensor, i0: int, i1: int, j0: int, count: int,
                      accumulate: bool = False) -> None:
        """Sources j0 .. j0+count-1 modulo len(x): one launch for a source run that wraps around."""
        _check_f4(x), _check_f4(a_out, i1 - i0)
        check(self._lib.nbody_accel_wrapped(self._h, _dptr(x), x.shape[0], _dptr(a_out), i0, i1, j0, count,
                                            1 if accumulate else 0))

    def accel_cross(self, x: torch.Tensor, a_i: torch.Tensor, i0: int, i1: int, accumulate_i: bool, j0: int, count: int,
                    a_j_out: torch.Tensor) -> None:
        """Symmetric evaluation of targets [i0,i1) x the source run j0..j0+count-1 (mod len(x)), disjoint sets:
        a_i (+)= what the sources do to the targets, a_j_out = what the targets do to the sources."""
        _check_f4(x), _check_f4(a_i, i1 - i0), _check_f4(a_j_out, count)
        check(self._lib.nbody_accel_cross(self._h, _dptr(x), x.shape[0], _dptr(a_i), i0, i1, 1 if accumulate_i else 0,
                                          j0, count, _dptr(a_j_out)))

    def integrate_range(self, x: torch.Tensor, v_own: torch.Tensor, a_own: torch.Tensor, i0: int, i1: int) -> None:
        _check_f4(x), _check_f4(v_own, i1 - i0), _check_f4(a_own, i1 - i0)
        if i1 > x.shape[0]:
            raise ValueError("range beyond the body array")
        check(self._lib.nbody_integrate_range(self._h, _dptr(x), _dptr(v_own), _dptr(a_own), i0, i1))

    def sync(self) -> None:
        check(self._lib.nbody_ctx_sync(self._h))

    def timing(self, enable: bool, clock: bool = False) -> None:
        """Bracket every force launch with hipEvents on the launch stream (bench.py's roofline); clock=True adds a clock-stamp
        launch on either side of the pair (shader cycles and 100-MHz ticks per XCD: clock_read)."""
        check(self._lib.nbody_ctx_timing(self._h, (2 if clock else 1) if enable else 0))

    def clock_read(self) -> dict:
        """What the clock stamps around the timed force launches since the last read say: shader cycles per launch, the shader
        clock held under that load (MHz; overall, slowest and fastest XCD), the launch's duration by the 100-MHz counter."""
        r = _lib.ClockReport()
        check(self._lib.nbody_ctx_clock_read(self._h, C.byref(r)))
        return {k: getattr(r, k) for k, _ in r._fields_}

    def timing_read(self):
        """(summed force-kernel ms, launches) since the last read; synchronises the stream."""
        ms, k = C.c_double(), C.c_int()
        check(self._lib.nbody_ctx_timing_read(self._h, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def close(self) -> None:
        if self._h:
            self._lib.nbody_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def simulate(d_bodies: torch.Tensor, d_accelerations: torch.Tensor, d_velocity: torch.Tensor, n: Optional[int] = None) -> None:
    """``simulate(d_bodies, d_accelerations, d_velocity, N)`` of TestProject/kernel.cuh:2 —
    one synchronous in-place step with DT / EPS2 of constants.h:25-26 on the current device."""
    n = d_bodies.shape[0] if n is None else n
    _check_f4(d_bodies), _check_f4(d_accelerations), _check_f4(d_velocity)
    if n > d_bodies.shape[0]:
        raise ValueError("N exceeds the allocation")
    check(_lib.load().nbody_simulate(_dptr(d_bodies), _dptr(d_accelerations), _dptr(d_velocity), n))


def simulate_autotuned(n: int) -> dict:
    """What simulate() — the default context — found when it measured whole steps of n bodies near a built-in switch-over size
    (nbody_ctx_autotuned): choice 0 = built-in decomposition kept, -1 = not measured, else the id of nbody_ctx_autotune."""
    lib = _lib.load()
    h = C.c_void_p()
    check(lib.nbody_default_ctx(C.byref(h)))
    choice, ub, ubest = C.c_int(), C.c_double(), C.c_double()
    check(lib.nbody_ctx_autotuned(h, n, C.byref(choice), C.byref(ub), C.byref(ubest)))
    return {"choice": choice.value, "us_builtin": ub.value, "us_best": ubest.value}


def simulate_pin(n: int, choice: int) -> None:
    """nbody_ctx_set_autotuned on the default context: pin what simulate() uses for n bodies (0 built-in, an autotune id, -1 forget)."""
    lib = _lib.load()
    h = C.c_void_p()
    check(lib.nbody_default_ctx(C.byref(h)))
    check(lib.nbody_ctx_set_autotuned(h, n, choice))


def force_choice(ctx: "Context", choice: int) -> None:
    """Put a context's knobs on the decomposition with autotune id `choice` (1 fused, 24/28/210 balanced runs with 4/8/10 bodies
    per lane, 3 unit runs, 4 block pairs or the two-kernel one-sided path; 0 or -1: leave the built-in choice)."""
    if choice <= 0:
        return
    if choice == 1:
        ctx.set_fused(1)
        return
    ctx.set_fused(0)
    if choice in (24, 28, 210):
        ctx.set_symmetric_runs(2)
        ctx.set_symmetric_shape(0, {24: 4, 28: 8, 210: 10}[choice])
    elif choice == 3:
        ctx.set_symmetric_runs(1)
    elif choice == 4:
        ctx.set_symmetric_runs(0)
    else:
        raise ValueError(f"unknown decomposition id {choice}")


def simulate_host_legacy(bodies: np.ndarray, accelerations3: np.ndarray, velocity3: np.ndarray) -> None:
    """The older snapshot's ``simulate(float4* bodies, float3* accelerations, float3* velocity, int N)``
    (Sim-Without-OpenGL-Integration/kernel.cuh:5): HOST arrays, updated in place (bodies, velocity);
    one step with that snapshot's DT = 0.01 / EPS2 = 0.002 double literals."""
    n = bodies.shape[0]
    for arr, w in ((bodies, 4), (accelerations3, 3), (velocity3, 3)):
        if not (arr.dtype == np.float32 and arr.shape == (n, w) and arr.flags.c_contiguous and arr.flags.writeable):
            raise ValueError("expected contiguous writable float32 arrays of shape (n,4), (n,3), (n,3)")
    check(_lib.load().nbody_simulate_host_legacy(C.c_void_p(bodies.ctypes.data), C.c_void_p(accelerations3.ctypes.data),
                                                 C.c_void_p(velocity3.ctypes.data), n))


class Simulation:
    """The headless run of main.cpp (alloc -> init -> H2D -> step loop), single GPU."""

    def __init__(self, bodies: np.ndarray, dt: float = _lib.DEFAULT_DT, eps2: float = _lib.DEFAULT_EPS2,
                 kernel: int = KERNEL_FAST, device: Optional[int] = None, **kernel_opts):
        bodies = np.ascontiguousarray(bodies, np.float32)
        if bodies.ndim != 2 or bodies.shape[1] != 4:
            raise ValueError("bodies must be (n,4) float32 {x,y,z,mass}")
        self.n = bodies.shape[0]
        self.ctx = Context(device=device, dt=dt, eps2=eps2, kernel=kernel, **kernel_opts)
        dev = torch.device("cuda", self.ctx.device)
        # main.cpp:250-283,352-354: bodies random, velocity/acceleration zero, all copied to the device
        self.x = torch.from_numpy(bodies).to(dev)
        self.v = torch.zeros((self.n, 4), dtype=torch.float32, device=dev)
        self.a = torch.zeros((self.n, 4), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        self.ctx.reserve(self.n)

    def run(self, steps: int, sync: bool = True) -> None:
        """simulationLoopNoVisual (main.cpp:142-160) without the per-step host sync."""
        self.ctx.step(self.x, self.a, self.v, steps)
        if sync:
            self.ctx.sync()

    def state(self):
        self.ctx.sync()
        return self.x.cpu().numpy(), self.v.cpu().numpy(), self.a.cpu().numpy()

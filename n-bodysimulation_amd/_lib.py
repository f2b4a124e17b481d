"""ctypes binding of ``libnbody_hip.so`` (the C-ABI declared in ``include/nbody.h``).

The library is the product: there is no Python or CPU fallback. If it has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C n-bodysimulation_amd/csrc``)
loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnbody_hip.so")

OK = 0
ERR_INVALID = 1
ERR_HIP = 2
ERR_CONFIG = 3
ERR_NOMEM = 4

KERNEL_FAST = 0       # by size: fused one-launch step (<= 8192 bodies), symmetric kernel (balanced runs / unit runs / block pairs) above
KERNEL_STRICT = 1
KERNEL_ONESIDED = 2   # fast arithmetic, every target evaluates all N sources
KERNEL_SYMMETRIC = 3  # fast arithmetic, every unordered pair once (Newton's third law)

DEFAULT_EPS2 = 0.002  # constants.h:25
DEFAULT_DT = 0.1      # constants.h:26


class NBodyError(RuntimeError):
    """Raised for any non-zero status of the C-ABI (the reference throws std::runtime_error
    from simulate(): TestProject/kernel.cu:633-641)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"nbody error {code}: {msg}")
        self.code = code


SCHEDULE_CANONICAL = 0   # strict kernel: own targets x [0, n) in index order
SCHEDULE_ONESIDED = 1    # own x own while gathering, then own x everybody else (one wrapped launch)
SCHEDULE_SYMMETRIC = 2   # every unordered pair once in the machine; J-side sums sent to their owners
MAX_RANKS = 64


class CrossLaunch(C.Structure):   # nbody_cross_launch
    _fields_ = [("i0", C.c_int), ("i1", C.c_int), ("j0", C.c_int), ("count", C.c_int), ("jbuf_offset", C.c_int)]


class Segment(C.Structure):       # nbody_shard_segment
    _fields_ = [("peer", C.c_int), ("offset", C.c_int), ("count", C.c_int), ("body0", C.c_int)]


class ClockReport(C.Structure):
    """include/nbody.h: nbody_clock_report"""
    _fields_ = [("launches", C.c_int), ("xcds", C.c_int), ("unpaired", C.c_int),
                ("cycles_per_launch", C.c_double), ("cycles_per_launch_min", C.c_double), ("cycles_per_launch_max", C.c_double),
                ("ticks_per_launch", C.c_double),
                ("sclk_mhz", C.c_double), ("sclk_mhz_min_xcd", C.c_double), ("sclk_mhz_max_xcd", C.c_double)]


class CommReport(C.Structure):
    """include/nbody.h: nbody_comm_report_t"""
    _fields_ = [("steps", C.c_int), ("gathers", C.c_int), ("exchanges", C.c_int), ("records_kept", C.c_int),
                ("gather_ms", C.c_double), ("gather_exposed_ms", C.c_double), ("exchange_ms", C.c_double), ("exchange_exposed_ms", C.c_double),
                ("gather_ms_max", C.c_double), ("gather_exposed_ms_max", C.c_double), ("exchange_ms_max", C.c_double),
                ("exchange_exposed_ms_max", C.c_double)]


class ShardPlan(C.Structure):     # nbody_shard_plan_t
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("n_total", C.c_int), ("schedule", C.c_int),
                ("shard", C.c_int), ("n_pad", C.c_int), ("i0", C.c_int), ("i1", C.c_int),
                ("n_launches", C.c_int), ("launch", CrossLaunch * 2), ("jbuf_bodies", C.c_int),
                ("n_sends", C.c_int), ("n_recvs", C.c_int), ("send", Segment * MAX_RANKS), ("recv", Segment * MAX_RANKS),
                ("rbuf_bodies", C.c_int)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Segment), C.c_int, C.c_void_p, C.POINTER(Segment), C.c_int,
                          C.c_void_p, C.c_void_p)


class Comm(C.Structure):          # nbody_comm
    _fields_ = [("user", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("exchange", EXCHANGE_FN)]


# every symbol include/nbody.h declares, with its signature
_p = C.c_void_p
_SIGNATURES = {
    "nbody_simulate": (C.c_int, [_p, _p, _p, C.c_int]),
    "nbody_simulate_prepare": (C.c_int, [_p, C.c_int]),
    "nbody_simulate_host_legacy": (C.c_int, [_p, _p, _p, C.c_int]),
    "nbody_default_ctx": (C.c_int, [C.POINTER(_p)]),
    "nbody_ctx_create": (C.c_int, [C.POINTER(_p), C.c_int]),
    "nbody_ctx_destroy": (C.c_int, [_p]),
    "nbody_ctx_set_params": (C.c_int, [_p, C.c_float, C.c_float]),
    "nbody_ctx_set_kernel": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nbody_ctx_set_symmetric_shape": (C.c_int, [_p, C.c_int, C.c_int]),
    "nbody_ctx_set_symmetric_runs": (C.c_int, [_p, C.c_int]),
    "nbody_ctx_autotune": (C.c_int, [_p, _p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "nbody_ctx_set_autotuned": (C.c_int, [_p, C.c_int, C.c_int]),
    "nbody_autotune_decide": (C.c_int, [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double]),
    "nbody_ctx_autotuned": (C.c_int, [_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nbody_ctx_set_fused": (C.c_int, [_p, C.c_int]),
    "nbody_ctx_set_fused_inplace": (C.c_int, [_p, C.c_int]),
    "nbody_ctx_fused_inplace_stats": (C.c_int, [_p, C.POINTER(C.c_ulonglong)]),
    "nbody_ctx_set_equal_mass": (C.c_int, [_p, C.c_int]),
    "nbody_ctx_equal_mass_verdict": (C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "nbody_ctx_set_workspace_limit": (C.c_int, [_p, C.c_size_t, C.c_int]),
    "nbody_ctx_set_inplace_sums": (C.c_int, [_p, C.c_int]),
    "nbody_plan_ticket_task": (C.c_int, [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4),
    "nbody_ctx_set_stream": (C.c_int, [_p, _p]),
    "nbody_ctx_reserve": (C.c_int, [_p, C.c_int]),
    "nbody_ctx_set_graph": (C.c_int, [_p, C.c_int]),
    "nbody_step": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int]),
    "nbody_accel_range": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nbody_accel_square_part": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nbody_accel_wrapped": (C.c_int, [_p, _p, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nbody_accel_cross": (C.c_int, [_p, _p, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "nbody_integrate_range": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int]),
    "nbody_ctx_sync": (C.c_int, [_p]),
    "nbody_ctx_get": (C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_p)]),
    "nbody_shard_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ShardPlan)]),
    "nbody_comm_rccl_unique_id": (C.c_int, [_p]),
    "nbody_comm_rccl_create": (C.c_int, [C.POINTER(Comm), C.c_int, C.c_int, _p, C.c_int]),
    "nbody_comm_rccl_destroy": (C.c_int, [C.POINTER(Comm)]),
    "nbody_shard_create": (C.c_int, [C.POINTER(_p), _p, C.c_int, C.c_int, C.c_int, C.POINTER(Comm)]),
    "nbody_shard_destroy": (C.c_int, [_p]),
    "nbody_shard_get_plan": (C.c_int, [_p, C.POINTER(ShardPlan)]),
    "nbody_shard_buffers": (C.c_int, [_p] + [C.POINTER(_p)] * 5),
    "nbody_shard_upload": (C.c_int, [_p, _p]),
    "nbody_shard_upload_velocity": (C.c_int, [_p, _p]),
    "nbody_shard_download": (C.c_int, [_p, _p, _p, _p]),
    "nbody_shard_step": (C.c_int, [_p, C.c_int]),
    "nbody_shard_step_phase": (C.c_int, [_p, C.c_int]),
    "nbody_shard_sync": (C.c_int, [_p]),
    "nbody_comm_local_group_create": (C.c_int, [C.POINTER(_p), C.c_int, C.c_double]),
    "nbody_comm_local_group_destroy": (C.c_int, [_p]),
    "nbody_comm_local_create": (C.c_int, [C.POINTER(Comm), _p, C.c_int, C.c_int]),
    "nbody_comm_local_destroy": (C.c_int, [C.POINTER(Comm)]),
    "nbody_comm_local_abort": (C.c_int, [_p]),
    "nbody_shard_comm_timing": (C.c_int, [_p, C.c_int]),
    "nbody_shard_set_comm_priority": (C.c_int, [_p, C.c_int]),
    "nbody_shard_comm_report": (C.c_int, [_p, C.POINTER(C.c_int)] + [C.POINTER(C.c_double)] * 4),
    "nbody_shard_comm_report_ex": (C.c_int, [_p, C.POINTER(CommReport)]),
    "nbody_ctx_timing": (C.c_int, [_p, C.c_int]),
    "nbody_ctx_timing_read": (C.c_int, [_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "nbody_ctx_clock_read": (C.c_int, [_p, C.POINTER(ClockReport)]),
    "nbody_step_f64": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_double, C.c_double]),
    "nbody_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "nbody_malloc_device": (C.c_int, [C.POINTER(_p), C.c_size_t]),
    "nbody_free_device": (C.c_int, [_p]),
    "nbody_malloc_host": (C.c_int, [C.POINTER(_p), C.c_size_t]),
    "nbody_free_host": (C.c_int, [_p]),
    "nbody_memcpy_h2d": (C.c_int, [_p, _p, C.c_size_t]),
    "nbody_memcpy_d2h": (C.c_int, [_p, _p, C.c_size_t]),
    "nbody_device_synchronize": (C.c_int, []),
    "nbody_fill_with_random4": (None, [_p, C.c_int]),
    "nbody_fill_with_zeroes4": (None, [_p, C.c_int]),
    "nbody_fill_with_zeroes3": (None, [_p, C.c_int]),
    "nbody_random_float": (C.c_float, [C.c_float, C.c_float]),
    "nbody_print_device_prop": (C.c_int, []),
    "nbody_verify_equality3": (C.c_int, [_p, _p, C.c_int]),
    "nbody_fill_seeded": (C.c_int, [_p, C.c_int, C.c_int, C.c_ulonglong]),
    "nbody_verify_still_bodies": (C.c_int, [_p, _p, C.c_int]),
    "nbody_verify_equality4": (C.c_int, [_p, _p, C.c_int]),
    "nbody_last_error": (C.c_char_p, []),
    "nbody_version": (C.c_char_p, []),
    "nbody_plan": (C.c_int, [C.c_int] * 7 + [C.POINTER(C.c_int)] * 4),
    "nbody_plan_symmetric": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_int)] * 4),
    "nbody_plan_fused": (C.c_int, [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4),
    "nbody_plan_symmetric_occupancy": (C.c_int, [C.c_int]),
    "nbody_ctx_step_info": (C.c_int, [_p, C.c_int] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_double)]),
    "nbody_ctx_step_info_f64": (C.c_int, [_p, C.c_int] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_double)]),
    "nbody_ctx_square_info": (C.c_int, [_p, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_double)]),
    "nbody_ctx_launch_info": (C.c_int, [_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP library first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (torch/lib) while
        # this library's DT_NEEDED resolves to /opt/rocm's. If torch is imported first the loader
        # reuses torch's copy for us (same SONAME) and streams/events/pointers are shared; the other
        # order maps two runtimes, and whichever initialises second sees "no ROCm-capable device".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc: int) -> None:
    if rc != OK:
        raise NBodyError(rc, load().nbody_last_error().decode(errors="replace"))

"""Multi-GPU step: bodies block-partitioned over the ranks of one node, one all-gather of
positions per step, overlapped with the local-block force pass.

The reference is single-device (device 0 hard-coded: TestProject/kernel.cu:630); this is the
build's own decomposition:

    rank r owns the contiguous index block [r*S, (r+1)*S), S = ceil(N / G)
    every rank holds the whole position/mass array X_full (16 B/body: 16 MiB at N = 1 M)
    V and A of the own block never leave the rank

    per step, on rank r                        stream
      all_gather(X_full <- own block)            comm     (in place, S*16 B per rank)
      A  = forces(own targets, own sources)      compute  (overlaps the all-gather)
      wait(all_gather)
      A += forces(own targets, all other blocks: sources i1, i1+1, ... wrapping around to i0-1)
      v += (dt/2) a ; x += dt v  (own block)     compute
      -> event for the next step's all-gather

The remote pass of step n ends before the integrate of step n (same stream), and the
all-gather of step n+1 waits on that integrate, so no second position buffer is needed.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, over xGMI). All compute
goes through a *backend*: ``HipBackend`` (the C-ABI, the only backend the product ships).
``tests/`` inject a CPU checker backend to exercise this schedule under ``gloo`` without a GPU.
Padding bodies (when G does not divide N) are massless and sit on the first body, so they add
exactly +-0 to every sum.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import KERNEL_FAST


def morton_order(bodies: np.ndarray, bits: int = 10) -> np.ndarray:
    """Permutation that sorts bodies along a 3-D Morton (Z-order) curve, so that a contiguous index
    block is also a compact block of space (BASELINE.json's "spatially block-partitioned"). The
    all-pairs cost does not depend on it; it only makes each rank's block a spatial one."""
    xyz = np.asarray(bodies[:, :3], np.float64)
    lo, hi = xyz.min(0), xyz.max(0)
    q = ((xyz - lo) / np.where(hi > lo, hi - lo, 1.0) * ((1 << bits) - 1)).astype(np.uint64)
    key = np.zeros(len(xyz), np.uint64)
    for b in range(bits):
        for axis in range(3):
            key |= ((q[:, axis] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + axis)
    return np.argsort(key, kind="stable")


class HipBackend:
    """Force/integrate on the rank's GPU through libnbody_hip.so, with a compute stream (the
    context's launch stream) and a communication stream for the all-gather."""

    def __init__(self, device: torch.device, dt: float, eps2: float, kernel: int = KERNEL_FAST, **kernel_opts):
        from .engine import Context
        self.device = device
        torch.cuda.set_device(device)
        self.compute = torch.cuda.Stream(device=device)
        self.comm = torch.cuda.Stream(device=device)
        self.ctx = Context(device=device.index, dt=dt, eps2=eps2, kernel=kernel, stream=self.compute, **kernel_opts)
        self._integrated = torch.cuda.Event()
        self._gathered = torch.cuda.Event()
        # optional per-step communication timing (bench.py): events around the all-gather on the comm
        # stream and after the own-block pass on the compute stream
        self.comm_timing = False
        self._comm_events = []   # (gather_start, gather_end, local_pass_end) per step

    def empty(self, n: int) -> torch.Tensor:
        return torch.zeros((n, 4), dtype=torch.float32, device=self.device)

    def from_numpy(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)

    def accel_range(self, x, a_own, i0, i1, j0, j1, accumulate):
        self.ctx.accel_range(x, a_own, i0, i1, j0, j1, accumulate)

    def accel_wrapped(self, x, a_own, i0, i1, j0, count, accumulate):
        self.ctx.accel_wrapped(x, a_own, i0, i1, j0, count, accumulate)

    def integrate_range(self, x, v_own, a_own, i0, i1):
        self.ctx.integrate_range(x, v_own, a_own, i0, i1)

    # -- stream choreography ---------------------------------------------------------------
    def all_gather(self, x_full: torch.Tensor, i0: int, i1: int, group) -> None:
        """Start the in-place all-gather of the own block on the comm stream, after the
        integrate that produced it."""
        self.comm.wait_event(self._integrated)
        with torch.cuda.stream(self.comm):
            if self.comm_timing:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
                      torch.cuda.Event(enable_timing=True))
                self._comm_events.append(ev)
                ev[0].record(self.comm)
            if dist.get_backend(group) == "nccl":
                dist.all_gather_into_tensor(x_full, x_full[i0:i1], group=group)   # RCCL, in place
            else:
                # gloo (rehearsals on a box without one GPU per rank): stage through a list
                world = dist.get_world_size(group)
                parts = [torch.empty_like(x_full[i0:i1]) for _ in range(world)]
                dist.all_gather(parts, x_full[i0:i1].clone(), group=group)
                x_full.copy_(torch.cat(parts))
            self._gathered.record(self.comm)
            if self.comm_timing:
                self._comm_events[-1][1].record(self.comm)

    def wait_gather(self) -> None:
        if self.comm_timing and self._comm_events:
            self._comm_events[-1][2].record(self.compute)   # the own-block pass has been queued before this point
        self.compute.wait_event(self._gathered)

    def comm_report(self) -> dict:
        """Mean all-gather time and the part of it NOT hidden behind the own-block force pass."""
        self.sync()
        if not self._comm_events:
            return {"steps": 0}
        gather = [a.elapsed_time(b) for a, b, _ in self._comm_events]
        exposed = [max(0.0, c.elapsed_time(b)) for _, b, c in self._comm_events]   # local pass end -> gather end
        self._comm_events = []
        return {"steps": len(gather), "all_gather_ms_avg": sum(gather) / len(gather),
                "exposed_ms_avg": sum(exposed) / len(exposed)}

    def ready(self) -> None:
        """Allocation/upload barrier: tensors were filled on torch's current stream, kernels run on
        self.compute / self.comm, which do not synchronise with it implicitly."""
        torch.cuda.synchronize(self.device)

    def mark_integrated(self) -> None:
        self._integrated.record(self.compute)

    def sync(self) -> None:
        self.compute.synchronize()
        self.comm.synchronize()


class ShardedSimulation:
    """`steps` x (all-gather, local forces, remote forces, integrate) over the ranks of `group`."""

    def __init__(self, bodies: np.ndarray, dt: float = _lib.DEFAULT_DT, eps2: float = _lib.DEFAULT_EPS2,
                 group=None, backend=None, kernel: int = KERNEL_FAST, spatial_sort: bool = False, **kernel_opts):
        bodies = np.ascontiguousarray(bodies, np.float32)
        if bodies.ndim != 2 or bodies.shape[1] != 4:
            raise ValueError("bodies must be (n,4) float32 {x,y,z,mass}")
        # optional one-off Morton sort: index blocks become spatial blocks; undone in gather_state()
        self.perm = morton_order(bodies) if spatial_sort and len(bodies) else None
        if self.perm is not None:
            bodies = np.ascontiguousarray(bodies[self.perm])
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = bodies.shape[0]
        self.shard = (self.n + self.world - 1) // self.world
        self.n_pad = self.shard * self.world
        self.i0 = self.rank * self.shard
        self.i1 = self.i0 + self.shard
        if backend is None:
            if not torch.cuda.is_available():
                raise _lib.NBodyError(_lib.ERR_HIP, "no HIP device: the sharded step has no CPU path")
            local = self.rank % max(torch.cuda.device_count(), 1)
            backend = HipBackend(torch.device("cuda", local), dt, eps2, kernel=kernel, **kernel_opts)
        self.backend = backend
        padded = np.zeros((self.n_pad, 4), np.float32)
        padded[: self.n] = bodies
        if self.n_pad > self.n and self.n > 0:
            padded[self.n:, :3] = bodies[0, :3]  # massless, on top of body 0: contributes +-0
        self.x = backend.from_numpy(padded)          # every rank starts from the same full array
        self.v = backend.empty(self.shard)
        self.a = backend.empty(self.shard)
        self._fresh = True                           # X_full already consistent: skip first gather
        # the uploads and zero fills above ran on the allocating stream: finish them before the first
        # kernel on the backend's own (non-blocking) streams reads or overwrites those arrays
        ready = getattr(backend, "ready", None)
        if ready is not None:
            ready()
        backend.mark_integrated()

    def step(self, steps: int = 1) -> None:
        b = self.backend
        for _ in range(steps):
            gather = self.world > 1 and not self._fresh
            if gather:
                b.all_gather(self.x, self.i0, self.i1, self.group)
            b.accel_range(self.x, self.a, self.i0, self.i1, self.i0, self.i1, False)
            if gather:
                b.wait_gather()
            if self.world > 1:
                # every other rank's block in ONE launch: sources i1, i1+1, ... wrapping around to i0-1
                b.accel_wrapped(self.x, self.a, self.i0, self.i1, self.i1 % self.n_pad, self.n_pad - self.shard, True)
            b.integrate_range(self.x, self.v, self.a, self.i0, self.i1)
            b.mark_integrated()
            self._fresh = False

    def sync(self) -> None:
        self.backend.sync()

    def gather_state(self):
        """(x, v, a) of all N bodies on every rank, as numpy arrays (test/diagnostic helper)."""
        self.sync()
        outs = []
        if self.world > 1:
            # bring X_full up to date first (the own block was advanced by the last integrate)
            xs = [torch.empty_like(self.x[self.i0:self.i1]) for _ in range(self.world)]
            dist.all_gather(xs, self.x[self.i0:self.i1].contiguous(), group=self.group)
            outs.append(torch.cat(xs))
            for t in (self.v, self.a):
                parts = [torch.empty_like(t) for _ in range(self.world)]
                dist.all_gather(parts, t.contiguous(), group=self.group)
                outs.append(torch.cat(parts))
        else:
            outs = [self.x, self.v, self.a]
        res = tuple(o.cpu().numpy()[: self.n] for o in outs)
        if self.perm is not None:       # back to the caller's body order
            inv = np.empty_like(self.perm)
            inv[self.perm] = np.arange(self.n)
            res = tuple(r[inv] for r in res)
        return res

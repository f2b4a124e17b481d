"""Multi-GPU step: bodies block-partitioned over the ranks of one node (one process per GPU).

The reference is single-device (TestProject/kernel.cu:630, main.cpp:287); this is the build's own
decomposition. The step itself is native: ``nbody_shard_*`` in libnbody_hip.so
(csrc/nbody_shard.hip) owns the rank's device arrays, the compute and communication streams and the
events between them, and orders the C-ABI force/integrate launches. This module is the thin host
side: it asks the library for the rank's plan, hands it the two collectives as callbacks over
``torch.distributed`` (backend "nccl" is RCCL on ROCm, over xGMI) and moves state in and out.

    rank r owns the contiguous index block [r*S, (r+1)*S); every rank holds all positions
    (16 B/body: 16 MiB at N = 1 M); velocities and accelerations never leave the rank.

    per step, on rank r (schedule SYMMETRIC, the default)          stream
      all_gather(X_full <- own block), in place, S*16 B per rank     comm     | overlapped
      own block x own block, first half of its block pairs           compute  |
      A  = own block x blocks r+1 .. r+(G-1)/2 (mod G), pairs once   compute  -> J-side sums for OTHER ranks
      exchange: J-side sums to their owners, mine arrive             comm     | overlapped (grouped send/recv)
      A += own block x own block (second half, then its slab sum)    compute  |
      A += received sums (fixed order); v += (dt/2) a ; x += dt v    compute

so every unordered pair of bodies is evaluated exactly once in the whole machine. With the ONESIDED
kernel a rank evaluates its targets against all N sources (no exchange); with the STRICT kernel it
does so in index order, bit-identical to the single-device strict step.

``shard_plan`` (pure host logic of the library) and ``TorchComm`` also work without a GPU: the CPU
tests drive the same plan and the same collectives under gloo with the checker as compute.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import KERNEL_FAST, SCHEDULE_CANONICAL, SCHEDULE_ONESIDED, SCHEDULE_SYMMETRIC, check  # noqa: F401


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


def shard_plan(rank: int, world: int, n_total: int, schedule: int = SCHEDULE_SYMMETRIC) -> _lib.ShardPlan:
    """nbody_shard_plan: what `rank` of `world` does for n_total bodies (no device needed)."""
    p = _lib.ShardPlan()
    rc = _lib.load().nbody_shard_plan(rank, world, n_total, schedule, C.byref(p))
    if rc != _lib.OK:
        raise _lib.NBodyError(rc, f"bad shard geometry: rank {rank} of {world}, {n_total} bodies, schedule {schedule}")
    return p


def schedule_of(kernel: int) -> int:
    return {_lib.KERNEL_STRICT: SCHEDULE_CANONICAL, _lib.KERNEL_ONESIDED: SCHEDULE_ONESIDED}.get(kernel, SCHEDULE_SYMMETRIC)


class TorchComm:
    """The two collectives of the sharded step over torch.distributed, on torch tensors of bodies
    ((n,4) float32, CPU or GPU). NCCL (= RCCL) works on device memory in place; with gloo, device
    tensors are staged through the host (rehearsals on a box with fewer GPUs than ranks)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"

    def all_gather(self, x_full: torch.Tensor, shard: int) -> None:
        own = x_full[self.rank * shard:(self.rank + 1) * shard]
        if self.world == 1:
            return
        if self.nccl:
            dist.all_gather_into_tensor(x_full, own, group=self.group)   # RCCL, in place
            return
        parts = [torch.empty((shard, 4), dtype=x_full.dtype) for _ in range(self.world)]
        dist.all_gather(parts, own.detach().cpu().contiguous(), group=self.group)
        x_full.copy_(torch.cat(parts))

    def exchange(self, sends, jbuf: torch.Tensor, recvs, rbuf: torch.Tensor) -> None:
        """sends / recvs: sequences of (peer, offset, count); one grouped send/recv."""
        if self.world == 1 or (not sends and not recvs):
            return
        if self.nccl:
            ops = [dist.P2POp(dist.isend, jbuf[o:o + c], p, self.group) for (p, o, c) in sends]
            ops += [dist.P2POp(dist.irecv, rbuf[o:o + c], p, self.group) for (p, o, c) in recvs]
            for req in dist.batch_isend_irecv(ops):
                req.wait()          # orders the current stream behind the transfer; the host does not block
            return
        out = [jbuf[o:o + c].detach().cpu().contiguous() for (_, o, c) in sends]
        inp = [torch.empty((c, 4), dtype=rbuf.dtype) for (_, _, c) in recvs]
        reqs = [dist.isend(t, p, group=self.group) for t, (p, _, _) in zip(out, sends)]
        reqs += [dist.irecv(t, p, group=self.group) for t, (p, _, _) in zip(inp, recvs)]
        for r in reqs:
            r.wait()
        for t, (_, o, c) in zip(inp, recvs):
            rbuf[o:o + c].copy_(t)


class NativeComm:
    """The library's own RCCL communicator (nbody_comm_rccl_*: librccl loaded at run time, ncclAllGather in place and one
    grouped ncclSend/ncclRecv per step, enqueued from C on the shard's communication stream — no Python in the step).
    The 128-byte unique id is made by rank 0 and broadcast through the existing torch.distributed group (any backend):
    that group stays the control plane (barriers, reductions of timings, gathering of results)."""

    def __init__(self, device: torch.device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.native = True
        lib = _lib.load()
        uid = C.create_string_buffer(128)
        if self.rank == 0:
            check(lib.nbody_comm_rccl_unique_id(uid))
        if self.world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = C.create_string_buffer(box[0], 128)
        self.struct = _lib.Comm()
        check(lib.nbody_comm_rccl_create(C.byref(self.struct), self.rank, self.world, uid, device.index))

    def close(self) -> None:
        if getattr(self, "struct", None) is not None:
            _lib.load().nbody_comm_rccl_destroy(C.byref(self.struct))
            self.struct = None


class _DeviceArray:
    """Zero-copy view of device memory the library owns, for torch.as_tensor."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}


class ShardedSimulation:
    """`steps` x (all-gather, own-block forces, cross/remote forces, exchange, integrate) over the ranks of `group`."""

    def __init__(self, bodies: np.ndarray, dt: float = _lib.DEFAULT_DT, eps2: float = _lib.DEFAULT_EPS2,
                 group=None, kernel: int = KERNEL_FAST, device: Optional[torch.device] = None, spatial_sort: bool = False,
                 comm=None, comm_priority: Optional[str] = None, **kernel_opts):
        """comm: None or "torch" = the two collectives over torch.distributed (TorchComm; RCCL when the group's backend
        is nccl); "native" = the library's own RCCL communicator (NativeComm); or a ready TorchComm / NativeComm.
        comm_priority: "normal" (the library's default) or "high" — the priority of the rank's communication stream
        (nbody_shard_set_comm_priority; only the library's own communicator launches RCCL's kernels on that stream)."""
        from .engine import Context
        bodies = np.ascontiguousarray(bodies, np.float32)
        if bodies.ndim != 2 or bodies.shape[1] != 4:
            raise ValueError("bodies must be (n,4) float32 {x,y,z,mass}")
        if not torch.cuda.is_available():
            raise _lib.NBodyError(_lib.ERR_HIP, "no HIP device: the sharded step has no CPU path")
        # optional one-off Morton sort: index blocks become spatial blocks; undone in gather_state()
        self.perm = morton_order(bodies) if spatial_sort and len(bodies) else None
        if self.perm is not None:
            bodies = np.ascontiguousarray(bodies[self.perm])
        if device is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
            device = torch.device("cuda", rank % max(torch.cuda.device_count(), 1))
        self.device = device
        torch.cuda.set_device(device)
        if comm is None or comm == "torch":
            comm = TorchComm(group)
        elif comm == "native":
            comm = NativeComm(device, group)
        self.comm = comm
        self.group, self.world, self.rank = self.comm.group, self.comm.world, self.comm.rank
        self.n = bodies.shape[0]
        sym_waves, sym_bpl = kernel_opts.pop("sym_waves", 0), kernel_opts.pop("sym_bpl", 0)
        self.ctx = Context(device=device.index, dt=dt, eps2=eps2, kernel=kernel, **kernel_opts)
        if sym_waves or sym_bpl:
            self.ctx.set_symmetric_shape(sym_waves, sym_bpl)
        self._lib = _lib.load()
        # the callbacks stay referenced for the life of the shard (ctypes does not keep them alive)
        self._cb_gather = _lib.ALL_GATHER_FN(self._on_all_gather)
        self._cb_exchange = _lib.EXCHANGE_FN(self._on_exchange)
        self._comm_struct = self.comm.struct if getattr(self.comm, "native", False) else _lib.Comm(None, self._cb_gather, self._cb_exchange)
        self._h = C.c_void_p()
        self._error = None
        check(self._lib.nbody_shard_create(C.byref(self._h), self.ctx._h, self.rank, self.world, self.n, C.byref(self._comm_struct)))
        self.plan = _lib.ShardPlan()
        check(self._lib.nbody_shard_get_plan(self._h, C.byref(self.plan)))
        self.shard, self.n_pad, self.i0, self.i1 = self.plan.shard, self.plan.n_pad, self.plan.i0, self.plan.i1
        ptrs = [C.c_void_p() for _ in range(5)]
        check(self._lib.nbody_shard_buffers(self._h, *[C.byref(p) for p in ptrs]))
        wrap = lambda p, n: torch.as_tensor(_DeviceArray(p.value, max(n, 1)), device=device)[:n]
        self.x = wrap(ptrs[0], self.n_pad)          # all positions (library-owned device memory)
        self.v = wrap(ptrs[1], self.shard)
        self.a = wrap(ptrs[2], self.shard)
        self._jbuf = wrap(ptrs[3], self.plan.jbuf_bodies)
        self._rbuf = wrap(ptrs[4], self.plan.rbuf_bodies)
        if comm_priority is not None:
            if comm_priority not in ("high", "normal"):
                raise ValueError("comm_priority must be 'high' or 'normal'")
            check(self._lib.nbody_shard_set_comm_priority(self._h, 1 if comm_priority == "high" else 0))
        check(self._lib.nbody_shard_upload(self._h, C.c_void_p(bodies.ctypes.data)))

    def set_comm_priority(self, priority: str) -> None:
        """nbody_shard_set_comm_priority between steps: "normal" or "high" (synchronises both streams)."""
        if priority not in ("high", "normal"):
            raise ValueError("comm_priority must be 'high' or 'normal'")
        self._check(self._lib.nbody_shard_set_comm_priority(self._h, 1 if priority == "high" else 0))

    def reset(self, bodies: np.ndarray) -> None:
        """New bodies (same count) for every rank, at rest: nbody_shard_upload again (synchronous; every rank calls it
        with the same array). The next step skips the all-gather, as the first one did."""
        bodies = np.ascontiguousarray(bodies, np.float32)
        if bodies.shape != (self.n, 4):
            raise ValueError(f"expected ({self.n},4) bodies")
        if self.perm is not None:
            bodies = np.ascontiguousarray(bodies[self.perm])
        self._check(self._lib.nbody_shard_upload(self._h, C.c_void_p(bodies.ctypes.data)))

    # -- the two collectives, called by the library with the communication stream to enqueue on -----------
    def _on_all_gather(self, user, d_x_full, bodies_per_rank, stream):
        try:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                self.comm.all_gather(self.x, bodies_per_rank)
            return 0
        except Exception as e:   # an exception must not unwind through the C frame
            self._error = e
            return 1

    def _on_exchange(self, user, send, n_sends, d_jbuf, recv, n_recvs, d_rbuf, stream):
        try:
            sends = [(send[k].peer, send[k].offset, send[k].count) for k in range(n_sends)]
            recvs = [(recv[k].peer, recv[k].offset, recv[k].count) for k in range(n_recvs)]
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                self.comm.exchange(sends, self._jbuf, recvs, self._rbuf)
            return 0
        except Exception as e:
            self._error = e
            return 1

    def _check(self, rc: int) -> None:
        if rc != _lib.OK and self._error is not None:
            e, self._error = self._error, None
            raise e
        check(rc)

    def step(self, steps: int = 1) -> None:
        self._check(self._lib.nbody_shard_step(self._h, steps))

    def sync(self) -> None:
        self._check(self._lib.nbody_shard_sync(self._h))

    def refresh_positions(self) -> None:
        """One all-gather of positions without a step (phase 0 alone) and a sync: afterwards every rank's copy of every
        block is the owner's current one. Diagnostic (bench.py's cross-rank check); call it with comm timing off."""
        self._check(self._lib.nbody_shard_step_phase(self._h, 0))
        self.sync()

    def block_checksums(self) -> torch.Tensor:
        """(world, 2) int64 on the device: for every rank's block of THIS rank's position array, the wrapping sum of the
        float bit patterns and the wrapping sum of pattern * (word index + 1). Equal rows on every rank <=> the copies
        are bit-identical (up to a 2^-64 accident)."""
        bits = self.x[: self.world * self.shard].contiguous().view(torch.int32).to(torch.int64).view(self.world, self.shard * 4)
        idx = torch.arange(1, self.shard * 4 + 1, device=bits.device, dtype=torch.int64)
        return torch.stack([bits.sum(1), (bits * idx).sum(1)], dim=1)

    def comm_timing(self, enable: bool) -> None:
        self._check(self._lib.nbody_shard_comm_timing(self._h, 1 if enable else 0))

    def comm_report(self) -> dict:
        """Over the timed steps, mean AND maximum: all-gather time, the part of it not hidden behind the own-block pass, exchange
        time and its exposed part (a single late collective shows in the maxima, not in the means)."""
        r = _lib.CommReport()
        self._check(self._lib.nbody_shard_comm_report_ex(self._h, C.byref(r)))
        return {"steps": r.steps, "all_gather_ms_avg": r.gather_ms, "exposed_ms_avg": r.gather_exposed_ms, "exchange_ms_avg": r.exchange_ms,
                "exchange_exposed_ms_avg": r.exchange_exposed_ms,
                "all_gather_ms_max": r.gather_ms_max, "exposed_ms_max": r.gather_exposed_ms_max, "exchange_ms_max": r.exchange_ms_max,
                "exchange_exposed_ms_max": r.exchange_exposed_ms_max,
                "gathers": r.gathers, "exchanges": r.exchanges, "records_kept": r.records_kept,
                "schedule": {0: "canonical", 1: "onesided", 2: "symmetric"}[self.plan.schedule]}

    def set_velocity(self, velocity: np.ndarray) -> None:
        """Velocities of all N bodies in the caller's order (to continue a run; construction starts at rest)."""
        v = np.ascontiguousarray(velocity, np.float32)
        if v.shape != (self.n, 4):
            raise ValueError(f"expected ({self.n},4) velocities")
        if self.perm is not None:
            v = np.ascontiguousarray(v[self.perm])
        self._check(self._lib.nbody_shard_upload_velocity(self._h, C.c_void_p(v.ctypes.data)))

    def own_state(self):
        """(x, v, a) of the own block (shard rows, padding included) as numpy arrays."""
        out = [np.zeros((self.shard, 4), np.float32) for _ in range(3)]
        self._check(self._lib.nbody_shard_download(self._h, *[C.c_void_p(o.ctypes.data) for o in out]))
        return tuple(out)

    def gather_state(self):
        """(x, v, a) of all N bodies on every rank, as numpy arrays (test/diagnostic helper)."""
        own = self.own_state()
        if self.world > 1:
            outs = []
            for o in own:
                t = torch.from_numpy(o)
                if self.comm.nccl:
                    t = t.to(self.device)
                parts = [torch.empty_like(t) for _ in range(self.world)]
                dist.all_gather(parts, t, group=self.group)
                outs.append(torch.cat(parts).cpu().numpy())
        else:
            outs = list(own)
        res = tuple(o[: self.n] for o in outs)
        if self.perm is not None:       # back to the caller's body order
            inv = np.empty_like(self.perm)
            inv[self.perm] = np.arange(self.n)
            res = tuple(r[inv] for r in res)
        return res

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.nbody_shard_destroy(self._h)
            self._h = C.c_void_p()
            self.x = self.v = self.a = self._jbuf = self._rbuf = None
            self.ctx.close()            # after the shard: it launches on the context's stream
            if getattr(self.comm, "native", False):
                self.comm.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

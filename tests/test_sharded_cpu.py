"""The N>1 schedule (partition, in-place all-gather, local/remote passes, integrate) under gloo
with world_size 2 and 3 on CPU. The compute backend injected here is the CPU checker — the
product itself ships only the HIP backend and refuses to run without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CheckerBackend:
    """Test double with the HipBackend interface, computing through oracle/ (fp32 sequential,
    continuing sums across calls exactly like the strict kernel)."""

    def __init__(self, dt, eps2):
        from oracle import oracle as O
        self.O, self.dt, self.eps2 = O, dt, eps2
        self.log = []

    def empty(self, n):
        return torch.zeros((n, 4), dtype=torch.float32)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32).copy())

    def accel_range(self, x, a_own, i0, i1, j0, j1, accumulate):
        self.log.append(("accel", i0, i1, j0, j1, bool(accumulate)))
        xn, an = x.numpy(), a_own.numpy()
        if not accumulate:
            an[:] = self.O.accel_range(xn, i0, i1, j0, j1, eps2=self.eps2)
            return
        # continue each sequential sum exactly, term by term, as the strict kernel does
        for t, i in enumerate(range(i0, i1)):
            acc = an[t].copy()
            for j in range(j0, j1):
                if j != i:
                    acc = self.O.pair(xn[i], xn[j], acc, eps2=self.eps2)
            an[t] = acc

    def accel_wrapped(self, x, a_own, i0, i1, j0, count, accumulate):
        self.log.append(("accel", i0, i1, j0, j0 + count, bool(accumulate)))
        assert accumulate
        xn, an = x.numpy(), a_own.numpy()
        n = len(xn)
        for t, i in enumerate(range(i0, i1)):
            acc = an[t].copy()
            for jj in range(j0, j0 + count):
                j = jj % n
                if j != i:
                    acc = self.O.pair(xn[i], xn[j], acc, eps2=self.eps2)
            an[t] = acc

    def integrate_range(self, x, v_own, a_own, i0, i1):
        self.log.append(("integrate", i0, i1))
        xs = x.numpy()[i0:i1].copy()
        vs = v_own.numpy()
        self.O.integrate(xs, vs, a_own.numpy(), dt=self.dt)
        x.numpy()[i0:i1] = xs

    def all_gather(self, x_full, i0, i1, group):
        self.log.append(("gather", i0, i1))
        world = dist.get_world_size(group)
        parts = [torch.empty_like(x_full[i0:i1]) for _ in range(world)]
        dist.all_gather(parts, x_full[i0:i1].clone(), group=group)
        x_full.copy_(torch.cat(parts))

    def wait_gather(self):
        self.log.append(("wait",))

    def mark_integrated(self):
        pass

    def sync(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, steps, q, spatial=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd
        from oracle import oracle as O
        O.set_threads(1)
        x0 = nbody_amd.engine.seeded_bodies(n, 0, 31)
        be = CheckerBackend(0.1, 0.002)
        sim = nbody_amd.sharded.ShardedSimulation(x0, dt=0.1, eps2=0.002, backend=be, spatial_sort=spatial)
        own0 = sim.x.numpy()[sim.i0:min(sim.i1, n), :3].copy()
        sim.step(steps)
        x, v, a = sim.gather_state()
        q.put((rank, x, v, a, be.log, sim.i0, sim.i1, sim.n_pad, own0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 96), (3, 100), (2, 1)])
def test_sharded_schedule_matches_single_rank(world, n):
    from oracle import oracle as O
    import nbody_amd
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    x0 = nbody_amd.engine.seeded_bodies(n, 0, 31)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    O.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=steps)
    amax = max(np.abs(ao[:, :3]).max(), 1e-30)
    for rank, x, v, a, log, i0, i1, n_pad, _own in res:
        # every rank ends with the same full state, equal to the single-rank Jacobi step up to the
        # summation order (own block first, then the blocks before and after it)
        assert np.abs(x - xo)[:, :3].max() / 1e5 <= 1e-6
        assert np.abs(a - ao)[:, :3].max() / amax <= 1e-5
        assert np.array_equal(x[:, 3], x0[:, 3])
        assert np.array_equal(x, res[0][1]) and np.array_equal(a, res[0][3])
        # schedule: no gather before the first step; afterwards gather -> local -> wait -> remote -> integrate
        kinds = [e[0] for e in log]
        assert kinds[0] == "accel" and kinds.count("gather") == steps - 1 and kinds.count("integrate") == steps
        first_gather = kinds.index("gather")
        assert kinds[first_gather + 1] == "accel" and log[first_gather + 1][3:5] == (i0, i1)   # local block while gathering
        assert kinds[first_gather + 2] == "wait"
        assert i1 - i0 == (n + world - 1) // world and n_pad == (i1 - i0) * world


def test_spatial_sort_gives_spatial_blocks_and_same_answer():
    """spatial_sort=True: each rank's index block is a compact region (Morton order), results come
    back in the caller's order and agree with the unsorted run to summation-order tolerance."""
    from oracle import oracle as O
    import nbody_amd
    world, n, steps = 2, 256, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, steps, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x0 = nbody_amd.engine.seeded_bodies(n, 0, 31)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    O.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=steps)
    whole = np.prod(x0[:, :3].max(0) - x0[:, :3].min(0))
    for rank, x, v, a, log, i0, i1, n_pad, own0 in res:
        assert np.abs(x - xo)[:, :3].max() / 1e5 <= 1e-6
        assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
        assert np.array_equal(x[:, 3], x0[:, 3])                    # caller's order restored
        assert np.prod(own0.max(0) - own0.min(0)) <= 0.62 * whole    # a half-space block, not the whole cube
    perm = nbody_amd.sharded.morton_order(x0)
    assert sorted(perm.tolist()) == list(range(n))


def test_sharded_refuses_cpu_without_backend():
    import nbody_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nbody_amd.NBodyError):
        nbody_amd.sharded.ShardedSimulation(np.zeros((4, 4), np.float32))

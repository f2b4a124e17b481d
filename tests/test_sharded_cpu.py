"""The N>1 path on CPU: the library's shard plan (pure host logic: who evaluates which pairs, who owes whom
which sums) checked exhaustively, and the three schedules executed under gloo with world sizes 2, 3 and 4 —
the product's own plan (nbody_shard_plan) and collectives (sharded.TorchComm), with the CPU checker standing in
for the device kernels. The product itself runs the step natively on the GPU (csrc/nbody_shard.hip) and refuses
to run without one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# ---- the plan ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 33, 64])
def test_symmetric_plan_covers_every_cross_pair_exactly_once(world):
    import nbody_amd
    from nbody_amd import sharded
    S = 6
    n = world * S - 3 if world > 1 else 5          # not a multiple: padding bodies take part like any other
    plans = [sharded.shard_plan(r, world, n) for r in range(world)]
    seen = {}
    for p in plans:
        assert p.shard % 2 == 0 and p.n_pad == p.shard * world and (p.i0, p.i1) == (p.rank * p.shard, (p.rank + 1) * p.shard)
        off = 0
        for l in range(p.n_launches):
            L = p.launch[l]
            assert p.i0 <= L.i0 < L.i1 <= p.i1 and L.jbuf_offset == off and L.count > 0
            off += L.count
            for i in range(L.i0, L.i1):
                for t in range(L.count):
                    j = (L.j0 + t) % p.n_pad
                    assert not (p.i0 <= j < p.i1)                      # never a body of the own block
                    key = (min(i, j), max(i, j))
                    assert key not in seen, (world, key, seen[key], p.rank)
                    seen[key] = p.rank
        assert off == p.jbuf_bodies
    S2 = plans[0].shard
    cross = sum(1 for i in range(world * S2) for j in range(i + 1, world * S2) if i // S2 != j // S2)
    assert len(seen) == cross                                          # every cross-block pair, once
    work = [sum((p.launch[l].i1 - p.launch[l].i0) * p.launch[l].count for l in range(p.n_launches)) for p in plans]
    assert max(work) == min(work)                                      # and the ranks share them evenly
    # what one rank sends is what the other expects, body for body
    for p in plans:
        for k in range(p.n_sends):
            s = p.send[k]
            q = plans[s.peer]
            match = [q.recv[m] for m in range(q.n_recvs) if q.recv[m].peer == p.rank]
            assert len(match) == 1 and (match[0].count, match[0].body0) == (s.count, s.body0)
            assert q.i0 <= s.body0 and s.body0 + s.count <= q.i1
        offs = [p.recv[m].offset for m in range(p.n_recvs)]
        assert offs == sorted(offs) and (p.n_recvs == 0 or p.recv[p.n_recvs - 1].offset + p.recv[p.n_recvs - 1].count == p.rbuf_bodies)
        order = [(p.rank - p.recv[m].peer) % world for m in range(p.n_recvs)]
        assert order == sorted(order)                                  # nearest preceding rank first


def test_plan_of_the_other_schedules_and_bad_arguments():
    import nbody_amd
    from nbody_amd import sharded
    for sched in (sharded.SCHEDULE_CANONICAL, sharded.SCHEDULE_ONESIDED):
        p = sharded.shard_plan(2, 8, 1048576, sched)
        assert (p.shard, p.n_pad, p.i0, p.i1) == (131072, 1048576, 262144, 393216)
        assert p.n_launches == 0 and p.n_sends == 0 and p.n_recvs == 0
    p = sharded.shard_plan(5, 8, 1048576)                              # BASELINE configs[3], an upper rank of an even ring
    assert p.n_launches == 2 and p.jbuf_bodies == 3 * 131072 + 131072
    assert (p.launch[1].i0, p.launch[1].i1, p.launch[1].j0, p.launch[1].count) == (5 * 131072 + 65536, 6 * 131072, 131072, 131072)
    p = sharded.shard_plan(1, 8, 1048576)
    assert p.n_launches == 1 and p.launch[0].count == 3 * 131072 + 65536
    for bad in ((3, 3, 10), (-1, 2, 10), (0, 0, 10), (0, 65, 10), (0, 2, -1)):
        with pytest.raises(nbody_amd.NBodyError):
            sharded.shard_plan(*bad)


# ---- the schedules under gloo ---------------------------------------------------------------------------------

class PlanExecutor:
    """Executes the library's plan with the CPU checker as compute and sharded.TorchComm as transport."""

    def __init__(self, bodies, dt, eps2, schedule):
        from nbody_amd import sharded
        from oracle import oracle as O
        self.O, self.dt, self.eps2 = O, dt, eps2
        self.comm = sharded.TorchComm()
        n = len(bodies)
        self.p = p = sharded.shard_plan(self.comm.rank, self.comm.world, n, schedule)
        padded = np.zeros((p.n_pad, 4), np.float32)
        padded[:n] = bodies
        if p.n_pad > n:
            padded[n:, :3] = bodies[0, :3]
        self.x = torch.from_numpy(padded)
        self.v = torch.zeros((p.shard, 4))
        self.a = torch.zeros((p.shard, 4))
        self.jbuf = torch.zeros((p.jbuf_bodies, 4))
        self.rbuf = torch.zeros((p.rbuf_bodies, 4))
        self.fresh = True
        self.log = []

    def _accel(self, targets, sources):
        """float32 accelerations of bodies `targets` (index array) from `sources` (index array), disjoint sets."""
        xs = np.ascontiguousarray(np.concatenate([self.x.numpy()[targets], self.x.numpy()[sources]]))
        nt = len(targets)
        return self.O.accel_range(xs, 0, nt, nt, nt + len(sources), eps2=self.eps2, f64acc=True)

    def step(self, steps):
        p, O = self.p, self.O
        for _ in range(steps):
            if p.world > 1 and not self.fresh:
                self.log.append("gather")
                self.comm.all_gather(self.x, p.shard)
            xn, an = self.x.numpy(), self.a.numpy()
            if p.schedule == 0:
                an[:] = O.accel_range(xn, p.i0, p.i1, 0, p.n_pad, eps2=self.eps2)          # index order, fp32 sequential
            else:
                an[:] = O.accel_range(xn, p.i0, p.i1, p.i0, p.i1, eps2=self.eps2, f64acc=True)
                own = np.arange(p.i0, p.i1)
                if p.schedule == 1 and p.world > 1:
                    an += self._accel(own, (np.arange(p.n_pad - p.shard) + p.i1) % p.n_pad)
                for l in range(p.n_launches):
                    L = p.launch[l]
                    tg, run = np.arange(L.i0, L.i1), (np.arange(L.count) + L.j0) % p.n_pad
                    an[L.i0 - p.i0:L.i1 - p.i0] += self._accel(tg, run)
                    self.jbuf.numpy()[L.jbuf_offset:L.jbuf_offset + L.count] = self._accel(run, tg)
            if p.schedule == 2 and p.world > 1:
                self.log.append("exchange")
                self.comm.exchange([(p.send[k].peer, p.send[k].offset, p.send[k].count) for k in range(p.n_sends)], self.jbuf,
                                   [(p.recv[k].peer, p.recv[k].offset, p.recv[k].count) for k in range(p.n_recvs)], self.rbuf)
                for k in range(p.n_recvs):
                    r = p.recv[k]
                    an[r.body0 - p.i0:r.body0 - p.i0 + r.count, :3] += self.rbuf.numpy()[r.offset:r.offset + r.count, :3]
            xs = xn[p.i0:p.i1].copy()
            O.integrate(xs, self.v.numpy(), an, dt=self.dt)
            xn[p.i0:p.i1] = xs
            self.fresh = False

    def gather_state(self, n):
        outs = []
        for t in (self.x[self.p.i0:self.p.i1], self.v, self.a):
            parts = [torch.empty_like(t) for _ in range(self.p.world)]
            dist.all_gather(parts, t.contiguous())
            outs.append(torch.cat(parts).numpy()[:n])
        return outs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, steps, schedule, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd
        from oracle import oracle as O
        O.set_threads(1)
        x0 = nbody_amd.engine.seeded_bodies(n, 0, 31)
        ex = PlanExecutor(x0, 0.1, 0.002, schedule)
        ex.step(steps)
        x, v, a = ex.gather_state(n)
        q.put((rank, x, v, a, ex.log, ex.p.shard, ex.p.n_pad))
    finally:
        dist.destroy_process_group()


def _run(world, n, steps, schedule):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, steps, schedule, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,n,schedule", [(2, 96, 2), (3, 100, 2), (4, 130, 2), (2, 1, 2), (2, 96, 1), (3, 100, 0), (2, 97, 0)])
def test_schedules_match_the_single_rank_step(world, n, schedule):
    from oracle import oracle as O
    import nbody_amd
    steps = 3
    res = _run(world, n, steps, schedule)
    x0 = nbody_amd.engine.seeded_bodies(n, 0, 31)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    O.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=steps)
    amax = max(np.abs(ao[:, :3]).max(), 1e-30)
    for rank, x, v, a, log, shard, n_pad in res:
        if schedule == 0:      # canonical order: the single-rank Jacobi step, bit for bit (padding adds +-0)
            assert np.array_equal(x, xo) and np.array_equal(v, vo) and np.array_equal(a, ao)
        else:                  # other summation orders: tolerance
            assert np.abs(x - xo)[:, :3].max() / 1e5 <= 1e-6
            assert np.abs(a - ao)[:, :3].max() / amax <= 1e-5
        assert np.array_equal(x[:, 3], x0[:, 3])
        assert np.array_equal(x, res[0][1]) and np.array_equal(a, res[0][3])      # every rank ends with the same state
        assert log.count("gather") == steps - 1                                  # none before the first step
        assert log.count("exchange") == (steps if schedule == 2 else 0)
        assert shard % 2 == 0 and n_pad == shard * world and shard >= (n + world - 1) // world


def test_morton_order_is_a_permutation_into_compact_blocks():
    import nbody_amd
    from nbody_amd import sharded
    x0 = nbody_amd.engine.seeded_bodies(256, 0, 31)
    perm = sharded.morton_order(x0)
    assert sorted(perm.tolist()) == list(range(256))
    whole = np.prod(x0[:, :3].max(0) - x0[:, :3].min(0))
    half = x0[perm[:128], :3]
    assert np.prod(half.max(0) - half.min(0)) <= 0.62 * whole       # a half-space block, not the whole cube


def test_sharded_refuses_cpu():
    import nbody_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nbody_amd.NBodyError):
        nbody_amd.sharded.ShardedSimulation(np.zeros((4, 4), np.float32))

"""The sharded step on the GPU: a 1-rank RCCL group (the only world size a 1-GPU box allows), and
the 2- and 3-block decomposition driven by hand on one device (same C-ABI calls a rank makes)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from _ranks import check_multi_gpu_line, free_port as _free_port, mark, run_bench, run_ranks, torchrun
from conftest import same_bits

pytestmark = pytest.mark.gpu


def test_one_rank_nccl_group_equals_single_gpu(nb):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n = 5000
        x0 = nb.engine.seeded_bodies(n, 1, 9)
        sh = nb.sharded.ShardedSimulation(x0, dt=0.01, eps2=0.002)
        sh.step(4)
        xs, vs, as_ = sh.gather_state()
        sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
        sim.ctx.set_fused(0)             # the shard builds its step from nbody_accel_range + nbody_integrate_range: the two-kernel path
        sim.run(4)
        x, v, a = sim.state()
        assert np.array_equal(xs, x) and np.array_equal(vs, v) and np.array_equal(as_, a)   # one rank == nbody_step, bit for bit
        fused = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)                               # what nbody_step runs by default at this size
        fused.run(4)
        assert np.abs(fused.state()[0] - x)[:, :3].max() <= 1e-6
        # the RCCL calls of the multi-rank path, as far as one rank can exercise them: the in-place
        # all_gather_into_tensor on the library's communication stream, MAX all-reduce, barrier
        before = sh.x.clone()
        assert sh._on_all_gather(None, None, sh.shard, torch.cuda.current_stream().cuda_stream) == 0
        dist.all_gather_into_tensor(sh.x, sh.x[sh.i0:sh.i1])
        torch.cuda.synchronize()
        assert torch.equal(sh.x, before)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(t.item()) == 1.5
        sh.step(2)                       # and stepping still works afterwards
        sh.sync()
        sh.close()
    finally:
        dist.destroy_process_group()


class _OneThreadRanks:
    """`world` ranks of the native sharded step (nbody_shard_*) driven by ONE host thread on one GPU: the step is
    taken phase by phase (nbody_shard_step_phase) for all ranks, and the two collectives are plain device copies
    made by the callbacks from what the plans say — exactly the transport contract of nbody_comm."""

    def __init__(self, nb, x0, world, kernel, dt, eps2, sym_shape=None):
        import ctypes as C
        self.C, self.nb, self.lib, self.world = C, nb, nb.load(), world
        L = nb._lib
        self.ctxs, self.shards, self.plans, self.bufs, self._keep = [], [], [], [], []
        for r in range(world):
            ctx = nb.engine.Context(dt=dt, eps2=eps2, kernel=kernel)
            if sym_shape:
                ctx.set_symmetric_shape(*sym_shape)
            g = L.ALL_GATHER_FN(lambda user, d_x, per_rank, stream, r=r: self._gather(r, per_rank))
            e = L.EXCHANGE_FN(lambda user, send, ns, d_j, recv, nr, d_r, stream, r=r: self._exchange(r, send, ns, recv, nr))
            comm = L.Comm(None, g, e)
            h = C.c_void_p()
            L.check(self.lib.nbody_shard_create(C.byref(h), ctx._h, r, world, len(x0), C.byref(comm)))
            plan = L.ShardPlan()
            L.check(self.lib.nbody_shard_get_plan(h, C.byref(plan)))
            ptrs = [C.c_void_p() for _ in range(5)]
            L.check(self.lib.nbody_shard_buffers(h, *[C.byref(p) for p in ptrs]))
            wrap = lambda p, n: torch.as_tensor(nb.sharded._DeviceArray(p.value, max(n, 1)), device="cuda")[:n]
            self.bufs.append(dict(x=wrap(ptrs[0], plan.n_pad), v=wrap(ptrs[1], plan.shard), a=wrap(ptrs[2], plan.shard),
                                  j=wrap(ptrs[3], plan.jbuf_bodies), r=wrap(ptrs[4], plan.rbuf_bodies)))
            L.check(self.lib.nbody_shard_upload(h, C.c_void_p(x0.ctypes.data)))
            self.ctxs.append(ctx); self.shards.append(h); self.plans.append(plan); self._keep.append((g, e, comm))
        self.gathers = self.exchanges = 0

    def _gather(self, r, per_rank):
        self.gathers += 1
        for q in range(self.world):                 # every other rank's advanced block
            if q != r:
                self.bufs[r]["x"][q * per_rank:(q + 1) * per_rank] = self.bufs[q]["x"][q * per_rank:(q + 1) * per_rank]
        return 0

    def _exchange(self, r, send, ns, recv, nr):
        self.exchanges += 1
        for k in range(nr):                         # pull what each peer's plan says it sends to r
            q, off, cnt = recv[k].peer, recv[k].offset, recv[k].count
            pq = self.plans[q]
            src = [pq.send[m] for m in range(pq.n_sends) if pq.send[m].peer == r]
            assert len(src) == 1 and src[0].count == cnt and src[0].body0 == recv[k].body0
            self.bufs[r]["r"][off:off + cnt] = self.bufs[q]["j"][src[0].offset:src[0].offset + cnt]
        return 0

    def step(self, steps):
        for _ in range(steps):
            for phase in range(4):
                torch.cuda.synchronize()            # one thread plays every rank: finish a phase everywhere first
                for h in self.shards:
                    self.nb._lib.check(self.lib.nbody_shard_step_phase(h, phase))
        torch.cuda.synchronize()

    def state(self, n):
        S = self.plans[0].shard
        x = torch.cat([self.bufs[r]["x"][r * S:(r + 1) * S] for r in range(self.world)]).cpu().numpy()[:n]
        v = torch.cat([b["v"] for b in self.bufs]).cpu().numpy()[:n]
        a = torch.cat([b["a"] for b in self.bufs]).cpu().numpy()[:n]
        return x, v, a

    def close(self):
        for h in self.shards:
            self.lib.nbody_shard_destroy(h)
        self.bufs = []


@pytest.mark.parametrize("world,kernel", [(2, "strict"), (3, "strict"), (8, "strict"), (2, "fast"), (3, "fast"), (4, "fast"),
                                          (8, "fast"), (8, "onesided")])
def test_callback_ranks_on_one_device(nb, oracle, world, kernel):
    """The native step with several ranks on one device. STRICT: canonical order, bit-identical to the single-device
    Jacobi oracle at every world size. FAST: the symmetric schedule (every pair once across the ranks, J-side sums
    exchanged), ONESIDED: own block then everybody else — tolerance."""
    n, steps, dt = 2037, 3, 0.1             # not divisible by 2, 3, 4 or 8: padding bodies take part
    x0 = nb.engine.seeded_bodies(n, 0, 17)
    k = {"strict": nb.KERNEL_STRICT, "fast": nb.KERNEL_SYMMETRIC, "onesided": nb.KERNEL_ONESIDED}[kernel]
    ranks = _OneThreadRanks(nb, x0, world, k, dt, 0.002, sym_shape=(1, 2) if kernel == "fast" else None)
    assert ranks.plans[0].schedule == {"strict": 0, "onesided": 1, "fast": 2}[kernel]
    ranks.step(steps)
    x, v, a = ranks.state(n)
    assert ranks.gathers == world * (steps - 1)
    assert ranks.exchanges == (world * steps if kernel == "fast" else 0)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=dt, eps2=0.002, steps=steps)
    if kernel == "strict":
        assert same_bits(x, xo) and same_bits(v, vo) and same_bits(a, ao)
    else:
        assert np.abs(x - xo)[:, :3].max() / 1e5 <= 1e-6
        assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
    ranks.close()


def test_callback_ranks_symmetric_at_block_sizes_the_bench_uses(nb, oracle):
    """4 ranks x 32768 bodies (the default block shapes: 2048-body blocks on both the own-block and the cross
    launches): sampled targets against the CPU over all sources, momentum balance over the whole system."""
    world, n = 4, 131072
    x0 = nb.engine.seeded_bodies(n, 1, 99)
    ranks = _OneThreadRanks(nb, x0, world, nb.KERNEL_FAST, 0.01, 0.002)
    assert ranks.plans[0].schedule == 2 and ranks.ctxs[0].step_info(ranks.plans[0].shard)["symmetric"]
    ranks.step(1)
    x, v, a = ranks.state(n)
    for i0 in (0, 70000, 131072 - 256):
        truth = oracle.accel_range(x0, i0, i0 + 256, 0, n, eps2=0.002, f64acc=True)
        assert np.abs(a[i0:i0 + 256] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * a[:, :3]).sum(0)).max() / (m * np.abs(a[:, :3])).sum() < 1e-6
    ranks.close()
    # the same run again: bit-identical (fixed task lists, fixed order of the slab sums and of the received sums)
    again = _OneThreadRanks(nb, x0, world, nb.KERNEL_FAST, 0.01, 0.002)
    again.step(1)
    x2, v2, a2 = again.state(n)
    again.close()
    assert np.array_equal(a, a2) and np.array_equal(x, x2) and np.array_equal(v, v2)


def test_configs3_decomposition_vs_the_reference_build(nb, oracle):
    """BASELINE configs[3] as the driver runs it — N = 1048576 bodies in EIGHT blocks of 131072, symmetric schedule (own-block halves,
    3 1/2 cross launches per rank, J-side sums exchanged, fixed-order adds), dt = 0.01 — one step through the native sharded step with
    eight callback ranks on this one device, against the REFERENCE build's CPU_compute (DT 0.01f) on 2048 sampled bodies
    (tests/golden/ref_cpu_plummer_n1048576_dt0.01_sample.npz: 78 minutes of one core, generated once). Same bar as the single-GPU test
    of that fixture: the difference stays within the reference's own in-place order and fp32 running-sum rounding, both measured here
    with the pinned restatement, + 1e-5 of max|a|; and the eight-rank result agrees with the single-GPU kernel to 2e-6 of max|a|."""
    from conftest import bits, load_golden
    g = load_golden("ref_cpu_plummer_n1048576_dt0.01_sample.npz")
    n, world = int(g["n"]), 8
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    assert np.array_equal(bits(x0[:8]), bits(g["x0_head"]))
    ranks = _OneThreadRanks(nb, x0, world, nb.KERNEL_FAST, 0.01, 0.002)
    assert ranks.plans[0].schedule == 2 and ranks.plans[0].shard == n // world
    ranks.step(1)
    x, v, a = ranks.state(n)
    ranks.close()
    idx = g["idx"]
    assert np.abs(x[idx] - g["x_1"])[:, :3].max() <= 1e-6
    for i0, i1 in ((0, 256), (n - 256, n)):
        sel = np.searchsorted(idx, np.arange(i0, i1))
        aj = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002)
        at = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002, f64acc=True)
        ar = g["a_1"][sel]
        amax = np.abs(at[:, :3]).max()
        bound = np.abs(aj - ar)[:, :3] + np.abs(aj - at)[:, :3] + 1e-5 * amax
        assert np.all(np.abs(a[i0:i1] - ar)[:, :3] <= bound), i0
        assert np.abs(a[i0:i1] - at)[:, :3].max() / amax <= 1e-5
    one = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    one.run(1)
    a1 = one.state()[2]
    assert np.abs(a[idx] - a1[idx])[:, :3].max() / np.abs(a1[:, :3]).max() <= 2e-6


def test_native_rccl_comm_with_one_rank(nb):
    """nbody_comm_rccl_* (librccl loaded at run time, ncclCommInitRank / ncclAllGather / grouped send-recv) through
    the C entry points, as far as a 1-GPU box allows: a world of one, bit-equal to nbody_step."""
    import ctypes as C
    lib, L = nb.load(), nb._lib
    uid = (C.c_char * 128)()
    L.check(lib.nbody_comm_rccl_unique_id(uid))
    comm = L.Comm()
    L.check(lib.nbody_comm_rccl_create(C.byref(comm), 0, 1, uid, -1))
    n = 20000                                   # >= 16384: the own-block pass is the symmetric kernel
    x0 = nb.engine.seeded_bodies(n, 1, 5)
    ctx = nb.engine.Context(dt=0.01, eps2=0.002)
    h = C.c_void_p()
    L.check(lib.nbody_shard_create(C.byref(h), ctx._h, 0, 1, n, C.byref(comm)))
    L.check(lib.nbody_shard_upload(h, C.c_void_p(x0.ctypes.data)))
    L.check(lib.nbody_shard_step(h, 3))
    out = [np.zeros((n, 4), np.float32) for _ in range(3)]
    L.check(lib.nbody_shard_download(h, *[C.c_void_p(o.ctypes.data) for o in out]))
    # the collectives themselves, on the shard's own buffers (a world of one: both leave the data as it is)
    ptrs = [C.c_void_p() for _ in range(5)]
    L.check(lib.nbody_shard_buffers(h, *[C.byref(p) for p in ptrs]))
    stream = torch.cuda.current_stream().cuda_stream
    assert comm.all_gather(comm.user, ptrs[0], n, stream) == 0
    assert comm.exchange(comm.user, None, 0, ptrs[3], None, 0, ptrs[4], stream) == 0
    torch.cuda.synchronize()
    out2 = [np.zeros((n, 4), np.float32) for _ in range(3)]
    L.check(lib.nbody_shard_download(h, *[C.c_void_p(o.ctypes.data) for o in out2]))
    assert all(np.array_equal(p, q) for p, q in zip(out, out2))
    L.check(lib.nbody_shard_destroy(h))
    L.check(lib.nbody_comm_rccl_destroy(C.byref(comm)))
    sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    sim.run(3)
    for got, want in zip(out, sim.state()):
        assert np.array_equal(got, want)



def _gloo_gpu_worker(rank, world, port, n, steps, kernel, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    mark("init_process_group gloo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd
        x0 = nbody_amd.engine.seeded_bodies(n, 1, 77)
        mark("ShardedSimulation")
        sim = nbody_amd.sharded.ShardedSimulation(x0, dt=0.01, eps2=0.002, kernel=kernel, device=torch.device("cuda", 0),
                                                  sym_waves=1, sym_bpl=2)      # every rank on the one GPU
        sim.comm_timing(True)
        mark(f"step({steps})")
        sim.step(steps)
        mark("gather_state")
        x, v, a = sim.gather_state()
        mark("comm_report")
        q.put((rank, x, v, a, sim.comm_report()))
        mark("close")
        sim.close()
    finally:
        mark("destroy_process_group")
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,kernel", [(2, 6000, 3), (3, 5001, 3), (4, 7000, 3), (2, 3000, 1)])
def test_sharded_simulation_multi_rank_over_gloo(nb, oracle, world, n, kernel):
    """The product path end to end (ShardedSimulation -> nbody_shard_* -> callbacks -> torch.distributed) with
    several processes; the box has one GPU, so all ranks share it and the collectives go over gloo instead of RCCL.
    kernel 3 = symmetric schedule with the exchange of J-side sums, 1 = strict (canonical order, bit-exact)."""
    steps = 3
    res = run_ranks(_gloo_gpu_worker, world, (world, _free_port(), n, steps, kernel))
    x0 = nb.engine.seeded_bodies(n, 1, 77)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.01, eps2=0.002, steps=steps)
    for rank, x, v, a, rep in res:
        if kernel == 1:
            assert same_bits(x, xo) and same_bits(v, vo) and same_bits(a, ao)
        else:
            assert np.abs(x - xo)[:, :3].max() <= 1e-6
            assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
        assert np.array_equal(x, res[0][1])
        assert rep["steps"] == steps and rep["schedule"] == ("canonical" if kernel == 1 else "symmetric")
        # mean and maximum over the timed steps (the first step after the upload carries no all-gather)
        assert rep["gathers"] == steps - 1 and rep["records_kept"] == steps
        assert rep["all_gather_ms_max"] >= rep["all_gather_ms_avg"] > 0 and rep["exposed_ms_max"] >= rep["exposed_ms_avg"] >= 0
        if kernel != 1:
            assert rep["exchanges"] == steps and rep["exchange_ms_max"] >= rep["exchange_ms_avg"] > 0
            assert rep["exchange_exposed_ms_max"] >= rep["exchange_exposed_ms_avg"] >= 0


def test_comm_timing_keeps_a_ring_of_step_records(nb):
    """200 timed steps of a two-rank machine (both ranks driven from this thread, the collectives plain device copies): the shard
    keeps 64 step records, folds the older ones into its sums, and the report still covers all 200 steps — mean and maximum."""
    import ctypes as C
    L = nb._lib
    n, world, steps = 4096, 2, 200
    x0 = nb.engine.seeded_bodies(n, 1, 5)
    m = _OneThreadRanks(nb, x0, world, nb.KERNEL_SYMMETRIC, 0.01, 0.002, sym_shape=(1, 2))
    for sh in m.shards:
        L.check(m.lib.nbody_shard_comm_timing(sh, 1))
    m.step(steps)
    for sh in m.shards:
        r = L.CommReport()
        L.check(m.lib.nbody_shard_comm_report_ex(sh, C.byref(r)))
        assert r.steps == steps and r.records_kept == 64 and r.gathers == steps - 1 and r.exchanges == steps
        assert r.gather_ms_max >= r.gather_ms >= 0 and r.exchange_ms_max >= r.exchange_ms >= 0      # (these callbacks copy on torch's stream: the bracketed interval is empty)
        assert r.gather_exposed_ms_max >= r.gather_exposed_ms >= 0 and r.exchange_exposed_ms_max >= r.exchange_exposed_ms >= 0
        k = C.c_int()
        g, ge, x, xe = (C.c_double() for _ in range(4))
        L.check(m.lib.nbody_shard_comm_report(sh, C.byref(k), C.byref(g), C.byref(ge), C.byref(x), C.byref(xe)))   # the means-only form agrees
        assert k.value == steps and g.value == r.gather_ms and xe.value == r.exchange_exposed_ms
        L.check(m.lib.nbody_shard_comm_timing(sh, 1))                                  # starts afresh
        L.check(m.lib.nbody_shard_comm_report_ex(sh, C.byref(r)))
        assert r.steps == 0 and r.gathers == 0 and r.gather_ms == 0
    m.close()


def _nccl_worker(rank, world, port, n, steps, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    mark("init_process_group nccl")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import nbody_amd
        x0 = nbody_amd.engine.seeded_bodies(n, 1, 77)
        mark("ShardedSimulation")
        sim = nbody_amd.sharded.ShardedSimulation(x0, dt=0.01, eps2=0.002, device=dev)
        sim.comm_timing(True)
        mark(f"step({steps})")
        sim.step(steps)
        mark("gather_state")
        x, v, a = sim.gather_state()
        q.put((rank, x, v, a, sim.comm_report()))
        mark("close")
        sim.close()
    finally:
        mark("destroy_process_group")
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs one GPU per rank (RCCL refuses two ranks on one device)")
def test_sharded_simulation_over_rccl_one_gpu_per_rank(nb, oracle):
    """Only on a node with several GPUs: the product path over RCCL itself (in-place all-gather of positions, grouped
    send/recv of the J-side sums), one process per GPU, against the CPU on sampled targets."""
    world = min(torch.cuda.device_count(), 5)     # (a GPU box admits six processes on its cards at once: this one + five ranks)
    n, steps = 65536, 2
    res = run_ranks(_nccl_worker, world, (world, _free_port(), n, steps), deadline_s=300)
    sim = nb.engine.Simulation(nb.engine.seeded_bodies(n, 1, 77), dt=0.01, eps2=0.002)
    sim.run(steps)
    x1, v1, a1 = sim.state()
    for rank, x, v, a, rep in res:
        assert np.abs(x - x1)[:, :3].max() <= 1e-6
        assert np.abs(a - a1)[:, :3].max() / np.abs(a1[:, :3]).max() <= 2e-5
        assert np.array_equal(x, res[0][1]) and rep["steps"] == steps and rep["schedule"] == "symmetric"


def test_bench_two_ranks_over_gloo_on_one_gpu():
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run, one process per
    rank; the launcher starts before anything touches the GPU), rehearsed with 2 ranks sharing this box's one
    GPU over gloo: one JSON line, n_gpus 2, strong scaling, the census, the in-run parity of the sharded step against the
    single-GPU kernel, the cross-rank bitwise check, the per-step communication report and the same-run single-GPU point."""
    line = run_bench(torchrun(2, "--backend", "gloo", "--bodies", "65536", "--steps", "3", "--warmup", "2", "--repeats", "2"))
    assert line["steps"] == 3 and line["warmup"] == 2 and line["repeats"] == 2
    check_multi_gpu_line(line, 2, 65536, "torch", distinct=False)       # two ranks share the one GPU here
    assert line["config"]["rccl"]["backend"] == "gloo"


@pytest.mark.parametrize("comm", ["torch", "native"])
def test_bench_sharded_path_with_one_rank_over_rccl(comm):
    """The multi-GPU code path of bench.py over REAL RCCL as far as one GPU allows without tricks: one rank
    (--force-sharded): nccl process group, census (one distinct device), the sharded step through torch's RCCL ops or
    through the library's own communicator, the in-run checks."""
    line = run_bench(torchrun(1, "--force-sharded", "--comm", comm, "--bodies", "65536", "--steps", "3", "--warmup", "2",
                              "--repeats", "2"))
    check_multi_gpu_line(line, 1, 65536, comm, distinct=True)
    assert line["config"]["rccl"]["backend"] == "nccl"
    assert line["config"]["multi_gpu_check"]["max_rel_da"] < 2e-5
    assert line["value"] > 1e12            # one rank alone on the GPU: a rate floor means something here (5-6e12 expected)

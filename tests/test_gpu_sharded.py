"""The sharded step on the GPU: a 1-rank RCCL group (the only world size a 1-GPU box allows), and
the 2- and 3-block decomposition driven by hand on one device (same C-ABI calls a rank makes)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from conftest import same_bits

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
        sim.run(4)
        x, v, a = sim.state()
        assert np.abs(xs - x)[:, :3].max() <= 1e-6
        assert np.abs(as_ - a)[:, :3].max() / np.abs(a[:, :3]).max() <= 1e-5
        # the RCCL calls of the multi-rank path, as far as one rank can exercise them: the in-place
        # all_gather_into_tensor on the communication stream, the event hand-over, MAX all-reduce, barrier
        before = sh.x.clone()
        sh.backend.all_gather(sh.x, sh.i0, sh.i1, None)
        sh.backend.wait_gather()
        sh.sync()
        assert torch.equal(sh.x, before)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(t.item()) == 1.5
        sh.step(2)                       # and stepping still works after a gather
        sh.sync()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("blocks,kernel", [(2, "strict"), (3, "strict"), (2, "fast"), (8, "fast")])
def test_block_decomposition_on_one_device(nb, oracle, blocks, kernel):
    """Each 'rank' = one context on the same GPU; the all-gather is a device copy. Strict kernel in
    canonical block order is bit-identical to the single-device Jacobi oracle."""
    n, steps, dt = 2040, 2, 0.1   # divisible by 2, 3 and 8
    x0 = nb.engine.seeded_bodies(n, 0, 17)
    k = nb.KERNEL_STRICT if kernel == "strict" else nb.KERNEL_FAST
    S = n // blocks
    ranks = []
    for r in range(blocks):
        ctx = nb.engine.Context(dt=dt, kernel=k)
        ranks.append(dict(ctx=ctx, x=torch.from_numpy(x0).cuda(), v=torch.zeros((S, 4), device="cuda"),
                          a=torch.zeros((S, 4), device="cuda"), i0=r * S, i1=(r + 1) * S))
    for _ in range(steps):
        for R in ranks:
            c = R["ctx"]
            if kernel == "strict":   # canonical order: the running sum continues block after block
                c.accel_range(R["x"], R["a"], R["i0"], R["i1"], 0, n, False)
            else:                    # the schedule of sharded.py: own block, then before, then after
                c.accel_range(R["x"], R["a"], R["i0"], R["i1"], R["i0"], R["i1"], False)
                if R["i0"] > 0:
                    c.accel_range(R["x"], R["a"], R["i0"], R["i1"], 0, R["i0"], True)
                if R["i1"] < n:
                    c.accel_range(R["x"], R["a"], R["i0"], R["i1"], R["i1"], n, True)
            c.integrate_range(R["x"], R["v"], R["a"], R["i0"], R["i1"])
            c.sync()
        # "all-gather": every rank receives every other rank's advanced block
        for R in ranks:
            for Q in ranks:
                if Q is not R:
                    R["x"][Q["i0"]:Q["i1"]] = Q["x"][Q["i0"]:Q["i1"]]
        torch.cuda.synchronize()
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=dt, eps2=0.002, steps=steps)
    xg = ranks[0]["x"].cpu().numpy()
    ag = torch.cat([R["a"] for R in ranks]).cpu().numpy()
    vg = torch.cat([R["v"] for R in ranks]).cpu().numpy()
    if kernel == "strict":
        assert same_bits(xg, xo) and same_bits(vg, vo) and same_bits(ag, ao)
    else:
        assert np.abs(xg - xo)[:, :3].max() / 1e5 <= 1e-6
        assert np.abs(ag - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
    for R in ranks[1:]:
        assert np.array_equal(R["x"].cpu().numpy(), xg)


def _gloo_gpu_worker(rank, world, port, n, steps, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd
        x0 = nbody_amd.engine.seeded_bodies(n, 1, 77)
        be = nbody_amd.sharded.HipBackend(torch.device("cuda", 0), 0.01, 0.002)   # every rank on the one GPU
        sim = nbody_amd.sharded.ShardedSimulation(x0, dt=0.01, eps2=0.002, backend=be)
        sim.step(steps)
        x, v, a = sim.gather_state()
        q.put((rank, x, v, a))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 6000), (3, 5001)])
def test_sharded_hip_backend_multi_rank_over_gloo(nb, oracle, world, n):
    """The real HipBackend (streams, events, C-ABI calls, padding) with several ranks; the box has one
    GPU, so all ranks share it and the collective goes over gloo instead of RCCL."""
    import torch.multiprocessing as mp
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_gpu_worker, args=(r, world, port, n, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x0 = nb.engine.seeded_bodies(n, 1, 77)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.01, eps2=0.002, steps=steps)
    for rank, x, v, a in res:
        assert np.abs(x - xo)[:, :3].max() <= 1e-6
        assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
        assert np.array_equal(x, res[0][1])


def _run_bench(args, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


def test_bench_two_ranks_over_gloo_on_one_gpu():
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run, one process per
    rank; the launcher starts before anything touches the GPU), rehearsed with 2 ranks sharing this box's one
    GPU over gloo: one JSON line, n_gpus 2, strong scaling, the per-step communication report present."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                       "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                       "--bodies", "65536", "--steps", "3", "--warmup", "1", "--repeats", "2"])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["repeats"] == 2
    assert line["scaling"] == "strong" and line["config"]["n_bodies"] == 65536
    assert line["config"]["comm_rank0"]["steps"] >= 3
    assert line["value"] > 1e11 and 0 < line["roofline"]["frac"] < 1

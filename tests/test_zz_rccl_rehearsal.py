"""REHEARSALS of the multi-rank RCCL path on a ONE-GPU box — collected LAST (file name + conftest's ordering hook), after every
oracle comparison and drop-in test, so that nothing here can mask a parity result.

Every rank reports its own NCCL_HOSTID, so RCCL takes the ranks for different hosts (no "Duplicate GPU") and moves the data
over its socket transport on the loopback interface. Slow transport, real library: ncclCommInitRank with world > 1, the
in-place ncclAllGather of positions and the grouped ncclSend/ncclRecv of the J-side sums run for real — through
torch.distributed (comm torch) and through nbody_comm_rccl_* (comm native). What these tests establish is the call sequence,
the buffer arithmetic and the stream ordering (parity against the CPU oracle / the single-GPU kernel, bit-identical positions
on every rank). They assert NO rate: several processes time-share one GPU's hardware queues. A stall is a
FAILURE that quotes every rank's last phase and Python stacks — never a skip."""
import os
import signal
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

from _ranks import ROOT, check_multi_gpu_line, free_port, mark, run_bench, run_ranks, torchrun

pytestmark = [pytest.mark.gpu, pytest.mark.rehearsal]


def _fake_hosts_work():
    """Can RCCL run several ranks on this box's one GPU when every rank reports its own NCCL_HOSTID (tools/rccl_hostid_probe.py)?"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), os.path.join(ROOT, "tools", "rccl_hostid_probe.py")],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)          # the session started above: the launcher and its two ranks
        out, err = p.communicate()
        return None, "probe stalled: " + (out + err)[-1200:]
    return p.returncode == 0, (out + err)[-1500:]


@pytest.fixture(scope="module")
def fake_hosts():
    ok, log = _fake_hosts_work()
    if ok is None:
        pytest.fail("the 2-rank RCCL probe (tools/rccl_hostid_probe.py) stalled: " + log)
    if not ok:
        pytest.skip("RCCL does not accept several ranks on one GPU here, even with distinct NCCL_HOSTIDs: " + log[-300:])
    return True


@pytest.mark.parametrize("comm,world", [("torch", 2), ("native", 2), ("native", 3)])
def test_bench_multi_rank_over_real_rccl_on_one_gpu(fake_hosts, comm, world):
    """bench.py as the driver launches it, 2 and 3 ranks over real RCCL: the line certifies itself (census, parity vs the
    single-GPU kernel, bit-identical positions on all ranks, same-run single-GPU point)."""
    line = run_bench(torchrun(world, "--fake-hosts", "--comm", comm, "--bodies", "49152", "--steps", "2", "--warmup", "2",
                              "--repeats", "2"))
    check_multi_gpu_line(line, world, 49152, comm, distinct=False)
    r = line["config"]["rccl"]
    assert r["backend"] == "nccl" and r["fake_hosts"]
    assert line["config"]["comm_rank0"]["all_gather_ms_avg"] > 0 and line["config"]["comm_rank0"]["exchange_ms_avg"] > 0


def test_bench_started_directly_launches_its_own_ranks(fake_hosts):
    """`python bench.py --gpus 2 ...` with NO launcher in the command (the form the driver uses at --gpus 1, extended to G > 1): bench.py
    starts `python -m torch.distributed.run ... bench.py --gpus 2 ...` as a child, relays its output and exit code. One line,
    n_gpus 2, RCCL world 2; the headline is the general pair arithmetic, the equal-mass path of the same run beside it."""
    line = run_bench([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--fake-hosts", "--bodies", "49152", "--steps", "2", "--warmup", "2", "--repeats", "2"])
    check_multi_gpu_line(line, 2, 49152, "torch", distinct=False)
    assert line["n_gpus"] == 2 and line["config"]["rccl"]["world"] == 2 and line["config"]["rccl"]["backend"] == "nccl"
    assert line["equal_mass_path"] is False and line["roofline"]["frac_path"] == "general pair arithmetic"


def test_bench_compares_stream_priorities_in_the_run(fake_hosts):
    """--comm-priority ab (what `auto` does on a GPU per rank with the library's communicator): a few untimed steps at normal and at
    the greatest stream priority, the faster kept on every rank, both timings in the line. Two ranks: the high priority is not
    pathological there, so either outcome is legitimate; the fields and the usual self-certification are checked."""
    line = run_bench(torchrun(2, "--fake-hosts", "--comm", "native", "--comm-priority", "ab", "--bodies", "49152", "--steps", "2", "--warmup", "2",
                              "--repeats", "2"))
    check_multi_gpu_line(line, 2, 49152, "native", distinct=False)
    ab = line["config"]["rccl"]["comm_priority_ab"]
    assert set(ab["ms_per_step"]) == {"normal", "high"} and all(v > 0 for v in ab["ms_per_step"].values())
    assert ab["kept"] == line["config"]["rccl"]["comm_priority"] and ab["kept"] in ("normal", "high")
    if ab["kept"] == "high":
        assert ab["ms_per_step"]["high"] < 0.98 * ab["ms_per_step"]["normal"]


def _nccl_fake_host_worker(rank, world, port, n, steps, comm, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NCCL_HOSTID"] = f"nbody-test-host-{rank}"
    os.environ["NCCL_SOCKET_IFNAME"] = "lo"
    os.environ["NCCL_IB_DISABLE"] = "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    mark("init_process_group nccl")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import nbody_amd
        x0 = nbody_amd.engine.seeded_bodies(n, 1, 77)
        mark(f"ShardedSimulation comm={comm}")
        sim = nbody_amd.sharded.ShardedSimulation(x0, dt=0.01, eps2=0.002, device=dev, comm=comm, sym_waves=1, sym_bpl=2)
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


@pytest.mark.parametrize("comm,world,n", [("torch", 2, 6000), ("native", 2, 6000), ("native", 4, 7001), ("torch", 3, 5001)])
def test_sharded_simulation_over_real_rccl_on_one_gpu(nb, oracle, fake_hosts, comm, world, n):
    """The product path (ShardedSimulation -> nbody_shard_* -> RCCL) with several ranks over REAL RCCL, symmetric schedule with
    the exchange of J-side sums, against the CPU oracle; identical positions on every rank."""
    steps = 3
    res = run_ranks(_nccl_fake_host_worker, world, (world, free_port(), n, steps, comm))
    x0 = nb.engine.seeded_bodies(n, 1, 77)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.01, eps2=0.002, steps=steps)
    for rank, x, v, a, rep in res:
        assert np.abs(x - xo)[:, :3].max() <= 1e-6
        assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
        assert np.array_equal(x, res[0][1]) and np.array_equal(a, res[0][3])
        assert rep["steps"] == steps and rep["schedule"] == "symmetric"

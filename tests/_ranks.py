"""Runners for the multi-process tests (several ranks of the sharded step, bench.py under the launcher).

The rules they implement: the parent never waits for a result that cannot come (a rank's exception or death fails the test
at once); a run that is still silent at its deadline — kept below the 7 minutes after which a GPU box takes a command for
hung — is ended (exactly the processes started here) and FAILS with what the ranks were doing: every rank keeps a progress
file (`mark()`: the phase it entered last) and arms a faulthandler dump of all its Python stacks. A stall is never a skip."""
import json
import os
import queue
import signal
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class RanksStalled(AssertionError):
    """The rank processes neither finished nor failed within the deadline (a test FAILURE: the message carries each rank's
    last phase and its Python stacks)."""


_progress_path = None


def mark(phase):
    """Rank side: note the phase this rank is entering (one line, appended and flushed; the parent quotes the tail on a stall)."""
    if _progress_path:
        with open(_progress_path, "a") as f:
            f.write(f"{time.time():.3f} {phase}\n")


def _guarded(worker, rank, args, q, dump_dir, stacks_after_s):
    """Rank process body: a Python exception travels to the parent through the queue, and a rank that is still running when the
    parent's deadline is near writes all its stacks to a file the parent quotes."""
    import faulthandler
    import traceback
    global _progress_path
    _progress_path = os.path.join(dump_dir, f"rank{rank}.progress")
    f = open(os.path.join(dump_dir, f"rank{rank}.stacks"), "w")
    faulthandler.dump_traceback_later(stacks_after_s, repeat=False, file=f)
    faulthandler.register(signal.SIGUSR1, file=f, all_threads=True)      # the parent asks a lingering rank for its stacks
    mark("started")
    try:
        worker(rank, *args, q)
        mark("worker returned")
    except BaseException:
        q.put(("error", rank, traceback.format_exc()))
        raise
    finally:
        faulthandler.cancel_dump_traceback_later()


def _tail(path, n=1500):
    try:
        return open(path).read()[-n:]
    except OSError:
        return "(none)\n"


def run_ranks(worker, world, args, deadline_s=240):
    """Starts `world` rank processes (spawn) running worker(rank, *args, q) and returns their results sorted by rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    dump_dir = tempfile.mkdtemp(prefix="nbody_ranks_")
    procs = [ctx.Process(target=_guarded, args=(worker, r, args, q, dump_dir, max(deadline_s - 20, 10))) for r in range(world)]
    for p in procs:
        p.start()
    res, t0, problem = [], time.time(), None
    while len(res) < world and problem is None:
        try:
            item = q.get(timeout=1.0)
            if item[0] == "error":
                problem = f"rank {item[1]} raised:\n{item[2]}"
            else:
                res.append(item)
        except queue.Empty:
            dead = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode not in (None, 0)]
            if dead:
                problem = f"rank process(es) died without a result: {dead}"
            elif time.time() - t0 > deadline_s:
                problem = "stalled"

    def end_all():
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.kill()
                p.join(10)

    def report():
        out = ""
        for r in range(world):
            out += f"--- rank {r}: progress\n{_tail(os.path.join(dump_dir, f'rank{r}.progress'), 600)}"
            out += f"--- rank {r}: stacks\n{_tail(os.path.join(dump_dir, f'rank{r}.stacks'))}"
        return out

    if problem is None:
        # every rank has delivered; what is left is tear-down (shard, communicator, process group, interpreter exit) — part of the
        # product path too: a rank that does not get through it within a minute fails the test with its stacks
        t1 = time.time()
        for p in procs:
            p.join(max(1.0, 60.0 - (time.time() - t1)))
        late = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode != 0]
        if late:
            for r, p in enumerate(procs):      # ask the lingering ranks for their stacks before ending them
                if p.is_alive():
                    try:
                        os.kill(p.pid, signal.SIGUSR1)
                    except OSError:
                        pass
            time.sleep(1.0)
            end_all()
            raise RanksStalled(f"every rank delivered its result, but tear-down did not finish cleanly within 60 s (rank, exit code): {late}\n{report()}")
        return sorted(res, key=lambda t: t[0])
    end_all()
    if problem == "stalled":
        raise RanksStalled(f"{len(res)} of {world} ranks reported within {deadline_s} s\n{report()}")
    raise AssertionError(problem + "\n" + report())


def run_bench(args, timeout=300):
    """bench.py (through the launcher) as a child in a session of its own; at the deadline the whole process group started here
    is ended and the test FAILS with the children's stderr — bench.py arms a faulthandler dump (NBODY_BENCH_STACKS_AFTER) and names
    its current phase there, so the stall locates itself."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env["NBODY_BENCH_STACKS_AFTER"] = str(max(timeout - 30, 10))
    p = subprocess.Popen([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)          # the launcher and its ranks: the session created above, nothing else
        out, err = p.communicate()
        raise RanksStalled(f"bench.py did not finish within {timeout} s: {' '.join(args[-12:])}\n{err[-4000:]}")
    assert p.returncode == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]          # rank 0 prints ONE JSON line, the other ranks nothing
    assert [ln for ln in out.splitlines() if ln.strip()] == lines, out[:2000]   # ... and nothing else reaches stdout (RCCL's banner goes to stderr)
    return json.loads(lines[0])


def torchrun(nproc, *bench_args):
    return ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *bench_args]


def check_multi_gpu_line(line, world, n, comm, distinct):
    """The self-certification fields of a multi-rank bench line (bench.py docstring: census, parity, cross-rank). FUNCTIONAL checks
    only: ranks that share one GPU (rehearsals) say nothing about throughput, so no rate floor is asserted here."""
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["config"]["n_bodies"] == n
    assert abs(line["per_gpu_value"] * world - line["value"]) <= 1e-6 * line["value"]
    r = line["config"]["rccl"]
    assert r["world"] == world and r["ranks_seen"] and r["comm"] == comm and len(r["devices"]) == world
    assert [d["rank"] for d in r["devices"]] == list(range(world)) and r["distinct_devices"] == distinct
    # (the marketing name is not reliable on these boxes — amdgpu.ids is absent and torch reports "AMD Radeon Graphics" — the ISA and CU count are)
    assert all(str(d.get("gcnArchName", "")).startswith("gfx950") and d["cus"] == 256 for d in r["devices"]), r["devices"]
    c = line["config"]["multi_gpu_check"]
    assert c["finite"] and c["x_bitwise_equal_across_ranks"] and 0 <= c["max_rel_da"] <= c["tolerance"] == 5e-5
    assert c["sampled_bodies_per_rank"] >= 1024 and c["steps_checked"] >= 1
    rm = c["random_masses"]            # the same step on bodies of UNEQUAL masses: mass-weighted J-side sums through the exchange
    assert rm["finite"] and 0 <= rm["max_rel_da"] <= 5e-5
    assert r["comm_priority"] in ("high", "normal")
    if world > 1:                      # the same-N single-GPU point of the scaling series comes from THIS run, never from a file
        s1 = line["single_gpu_same_n"]
        assert s1["measured_in_this_run"] is True and s1["n_bodies"] == n and s1["value"] > 0 and s1["ms_per_step"] > 0
    else:
        assert "single_gpu_same_n" not in line
    assert line["config"]["comm_rank0"]["steps"] >= line["steps"]
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1 and 0 < line["roofline"]["frac_evaluated"] <= line["roofline"]["frac"]
    # the launch description is what the own-block pass really launches: block pairs (never "runs") when it is issued in parts
    if world > 1 and line["config"]["launch"]["schedule"] == "symmetric":
        assert line["config"]["launch"]["symmetric"] and not line["config"]["launch"]["runs"]

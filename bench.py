#!/usr/bin/env python3
"""bench.py — body-pair interactions/s of the all-pairs step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bodies B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
Both forms work for N > 1: started without the launcher, `bench.py --gpus N` starts that launcher line itself as a child process
(one rank per GPU, rendezvous on 127.0.0.1 and a free port), relays its output and ends with its exit code.

A "step" is one pass of the hot path over all bodies: the O(N^2) force accumulation followed by
the half-kick/drift integrate (TestProject/kernel.cu:80-130 in the reference).

Workload (BASELINE.json):
  --gpus 1   N = 262144 bodies, fp32 (configs[2], the size the metric is quoted on).
  --gpus G>1 N = 1048576 bodies (configs[3]) block-partitioned over the G ranks, positions all-gathered
             once per step (RCCL); STRONG scaling: the same system at every G >= 2. The 1-GPU point of THAT series is
             `--gpus 1 --bodies 1048576` (the default 1-GPU run is configs[2]; an offline figure for it is reported
             as `single_gpu_same_n`). `--scaling weak` instead runs N(G) = 262144*sqrt(G) rounded to 8192*G bodies
             (equal pairs per GPU, the default 1-GPU run being its first point).
  --bodies overrides N; --dtype f64 is configs[4] (single GPU).

Multi-GPU runs certify themselves BEFORE the timed loop (this process is the only one that ever sees 8 GPUs):
  census   every rank's device (PCI bus id, uuid, name) gathered and required to be distinct: `config.rccl`;
  parity   one untimed sharded step, then every rank compares its own-block accelerations on 4096 sampled bodies with the
           SINGLE-GPU one-sided kernel over all N sources (product kernel vs product kernel, no CPU checker involved);
  cross    after the warm-up steps one more all-gather, then per-block checksums of every rank's position array must be
           identical on all ranks (bit-identical copies): `config.multi_gpu_check`. A failed check exits non-zero.
`--comm torch` runs the two collectives through torch.distributed (backend nccl = RCCL); `--comm native` through the
library's own RCCL communicator (nbody_comm_rccl_*, no Python in the step), its id broadcast over the process group.

Timing: the CPU baseline first (rank 0, single GPU runs), then W warm-up steps, then R repeats (enough for >= 10 s of timed
GPU work, 3 <= R <= 64; --repeats overrides) of EXACTLY K steps each, every repeat bracketed by a barrier + device
synchronise on both sides and reduced with MAX over ranks. `ms_per_step` and `value` come from the MEDIAN repeat; min/max
are reported beside it. `value` = N^2 * K / (median repeat time): interactions applied per second, inputs resident in HBM
before the first repeat starts.

`roofline` is the force kernel's algorithmic FLOP rate — 20 FLOP per interaction times
the N^2 interactions one launch applies — over its own HIP-event time on its launch stream, against the
157.3 TFLOP/s fp32 vector peak. The symmetric kernel EVALUATES each unordered pair once (about N^2/2
evaluations for the same N^2 interactions); `evaluated_pairs_per_launch` and `frac_evaluated` state the
figure on that count as well. `roofline` also carries THE CLOCK the number was measured at: around every timed force launch one tiny
launch per side reads each XCD's shader-cycle counter and the 100-MHz counter (nbody_ctx_timing(ctx, 2); no stamp inside the
measured kernel), giving `kernel_cycles_per_launch` (a property of the CODE: about 2.43e7 at N = 262144 on every box),
`sclk_mhz_under_load` (a property of the BOX: what the part held under this load) and `frac_at_measured_clock` (achieved over the
peak scaled to that clock) beside the unchanged `frac` — a slow box shows in the clock, slow code in the cycles. `cpu_baseline` times the reference's own CPU path (oracle/_ref, kind
"reference") or the checker's restatement (kind "port") on a bounded sample on the host cores — a
reported baseline, not the product.
"""
import argparse
import fcntl
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PHASE = ["start"]                # what this rank is doing (named by the stall dump and by the tear-down watchdog)


def phase(name: str) -> None:
    PHASE[0] = name


def arm_stall_dump() -> None:
    """NBODY_BENCH_STACKS_AFTER=<seconds>: a run still alive after that long writes its phase and all its Python stacks to
    stderr (the multi-rank tests set it just below their own deadline, so a stall locates itself)."""
    after = os.environ.get("NBODY_BENCH_STACKS_AFTER")
    if not after:
        return
    import faulthandler
    import threading

    def say():
        print(f"[bench.py rank {os.environ.get('RANK', '0')} pid {os.getpid()}] still running after {after} s in phase: {PHASE[0]}",
              file=sys.stderr, flush=True)
    t = threading.Timer(float(after), say)
    t.daemon = True
    t.start()
    faulthandler.dump_traceback_later(float(after) + 1.0, repeat=False, file=sys.stderr)


FLOP_PER_PAIR = 20.0             # the agreed algorithmic count
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
FP64_VECTOR_PEAK_TFLOPS = 78.6
N_SINGLE = 262144                # BASELINE.json configs[2]
N_MULTI = 1048576                # BASELINE.json configs[3]


def weak_n(gpus: int) -> int:
    if gpus == 1:
        return N_SINGLE
    # every rank's block a multiple of 8192 bodies
    q = 8192 * gpus
    return int(round(N_SINGLE * math.sqrt(gpus) / q)) * q


def default_workload(world: int, bodies: int, scaling: str):
    """(N, scaling label) for a run on `world` GPUs: BASELINE.json configs[2] on one GPU, configs[3] strong-scaled on
    several, unless --bodies / --scaling weak say otherwise. The label follows --scaling at every G (a 1-GPU run is a
    point of whichever series it is compared with): "strong" = the same N at every G (N = 1048576 for G >= 2; its 1-GPU
    point is `--gpus 1 --bodies 1048576`), "weak" = N(G) = 262144*sqrt(G), per-GPU pair count fixed."""
    if bodies:
        return bodies, "strong"
    if scaling == "weak":
        return weak_n(world), "weak"
    if world == 1:
        return N_SINGLE, "strong"
    return N_MULTI, "strong"


def device_census(dev):
    """What this rank runs on, for the cross-rank census (config.rccl.devices)."""
    import torch
    p = torch.cuda.get_device_properties(dev)
    rec = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "index": dev.index,
           "name": p.name, "cus": p.multi_processor_count, "hbm_gib": round(p.total_memory / 2 ** 30, 1)}
    for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id", "gcnArchName"):
        if hasattr(p, k):
            v = getattr(p, k)
            rec[k] = v if isinstance(v, (int, str)) else str(v)
    rec["visible"] = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or ""
    rec["pid"] = os.getpid()
    return rec


def device_key(rec):
    """Identity of the physical device behind a census record."""
    if "uuid" in rec and rec["uuid"] not in ("", "None"):
        return ("uuid", rec["uuid"])
    return ("pci", rec.get("pci_domain_id"), rec.get("pci_bus_id"), rec.get("pci_device_id"), rec.get("visible"), rec["index"])


def ensure_built() -> None:
    """Build the in-tree binaries if a snapshot lacks them — under an exclusive file lock, so that with
    several ranks starting at once exactly one compiles and the others wait for the finished library
    (never a half-written one). Runs before anything touches the GPU or the process group."""
    import nbody_amd
    need = [nbody_amd._lib.LIB_PATH]
    if all(os.path.exists(p) for p in need):
        return
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    with open(os.path.join(ROOT, "build", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not all(os.path.exists(p) for p in need):
                import __graft_entry__
                __graft_entry__.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def rank_launch_command(gpus: int, argv, port: int):
    """What `python bench.py --gpus G ...` runs when nobody wrapped it in the launcher: the driver's own form, one rank per GPU."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def spawn_ranks(gpus: int) -> int:
    """Starts the ranks as a CHILD process (never an exec: this process may not be replaced once a GPU runtime is loaded, and a child
    keeps the exit code honest) in a session of its own, lets them write to this process's stdout / stderr, forwards SIGTERM / SIGINT to
    the whole group, returns their exit code. However this function is left (an exception, a signal), the group is ended: first
    SIGTERM, after a grace period SIGKILL — no launcher and no rank outlives this process to hold the GPUs."""
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL between processes needs it)
    env.setdefault("OMP_NUM_THREADS", "1")              # what the launcher would set itself, with a warning
    child = {"proc": None, "pending": None}

    import ctypes
    libc = ctypes.CDLL("libc.so.6", use_errno=True)       # loaded HERE: nothing is imported between fork and exec (this process may have threads)
    sigterm = int(signal.SIGTERM)

    def die_with_parent():                               # in the child, before exec: a parent that is SIGKILLed takes the launcher with it
        libc.prctl(1, sigterm)                           # PR_SET_PDEATHSIG

    def signal_group(signum) -> None:
        proc = child["proc"]
        if proc is None or proc.poll() is not None:
            return
        try:
            os.killpg(proc.pid, signum)                  # start_new_session: the child's pid is its process group
        except (ProcessLookupError, PermissionError):
            pass

    def forward(signum, _frame):
        if child["proc"] is None:
            child["pending"] = signum                   # arrived before the child existed: delivered right after Popen
        else:
            signal_group(signum)
    for sg in (signal.SIGTERM, signal.SIGINT):          # BEFORE Popen: no window in which a signal ends the parent alone
        signal.signal(sg, forward)
    try:
        child["proc"] = subprocess.Popen(rank_launch_command(gpus, sys.argv[1:], port), env=env, start_new_session=True, preexec_fn=die_with_parent)
        if child["pending"] is not None:
            signal_group(child["pending"])
        return child["proc"].wait()
    finally:
        proc = child["proc"]
        if proc is not None and proc.poll() is None:    # left through an exception: end the launcher and its ranks
            signal_group(signal.SIGTERM)
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                signal_group(signal.SIGKILL)
                proc.wait()


def cpu_baseline(seconds_budget: float = 12.0):
    """Times the reference's CPU step (serial, as the reference builds it) on a bounded sample."""
    import numpy as np
    from oracle import oracle as O   # checker / baseline only — never the measured product
    import nbody_amd
    n = 16384
    x0 = nbody_amd.engine.seeded_bodies(n, 0, 12345)
    out = {}
    if O.have_ref():
        fn, kind = O.ref_step, "reference"
    else:
        fn, kind = (lambda x, a, v, steps=1: O.step_inplace(x, a, v, steps=steps)), "port"
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    fn(x, a, v, steps=1)            # warm caches
    t0 = time.perf_counter()
    done = 0
    while done < 64:
        fn(x, a, v, steps=1)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    out.update({"value": n * (n - 1) * done / dt, "unit": "pairs/s", "cores": 1, "kind": kind,
                "sample": f"{done} serial in-place steps of validation.cpp:28-52 at N={n} (reference init, DT=0.1, EPS2=0.002), {dt:.1f} s"})
    # context only: our Jacobi restatement, SIMD over targets, OpenMP over all cores (varies a lot
    # between boxes of the pool — the serial reference figure above is the stated baseline)
    thr = O.max_threads()
    O.set_threads(thr)
    O.accel_range(x0, 0, n)          # spin up the thread pool
    t0 = time.perf_counter()
    reps = 0
    while reps < 200:
        O.accel_range(x0, 0, n)
        reps += 1
        if time.perf_counter() - t0 > seconds_budget / 3:
            break
    dt2 = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out.update({"port_all_cores_value": n * (n - 1) * reps / dt2, "port_all_cores": thr,
                "host_cpus": os.cpu_count(), "cpu_model": model})
    return out


def cpu_baseline_f64(seconds_budget: float = 10.0):
    """The checker's all-double Jacobi step (the fp64 variant has no reference counterpart: kind "port")."""
    import numpy as np
    from oracle import oracle as O
    import nbody_amd
    n = 8192
    x0 = nbody_amd.engine.seeded_bodies(n, 1, 12345).astype(np.float64)
    O.set_threads(1)
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    O.step_jacobi_f64(x, a, v, 0.01, 0.002, steps=1)
    t0 = time.perf_counter()
    done = 0
    while done < 64:
        O.step_jacobi_f64(x, a, v, 0.01, 0.002, steps=1)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return {"value": float(n) * n * done / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": f"{done} all-double Jacobi steps of the checker at N={n} (Plummer), one thread, {dt:.1f} s"}

SCLK_NOMINAL_MHZ = 2400.0        # the clock the 157.3 / 78.6 TFLOP/s vector peaks are quoted at (256 CUs x 4 SIMDs x 32 packed lanes x 2 FLOP x 2.4 GHz)
# compute-side efficiency of ONE rank's sharded step against 1/G of the single-GPU step at the same N (measured per kernel
# on one GPU with no-op collectives, general arithmetic): what the multi-GPU line's prediction is built from
RANK_COMPUTE_EFFICIENCY = {1: 1.0, 2: 0.993, 4: 0.981, 8: 0.976}


class SysfsClockSampler:
    """Context beside the in-kernel clock: the driver's own view (sysfs pp_dpm_sclk / hwmon freq1_input, power1_average) sampled from a
    thread during the timed repeats. NOT the measurement (MI355X_MICROARCH.md: pp_dpm_sclk reads up to ~10 % above the in-kernel clock);
    absent files or permissions simply leave the fields out."""

    def __init__(self, props):
        self.dir = None
        self.samples, self.power = [], []
        self._stop = None
        try:
            bdf = "%04x:%02x:%02x.0" % (int(getattr(props, "pci_domain_id", 0)), int(props.pci_bus_id), int(props.pci_device_id))
            d = os.path.join("/sys/bus/pci/devices", bdf)
            if os.path.isdir(d):
                self.dir = d
        except Exception:
            self.dir = None

    def _read_once(self):
        import glob
        mhz = None
        for f in glob.glob(os.path.join(self.dir, "hwmon", "hwmon*", "freq1_input")):
            try:
                mhz = float(open(f).read().strip()) / 1e6
            except (OSError, ValueError):
                pass
        if mhz is None:
            try:
                for ln in open(os.path.join(self.dir, "pp_dpm_sclk")):
                    if ln.strip().endswith("*"):
                        mhz = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (OSError, ValueError, IndexError):
                pass
        if mhz:
            self.samples.append(mhz)
        for name in ("power1_average", "power1_input"):
            for f in glob.glob(os.path.join(self.dir, "hwmon", "hwmon*", name)):
                try:
                    self.power.append(float(open(f).read().strip()) / 1e6)
                    return
                except (OSError, ValueError):
                    pass

    def start(self):
        if not self.dir:
            return
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.wait(0.2):
                self._read_once()
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._stop is None:
            return None
        self._stop.set()
        self._thread.join(timeout=2.0)
        out = {"samples": len(self.samples)}
        if self.samples:
            out["sclk_mhz_median"] = statistics.median(self.samples)
        if self.power:
            out["board_power_w_mean"] = sum(self.power) / len(self.power)
        out["note"] = "the driver's view (sysfs), sampled every 0.2 s during the timed repeats: context only, the in-kernel counters above are the measurement"
        return out if self.samples or self.power else None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of --steps steps (0 = auto: >= 3 and >= --min-seconds in all)")
    ap.add_argument("--min-seconds", type=float, default=10.0, help="auto repeats: timed GPU work to accumulate (sustained clocks, "
                    "and long enough for an external utilisation sampler to see)")
    ap.add_argument("--bodies", dest="n", type=int, default=0, help="number of bodies (default: 262144 on 1 GPU, 1048576 on several)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="without --bodies: strong = N 1048576 at every "
                    "G >= 2 (default, BASELINE configs[3]; G = 1 runs configs[2], N = 262144); weak = 262144*sqrt(G) bodies")
    ap.add_argument("--dt", type=float, default=0.01)
    ap.add_argument("--eps2", type=float, default=0.002)
    ap.add_argument("--init", type=int, default=1, help="0 reference cube, 1 Plummer")
    ap.add_argument("--kernel", default="fast", choices=["fast", "onesided", "symmetric"])
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--bpl", type=int, default=0)
    ap.add_argument("--jsplit", type=int, default=0)
    ap.add_argument("--sym-waves", type=int, default=0)
    ap.add_argument("--sym-bpl", type=int, default=0)
    ap.add_argument("--workspace-limit-mib", type=float, default=0.0, help="nbody_ctx_set_workspace_limit: cap on the context's partial-sum workspace "
                    "(0 = automatic); config.launch says what the step does under it")
    ap.add_argument("--inplace-sums", default="auto", choices=["auto", "on", "off"], help="nbody_ctx_set_inplace_sums: block pairs with the partial sums added in "
                    "place (no slab workspace). auto: only where the slab workspace does not fit its cap; on: wherever unit runs / block pairs would run; off: never")
    ap.add_argument("--masses", default="init", choices=["init", "random"], help="init: as the initial conditions give them (Plummer: every body 1/N, "
                    "which the symmetric kernels' equal-mass path picks up; cube: the reference's random masses); random: Plummer positions "
                    "with masses drawn uniformly over a decade, as the reference's fill_with_random4 does, total 1 - the general path")
    ap.add_argument("--equal-mass", default="headline", choices=["headline", "auto", "on", "off"], help="which pair arithmetic the TIMED steps run. "
                    "headline (default): the GENERAL arithmetic (nbody_ctx_set_equal_mass(0): what unequal masses, e.g. the reference's own "
                    "fill_with_random4 bodies, get) - value / ms_per_step / roofline.frac do not depend on the input's masses; when the bodies do "
                    "all carry one mass, three extra repeats with the library's default switched back on are reported beside it "
                    "(equal_mass_value, roofline.frac_equal_mass). auto: the library's default for the timed steps (launches of 32768 bodies or "
                    "more are scanned for one common mass); on: from 4096 bodies; off: general arithmetic, no extras")
    ap.add_argument("--no-equal-mass-extras", "--no-general-path", dest="no_extras", action="store_true", help="skip the three extra repeats on the "
                    "equal-mass path (profile runs: only the timed kernels in the trace)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock", action="store_true", help="time the force launches with events alone (no clock-stamp launches around them): "
                    "the roofline block then carries no sclk / cycle fields")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="f64 = the build's own double-precision variant "
                    "(BASELINE configs[4]; single GPU only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1: nccl (RCCL; default) or gloo "
                    "(rehearsal of the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--comm", default="torch", choices=["torch", "native"], help="who runs the step's two collectives: torch = "
                    "torch.distributed on the --backend group; native = the library's own RCCL communicator (nbody_comm_rccl_*), "
                    "its unique id broadcast through the --backend group, which then only carries barriers and reductions")
    ap.add_argument("--comm-priority", default="auto", choices=["auto", "high", "normal", "ab"], help="priority of the stream RCCL's kernels run on: normal "
                    "(the library's default), high = the device's greatest (RCCL's few workgroups are placed ahead of the queued force workgroups; "
                    "MEASURED PATHOLOGICAL when three or more processes share one GPU: profiles/r04a_rehearsal_priority_probe.txt), auto = normal, "
                    "except with --comm native on a GPU per rank, where a few untimed steps are run both ways before the timed repeats and the faster "
                    "setting is kept (both timings reported as config.rccl.comm_priority_ab); ab = that in-run comparison wherever the ranks run "
                    "(--comm native only)")
    ap.add_argument("--no-single-gpu-point", action="store_true", help="multi-GPU runs: skip rank 0's same-N single-GPU timing before the sharded phase")
    ap.add_argument("--fake-hosts", action="store_true", help="rehearsal only: give every rank its own NCCL_HOSTID so that RCCL "
                    "accepts several ranks on ONE GPU (it then talks over its socket transport on the loopback interface)")
    ap.add_argument("--force-sharded", action="store_true", help="take the multi-GPU code path (process group, census, sharded step, "
                    "checks) even with ONE rank — what a 1-GPU box can exercise of it over real RCCL")
    ap.add_argument("--no-multi-gpu-check", action="store_true", help="skip the in-run parity / cross-rank checks (world > 1)")
    return ap.parse_args()


class Run:
    """What the phases of one bench run share (one process = one rank)."""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.multi = self.world > 1 or args.force_sharded          # the sharded code path
        self.f64 = args.dtype == "f64"
        self.real_stdout = None
        self.steps_done = 0          # every step this process asks for (the fp64 line compares with an fp32 run of the same length)
        self.comm_priority = args.comm_priority if args.comm_priority in ("high", "normal") else "normal"
        self.priority_ab = None
        self.eq_mode = {"headline": 0, "auto": -1, "on": 1, "off": 0}[args.equal_mass]

    # -- output ------------------------------------------------------------------------------------------------------------------
    def redirect_stdout(self) -> None:
        """Multi-rank runs: RCCL prints a version banner to STDOUT when its first communicator comes up. The contract is ONE JSON line on
        stdout, so until that line is printed everything written to file descriptor 1 (by any library, from any language) goes to stderr."""
        if self.multi:
            sys.stdout.flush()
            self.real_stdout = os.dup(1)
            os.dup2(2, 1)

    def emit(self, obj) -> None:
        """the run's one line (or the evidence of a failed self-check) on the real stdout"""
        if self.real_stdout is not None:
            sys.stdout.flush()
            os.dup2(self.real_stdout, 1)
        print(json.dumps(obj), flush=True)

    # -- stepping ----------------------------------------------------------------------------------------------------------------
    def run(self, k: int) -> None:
        self.steps_done += k
        self._run_steps(k)

    def barrier(self) -> None:
        import torch
        import torch.distributed as dist
        self.sync()
        torch.cuda.synchronize(self.dev)
        if self.multi:
            dist.barrier()

    def max_over_ranks(self, v: float) -> float:
        if not self.multi:
            return v
        import torch
        import torch.distributed as dist
        t = torch.tensor([v], dtype=torch.float64, device=self.red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def select_device(R) -> None:
    import torch
    args = R.args
    args.gpus = R.world   # (under the launcher the environment decides; `--gpus G` without it has started its own ranks above)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    R.shared_gpu = args.fake_hosts or (args.backend != "nccl" and args.comm != "native")
    if args.comm_priority == "ab" and args.comm != "native":
        raise SystemExit("--comm-priority ab compares stream priorities of the library's own communicator: add --comm native")
    if R.world > ndev and not R.shared_gpu:
        raise SystemExit(f"{R.world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank (use --backend gloo, or --fake-hosts, to rehearse)")
    R.dev = torch.device("cuda", R.local_rank % ndev)
    torch.cuda.set_device(R.dev)
    if R.f64 and R.multi:
        raise SystemExit("--dtype f64 is a single-GPU variant")
    R.red_dev = R.dev if args.backend == "nccl" else "cpu"


def init_process_group(R) -> None:
    import datetime
    import torch.distributed as dist
    args = R.args
    if R.world == 1:   # a single rank not started by torch.distributed.run: rendezvous with itself
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    limit = datetime.timedelta(minutes=5)   # a collective that never completes becomes an error, not a hang
    phase("init_process_group")
    if args.backend == "nccl":
        opts = None
        try:    # RCCL's kernels on a high-priority stream: placed as soon as a slot frees, not behind the queued force workgroups
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True) if R.comm_priority == "high" else None
        except Exception:
            opts = None
        dist.init_process_group("nccl", device_id=R.dev, timeout=limit, **({"pg_options": opts} if opts is not None else {}))
    else:
        dist.init_process_group(args.backend, timeout=limit)


def make_workload(R) -> None:
    import numpy as np
    import nbody_amd
    args = R.args
    R.n, R.scaling = default_workload(R.world, args.n, args.scaling)
    R.x0 = nbody_amd.engine.seeded_bodies(R.n, args.init, 12345)
    if args.masses == "random" and args.init == 1:
        u = np.random.RandomState(12345).uniform(1.0, 10.0, R.n)
        R.x0[:, 3] = (u / u.sum()).astype(np.float32)
    R.kernel = {"fast": nbody_amd.KERNEL_FAST, "onesided": nbody_amd.KERNEL_ONESIDED, "symmetric": nbody_amd.KERNEL_SYMMETRIC}[args.kernel]
    R.kopts = dict(kernel=R.kernel, tile=args.tile, bodies_per_lane=args.bpl, jsplit=args.jsplit)


def census(R):
    """did the collective library see `world` ranks on `world` different devices?"""
    import torch.distributed as dist
    mine = device_census(R.dev)
    recs = [None] * R.world
    dist.all_gather_object(recs, mine)
    recs = sorted(recs, key=lambda r: r["rank"])
    distinct = len({device_key(r) for r in recs}) == R.world
    rccl = {"world": dist.get_world_size(), "backend": dist.get_backend(), "comm": R.args.comm, "devices": recs,
            "distinct_devices": distinct, "ranks_seen": sorted(r["rank"] for r in recs) == list(range(R.world)),
            "fake_hosts": bool(R.args.fake_hosts)}
    if not rccl["ranks_seen"] or rccl["world"] != R.world:
        raise SystemExit(f"census: expected ranks 0..{R.world - 1}, the group reports {rccl['world']} ranks: {recs}")
    if not distinct and not R.shared_gpu:
        raise SystemExit(f"census: {R.world} ranks but the devices are not distinct: {recs}")
    return rccl


def single_gpu_same_n(R):
    """the same-N single-GPU point of the strong-scaling series, measured in THIS run (rank 0 alone, the others wait)"""
    import torch
    import torch.distributed as dist
    import nbody_amd
    args, n = R.args, R.n
    phase("single_gpu_same_n")
    rec = [None]
    if R.rank == 0:
        s1 = nbody_amd.engine.Simulation(R.x0, dt=args.dt, eps2=args.eps2, device=R.dev.index, **R.kopts)
        if args.sym_waves or args.sym_bpl:
            s1.ctx.set_symmetric_shape(args.sym_waves, args.sym_bpl)
        s1.ctx.set_equal_mass(R.eq_mode)
        k1 = max(1, min(args.steps, int(math.ceil(2.0 / (float(n) * n / 6.5e12)))))   # about 2 s of steps, at most --steps
        s1.run(1, sync=False)
        s1.ctx.sync()
        t0 = time.perf_counter()
        s1.run(k1, sync=False)
        s1.ctx.sync()
        t1 = time.perf_counter() - t0
        v1 = s1.ctx.equal_mass_verdict()
        rec[0] = {"ms_per_step": t1 / k1 * 1e3, "value": float(n) * n * k1 / t1, "unit": "pairs/s", "steps": k1, "n_bodies": n,
                  "measured_in_this_run": True, "equal_mass_path": bool(v1["scanned"] and v1["uniform"]),
                  "launch": {k: v for k, v in s1.ctx.step_info(n).items() if k in ("symmetric", "runs", "balanced", "fused", "block_bodies", "slabs")},
                  "where": "rank 0's GPU, plain nbody_step of all N bodies, before the sharded phase" +
                           (" (the other ranks of this rehearsal share that GPU but are idle in a barrier)" if R.shared_gpu else "")}
        s1.ctx.close()
        del s1
        torch.cuda.empty_cache()
    dist.broadcast_object_list(rec, src=0)
    return rec[0]


def build_simulation(R) -> None:
    """R.sim / R.ctx / R.sync / R._run_steps / R.info for the three kinds of run: fp64, single GPU, sharded"""
    import numpy as np
    import torch
    import nbody_amd
    args, n, dev = R.args, R.n, R.dev
    if R.f64:
        class _F64Sim:   # same alloc/init/H2D sequence as engine.Simulation, double state
            def __init__(self):
                self.ctx = nbody_amd.engine.Context(device=dev.index, kernel=R.kernel, jsplit=args.jsplit)
                if args.sym_waves or args.sym_bpl:
                    self.ctx.set_symmetric_shape(args.sym_waves, args.sym_bpl)
                self.x = torch.from_numpy(R.x0.astype(np.float64)).to(dev)
                self.v = torch.zeros_like(self.x)
                self.a = torch.zeros_like(self.x)
                self.shard, self.n_pad = n, n
                torch.cuda.synchronize(dev)

            def run(self, k, sync=False):
                self.ctx.step_f64(self.x, self.a, self.v, args.dt, args.eps2, k)

        R.sim = _F64Sim()
        R.ctx = R.sim.ctx
        R._run_steps = lambda k: R.sim.run(k)
        R.sync = R.ctx.sync
        R.info = R.ctx.step_info_f64(n)   # the library's own account of what nbody_step_f64 launches
    elif not R.multi:
        R.sim = nbody_amd.engine.Simulation(R.x0, dt=args.dt, eps2=args.eps2, device=dev.index, **R.kopts)
        R.ctx = R.sim.ctx
        if args.workspace_limit_mib > 0:
            R.ctx.set_workspace_limit(int(args.workspace_limit_mib * 2 ** 20))
        if args.inplace_sums != "auto":
            R.ctx.set_inplace_sums(1 if args.inplace_sums == "on" else 0)
        if args.sym_waves or args.sym_bpl:
            R.ctx.set_symmetric_shape(args.sym_waves, args.sym_bpl)
            R.ctx.reserve(n)
        R._run_steps = lambda k: R.sim.run(k, sync=False)
        R.sync = R.ctx.sync
        R.info = R.ctx.step_info(n)
    else:
        R.sim = sim = nbody_amd.sharded.ShardedSimulation(R.x0, dt=args.dt, eps2=args.eps2, device=dev, sym_waves=args.sym_waves,
                                                          sym_bpl=args.sym_bpl, comm=args.comm, comm_priority=R.comm_priority, **R.kopts)
        R.ctx = sim.ctx
        R._run_steps = sim.step
        R.sync = sim.sync
        plan = sim.plan
        symmetric_schedule = plan.schedule == nbody_amd.sharded.SCHEDULE_SYMMETRIC
        info = R.ctx.square_info(sim.shard, 2 if symmetric_schedule and R.world > 1 else 1)   # the own-block pass as it is launched (two parts)
        if symmetric_schedule:   # + the cross launches: each pair once, both sides
            cross = sum(float(plan.launch[l].i1 - plan.launch[l].i0) * plan.launch[l].count for l in range(plan.n_launches))
        else:                    # + own targets x every other block, one-sided
            cross = float(sim.shard) * (sim.n_pad - sim.shard)
        info["evaluated_pairs"] += cross
        info["schedule"] = {0: "canonical", 1: "onesided", 2: "symmetric"}[plan.schedule]
        info["cross_launches"] = [[plan.launch[l].i0, plan.launch[l].i1, plan.launch[l].j0, plan.launch[l].count] for l in range(plan.n_launches)]
        R.info = info
    R.ctx.set_equal_mass(R.eq_mode)


def multi_gpu_check_one_step(R, rccl):
    """multi-GPU self-check, part 1: one sharded step against the single-GPU kernel — twice: on Plummer positions with masses drawn over a
    decade (the cross launches and the exchange then carry MASS-WEIGHTED J-side sums, what the reference's own initial conditions give),
    and on the bench's own bodies (which the run continues from). Returns the check record; a failed check ends the run (exit 3)."""
    import numpy as np
    import torch
    import nbody_amd
    args, n, dev, sim = R.args, R.n, R.dev, R.sim
    phase("multi_gpu_check")
    ref_ctx = nbody_amd.engine.Context(device=dev.index, dt=args.dt, eps2=args.eps2, kernel=nbody_amd.KERNEL_ONESIDED)

    def one_step_against_single_gpu(bodies):
        sim.reset(bodies)
        R.run(1)                 # accelerations at the initial positions, through the whole sharded machinery
        R.barrier()
        xfull = torch.zeros((sim.n_pad, 4), dtype=torch.float32, device=dev)
        xfull[:n] = torch.from_numpy(bodies).to(dev)
        if sim.n_pad > n:        # the shard's padding bodies: massless, on top of body 0
            xfull[n:] = xfull[0]
            xfull[n:, 3] = 0.0
        samples = min(4096, sim.shard)
        pieces = 4 if samples >= 1024 else 1
        per = samples // pieces
        worst, amax = 0.0, 0.0
        for k in range(pieces):  # sample ranges spread over the own block
            off = (sim.shard - per) * k // max(pieces - 1, 1)
            i0 = sim.i0 + off
            ref = torch.empty((per, 4), dtype=torch.float32, device=dev)
            ref_ctx.accel_range(xfull, ref, i0, i0 + per, 0, sim.n_pad)
            ref_ctx.sync()
            got = sim.a[off:off + per]
            worst = max(worst, float((got - ref)[:, :3].abs().max().item()))
            amax = max(amax, float(ref[:, :3].abs().max().item()))
        del xfull
        rel = worst / amax if amax > 0 else float("inf")
        rel_all = R.max_over_ranks(rel)
        finite = R.max_over_ranks(0.0 if bool(torch.isfinite(sim.a).all().item()) else 1.0) == 0.0
        return rel_all, finite, per * pieces

    xr = R.x0.copy()
    u = np.random.RandomState(777).uniform(1.0, 10.0, n)
    xr[:, 3] = (u / u.sum() * float(R.x0[:, 3].astype(np.float64).sum())).astype(np.float32)     # same total mass, one decade of spread
    rel_rand, finite_rand, _ = one_step_against_single_gpu(xr)
    rel_all, finite, sampled = one_step_against_single_gpu(R.x0)
    ref_ctx.close()
    check = {"sampled_bodies_per_rank": sampled, "reference": "single-GPU one-sided kernel (nbody_accel_range) over all sources, on every rank's own GPU",
             "max_rel_da": rel_all, "tolerance": 5e-5, "finite": finite,
             "random_masses": {"max_rel_da": rel_rand, "finite": finite_rand,
                               "bodies": "the same positions, masses uniform over a decade (seed 777): mass-weighted J-side sums through the cross launches and the exchange"}}
    if not (rel_all <= 5e-5 and finite and rel_rand <= 5e-5 and finite_rand):
        if R.rank == 0:
            R.emit({"error": "multi_gpu_check failed", "multi_gpu_check": check, "rccl": rccl})
        raise SystemExit(3)
    return check


def comm_priority_ab(R) -> None:
    """--comm-priority auto, library communicator, one GPU per rank: measure both settings on this machine, keep the faster"""
    args, sim = R.args, R.sim
    phase("comm priority A/B")
    k_ab = max(2, min(args.steps, 5))
    ab = {}
    for setting in ("normal", "high"):
        sim.set_comm_priority(setting)
        R.run(1)
        R.barrier()
        t0 = time.perf_counter()
        R.run(k_ab)
        R.barrier()
        ab[setting] = R.max_over_ranks(time.perf_counter() - t0) / k_ab * 1e3
    R.comm_priority = "high" if ab["high"] < 0.98 * ab["normal"] else "normal"      # the same on every rank: from reduced times
    sim.set_comm_priority(R.comm_priority)
    R.priority_ab = {"ms_per_step": ab, "steps_each": k_ab, "kept": R.comm_priority, "rule": "high only if more than 2 % faster"}
    R.barrier()


def cross_rank_check(R, check, rccl, steps_before_timing) -> None:
    """part 2: after the warm-up steps every rank's copy of every block must be bit-identical"""
    import torch.distributed as dist
    sim = R.sim
    sim.refresh_positions()
    sums = sim.block_checksums().cpu()
    alls = [None] * R.world
    dist.all_gather_object(alls, sums.tolist())
    equal = all(a == alls[0] for a in alls)
    check["x_bitwise_equal_across_ranks"] = equal
    check["steps_checked"] = 1 + steps_before_timing
    if not equal:
        if R.rank == 0:
            R.emit({"error": "positions differ between ranks", "multi_gpu_check": check, "rccl": rccl})
        raise SystemExit(3)
    R.barrier()


def merge_clock(parts):
    """launch-weighted mean of several nbody_ctx_clock_read records (one per repeat)"""
    parts = [p for p in parts if p and p["launches"] > 0 and p["sclk_mhz"] > 0]
    if not parts:
        return None
    L = sum(p["launches"] for p in parts)
    cyc = sum(p["cycles_per_launch"] * p["launches"] for p in parts) / L
    tk = sum(p["ticks_per_launch"] * p["launches"] for p in parts) / L
    return {"launches": L, "xcds": max(p["xcds"] for p in parts), "unpaired": sum(p["unpaired"] for p in parts),
            "cycles_per_launch": cyc, "ticks_per_launch": tk, "sclk_mhz": cyc / tk * 100.0,
            "cycles_per_launch_min": min(p["cycles_per_launch_min"] for p in parts), "cycles_per_launch_max": max(p["cycles_per_launch_max"] for p in parts),
            "sclk_mhz_min_xcd": min(p["sclk_mhz_min_xcd"] for p in parts), "sclk_mhz_max_xcd": max(p["sclk_mhz_max_xcd"] for p in parts),
            "sclk_mhz_min_repeat": min(p["sclk_mhz"] for p in parts), "sclk_mhz_max_repeat": max(p["sclk_mhz"] for p in parts)}


def timed_repeats(R):
    """R repeats of EXACTLY K steps, each bracketed by a barrier + device synchronise on both sides and reduced with MAX over ranks.
    The force kernel's own time comes from a pair of HIP events around every launch, its cycle count and the shader clock from a pair of
    clock-stamp launches outside those events. At the sizes the metric is quoted on that costs nothing measurable; below ~32k bodies (steps
    of tens of microseconds) the events themselves would slow the timed region by 10-30 %, so there the timed repeats run un-instrumented
    and the kernel time is taken from separate instrumented repeats."""
    import torch
    args, ctx = R.args, R.ctx
    phase("timed repeats")
    if R.multi:
        R.sim.comm_timing(True)
    clock_on = not args.no_clock
    inline = R.n >= 32768 or R.multi
    ctx.timing(inline, clock=clock_on and inline)
    sampler = SysfsClockSampler(torch.cuda.get_device_properties(R.dev))
    sampler.start()
    T = {"inline_events": inline, "repeats": [], "kernel_ms": [], "kernel_launches": 0, "clock_parts": []}
    target = args.repeats if args.repeats > 0 else 3
    while len(T["repeats"]) < target:
        R.barrier()
        t0 = time.perf_counter()
        R.run(args.steps)              # EXACTLY K steps per timed region
        R.barrier()
        elapsed = R.max_over_ranks(time.perf_counter() - t0)
        T["repeats"].append(elapsed)
        if inline:
            ms, launches = ctx.timing_read()
            T["kernel_ms"].append(ms)
            T["kernel_launches"] += launches
            if clock_on:
                T["clock_parts"].append(ctx.clock_read())
        if args.repeats <= 0 and len(T["repeats"]) == 1:   # same on every rank: from the reduced time
            target = min(max(3, int(math.ceil(args.min_seconds / max(elapsed, 1e-6)))), 64)
    T["sysfs"] = sampler.stop()
    T["kernel_repeats"] = len(T["repeats"])
    if not inline:
        ctx.timing(True, clock=clock_on)
        T["kernel_repeats"] = 3
        for _ in range(3):
            R.run(args.steps)
            R.barrier()
            ms, launches = ctx.timing_read()
            T["kernel_ms"].append(ms)
            T["kernel_launches"] += launches
            if clock_on:
                T["clock_parts"].append(ctx.clock_read())
    ctx.timing(False)
    T["clock"] = merge_clock(T["clock_parts"])
    return T


def equal_mass_report(R, T):
    """Equal masses (a Plummer sphere: every body 1/N) let the symmetric kernels factor the common mass out of the pair sums; the
    decision is taken on the device per launch. The TIMED steps ran the general arithmetic (--equal-mass headline / off) unless
    the caller asked for the library's default; say which, and — headline mode — time the library's default on the same bodies right
    here, as an extra: it is what THIS input gets from the library, but not a figure an input with unequal masses can reach."""
    args, ctx, n = R.args, R.ctx, R.n
    inline = T["inline_events"]
    v = ctx.equal_mass_verdict()
    took = bool(R.eq_mode != 0 and v["scanned"] and v["uniform"])
    em = {"path_taken": took, "timed_steps_ran": "equal-mass path" if took else "general pair arithmetic",
          "decided": ("--equal-mass %s: the timed steps ran with nbody_ctx_set_equal_mass(0), the general arithmetic whatever the masses" % args.equal_mass) if R.eq_mode == 0 else
                     "on the device, per launch (nbk::mass_scan); nbody_ctx_set_equal_mass(0) disables it"}
    em["path_taken_on_every_rank"] = R.max_over_ranks(0.0 if took else 1.0) == 0.0
    if not (args.equal_mass == "headline" and not args.no_extras):
        return em
    phase("equal_mass_extras")
    ctx.set_equal_mass(-1)   # the library's default
    R.run(2)
    R.barrier()
    v = ctx.equal_mass_verdict()
    # (a collective decision: every rank times the extras or none does)
    on_all = R.max_over_ranks(0.0 if (v["scanned"] and v["uniform"]) else 1.0) == 0.0
    em["masses"] = ("all equal: %.9g" % v["mass"]) if v["scanned"] and v["uniform"] else "not all equal (or below the size the library scans from): general pair arithmetic is all there is"
    if on_all:
        eqt, eq_kernel_ms, eq_clock = [], [], []
        if inline:
            ctx.timing(True, clock=not args.no_clock)
            R.barrier()
            ctx.timing_read()
            ctx.clock_read()
        for _ in range(3):
            R.barrier()
            t0 = time.perf_counter()
            R.run(args.steps)
            R.barrier()
            eqt.append(R.max_over_ranks(time.perf_counter() - t0))
            eq_kernel_ms.append(ctx.timing_read()[0] if inline else 0.0)
            if inline and not args.no_clock:
                eq_clock.append(ctx.clock_read())
        ctx.timing(False)
        g = statistics.median(eqt)
        # force-kernel seconds per step by HIP events (this rank); small systems run un-instrumented: the whole step stands in
        gk = statistics.median(eq_kernel_ms) * 1e-3 / args.steps if inline else g / args.steps
        pk = FP64_VECTOR_PEAK_TFLOPS if R.f64 else FP32_VECTOR_PEAK_TFLOPS
        rp = float(R.sim.shard) * R.sim.n_pad if R.multi else float(n) * n
        ck = merge_clock(eq_clock)
        em["equal_mass_path"] = {"ms_per_step": g / args.steps * 1e3, "value": float(n) * n * args.steps / g, "unit": "pairs/s",
                                 "kernel_ms_per_step": gk * 1e3,
                                 "frac_of_peak_at_20_flop": FLOP_PER_PAIR * rp / gk / 1e12 / pk if gk > 0 else None,
                                 "frac_source": "force kernel's HIP-event time, as roofline.frac" if inline else "whole-step wall time (no per-launch events at this size)",
                                 **({"kernel_cycles_per_launch": ck["cycles_per_launch"], "sclk_mhz_under_load": ck["sclk_mhz"]} if ck else {}),
                                 "repeats": 3, "note": "same bodies, same run, the library's default (equal-mass path where the device-side scan finds one common "
                                                       "mass): 14 instead of 16 packed ops per two pair evaluations; an input-dependent figure, not the headline"}
    ctx.set_equal_mass(R.eq_mode)
    return em


def offline_traffic(R, symmetric):
    """PMC passes are taken offline (tools/gpu_round.sh pmc); valid only for the launch shape they were taken at"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if R.multi or R.f64 or not os.path.exists(tpath):
        return None, None, None
    try:
        top = json.load(open(tpath))
        for t in [top] + list(top.get("other_shapes", [])):
            if (t.get("n_bodies") == R.n and t.get("symmetric") == symmetric and t.get("slabs") == R.info.get("slabs")
                    and bool(t.get("ticket")) == bool(R.info.get("ticket"))):
                return (t.get("force_kernel_hbm_bytes_per_launch"), t.get("note"),
                        "OFFLINE PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/gpu_round.sh pmc), kept in profiles/traffic.json for this launch shape; NOT measured in this run")
    except Exception:
        pass
    return None, None, None


def clock_fields(R, T, achieved, peak, kernel_s_step, launches_per_step):
    """the clock the figure was measured at: shader cycles per force launch (the code) and the shader clock under load (the box)"""
    ck = T.get("clock")
    if not ck:
        return {}
    by_clock_ms = ck["ticks_per_launch"] * 1e-5 * launches_per_step
    out = {
        "sclk_mhz_under_load": ck["sclk_mhz"],
        "kernel_cycles_per_launch": ck["cycles_per_launch"],
        "kernel_cycles_per_step": ck["cycles_per_launch"] * launches_per_step,
        "frac_at_measured_clock": achieved / (peak * ck["sclk_mhz"] / SCLK_NOMINAL_MHZ) if ck["sclk_mhz"] > 0 else None,
        "clock": {
            "source": "nbk::clock_begin / clock_end launches in front of and behind every timed force launch, outside its event pair, on its stream "
                      "(nbody_ctx_timing(ctx, 2)): s_memtime (shader cycles, a per-CU counter: the two readings are paired CU by CU) and s_memrealtime (100 MHz), "
                      "one pair per XCD, mean over the XCDs; no stamp executes inside the measured kernel",
            "nominal_mhz": SCLK_NOMINAL_MHZ, "peak_at_measured_clock": peak * ck["sclk_mhz"] / SCLK_NOMINAL_MHZ,
            "stamped_launches": ck["launches"], "xcds": ck["xcds"], "launch_xcd_records_missing": ck["unpaired"],
            "sclk_mhz_slowest_xcd": ck["sclk_mhz_min_xcd"], "sclk_mhz_fastest_xcd": ck["sclk_mhz_max_xcd"],
            "sclk_mhz_min_repeat": ck["sclk_mhz_min_repeat"], "sclk_mhz_max_repeat": ck["sclk_mhz_max_repeat"],
            "cycles_per_launch_min": ck["cycles_per_launch_min"], "cycles_per_launch_max": ck["cycles_per_launch_max"],
            "kernel_ms_per_step_by_device_clock": by_clock_ms,
            "cycles_over_sclk_vs_event_ms": (by_clock_ms / (kernel_s_step * 1e3)) if kernel_s_step > 0 else None,
            "reading": "cycles x 1/sclk reproduces kernel_ms_per_step (the stamps sit outside the events: + two launch boundaries); same cycles and a lower sclk "
                       "= a slower box, more cycles at the same sclk = slower code",
            **({"sysfs": T["sysfs"]} if T.get("sysfs") else {}),
        },
    }
    return out


def roofline_block(R, T, em):
    """roofline of the dominant kernel (force accumulation), from its own event time on this rank"""
    args, info, n = R.args, R.info, R.n
    pairs_step = float(n) * n
    rank_pairs = float(R.sim.shard) * R.sim.n_pad if R.multi else pairs_step            # interactions this rank applies per step
    kr, kl = T["kernel_repeats"], T["kernel_launches"]
    launches_per_step = max(kl // (kr * args.steps), 1)
    kernel_s_step = sum(T["kernel_ms"]) * 1e-3 / (kr * args.steps)               # force-kernel time per step
    achieved = FLOP_PER_PAIR * rank_pairs / kernel_s_step / 1e12 if kernel_s_step > 0 else 0.0
    evaluated = float(info["evaluated_pairs"])
    achieved_eval = FLOP_PER_PAIR * evaluated / kernel_s_step / 1e12 if kernel_s_step > 0 else 0.0
    peak = FP64_VECTOR_PEAK_TFLOPS if R.f64 else FP32_VECTOR_PEAK_TFLOPS
    symmetric = bool(info.get("symmetric"))
    traffic, traffic_note, traffic_source = offline_traffic(R, symmetric)
    eqp = em.get("equal_mass_path")
    inline = T["inline_events"]
    return {
        "bound": "valu",
        "achieved": achieved,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        **clock_fields(R, T, achieved, peak, kernel_s_step, launches_per_step),
        # which pair arithmetic `frac` was measured on (default: the general one — the figure the >= 70 % target is read on), and the
        # equal-mass path of the same run beside it when the bodies all carry one mass
        "frac_path": ("equal-mass path (the bodies all carry one mass): 20 FLOP x N^2 interactions applied; the kernel executes 14 instead of "
                      "16 packed ops per two pair evaluations" if em["path_taken"] else "general pair arithmetic"),
        **({"frac_equal_mass": eqp["frac_of_peak_at_20_flop"], "equal_mass_kernel_ms_per_step": eqp["kernel_ms_per_step"],
            "frac_equal_mass_source": eqp["frac_source"]} if eqp else {}),
        "traffic": traffic,
        **({"traffic_source": traffic_source} if traffic_source else {}),
        **({"traffic_note": traffic_note} if traffic_note else {}),
        "kernel": ("nbk::force_sym<SymF64> (fp64, each unordered pair once)" if symmetric else "nbk::force_f64 (one-sided)") if R.f64 else
                  ("nbk::step_fused (fp32 packed, one-sided, force + integrate in one launch)" if info.get("fused") else
                   "nbk::force_sym_bal (fp32 packed, each unordered pair once, balanced runs)" if info.get("balanced") else
                   "nbk::force_sym_run (fp32 packed, each unordered pair once, unit runs)" if info.get("runs") else
                   "nbk::force_sym_ticket (fp32 packed, each unordered pair once, block sums added in place: no slab workspace)" if info.get("ticket") else
                   "nbk::force_sym / force_sym_square (fp32 packed, each unordered pair once)" if symmetric else "nbk::force_lds (fp32 packed, one-sided)"),
        "kernel_ms_per_step": kernel_s_step * 1e3,
        "kernel_time_source": ("HIP events around every force launch of the timed repeats" if inline else
                               "HIP events around every launch of 3 separate instrumented repeats (at this size the events would slow the timed repeats, which ran without them)"),
        "kernel_launches_per_step": launches_per_step,
        "kernel_launches": kl,
        "flop_per_pair": FLOP_PER_PAIR,
        "interactions_per_step": rank_pairs,
        "evaluated_pairs_per_step": evaluated,
        "achieved_evaluated": achieved_eval,
        "frac_evaluated": achieved_eval / peak,
        "note": ("fp64 vector-ALU bound; peak = 78.6 TFLOP/s fp64 vector" if R.f64 else
                 "fp32 vector-ALU bound (no MFMA, HBM traffic is O(N) per step); peak = 157.3 TFLOP/s fp32 vector = fp32 MFMA peak. "
                 "achieved/frac: 20 FLOP x interactions applied (N^2, the metric's convention: SURVEY.md 8d's per-unit figure x the units one launch processes); "
                 "achieved_evaluated/frac_evaluated: 20 FLOP x pair evaluations actually EXECUTED (the symmetric kernel evaluates each unordered pair once and "
                 "applies it to both bodies) — the figure to read as ALU work done per second" +
                 ("; the bodies all carry the same mass and --equal-mass asked for the library's default, so the kernel took its equal-mass path (14 instead of 16 "
                  "packed ops per two pair evaluations: the common mass is factored out of the sums and applied once per stored partial sum)"
                  if em["path_taken"] else "; the timed steps ran the general pair arithmetic (the headline does not depend on the input's masses)")),
    }


def clock_per_rank(R, roof):
    """multi-GPU lines: every rank's clock beside rank 0's roofline (eight parts under RCCL load each pick their own clock)"""
    import torch.distributed as dist
    mine = {"rank": R.rank, **{k: roof.get(k) for k in ("sclk_mhz_under_load", "kernel_cycles_per_step", "kernel_ms_per_step", "frac", "frac_at_measured_clock")}}
    recs = [None] * R.world
    dist.all_gather_object(recs, mine)
    return sorted(recs, key=lambda r: r["rank"])


def scaling_fields(R, value, ms_per_step, same_n):
    """the multi-GPU line read against its own single-GPU point and against DESIGN.md 5's prediction"""
    if not same_n or R.world < 2:
        return {}
    g = R.world
    eff = RANK_COMPUTE_EFFICIENCY.get(g, 0.975)
    low = same_n["ms_per_step"] / g / eff
    return {"scaling_efficiency": value / (g * same_n["value"]),
            "scaling_efficiency_note": "value / (n_gpus x single_gpu_same_n.value): both measured in this run, same N, same pair arithmetic",
            "predicted_ms_per_step": {"low": low, "high": low + 0.3, "measured": ms_per_step,
                                      "basis": f"single_gpu_same_n.ms_per_step / {g} / {eff} (one rank's compute-side efficiency at G = {g}, measured per kernel on one GPU: "
                                               "DESIGN.md 5) with both collectives hidden (low) or exposed by up to 0.3 ms (high); DESIGN.md 5 quotes 22.0-22.3 ms for "
                                               "N = 1048576 at G = 8 from a 171.46-ms single-GPU step"}}


def fp32_vs_fp64(R):
    """configs[4]'s tolerance check: the fp32 engine from the same start, same number of steps"""
    import numpy as np
    import nbody_amd
    args = R.args
    total_steps = R.steps_done
    s32 = nbody_amd.engine.Simulation(R.x0, dt=args.dt, eps2=args.eps2, device=R.dev.index)
    s32.run(total_steps)
    x32 = s32.state()[0].astype(np.float64)
    x64 = R.sim.x.cpu().numpy()
    scale = float(np.abs(x64[:, :3]).max()) if args.init == 0 else 1.0     # Plummer: scale radius a = 1
    return {"steps": total_steps, "max_abs_dx": float(np.abs(x32 - x64)[:, :3].max()),
            "max_rel_dx": float(np.abs(x32 - x64)[:, :3].max() / scale), "scale": scale}


def build_line(R, T, em, roof, same_n, comm, rccl, check, fp_diff):
    import nbody_amd
    args, n, world, scaling, info = R.args, R.n, R.world, R.scaling, R.info
    repeats = T["repeats"]
    elapsed = statistics.median(repeats)
    pairs_step = float(n) * n
    value = pairs_step * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    eqp = em.get("equal_mass_path")
    return {
        "metric": "body_pair_interactions_per_s",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "repeats": len(repeats),
        "timed_seconds": sum(repeats),
        "ms_per_step_min": min(repeats) / args.steps * 1e3,
        "ms_per_step_max": max(repeats) / args.steps * 1e3,
        "per_gpu_value": value / world,
        # which pair arithmetic the timed steps ran (false by default: the general one, whatever the masses), and the library's default
        # on THIS input beside it
        "equal_mass_path": em["path_taken"],
        **({"equal_mass_value": eqp["value"], "equal_mass_ms_per_step": eqp["ms_per_step"]} if eqp else {}),
        **({"single_gpu_same_n": same_n} if same_n else {}),
        **scaling_fields(R, value, ms_per_step, same_n),
        "config": {
            "workload": f"all-pairs gravity step, N={n} bodies, {'fp64' if R.f64 else 'fp32'}, {'Plummer' if args.init == 1 else 'reference-cube'} init seed 12345, "
                        f"dt={args.dt}, eps2={args.eps2}" + (f", {world} GPUs, {scaling} scaling" if world > 1 else ""),
            "n_bodies": n,
            "pairs_per_step": pairs_step,
            "scaling_series": ("strong: N = 1048576 at every G >= 2 (BASELINE configs[3]); the default 1-GPU run is configs[2] (N = 262144), "
                               "the same-N 1-GPU point is `--gpus 1 --bodies 1048576`" if scaling == "strong" and not args.n else
                               "weak: N(G) = 262144*sqrt(G) rounded to 8192*G, equal pairs per GPU" if scaling == "weak" else f"--bodies {n} at every G"),
            "partition": "single GPU" if not R.multi else (f"{world} contiguous blocks of {R.sim.shard} bodies, all-gather of positions per step over {args.comm} "
                                                           f"({'library RCCL communicator' if args.comm == 'native' else 'torch.distributed ' + args.backend})" +
                                                           (", every unordered pair once across the ranks, J-side sums exchanged (grouped send/recv)" if info.get("schedule") == "symmetric" else "")),
            "kernel": nbody_amd.load().nbody_version().decode(),
            "launch": info,
            "masses": ("Plummer: every body 1/N" if args.init == 1 and args.masses == "init" else "Plummer positions, masses uniform over a decade (total 1)"
                       if args.init == 1 else "the reference's fill_with_random4 range"),
            "equal_mass": em,
            "gflops_at_20_flop_per_pair": value * FLOP_PER_PAIR / 1e9,
            **({"fp32_vs_fp64": fp_diff} if fp_diff else {}),
            **({"comm_rank0": comm} if comm else {}),
            **({"rccl": rccl} if rccl else {}),
            **({"multi_gpu_check": check} if check else {}),
        },
        "roofline": roof,
    }


def teardown(R) -> None:
    """The line is out. Tear-down (shard, communicator, process group): if it has not finished within a minute the process says where
    it hangs (phase + all Python stacks on stderr) and ends itself with a NON-ZERO exit code instead of leaving the launcher waiting."""
    import faulthandler
    import threading
    import torch.distributed as dist
    sys.stdout.flush()

    def give_up():   # NOT a success: the line stands, but a tear-down that hangs is a defect and the exit code says so
        print(f"[bench.py rank {R.rank} pid {os.getpid()}] tear-down did not finish within 60 s; hung in: {PHASE[0]}", file=sys.stderr, flush=True)
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        sys.stderr.flush()
        os._exit(3)
    watchdog = threading.Timer(60.0, give_up)
    watchdog.daemon = True
    watchdog.start()
    phase("teardown: barrier")
    R.barrier()
    phase("teardown: sim.close")
    R.sim.close()
    phase("teardown: destroy_process_group")
    dist.destroy_process_group()
    watchdog.cancel()


def main():
    args = parse_args()
    arm_stall_dump()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus G` started directly: this process becomes the launcher of its own G ranks (before torch is imported
        # and before anything touches a GPU) and ends with their exit code
        ensure_built()
        raise SystemExit(spawn_ranks(args.gpus))

    R = Run(args)
    R.redirect_stdout()
    if args.fake_hosts:   # before RCCL is loaded
        os.environ["NCCL_HOSTID"] = f"nbody-bench-host-{R.rank}"
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    ensure_built()   # before the process group and before any GPU call; ranks serialise on a file lock
    select_device(R)

    # The CPU baseline goes FIRST (rank 0, single-GPU runs): the GPU phase then runs uninterrupted to the end of the process
    cpu = None
    if R.rank == 0 and not R.multi and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline_f64() if R.f64 else cpu_baseline()
        except Exception as e:  # the baseline must never take the GPU number down with it
            cpu = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    if R.multi:
        init_process_group(R)
    make_workload(R)
    rccl = census(R) if R.multi else None
    same_n = single_gpu_same_n(R) if R.multi and R.world > 1 and not args.no_single_gpu_point else None
    build_simulation(R)

    check = None
    steps_before_timing = args.warmup
    if R.multi and not args.no_multi_gpu_check:
        check = multi_gpu_check_one_step(R, rccl)
        steps_before_timing = max(args.warmup - 1, 0)
    phase("warmup")
    R.run(steps_before_timing)
    R.barrier()
    if R.multi and R.world > 1 and args.comm == "native" and (args.comm_priority == "ab" or (args.comm_priority == "auto" and not R.shared_gpu)):
        comm_priority_ab(R)
    if check is not None:
        cross_rank_check(R, check, rccl, steps_before_timing)

    T = timed_repeats(R)
    comm = R.sim.comm_report() if R.multi else None
    if R.multi:
        rccl["comm_priority"] = R.comm_priority
        if R.priority_ab:
            rccl["comm_priority_ab"] = R.priority_ab
    em = equal_mass_report(R, T)
    roof = roofline_block(R, T, em)
    if R.multi:
        roof["clock_per_rank"] = clock_per_rank(R, roof)
    fp_diff = fp32_vs_fp64(R) if R.f64 else None
    line = build_line(R, T, em, roof, same_n, comm, rccl, check, fp_diff)
    if R.rank == 0:
        if cpu is not None:
            if cpu.get("value"):      # the GPU / CPU ratios, for the serial reference build and for the all-core restatement
                cpu["gpu_over_cpu"] = line["value"] / cpu["value"]
                if cpu.get("port_all_cores_value"):
                    cpu["gpu_over_cpu_all_cores"] = line["value"] / cpu["port_all_cores_value"]
            line["cpu_baseline"] = cpu
        R.emit(line)
    if R.multi:
        teardown(R)


if __name__ == "__main__":
    main()

"""The C-ABI library: loads, exports everything include/nbody.h declares, and fails loudly
(never falls back to a CPU path) when no HIP device is present. No GPU compute here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "nbody.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nbody_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(nb):
    declared = _declared()
    assert len(declared) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", nb._lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (nbody_[a-z0-9_]+)", out))
    missing = [s for s in declared if s not in exported]
    assert not missing, f"declared in include/nbody.h but not exported: {missing}"
    # the Python binding covers the same set
    assert sorted(nb.exported_symbols()) == declared


def test_library_contains_gfx950_code(nb):
    blob = open(nb._lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"force_lds" in blob and b"force_strict" in blob and b"integrate" in blob


def test_no_oracle_in_product():
    """The product must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "n-bodysimulation_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and False, f"{f} mentions the oracle"
    out = subprocess.run(["ldd", os.path.join(pkg, "libnbody_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "libref_cpu" not in out


def test_fails_loudly_without_a_device(nb):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = nb.load()
    h = C.c_void_p()
    rc = lib.nbody_ctx_create(C.byref(h), 0)
    assert rc == nb._lib.ERR_HIP and not h.value
    assert b"hip" in lib.nbody_last_error().lower()
    buf = np.zeros((8, 4), np.float32)
    p = C.c_void_p(buf.ctypes.data)
    assert lib.nbody_simulate(p, p, p, 8) == nb._lib.ERR_HIP   # no silent CPU step
    assert np.all(buf == 0)
    with pytest.raises(nb.NBodyError):
        nb.engine.Context()


def test_argument_validation_messages(nb):
    lib = nb.load()
    assert lib.nbody_ctx_set_params(None, 0.1, 0.002) == nb._lib.ERR_INVALID
    assert b"null context" in lib.nbody_last_error()
    assert lib.nbody_step(None, None, None, None, 4, 1) == nb._lib.ERR_INVALID
    assert lib.nbody_device_count(None) == nb._lib.ERR_INVALID


def test_seeded_generators_are_deterministic(nb):
    a = nb.engine.seeded_bodies(1000, 0, 42)
    b = nb.engine.seeded_bodies(1000, 0, 42)
    c = nb.engine.seeded_bodies(1000, 0, 43)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    # the reference's ranges (constants.h:15-19)
    assert np.all(np.abs(a[:, :3]) <= 1e5) and a[:, 3].min() >= 1e5 and a[:, 3].max() <= 1e9
    p = nb.engine.seeded_bodies(4096, 1, 7)
    assert np.allclose(p[:, 3].sum(), 1.0, rtol=1e-5)
    r = np.linalg.norm(p[:, :3], axis=1)
    assert 1.15 < np.median(r) < 1.45        # Plummer 3-D half-mass radius = 1.305 a
    # pinned values (guards against accidental generator changes; goldens depend on it)
    assert np.allclose(nb.engine.seeded_bodies(2, 0, 12345)[0], [-7.3384062e+04, -5.9036672e+04, -7.6091484e+04, 1.7620019e+08], rtol=1e-7)


def test_zero_fill_and_types(nb):
    buf = np.ones((5, 4), np.float32)
    nb.load().nbody_fill_with_zeroes4(C.c_void_p(buf.ctypes.data), 5)
    assert np.all(buf == 0)


def _plan(nb, nt, ns, kernel=0, tile=0, bpl=0, jsplit=0, cus=256):
    o = [C.c_int() for _ in range(4)]
    rc = nb.load().nbody_plan(nt, ns, kernel, tile, bpl, jsplit, cus, *[C.byref(x) for x in o])
    assert rc == 0
    return tuple(x.value for x in o)  # bpl, tile, jsplit, blocks_x


def _slab_ranges(j0, j1, tile, nslab):
    """Python mirror of nbk::slab_range (nbody_kernels.hip.h)."""
    ntile = (j1 - j0 + tile - 1) // tile
    out = []
    for s in range(nslab):
        a = min(j0 + (s * ntile // nslab) * tile, j1)
        b = min(j0 + ((s + 1) * ntile // nslab) * tile, j1)
        out.append((a, b))
    return out


def test_launch_plan_is_always_buildable_and_covers_every_source(nb):
    """Host logic of the launcher, no GPU: every auto plan names an instantiated kernel, the grid
    covers all targets, and the slabs tile [j0,j1) exactly once."""
    built = {(1, 256), (1, 512), (1, 1024), (2, 256), (2, 512), (2, 1024), (4, 256), (4, 512), (4, 1024), (4, 2048)}
    rng = np.random.default_rng(0)
    sizes = [1, 2, 63, 64, 255, 256, 257, 1000, 1024, 4095, 8192, 16384, 32768, 65536, 100000, 262144, 1048576, 3000001]
    sizes += [int(x) for x in rng.integers(1, 2_000_000, 60)]
    for nt in sizes:
        for ns in (nt, max(1, nt // 7), nt * 3 + 5):
            for user in ({}, {"tile": 2048}, {"bpl": 1}, {"bpl": 2}, {"tile": 256}, {"jsplit": 5}, {"tile": 512, "bpl": 4}):
                bpl, tile, js, bx = _plan(nb, nt, ns, **user)
                assert (bpl, tile) in built, (nt, ns, user, bpl, tile)
                assert 1 <= js <= 64 and bx * 256 * bpl >= nt > (bx - 1) * 256 * bpl
                if "jsplit" in user:
                    assert js == user["jsplit"]
                r = _slab_ranges(0, ns, tile, js)
                assert r[0][0] == 0 and r[-1][1] == ns
                assert all(r[k][1] == r[k + 1][0] for k in range(js - 1)) and all(a <= b for a, b in r)
                lens = [-(-(b - a) // tile) for a, b in r]
                assert max(lens) - min(lens) <= 1                     # slabs balanced to within one tile
    # strict: one target per lane, one slab, whatever the user asked
    assert _plan(nb, 5000, 5000, kernel=1, tile=256, bpl=4, jsplit=8)[:3] == (1, 1024, 1)
    # the sizes the docs quote
    assert _plan(nb, 262144, 262144)[:2] == (4, 2048) and _plan(nb, 8192, 8192)[:2] == (1, 256)


def test_header_says_which_symbols_are_the_boundary(nb):
    """include/nbody.h lists, at its top, the drop-in boundary (what a maintainer of the reference binds) apart from the extensions:
    the two lists together are exactly the declared symbols, and the boundary stays small."""
    import re
    hdr = open(os.path.join(ROOT, "include", "nbody.h")).read()
    head = hdr[:hdr.index("#ifndef NBODY_H")]
    b0, e0 = head.index("THE DROP-IN BOUNDARY"), head.index("EXTENSIONS")
    boundary = set(re.findall(r"\bnbody_[a-z0-9_]+", head[b0:e0]))
    extensions = set(re.findall(r"\bnbody_[a-z0-9_]+", head[e0:]))
    declared = set(re.findall(r"^(?:int|float|void|const char\*) (nbody_[a-z0-9_]+)\(", hdr, re.M))
    assert not (boundary & extensions)
    assert boundary | extensions == declared == set(nb.exported_symbols()), (sorted(declared - boundary - extensions), sorted((boundary | extensions) - declared))
    assert len(boundary) == 20 and {"nbody_simulate", "nbody_malloc_device", "nbody_memcpy_d2h", "nbody_fill_with_random4", "nbody_verify_still_bodies"} <= boundary
    assert "(20)" in head[b0:e0] and "(%d)" % len(extensions) in head[e0:]


def test_library_exports_no_stray_c_symbols(nb):
    """Besides the `nbody_*` entry points of include/nbody.h the library may export C++-mangled names (kernel stubs, inline
    members) and the HIP runtime's registration symbols — but no bare C identifier: a helper that leaks out as `accel_impl` or
    `rccl_load` (an unnamed namespace nested inside an `extern "C"` block does exactly that) can collide with the host program's own."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", nb._lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    stray = [n for n in names if not (n.startswith("nbody_") or n.startswith("_Z") or n.startswith("__hip_") or n in ("_init", "_fini"))]
    assert not stray, stray


def test_bench_defaults_name_the_baseline_configs():
    """--gpus 1 runs configs[2] (N=262144); a dry look at what --gpus 8 would run: configs[3], N=1048576, strong."""
    import importlib.util
    root = ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.N_SINGLE == 262144 and b.N_MULTI == 1048576
    assert b.default_workload(1, 0, "strong") == (262144, "strong")     # a point of the series it is compared with, never "weak" by default
    assert b.default_workload(1, 0, "weak") == (262144, "weak")          # ... and the first point of the weak series
    assert b.default_workload(1, 1048576, "strong") == (1048576, "strong")   # the same-N 1-GPU point of the strong series
    for g in (2, 4, 8):
        assert b.default_workload(g, 0, "strong") == (1048576, "strong")
    assert b.default_workload(8, 0, "weak") == (720896, "weak")
    assert b.default_workload(4, 65536, "strong") == (65536, "strong")


def test_bench_starts_its_own_ranks_when_not_under_the_launcher(tmp_path):
    """`python bench.py --gpus G` (G > 1) without torch.distributed.run around it becomes the launcher of its own ranks — a child
    process, the driver's own command line, rendezvous on 127.0.0.1 — instead of refusing to start. Without a GPU the ranks then
    stop where every bench run stops here ("needs a GPU"), and the parent hands their non-zero exit code on."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    cmd = b.rank_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29400)
    assert cmd[0] == sys.executable and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29400"
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "launch through torch.distributed.run" not in src and "os.exec" not in src       # never a refusal, never an exec
    import torch
    if torch.cuda.is_available():
        return                                                                                # (the GPU box runs the real thing: test_zz_rccl_rehearsal)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode != 0 and "bench.py needs a GPU" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_symmetric_shape_choice_without_a_device():
    """nbody_plan_symmetric: the block shape the cost estimate picks at the sizes the sweeps were measured at
    (profiles/r02_shape_probe_{mid,large}.jsonl: the measured best at each), and how restrictions narrow it."""
    import nbody_amd as nb
    lib = nb.load()

    def plan(n, waves=0, bpl=0, cus=256):
        o = [C.c_int() for _ in range(4)]
        rc = lib.nbody_plan_symmetric(n, cus, waves, bpl, *[C.byref(x) for x in o])
        return rc, tuple(x.value for x in o)

    assert plan(262144) == (0, (4, 10, 103, 5356))            # BASELINE configs[2]
    assert plan(1048576) == (0, (4, 10, 410, 84255))          # configs[3] on one GPU
    assert plan(65536)[1][:2] == (1, 8)                       # configs[1]
    assert plan(32768)[1][:2] == (1, 4) and plan(98304)[1][:2] == (2, 10) and plan(16384)[1][:2] == (1, 2)
    assert plan(8192) == (0, (1, 2, 64, 2080))                # the reference's N_BODIES, when the symmetric kernel is forced
    assert plan(262144, 4, 8) == (0, (4, 8, 128, 8256)) and plan(262144, 0, 8)[1][1] == 8
    for n in (130, 1000, 5000, 100003, 3000000):
        rc, (w, b, blocks, wgs) = plan(n)
        assert rc == 0 and blocks == -(-n // (64 * w * b)) and blocks >= 2 and wgs == blocks * (blocks + 1) // 2
    assert plan(100)[0] != 0                                  # one block: nothing to pair up
    assert plan(262144, 3, 8)[0] != 0 and plan(-1)[0] != 0 and plan(1000, cus=0)[0] != 0
    assert plan(40_000_000)[0] != 0                           # beyond the 96 GiB slab cap: the one-sided kernel takes over


def test_fused_step_shape_without_a_device(nb):
    """nbody_plan_fused: the fused small-N step's launch shape. Every size gets a built instantiation (even wave counts 2 ... 16,
    the tile each is compiled with), the grid covers all targets, and up to 8192 bodies on 256 CUs it is one workgroup per CU."""
    lib = nb.load()
    tiles = {2: 2048, 4: 2048, 6: 2304, 8: 2048, 10: 2560, 12: 2304, 14: 2688, 16: 2048}

    def plan(n, cus=256):
        o = [C.c_int() for _ in range(4)]
        assert lib.nbody_plan_fused(n, cus, *[C.byref(x) for x in o]) == 0
        return tuple(x.value for x in o)   # T, waves, tile, workgroups

    assert plan(8192) == (2, 16, 2048, 256) and plan(4096) == (2, 8, 2048, 256) and plan(2048) == (2, 4, 2048, 256)
    assert plan(6144) == (2, 12, 2304, 256) and plan(1024)[:2] == (2, 2) and plan(1) == (2, 2, 2048, 1)
    for n in list(range(1, 300)) + list(range(300, 20000, 97)) + [8191, 8192, 8193, 10240, 16384, 65536]:
        t, wv, tile, grid = plan(n)
        assert t in (2, 4) and wv in tiles and tile == tiles[wv] and tile % (64 * wv) == 0
        assert grid * wv * t >= n > (grid - 1) * wv * t
        if n <= 8192:
            assert t == 2 and grid <= 256
    assert lib.nbody_plan_fused(0, 256, None, None, None, None) == nb._lib.ERR_INVALID


def test_local_group_and_pin_arguments_without_a_device(nb):
    """The host-only parts of the round-4 entry points: a local (rank threads, no RCCL) group is created for 1 ... 64 ranks, refused
    outside that, destroyed once no rank holds a communicator; abort on a null group is a no-op."""
    lib, L = nb.load(), nb._lib
    g = C.c_void_p()
    assert lib.nbody_comm_local_group_create(C.byref(g), 0, 0.0) == L.ERR_INVALID and not g.value
    assert lib.nbody_comm_local_group_create(C.byref(g), L.MAX_RANKS + 1, 0.0) == L.ERR_INVALID
    assert lib.nbody_comm_local_group_create(None, 2, 0.0) == L.ERR_INVALID
    assert lib.nbody_comm_local_group_create(C.byref(g), 3, 5.0) == L.OK and g.value
    assert lib.nbody_comm_local_abort(g) == L.OK and lib.nbody_comm_local_abort(None) == L.OK
    comm = L.Comm()
    assert lib.nbody_comm_local_create(C.byref(comm), g, 3, 0) == L.ERR_INVALID      # rank 3 of 3
    assert lib.nbody_comm_local_destroy(C.byref(comm)) == L.OK                         # never created: nothing to do
    assert lib.nbody_comm_local_group_destroy(g) == L.OK and lib.nbody_comm_local_group_destroy(None) == L.OK


def test_autotune_decision_rule(nb):
    """nbody_autotune_decide: a measurement may override the built-in decomposition only by a clear and repeatable win on a quiet
    machine — the choice decides the low-order bits of every later result (a busy GPU flipped one run in six under the first rule)."""
    lib = nb.load()
    arr = lambda *v: (C.c_double * len(v))(*v)
    decide = lambda bf, bl, ch, cb=None, cc=None, m=0.03: lib.nbody_autotune_decide(bf, bl, ch, cb, cc, len(cb) if cb is not None else 0, m)
    assert decide(20.0, 20.2, 19.0) == 1                                   # quiet, 5 % faster
    assert decide(20.0, 20.2, 19.5) == 0                                   # 2.5 %: inside the margin
    assert decide(20.0, 23.0, 15.0) == 0                                   # the built-in choice disagrees with itself by 15 %: not quiet
    assert decide(23.0, 20.0, 15.0) == 0
    assert decide(20.0, 20.0, 19.0, arr(20.1, 20.0, 20.3), arr(19.0, 19.1, 18.9)) == 1
    assert decide(20.0, 20.0, 19.0, arr(20.1, 20.0, 20.3), arr(19.0, 19.6, 18.9)) == 0    # one confirmation trial too slow
    assert decide(20.0, 20.0, 19.0, arr(20.1, 19.2, 20.3), arr(19.0, 19.1, 18.9)) == 0    # one built-in trial as fast as the challenger
    assert decide(20.0, 20.0, 19.0, arr(20.1, -1.0, 20.3), arr(19.0, 19.1, 18.9)) == 0    # a failed trial
    assert decide(0.0, 20.0, 19.0) == 0 and decide(20.0, 20.0, 0.0) == 0 and decide(20.0, 20.0, 19.0, m=0.0) == 0


def test_report_structs_have_the_layout_the_python_binding_assumes(tmp_path):
    """nbody_clock_report / nbody_comm_report_t are filled by the library and read through ctypes: sizes and field offsets of the two
    sides must agree (compiled from include/nbody.h with the host C compiler; no GPU)."""
    import ctypes
    import subprocess
    import nbody_amd as nb
    src = tmp_path / "layout.c"
    fields = {"nbody_clock_report": [f for f, _ in nb._lib.ClockReport._fields_], "nbody_comm_report_t": [f for f, _ in nb._lib.CommReport._fields_]}
    body = "".join(f'printf("{t} %zu", sizeof({t}));' + "".join(f'printf(" %zu", offsetof({t}, {f}));' for f in fs) + 'printf("\\n");' for t, fs in fields.items())
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nbody.h"\nint main(void){' + body + 'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for line, (t, cls) in zip(out, (("nbody_clock_report", nb._lib.ClockReport), ("nbody_comm_report_t", nb._lib.CommReport))):
        nums = [int(v) for v in line.split()[1:]]
        assert nums[0] == ctypes.sizeof(cls), (t, nums[0], ctypes.sizeof(cls))
        assert nums[1:] == [getattr(cls, f).offset for f, _ in cls._fields_], t


def test_multi_gpu_line_reads_against_its_own_single_gpu_point_and_prediction():
    """bench.py's multi-GPU extras (pure host logic): scaling_efficiency = value / (G x single_gpu_same_n.value) and the predicted step
    time from DESIGN.md 5's per-rank compute-side efficiency; the launch-weighted merge of per-repeat clock records."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    same_n = {"ms_per_step": 171.46, "value": 1048576.0 ** 2 / 171.46e-3}
    R = types.SimpleNamespace(world=8)
    f = b.scaling_fields(R, value=1048576.0 ** 2 / 22.1e-3, ms_per_step=22.1, same_n=same_n)
    assert abs(f["scaling_efficiency"] - 171.46 / (8 * 22.1)) < 1e-12
    p = f["predicted_ms_per_step"]
    assert abs(p["low"] - 171.46 / 8 / 0.976) < 1e-9 and abs(p["high"] - p["low"] - 0.3) < 1e-12 and p["measured"] == 22.1
    assert 21.9 < p["low"] < 22.0 and 22.2 < p["high"] < 22.3                      # 22.0-22.3 ms at G = 8 from a 171.46-ms step
    assert b.scaling_fields(types.SimpleNamespace(world=1), 1.0, 1.0, same_n) == {} and b.scaling_fields(R, 1.0, 1.0, None) == {}
    one = {"launches": 20, "xcds": 8, "unpaired": 0, "cycles_per_launch": 2.4e7, "cycles_per_launch_min": 2.39e7, "cycles_per_launch_max": 2.41e7,
           "ticks_per_launch": 1.0e6, "sclk_mhz": 2400.0, "sclk_mhz_min_xcd": 2380.0, "sclk_mhz_max_xcd": 2420.0}
    two = dict(one, launches=60, cycles_per_launch=2.4e7, ticks_per_launch=1.2e6, sclk_mhz=2000.0, sclk_mhz_min_xcd=1990.0, sclk_mhz_max_xcd=2010.0)
    m = b.merge_clock([one, two, {"launches": 0, "sclk_mhz": 0}, None])
    assert m["launches"] == 80 and abs(m["ticks_per_launch"] - (20 * 1.0e6 + 60 * 1.2e6) / 80) < 1e-6
    assert abs(m["sclk_mhz"] - 2.4e7 / m["ticks_per_launch"] * 100.0) < 1e-9 and m["sclk_mhz_min_xcd"] == 1990.0 and m["sclk_mhz_max_repeat"] == 2400.0
    assert b.merge_clock([]) is None


def test_ticket_task_list_gives_every_block_its_contributions_in_order():
    """nbody_plan_ticket_task (the closed forms nbk::force_sym_ticket uses): walking the task list in launch order, every block receives
    contributions 0, 1, ..., nb-1 exactly once and in that order — so a contribution's predecessor always belongs to an EARLIER task (a
    waiting workgroup never waits for one that starts after it) — every unordered block pair occurs once, the diagonal block is each
    block's last contribution, and a block meets at most two tasks per anti-diagonal (what keeps concurrent tasks off one ticket)."""
    import ctypes as C
    import nbody_amd as nb
    lib = nb.load()
    for nbk in (2, 3, 4, 7, 16, 103, 410):
        count = [0] * nbk
        pairs = set()
        ntasks = nbk * (nbk - 1) // 2 + nbk
        last_d, seen_in_d = 0, {}
        for t in range(ntasks):
            i, j, si, sj = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            nb._lib.check(lib.nbody_plan_ticket_task(nbk, t, C.byref(i), C.byref(j), C.byref(si), C.byref(sj)))
            I, J = i.value, j.value
            assert 0 <= I <= J < nbk and (I, J) not in pairs
            pairs.add((I, J))
            assert si.value == count[I], (nbk, t, I, J, si.value, count[I])
            count[I] += 1
            if I != J:
                assert sj.value == count[J], (nbk, t, I, J, sj.value, count[J])
                count[J] += 1
                d = J - I
                assert d >= last_d                                       # anti-diagonal by anti-diagonal
                if d != last_d:
                    last_d, seen_in_d = d, {}
                for b in (I, J):
                    seen_in_d[b] = seen_in_d.get(b, 0) + 1
                    assert seen_in_d[b] <= 2
            else:
                assert sj.value == -1 and si.value == nbk - 1           # the diagonal block closes the count
        assert count == [nbk] * nbk and len(pairs) == ntasks
    assert lib.nbody_plan_ticket_task(1, 0, None, None, None, None) != 0 and lib.nbody_plan_ticket_task(4, 10, None, None, None, None) != 0

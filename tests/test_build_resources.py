"""Compiler-dependent assumptions of the launch-shape logic, checked against the compiler's own report.

`sym_cost` / `run_resolve` (n-bodysimulation_amd/csrc/nbody_plan.hip) rank the symmetric kernel's block shapes with the number
of waves per SIMD each instantiation's register allocation allows — a COMPILER OUTPUT, not a property of the source. A ROCm
update that pushed `SymPacked<10>` past 256 VGPRs, or made any hot kernel spill to scratch, would halve the rate silently.
`make -C n-bodysimulation_amd/csrc resources` compiles the product's device code with the product's flags plus
`-Rpass-analysis=kernel-resource-usage`; this file parses that report. No GPU needed (hipcc cross-compiles gfx950).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "build", "kernel_resources.txt")


@pytest.fixture(scope="module")
def kernels():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "n-bodysimulation_amd", "csrc"), "resources"], check=True)
    cur, res = None, {}
    for ln in open(REPORT, errors="replace"):
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", ln)
        if m and cur:
            res[cur][m.group(1).strip()] = m.group(2)
    assert res, "no kernel-resource-usage remarks in " + REPORT
    names = subprocess.run(["c++filt"] + list(res), check=True, capture_output=True, text=True).stdout.splitlines()
    out = {}
    for mangled, name in zip(res, names):
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)             # drop the parameter list
        out[name] = {k: (int(v) if v.lstrip("-").isdigit() else v) for k, v in res[mangled].items()}
    return out


def _get(kernels, name):
    assert name in kernels, f"{name} is not among the compiled kernels: {sorted(kernels)}"
    return kernels[name]


def test_every_kernel_is_free_of_scratch_and_spills(kernels):
    assert len(kernels) >= 30
    for name, r in kernels.items():
        assert r["ScratchSize"] == 0, (name, r)
        assert r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0, (name, r)
        assert r["Dynamic Stack"] == "False", (name, r)


@pytest.mark.parametrize("bpl,waves_list", [(10, (4, 2, 1)), (8, (4, 2, 1)), (4, (2, 1)), (2, (1,))])
def test_symmetric_kernel_occupancy_is_what_the_cost_model_assumes(nb, kernels, bpl, waves_list):
    want = nb.load().nbody_plan_symmetric_occupancy(bpl)
    assert want == {10: 2, 8: 3, 4: 5, 2: 8}[bpl]
    for w in waves_list:
        r = _get(kernels, f"nbk::force_sym<nbk::SymPacked<{bpl}>, {w}, 1>")
        assert r["Occupancy"] == want, (bpl, w, r)
        assert r["AGPRs"] == 0 and r["VGPRs"] <= 512 // want, (bpl, w, r)
        assert r["LDS Size"] == 64 * w * bpl * 16        # one float4 of J-side sums per body of the block


def test_run_kernels_occupancy(nb, kernels):
    for bpl in (10, 8):
        r = _get(kernels, f"nbk::force_sym_run<nbk::SymPacked<{bpl}> >")
        assert r["Occupancy"] == nb.load().nbody_plan_symmetric_occupancy(bpl), r
        assert r["LDS Size"] == 0


def test_fp64_kernel_occupancy(kernels):
    r = _get(kernels, "nbk::force_sym<nbk::SymF64<6>, 4, 1>")       # the default fp64 shape: two waves per SIMD
    assert r["Occupancy"] == 2 and r["AGPRs"] == 0, r
    r = _get(kernels, "nbk::force_sym<nbk::SymF64<8>, 4, 1>")       # on request only: one wave per SIMD
    assert r["Occupancy"] == 1, r


def test_one_sided_default_shape(kernels):
    r = _get(kernels, "nbk::force_lds<nbk::MathPacked<4>, 2048, 8, 1, 0, 256>")
    assert r["LDS Size"] == 2 * 2048 * 16 and r["Occupancy"] >= 2, r


def test_square_only_kernels_keep_the_occupancy_of_the_general_ones(nb, kernels):
    for bpl in (10,):
        r = _get(kernels, f"nbk::force_sym_square<nbk::SymPacked<{bpl}>, 4>")
        assert r["Occupancy"] == nb.load().nbody_plan_symmetric_occupancy(bpl), r
        assert r["LDS Size"] == 64 * 4 * bpl * 16


def test_rectangular_only_kernel(kernels):
    """nbody_accel_cross's build of the (4,10) shape: at least the two waves per SIMD the cost model counts on, LDS for three
    workgroups per CU."""
    r = _get(kernels, "nbk::force_sym_rect<nbk::SymPacked<10>, 4>")
    assert r["Occupancy"] >= 2 and r["LDS Size"] == 64 * 4 * 10 * 16 and r["AGPRs"] == 0, r


@pytest.mark.parametrize("bpl", [2, 4, 8, 10])
def test_balanced_run_kernels_fit_two_waves_per_simd(kernels, bpl):
    """The balanced-run plan puts exactly two workers on every SIMD (kBalWavesPerSimd): every instantiation must allow that,
    and its LDS (the I-side sums of the four workers of a workgroup) must leave room for two workgroups per CU."""
    r = _get(kernels, f"nbk::force_sym_bal<nbk::SymPacked<{bpl}>, 4>")
    assert r["Occupancy"] >= 2, r
    assert r["LDS Size"] <= 4 * 64 * bpl * 16 + 64 and 2 * r["LDS Size"] <= 160 * 1024


def test_fused_step_kernels_fit_one_workgroup_per_cu(kernels):
    """The fused small-N step runs one workgroup of 2 ... 16 waves per CU: every instantiation must fit (VGPRs of 16 waves:
    at most 128 each; LDS: the double-buffered source tile), without scratch."""
    names = [k for k in kernels if k.startswith("nbk::step_fused<")]
    assert len(names) == 32, names                           # 2 targets-per-wave x 8 workgroup sizes x {two arrays, in place}
    for name in names:
        r = kernels[name]
        args = name[len("nbk::step_fused<"):-1].split(",")    # <targets per wave, waves, tile, unroll, min waves, in place>
        t, wv, tile = (int(v) for v in args[:3])
        assert args[5].strip() in ("true", "false")
        assert r["LDS Size"] == 2 * tile * 16 and r["LDS Size"] <= 160 * 1024, (name, r)
        assert r["Occupancy"] * 4 >= wv, (name, r)            # all waves of the workgroup resident on the CU's four SIMDs


def test_every_global_kernel_of_the_product_header_is_instantiated_by_the_library(kernels):
    """No orphans: every `__global__` template of the product's device header (csrc/nbody_kernels.hip.h) is instantiated by
    libnbody_hip.so's sources (it shows up in the compiler's resource report of nbody_step.hip / nbody_shard.hip). Measured
    alternatives that nothing ships live in tools/nbody_experiments.hip.h, not in the product header."""
    hdr = open(os.path.join(ROOT, "n-bodysimulation_amd", "csrc", "nbody_kernels.hip.h")).read()
    declared = set(re.findall(r"__global__\s+void[^;{]*?\b([a-z_0-9]+)\s*\(const", hdr))
    assert len(declared) >= 15, declared
    compiled = {re.sub(r"<.*$", "", k).replace("nbk::", "") for k in kernels}
    assert declared <= compiled, f"kernels of the product header that the library never instantiates: {sorted(declared - compiled)}"
    exp = open(os.path.join(ROOT, "tools", "nbody_experiments.hip.h")).read()
    probes = set(re.findall(r"__global__\s+void[^;{]*?\b([a-z_0-9]+)\s*\(const", exp))
    assert probes and not (probes & compiled), f"tools-only kernels found in the library: {sorted(probes & compiled)}"


@pytest.mark.parametrize("tool", ["symbench", "balbench", "kbench", "f64bench", "f64shapes", "bal_sim", "sync_probe", "dp_mb", "valu_mb", "mfma_mb", "rsq64_probe", "pkbank_mb", "clock_probe"])
def test_developer_probes_still_compile_against_the_product_headers(tool):
    """tools/*.hip (the measured alternatives and the instruction-cost probes) are kept because DESIGN.md quotes their results: they must
    keep compiling against the product's device header and C-ABI as those move (front end only, host and gfx950 passes; no GPU needed)."""
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "--offload-arch=gfx950", "-fsyntax-only", "-Wno-unused-command-line-argument",
                        "-I", os.path.join(ROOT, "n-bodysimulation_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tools", tool + ".hip")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_in_place_block_pair_kernels_keep_the_slab_kernel_s_occupancy(nb, kernels):
    """nbk::force_sym_ticket (sums added in place): same rotation pass, same registers-per-wave class (two waves per SIMD at ten bodies
    per lane) and the same LDS as the slab kernel of its shape."""
    for w in (4, 1):
        r = _get(kernels, f"nbk::force_sym_ticket<nbk::SymPacked<10>, {w}>")
        assert r["Occupancy"] == nb.load().nbody_plan_symmetric_occupancy(10) and r["AGPRs"] == 0, r
        assert r["LDS Size"] == 64 * w * 10 * 16

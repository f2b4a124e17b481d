"""The balanced-run decomposition (nbk::force_sym_bal + nbk::bal_reduce) emulated on the HOST.

tools/bal_sim.hip includes the product's device header and replays, with the lanes played by a loop and the pair force replaced
by a random antisymmetric weight, exactly the control flow and index arithmetic of the kernel and of the reducer: the plan
(nbk::bal_plan), the split of units between workers at any rotation step, the lane permutation of a chunk entered mid-unit, the
workgroup-level combination of I-side sums, the inbox record each partial sum lands in, and the reducer's enumeration of the records
that exist. It checks that every worker's step range adds up to the whole list, that no record is written twice, that the set of
records written equals the set the reducer reads, and that every body receives sum_j w(i, j) — for 576 layouts (sizes that are not
multiples of anything, 2/4/8/10 bodies per lane, 1 ... 100000 workers, workgroups of 1/4/8). No GPU: hipcc compiles, nothing is
launched."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_balanced_run_decomposition_covers_every_pair_once(tmp_path):
    exe = str(tmp_path / "bal_sim")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "n-bodysimulation_amd", "csrc"),
                        os.path.join(ROOT, "tools", "bal_sim.hip"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ALL OK"), r.stdout[-2000:]
    assert r.stdout.count(": ok,") >= 500

#!/usr/bin/env python3
"""One table of fraction-of-peak across N from the bench lines of `tools/gpu_round.sh <tag> sizes` (one box, general pair arithmetic,
queued steps): what each size launches, its step time, the fraction of the fp32 vector peak on the step's wall time and on the force
kernel's event time, and the shader clock the box held (bench.py's clock stamps).

    python tools/sizes_table.py gpurun_out/<tag>      # markdown on stdout
"""
import glob
import json
import os
import re
import sys

PEAK = 157.3e12
FLOP = 20.0


def decomposition(launch):
    if launch.get("fused"):
        return "fused step (one launch, one-sided)"
    if launch.get("balanced"):
        return f"balanced runs ({launch['block_bodies'] // 64}/lane) + bal_reduce"
    if launch.get("runs"):
        return f"unit runs ({launch['block_bodies'] // 64}/lane) + integrate"
    if launch.get("ticket"):
        return f"block pairs, sums in place ({launch['block_bodies']}-body blocks, {launch['slabs'] or 1} lane(s)) + integrate"
    if launch.get("symmetric"):
        return f"block pairs ({launch['block_bodies']}-body blocks, {launch['slabs']} slabs) + integrate"
    return "one-sided LDS tiles + integrate"


def main():
    src = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(src, "n*_bench.json")):
        m = re.match(r"n(\d+)(?:_inplace)?_bench\.json$", os.path.basename(f))
        if not m:
            continue
        lines = [ln for ln in open(f) if ln.startswith("{")]
        if not lines:
            continue
        d = json.loads(lines[-1])
        n = d["config"]["n_bodies"]
        r = d["roofline"]
        rows.append((n, d, r))
    rows.sort(key=lambda t: (t[0], bool(t[1]['config']['launch'].get('ticket'))))
    print("| N | decomposition | µs / step | interactions / s | frac of peak (step wall time) | frac (force kernel events) | sclk under load (MHz) | frac at that clock | kernel cycles / step |")
    print("|---|---|---|---|---|---|---|---|---|")
    for n, d, r in rows:
        wall = FLOP * float(n) * n / (d["ms_per_step"] * 1e-3) / PEAK
        sclk = r.get("sclk_mhz_under_load")
        print(f"| {n} | {decomposition(d['config']['launch'])} | {d['ms_per_step'] * 1e3:.1f} | {d['value']:.3g} | {wall:.3f} | {r['frac']:.3f} | "
              f"{sclk:.0f} | {r.get('frac_at_measured_clock', float('nan')):.3f} | {r.get('kernel_cycles_per_step', float('nan')):.4g} |" if sclk else
              f"| {n} | {decomposition(d['config']['launch'])} | {d['ms_per_step'] * 1e3:.1f} | {d['value']:.3g} | {wall:.3f} | {r['frac']:.3f} | - | - | - |")


if __name__ == "__main__":
    main()

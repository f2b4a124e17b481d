#!/usr/bin/env python3
"""Condense the counter_collection CSVs of tools/pmc.sh passes into one small CSV per pass
(kernel, counter, dispatches, mean value per dispatch, grid, VGPR column, LDS) under profiles/.

    python tools/pmc_summary.py gpurun_out/<tag> profiles/<prefix>      # writes <prefix>_pmc_<pass>.csv

The VGPR_Count column of rocprofv3 is the ARCHITECTED VGPR allocation of the dispatch (accumulation
registers not included): the ISA's .vgpr_count / next_free_vgpr is the figure that sets occupancy."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    for d in sorted(glob.glob(os.path.join(src, "*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(lambda: [0, 0.0, None, None, None])
        for f in files:
            for row in csv.DictReader(open(f)):
                key = (row["Kernel_Name"], row["Counter_Name"])
                a = acc[key]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                a[2], a[3], a[4] = row.get("Grid_Size"), row.get("VGPR_Count"), row.get("LDS_Block_Size")
        out = f"{prefix}_pmc_{os.path.basename(d)}.csv"
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean_Value_per_dispatch", "Grid_Size", "VGPR_Count", "LDS_Block_Size"])
            for (k, c), a in sorted(acc.items()):
                if "nbk::" not in k:
                    continue
                w.writerow([k, c, a[0], a[1] / a[0], a[2], a[3], a[4]])
        print("wrote", out)


if __name__ == "__main__":
    main()

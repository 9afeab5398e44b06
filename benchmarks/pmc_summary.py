#!/usr/bin/env python3
"""Summarise rocprofv3 counter-collection CSVs: per-launch mean of every counter for one kernel.

Usage (on the GPU box; one --pmc group per rocprofv3 pass, never together with trace domains):
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d gpurun_out/pmc1 --output-format csv -- \
        python3 bench.py --steps 50 --warmup 5 --no-cpu --no-kernel-events
    python3 benchmarks/pmc_summary.py gpurun_out/pmc1 gpurun_out/pmc2 ... --kernel k_tridiag_seg --grid 1048576
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", default="k_tridiag_seg")
    ap.add_argument("--grid", type=int, default=0, help="keep only launches with this Grid_Size (0 = all)")
    args = ap.parse_args()
    sums = defaultdict(float)
    cnts = defaultdict(int)
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    if args.kernel not in row.get("Kernel_Name", ""):
                        continue
                    if args.grid and int(row.get("Grid_Size", 0)) != args.grid:
                        continue
                    sums[row["Counter_Name"]] += float(row["Counter_Value"])
                    cnts[row["Counter_Name"]] += 1
    out = {k: {"launches": cnts[k], "mean_per_launch": sums[k] / cnts[k]} for k in sorted(sums)}
    print(json.dumps({"kernel": args.kernel, "grid": args.grid, "counters": out}, indent=1))


if __name__ == "__main__":
    main()

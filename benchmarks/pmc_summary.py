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
    ap.add_argument("--sweeps-per-launch", type=int, default=1, help="sweeps one launch carries (per-sweep figures = per-launch / this)")
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
    rec = {"kernel": args.kernel, "grid": args.grid, "sweeps_per_launch": args.sweeps_per_launch, "counters": out}
    spl = args.sweeps_per_launch
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        # KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)
        per_launch = (2 * out["FETCH_SIZE"]["mean_per_launch"] + out["WRITE_SIZE"]["mean_per_launch"]) * 1024
        rec["hbm_bytes_per_launch"] = per_launch
        rec["hbm_bytes_per_sweep"] = per_launch / spl
    if "SQ_INSTS_VALU" in out and "SQ_WAVES" in out:
        rec["valu_instructions_per_wave"] = out["SQ_INSTS_VALU"]["mean_per_launch"] / out["SQ_WAVES"]["mean_per_launch"]
        # one wave = one sweep of 64 lanes' share of a chain: with self-restarting workgroups a hardware wave lives through all
        # the sweeps of the launch, so the count per wave AND SWEEP is what compares across launch forms
        chains = args.grid // 1024 if args.grid else 0
        restarting = chains and out["SQ_WAVES"]["mean_per_launch"] <= 16 * chains
        rec["valu_instructions_per_wave_and_sweep"] = rec["valu_instructions_per_wave"] / (spl if restarting else 1)
    if "SQ_ACTIVE_INST_VALU" in out and "SQ_WAVE_CYCLES" in out:
        rec["valu_active_share_of_wave_cycles_x4waves"] = 4 * out["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / out["SQ_WAVE_CYCLES"]["mean_per_launch"]
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()

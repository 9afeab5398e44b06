#!/bin/bash
# quick configuration probes of the headline kernel (kernel us per sweep)
out=${1:-gpurun_out/r04i}
mkdir -p $out
run() { name=$1; shift; timeout -k 10 200 python3 bench.py "$@" --no-cpu --secondary-ms 0 --repeat-ms 100 > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; return 0; }
  python3 - $out/$name.json $name <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["config"]["repeats"]
print("%-22s ms/step %.5f  kernel us/sweep min %.2f med %.2f  form %s  lam %.4f" % (sys.argv[2], d["ms_per_step"], 1e3 * r["kernel_ms_min"], 1e3 * r["kernel_ms_median"], d["config"]["diagnostics"]["headline_run"].get("launch_form"), d["config"]["check"]["mean_lambda"]))
P
}
run base
run seg8 --seg 8
run seg16 --seg 16
run seg20 --seg 20
run spl16 --sweeps-per-launch 16
run blk8 --block-sweeps 8
run reenter1 --reenter 1
run reenter0 --reenter 0

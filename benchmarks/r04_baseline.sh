#!/bin/bash
# Round-4 opening measurement: the driver's command at 1024 and 128 chains (the strong-scaling ratio under --steps 20),
# and the what-if without draw generation (the draws' share of the sweep as a measurement).
#   gpurun --timeout 900 -- 'bash benchmarks/r04_baseline.sh gpurun_out/r04a'
out=${1:-gpurun_out/r04a}
mkdir -p $out
set -o pipefail
run() { name=$1; shift; timeout -k 10 200 python3 bench.py "$@" --no-cpu --secondary-ms 0 > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }
  python3 - $out/$name.json $name <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["config"]["repeats"]
print("%-22s ms/step %.5f  kernel us/sweep min %.2f med %.2f max %.2f  wall med %.5f  form %s" % (sys.argv[2], d["ms_per_step"], 1e3 * r["kernel_ms_min"], 1e3 * r["kernel_ms_median"], 1e3 * r["kernel_ms_max"], r["ms_per_step_median"], d["config"]["diagnostics"]["headline_run"].get("launch_form")))
P
}
run steps20_1024 --steps 20 --warmup 5 &&
run steps20_128 --chains 128 --steps 20 --warmup 5 &&
run steps200_1024 &&
run steps200_128 --chains 128 &&
run zeroz_1024 --zero-z &&
run zeroz_128 --chains 128 --zero-z &&
run steps20_1024_b --steps 20 --warmup 5 &&
run steps20_128_b --chains 128 --steps 20 --warmup 5

#!/bin/bash
# One GPU's view of strong scaling: the headline kernel at the per-GPU chain counts of a 1024-chain job on 1/2/4/8/16 GPUs.
#   gpurun -- 'bash benchmarks/strong_scaling_probe.sh gpurun_out/tag'
out=${1:-gpurun_out/strong}
mkdir -p $out
for c in 1024 512 256 128 64; do
  timeout -k 10 120 python3 bench.py --chains $c --steps 200 --warmup 20 --no-cpu --secondary-ms 0 --repeat-ms 100 > $out/chains_$c.json 2> $out/chains_$c.err || { echo "chains $c failed"; tail -5 $out/chains_$c.err; exit 1; }
  python3 - $out/chains_$c.json $c <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["config"]["repeats"]
print("chains %5s  kernel us/sweep  min %.2f  median %.2f  max %.2f   launch form %s" % (sys.argv[2], 1e3 * r["kernel_ms_min"], 1e3 * r["kernel_ms_median"], 1e3 * r["kernel_ms_max"], d["config"]["diagnostics"]["headline_run"].get("launch_form")))
P
done

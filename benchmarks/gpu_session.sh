#!/bin/bash
# One measurement session on the GPU box: parity tests, the headline line twice (default and the driver's command), the
# kernel trace of the profiled run and the counter passes.  Everything lands under gpurun_out/<tag>/; the summaries that
# are to be judged are copied into profiles/ afterwards (by hand, named per round).
#   gpurun --timeout 1200 -- 'bash benchmarks/gpu_session.sh r03a'
# Steps are joined with &&: after a step that fails or is killed nothing else touches the GPU.
tag=${1:-sess}
what=${2:-all}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
set -o pipefail
step() { echo "== $(date +%T) $*" | tee -a $out/progress.log; }

if [ "$what" = all ] || [ "$what" = tests ]; then
  step "pytest -m gpu"
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; exit 1; }
  tail -3 $out/pytest_gpu.log
fi
if [ "$what" = all ] || [ "$what" = bench ]; then
  step "bench (driver's command)" &&
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err &&
  step "bench (default)" &&
  timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err &&
  python3 - $out <<'EOF'
import json, sys
for f in ("bench_steps20.json", "bench.json"):
    d = json.loads(open(sys.argv[1] + "/" + f).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(f, "value %.3e  ms/step %.4f  frac %.3f  repeats" % (d["value"], d["ms_per_step"], r.get("frac", 0)), d["config"]["repeats"])
    print("   secondary:", {k: (v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("error")) for k, v in (d["config"].get("secondary") or {}).items()})
    print("   cpu:", (d.get("cpu_baseline") or {}).get("value"))
EOF
fi
if [ "$what" = all ] || [ "$what" = prof ]; then
  step "rocprofv3 kernel trace of the headline" &&
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/trace --output-format csv -- python3 $OLDPWD/bench.py --steps 192 --warmup 32 --no-cpu --secondary-ms 0 > $OLDPWD/$out/bench_profiled.json 2> $OLDPWD/$out/bench_profiled.err) &&
  find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \; &&
  head -8 $out/kernel_stats.csv &&
  step "pmc passes" &&
  timeout -k 10 600 bash benchmarks/pmc_headline.sh $out/pmc > $out/pmc.log 2>&1 && tail -30 $out/pmc.log
fi
step done

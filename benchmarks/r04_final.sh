#!/bin/bash
# Round-4 closing session on one box: the two-rank gloo rehearsal of the multi-GPU launch contract (ranks share the GPU), then the
# hierarchical smoother under the kernel trace, the lattice / RW2 band numbers and the truncated scan on the final library.
#   gpurun --timeout 1200 -- 'bash benchmarks/r04_final.sh gpurun_out/r04k'
out=${1:-gpurun_out/r04k}
mkdir -p $out
export TMPDIR=/tmp
set -o pipefail
step() { echo "== $(date +%T) $*" | tee -a $out/progress.log; }
root=$PWD
step "two gloo ranks on one GPU: bench.py --gpus 2 (driver's command)" &&
OMC_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > $out/bench_gpus2.json 2> $out/bench_gpus2.err; echo "exit code $?" | tee -a $out/progress.log
python3 - $out/bench_gpus2.json <<'P'
import json, sys
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print("n_gpus", d["n_gpus"], "scaling", d["scaling"], "value %.3e" % d["value"], "ms/step %.4f" % d["ms_per_step"], "gather", (d["config"].get("store_gather") or {}).get("collective"), "other", (d["config"].get("other_scaling") or {}).get("value"))
P
step "hierarchical smoother under the kernel trace" &&
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/$out/hier_trace --output-format csv -- python3 $root/benchmarks/hierarchical_smoother.py > $root/$out/hier.txt 2> $root/$out/hier.err) ; tail -2 $out/hier.txt
find $out/hier_trace -name '*kernel_stats.csv' -exec cp {} $out/hier_kernel_stats.csv \; ; head -14 $out/hier_kernel_stats.csv | cut -c1-150
step "band: RW2, lattice; truncated scan" &&
timeout -k 10 200 python3 benchmarks/band_profile.py > $out/band.jsonl 2> $out/band.err &&
timeout -k 10 400 python3 benchmarks/band_profile.py --lattice 100 --steps 5 >> $out/band.jsonl 2>> $out/band.err &&
timeout -k 10 400 python3 benchmarks/band_profile.py --lattice 100 --steps 5 --chains 256 >> $out/band.jsonl 2>> $out/band.err &&
timeout -k 10 200 python3 benchmarks/trunc_profile.py >> $out/band.jsonl 2>> $out/band.err && cat $out/band.jsonl
step done

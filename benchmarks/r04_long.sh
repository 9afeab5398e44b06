#!/bin/bash
# two-rank gloo rehearsal of bench.py --gpus 2 after the gather reordering, and the long-chain smoother under the kernel trace
out=${1:-gpurun_out/r04n}
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
OMC_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 20 --warmup 5 > $out/gpus2.json 2> $out/gpus2.err; echo "exit $?"
python3 - $out/gpus2.json <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["n_gpus"], "%.3e" % d["value"], d["config"]["store_gather"], d["config"]["check"])
P
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/$out/long_trace --output-format csv -- python3 $root/benchmarks/long_chain_smoother.py > $root/$out/long.txt 2> $root/$out/long.err); tail -3 $out/long.txt
find $out/long_trace -name '*kernel_stats.csv' -exec cp {} $out/long_kernel_stats.csv \;
python3 - $out/long_kernel_stats.csv <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "avg us %8.1f" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
P

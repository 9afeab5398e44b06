#!/bin/bash
# Round-4 secondary measurements on one box: cfg4 kernel trace, band route (RW2 at two warm-ups, 100 x 100 lattice), store summaries,
# one GPU's view of strong scaling.   gpurun --timeout 1200 -- 'bash benchmarks/r04_secondary.sh gpurun_out/r04h'
out=${1:-gpurun_out/r04h}
mkdir -p $out
export TMPDIR=/tmp
set -o pipefail
step() { echo "== $(date +%T) $*" | tee -a $out/progress.log; }
root=$PWD
step "cfg4 kernel trace" &&
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/$out/cfg4_trace --output-format csv -- python3 $root/bench.py --config cfg4 --steps 2048 --warmup 128 --no-cpu > $root/$out/cfg4.json 2> $root/$out/cfg4.err) &&
find $out/cfg4_trace -name '*kernel_stats.csv' -exec cp {} $out/cfg4_kernel_stats.csv \; && head -4 $out/cfg4_kernel_stats.csv | cut -c1-160 &&
step "band RW2 (default warm-up, 96)" &&
timeout -k 10 200 python3 benchmarks/band_profile.py > $out/band.jsonl 2> $out/band.err &&
timeout -k 10 200 python3 benchmarks/band_profile.py --overlap 96 >> $out/band.jsonl 2>> $out/band.err &&
step "lattice 100 x 100" &&
timeout -k 10 400 python3 benchmarks/band_profile.py --lattice 100 --steps 5 >> $out/band.jsonl 2>> $out/band.err &&
timeout -k 10 400 python3 benchmarks/band_profile.py --lattice 100 --steps 5 --chains 256 >> $out/band.jsonl 2>> $out/band.err &&
cat $out/band.jsonl &&
step "lattice kernel trace" &&
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $root/$out/lattice_trace --output-format csv -- python3 $root/benchmarks/band_profile.py --lattice 100 --steps 5 > /dev/null 2>&1) &&
find $out/lattice_trace -name '*kernel_stats.csv' -exec cp {} $out/lattice_kernel_stats.csv \; && head -5 $out/lattice_kernel_stats.csv | cut -c1-160 &&
step "store summaries" &&
timeout -k 10 400 python3 benchmarks/store_summaries.py > $out/store_summaries.json 2> $out/store.err && cat $out/store_summaries.json &&
step "strong scaling, one GPU's view" &&
bash benchmarks/strong_scaling_probe.sh $out/strong > $out/strong.txt 2>&1 && cat $out/strong.txt
step done

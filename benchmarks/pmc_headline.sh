#!/bin/bash
# PMC passes for the headline kernel (one counter group per pass; no trace domains besides --kernel-trace).
# Run from the repo root on the GPU box:  bash benchmarks/pmc_headline.sh gpurun_out/pmc
# The bench issues 32 sweeps per launch as ONE grid of 1024 self-restarting workgroups x 1024 lanes (1048576 threads;
# with run_reenter = 0 it is 32 x 1024 workgroups = 33554432): the summary keeps those launches only and reports
# per-launch means; benchmarks/pmc_summary.py divides by the sweeps per launch.
set -e
out=${1:-gpurun_out/pmc}
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i --output-format csv -- python3 bench.py --steps 64 --warmup 32 --no-cpu --no-kernel-events --condition-sweeps 256 --repeat-ms 0 --secondary-ms 0 > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out.p$i.log; }
done
python3 benchmarks/pmc_summary.py $out/p* --kernel k_tridiag_seg --grid 1048576 --sweeps-per-launch 32 > $out.summary.json
cat $out.summary.json

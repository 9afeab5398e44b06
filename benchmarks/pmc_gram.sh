#!/bin/bash
# Matrix-core counters for the own Gram kernel (k_gram_mfma), one counter group per pass:
#   bash benchmarks/pmc_gram.sh gpurun_out/pmc_gram
set -e
out=${1:-gpurun_out/pmc_gram}
export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i --output-format csv -- python3 benchmarks/gram_bench.py --reps 10 > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out.p$i.log; }
done
python3 benchmarks/pmc_summary.py $out/p* --kernel k_gram_mfma > $out.summary.json
cat $out.summary.json

#!/bin/bash
# Counters of the block product of omc_mala_run_white (k_dgemm_64x64), one group per pass:
#   bash benchmarks/pmc_cfg4.sh gpurun_out/pmc_cfg4
out=${1:-gpurun_out/pmc_cfg4}
export TMPDIR=/tmp
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i --output-format csv -- python3 bench.py --config cfg4 --steps 512 --warmup 64 --no-cpu > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out.p$i.log; }
done
python3 benchmarks/pmc_summary.py $out/p* --kernel k_dgemm_64x64 > $out.summary.json
cat $out.summary.json

#!/bin/bash
# PMC passes for the segmented band kernels (one counter group per pass; no trace domains besides --kernel-trace).
# Run from the repo root on the GPU box:  bash benchmarks/pmc_band.sh gpurun_out/pmc_band [band_profile.py flags]
set -e
out=${1:-gpurun_out/pmc_band}; shift || true
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i --output-format csv -- python3 benchmarks/band_profile.py --steps 4 "$@" > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out.p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_band_seg" not in k:
            continue
        name = k[k.index("k_band_seg"):k.index(">") + 1]
        if int(r["Grid_Size"]) < 64 * 64:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c, v in sorted(acc[name].items()):
        v = sorted(v)
        big = [x for x in v if x > 0.2 * v[-1]] or v  # (the gated second attempts return at once)
        print(f"   {c:24s} per launch {sum(big) / len(big):16.0f}   ({len(big)} launches)")
PY

#!/bin/bash
# narrow bands (4 <= w <= 15) on 10 000-node lattices: lane-per-chain kernel (algo 1, w <= 8) against the blocked workgroup-per-chain
# kernel (algo 3) with four waves per chain (default below w = 16) and with eight (OMC_BLOCKED_THREADS=512)
for chains in 64 256 1024 4096; do
  for k in 4 8 12 15; do
    r=$((10000 / k))
    for cfg in "1 0" "3 0" "3 512"; do
      set -- $cfg
      [ $1 = 1 ] && [ $k -gt 8 ] && continue
      OMC_BLOCKED_THREADS=$2 timeout -k 10 200 python3 benchmarks/band_profile.py --lattice $k --rows $r --steps 3 --algo $1 --chains $chains 2>/dev/null | tail -1 > /tmp/bc.json
      python3 - $k $1 $chains $2 <<'P'
import json, sys
d = json.loads(open("/tmp/bc.json").read())
print("chains", sys.argv[3], "w", sys.argv[1], "algo", sys.argv[2], "threads", sys.argv[4] if sys.argv[2] == "3" and sys.argv[4] != "0" else ("256" if sys.argv[2] == "3" else "-"), "ms_per_draw %.2f" % d["ms_per_draw"])
P
    done
  done
done

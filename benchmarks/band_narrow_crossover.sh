#!/bin/bash
# narrow bands (4 <= w <= 8) on 10 000-node lattices: lane-per-chain kernel (default) against the blocked workgroup-per-chain kernel (algo 3)
for chains in 64 256 1024; do
  for k in 4 6 8; do
    r=$((10000 / k))
    for algo in 1 3; do
      timeout -k 10 200 python3 benchmarks/band_profile.py --lattice $k --rows $r --steps 3 --algo $algo --chains $chains 2>/dev/null | tail -1 > /tmp/bc.json
      python3 - $k $algo $chains <<'P'
import json, sys
d = json.loads(open("/tmp/bc.json").read())
print("chains", sys.argv[3], "w", sys.argv[1], "algo", sys.argv[2], "ms_per_draw %.2f" % d["ms_per_draw"])
P
    done
  done
done

#!/bin/bash
# blocked (band_algo 3) against column-at-a-time (2) workgroup-per-chain band kernels on R x K lattices of 10 000 nodes
for k in 9 12 16 24 32 64; do
  r=$((10000 / k))
  for algo in 3 2; do
    timeout -k 10 200 python3 benchmarks/band_profile.py --lattice $k --rows $r --steps 3 --algo $algo --chains 256 2>/dev/null | tail -1 > /tmp/bc.json
    python3 - $k $algo <<'P'
import json, sys
d = json.loads(open("/tmp/bc.json").read())
print("w", sys.argv[1], "algo", sys.argv[2], "ms_per_draw %.2f" % d["ms_per_draw"])
P
  done
done

#!/bin/bash
# blocked band kernel with look-ahead: parity tests, stamps, lattice, narrow-lattice crossover
set -o pipefail
mkdir -p gpurun_out/r04q
timeout -k 10 600 python3 -m pytest tests/test_band_gpu.py tests/test_round4_edges_gpu.py tests/test_band_seg_gpu.py -x -q -m gpu > gpurun_out/r04q/pytest_band.log 2>&1 || { tail -30 gpurun_out/r04q/pytest_band.log; exit 1; }
tail -2 gpurun_out/r04q/pytest_band.log
OMC_WIDE_STAMPS=1 timeout -k 10 200 python3 benchmarks/band_profile.py --lattice 100 --chains 1024 > gpurun_out/r04q/lattice.txt 2>&1 && grep -v amdgpu.ids gpurun_out/r04q/lattice.txt
bash benchmarks/band_crossover.sh > gpurun_out/r04q/crossover.txt 2>&1; cat gpurun_out/r04q/crossover.txt

#!/bin/bash
# Build a variant of libomcmc_hip.so with extra -D flags on omc_tridiag.hip (A/B timing, benchmarks/ab_headline.py):
#   bash benchmarks/build_variant.sh NAME "-DOMC_WHATIF_NOSTORE=1"   ->  build/ab/libomcmc_hip_NAME.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2
mkdir -p $root/build/ab
make -s -C $root/openmcmc_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$root/include -Wall -Wno-unused-result -Wno-unused-value -mllvm -instcombine-max-copied-from-constant-users=100000 $flags \
  -c $root/openmcmc_amd/csrc/omc_tridiag.hip -o $root/build/ab/omc_tridiag_$name.o
objs=$(ls $root/openmcmc_amd/csrc/*.o | grep -v omc_tridiag.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $root/build/ab/omc_tridiag_$name.o -o $root/build/ab/libomcmc_hip_$name.so -L/opt/rocm/lib -lrocblas -lrocsolver -lrccl
rm -f $root/build/ab/omc_tridiag_$name.o
echo built build/ab/libomcmc_hip_$name.so

#!/bin/bash
# Build libomcmc_hip.so as of a git revision into build/ab/libomcmc_hip_NAME.so (A/B timing against the working tree):
#   bash benchmarks/build_rev.sh HEAD~3 before
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
rev=$1; name=$2
tmp=$(mktemp -d)
git -C $root archive $rev openmcmc_amd/csrc include | tar -x -C $tmp
make -s -j4 -C $tmp/openmcmc_amd/csrc ROOT=$tmp OUT=$tmp/lib.so > /dev/null
mkdir -p $root/build/ab
cp $tmp/lib.so $root/build/ab/libomcmc_hip_$name.so
rm -rf $tmp
echo built build/ab/libomcmc_hip_$name.so from $rev

#!/bin/bash
# an alternative build of the library with ONE source compiled with extra flags (experiments; A/B with scratch/ab_libs.sh):
#   bash scratch/build_alt.sh <name> <source-stem> "<extra flags>"   ->  scratch/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; stem=$2; flags=$3
make -j16 all > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $flags -x hip -c buctd_amd/csrc/$stem.hip -o /tmp/alt_${name}_$stem.o
objs=$(ls buctd_amd/csrc/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_$name.so $objs /tmp/alt_${name}_$stem.o
echo built scratch/lib_$name.so

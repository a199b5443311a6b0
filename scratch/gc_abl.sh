#!/bin/bash
# builds scratch/libbuctd_gcabl_<k>.so: the product library with conv_gather_x6.hip compiled with -DGC_ABL=k
set -e
cd $(dirname $0)/..
for k in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DGC_ABL=$k -x hip -c buctd_amd/csrc/conv_gather_x6.hip -o /tmp/cg_abl_$k.o
    objs=$(ls buctd_amd/csrc/*.o | grep -v conv_gather_x6.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libbuctd_gcabl_$k.so $objs /tmp/cg_abl_$k.o ) &
done
wait

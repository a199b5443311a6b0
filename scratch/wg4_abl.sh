#!/bin/bash
# builds scratch/wg4_abl_<k>.so: conv3x3_wgrad4.hip alone with -DW4_ABL=k (ablations of the planes weight-gradient kernel)
set -e
cd $(dirname $0)/..
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DW4_ABL=$k -x hip -shared -Wl,-Bsymbolic -fvisibility=hidden \
    buctd_amd/csrc/conv3x3_wgrad4.hip buctd_amd/csrc/error.cpp -o scratch/wg4_abl_$k.so &
done
wait

#!/bin/bash
# builds scratch/mha_abl_<k>.so: attn_mha.hip alone with -DMHA_ABL=k (what-if switches of mha_fwd_x6q_kernel)
set -e
cd $(dirname $0)/..
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMHA_ABL=$k -DMHA_ABL_BUILD -x hip -shared -Wl,-Bsymbolic -fvisibility=hidden \
    buctd_amd/csrc/attn_mha.hip buctd_amd/csrc/error.cpp -o scratch/mha_abl_$k.so &
done
wait

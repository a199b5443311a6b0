#!/bin/bash
# ablations of the 3x3 weight-gradient kernel (scratch/patches/r6_wgrad_ablation_switches.patch builds; WGX bits: 1 no MFMA,
# 2 staging of the first stage only, 4 linear positions, 8 no slab store, 16 no B fragment reads): group {0,1} and single 0
cd "$(dirname "$0")/.."
for l in product "$@"; do
  echo "== $l"
  if [ $l = product ]; then timeout 200 python scratch/time_group_wgrad.py w48 32 01 0 23 2>&1 | grep -v amdgpu.ids
  else timeout 200 python scratch/run_alt.py lib_$l.so scratch/time_group_wgrad.py w48 32 01 0 23 2>&1 | grep -v amdgpu.ids; fi
done

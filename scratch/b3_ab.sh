#!/bin/bash
# tile plans of the low-resolution branches: the product build against scratch/lib_<name>.so builds (scratch/build_alt.sh)
cd "$(dirname "$0")/.."
for l in product "$@"; do
  echo "== $l"
  if [ $l = product ]; then timeout 300 python scratch/time_c3.py w48 32 3 23 2 0123 012 2>&1 | grep -v amdgpu.ids
  else timeout 300 python scratch/run_alt.py lib_$l.so scratch/time_c3.py w48 32 3 23 2 0123 012 2>&1 | grep -v amdgpu.ids; fi
done

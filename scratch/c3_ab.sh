#!/bin/bash
# the train-mode 3x3 launches of the product build against scratch/lib_<name>.so builds: bash scratch/c3_ab.sh <sets...> -- <names...>
cd "$(dirname "$0")/.."
sets=(); while [ "$1" != "--" ] && [ -n "$1" ]; do sets+=("$1"); shift; done; shift
for l in product "$@"; do
  echo "== $l"
  if [ $l = product ]; then timeout 300 python scratch/time_c3.py w48 32 "${sets[@]}" 2>&1 | grep -v amdgpu.ids
  else timeout 300 python scratch/run_alt.py lib_$l.so scratch/time_c3.py w48 32 "${sets[@]}" 2>&1 | grep -v amdgpu.ids; fi
done

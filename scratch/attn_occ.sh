#!/bin/bash
# builds scratch/libbuctd_occ_<q>_<kv>.so: the product library with attn_smallqk.hip compiled for other occupancy caps
set -e
cd $(dirname $0)/..
for v in "$@"; do
  q=${v%_*}; kv=${v#*_}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-pass-failed -DATT_BQ_OCC=$q -DATT_BKV_OCC=$kv -x hip -c buctd_amd/csrc/attn_smallqk.hip -o /tmp/attn_smallqk_$v.o
  objs=$(ls buctd_amd/csrc/*.o | grep -v attn_smallqk.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libbuctd_occ_$v.so $objs /tmp/attn_smallqk_$v.o
done

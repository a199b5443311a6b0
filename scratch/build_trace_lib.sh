#!/bin/bash
# builds scratch/libbuctd_hip_trace.so (run before the sweep / trace scripts of this directory)
# builds scratch/libbuctd_hip_trace.so = the product library with every kernel file compiled -DBUCTD_TUNING (the experiment
# switches BUCTD_C3_FORCE / BUCTD_C3_LEAN / BUCTD_C3_COLMAJOR / BUCTD_WG3_SPLIT / BUCTD_GX_ROWMAJOR / BUCTD_FWD_THIN /
# BUCTD_WGRAD_THIN exist only in this build) and conv3x3.hip additionally -DC3_TRACE (cycle stamps).
# Scripts select it with  _C.LIB_PATH = "scratch/libbuctd_hip_trace.so"  (see scratch/c3_trace.py) or BUCTD_LIB=<path>.
set -e
cd $(dirname $0)/..
mkdir -p /tmp/buctd_tune
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DBUCTD_TUNING"
for f in buctd_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""; [ "$b" = conv3x3 ] && extra="-DC3_TRACE"
  /opt/rocm/bin/hipcc $FLAGS $extra -x hip -c $f -o /tmp/buctd_tune/$b.o &
done
wait
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -c buctd_amd/csrc/error.cpp -o /tmp/buctd_tune/error.o 2>/dev/null || cp buctd_amd/csrc/error.o /tmp/buctd_tune/error.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libbuctd_hip_trace.so /tmp/buctd_tune/*.o

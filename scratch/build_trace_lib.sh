#!/bin/bash
# builds buctd_amd/lib/libbuctd_hip_trace.so = the product library with conv3x3.hip compiled -DC3_TRACE (cycle stamps)
set -e
cd $(dirname $0)/..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DC3_TRACE -x hip -c buctd_amd/csrc/conv3x3.hip -o /tmp/conv3x3_trace.o
objs=$(ls buctd_amd/csrc/*.o | grep -v "/conv3x3.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libbuctd_hip_trace.so $objs /tmp/conv3x3_trace.o

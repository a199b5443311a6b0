#!/bin/bash
# usage: scratch/pcsamp_run.sh <tag> <method: stochastic|host_trap> <one_op args...>  -> gpurun_out/pcs_<tag>/
tag=$1; method=$2; shift; shift
root=$(pwd); out=$root/gpurun_out/pcs_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
unit=cycles; interval=1048576
if [ "$method" = host_trap ]; then unit=time; interval=1; fi
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit \
  --pc-sampling-interval $interval --kernel-trace --output-format csv -d /tmp/pcs_$tag -o p -- python $root/scratch/one_op.py "$@" > $out/run.log 2>&1
echo rc $? >> $out/run.log
find /tmp/pcs_$tag -type f | head -20 >> $out/run.log
for f in $(find /tmp/pcs_$tag -name "*.csv"); do sz=$(stat -c %s $f); if [ $sz -lt 30000000 ]; then cp $f $out/$(basename $f); else head -c 30000000 $f > $out/$(basename $f); fi; done
cd $root

#!/bin/bash
# usage: scratch/pmc_gconv.sh <tag> N H W Ci Co k stride pad [d]  -> gpurun_out/pmc_<tag>/  (counters of gconv_x6_kernel, time_gconv.py workload)
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  PMC_SHORT=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcg_${tag}_$name -o p -- python $root/scratch/time_gconv.py $ARGS > $out/$name.log 2>&1
  for f in $(find /tmp/pmcg_${tag}_$name -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do cp $f $out/${name}_$(basename $f); done
}
ARGS="$*"
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM
run c GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $root

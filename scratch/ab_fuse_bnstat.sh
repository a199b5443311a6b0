# the fuse rows' BatchNorm sums formed by fuse_sum_bwd (1) against a reduction launch per BatchNorm (0): test, then interleaved A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_streams.py tests/test_gpu_step_graph.py -x -q 2>&1 | tail -15 > gpurun_out/ab_fuse_bnstat.txt
export BUCTD_TUNING=1
bash scratch/ab_env3.sh ${1:-3} ${2:-train_c4} "BUCTD_FUSE_BWD_BNSTAT=1" "BUCTD_FUSE_BWD_BNSTAT=0" >> gpurun_out/ab_fuse_bnstat.txt 2>&1
cat gpurun_out/ab_fuse_bnstat.txt

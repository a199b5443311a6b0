# interleaved A/B: decoded statistics written to pinned memory by the kernels (1) against copy-engine transfers (0)
export BUCTD_TUNING=1
bash scratch/ab_env3.sh ${1:-3} ${2:-train_c4} "BUCTD_STATS_ZERO_COPY=1" "BUCTD_STATS_ZERO_COPY=0" > gpurun_out/ab_stats.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_entry.py tests/test_gpu_step_graph.py -x -q 2>&1 | tail -3 >> gpurun_out/ab_stats.txt
cat gpurun_out/ab_stats.txt

mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_step_graph.py -x -q 2>&1 | tail -25 > gpurun_out/fg_tests.txt
: > gpurun_out/fg_batches.txt
for wl in train_c2 train_c4 infer_c5; do
  timeout 300 python scratch/forward_graph_probe.py --workload $wl 2>&1 | grep "eval forward\|Error\|error" >> gpurun_out/fg_batches.txt
done
tail -5 gpurun_out/fg_tests.txt; cat gpurun_out/fg_batches.txt

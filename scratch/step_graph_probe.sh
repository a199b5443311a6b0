#!/bin/bash
# one gpurun call: the StepGraph tests, then eager vs linear graph by batch size
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_step_graph.py -x -q 2>&1 | tail -30 > gpurun_out/sg_tests.txt
: > gpurun_out/sg_batches.txt
for wl in train_c2 train_c3; do for b in 2 4 8 16 32; do
  timeout 300 python scratch/step_graph_probe.py --workload $wl --batch $b --steps 15 2>&1 | grep "img/s" | tail -2 >> gpurun_out/sg_batches.txt
done; done
for b in 2 8; do
  timeout 300 python scratch/step_graph_probe.py --workload train_c4 --repeat-masks --batch $b --steps 15 2>&1 | grep "img/s" | tail -2 | sed 's/$/ [masks repeat: timing only]/' >> gpurun_out/sg_batches.txt
done
tail -3 gpurun_out/sg_tests.txt; cat gpurun_out/sg_batches.txt

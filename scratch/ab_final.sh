mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_branches.py tests/test_gpu_step_graph.py -x -q 2>&1 | tail -6 > gpurun_out/ab_final.txt
bash scratch/forward_graph_probe.sh > /dev/null 2>&1; cat gpurun_out/fg_batches.txt >> gpurun_out/ab_final.txt
bash scratch/ab_trees.sh 4 train_c2 . scratch/base_tree 2>&1 | tail -2 >> gpurun_out/ab_final.txt
bash scratch/ab_trees.sh 2 infer_c5 . scratch/base_tree 2>&1 | tail -2 >> gpurun_out/ab_final.txt
cat gpurun_out/ab_final.txt

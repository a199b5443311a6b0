# the PCIe-inclusive rate: batch resident in HBM / copied on the compute stream at the top of the step / prefetched beside the step
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_entry.py -x -q -k "prefetch or train_entry" 2>&1 | tail -4 > gpurun_out/ab_host_input.txt
W=${1:-train_c4}
for r in 1 2 3; do for m in resident copy prefetch; do
  v=$(timeout 200 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --host-input $m 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$m $v"
done; done | tee -a gpurun_out/ab_host_input.txt
cat gpurun_out/ab_host_input.txt

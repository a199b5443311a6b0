# collects the round-4 profile artefacts into gpurun_out/r04 (copied to profiles/ afterwards)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
for w in train_c4 train_c3 train_c2 infer_c5; do
  extra=""; [ $w = train_c4 ] || extra="--no-cpu-baseline"
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 $extra 2>/dev/null | tail -1 > $O/bench_line_$w.json
done
BATCH=32 timeout 200 python scratch/cpu_split.py 2>/dev/null | grep batch > $O/host_split.txt
BATCH=2 timeout 200 python scratch/cpu_split.py 2>/dev/null | grep batch >> $O/host_split.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr4 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > /tmp/tr4.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tr4 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scratch/trace_csv_stats.py $f > $O/kernel_trace_stats.txt 2>&1 && python scratch/timeline_gaps.py $f > $O/timeline.txt 2>&1 && python scratch/critical_path.py $f 1 > $O/critical_path.txt 2>&1
bash scratch/pmc_run.sh r4fwd bf16x6 fwd > /dev/null 2>&1; python scratch/pmc_summary.py gpurun_out/pmc_r4fwd conv3x3 > $O/pmc_fwd.txt 2>&1
bash scratch/pmc_run.sh r4wg bf16x6 wgrad > /dev/null 2>&1; python scratch/pmc_summary.py gpurun_out/pmc_r4wg "" > $O/pmc_wgrad.txt 2>&1
bash scratch/pmc_run.sh r4dg bf16x6 dgrad > /dev/null 2>&1; python scratch/pmc_summary.py gpurun_out/pmc_r4dg conv3x3 > $O/pmc_dgrad.txt 2>&1
python scratch/pmc_traffic_json.py gpurun_out $O/kernel_trace_stats.txt $O $(date +%F) > $O/traffic_line.txt 2>&1
bash scratch/serial_census.sh r04 > /dev/null 2>&1
ls -la $O

# collects the round-6 profile artefacts into gpurun_out/r06 (scratch/r06_collect.py copies them to profiles/ afterwards)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
D=$(date +%F)
BATCH=32 timeout 200 python scratch/cpu_split.py 2>/dev/null | grep batch > $O/host_split.txt
BATCH=2 timeout 200 python scratch/cpu_split.py 2>/dev/null | grep batch >> $O/host_split.txt
cd /tmp && export TMPDIR=/tmp
for t in c4:train_c4 c3:train_c3 c2:train_c2; do
  tag=${t%%:*}; w=${t##*:}
  rm -rf /tmp/tr6_$tag
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr6_$tag -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > /tmp/tr6_$tag.log 2>&1
  f=$(find /tmp/tr6_$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/scratch/in_step_json.py $f $O/in_step_kernel_us_$tag.json "profiles/r06_kernel_trace_stats_bench_$tag.txt (rocprofv3 --kernel-trace of python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer; scratch/in_step_json.py)" $D
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/scratch/trace_csv_stats.py $f > $O/kernel_trace_stats_$tag.txt 2>&1
  if [ $tag = c4 ] && [ -n "$f" ]; then
    python $GRAFT_REPO_ROOT/scratch/timeline_gaps.py $f > $O/timeline.txt 2>&1
    python $GRAFT_REPO_ROOT/scratch/critical_path.py $f 1 > $O/critical_path.txt 2>&1
    python $GRAFT_REPO_ROOT/scratch/trace_by_grid.py $f conv3x3 wg3 gconv bn_ > $O/by_grid.txt 2>&1
  fi
done
cd $GRAFT_REPO_ROOT
bash scratch/pmc_run.sh r6fwd bf16x6 fwd > /dev/null 2>&1; python scratch/pmc_summary.py gpurun_out/pmc_r6fwd conv3x3 > $O/pmc_fwd.txt 2>&1
bash scratch/pmc_run.sh r6wg bf16x6 wgrad > /dev/null 2>&1; python scratch/pmc_summary.py gpurun_out/pmc_r6wg "" > $O/pmc_wgrad.txt 2>&1
bash scratch/pmc_run.sh r6dg bf16x6 dgrad > /dev/null 2>&1; python scratch/pmc_summary.py gpurun_out/pmc_r6dg conv3x3 > $O/pmc_dgrad.txt 2>&1
bash scratch/pmc_run.sh r6fwdg bf16x6 fwd_group > /dev/null 2>&1; python scratch/pmc_summary.py gpurun_out/pmc_r6fwdg conv3x3 > $O/pmc_fwd_group.txt 2>&1
bash scratch/pmc_run.sh r6wgg bf16x6 wgrad_group > /dev/null 2>&1; python scratch/pmc_summary.py gpurun_out/pmc_r6wgg "" > $O/pmc_wgrad_group.txt 2>&1
python scratch/pmc_traffic_json.py gpurun_out $O/kernel_trace_stats_c4.txt $O $D > $O/traffic_line.txt 2>&1
bash scratch/serial_census.sh r06 > /dev/null 2>&1
# the bench lines quote the census, the PMC traffic and the in-step durations of THIS tree: everything above goes to profiles/
# first (on this box's copy), then the lines
python scratch/r06_collect.py > /dev/null 2>&1
for w in train_c4 train_c3 train_c2 infer_c5; do
  extra=""; [ $w = train_c4 ] || extra="--no-cpu-baseline"
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 $extra 2>/dev/null | tail -1 > $O/bench_line_$w.json
done
timeout 300 python bench.py --condition mono --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_train_c4_mono.json
ls -la $O

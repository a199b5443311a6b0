# kernel-trace stats + timeline of bench.py with the group launches off / on, on one box: bash scratch/trace_ab.sh [workload]
cd $GRAFT_REPO_ROOT
W=${1:-train_c4}
for m in off on; do
  O=$GRAFT_REPO_ROOT/gpurun_out/trab_$m; mkdir -p $O
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trab_$m && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trab_$m -o t -- python $GRAFT_REPO_ROOT/scratch/bench_ab.py $m --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > /tmp/trab_$m.log 2>&1)
  f=$(find /tmp/trab_$m -name "*kernel_trace.csv" | head -1)
  python scratch/trace_csv_stats.py $f > $O/kernel_trace_stats.txt 2>&1
  python scratch/timeline_gaps.py $f > $O/timeline.txt 2>&1
  tail -1 /tmp/trab_$m.log | cut -c1-120
done

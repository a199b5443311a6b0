# kernel trace of the default bench + critical path + stats into gpurun_out/<tag>
cd $GRAFT_REPO_ROOT
tag=${1:-cp}; O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > /tmp/tr_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then
  python scratch/critical_path.py $f 1 > $O/critical_path.txt 2>&1
  python scratch/critical_path.py $f 3 >> $O/critical_path.txt 2>&1
  python scratch/trace_csv_stats.py $f > $O/kernel_trace_stats.txt 2>&1
  python scratch/timeline_gaps.py $f > $O/timeline.txt 2>&1
fi
tail -2 /tmp/tr_$tag.log | cut -c1-200

# kernel-trace stats of two trees on one box: bash scratch/trace_two.sh <treeA> <treeB>
cd $GRAFT_REPO_ROOT
for tr in "$@"; do
  tag=$(echo $tr | tr '/.' '__'); O=$GRAFT_REPO_ROOT/gpurun_out/tr_$tag; mkdir -p $O
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$tag && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- python $GRAFT_REPO_ROOT/$tr/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > /tmp/tr_$tag.log 2>&1)
  f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
  python scratch/trace_csv_stats.py $f > $O/kernel_trace_stats.txt 2>&1
  python scratch/timeline_gaps.py $f > $O/timeline.txt 2>&1
  tail -1 /tmp/tr_$tag.log | cut -c1-120
done

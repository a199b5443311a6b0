cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_c5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_c5 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload infer_c5 --steps 10 --warmup 3 > /tmp/tr_c5.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tr_c5 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -32 $f | cut -c1-200 > $O/kernel_stats_head.txt
tail -1 /tmp/tr_c5.log | cut -c1-200

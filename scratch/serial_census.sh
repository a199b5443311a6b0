# kernel census of the train step with every stream serialised (kernel durations ~ solo): gpurun_out/<tag>/serial_stats.txt
cd $GRAFT_REPO_ROOT
tag=${1:-ser}; W=${2:-train_c4}; O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
export BUCTD_TUNING=1 BUCTD_WGRAD_STREAM=0 BUCTD_BRANCH_STREAMS=0
timeout 200 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serialised streams:', d['value'], d['ms_per_step'])" > $O/serial_bench.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trs_$tag
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs_$tag -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > /tmp/trs_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/trs_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scratch/trace_csv_stats.py $f > $O/serial_stats.txt 2>&1 && python scratch/timeline_gaps.py $f 2>&1 | head -12 > $O/serial_timeline.txt
cat $O/serial_bench.txt; head -60 $O/serial_stats.txt

cd $GRAFT_REPO_ROOT
run() { wl=$1; shift; echo "== $wl $*"; env "$@" timeout 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run train_c2 A=1
run train_c2 BUCTD_WGRAD_STREAM=0
run train_c2 BUCTD_BRANCH_STREAMS=0
run train_c2 BUCTD_WGRAD_STREAM=0 BUCTD_BRANCH_STREAMS=0
run train_c4 A=1
run train_c3 A=1

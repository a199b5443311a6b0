cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 200 python scratch/run_alt.py libbuctd_hip_trace.so bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run A=1
run BUCTD_SKIP=8
run BUCTD_SKIP=16
run BUCTD_SKIP=32
run BUCTD_SKIP=56
run BUCTD_SKIP=63
run BUCTD_SKIP=63 BUCTD_BRANCH_STREAMS=0

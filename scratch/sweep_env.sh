cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run BUCTD_PLANES=0
run BUCTD_PLANES=1
run BUCTD_PLANES=0 GPU_MAX_HW_QUEUES=8
run BUCTD_PLANES=0 GPU_MAX_HW_QUEUES=8 BUCTD_BRANCH_MAX=3
run BUCTD_PLANES=0 GPU_MAX_HW_QUEUES=8 BUCTD_BRANCH_MAX=3 BUCTD_WGRAD_STREAMS=2
run BUCTD_PLANES=1 GPU_MAX_HW_QUEUES=8 BUCTD_BRANCH_MAX=3

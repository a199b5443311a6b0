# sweep of one environment variable on the default bench with an alternative library: bash scratch/ab_env2.sh LIB VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
lib=$1; var=$2; shift 2
for v in "$@"; do
  r=$(env $var=$v timeout 200 python scratch/run_alt.py $lib bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$var=$v: $r"
done

# A/B of an environment variable on the default bench: bash scratch/ab_env.sh VAR v1 v2 [repeats]
cd $GRAFT_REPO_ROOT
var=$1; a=$2; b=$3; n=${4:-2}
for i in $(seq $n); do
  for v in $a $b; do
    r=$(env $var=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$var=$v: $r"
  done
done

# like ab_env3.sh, but each setting is a comma-separated list of VAR=value pairs: bash scratch/ab_env4.sh <rounds> <workload> "A=1,B=2" ...
cd $GRAFT_REPO_ROOT
rounds=$1; W=$2; shift 2
for r in $(seq $rounds); do
  for e in "$@"; do
    v=$(env $(echo $e | tr ',' ' ') timeout 200 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
    echo "$e $v"
  done
done | tee /tmp/ab.txt
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    k, v = l.split(); d[k].append(float(v))
for k, v in d.items(): print(f"{k}: median {statistics.median(v):.1f} img/s  min {min(v):.1f} max {max(v):.1f}  n={len(v)}")
PY

# interleaved A/B of environment settings on one box: bash scratch/ab_env3.sh <rounds> <workload> "VAR=a" "VAR=b" ...
cd $GRAFT_REPO_ROOT
rounds=$1; W=$2; shift 2
for r in $(seq $rounds); do
  for e in "$@"; do
    v=$(env $e timeout 200 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
    echo "$(echo $e | tr ' ' ',') $v"
  done
done | tee /tmp/ab.txt
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    k, v = l.split(); d[k].append(float(v))
for k, v in d.items(): print(f"{k}: median {statistics.median(v):.1f} img/s  min {min(v):.1f} max {max(v):.1f}  n={len(v)}")
PY

# interleaved A/B of engine switches on one box (scratch/bench_ab.py specs):  bash scratch/ab_specs.sh <rounds> <workload> spec1 spec2 ...
# ("on" = the shipped defaults).  Prints img/s per run and the medians.
cd $GRAFT_REPO_ROOT
rounds=$1; W=$2; shift 2
for r in $(seq $rounds); do
  for spec in "$@"; do
    v=$(timeout 200 python scratch/bench_ab.py $spec --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
    echo "$spec $v"
  done
done | tee /tmp/abs.txt
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/abs.txt'):
    k, v = l.split(); d[k].append(float(v))
for k, v in d.items(): print(f"{k}: median {statistics.median(v):.1f} img/s  min {min(v):.1f} max {max(v):.1f}  n={len(v)}")
PY

# interleaved A/B of whole trees on one box:  bash scratch/ab_trees.sh <rounds> <workload> <treeA> <treeB> ...   ("." = this tree)
cd $GRAFT_REPO_ROOT
rounds=$1; W=$2; shift 2
for r in $(seq $rounds); do
  for tr in "$@"; do
    v=$(cd $tr && timeout 200 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
    echo "$tr $v"
  done
done | tee /tmp/ab.txt
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    k, v = l.split(); d[k].append(float(v))
for k, v in d.items(): print(f"{k}: median {statistics.median(v):.1f} img/s  min {min(v):.1f} max {max(v):.1f}  n={len(v)}")
PY

"""Per-kernel summary of a rocprofv3 --kernel-trace CSV (steady-state steps between Adam launches):
python scratch/trace_csv_stats.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if k[2].startswith("adam_kernel")]
sel = ks[adam[3] + 1:adam[-1] + 1]
nsteps = len(adam) - 4
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    a = agg[n[:100]]; a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"{nsteps} steady-state steps; kernel time {tot/1e6/nsteps:.2f} ms per step (sum of durations, concurrent kernels overlap), {sum(v[0] for v in agg.values())//nsteps} launches per step")
print(f"{'kernel':100s} {'calls/step':>10s} {'ms/step':>9s} {'avg_us':>8s} {'share':>6s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n:100s} {c/nsteps:10.1f} {t/1e6/nsteps:9.3f} {t/1e3/c:8.1f} {100*t/tot:5.1f}%")

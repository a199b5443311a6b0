"""Per-kernel summary of a whole rocprofv3 --kernel-trace CSV (no step splitting): python scratch/trace_csv_all.py <csv> [skip_first_n_kernels]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
ks = ks[len(ks) // 2:]          # second half: steady state
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ks:
    a = agg[n[:100]]; a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
wall = ks[-1][1] - ks[0][0]
print(f"second half of the trace: {len(ks)} launches, kernel time {tot/1e6:.1f} ms in {wall/1e6:.1f} ms of wall clock")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{n:100s} {c:6d} {t/1e6:9.2f} ms {t/1e3/c:8.1f} us {100*t/tot:5.1f}%")

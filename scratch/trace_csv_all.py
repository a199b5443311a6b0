"""Per-kernel summary of a whole rocprofv3 --kernel-trace CSV divided by a run count: python scratch/trace_csv_all.py <csv> <runs> [rows]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
runs = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[(r["Kernel_Name"][:90], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"kernel time {tot / 1e6 / runs:.2f} ms per run, {sum(v[0] for v in agg.values()) / runs:.0f} launches per run")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / 1e3 / runs:9.1f} us/run  n/run {c / runs:5.1f}  avg {t / 1e3 / c:8.1f} us  grid {k[1]}x{k[2]}x{k[3]}  {k[0]}")

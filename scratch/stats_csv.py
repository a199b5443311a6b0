"""rocprofv3 --stats kernel_stats.csv -> per-run table: python scratch/stats_csv.py <kernel_stats.csv> <runs> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
runs = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per run {tot / runs / 1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f"{float(r['TotalDurationNs']) / runs / 1e3:9.1f} us/run  calls/run {int(r['Calls']) / runs:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")

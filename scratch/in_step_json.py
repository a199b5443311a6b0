"""rocprofv3 --kernel-trace CSV of `python bench.py ...` -> the JSON bench.py quotes its in-step kernel durations from
(profiles/r05_in_step_kernel_us_<tag>.json): per kernel name and per (kernel name, grid), steady-state steps only.
   python scratch/in_step_json.py <kernel_trace.csv> <out.json> <source text> <date>"""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
              "%s,%s,%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if k[2].startswith("adam_kernel")]
sel = ks[adam[3] + 1:adam[-1] + 1]
n = len(adam) - 4
by_name, by_grid = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
for s, e, nm, grid in sel:
    a = by_name[nm[:120]]; a[0] += 1; a[1] += e - s
    b = by_grid[nm[:120] + "|" + grid]; b[0] += 1; b[1] += e - s
def pack(d, top):
    out = {}
    for k, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:top]:
        out[k] = {"calls_per_step": round(c / n, 2), "ms_per_step": round(t / 1e6 / n, 4), "avg_us": round(t / 1e3 / c, 1)}
    return out
json.dump({"date": sys.argv[4], "source": sys.argv[3], "steps": n, "launches_per_step": round(len(sel) / n, 1),
           "grid_unit": "threads (Grid_Size_X,Y,Z of the trace)", "kernels": pack(by_name, 60), "by_grid": pack(by_grid, 120)},
          open(sys.argv[2], "w"), indent=1)

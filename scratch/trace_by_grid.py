"""Per (kernel, grid) census of a rocprofv3 --kernel-trace CSV over the steady-state steps: python scratch/trace_by_grid.py <csv> [name filter ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2:]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z")), r.get("LDS_Block_Size") or "") for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if k[2].startswith("adam_kernel")]
sel = ks[adam[3] + 1:adam[-1] + 1]
n = len(adam) - 4
agg = collections.defaultdict(lambda: [0, 0])
for s, e, nm, grid, lds in sel:
    if flt and not any(f in nm for f in flt):
        continue
    a = agg[(nm[:60], grid, lds)]; a[0] += 1; a[1] += e - s
for (nm, grid, lds), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{nm:60s} grid {str(grid):28s} lds {lds:>6s} {c / n:6.1f}/step {t / 1e6 / n:8.3f} ms/step {t / 1e3 / c:8.1f} us")

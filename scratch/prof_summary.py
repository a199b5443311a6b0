"""Summarise a rocprofv3 rocpd database: per-kernel total/avg time, share. usage: prof_summary.py db [steps]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels").fetchall() if 'name' in cols else None
if rows is None:
    print(cols); sys.exit()
agg = {}
for name, s, e in rows:
    n = re.sub(r"\(.*", "", name)
    n = n.replace("void ", "")
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} launches; per step ({steps:g} steps): {tot/1e3/steps:.2f} ms, {len(rows)/steps:.0f} launches")
print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'share':>6s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n[:110]:110s} {c:7d} {t/1e3:10.2f} {t/c:9.1f} {100*t/tot:5.1f}%")

import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib
os.environ["SWEEP_NO_MAIN"] = "1"
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_c3_plan.py")).read().split("shapes = [")[0]
exec(src)
for shp in [(32,24,18,192,192),(32,12,9,384,384),(32,48,36,96,96)]:
    line = f"{shp}:"
    for cm in ("0", "1"):
        os.environ["BUCTD_C3_COLMAJOR"] = cm
        for force in (None, "2,1", "4,1", "2,2", "4,2"):
            r = run(*shp, force)
            line += f" | cm{cm} {force or 'default'}: " + ("n/a" if r is None else f"{r[0]:.1f}")
    print(line, flush=True)

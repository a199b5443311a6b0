"""HBM traffic per launch of the roofline kernels from the FETCH_SIZE / WRITE_SIZE passes of scratch/pmc_run.sh, and their
in-step kernel durations from a rocprofv3 kernel trace of bench.py -> the two small JSON files bench.py quotes:
    python scratch/pmc_traffic_json.py <gpurun_out dir with pmc_r6{fwd,dg,wg,fwdg,wgg}> <unused> <out dir> <date>"""
import csv, glob, json, os, re, sys
from collections import defaultdict
base, stats_txt, outdir, date = sys.argv[1:5]

def kb(dirn, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirn, f"*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in acc.items()}

res = {"date": date, "unit": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; gfx950: FETCH_SIZE counts 64 B per "
                             "128-B request of a wide coalesced read, MI355X_MICROARCH.md), standalone launches at N = 32: 48->48 @96x72 (fwd / dgrad / wgrad), the two-member group launches 48->48 @96x72 + 96->96 @48x36 (fwd_group / wgrad_group)",
       "source": "profiles/r06_pmc_conv3x3.txt (scratch/pmc_run.sh, one rocprofv3 --pmc pass per counter group)"}
for kind, tag in (("fwd", "r6fwd"), ("dgrad", "r6dg"), ("wgrad", "r6wg"), ("fwd_group", "r6fwdg"), ("wgrad_group", "r6wgg")):
    d = os.path.join(base, "pmc_" + tag)
    f, w = kb(d, "FETCH_SIZE"), kb(d, "WRITE_SIZE")
    tot, detail = 0.0, {}
    for k in sorted(set(f) | set(w)):
        if "conv3x3" in k or "wg3_reduce" in k:
            b = (2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024
            detail[k[-60:]] = {"fetch_kb": round(f.get(k, 0.0), 1), "write_kb": round(w.get(k, 0.0), 1), "bytes": round(b)}
            if "prep" not in k:
                tot += b
    res[kind] = {"bytes": round(tot), "kernels": detail}
json.dump(res, open(os.path.join(outdir, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: res[k]["bytes"] for k in ("fwd", "dgrad", "wgrad", "fwd_group", "wgrad_group")}))

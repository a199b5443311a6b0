"""Summarise the csv files of scratch/pmc_run.sh: per kernel name, mean of every counter over the launches (first
launch dropped) and the mean kernel duration.  usage: pmc_summary.py gpurun_out/pmc_<tag> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "conv3x3"
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if want in k:
            acc[k.split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(os.path.basename(f), k)
        for c, v in cs.items():
            v = v[1:] if len(v) > 1 else v
            print(f"    {c:36s} n={len(v):3d} mean {sum(v) / len(v):16.1f}")
for f in sorted(glob.glob(os.path.join(d, "*kernel_trace.csv")))[:1]:
    dur = defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if want in k:
            dur[k.split("(")[0][-70:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        v = v[1:] if len(v) > 1 else v
        print(f"duration {k}: n={len(v)} mean {sum(v) / len(v):.1f} us  min {min(v):.1f}")

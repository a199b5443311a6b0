"""Kernel resource table of one .hip file: python scratch/kres.py buctd_amd/csrc/conv3x3.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name: continue
    print(f"{name[:90]:90s} vgpr {v.get('VGPRs',0):3d} agpr {v.get('AGPRs',0):3d} spill {v.get('VGPRs Spill',0):3d} sgpr {v.get('SGPRs',0):3d} sspill {v.get('SGPRs Spill',0):3d} scratch {v.get('ScratchSize',0):4d} occ {v.get('Occupancy',0)}")

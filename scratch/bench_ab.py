"""bench.py with engine switches set from the command line (A/B on one box):
   python scratch/bench_ab.py <spec> [bench.py arguments]      spec: comma list of  off | on | parts=N | nofuse | nobranch"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from buctd_amd import ops
for tok in sys.argv[1].split(","):
    if tok == "off":
        ops.set_group_branches(False)
    elif tok.startswith("parts="):
        ops.set_group_parts(int(tok[6:]))
    elif tok == "nofuse":
        ops.conv_bn_group_ok = lambda xs, layers: False
    elif tok.startswith("p3=") or tok.startswith("p4=") or tok.startswith("p2="):       # e.g. p3=01-2  p4=012-3
        ops.set_group_partition(int(tok[1]), [[int(c) for c in part] for part in tok[3:].split("-")])
    elif tok == "nogwg":
        ops._GCONV_WGRAD["on"] = False
    elif tok == "nobranch":
        ops.group_branches_ok = lambda xs, chains: False
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

"""bench.py with the cross-branch group launches switched off / on (A/B on one box):
   python scratch/bench_ab.py off|on [bench.py arguments]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from buctd_amd import ops
ops.set_group_branches(sys.argv[1] != "off")
if sys.argv[1].isdigit():
    ops.set_group_parts(int(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

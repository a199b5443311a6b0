"""Run a script of this repository against another build of the library (experiments only; the product loader reads no
environment):   python scratch/run_alt.py <path-or-name-under-scratch/> <script.py> [args ...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib, script = sys.argv[1], sys.argv[2]
if not os.path.isabs(lib):
    lib = os.path.join(ROOT, "scratch", lib)
from buctd_amd import _C  # noqa: E402
_C.LIB_PATH = lib if os.path.exists(lib) else os.path.join(ROOT, sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")

"""scratch/time_c3_variants.py for several library builds: python scratch/time_c3_alt.py lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:]:
    if not os.path.isabs(lib): lib = os.path.join(ROOT, lib)
    print("==", os.path.basename(lib), flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "run_alt.py"), lib, os.path.join(ROOT, "scratch", "time_c3_variants.py")])

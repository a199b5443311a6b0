"""Run a script under a faulthandler watchdog: python scratch/watchdog_run.py SECONDS script.py [args...]"""
import faulthandler
import runpy
import sys

secs = int(sys.argv[1])
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

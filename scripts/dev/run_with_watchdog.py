"""Run a script with a watchdog: after N seconds dump every thread's Python stack and exit.
usage: python scripts/dev/run_with_watchdog.py SECONDS script.py [args...]"""
import faulthandler
import runpy
import sys

secs = float(sys.argv[1])
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

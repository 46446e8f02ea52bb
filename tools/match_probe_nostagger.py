"""tools/match_probe.py with the int8 kernel forced and its wave stagger switched off for the first variant (separate
script because a launcher under rocprofv3 must be the program itself)."""
import runpy
import sys
sys.argv = [sys.argv[0], "match_use_i8=1", "match_no_stagger=1"]
runpy.run_path(__file__.replace("match_probe_nostagger.py", "match_probe.py"), run_name="__main__")

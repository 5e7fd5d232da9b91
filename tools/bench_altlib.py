"""bench.py with another build of the library (ALTLIB=<file under tools/>): python tools/bench_altlib.py <bench.py arguments>"""
import os
import runpy
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
if os.environ.get("ALTLIB"):
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["ALTLIB"])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

"""bench.py against another build of the library (same-call A/B of whole steps): PROBE_LIB=/path/libfmmt_hip_x.so python tools/probes/bench_with_lib.py [bench.py flags]"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

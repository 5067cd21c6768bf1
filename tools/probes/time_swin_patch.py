"""tools/probes/time_swin.py with module constants of facialmmt_amd.ops patched first: PATCH="ops._MLP_FUSED_WIDTHS=(96,)" python tools/probes/time_swin_patch.py 640"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
import facialmmt_amd.ops as ops                      # noqa: E402
for item in filter(None, os.environ.get("PATCH", "").split(";")):
    name, value = item.split("=", 1)
    mod, attr = name.strip().split(".")
    assert mod == "ops" and hasattr(ops, attr), name
    setattr(ops, attr, eval(value))
    print(f"[patch] {name} = {getattr(ops, attr)!r}", file=sys.stderr)
sys.argv = [os.path.join(root, "tools", "probes", "time_swin.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

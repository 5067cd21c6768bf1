"""bench.py with module constants of facialmmt_amd.ops / train_step patched (same-call A/B of a host-side switch; the library reads no environment variable and the
modules keep constants, not env switches):  PATCH="ops._MLP_SAVE_DG=False,ops._WBLOCK_WIDTHS=(96,192)" python tools/probes/bench_patch.py [bench.py flags]
PROBE_LIB=... selects another build of the library as well."""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
import facialmmt_amd.ops as ops                      # noqa: E402
import facialmmt_amd.train_step as train_step        # noqa: E402
import facialmmt_amd.modules.SwinTransformer.Swin_Transformer as swin_mod        # noqa: E402
for item in filter(None, os.environ.get("PATCH", "").split(";")):
    name, value = item.split("=", 1)
    mod, *path, attr = name.strip().split(".")        # ops._X, or train_step.GraphedTargetStep.TEXT_FORK_AT
    target = {"ops": ops, "train_step": train_step, "swin": swin_mod}[mod]
    for part in path:
        target = getattr(target, part)
    assert hasattr(target, attr), name
    setattr(target, attr, eval(value))
    print(f"[bench_patch] {name} = {getattr(target, attr)!r}", file=sys.stderr)
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

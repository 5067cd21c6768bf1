"""The finish pass of the weight-gradient launches (reduce_partials_kernel) per Swin shape: whole launch minus the contraction alone.  PROBE_LIB selects a build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
dev = torch.device("cuda:0")
SHAPES = [(2007040, 288, 96), (2007040, 96, 96), (2007040, 384, 96), (2007040, 96, 384), (501760, 576, 192), (501760, 192, 192), (501760, 768, 192), (501760, 192, 768),
          (125440, 1152, 384), (125440, 384, 384), (125440, 1536, 384), (125440, 384, 1536), (31360, 2304, 768), (31360, 768, 768), (31360, 3072, 768), (31360, 768, 3072)]
def t(fn, n=10):
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
tot = 0.0
for (M, N, K) in SHAPES:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16); x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    nbytes = _lib.load().fmmt_linear_wgrad_workspace(ops.dtype_code(dy.dtype), M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for _ in range(3): ops.wgrad_raw(dy, x, True)
    part = t(lambda: ops.wgrad_partials_raw(dy, x, True, ws, nbytes))
    full = t(lambda: ops.wgrad_raw(dy, x, True))
    tot += full - part
    print(f"{M:8d} x {N:5d} x {K:5d}: contraction {part:7.1f} us, with the finish pass {full:7.1f} us ({full - part:6.1f}); partials {nbytes / 2**20:6.1f} MiB", flush=True)
print(f"sum of the finish passes {tot:.1f} us")

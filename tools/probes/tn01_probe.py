"""The stage-0 / stage-1 weight-gradient launches (token contraction over 0.5-2 M tokens, small N x K: HBM streams) alone: us and the
algorithmic GB/s of each, plain and with the DropPath scale.  PROBE_LIB=... for a same-call A/B of two builds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
dev = torch.device("cuda:0")
SHAPES = [(2007040, 288, 96, False), (2007040, 96, 96, True), (2007040, 384, 96, False), (2007040, 96, 384, True),
          (501760, 576, 192, False), (501760, 192, 192, True), (501760, 768, 192, False), (501760, 192, 768, True), (501760, 192, 384, False)]
tot = 0.0
for (M, N, K, scaled) in SHAPES:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16); x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    rps = 3136 if M > 1000000 else 784
    rs = torch.full((M // rps,), 1.0 / 0.9, device=dev) if scaled else None
    for _ in range(3):
        ops.wgrad_raw(dy, x, True, rs, rps)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.wgrad_raw(dy, x, True, rs, rps)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    tot += best
    gb = M * (N + K) * 2 / 1e6       # MB; / ms = GB/s
    print(f"  tn01 {M:8d}x{N:4d}x{K:4d}{' scaled' if scaled else '       '}: {best*1e3:7.1f} us  {gb/best:6.0f} GB/s  {2.0*M*N*K/best/1e9:6.1f} TF/s", flush=True)
    del dy, x
print(f"  total {tot*1e3:.1f} us")

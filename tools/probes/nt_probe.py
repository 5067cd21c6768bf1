"""Development aid: the many-token NT launches of Swin stages 1-3 alone, timed with HIP events (A/B of the FMMT_NT_* switches: one
process per setting, same gpurun call); --vendor adds hipBLASLt (torch.nn.functional.linear) on the same operands."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):                                # A/B of two builds in one gpurun call
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
from facialmmt_amd._lib import EPI_GELU, EPI_GELU_BWD
dev = torch.device("cuda:0")
vendor = "--vendor" in sys.argv
SHAPES = [(501760, 576, 192), (501760, 768, 192), (501760, 192, 768), (125440, 1152, 384), (125440, 1536, 384), (125440, 384, 384),
          (125440, 384, 1536), (125440, 384, 1152), (31360, 2304, 768), (31360, 3072, 768), (31360, 768, 3072), (31360, 768, 768)]
if os.environ.get("NT_PROBE_SHAPES"):                         # "M,N,K;M,N,K;..." instead of the Swin list
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["NT_PROBE_SHAPES"].split(";") if t]


def ev(fn, n=10, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


tot = 0.0
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
    b = torch.randn(N, device=dev)
    t = ev(lambda: ops.linear_raw(x, w, b)); tot += t
    line = f"  nt {M:7d}x{N:5d}x{K:5d}: {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:6.1f} TF/s"
    if N == 4 * K:
        pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t4 = ev(lambda: ops.linear_raw(x, w, b, epi=EPI_GELU, y_pre=pre)); tot += t4
        line += f" | gelu+pre {t4*1e3:7.1f} us"
    rps = 196 if M == 125440 else (49 if M % 49 == 0 else 64)
    rs = torch.full((M // rps,), 1.0 / 0.9, device=dev)
    if N == 4 * K:                                             # fc2 input gradient: GELU' operand
        aux = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        t5 = ev(lambda: ops.linear_raw(x, w, None, epi=EPI_GELU_BWD, aux=aux)); tot += t5
        line += f" | gelu' {t5*1e3:7.1f} us"
    if K == 4 * N or K == N:                                   # fc2 / proj forward: residual + DropPath scale; proj input gradient: scale only
        res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        t6 = ev(lambda: ops.linear_raw(x, w, b, res=res, rowscale=rs, rows_per_scale=rps)); tot += t6
        t7 = ev(lambda: ops.linear_raw(x, w, None, rowscale=rs, rows_per_scale=rps)); tot += t7
        line += f" | res+scale {t6*1e3:7.1f} us | scale {t7*1e3:7.1f} us"
    if vendor:
        tv = ev(lambda: torch.nn.functional.linear(x, w))
        line += f" || hipblaslt {tv*1e3:7.1f} us {2.0*M*N*K/tv/1e9:6.1f} TF/s"
    print(line, flush=True)
print(f"  total {tot*1e3:.1f} us   FMMT_NT_P256_BATCH={os.environ.get('FMMT_NT_P256_BATCH', '')} RING={os.environ.get('FMMT_NT_P256_RING', '')} OPS={os.environ.get('FMMT_NT_P256_OPS', '')}")

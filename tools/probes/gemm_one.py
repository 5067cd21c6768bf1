"""Development aid: run ONE Linear problem a few times (for rocprofv3 --pmc passes).
usage: gpu_gemm_one.py M N K mode   (mode: plain | gelu | gelubwd | res | tn | tnscale)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
from facialmmt_amd._lib import EPI_GELU, EPI_GELU_BWD
M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "plain"
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
b = torch.randn(N, device=dev)
pre = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
rs = torch.full((M // 49 + 1,), 1.25, device=dev)
for _ in range(5):
    if mode == "plain":
        ops.linear_raw(x, w, b)
    elif mode == "gelu":
        ops.linear_raw(x, w, b, epi=EPI_GELU, y_pre=pre)
    elif mode == "gelubwd":
        ops.linear_raw(x, w, None, epi=EPI_GELU_BWD, aux=pre)
    elif mode == "res":
        ops.linear_raw(x, w, b, res=pre, rowscale=rs, rows_per_scale=49)
    elif mode == "tn":
        ops.wgrad_raw(pre, x, True)
    elif mode == "tnscale":
        ops.wgrad_raw(pre, x, True, rs, 49)
torch.cuda.synchronize()

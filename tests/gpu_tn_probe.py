"""Development aid: the stage-2 weight-gradient launches alone (for PMC passes: rocprofv3 --pmc ... -- python tests/gpu_tn_probe.py)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facialmmt_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(125440, 1536, 384), (125440, 1152, 384), (125440, 384, 1536)]:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16); x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for _ in range(4):
        ops.wgrad_raw(dy, x, True)
    torch.cuda.synchronize()

"""Development aid: the stage-0 fused Mlp forward (training form) alone, a few launches (for tools/pmc_kernel.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
dev = torch.device("cuda:0")
M, C = 2007040, 96
x = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
w1 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).bfloat16(); b1 = torch.randn(4 * C, device=dev) * 0.1
w2 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).bfloat16(); b2 = torch.randn(C, device=dev) * 0.1
g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev)
for _ in range(5):
    out = ops.mlp_ln(x, g, b, 1e-5, w1, b1, w2, b2, None, 1)
    del out
torch.cuda.synchronize()

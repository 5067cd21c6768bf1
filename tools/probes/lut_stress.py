"""Development aid: the kernels that read the GELU table from LDS, many launches at the bench size, outputs compared bit for bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, C) in ((2007040, 96), (501760, 192)):
    x = torch.randn(M, C, device=dev).bfloat16()
    w1 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).bfloat16(); b1 = 0.1 * torch.randn(4 * C, device=dev)
    w2 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).bfloat16(); b2 = 0.1 * torch.randn(C, device=dev)
    res = torch.randn(M, C, device=dev).bfloat16()
    first, bad = None, 0
    for it in range(300):
        hp = torch.empty(M, 4 * C, device=dev, dtype=torch.bfloat16); ha = torch.empty_like(hp)
        y = ops.mlp_fused_raw(x, w1, b1, w2, b2, res, None, 1, hp, ha)
        if first is None:
            first = (y.clone(), ha.clone())
        elif not (torch.equal(first[0], y) and torch.equal(first[1], ha)):
            bad += 1
    print(f"fused Mlp forward {M}x{C}: {bad} of 299 launches differ", flush=True)
    dy = torch.randn(M, C, device=dev).bfloat16(); hpre = torch.randn(M, 4 * C, device=dev).bfloat16()
    if C == 96:
        first, bad = None, 0
        for it in range(300):
            dh, dx = ops.mlp_bwd_input_raw(dy, hpre, w1, w2, None, 1)
            if first is None:
                first = (dh.clone(), dx.clone())
            elif not (torch.equal(first[0], dh) and torch.equal(first[1], dx)):
                bad += 1
        print(f"fused Mlp input gradient {M}x{C}: {bad} of 299 launches differ", flush=True)

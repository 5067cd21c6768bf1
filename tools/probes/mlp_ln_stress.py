"""Development aid: the fused Mlp backward (fmmt_mlp_ln_bwd_input hand-counts its in-flight loads) repeated at the bench size, every
result compared bit for bit with the first; other work is queued on a second stream to vary the timing."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for M, C in ((2007040, 96), (8192 + 77, 96), (256 * 300 + 5, 96), (501760, 192), (128 * 300 + 9, 192)):
    x = torch.randn(M, C, device=dev).bfloat16().requires_grad_(True)
    g = (1 + 0.2 * torch.randn(C, device=dev)).requires_grad_(True); b = (0.1 * torch.randn(C, device=dev)).requires_grad_(True)
    w1 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).requires_grad_(True); b1 = (0.1 * torch.randn(4 * C, device=dev)).requires_grad_(True)
    w2 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).requires_grad_(True); b2 = (0.1 * torch.randn(C, device=dev)).requires_grad_(True)
    dy = torch.randn(M, C, device=dev).bfloat16()
    leaves = [x, g, b, w1, b1, w2, b2]
    side = torch.cuda.Stream()
    junk = torch.randn(8192, 8192, device=dev)
    first = None
    n = 40 if M > 400000 else 200
    for it in range(n):
        if it % 3 == 1:
            with torch.cuda.stream(side):
                for _ in range(it % 5):
                    junk = junk @ junk * 1e-4
        y = ops.mlp_ln(x, g, b, 1e-5, w1, b1, w2, b2, None, 1)
        gr = torch.autograd.grad(y, leaves, dy)
        if first is None:
            first = [t.clone() for t in gr]
        else:
            for k, (a, c) in enumerate(zip(first, gr)):
                if not torch.equal(a, c):
                    bad += 1
                    print(f"M={M} C={C} iteration {it}: gradient {k} differs, max |diff| {(a.float() - c.float()).abs().max().item():.3e}", flush=True)
    torch.cuda.synchronize()
    print(f"M={M} C={C}: {n} repetitions done, mismatches so far {bad}", flush=True)
sys.exit(1 if bad else 0)

"""Development measurement behind INTEGRATION.md's choice of front end: host cost per call of the two ways into the same C-ABI launch,
`ops.linear` (torch.autograd.Function -> ctypes) and `torch.ops.fmmt.linear` (torch.library custom operator -> the same ctypes call),
on a problem small enough for the host to be the bound (64 x 96 x 96), forward only and forward + backward."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
import facialmmt_amd.torch_ops  # noqa: F401
dev = torch.device("cuda:0")
x = torch.randn(64, 96, device=dev, dtype=torch.bfloat16, requires_grad=True)
w = torch.randn(96, 96, device=dev, requires_grad=True)
b = torch.randn(96, device=dev, requires_grad=True)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
f_fn = lambda: ops.linear(x, w, b)
f_op = lambda: torch.ops.fmmt.linear(x, w, b, None, None, 1)
def fb(f):
    def g():
        y = f()
        torch.autograd.grad(y, (x, w, b), y)
    return g
with torch.no_grad():
    a, c = t(f_fn), t(f_op)
print(f"forward only  (no_grad): autograd.Function {a:6.1f} us/call   torch.ops.fmmt {c:6.1f} us/call")
a, c = t(fb(f_fn), 1000), t(fb(f_op), 1000)
print(f"forward + backward     : autograd.Function {a:6.1f} us/call   torch.ops.fmmt {c:6.1f} us/call")

"""Development aid: PyTorch's TunableOp on the text encoder's GEMM shapes (M = 2048 tokens): default heuristic pick of the vendor library vs the tuned pick."""
import os, sys, time
import torch
dev = torch.device("cuda:0")


def ev(fn, n=20, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


M = 2048
shapes = [("qkv fwd", (M, 1024), (3072, 1024), "nt"), ("out fwd", (M, 1024), (1024, 1024), "nt"), ("fc1 fwd", (M, 1024), (4096, 1024), "nt"), ("fc2 fwd", (M, 4096), (1024, 4096), "nt"),
          ("qkv dgrad", (M, 3072), (3072, 1024), "nn"), ("out dgrad", (M, 1024), (1024, 1024), "nn"), ("fc1 dgrad", (M, 4096), (4096, 1024), "nn"), ("fc2 dgrad", (M, 1024), (1024, 4096), "nn"),
          ("qkv wgrad", (M, 3072), (M, 1024), "tn"), ("out wgrad", (M, 1024), (M, 1024), "tn"), ("fc1 wgrad", (M, 4096), (M, 1024), "tn"), ("fc2 wgrad", (M, 1024), (M, 4096), "tn")]
ops = []
for name, sa, sb, kind in shapes:
    a = torch.randn(*sa, device=dev, dtype=torch.bfloat16); b = torch.randn(*sb, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(sb[0], device=dev, dtype=torch.bfloat16)
    if kind == "nt":
        fn = (lambda a=a, b=b, bias=bias: torch.nn.functional.linear(a, b, bias))
    elif kind == "nn":
        fn = (lambda a=a, b=b: a.matmul(b))
    else:
        fn = (lambda a=a, b=b: a.t().mm(b))
    ops.append((name, fn))
base = [ev(fn) for _, fn in ops]
import torch.cuda.tunable as T
T.enable(True); T.tuning_enable(True)
try:
    T.set_max_tuning_duration(30); T.set_max_tuning_iterations(20)
except Exception as e:
    print("tuning limits:", e)
t0 = time.time()
for _, fn in ops:
    fn()
torch.cuda.synchronize()
print(f"tuning took {time.time() - t0:.1f} s")
T.tuning_enable(False)
tuned = [ev(fn) for _, fn in ops]
for (name, _), b, t in zip(ops, base, tuned):
    print(f"{name:10s} default {b:6.1f} us  tuned {t:6.1f} us")
print(f"sum default {sum(base):.1f} us tuned {sum(tuned):.1f} us  (x24 layers: {24*(sum(base)-sum(tuned))/1e3:.2f} ms)")

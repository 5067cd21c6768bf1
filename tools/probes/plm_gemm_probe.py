"""The text encoder's GEMM shapes (RoBERTa-large, 4 x 512 tokens): our NT / TN kernels against hipBLASLt through torch, us per launch inside a
HIP graph of 20 launches (what a replayed step pays)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
dev = torch.device("cuda:0")
dt = torch.bfloat16
def graph_time(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for (N, K) in [(1024, 1024), (4096, 1024), (1024, 4096)]:
    x = torch.randn(M, K, device=dev, dtype=dt); w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5; b = torch.randn(N, device=dev, dtype=dt)
    bf = b.float()
    dy = torch.randn(M, N, device=dev, dtype=dt)
    wt = w.t().contiguous()
    t_v = graph_time(lambda: torch.nn.functional.linear(x, w, b))
    t_o = graph_time(lambda: ops.linear_raw(x, w, bf))
    t_vd = graph_time(lambda: dy.matmul(w))
    t_od = graph_time(lambda: ops.linear_raw(dy, wt, None))
    t_vw = graph_time(lambda: dy.t().mm(x))
    t_ow = graph_time(lambda: ops.wgrad_raw(dy, x, True))
    err = (ops.linear_raw(x, w, bf).float() - torch.nn.functional.linear(x, w, b).float()).abs().max().item()
    print(f"M={M} N={N} K={K}: fwd vendor {t_v:6.1f} ours {t_o:6.1f} | dgrad vendor {t_vd:6.1f} ours {t_od:6.1f} | wgrad vendor {t_vw:6.1f} ours(+bias grad) {t_ow:6.1f} us   max|diff| {err:.3f}", flush=True)

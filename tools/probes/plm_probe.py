"""Development aid: the text encoder's Linear shapes (2048 tokens; RoBERTa-large / BERT-large) on our NT / TN kernels against the
vendor library, per launch from a captured graph of 20 launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
from facialmmt_amd._lib import EPI_GELU
dev = torch.device("cuda:0")


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    keep.append(g)
    return (time.perf_counter() - t0) / 5 / reps


keep = []
for M in (2048,):
    for (N, K) in [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)]:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
        b = torch.randn(N, device=dev); bb = b.bfloat16()
        dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        wt = w.t().contiguous()
        t_nt = graph_time(lambda: ops.linear_raw(x, w, b))
        t_v = graph_time(lambda: torch.nn.functional.linear(x, w, bb))
        t_dg = graph_time(lambda: ops.linear_raw(dy, wt, None))
        t_dgv = graph_time(lambda: dy.matmul(w))
        t_tn = graph_time(lambda: ops.wgrad_raw(dy, x, True))
        t_tnv = graph_time(lambda: (dy.t().mm(x), ops.colsum_raw(dy)))
        t_cs = graph_time(lambda: ops.colsum_raw(dy))
        t_sum = graph_time(lambda: dy.sum(0))
        print(f"  {M}x{N}x{K}: fwd ours {t_nt*1e6:6.1f} us vendor {t_v*1e6:6.1f} | dgrad ours {t_dg*1e6:6.1f} vendor {t_dgv*1e6:6.1f} | wgrad+db ours {t_tn*1e6:6.1f} vendor+colsum {t_tnv*1e6:6.1f}"
              f" | colsum {t_cs*1e6:5.1f} torch.sum {t_sum*1e6:5.1f}", flush=True)

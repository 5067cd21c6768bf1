"""Development aid: the few-token NT launches of the fusion stack (152-1328 tokens x 768 / 3072 channels) alone, GPU time per launch from
a captured graph of 20 launches, with a value check against torch (A/B of FMMT_NT_SMALL: one process per setting, same gpurun call)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
from facialmmt_amd._lib import EPI_GELU, EPI_GELU_BWD
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
    return (time.perf_counter() - t0) / 5 / reps


tot, bad = 0.0, 0
for (M, N, K) in [(152, 768, 768), (166, 768, 768), (512, 768, 768), (640, 768, 768), (1328, 768, 768), (664, 1536, 768), (512, 2304, 768), (512, 3072, 768), (512, 768, 3072),
                  (1328, 3072, 768), (1328, 768, 3072), (640, 512, 768), (600, 768, 768)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5; b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    ref = x.float() @ w.float().t() + b
    y = ops.linear_raw(x, w, b)
    y2 = ops.linear_raw(x, w, b, res=res)
    y3 = ops.linear_raw(x, w, b, epi=EPI_GELU)
    e = max((y.float() - ref).abs().max().item(), (y2.float() - ref - res.float()).abs().max().item(),
            (y3.float() - torch.nn.functional.gelu(ref)).abs().max().item())
    ok = e < 0.06
    bad += 0 if ok else 1
    t = graph_time(lambda: ops.linear_raw(x, w, b)); tot += t
    t2 = graph_time(lambda: ops.linear_raw(x, w, b, res=res)); tot += t2
    print(f"  {M:6d}x{N:5d}x{K:5d}: nt {t*1e6:6.1f} us | +res {t2*1e6:6.1f} us | max err {e:.4f} {'ok' if ok else 'BAD'}", flush=True)
print(f"  total {tot*1e6:.1f} us  bad {bad}  FMMT_NT_SMALL={os.environ.get('FMMT_NT_SMALL', '')}")
sys.exit(1 if bad else 0)

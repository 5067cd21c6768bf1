"""Development micro-benchmark of the Linear kernels at the bench shapes (bf16)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):                                # A/B of two builds in one gpurun call
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
SHAPES_ALL = [(2007040, 288, 96), (2007040, 96, 96), (2007040, 384, 96), (2007040, 96, 384), (2007040, 96, 288),
          (501760, 576, 192), (501760, 768, 192), (501760, 192, 768), (125440, 1152, 384), (125440, 1536, 384), (125440, 384, 1536), (125440, 384, 384), (125440, 384, 1152),
          (31360, 2304, 768), (31360, 3072, 768), (31360, 768, 3072), (664, 768, 768), (1280, 3072, 768)]
SHAPES = [sh for sh in SHAPES_ALL if not os.environ.get("GEMM_BENCH_M") or sh[0] == int(os.environ["GEMM_BENCH_M"])]
if os.environ.get("GEMM_BENCH_FEW"):
    SHAPES = []
print("cfg", os.environ.get("FMMT_NT_CFG", "0"))
from facialmmt_amd._lib import EPI_GELU, EPI_GELU_BWD
tot = 0
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, device=dev)
    t = timeit(lambda: ops.linear_raw(x, w, b)); tot += t
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    rs = torch.full((M // 49,), 1.25, device=dev)
    t2 = timeit(lambda: ops.wgrad_raw(dy, x, True))
    t3 = timeit(lambda: ops.wgrad_raw(dy, x, True, rs, 49))
    line = f"  {M:8d}x{N:5d}x{K:5d}: nt {t*1e3:7.3f} ms {2.0*M*N*K/t/1e12:6.1f} TF/s {(M*K+M*N+N*K)*2/t/1e9:6.0f} GB/s | tn {t2*1e3:7.3f} ms {2.0*M*N*K/t2/1e12:6.1f} TF/s | tn+rowscale {t3*1e3:7.3f} ms"
    if N == 4 * K:      # fc1 forward (GELU, pre-activation saved) and fc2 input gradient (GELU')
        pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t4 = timeit(lambda: ops.linear_raw(x, w, b, epi=EPI_GELU, y_pre=pre))
        t5 = timeit(lambda: ops.linear_raw(x, w, None, epi=EPI_GELU_BWD, aux=pre))
        line += f" | gelu+pre {t4*1e3:7.3f} ms | gelu' {t5*1e3:7.3f} ms"
    if K == 4 * N or K == N:      # fc2 / proj forward: residual + DropPath scale
        res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        t6 = timeit(lambda: ops.linear_raw(x, w, b, res=res, rowscale=rs, rows_per_scale=49))
        line += f" | res+scale {t6*1e3:7.3f} ms"
    if os.environ.get("GEMM_BENCH_VENDOR"):        # vendor-library reference points (hipBLASLt via torch), same operands
        tv = timeit(lambda: torch.nn.functional.linear(x, w))
        tv2 = timeit(lambda: torch.matmul(dy.t(), x))
        line += f" || hipblaslt nt {tv*1e3:7.3f} ms {2.0*M*N*K/tv/1e12:6.1f} TF/s  tn {tv2*1e3:7.3f} ms {2.0*M*N*K/tv2/1e12:6.1f} TF/s"
    print(line, flush=True)
    del x, w, dy
print(f"  sum nt {tot*1e3:.3f} ms")

# few-token problems (cross-modal / self-attention encoders): GPU time per launch from a captured graph of 20 launches
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
if not os.environ.get("GEMM_BENCH_M"):
    print("few-token (graph-replayed, per launch):")
    for (M, N, K) in [(152, 768, 768), (512, 768, 768), (640, 768, 768), (1328, 768, 768), (664, 1536, 768), (512, 3072, 768), (512, 768, 3072), (1328, 3072, 768), (1328, 768, 3072)]:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        t = graph_time(lambda: ops.linear_raw(x, w, b))
        t2 = graph_time(lambda: ops.wgrad_raw(dy, x, True))
        tv = graph_time(lambda: torch.nn.functional.linear(x, w, b.bfloat16()))
        print(f"  {M:6d}x{N:5d}x{K:5d}: nt {t*1e6:6.1f} us {2.0*M*N*K/t/1e12:6.1f} TF/s | tn(+reduce) {t2*1e6:6.1f} us | hipblaslt nt {tv*1e6:6.1f} us", flush=True)

"""Development micro-benchmark: fused Mlp forward (fmmt_mlp_fwd) against the two-launch form, and the weight gradient with the
activation recomputed on load, at the Swin stage-0 / stage-1 sizes of the bench (640 frames)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):                                # A/B of two builds in one gpurun call
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
from facialmmt_amd._lib import EPI_GELU
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, C) in [(2007040, 96), (501760, 192)]:
    dt = torch.bfloat16
    x = torch.randn(M, C, device=dev, dtype=dt); res = torch.randn(M, C, device=dev, dtype=dt)
    w1 = torch.randn(4 * C, C, device=dev, dtype=dt) * C ** -0.5; b1 = torch.randn(4 * C, device=dev)
    w2 = torch.randn(C, 4 * C, device=dev, dtype=dt) * (4 * C) ** -0.5; b2 = torch.randn(C, device=dev)
    rs = torch.full((M // 49,), 1.25, device=dev)
    hp = torch.empty(M, 4 * C, device=dev, dtype=dt)
    tf = timeit(lambda: ops.mlp_fused_raw(x, w1, b1, w2, b2, res, rs, 49, hp))
    tfn = timeit(lambda: ops.mlp_fused_raw(x, w1, b1, w2, b2, res, rs, 49, None))
    ha = torch.empty(M, 4 * C, device=dev, dtype=dt)
    tfa = timeit(lambda: ops.mlp_fused_raw(x, w1, b1, w2, b2, res, rs, 49, hp, ha))
    del ha
    t1 = timeit(lambda: ops.linear_raw(x, w1, b1, epi=EPI_GELU, y_pre=hp))
    h = ops.linear_raw(x, w1, b1, epi=EPI_GELU, y_pre=hp)
    t2 = timeit(lambda: ops.linear_raw(h, w2, b2, res=res, rowscale=rs, rows_per_scale=49))
    fl = 2.0 * M * C * 4 * C * 2
    by = M * C * 2 * 3 + M * 4 * C * 2
    print(f"{M}x{C}: fused {tf*1e3:.3f} ms ({fl/tf/1e12:.0f} TF/s, {by/tf/1e9:.0f} GB/s algorithmic) | fused, no pre-activation {tfn*1e3:.3f} ms | fused + activation stored {tfa*1e3:.3f} ms | "
          f"two launches {t1*1e3:.3f} + {t2*1e3:.3f} = {(t1+t2)*1e3:.3f} ms", flush=True)
    dy = torch.randn(M, C, device=dev, dtype=dt)
    tw = timeit(lambda: ops.wgrad_raw(dy, h, True, rs, 49))
    twg = timeit(lambda: ops.wgrad_raw(dy, hp, True, rs, 49, x_gelu=True))
    print(f"   weight gradient of fc2: stored activation {tw*1e3:.3f} ms | recomputed gelu(pre) {twg*1e3:.3f} ms", flush=True)
    # input gradient: fused launch (C = 96; C = 192 goes to the two GELU' / plain GEMM launches unless ops._MLP_BWD_WIDTHS = (96, 192))
    tb = timeit(lambda: ops.mlp_bwd_input_raw(dy, hp, w1, w2, rs, 49))
    print(f"   input gradient (dh stored): {tb*1e3:.3f} ms", flush=True)
    del x, res, hp, h, dy
# GELU / GELU' epilogues of the stage-2 / stage-3 GEMMs against the plain launch of the same shape
from facialmmt_amd._lib import EPI_GELU_BWD
for (M, N, K) in [(125440, 1536, 384), (31360, 3072, 768)]:
    dt = torch.bfloat16
    x = torch.randn(M, K, device=dev, dtype=dt); w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5; b = torch.randn(N, device=dev)
    pre = torch.empty(M, N, device=dev, dtype=dt); aux = torch.randn(M, N, device=dev, dtype=dt)
    t0 = timeit(lambda: ops.linear_raw(x, w, b))
    t1 = timeit(lambda: ops.linear_raw(x, w, b, epi=EPI_GELU, y_pre=pre))
    t2 = timeit(lambda: ops.linear_raw(x, w, None, epi=EPI_GELU_BWD, aux=aux))
    print(f"{M}x{N}x{K}: plain {t0*1e3:.3f} ms | gelu + pre-activation {t1*1e3:.3f} ms | gelu' {t2*1e3:.3f} ms", flush=True)
    del x, w, pre, aux

import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"): _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
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
    hp = torch.empty(M, 4 * C, device=dev, dtype=dt); ha = torch.empty(M, 4 * C, device=dev, dtype=dt)
    tfn = timeit(lambda: ops.mlp_fused_raw(x, w1, b1, w2, b2, res, rs, 49, None))
    tf = timeit(lambda: ops.mlp_fused_raw(x, w1, b1, w2, b2, res, rs, 49, hp))
    tfa = timeit(lambda: ops.mlp_fused_raw(x, w1, b1, w2, b2, res, rs, 49, hp, ha))
    print(f"{os.environ.get('PROBE_LIB','new').split('_')[-1]:16s} {M}x{C}: inference {tfn*1e3:.3f} | +pre {tf*1e3:.3f} | +pre+act {tfa*1e3:.3f} ms", flush=True)
    del x, res, hp, ha
# the LayerNorm form the models call (fmmt_mlp_ln_fwd): inference (no_grad) and training (saves LN(x), statistics, both hidden tensors)
for (M, C) in [(2007040, 96), (501760, 192)]:
    dt = torch.bfloat16
    x = torch.randn(M, C, device=dev, dtype=dt)
    g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    w1 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).requires_grad_(True); b1 = torch.randn(4 * C, device=dev)
    w2 = torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5; b2 = torch.randn(C, device=dev)
    rs = torch.full((M // 49,), 1.25, device=dev)
    def fwd():
        return ops.mlp_ln(x, g, b, 1e-5, w1, b1, w2, b2, rs, 49)
    def fwd_ng():
        with torch.no_grad():
            return ops.mlp_ln(x, g, b, 1e-5, w1, b1, w2, b2, rs, 49)
    ti, tt = timeit(fwd_ng), timeit(fwd)
    print(f"{os.environ.get('PROBE_LIB','new').split('_')[-1]:16s} {M}x{C} mlp_ln: inference {ti*1e3:.3f} | training {tt*1e3:.3f} ms", flush=True)
    del x

"""Round 6 probe: what storing gelu'(pre-activation) instead of the pre-activation costs the forward and saves the backward.
Fused Mlp (stage 0 / 1 sizes): fmmt_mlp_ln_fwd / fmmt_mlp_ln_bwd_input with and without FMMT_SAVE_DG; two-launch Mlp (stage 2 / 3 shapes):
FMMT_EPI_GELU + pre-activation vs FMMT_EPI_GELU_DG, FMMT_EPI_GELU_BWD vs FMMT_EPI_MUL_AUX.  PROBE_LIB=... for another build."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
from facialmmt_amd._lib import EPI_GELU, EPI_GELU_BWD, EPI_GELU_DG, EPI_MUL_AUX
dev = torch.device("cuda:0")
dt = torch.bfloat16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for (M, C) in [(2007040, 96), (501760, 192)]:
    x = torch.randn(M, C, device=dev, dtype=dt)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1 = torch.randn(4 * C, C, device=dev, dtype=dt) * C ** -0.5; b1 = torch.randn(4 * C, device=dev) * 0.1
    w2 = torch.randn(C, 4 * C, device=dev, dtype=dt) * (4 * C) ** -0.5; b2 = torch.randn(C, device=dev) * 0.1
    rs = torch.full((M // 49,), 1.25, device=dev)
    dy = torch.randn(M, C, device=dev, dtype=dt)
    for dg in (False, True):
        ops._MLP_SAVE_DG = dg
        ops._MLP_FUSED_DG_WIDTHS = (96, 192)
        xr = x.clone().requires_grad_(True)
        w1r, w2r = w1.float().requires_grad_(True), w2.float().requires_grad_(True)
        def fwd():
            return ops.mlp_ln(xr, g, b, 1e-5, w1r, b1, w2r, b2, rs, 49)
        tf = timeit(fwd)
        y = fwd()
        fn = y.grad_fn
        def bwd():
            return fn.apply(dy) if hasattr(fn, "apply") else None
        # the backward through autograd (input gradient + LayerNorm' + the two weight gradients): time the whole, and the input-gradient launch alone
        def whole():
            yy = fwd(); yy.backward(dy)
        tw = timeit(whole, 6)
        ctx_x2, xn, mean, rstd, gg, _, _, h_pre, h, _ = fn.saved_tensors
        lib = _lib.load()
        dh = torch.empty((M, 4 * C), dtype=dt, device=dev); dx = torch.empty_like(x)
        dgm = torch.empty(C, device=dev); dbt = torch.empty(C, device=dev)
        nbytes = lib.fmmt_mlp_ln_bwd_input_workspace(C); ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
        P = lambda t: t.data_ptr()
        def bin_():
            rc = lib.fmmt_mlp_ln_bwd_input(ops.dtype_code(dt) | (_lib.SAVE_DG if dg else 0), M, C, P(dy), P(h_pre), P(w2t), P(w1t), P(rs), 49, P(x), P(mean), P(rstd), P(gg),
                                           P(dh), P(dx), P(dgm), P(dbt), P(ws), nbytes, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        tb = timeit(bin_)
        print(f"fused Mlp {M}x{C} dg={int(dg)}: fwd (train) {tf:.3f} ms | bwd input+LN' {tb:.3f} ms | fwd+bwd through autograd {tw:.3f} ms", flush=True)
        del xr, y, fn, h_pre, h, dh, dx
    del x, dy

for (M, N, K) in [(125440, 1536, 384), (31360, 3072, 768)]:
    x = torch.randn(M, K, device=dev, dtype=dt); w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5; b = torch.randn(N, device=dev) * 0.1
    pre = torch.empty(M, N, device=dev, dtype=dt)
    aux = torch.randn(M, N, device=dev, dtype=dt)
    rs = torch.full((M // 49 + 1,), 1.1, device=dev)
    t0 = timeit(lambda: ops.linear_raw(x, w, b))
    t1 = timeit(lambda: ops.linear_raw(x, w, b, epi=EPI_GELU, y_pre=pre))
    t2 = timeit(lambda: ops.linear_raw(x, w, b, epi=EPI_GELU_DG, y_pre=pre))
    t3 = timeit(lambda: ops.linear_raw(x, w, None, epi=EPI_GELU_BWD, aux=aux, rowscale=rs, rows_per_scale=49))
    t4 = timeit(lambda: ops.linear_raw(x, w, None, epi=EPI_MUL_AUX, aux=aux, rowscale=rs, rows_per_scale=49))
    t5 = timeit(lambda: ops.linear_raw(x, w, None, epi=EPI_MUL_AUX, aux=aux))
    print(f"nt {M}x{N}x{K}: plain {t0*1e3:.1f} us | gelu+pre {t1*1e3:.1f} | gelu+dg {t2*1e3:.1f} | gelu' (+scale) {t3*1e3:.1f} | mul_aux+scale {t4*1e3:.1f} | mul_aux {t5*1e3:.1f}", flush=True)

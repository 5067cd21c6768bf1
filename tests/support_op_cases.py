"""Development probe (not a pytest file): runs every C-ABI entry point against a plain torch
reference on the GPU and prints max errors without stopping at the first failure.
Usage on the GPU box:  python tests/support_op_cases.py > gpurun_out/probe.log 2>&1"""
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facialmmt_amd import ops, synth  # noqa: E402
from facialmmt_amd._lib import EPI_GELU, EPI_GELU_BWD, EPI_GELU_DG, EPI_MUL_AUX  # noqa: E402
from oracle import crossmodal as OC  # noqa: E402
from oracle import swin as OS  # noqa: E402

dev = torch.device("cuda:0")
RES = []


def rnd(name, shape, seed=0, scale=1.0, dtype=torch.float32):
    return (synth.tensor(name, shape, seed=seed) * scale).to(dev).to(dtype)


def one_ulp(a, b):
    """bf16 tensors equal to within one rounding step at the largest magnitude"""
    return bool((a.float() - b.float()).abs().max() <= b.float().abs().max() * 2.0 ** -7)


def report(name, got, ref, tol):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    ok = err <= tol * max(scale, 1e-6) and torch.isfinite(got).all().item()
    RES.append((name, ok))
    print(f"{'OK  ' if ok else 'FAIL'} {name:58s} max|err|={err:.3e} ref_scale={scale:.3e} tol={tol:g}", flush=True)


def section(fn):
    try:
        fn()
    except Exception:
        RES.append((fn.__name__, False))
        print(f"EXC  {fn.__name__}")
        traceback.print_exc()
    torch.cuda.synchronize()


def t_linear():
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        for (M, N, K) in [(200, 96, 96), (333, 288, 96), (128, 128, 64), (1000, 384, 96), (77, 96, 384), (50, 512, 37632 // 8), (130, 768, 3072),
                          (1, 96, 96), (7, 768, 768), (1, 512, 4704), (3, 8, 8)]:      # degenerate token counts / tiny channels
            x = rnd("x", (M, K), 1, dtype=dt)
            w = rnd("w", (N, K), 2, K ** -0.5, dtype=dt)
            b = rnd("b", (N,), 3, 0.1)
            ref = x.double() @ w.double().t() + b.double()
            report(f"linear {dt} {M}x{N}x{K}", ops.linear_raw(x, w, b), ref, tol)
        for (M, N, K) in ((300, 384, 96), (300, 768, 768), (166, 768, 3072)):      # the latter two: quarter tiles of the few-token dispatch (bf16)
            x = rnd("x", (M, K), 1, dtype=dt)
            w = rnd("w", (N, K), 2, K ** -0.5, dtype=dt)
            b = rnd("b", (N,), 3, 0.1)
            pre = x.double() @ w.double().t() + b.double()
            ypre = torch.empty((M, N), dtype=dt, device=dev)
            y = ops.linear_raw(x, w, b, epi=EPI_GELU, y_pre=ypre)
            report(f"linear gelu {dt} {M}x{N}x{K}", y, OS.gelu_erf(pre), tol)
            report(f"linear gelu pre {dt} {M}x{N}x{K}", ypre, pre, tol)
            res = rnd("res", (M, N), 4, dtype=dt)
            rs = rnd("rs", (3,), 5).abs() + 0.5
            y = ops.linear_raw(x, w, b, res=res, rowscale=rs, rows_per_scale=100)
            report(f"linear res+rowscale {dt} {M}x{N}x{K}", y, res.double() + rs.double().repeat_interleave(100)[:M, None] * pre, tol)
            aux = rnd("aux", (M, N), 6, dtype=dt)
            a64 = aux.double().requires_grad_(True)
            g = torch.autograd.grad(OS.gelu_erf(a64).sum(), a64)[0]
            y = ops.linear_raw(x, w, None, epi=EPI_GELU_BWD, aux=aux)
            report(f"linear gelu_bwd {dt} {M}x{N}x{K}", y, (x.double() @ w.double().t()) * g, tol)
            # round 6: the forward stores the derivative (FMMT_EPI_GELU_DG), the backward multiplies by it (FMMT_EPI_MUL_AUX)
            ydg = torch.empty((M, N), dtype=dt, device=dev)
            y = ops.linear_raw(x, w, b, epi=EPI_GELU_DG, y_pre=ydg)
            p64 = pre.clone().requires_grad_(True)
            g64 = torch.autograd.grad(OS.gelu_erf(p64).sum(), p64)[0]
            report(f"linear gelu_dg {dt} {M}x{N}x{K}", y, OS.gelu_erf(pre), tol)
            report(f"linear gelu_dg derivative {dt} {M}x{N}x{K}", ydg, g64, tol)
            y = ops.linear_raw(x, w, None, epi=EPI_MUL_AUX, aux=aux, rowscale=rs, rows_per_scale=100)
            report(f"linear mul_aux {dt} {M}x{N}x{K}", y, rs.double().repeat_interleave(100)[:M, None] * (x.double() @ w.double().t()) * aux.double(), tol)


def t_linear_large():
    """many-token bf16 problems (>= 65536 rows: the 256-row deep-pipelined kernels; FMMT_NT_DEEP selects the
    variant) with ragged M, every epilogue, against fp32 torch matmuls of the same bf16 operands"""
    dt, tol = torch.bfloat16, 2e-2
    for (M, N, K) in [(65536 + 77, 384, 384), (70000, 1536, 384), (65536, 384, 1536), (66000, 128, 256), (65600, 768, 192),
                      (65536 + 77, 192, 192), (70000, 576, 192), (65600, 96, 384), (65536, 96, 288), (66000, 288, 768),
                      (65536 + 77, 384, 96), (70000, 288, 96), (65600, 96, 96),
                      (31360, 768, 3072), (31360 + 16, 2304, 768), (20000, 1536, 384), (125440, 1152, 384), (31360, 384, 1536),   # persistent 256-row-tile kernel: partial rounds, ragged panel
                      (7840, 1536, 1536), (7840 + 8, 6144, 1536), (7840, 4608, 1536), (7848, 1536, 6144), (4104, 2304, 768),   # Swin stage 3 (and a 153-tile case): four-phase kernel (gemm_ph.h), plain / bias and GELU + pre-activation, ragged last panel
                      (640, 37632, 512), (300, 16384, 64)]:   # few rows, very wide output (input gradient of the embedding head): 128-row tiles
        x = rnd("x", (M, K), 1, dtype=dt)
        w = rnd("w", (N, K), 2, K ** -0.5, dtype=dt)
        b = rnd("b", (N,), 3, 0.1)
        pre = x.float() @ w.float().t() + b
        report(f"linear large {M}x{N}x{K}", ops.linear_raw(x, w, b), pre, tol)
        ypre = torch.empty((M, N), dtype=dt, device=dev)
        y = ops.linear_raw(x, w, b, epi=EPI_GELU, y_pre=ypre)
        report(f"linear large gelu {M}x{N}x{K}", y, torch.nn.functional.gelu(pre), tol)
        report(f"linear large gelu pre {M}x{N}x{K}", ypre, pre, tol)
        res = rnd("res", (M, N), 4, dtype=dt)
        rs = rnd("rs", (M // 196 + 1,), 5).abs() + 0.5
        y = ops.linear_raw(x, w, b, res=res, rowscale=rs, rows_per_scale=196)
        report(f"linear large res+rowscale {M}x{N}x{K}", y, res.float() + rs.repeat_interleave(196)[:M, None] * pre, tol)
        aux = rnd("aux", (M, N), 6, dtype=dt)
        a32 = aux.float().requires_grad_(True)
        g = torch.autograd.grad(torch.nn.functional.gelu(a32).sum(), a32)[0]
        y = ops.linear_raw(x, w, None, epi=EPI_GELU_BWD, aux=aux)
        report(f"linear large gelu_bwd {M}x{N}x{K}", y, (x.float() @ w.float().t()) * g, tol)
        # round 6: derivative stored by the forward / product in the backward (the latter on gemm_ph3.h's EPI 6 where the shape allows)
        ydg = torch.empty((M, N), dtype=dt, device=dev)
        y = ops.linear_raw(x, w, b, epi=EPI_GELU_DG, y_pre=ydg)
        p32 = pre.clone().requires_grad_(True)
        g32 = torch.autograd.grad(torch.nn.functional.gelu(p32).sum(), p32)[0]
        report(f"linear large gelu_dg {M}x{N}x{K}", y, torch.nn.functional.gelu(pre), tol)
        report(f"linear large gelu_dg derivative {M}x{N}x{K}", ydg, g32, tol)
        y = ops.linear_raw(x, w, None, epi=EPI_MUL_AUX, aux=aux, rowscale=rs, rows_per_scale=196)
        report(f"linear large mul_aux+rowscale {M}x{N}x{K}", y, rs.repeat_interleave(196)[:M, None] * (x.float() @ w.float().t()) * aux.float(), tol)
        y = ops.linear_raw(x, w, None, epi=EPI_MUL_AUX, aux=aux)
        report(f"linear large mul_aux {M}x{N}x{K}", y, (x.float() @ w.float().t()) * aux.float(), tol)
        del ydg, p32, g32
        del x, w, pre, y, ypre, res, aux, a32, g


def t_wgrad():
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        for (M, N, K) in [(500, 96, 96), (3136, 288, 96), (777, 384, 96), (1000, 96, 384), (100, 512, 1024), (6272, 96, 48),
                          (1, 96, 96), (7, 768, 768), (769, 768, 768), (5000, 96, 384),      # single token; both sides of the 768-row direct/split boundary
                          (640, 768, 768), (664, 1536, 768), (2048, 768, 1024), (1328, 3072, 768), (129, 512, 512), (2047, 512, 576), (1000, 448, 512)]:   # linear_tn_few_kernel (bf16): wave-split tokens, ragged tails
            dy = rnd("dy", (M, N), 1, dtype=dt)
            x = rnd("x", (M, K), 2, dtype=dt)
            dw, db = ops.wgrad_raw(dy, x, True)
            report(f"wgrad dw {dt} {M}x{N}x{K}", dw, dy.double().t() @ x.double(), tol)
            report(f"wgrad db {dt} {M}x{N}x{K}", db, dy.double().sum(0), tol)
        M, N, K = 600, 192, 96
        dy = rnd("dy", (M, N), 1, dtype=dt)
        x = rnd("x", (M, K), 2, dtype=dt)
        rs = rnd("rs", (6,), 5).abs() + 0.5
        dw, db = ops.wgrad_raw(dy, x, True, rs, 100)
        s = rs.double().repeat_interleave(100)[:, None]
        report(f"wgrad rowscale dw {dt}", dw, (dy.double() * s).t() @ x.double(), tol)
        report(f"wgrad rowscale db {dt}", db, (dy.double() * s).sum(0), tol)


def t_wgrad_large():
    """many-token bf16 weight gradients (ragged token counts, bias gradient, DropPath row scale with a dropped sample) against fp32
    torch matmuls of the same bf16 operands (both sides accumulate exact products in fp32).  The unscaled launches of the
    stage-2/3 shapes take the DMA-staged kernel (256x256 / 192x384 tiles, bias gradient through the ones-fragment MFMA), the
    scaled ones and the other shapes the register-staged kernel; tests/test_gpu_ops.py runs the list once more with
    FMMT_TN_DMA=0 (everything register-staged)."""
    dt, tol = torch.bfloat16, 1e-3
    for (M, N, K) in [(125440, 1536, 384), (125440, 384, 1536), (125440, 1152, 384), (125440, 384, 384), (31360, 768, 768),
                      (20008, 1536, 384), (17000, 384, 1536), (31360, 2304, 768), (501760, 192, 384), (62720, 768, 384),
                      (31360, 768, 3072), (125440, 384, 768), (31360, 768, 1536), (20480, 1024, 256),
                      (15680, 2304, 768), (15680, 768, 3072), (12544, 3072, 768), (8256, 768, 768), (15680, 768, 768),      # 320-frame utterances (configs[4]): stage 3
                      (125440, 768, 192), (62720, 1152, 192), (501760, 768, 192), (501760, 576, 192), (125440, 960, 192)]:      # round 6: the 384 x 192 tile of the DMA-staged kernel (K = 192: stage 1's fc1 weight gradient), three bias blocks per wave; N = 576 / 960: 192 short of a whole tile (padded partial rows, dropped by the finish pass)
        dy = rnd("dy", (M, N), 1, dtype=dt)
        x = rnd("x", (M, K), 2, dtype=dt)
        dw, db = ops.wgrad_raw(dy, x, True)
        report(f"wgrad large dw {M}x{N}x{K}", dw, dy.float().t() @ x.float(), tol)
        report(f"wgrad large db {M}x{N}x{K}", db, dy.float().sum(0), tol)
        dw2, none = ops.wgrad_raw(dy, x, False)
        RES.append((f"wgrad large no-bias identical {M}x{N}x{K}", bool(torch.equal(dw, dw2)) and none is None))
        rs = rnd("rs", (M // 196 + 1,), 5).abs() + 0.5
        rs[1] = 0.0                                            # a dropped sample
        dw, db = ops.wgrad_raw(dy, x, True, rs, 196)
        sdy = dy.float() * rs.repeat_interleave(196)[:M, None]
        report(f"wgrad large rowscale dw {M}x{N}x{K}", dw, sdy.t() @ x.float(), 5e-3)      # s * dy is rounded to bf16 before the MFMA
        report(f"wgrad large rowscale db {M}x{N}x{K}", db, sdy.sum(0), 5e-3)
        # what DropPath hands over: 0 for a dropped image, 1 / keep for the others (two-valued: the DMA-staged kernel zeroes the dropped images'
        # rows and applies the common factor to its accumulators -- nothing is rounded, so the fp32 product of the bf16 operands is exact to 1e-3);
        # images of 196 and of 49 tokens (stage 2 / 3), first and last image dropped, and the all-dropped vector
        for rps in (196, 49):
            nimg = (M + rps - 1) // rps
            rs2 = torch.full((nimg,), 1.0 / 0.9, device=dy.device)
            rs2[torch.arange(0, nimg, 7, device=dy.device)] = 0.0
            rs2[-1] = 0.0
            dw, db = ops.wgrad_raw(dy, x, True, rs2, rps)
            s2 = rs2.repeat_interleave(rps)[:M, None]
            tol2 = tol if M in (125440, 31360, 15680) and (N, K) != (768, 768) and K != 192 else 5e-3      # the register-staged kernel rounds s * dy to bf16 (K = 192: the <384,192> tile is unscaled-only)
            report(f"wgrad large droppath dw {M}x{N}x{K} rps={rps}", dw, (dy.float() * s2).t() @ x.float(), tol2)
            report(f"wgrad large droppath db {M}x{N}x{K} rps={rps}", db, (dy.float() * s2).sum(0), tol2)
        dw, db = ops.wgrad_raw(dy, x, True, torch.zeros_like(rs2), 49)
        RES.append((f"wgrad large all-dropped {M}x{N}x{K}", bool((dw == 0).all()) and bool((db == 0).all())))
        # the public entry point takes ANY vector (round-5 ADVICE): zeros and one NEGATIVE value (two-valued: the exact pass with a negative common
        # factor -- used to be taken for all-zero), only negatives, and mixed signs (general pass)
        for name, vec in (("zeros+negative", torch.where(rs2 == 0, rs2, torch.full_like(rs2, -1.25))), ("all negative", torch.full_like(rs2, -0.75)),
                          ("mixed signs", torch.where(torch.arange(rs2.numel(), device=rs2.device) % 3 == 0, -rs2 - 0.5, rs2 + 0.25))):
            dw, db = ops.wgrad_raw(dy, x, True, vec, 49)
            s3 = vec.repeat_interleave(49)[:M, None]
            report(f"wgrad large rowscale {name} dw {M}x{N}x{K}", dw, (dy.float() * s3).t() @ x.float(), 5e-3)
            report(f"wgrad large rowscale {name} db {M}x{N}x{K}", db, (dy.float() * s3).sum(0), 5e-3)
        del dy, x, dw, db, dw2, sdy


def t_mlp_fused():
    """fmmt_mlp_fwd (Swin stage 0 / 1 Mlp in one launch) against the two-launch form -- bit-identical by construction -- and
    against an fp32 torch restatement; ragged token counts, with / without residual, DropPath scale, pre-activation output;
    then the module-level backward (activation recomputed inside the weight-gradient kernel) against fp32 autograd"""
    dt = torch.bfloat16
    for (M, C) in [(2007040 // 8, 96), (4096 + 40, 96), (501760 // 4, 192), (5000, 192), (256, 96), (300, 192)]:
        x = rnd("x", (M, C), 1, dtype=dt)
        w1 = rnd("w1", (4 * C, C), 2, C ** -0.5, dtype=dt)
        b1 = rnd("b1", (4 * C,), 3, 0.1)
        w2 = rnd("w2", (C, 4 * C), 4, (4 * C) ** -0.5, dtype=dt)
        b2 = rnd("b2", (C,), 5, 0.1)
        res = rnd("res", (M, C), 6, dtype=dt)
        rs = rnd("rs", (M // 49 + 1,), 7).abs() + 0.5
        rs[0] = 0.0
        for variant in ("full", "plain"):
            r_, s_ = (res, rs) if variant == "full" else (None, None)
            hp_f = torch.empty((M, 4 * C), dtype=dt, device=dev)
            y_f = ops.mlp_fused_raw(x, w1, b1, w2, b2, r_, s_, 49, hp_f)
            hp_u = torch.empty((M, 4 * C), dtype=dt, device=dev)
            h_u = ops.linear_raw(x, w1, b1, epi=EPI_GELU, y_pre=hp_u)
            y_u = ops.linear_raw(h_u, w2, b2, res=r_, rowscale=s_, rows_per_scale=49)
            # pre-activation: same products, same rounding -> bit-identical; y: the fused kernel's GELU is the LDS table, the GEMM epilogue's
            # the table or the polynomial depending on the kernel -> equal to within one bf16 rounding of the hidden activation
            RES.append((f"mlp fused == two launches {variant} {M}x{C}", one_ulp(y_f, y_u) and bool(torch.equal(hp_f, hp_u))))
            if not RES[-1][1]:
                print(f"FAIL mlp fused vs two launches {variant} {M}x{C}: y {(y_f.float() - y_u.float()).abs().max().item():.3e} "
                      f"pre {(hp_f.float() - hp_u.float()).abs().max().item():.3e}", flush=True)
            y_n = ops.mlp_fused_raw(x, w1, b1, w2, b2, r_, s_, 49, None)             # inference form: no pre-activation output
            RES.append((f"mlp fused no-pre identical {variant} {M}x{C}", bool(torch.equal(y_n, y_f))))
            ha = torch.empty((M, 4 * C), dtype=dt, device=dev)
            y_a = ops.mlp_fused_raw(x, w1, b1, w2, b2, r_, s_, 49, hp_f, ha)         # both hidden tensors stored
            RES.append((f"mlp fused + activation identical {variant} {M}x{C}", bool(torch.equal(y_a, y_f)) and one_ulp(ha, h_u)))
            del ha, y_a
            pre = x.float() @ w1.float().t() + b1
            ref = torch.nn.functional.gelu(pre).to(dt).float() @ w2.float().t() + b2
            if variant == "full":
                ref = res.float() + rs.repeat_interleave(49)[:M, None] * ref
            report(f"mlp fused vs fp32 {variant} {M}x{C}", y_f, ref, 2e-2)
            report(f"mlp fused pre vs fp32 {variant} {M}x{C}", hp_f, pre, 2e-2)
        del x, res, hp_f, hp_u, h_u, y_f, y_u, y_n, pre, ref
    # autograd through ops.mlp (fused forward; GELU' from the pre-activation; activation recomputed in the weight gradient)
    for (M, C) in [(12544, 96), (6272, 192)]:
        x = rnd("x", (M, C), 1, dtype=dt).requires_grad_(True)
        w1 = rnd("w1", (4 * C, C), 2, C ** -0.5).requires_grad_(True)
        b1 = rnd("b1", (4 * C,), 3, 0.1).requires_grad_(True)
        w2 = rnd("w2", (C, 4 * C), 4, (4 * C) ** -0.5).requires_grad_(True)
        b2 = rnd("b2", (C,), 5, 0.1).requires_grad_(True)
        res = rnd("res", (M, C), 6, dtype=dt).requires_grad_(True)
        rs = rnd("rs", (M // 49,), 7).abs() + 0.5
        y = ops.mlp(x, w1, b1, w2, b2, res, rs, 49)
        dy = rnd("dy", (M, C), 8, dtype=dt)
        y.backward(dy)
        xr = x.detach().float().requires_grad_(True)
        w1r, b1r, w2r, b2r = (t.detach().clone().requires_grad_(True) for t in (w1, b1, w2, b2))
        rr = res.detach().float().requires_grad_(True)
        h = torch.nn.functional.gelu(xr @ w1r.to(dt).float().t() + b1r)
        yr = rr + rs.repeat_interleave(49)[:, None] * (h @ w2r.to(dt).float().t() + b2r)
        yr.backward(dy.float())
        report(f"mlp autograd y {M}x{C}", y, yr, 2e-2)
        for name, g, r in (("dx", x.grad, xr.grad), ("dw1", w1.grad, w1r.grad), ("db1", b1.grad, b1r.grad), ("dw2", w2.grad, w2r.grad),
                           ("db2", b2.grad, b2r.grad), ("dres", res.grad, rr.grad)):
            report(f"mlp autograd {name} {M}x{C}", g, r, 4e-2)


def t_gelu_tail():
    """bf16 GELU / GELU' epilogues over x in [-8, 8] with a RELATIVE and a SIGN check on the negative tail (round-4 ADVICE, round-5 VERDICT weak 1):
    the round-4 odd polynomials were accurate to 5e-5 absolute -- tens of percent, either sign, where |gelu| is smaller than that (x < -3.5).  The
    sweep reaches the epilogues exactly: identity weights (stacked identities for the Mlp) make the fp32 pre-activation the bf16 sweep value itself,
    so the kernel's output is bf16(gelu(x)) and is compared with the bf16-rounded fp64 erf-GELU at 8.5e-3 relative (ONE bf16 step: the form's 4.6e-4 / 7.1e-4 may
    tip a rounding),
    gelu(x) <= 0 for x < 0, gelu'(x) < 0 for x in [-8, -1.5].  Kernel families: few-token tiles, the many-token persistent / phase kernels
    (stage-2 / stage-3 shapes, GELU + pre-activation and GELU'), the fused Mlp forward (C = 96, 192) and its fused backward."""
    import math
    dt = torch.bfloat16

    def sweep(M, N):
        # every row the same N-point sweep of [-8, 8] (bf16-representable values), dense on the negative tail; rows shifted so that every lane / tile position sees the tail
        base = torch.cat([torch.linspace(-8.0, -3.0, N // 2), torch.linspace(-3.0, 8.0, N - N // 2)]).to(dt)
        idx = (torch.arange(N)[None, :] + torch.arange(M)[:, None] * 7) % N
        return base[idx].to(dev).contiguous()

    def gelu64(v):
        v = v.double()
        return 0.5 * v * torch.special.erfc(-v / math.sqrt(2.0))          # erfc: no cancellation on the negative tail

    def grad64(v):
        v = v.double()
        return 0.5 * torch.special.erfc(-v / math.sqrt(2.0)) + v * torch.exp(-0.5 * v * v) / math.sqrt(2.0 * math.pi)

    def check(name, got, x, ref, lo, hi, want_neg):
        got, ref = got.double(), ref.to(dt).double()                         # reference rounded like the output: what is left is half a step + the form
        m = (x.double() >= lo) & (x.double() <= hi)
        rel = ((got[m] - ref[m]).abs() / ref[m].abs().clamp_min(1e-300)).max().item()
        sign_ok = bool((got[m] < 0).all()) if want_neg else True
        absr = (got - ref).abs().max().item()
        ok = rel <= 8.5e-3 and sign_ok and absr <= 4e-2 and bool(torch.isfinite(got).all())
        RES.append((name, ok))
        print(f"{'OK  ' if ok else 'FAIL'} {name:58s} tail rel={rel:.3e} sign={'ok' if sign_ok else 'WRONG'} max|err|={absr:.3e}", flush=True)

    for (M, N) in [(300, 384), (300, 768), (70000, 384), (31360, 768), (125440, 1536), (7840, 1536)]:
        x = sweep(M, N)
        w = torch.eye(N, dtype=dt, device=dev)
        ypre = torch.empty((M, N), dtype=dt, device=dev)
        y = ops.linear_raw(x, w, None, epi=EPI_GELU, y_pre=ypre)
        RES.append((f"gelu tail pre exact {M}x{N}", bool(torch.equal(ypre, x))))
        check(f"gelu tail fwd {M}x{N}", y, x, gelu64(x), -8.0, -1e-3, True)
        ones = torch.ones((M, N), dtype=dt, device=dev)
        g = ops.linear_raw(ones, w, None, epi=EPI_GELU_BWD, aux=x)
        check(f"gelu tail bwd {M}x{N}", g, x, grad64(x), -8.0, -1.5, True)
        two = ops.linear_raw(ones * 2, w, None, epi=EPI_GELU_BWD, aux=x)
        check(f"gelu tail bwd x2 {M}x{N}", two, x, 2.0 * grad64(x), -8.0, -1.5, True)
        ydg = torch.empty((M, N), dtype=dt, device=dev)
        y = ops.linear_raw(x, w, None, epi=EPI_GELU_DG, y_pre=ydg)      # the forward that stores the derivative (shared exponential)
        check(f"gelu tail dg fwd {M}x{N}", y, x, gelu64(x), -8.0, -1e-3, True)
        check(f"gelu tail dg derivative {M}x{N}", ydg, x, grad64(x), -8.0, -1.5, True)
        del x, w, ypre, y, ones, g, two, ydg
    for (M, C) in [(256 * 40, 96), (256 * 12 + 40, 96), (256 * 24, 192), (5000, 192)]:
        x = sweep(M, C)
        w1 = torch.eye(C, dtype=dt, device=dev).repeat(4, 1).contiguous()      # hidden channel h sees x[:, h % C]
        b1 = torch.zeros(4 * C, device=dev)
        w2 = rnd("w2", (C, 4 * C), 4, (4 * C) ** -0.5, dtype=dt)
        b2 = torch.zeros(C, device=dev)
        hp = torch.empty((M, 4 * C), dtype=dt, device=dev)
        ha = torch.empty((M, 4 * C), dtype=dt, device=dev)
        ops.mlp_fused_raw(x, w1, b1, w2, b2, None, None, 49, hp, ha)
        xx = x.repeat(1, 4)
        RES.append((f"gelu tail mlp pre exact {M}x{C}", bool(torch.equal(hp, xx))))
        check(f"gelu tail mlp fwd {M}x{C}", ha, xx, gelu64(xx), -8.0, -1e-3, True)
        # fused backward: dy = first unit vector, W2 row 0 = ones  ->  (dy . W2) = 1 for every hidden channel, dh = gelu'(h_pre)
        dy = torch.zeros((M, C), dtype=dt, device=dev)
        dy[:, 0] = 1.0
        w2b = torch.zeros((C, 4 * C), dtype=dt, device=dev)
        w2b[0] = 1.0
        dh, _ = ops.mlp_bwd_input_raw(dy, xx.contiguous(), w1, w2b, None, 49)
        check(f"gelu tail mlp bwd {M}x{C}", dh, xx, grad64(xx), -8.0, -1.5, True)
        del x, hp, ha, xx, dy, dh


def t_layernorm():
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)):
        for (M, C) in [(100, 96), (77, 192), (50, 384), (33, 768), (20, 1536), (10, 500 if dt == torch.float32 else 504), (1, 96), (1, 768)]:
            x = rnd("x", (M, C), 1, dtype=dt).requires_grad_(True)
            g = (rnd("g", (C,), 2) * 0.2 + 1).requires_grad_(True)
            b = (rnd("b", (C,), 3) * 0.1).requires_grad_(True)
            y = ops.layer_norm(x, g, b)
            xr = x.detach().double().requires_grad_(True)
            gr, br = g.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
            yr = OS.layer_norm(xr, gr, br)
            report(f"ln fwd {dt} {M}x{C}", y, yr, tol)
            dy = rnd("dy", (M, C), 4, dtype=dt)
            y.backward(dy)
            yr.backward(dy.double())
            report(f"ln bwd dx {dt} {M}x{C}", x.grad, xr.grad, tol * 2)
            report(f"ln bwd dg {dt} {M}x{C}", g.grad, gr.grad, tol * 2)
            report(f"ln bwd db {dt} {M}x{C}", b.grad, br.grad, tol * 2)
        for (n, H, Cq) in [(2, 56, 96), (3, 14, 384)]:
            x = rnd("x", (n, H * H, Cq), 1, dtype=dt).requires_grad_(True)
            g = (rnd("g", (4 * Cq,), 2) * 0.2 + 1).requires_grad_(True)
            b = (rnd("b", (4 * Cq,), 3) * 0.1).requires_grad_(True)
            y = ops.layer_norm(x, g, b, 1e-5, H)
            xr = x.detach().double().requires_grad_(True)
            gq = xr.reshape(n, H // 2, 2, H // 2, 2, Cq)
            cat = torch.cat([gq[:, :, 0, :, 0], gq[:, :, 1, :, 0], gq[:, :, 0, :, 1], gq[:, :, 1, :, 1]], -1).reshape(n, -1, 4 * Cq)
            yr = OS.layer_norm(cat, g.detach().double(), b.detach().double())
            report(f"ln merge fwd {dt} H={H}", y, yr, tol)
            dy = rnd("dy", tuple(y.shape), 4, dtype=dt)
            y.backward(dy)
            yr.backward(dy.double())
            report(f"ln merge bwd dx {dt} H={H}", x.grad, xr.grad, tol * 2)


def _wattn_ref(qkv, table, mask, n_img, H, C, nh, shift):
    idx = OS.window_token_index(H, H, 7, shift).to(qkv.device)
    nW = idx.shape[0]
    hd = C // nh
    t = qkv.reshape(n_img, H * H, 3, nh, hd)[:, idx.reshape(-1)].reshape(n_img * nW, 49, 3, nh, hd)
    q = t[:, :, 0].transpose(1, 2) * hd ** -0.5
    k, v = t[:, :, 1].transpose(1, 2), t[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) + table[OS.relative_position_index(7).to(qkv.device).reshape(-1)].reshape(49, 49, nh).permute(2, 0, 1)
    if mask is not None:
        s = (s.reshape(n_img, nW, nh, 49, 49) + mask[None, :, None]).reshape(n_img * nW, nh, 49, 49)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(n_img, nW * 49, C)
    out = torch.empty(n_img, H * H, C, dtype=o.dtype, device=o.device)
    out[:, idx.reshape(-1)] = o
    return out.reshape(-1, C)


def t_wattn():
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        for (n_img, H, C, nh, shift) in [(2, 14, 96, 3, 0), (2, 14, 96, 3, 3), (1, 28, 192, 6, 3), (3, 7, 768, 24, 0), (5, 14, 384, 12, 3),
                                         (1, 56, 96, 3, 3), (2, 21, 96, 3, 2), (1, 7, 768, 24, 0), (1, 7, 96, 3, 0)]:      # incl. a single window
            for mis in ((False, True) if shift else (False,)):
                qkv = rnd("qkv", (n_img * H * H, 3 * C), 1, dtype=dt).requires_grad_(True)
                table = (rnd("tab", (169, nh), 2) * 0.5).requires_grad_(True)
                mask = OS.shift_mask(H, H, 7, shift).to(dev) if shift else None
                out = ops.window_attn_core(qkv, table, index, mask, n_img, H, H, nh, shift, 32 ** -0.5, mis)
                q64 = qkv.detach().double().requires_grad_(True)
                t64 = table.detach().double().requires_grad_(True)
                ref = _wattn_ref(q64, t64, mask.double() if mask is not None else None, n_img, H, C, nh, shift)
                tag = f"{dt} n{n_img} H{H} C{C} s{shift} std{int(mis)}"
                report(f"wattn fwd {tag}", out, ref, tol)
                dy = rnd("dy", (n_img * H * H, C), 3, dtype=dt)
                out.backward(dy)
                ref.backward(dy.double())
                report(f"wattn bwd dqkv {tag}", qkv.grad, q64.grad, tol * 2)
                report(f"wattn bwd dtable {tag}", table.grad, t64.grad, tol * 2)


def t_mha():
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        for (Lq, Lk, B, E, nh) in [(38, 128, 2, 768, 12), (128, 38, 1, 768, 12), (166, 160, 4, 768, 12), (320, 166, 1, 768, 12), (166, 320, 2, 768, 12), (70, 33, 2, 256, 8),
                                   (1, 1, 1, 768, 12), (1, 65, 2, 768, 12), (64, 1, 1, 768, 12)]:      # single query / single key
            hd = E // nh
            q = rnd("q", (Lq, B, E), 1, dtype=dt).requires_grad_(True)
            kv = rnd("kv", (Lk, B, 2 * E), 2, dtype=dt).requires_grad_(True)
            out = ops.mha_core(q, kv, None, nh, hd ** -0.5)
            q64, kv64 = q.detach().double().requires_grad_(True), kv.detach().double().requires_grad_(True)
            qq = (q64 * hd ** -0.5).reshape(Lq, B * nh, hd).transpose(0, 1)
            kk = kv64[:, :, :E].reshape(Lk, B * nh, hd).transpose(0, 1)
            vv = kv64[:, :, E:].reshape(Lk, B * nh, hd).transpose(0, 1)
            ref = (torch.softmax(qq @ kk.transpose(1, 2), -1) @ vv).transpose(0, 1).reshape(Lq, B, E)
            tag = f"{dt} {Lq}x{Lk} B{B} E{E}"
            report(f"mha fwd {tag}", out, ref, tol)
            dy = rnd("dy", (Lq, B, E), 3, dtype=dt)
            out.backward(dy)
            ref.backward(dy.double())
            if Lk == 1:       # softmax over a single key is constant: dq is exactly 0 -- absolute check
                ok = q.grad.float().abs().max().item() <= 1e-5
                RES.append((f"mha bwd dq {tag}", ok))
                print(f"{'OK  ' if ok else 'FAIL'} mha bwd dq {tag} (single key: |dq| <= 1e-5)", flush=True)
            else:
                report(f"mha bwd dq {tag}", q.grad, q64.grad, tol * 2)
            report(f"mha bwd dkv {tag}", kv.grad, kv64.grad, tol * 2)
        # additive key bias (extended attention mask of the self-attention encoders): separate k, v, self-attention
        # lengths incl. ragged tails, -10000 on padded keys plus a smooth bias to exercise the general case
        for (L, B, E, nh) in [(128, 4, 768, 12), (37, 3, 768, 12), (160, 2, 768, 12), (320, 2, 768, 12), (70, 2, 256, 8)]:
            hd = E // nh
            q = rnd("q", (L, B, E), 1, dtype=dt).requires_grad_(True)
            k = rnd("k", (L, B, E), 2, dtype=dt).requires_grad_(True)
            v = rnd("v", (L, B, E), 3, dtype=dt).requires_grad_(True)
            kb = rnd("kb", (B, L), 4) * 0.5
            for b in range(B):
                kb[b, L - 1 - 7 * b:] += -10000.0
            out = ops.mha_core(q, k, v, nh, hd ** -0.5, 0.0, 0, kb)
            q64, k64, v64 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
            def hs(t):
                return t.reshape(L, B, nh, hd).permute(1, 2, 0, 3)
            sc = hs(q64) @ hs(k64).transpose(-1, -2) * hd ** -0.5 + kb.double()[:, None, None, :]
            ref = (torch.softmax(sc, -1) @ hs(v64)).permute(2, 0, 1, 3).reshape(L, B, E)
            tag = f"{dt} L{L} B{B} E{E} key_bias"
            report(f"mha fwd {tag}", out, ref, tol)
            dy = rnd("dy", (L, B, E), 5, dtype=dt)
            out.backward(dy)
            ref.backward(dy.double())
            report(f"mha bwd dq {tag}", q.grad, q64.grad, tol * 2)
            report(f"mha bwd dk {tag}", k.grad, k64.grad, tol * 2)
            report(f"mha bwd dv {tag}", v.grad, v64.grad, tol * 2)
        # FMMT_BATCH_MAJOR (round 6): the same launch on (batch, tokens, hidden) operands is the time-major one on transposed data, bit for bit -- both kernel
        # families (MFMA: bf16 head_dim 64; VALU: fp32, head_dim 32), cross lengths with ragged tails, packed column slices, key bias, replayed dropout
        from facialmmt_amd import _lib
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        for (Lq, Lk, B, E, nh, p) in [(77, 130, 3, 768, 12, 0.0), (130, 77, 2, 768, 12, 0.2), (64, 64, 4, 256, 8, 0.1), (200, 200, 2, 1024, 16, 0.1)]:
            es = 4 if dt == torch.float32 else 2
            code = _lib.dtype_code(dt)
            qb = rnd("qb", (B, Lq, E), 1, dtype=dt)
            kvb = rnd("kvb", (B, Lk, 2 * E), 2, dtype=dt)
            kb = rnd("kbb", (B, Lk), 3) * 0.5
            kb[:, Lk - 3:] = -10000.0
            dob = rnd("dob", (B, Lq, E), 4, dtype=dt)
            seed = torch.tensor([77], device=dev, dtype=torch.int64)
            res = []
            for bm in (1, 0):
                q_, kv_, do_ = (t if bm else t.transpose(0, 1).contiguous() for t in (qb, kvb, dob))
                out, dq, dkv = torch.empty_like(q_), torch.empty_like(q_), torch.empty_like(kv_)
                lse = torch.empty(B * nh * Lq, device=dev, dtype=torch.float32)
                flag = code | (_lib.BATCH_MAJOR if bm else 0)
                _lib.check(lib.fmmt_mha_fwd(flag, Lq, Lk, B, E, nh, q_.data_ptr(), E, kv_.data_ptr(), kv_.data_ptr() + E * es, 2 * E, (E // nh) ** -0.5, kb.data_ptr(), p, 0,
                                            seed.data_ptr(), out.data_ptr(), E, lse.data_ptr(), st), "mha fwd")
                _lib.check(lib.fmmt_mha_bwd(flag, Lq, Lk, B, E, nh, q_.data_ptr(), E, kv_.data_ptr(), kv_.data_ptr() + E * es, 2 * E, (E // nh) ** -0.5, kb.data_ptr(), p, 0,
                                            seed.data_ptr(), out.data_ptr(), do_.data_ptr(), E, lse.data_ptr(), dq.data_ptr(), E, dkv.data_ptr(), dkv.data_ptr() + E * es, 2 * E,
                                            st), "mha bwd")
                res.append(tuple(t if bm else t.transpose(0, 1) for t in (out, dq, dkv)) + (lse,))
            ok = all(torch.equal(a, b) for a, b in zip(*res)) and bool(torch.isfinite(res[0][0].float()).all())
            RES.append((f"mha batch-major {dt} {Lq}x{Lk} B{B} E{E} p{p}", ok))
            print(f"{'OK  ' if ok else 'FAIL'} mha batch-major == time-major {dt} {Lq}x{Lk} B{B} E{E} heads{nh} p={p}", flush=True)
        # separate k, v tensors + dropout statistics
        q = rnd("q", (64, 2, 768), 1, dtype=dt)
        k = rnd("k", (96, 2, 768), 2, dtype=dt)
        v = torch.ones((96, 2, 768), dtype=dt, device=dev)
        o0 = ops.mha_core(q, k, v, 12, 0.125, 0.0, 0)
        report(f"mha v=1 p=0 {dt}", o0, torch.ones_like(o0), tol)
        o1 = ops.mha_core(q, k, v, 12, 0.125, 0.25, 1234)
        print(f"     dropout p=0.25: mean(out)={o1.float().mean().item():.4f} (expect ~1), std={o1.float().std().item():.4f}")


def t_misc():
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-6)):
        img = rnd("img", (2, 3, 224, 224), 1, dtype=dt).requires_grad_(True)
        cols = ops.patch_im2col(img)
        ref = img.detach().reshape(2, 3, 56, 4, 56, 4).permute(0, 2, 4, 1, 3, 5).reshape(2 * 3136, 48)
        report(f"im2col {dt}", cols, ref, tol)
        cols.backward(cols.detach())
        report(f"col2im {dt}", img.grad, img.detach(), tol)
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)):
        for training in (True, False):
            x = rnd("x", (7, 512), 1, dtype=dt).requires_grad_(True)
            g = (rnd("g", (512,), 2) * 0.2 + 1).requires_grad_(True)
            b = (rnd("b", (512,), 3) * 0.1).requires_grad_(True)
            rm, rv = rnd("rm", (512,), 4) * 0.1, rnd("rv", (512,), 5).abs() + 0.5
            rm2, rv2 = rm.clone(), rv.clone()
            y = ops.batch_norm_1d(x, g, b, rm, rv, 0.1, 1e-5, training)
            xr, gr, br = x.detach().float().requires_grad_(True), g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
            yr = torch.nn.functional.batch_norm(xr, rm2, rv2, gr, br, training, 0.1, 1e-5)
            report(f"bn fwd {dt} train={training}", y, yr, tol)
            report(f"bn running_mean {dt} train={training}", rm, rm2, 1e-5)
            report(f"bn running_var {dt} train={training}", rv, rv2, 1e-5 if dt == torch.float32 else 2e-2)
            dy = rnd("dy", (7, 512), 6, dtype=dt)
            y.backward(dy)
            yr.backward(dy.float())
            report(f"bn bwd dx {dt} train={training}", x.grad, xr.grad, tol * 4)
            report(f"bn bwd dg {dt} train={training}", g.grad, gr.grad, tol * 4)
        x = rnd("x", (38, 2, 768), 1, dtype=dt)
        x[30:] = 0
        x[1, 0, 0] = 0
        tab = OC.sinusoidal_table(39, 768).to(dev)
        y = ops.posemb_scale(x, tab, 768 ** 0.5)
        report(f"posemb {dt}", y, OC.embed(x.float(), 768), tol)


def t_speed():
    """first timing impressions of the dominant kernels at bench-like sizes (bf16)"""
    def timeit(fn, n=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    dt = torch.bfloat16
    for (M, N, K) in [(2007040, 288, 96), (2007040, 96, 96), (2007040, 384, 96), (2007040, 96, 384),
                      (501760, 768, 192), (125440, 1536, 384), (125440, 384, 1536), (31360, 3072, 768), (640, 512, 37632)]:
        x = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt)
        b = torch.randn(N, device=dev)
        t = timeit(lambda: ops.linear_raw(x, w, b))
        dy = torch.randn(M, N, device=dev, dtype=dt)
        t2 = timeit(lambda: ops.wgrad_raw(dy, x, True))
        fl = 2.0 * M * N * K
        byts = (M * K + M * N + N * K) * 2
        print(f"     gemm {M}x{N}x{K}: fwd {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TF/s {byts/t/1e9:7.0f} GB/s | wgrad {t2*1e3:8.3f} ms {fl/t2/1e12:7.1f} TF/s", flush=True)
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    for (n_img, H, C, nh) in [(640, 56, 96, 3), (640, 28, 192, 6), (640, 14, 384, 12), (640, 7, 768, 24)]:
        qkv = torch.randn(n_img * H * H, 3 * C, device=dev, dtype=dt).requires_grad_(True)
        table = torch.randn(169, nh, device=dev).requires_grad_(True)
        mask = OS.shift_mask(H, H, 7, 3).to(dev) if H > 7 else None
        t = timeit(lambda: ops.window_attn_core(qkv, table, index, mask, n_img, H, H, nh, 3 if H > 7 else 0, 32 ** -0.5, True), 5)
        out = ops.window_attn_core(qkv, table, index, mask, n_img, H, H, nh, 3 if H > 7 else 0, 32 ** -0.5, True)
        dy = torch.randn_like(out)
        t2 = timeit(lambda: torch.autograd.grad(out, (qkv, table), dy, retain_graph=True), 5)
        print(f"     wattn n{n_img} H{H} C{C}: fwd {t*1e3:8.3f} ms | bwd {t2*1e3:8.3f} ms", flush=True)
    for (M, C) in [(2007040, 96), (125440, 384)]:
        x = torch.randn(M, C, device=dev, dtype=dt).requires_grad_(True)
        g = torch.ones(C, device=dev, requires_grad=True)
        b = torch.zeros(C, device=dev, requires_grad=True)
        t = timeit(lambda: ops.layer_norm(x, g, b))
        y = ops.layer_norm(x, g, b)
        dy = torch.randn_like(y)
        t2 = timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True))
        print(f"     ln {M}x{C}: fwd {t*1e3:8.3f} ms {M*C*4/t/1e9:7.0f} GB/s | bwd {t2*1e3:8.3f} ms {M*C*6/t2/1e9:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.version.hip)
    for f in (t_linear, t_linear_large, t_wgrad, t_wgrad_large, t_mlp_fused, t_layernorm, t_wattn, t_mha, t_misc):
        section(f)
    bad = [n for n, ok in RES if not ok]
    print(f"\nSUMMARY: {len(RES) - len(bad)} ok, {len(bad)} failed")
    for n in bad:
        print("  FAILED:", n)
    if "--speed" in sys.argv:
        section(t_speed)

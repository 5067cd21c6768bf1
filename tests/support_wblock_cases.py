"""Development probe of the fused attention half of a Swin block (csrc/wblock.hip), not a pytest file.
  python tests/support_wblock_cases.py            correctness: fused vs the four-launch form (bit-identical given the same LN output),
                                        vs an fp64 restatement, gradients through both paths
  python tests/support_wblock_cases.py --speed    timing at the bench geometry (640 frames), fused vs four launches, fwd and fwd+bwd"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facialmmt_amd import _lib  # noqa: E402
if os.environ.get("PROBE_LIB"):                                # --speed: same-call A/B of two builds
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops, synth  # noqa: E402
from oracle import swin as OS  # noqa: E402

dev = torch.device("cuda:0")
BAD = []


def rnd(name, shape, seed=0, scale=1.0, dtype=torch.float32):
    return (synth.tensor(name, shape, seed=seed) * scale).to(dev).to(dtype)


def report(name, got, ref, tol, exact=False):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    ok = (err == 0.0) if exact else (err <= tol * max(scale, 1e-6))
    ok = ok and torch.isfinite(got).all().item()
    if not ok:
        BAD.append(name)
    print(f"{'OK  ' if ok else 'FAIL'} {name:64s} max|err|={err:.3e} ref_scale={scale:.3e} {'exact' if exact else f'tol={tol:g}'}", flush=True)


def params(C, nh, seed=0):
    P = {}
    P["g"] = (1.0 + 0.2 * rnd("g", (C,), seed + 1)).requires_grad_(True)
    P["b"] = (0.1 * rnd("b", (C,), seed + 2)).requires_grad_(True)
    P["wqkv"] = rnd("wqkv", (3 * C, C), seed + 3, C ** -0.5).requires_grad_(True)
    P["bqkv"] = (0.1 * rnd("bqkv", (3 * C,), seed + 4)).requires_grad_(True)
    P["wproj"] = rnd("wproj", (C, C), seed + 5, C ** -0.5).requires_grad_(True)
    P["bproj"] = (0.1 * rnd("bproj", (C,), seed + 6)).requires_grad_(True)
    P["table"] = (0.5 * rnd("tab", (169, nh), seed + 7)).requires_grad_(True)
    return P


def four_launch(x, P, index, mask, n_img, H, nh, shift, rs, xn_override=None):
    """the reference composition on the same kernels: LN -> qkv -> attention core -> proj(+res, rowscale)"""
    C = x.shape[-1]
    xr, xn = ops.residual_layer_norm(x, P["g"], P["b"], 1e-5)
    if xn_override is not None:
        xn = xn_override
    qkv = ops.linear(xn, P["wqkv"], P["bqkv"])
    o = ops.window_attn_core(qkv.view(-1, 3 * C), P["table"], index, mask, n_img, H, H, nh, shift, 32 ** -0.5, mask is not None)
    return ops.linear(o.view(n_img, H * H, C), P["wproj"], P["bproj"], xr, rs, H * H), xn, o


def ref64(x, P, mask, n_img, H, nh, shift, rs):
    C = x.shape[-1]
    x64 = x.double()
    xn = torch.nn.functional.layer_norm(x64, (C,), P["g"].double(), P["b"].double(), 1e-5)
    qkv = xn.reshape(-1, C) @ P["wqkv"].double().t() + P["bqkv"].double()
    from tests.support_op_cases import _wattn_ref as wattn_ref
    o = wattn_ref(qkv, P["table"].double(), mask.double() if mask is not None else None, n_img, H, C, nh, shift)
    y = o @ P["wproj"].double().t() + P["bproj"].double()
    s = rs.double().repeat_interleave(H * H)[:, None] if rs is not None else 1.0
    return x64.reshape(-1, C) + s * y


def correctness():
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    C, nh = 96, 3
    for (n_img, H, shift, use_rs) in [(2, 14, 0, False), (2, 14, 3, True), (1, 56, 3, False), (3, 7, 0, True), (2, 21, 2, True), (5, 28, 3, False),
                                      (1, 7, 0, False), (33, 14, 3, True)]:
        tag = f"n{n_img} H{H} s{shift} rs{int(use_rs)}"
        P = params(C, nh, seed=n_img)
        x = rnd("x", (n_img, H * H, C), 11, dtype=torch.bfloat16).requires_grad_(True)
        mask = OS.shift_mask(H, H, 7, shift).to(dev) if shift else None
        rs = None
        if use_rs:
            rs = (rnd("rs", (n_img,), 5).abs() + 0.5)
            rs[0] = 0.0                                                       # a dropped sample
        y = ops.window_block(x, P["g"], P["b"], 1e-5, P["wqkv"], P["bqkv"], P["wproj"], P["bproj"], P["table"], index, mask, n_img, H, H, nh, shift,
                             32 ** -0.5, rs)
        # saved tensors of the fused forward
        y_raw, xn_f, o_f, mean_f, rstd_f, lse_f = ops.window_block_raw(
            x.detach().reshape(-1, C), n_img, H, H, nh, shift, P["g"].detach(), P["b"].detach(), 1e-5, P["wqkv"].detach().bfloat16(), P["bqkv"].detach(),
            P["wproj"].detach().bfloat16(), P["bproj"].detach(), P["table"].detach(), index, 32 ** -0.5, rs, True)
        report(f"wblock y == raw call {tag}", y.reshape(-1, C), y_raw, 0, exact=True)
        with torch.no_grad():
            y4, xn4, o4 = four_launch(x.detach(), P, index, mask, n_img, H, nh, shift, rs)
            report(f"wblock xn vs LN kernel {tag}", xn_f, xn4.reshape(-1, C), 1e-2)
            y4b, _, o4b = four_launch(x.detach(), P, index, mask, n_img, H, nh, shift, rs, xn_override=xn_f.view(n_img, H * H, C))
            report(f"wblock attn_out vs four-launch (same xn) {tag}", o_f, o4b, 1e-2)
            report(f"wblock y vs four-launch (same xn) {tag}", y_raw, y4b.reshape(-1, C), 1e-2)
            report(f"wblock y vs four-launch {tag}", y_raw, y4.reshape(-1, C), 2e-2)
        r64 = ref64(x.detach(), P, mask, n_img, H, nh, shift, rs)
        report(f"wblock y vs fp64 {tag}", y_raw, r64, 2e-2)
        # gradients: fused forward + its backward vs the four-launch autograd, both vs fp64 autograd
        dy = rnd("dy", (n_img, H * H, C), 13, dtype=torch.bfloat16)
        names = ["x", "g", "b", "wqkv", "bqkv", "wproj", "bproj", "table"]
        leaves = [x] + [P[k] for k in names[1:]]
        gf = torch.autograd.grad(y, leaves, dy)
        y4g, _, _ = four_launch(x, P, index, mask, n_img, H, nh, shift, rs)
        g4 = torch.autograd.grad(y4g, leaves, dy)
        x64 = x.detach().double().requires_grad_(True)
        P64 = {k: v.detach().double().requires_grad_(True) for k, v in P.items()}
        r = ref64(x64, P64, mask, n_img, H, nh, shift, rs)
        g64 = torch.autograd.grad(r, [x64] + [P64[k] for k in names[1:]], dy.double().reshape(-1, C))
        for nm, a, b4, c in zip(names, gf, g4, g64):
            report(f"wblock grad {nm} vs fp64 {tag}", a, c.reshape(a.shape), 4e-2)
            report(f"four-launch grad {nm} vs fp64 {tag}", b4, c.reshape(a.shape), 4e-2)
            c = c.reshape(a.shape)
            l2 = lambda u: ((u.double() - c).norm() / c.norm()).item()
            print(f"     rel-L2 error of grad {nm}: fused {l2(a):.4f}  four-launch {l2(b4):.4f}", flush=True)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def speed():
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    C, nh, N, H = 96, 3, int(os.environ.get("WB_FRAMES", "640")), 56
    P = params(C, nh)
    for shift in (0, 3):
        x = torch.randn(N, H * H, C, device=dev, dtype=torch.bfloat16).requires_grad_(True)
        mask = OS.shift_mask(H, H, 7, shift).to(dev) if shift else None
        rs = torch.ones(N, device=dev)
        dy = torch.randn_like(x)
        leaves = [x] + list(P.values())

        def fused():
            return ops.window_block(x, P["g"], P["b"], 1e-5, P["wqkv"], P["bqkv"], P["wproj"], P["bproj"], P["table"], index, mask, N, H, H, nh, shift, 32 ** -0.5, rs)

        def four():
            return four_launch(x, P, index, mask, N, H, nh, shift, rs)[0]

        only = os.environ.get("WB_FUSED_ONLY") == "1"
        for nm, f in ((("fused", fused),) if only else (("fused", fused), ("four-launch", four))):
            with torch.no_grad():
                tf = timeit(f, 10)
            tfb = 0.0 if only else timeit(lambda: torch.autograd.grad(f(), leaves, dy))
            by = N * H * H * C * 2
            print(f"stage0 shift{shift} {nm:12s}: fwd {tf*1e3:7.3f} ms ({2*by/tf/1e9:6.0f} GB/s of x+y) | fwd+bwd {tfb*1e3:7.3f} ms", flush=True)


def fwd_only():
    """a few fused forward launches at the bench geometry and nothing else: the target of rocprofv3 --pmc passes"""
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    C, nh, N, H = 96, 3, int(os.environ.get("WB_FRAMES", "640")), 56
    P = params(C, nh)
    x = torch.randn(N, H * H, C, device=dev, dtype=torch.bfloat16)
    save = "--save" in sys.argv
    for shift in (0, 3):
        for _ in range(3):
            ops.window_block_raw(x.reshape(-1, C), N, H, H, nh, shift, P["g"].detach(), P["b"].detach(), 1e-5, P["wqkv"].detach().bfloat16(), P["bqkv"].detach(),
                                 P["wproj"].detach().bfloat16(), P["bproj"].detach(), P["table"].detach(), index, 32 ** -0.5, None, save)
    torch.cuda.synchronize()


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.version.hip)
    if "--fwd-only" in sys.argv:
        fwd_only()
    elif "--speed" in sys.argv:
        speed()
    else:
        correctness()
        print(f"\nSUMMARY: {len(BAD)} failed")
        for n in BAD:
            print("  FAILED:", n)
        sys.exit(1 if BAD else 0)

"""fmmt_window_block_fwd (csrc/wblock.hip): the attention half of a stage-0 Swin block in one launch -- norm1, qkv, (shifted-)window
attention, proj, DropPath scale, residual (Swin_Transformer.py:233-266).  Parity, all through the C ABI:
  * against an fp64 restatement of the reference maths (the same helpers the per-op probe uses; window gather by index tables),
  * against the four-launch composition on the same kernels (LayerNorm -> Linear -> attention core -> Linear), forward and every
    gradient, including a dropped DropPath sample, ragged last workgroups and a single window,
  * through the module: SwinTransformerBlock / BasicLayer with the fused path on and off (ops._WBLOCK), and the
    reference-generated block goldens in bf16."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from facialmmt_amd import ops, synth  # noqa: E402
from oracle import swin as OS  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    return (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-6)


CASES = [(96, 2, 14, 0, False), (96, 2, 14, 3, True), (96, 1, 56, 3, False), (96, 3, 7, 0, True), (96, 2, 21, 2, True), (96, 5, 28, 3, False), (96, 1, 7, 0, False),
         (96, 33, 14, 3, True), (96, 8, 56, 3, True),
         (192, 2, 14, 0, False), (192, 3, 28, 3, True), (192, 1, 7, 0, True), (192, 9, 14, 3, False)]     # stage-1 width: four forward launches, the same recompute backward


@pytest.mark.parametrize("C,n_img,H,shift,use_rs", CASES)
def test_fused_block_half_forward_and_gradients(dev, C, n_img, H, shift, use_rs, monkeypatch):
    import support_wblock_cases as W
    nh = C // 32
    monkeypatch.setattr(ops, "_WBLOCK_WIDTHS", (96, 192))
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    P = W.params(C, nh, seed=n_img)
    x = W.rnd("x", (n_img, H * H, C), 11, dtype=torch.bfloat16).requires_grad_(True)
    mask = OS.shift_mask(H, H, 7, shift).to(dev) if shift else None
    rs = None
    if use_rs:
        rs = W.rnd("rs", (n_img,), 5).abs() + 0.5
        rs[0] = 0.0                                                                      # a dropped sample
    y = ops.window_block(x, P["g"], P["b"], 1e-5, P["wqkv"], P["bqkv"], P["wproj"], P["bproj"], P["table"], index, mask, n_img, H, H, nh, shift, 32 ** -0.5, rs)
    r64 = W.ref64(x.detach(), P, mask, n_img, H, nh, shift, rs)
    assert _rel(y.reshape(-1, C), r64) <= 2e-2
    y_raw, xn_f, o_f, mean_f, rstd_f, lse_f = ops.window_block_raw(
        x.detach().reshape(-1, C), n_img, H, H, nh, shift, P["g"].detach(), P["b"].detach(), 1e-5, P["wqkv"].detach().bfloat16(), P["bqkv"].detach(),
        P["wproj"].detach().bfloat16(), P["bproj"].detach(), P["table"].detach(), index, 32 ** -0.5, rs, True, mask)
    assert torch.equal(y.reshape(-1, C), y_raw)                                         # the saved-tensor form computes the same y
    with torch.no_grad():
        y4, xn4, o4 = W.four_launch(x.detach(), P, index, mask, n_img, H, nh, shift, rs)
        assert _rel(xn_f, xn4.reshape(-1, C)) <= 1e-2                                    # one bf16 rounding apart at most
        y4b, _, o4b = W.four_launch(x.detach(), P, index, mask, n_img, H, nh, shift, rs, xn_override=xn_f.view(n_img, H * H, C))
        assert _rel(o_f, o4b) <= 1e-2 and _rel(y_raw, y4b.reshape(-1, C)) <= 1e-2
    x64 = x.detach().double().reshape(-1, C)
    mu = x64.mean(-1)
    assert (mean_f.double() - mu).abs().max().item() <= 1e-5 * max(1.0, mu.abs().max().item())
    assert _rel(rstd_f, (x64.var(-1, unbiased=False) + 1e-5).rsqrt()) <= 1e-4
    # gradients of the fused forward + its backward against fp64 autograd, next to the four-launch autograd
    dy = W.rnd("dy", (n_img, H * H, C), 13, dtype=torch.bfloat16)
    names = ["x", "g", "b", "wqkv", "bqkv", "wproj", "bproj", "table"]
    leaves = [x] + [P[k] for k in names[1:]]
    gf = torch.autograd.grad(y, leaves, dy)
    xd = x.detach().double().requires_grad_(True)
    P64 = {k: v.detach().double().requires_grad_(True) for k, v in P.items()}
    g64 = torch.autograd.grad(W.ref64(xd, P64, mask, n_img, H, nh, shift, rs), [xd] + [P64[k] for k in names[1:]], dy.double().reshape(-1, C))
    for nm, a, c in zip(names, gf, g64):
        assert _rel(a, c.reshape(a.shape)) <= 4e-2, nm


def test_entry_point_refuses_what_it_does_not_cover(dev):
    from facialmmt_amd import _lib
    lib = _lib.load()
    x = torch.zeros(49, 192, dtype=torch.bfloat16, device=dev)
    z = torch.zeros(1024, dtype=torch.float32, device=dev)
    w = torch.zeros(576 * 192, dtype=torch.bfloat16, device=dev)
    idx = torch.zeros(49 * 49, dtype=torch.int32, device=dev)
    args = lambda dtype, C, nh: (dtype, 1, 7, 7, C, nh, 0, x.data_ptr(), z.data_ptr(), z.data_ptr(), 1e-5, w.data_ptr(), None, w.data_ptr(), None,
                                 z.data_ptr(), idx.data_ptr(), 0.17, None, x.data_ptr(), None, None, None, None, z.data_ptr(), None)
    assert lib.fmmt_window_block_fwd(*args(_lib.BF16, 192, 6)) == -1                     # FMMT_EINVAL: other widths take the four-launch form
    assert lib.fmmt_window_block_fwd(*args(_lib.F32, 192, 6)) == -1
    assert lib.fmmt_window_block_fwd_ref(*args(_lib.BF16, 192, 6)) == -1
    assert ops.window_block_fusable(x[:, :96].float(), 96, 3, (7, 7), 0, None, False)    # fp32 (parity): the generic instantiation, C = 96 only
    assert not ops.window_block_fusable(x.float(), 192, 6, (7, 7), 0, None, False)
    assert not ops.window_block_fusable(x, 96, 3, (7, 7), 3, torch.zeros(1, 49, 49, device=dev), False)     # a non-standard mask tensor


def _raw_block(lib, entry, dtype, x2, P, index, n_img, H, nh, shift, rs):
    """one call of a fused block-half entry point on explicit operands; returns (y, xn, attn_out, mean, rstd, lse)"""
    from facialmmt_amd import _lib
    code = _lib.dtype_code(dtype)
    M, C = x2.shape
    dev = x2.device
    y, xn, o = torch.empty_like(x2), torch.empty_like(x2), torch.empty_like(x2)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    lse = torch.empty(n_img * (H // 7) ** 2 * nh * 49, device=dev)
    keep = [P["g"].detach().float().contiguous(), P["b"].detach().float().contiguous(), P["wqkv"].detach().to(dtype).contiguous(), P["bqkv"].detach().float().contiguous(),
            P["wproj"].detach().to(dtype).contiguous(), P["bproj"].detach().float().contiguous(), P["table"].detach().float().contiguous()]
    rc = getattr(lib, entry)(code, n_img, H, H, C, nh, shift, x2.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), 1e-5, keep[2].data_ptr(), keep[3].data_ptr(),
                             keep[4].data_ptr(), keep[5].data_ptr(), keep[6].data_ptr(), index.data_ptr(), 32 ** -0.5, rs.data_ptr() if rs is not None else None,
                             y.data_ptr(), xn.data_ptr(), o.data_ptr(), mean.data_ptr(), rstd.data_ptr(), lse.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, (entry, rc)
    torch.cuda.synchronize()
    return y, xn, o, mean, rstd, lse


GENERIC_CASES = [(1, 7, 0, False), (2, 14, 3, True), (3, 56, 0, True), (2, 56, 3, False), (5, 28, 3, True)]


@pytest.mark.parametrize("n_img,H,shift,use_rs", GENERIC_CASES)
def test_generic_template_in_bf16_reproduces_the_production_kernel(dev, n_img, H, shift, use_rs):
    """csrc/wblock_ref.hip is wblock.hip's kernel written over an element-type trait.  Its bf16 instantiation must reproduce the kernel the
    benchmark runs: the attention output, log-sum-exp and y to at most one bf16 rounding step (the LayerNorm affine is packed fp32
    arithmetic in one and scalar in the other; everything behind it is the same instruction sequence on the same operands)."""
    import support_wblock_cases as W
    from facialmmt_amd import _lib
    from oracle import swin as OS
    lib = _lib.load()
    C, nh = 96, 3
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    P = W.params(C, nh, seed=20 + n_img)
    x2 = W.rnd("xg", (n_img * H * H, C), 31, dtype=torch.bfloat16)
    rs = None
    if use_rs:
        rs = W.rnd("rs", (n_img,), 5).abs() + 0.5
        rs[0] = 0.0
    prod = _raw_block(lib, "fmmt_window_block_fwd", torch.bfloat16, x2, P, index, n_img, H, nh, shift, rs)
    gen = _raw_block(lib, "fmmt_window_block_fwd_ref", torch.bfloat16, x2, P, index, n_img, H, nh, shift, rs)
    for name, a, b in zip(("y", "xn", "attn_out", "mean", "rstd", "lse"), prod, gen):
        a, b = a.float(), b.float()
        scale = b.abs().max().item()
        tol = 2.0 ** -7 * scale if name in ("y", "xn", "attn_out") else 1e-5 * max(1.0, scale)
        assert (a - b).abs().max().item() <= tol, (name, (a - b).abs().max().item(), scale)
        if name in ("y", "xn", "attn_out"):                  # and almost everywhere identical
            assert (a != b).float().mean().item() <= 0.05, name


@pytest.mark.parametrize("n_img,H,shift,use_rs", GENERIC_CASES)
def test_generic_backward_template_in_bf16_reproduces_the_production_kernel(dev, n_img, H, shift, use_rs):
    """csrc/wattn_bwd_ref.hip (the recompute backward over an element-type trait) in bf16 against the kernel behind
    fmmt_window_block_attn_bwd on the same saved tensors: dqkv to one bf16 rounding step and almost everywhere identical (same products,
    same rounding points; the fast kernel's transposing LDS reads and prefetch change no value), d(table) to fp32 summation order."""
    import support_wblock_cases as W
    from facialmmt_amd import _lib
    from oracle import swin as OS
    lib = _lib.load()
    C, nh = 96, 3
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    P = W.params(C, nh, seed=60 + n_img)
    x2 = W.rnd("xb", (n_img * H * H, C), 61, dtype=torch.bfloat16)
    rs = None
    if use_rs:
        rs = W.rnd("rs", (n_img,), 5).abs() + 0.5
        rs[0] = 0.0
    _, xn, o, _, _, lse = _raw_block(lib, "fmmt_window_block_fwd", torch.bfloat16, x2, P, index, n_img, H, nh, shift, rs)
    dy = W.rnd("dyb", (n_img * H * H, C), 63, dtype=torch.bfloat16)
    wq, bq, wp, tab = P["wqkv"].detach().bfloat16().contiguous(), P["bqkv"].detach().float().contiguous(), P["wproj"].detach().bfloat16().contiguous(), P["table"].detach().float().contiguous()
    nbytes = lib.fmmt_window_attn_bwd_workspace(nh)
    res = []
    for entry in ("fmmt_window_block_attn_bwd", "fmmt_window_block_attn_bwd_ref"):
        dqkv = torch.zeros(x2.shape[0], 3 * C, dtype=torch.bfloat16, device=dev)
        dtab = torch.empty_like(tab)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = getattr(lib, entry)(_lib.BF16, n_img, H, H, C, nh, shift, xn.data_ptr(), dy.data_ptr(), o.data_ptr(), lse.data_ptr(), wq.data_ptr(), bq.data_ptr(), wp.data_ptr(),
                                 tab.data_ptr(), index.data_ptr(), 32 ** -0.5, rs.data_ptr() if rs is not None else None, dqkv.data_ptr(), dtab.data_ptr(),
                                 ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, (entry, rc)
        torch.cuda.synchronize()
        res.append((dqkv.float(), dtab))
    (d0, t0), (d1, t1) = res
    scale = d1.abs().max().item()
    assert (d0 - d1).abs().max().item() <= 2.0 ** -7 * scale and (d0 != d1).float().mean().item() <= 0.02
    assert (t0 - t1).abs().max().item() <= 1e-4 * max(1e-6, t1.abs().max().item())


MATERIALISED_CASES = [(192, 6, 2, 28, 0), (192, 6, 3, 28, 3), (384, 12, 2, 14, 3), (768, 24, 3, 7, 0), (96, 3, 1, 56, 3)]


@pytest.mark.parametrize("C,nh,n_img,H,shift", MATERIALISED_CASES)
def test_generic_twin_of_the_materialised_window_attention_backward(dev, C, nh, n_img, H, shift):
    """fmmt_window_attn_bwd (q / k / v / d(out) READ from the qkv tensor: Swin stages 1-3, Swin_Transformer.py:113-144) has the same generic
    twin as the recompute form -- wattn_bwd_ref_kernel<T, MM, 0> -- reachable with FMMT_BF16 | FMMT_GENERIC and run by default for FMMT_F32:
      * its bf16 instantiation against the production kernel wattn_mfma_bwd_kernel<MM, 2, 0>: dqkv to one bf16 rounding step, almost
        everywhere identical; d(table) to fp32 summation order;
      * its fp32 instantiation against fp64 autograd of the oracle's attention core at 1e-4, and against the VALU fp32 kernel (a different
        algorithm, reached by handing the same mask over as an explicit tensor)."""
    from facialmmt_amd import _lib
    from oracle import swin as OS
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(700 + C + shift)
    rnd = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    M, nW = n_img * H * H, (H // 7) ** 2
    qkv = rnd(M, 3 * C)
    dout = rnd(M, C)
    table = (rnd(169, nh) * 0.5).contiguous()
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    mask = OS.shift_mask(H, H, 7, shift).to(dev).float().contiguous() if shift > 0 else None
    scale = 32 ** -0.5
    st = lambda: torch.cuda.current_stream().cuda_stream
    nbytes = lib.fmmt_window_attn_bwd_workspace(nh)

    def run(code, dt, explicit_mask=False):
        q, do = qkv.to(dt).contiguous(), dout.to(dt).contiguous()
        out = torch.empty(M, C, dtype=dt, device=dev)
        lse = torch.empty(n_img * nW * nh * 49, device=dev)
        mp = mask.data_ptr() if mask is not None else None
        is_shift = 0 if explicit_mask else (1 if mask is not None else 0)
        rc = lib.fmmt_window_attn_fwd(code & 0xff, n_img, H, H, C, nh, shift, q.data_ptr(), table.data_ptr(), index.data_ptr(), mp, nW if mask is not None else 0,
                                      is_shift, scale, out.data_ptr(), lse.data_ptr(), st())
        assert rc == 0
        dqkv = torch.zeros(M, 3 * C, dtype=dt, device=dev)
        dtab = torch.empty_like(table)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = lib.fmmt_window_attn_bwd(code, n_img, H, H, C, nh, shift, q.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(), table.data_ptr(), index.data_ptr(),
                                      mp, nW if mask is not None else 0, is_shift, scale, dqkv.data_ptr(), dtab.data_ptr(), ws.data_ptr(), nbytes, st())
        assert rc == 0, rc
        torch.cuda.synchronize()
        return dqkv.float(), dtab

    d0, t0 = run(_lib.BF16, torch.bfloat16)
    d1, t1 = run(_lib.BF16 | _lib.GENERIC, torch.bfloat16)
    sc = d1.abs().max().item()
    assert (d0 - d1).abs().max().item() <= 2.0 ** -7 * sc and (d0 != d1).float().mean().item() <= 0.03, ((d0 - d1).abs().max().item(), sc, (d0 != d1).float().mean().item())
    assert (t0 - t1).abs().max().item() <= 2e-4 * max(1e-6, t1.abs().max().item())
    # fp32 instantiation: fp64 autograd through the plain attention core
    d32, t32 = run(_lib.F32, torch.float32)
    q64 = qkv.double().requires_grad_(True)
    tab64 = table.double().requires_grad_(True)
    tok = OS.window_token_index(H, H, 7, shift).to(dev)                      # (nW, 49): roll + window_partition as a gather
    xw = q64.reshape(n_img, H * H, 3 * C)[:, tok].reshape(n_img * nW, 49, 3, nh, 32)
    q_, k_, v_ = xw[:, :, 0].transpose(1, 2) * scale, xw[:, :, 1].transpose(1, 2), xw[:, :, 2].transpose(1, 2)
    s_ = q_ @ k_.transpose(-1, -2) + tab64[index.long().reshape(-1)].reshape(49, 49, nh).permute(2, 0, 1)
    if mask is not None:
        s_ = (s_.reshape(n_img, nW, nh, 49, 49) + mask.double()[None, :, None]).reshape(n_img * nW, nh, 49, 49)
    ow = (torch.softmax(s_, -1) @ v_).transpose(1, 2).reshape(n_img, nW * 49, C)
    o64 = torch.zeros(n_img, H * H, C, dtype=torch.float64, device=dev).index_add(1, tok.reshape(-1), ow).reshape(M, C)
    gq, gt = torch.autograd.grad(o64, [q64, tab64], dout.double())
    assert _rel(d32, gq) <= 1e-4 and _rel(t32, gt) <= 1e-4
    if mask is not None:                                     # the VALU kernel on the same problem (explicit mask tensor)
        dv, tv = run(_lib.F32, torch.float32, explicit_mask=True)
        assert _rel(d32, dv) <= 1e-5 and _rel(t32, tv) <= 1e-4
        # the bf16 twin does not take an explicit mask: refused, not mis-computed
        q = qkv.bfloat16()
        assert lib.fmmt_window_attn_bwd(_lib.BF16 | _lib.GENERIC, n_img, H, H, C, nh, shift, q.data_ptr(), q.data_ptr(), q.data_ptr(), t0.data_ptr(), table.data_ptr(),
                                        index.data_ptr(), mask.data_ptr(), nW, 0, scale, q.data_ptr(), t0.data_ptr(), q.data_ptr(), nbytes, st()) == -1


@pytest.mark.parametrize("n_img,H,shift,use_rs", GENERIC_CASES)
def test_fp32_instantiation_of_the_fused_block_against_fp64(dev, n_img, H, shift, use_rs):
    """The fp32 instantiation of the same template (fp32 fragments, 8 x v_mfma_f32_16x16x4_f32 per 32-deep block, nothing rounded to bf16):
    north_star's 1e-3 on the fused kernel's ALGORITHM -- window / shift addressing, fragment-order weights, mask derivation, base-2
    softmax normalised after the second product -- against an fp64 restatement: forward, the saved LayerNorm output, statistics and
    log-sum-exp; and every gradient of the op against fp64 autograd -- its backward is the fp32 instantiation of the recompute kernel
    (csrc/wattn_bwd_ref.hip, through fmmt_window_block_attn_bwd) plus the fp32 GEMM / LayerNorm launches."""
    import support_wblock_cases as W
    from facialmmt_amd import _lib
    from oracle import swin as OS
    lib = _lib.load()
    C, nh = 96, 3
    index = OS.relative_position_index(7).to(dev).int().contiguous()
    P = W.params(C, nh, seed=40 + n_img)
    x = W.rnd("xf", (n_img, H * H, C), 41).requires_grad_(True)
    mask = OS.shift_mask(H, H, 7, shift).to(dev) if shift else None
    rs = None
    if use_rs:
        rs = W.rnd("rs", (n_img,), 5).abs() + 0.5
        rs[0] = 0.0
    y, xn, o, mean, rstd, lse = _raw_block(lib, "fmmt_window_block_fwd", torch.float32, x.detach().reshape(-1, C), P, index, n_img, H, nh, shift, rs)
    r64 = W.ref64(x.detach(), P, mask, n_img, H, nh, shift, rs)
    assert (y.double() - r64).abs().max().item() <= 1e-4 * max(1.0, r64.abs().max().item())      # fp32 arithmetic: well inside north_star's 1e-3
    x64 = x.detach().double().reshape(-1, C)
    xn64 = torch.nn.functional.layer_norm(x64, (C,), P["g"].detach().double(), P["b"].detach().double(), 1e-5)
    assert (xn.double() - xn64).abs().max().item() <= 1e-4
    assert (mean.double() - x64.mean(-1)).abs().max().item() <= 1e-5
    # through autograd: the op with fp32 operands = this forward + the fp32 parity backward
    assert ops.window_block_fusable(x, C, nh, (7, 7), shift, mask, shift > 0)
    yo = ops.window_block(x, P["g"], P["b"], 1e-5, P["wqkv"], P["bqkv"], P["wproj"], P["bproj"], P["table"], index, mask, n_img, H, H, nh, shift, 32 ** -0.5, rs)
    assert torch.equal(yo.reshape(-1, C), y)
    dy = W.rnd("dyf", (n_img, H * H, C), 43)
    names = ["x", "g", "b", "wqkv", "bqkv", "wproj", "bproj", "table"]
    gf = torch.autograd.grad(yo, [x] + [P[k] for k in names[1:]], dy)
    xd = x.detach().double().requires_grad_(True)
    P64 = {k: v.detach().double().requires_grad_(True) for k, v in P.items()}
    g64 = torch.autograd.grad(W.ref64(xd, P64, mask, n_img, H, nh, shift, rs), [xd] + [P64[k] for k in names[1:]], dy.double().reshape(-1, C))
    for nm, a, c in zip(names, gf, g64):
        assert _rel(a, c.reshape(a.shape)) <= 1e-3, nm


@pytest.mark.parametrize("shift", [0, 3])
def test_fp32_swin_block_reaches_the_golden_through_the_fused_kernel(dev, golden, shift, monkeypatch):
    """`block_s0_shift{0,3}` (outputs of the reference's SwinTransformerBlock) at 1e-3 through the fp32 module, whose attention half must
    have been ONE call of fmmt_window_block_fwd (the generic instantiation), not the four parity launches."""
    from facialmmt_amd import _lib
    from facialmmt_amd.modules.SwinTransformer.Swin_Transformer import SwinTransformerBlock
    blk = SwinTransformerBlock(96, (56, 56), 3, window_size=7, shift_size=shift, drop_path=0.0).eval()
    synth.fill_state_dict(blk, seed=10, prefix="blk0.")
    blk.to(dev)
    x = synth.tensor("blk_in0", (2, 3136, 96), seed=0).to(dev)
    lib = _lib.load()
    calls = []
    real = lib.fmmt_window_block_fwd
    monkeypatch.setattr(lib, "fmmt_window_block_fwd", lambda *a: (calls.append(a[0]), real(*a))[1])
    with torch.no_grad():
        y32 = blk(x)
    assert calls == [_lib.F32]
    golden.check("swin_parts", f"block_s0_shift{shift}", y32, atol=1e-3, rtol=1e-3)
    monkeypatch.setattr(ops, "_WBLOCK_F32", False)           # and the four fp32 launches agree with it
    with torch.no_grad():
        y4 = blk(x)
    assert len(calls) == 1 and (y4 - y32).abs().max().item() <= 2e-5 * max(1.0, y32.abs().max().item())


@pytest.mark.parametrize("shift", [0, 3])
def test_swin_block_module_takes_the_fused_path(dev, golden, shift, monkeypatch):
    """SwinTransformerBlock (C = 96) in bf16: the module with the fused attention half against the same module on the four-launch
    path, forward and parameter gradients; and against the reference-generated block golden (fp32 reference, bf16 tolerance)."""
    from facialmmt_amd.modules.SwinTransformer.Swin_Transformer import SwinTransformerBlock
    torch.manual_seed(0)
    blk = SwinTransformerBlock(96, (56, 56), 3, window_size=7, shift_size=shift).to(dev)
    synth.fill_state_dict(blk, seed=7 + shift, prefix=f"blk{shift}.")
    blk.to(dev).train()
    x = synth.tensor("bx", (2, 3136, 96), seed=3).to(dev).bfloat16()
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "_WBLOCK", fused)
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        y.float().square().mean().backward()
        outs.append((y.detach(), xi.grad, {k: p.grad.clone() for k, p in blk.named_parameters()}))
    (yf, dxf, gf), (y4, dx4, g4) = outs
    assert _rel(yf, y4) <= 2e-2 and _rel(dxf, dx4) <= 4e-2
    for k in gf:
        assert _rel(gf[k], g4[k]) <= 5e-2, k


@pytest.mark.parametrize("shift", [0, 3])
def test_fused_block_against_the_reference_generated_golden(dev, golden, shift):
    """`block_s0_shift{0,3}` of tests/golden/swin_parts.npz (outputs of the reference's SwinTransformerBlock on hash-generated weights,
    oracle/gen_golden.py) through the bf16 module, whose attention half is the fused launch and whose Mlp is the fused Mlp launch:
    held to 3e-2 of the output scale (the bf16 bar of this suite; the fp32 instantiation of the same block -- the four parity
    kernels -- is held to the same fixture at 1e-3 by tests/test_gpu_swin.py::test_block_and_window_attention)."""
    from facialmmt_amd.modules.SwinTransformer.Swin_Transformer import SwinTransformerBlock
    blk = SwinTransformerBlock(96, (56, 56), 3, window_size=7, shift_size=shift, drop_path=0.0).eval()
    synth.fill_state_dict(blk, seed=10, prefix="blk0.")
    blk.to(dev)
    x = synth.tensor("blk_in0", (2, 3136, 96), seed=0).to(dev)
    assert ops.window_block_fusable(x.bfloat16(), 96, 3, (7, 7), shift, blk.attn_mask, blk._mask_is_standard())
    with torch.no_grad():
        y16 = blk(x.bfloat16())
        y32 = blk(x)
    ref, stride = golden.expected("swin_parts", f"block_s0_shift{shift}")
    got = y16.float().cpu().numpy()
    got = got if stride is None else got.reshape(-1)[::stride]
    import numpy as np
    assert np.abs(got - ref).max() <= 3e-2 * np.abs(ref).max()
    golden.check("swin_parts", f"block_s0_shift{shift}", y32, atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("C,M,use_rs", [(96, 8192 + 77, True), (192, 4096, False), (96, 3136 * 2, False), (96, 8192, True)])   # last: M % 256 == 0, the guard-free instantiation
def test_mlp_half_with_layernorm_prologue(dev, C, M, use_rs, monkeypatch):
    """fmmt_mlp_ln_fwd (norm2 -> Mlp -> DropPath -> residual in one launch) against fmmt_layernorm_fwd followed by fmmt_mlp_fwd, and
    both against fp64: forward, the saved LayerNorm output and statistics, every gradient; ragged last tile, dropped sample.
    (Both widths the fused kernels exist for: the model itself runs them at C = 96 only since round 6, ops._MLP_FUSED_WIDTHS.)"""
    import support_wblock_cases as W
    monkeypatch.setattr(ops, "_MLP_FUSED_WIDTHS", (96, 192))
    x = W.rnd("x", (M, C), 21, dtype=torch.bfloat16).requires_grad_(True)
    P = [(1.0 + 0.2 * W.rnd("g", (C,), 22)).requires_grad_(True), (0.1 * W.rnd("b", (C,), 23)).requires_grad_(True),
         W.rnd("w1", (4 * C, C), 24, C ** -0.5).requires_grad_(True), (0.1 * W.rnd("b1", (4 * C,), 25)).requires_grad_(True),
         W.rnd("w2", (C, 4 * C), 26, (4 * C) ** -0.5).requires_grad_(True), (0.1 * W.rnd("b2", (C,), 27)).requires_grad_(True)]
    rps = 1024
    rs = None
    if use_rs:
        rs = W.rnd("rs", ((M + rps - 1) // rps,), 5).abs() + 0.5
        rs[1] = 0.0
    assert ops.mlp_ln_fusable(x, P[2], P[4], P[3], P[5])
    y = ops.mlp_ln(x, P[0], P[1], 1e-5, P[2], P[3], P[4], P[5], rs, rps)
    xr, xn = ops.residual_layer_norm(x, P[0], P[1], 1e-5)
    y2 = ops.mlp(xn, P[2], P[3], P[4], P[5], res=xr, rowscale=rs, rows_per_scale=rps)
    x64 = x.detach().double().requires_grad_(True)
    P64 = [p.detach().double().requires_grad_(True) for p in P]
    h64 = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x64, (C,), P64[0], P64[1], 1e-5) @ P64[2].t() + P64[3])
    s64 = rs.double().repeat_interleave(rps)[:M, None] if rs is not None else 1.0
    r64 = x64 + s64 * (h64 @ P64[4].t() + P64[5])
    assert _rel(y, r64) <= 2e-2 and _rel(y, y2) <= 1e-2
    dy = W.rnd("dy", (M, C), 29, dtype=torch.bfloat16)
    leaves = [x] + P
    g1 = torch.autograd.grad(y, leaves, dy)
    g2 = torch.autograd.grad(y2, leaves, dy)
    g64 = torch.autograd.grad(r64, [x64] + P64, dy.double())
    for a, b, c in zip(g1, g2, g64):
        assert _rel(a, c) <= 4e-2 and _rel(b, c) <= 4e-2
        l2 = lambda u: ((u.double() - c).norm() / c.norm()).item()
        assert l2(a) <= 1.5 * l2(b) + 1e-3                                            # as accurate as the two-launch form
    # the fused input-gradient launch hand-counts its in-flight loads: the same backward five more times, bit for bit
    for _ in range(5):
        y3 = ops.mlp_ln(x, P[0], P[1], 1e-5, P[2], P[3], P[4], P[5], rs, rps)
        g3 = torch.autograd.grad(y3, leaves, dy)
        for a, b in zip(g1, g3):
            assert torch.equal(a, b)


def _mlp_operands(C, M, use_rs, dtype, seed):
    import support_wblock_cases as W
    x = W.rnd("mx", (M, C), seed, dtype=dtype)
    P = dict(g=1.0 + 0.2 * W.rnd("g", (C,), seed + 1), b=0.1 * W.rnd("b", (C,), seed + 2), w1=W.rnd("w1", (4 * C, C), seed + 3, C ** -0.5),
             b1=0.1 * W.rnd("b1", (4 * C,), seed + 4), w2=W.rnd("w2", (C, 4 * C), seed + 5, (4 * C) ** -0.5), b2=0.1 * W.rnd("b2", (C,), seed + 6))
    rps, rs = 1024, None
    if use_rs:
        rs = W.rnd("rs", ((M + rps - 1) // rps,), 5).abs() + 0.5
        rs[1] = 0.0
    return x, P, rs, rps


@pytest.mark.parametrize("dg", [False, True])
@pytest.mark.parametrize("C,M,use_rs", [(96, 8192 + 77, True), (192, 4096 + 130, False), (192, 4096, False)])
def test_generic_mlp_templates_in_bf16_reproduce_the_production_kernels(dev, C, M, use_rs, dg):
    """csrc/mlp_ref.hip (the fused Mlp forward / input-gradient kernels over an element-type trait, weights read as fragments instead of
    through the DMA ring) in bf16 -- dtype FMMT_BF16 | FMMT_GENERIC on the same entry points -- against the kernels the benchmark runs:
    every output to one bf16 rounding step and almost everywhere identical; the LayerNorm-backward partial sums to fp32 summation order.
    dg: the same with FMMT_SAVE_DG (round 6: h_pre holds gelu'(pre-activation), the backward multiplies) -- what the model runs since."""
    from facialmmt_amd import _lib
    lib = _lib.load()
    x, P, rs, rps = _mlp_operands(C, M, use_rs, torch.bfloat16, 70)
    w1, w2 = P["w1"].bfloat16().contiguous(), P["w2"].bfloat16().contiguous()
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    st = lambda: torch.cuda.current_stream().cuda_stream
    outs = []
    for code in (_lib.BF16 | (_lib.SAVE_DG if dg else 0), _lib.BF16 | _lib.GENERIC | (_lib.SAVE_DG if dg else 0)):
        y, xn, hp, ha = torch.empty_like(x), torch.empty_like(x), torch.empty(M, 4 * C, dtype=x.dtype, device=dev), torch.empty(M, 4 * C, dtype=x.dtype, device=dev)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        rc = lib.fmmt_mlp_ln_fwd(code, M, C, x.data_ptr(), P["g"].data_ptr(), P["b"].data_ptr(), 1e-5, w1.data_ptr(), P["b1"].data_ptr(), w2.data_ptr(), P["b2"].data_ptr(),
                                 rs.data_ptr() if rs is not None else None, rps, y.data_ptr(), xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(), hp.data_ptr(), ha.data_ptr(), st())
        assert rc == 0
        res = [("y", y), ("xn", xn), ("h_pre", hp), ("h_act", ha), ("mean", mean), ("rstd", rstd)]
        if C == 96:                                          # the input gradient with the norm2 backward as its epilogue (C = 96 in production)
            dy = torch.randn(M, C, device=dev, generator=torch.Generator(device=dev).manual_seed(3)).bfloat16()
            dh, dx = torch.empty(M, 4 * C, dtype=x.dtype, device=dev), torch.empty_like(x)
            dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
            nb = lib.fmmt_mlp_ln_bwd_input_workspace(C)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            rc = lib.fmmt_mlp_ln_bwd_input(code, M, C, dy.data_ptr(), outs[0][2][1].data_ptr() if outs else hp.data_ptr(), w2t.data_ptr(), w1t.data_ptr(),
                                           rs.data_ptr() if rs is not None else None, rps, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), P["g"].data_ptr(),
                                           dh.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, st())
            assert rc == 0
            res += [("dh", dh), ("dx", dx), ("dgamma", dg), ("dbeta", db)]
        else:                                                # C = 192 in production: the input gradient without the LayerNorm epilogue (mlp_fused_bwd_kernel<192>)
            dy = torch.randn(M, C, device=dev, generator=torch.Generator(device=dev).manual_seed(3)).bfloat16()
            dh, dx = torch.empty(M, 4 * C, dtype=x.dtype, device=dev), torch.empty_like(x)
            rc = lib.fmmt_mlp_bwd_input(code, M, C, dy.data_ptr(), outs[0][2][1].data_ptr() if outs else hp.data_ptr(), w2t.data_ptr(), w1t.data_ptr(),
                                        rs.data_ptr() if rs is not None else None, rps, dh.data_ptr(), dx.data_ptr(), st())
            assert rc == 0
            res += [("dh", dh), ("dx", dx)]
        torch.cuda.synchronize()
        outs.append(res)
    for (name, a), (_, b) in zip(*outs):
        a, b = a.float(), b.float()
        scale = max(b.abs().max().item(), 1e-6)
        if a.dtype == torch.float32 and name in ("mean", "rstd", "dgamma", "dbeta"):
            assert (a - b).abs().max().item() <= 2e-4 * scale, name
        else:
            assert (a - b).abs().max().item() <= 2.0 ** -7 * scale and (a != b).float().mean().item() <= 0.02, (name, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("C,M,use_rs", [(96, 8192 + 77, True), (192, 4096 + 130, False), (96, 3136 * 2, False)])
def test_fp32_instantiation_of_the_fused_mlp_against_fp64(dev, C, M, use_rs, monkeypatch):
    """The fp32 instantiations of the same templates (what fmmt_mlp_ln_fwd / fmmt_mlp_ln_bwd_input / fmmt_mlp_fwd / fmmt_mlp_bwd_input run for
    dtype FMMT_F32; erf GELU, nothing rounded): the op x + s * Mlp(LayerNorm(x)) forward and every gradient against fp64 at 1e-3 or better,
    and the op without the LayerNorm (ops.mlp) likewise."""
    monkeypatch.setattr(ops, "_MLP_FUSED_WIDTHS", (96, 192))
    x, P, rs, rps = _mlp_operands(C, M, use_rs, torch.float32, 80)
    x.requires_grad_(True)
    leaves = [x] + [P[k].requires_grad_(True) for k in ("g", "b", "w1", "b1", "w2", "b2")]
    assert ops.mlp_ln_fusable(x, P["w1"], P["w2"], P["b1"], P["b2"])
    y = ops.mlp_ln(x, P["g"], P["b"], 1e-5, P["w1"], P["b1"], P["w2"], P["b2"], rs, rps)
    l64 = [t.detach().double().requires_grad_(True) for t in leaves]
    x64, g64, b64, w164, b164, w264, b264 = l64
    s64 = rs.double().repeat_interleave(rps)[:M, None] if rs is not None else 1.0
    r64 = x64 + s64 * (torch.nn.functional.gelu(torch.nn.functional.layer_norm(x64, (C,), g64, b64, 1e-5) @ w164.t() + b164) @ w264.t() + b264)
    assert (y.double() - r64).abs().max().item() <= 1e-4 * max(1.0, r64.abs().max().item())
    dy = torch.randn(M, C, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    ga = torch.autograd.grad(y, leaves, dy)
    gr = torch.autograd.grad(r64, l64, dy.double())
    for i, (a, c) in enumerate(zip(ga, gr)):
        assert _rel(a, c) <= 1e-3, i
    # without the LayerNorm: ops.mlp (fmmt_mlp_fwd / fmmt_mlp_bwd_input in fp32)
    res = torch.randn(M, C, device=dev, generator=torch.Generator(device=dev).manual_seed(6))
    y2 = ops.mlp(x, P["w1"], P["b1"], P["w2"], P["b2"], res=res, rowscale=rs, rows_per_scale=rps)
    r2 = res.double() + s64 * (torch.nn.functional.gelu(x64 @ w164.t() + b164) @ w264.t() + b264)
    assert (y2.double() - r2).abs().max().item() <= 1e-4 * max(1.0, r2.abs().max().item())
    g2 = torch.autograd.grad(y2, [x, P["w1"], P["b1"], P["w2"], P["b2"]], dy)
    gr2 = torch.autograd.grad(r2, [x64, w164, b164, w264, b264], dy.double())
    for i, (a, c) in enumerate(zip(g2, gr2)):
        assert _rel(a, c) <= 1e-3, i


def test_fused_mlp_input_gradient_launch(dev):
    """fmmt_mlp_bwd_input (dh and dx of the Mlp in one launch, C = 96) against the two GEMM launches it replaces (GELU' epilogue, then
    the input gradient): same products, same rounding points; gelu' comes from the LDS table in the fused kernel and from the table or
    the polynomial in the GEMM epilogue (by kernel), so dh agrees to one bf16 rounding step and dx to the accumulation of those;
    ragged token counts, a dropped sample, the bench size (2 M tokens).  Against fp64: test_mlp_half_* below."""
    import support_wblock_cases as W
    from facialmmt_amd._lib import EPI_GELU_BWD
    C = int(os.environ.get("MLP_BWD_TEST_C", "96"))
    for M, rps in ((4096 + 33, 1000), (256 * 40, 3136), (3136 * 640 * 96 // C, 3136 * 96 // C)):
        g = torch.Generator(device=dev).manual_seed(M)
        dy = torch.randn(M, C, device=dev, generator=g).bfloat16()
        hp = torch.randn(M, 4 * C, device=dev, generator=g).bfloat16()
        w1 = (torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5).bfloat16()
        w2 = (torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5).bfloat16()
        rs = torch.rand((M + rps - 1) // rps, device=dev, generator=g) + 0.5
        rs[0] = 0.0
        dh, dx = ops.mlp_bwd_input_raw(dy, hp, w1, w2, rs, rps)
        dh2 = ops.linear_raw(dy, w2.t().contiguous(), None, epi=EPI_GELU_BWD, aux=hp, rowscale=rs, rows_per_scale=rps)
        dx2 = ops.linear_raw(dh2, w1.t().contiguous(), None)
        for got, want, ulps in ((dh, dh2, 1.0), (dx, dx2, 2.0)):
            err, scale = (got.float() - want.float()).abs().max().item(), want.float().abs().max().item()
            assert err <= ulps * scale * 2.0 ** -7, (M, err, scale)
        assert (dh[:rps] == 0).all() and (dx[:rps] == 0).all()                                  # the dropped sample
        del dh, dx, dh2, dx2


@pytest.mark.parametrize("n", [2, 5])
def test_patch_embed_projection_and_layernorm_in_one_launch(dev, monkeypatch, golden, n):
    """fmmt_patch_embed_ln_fwd (PatchEmbed: proj + bias + LayerNorm, Swin_Transformer.py:392-422) in bf16: against the reference-generated
    golden (n = 2), against the two-launch form, and forward / every gradient against an fp64 restatement; ragged last tile (n = 5:
    15680 patches)."""
    from facialmmt_amd import synth
    from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as S
    pe = S.PatchEmbed(224, 4, 3, 96, torch.nn.LayerNorm)
    synth.fill_state_dict(pe, seed=30, prefix="pe.")
    pe.to(dev)
    frames = synth.tensor("frames", (n, 3, 224, 224), seed=1).to(dev).bfloat16().requires_grad_(True)
    params = list(pe.parameters())
    assert ops.patch_proj_ln_fusable(ops.patch_im2col(frames.detach()), pe.proj.weight.view(96, -1), pe.norm)
    y = pe(frames)
    if n == 2:
        import numpy as np
        ref, stride = golden.expected("swin_parts", "patch_embed")
        got = y.detach().float().cpu().numpy()
        got = got if stride is None else got.reshape(-1)[::stride]
        assert np.abs(got - ref).max() <= 3e-2 * np.abs(ref).max()
    w = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3)).bfloat16()
    g1 = torch.autograd.grad(y, [frames] + params, w)
    monkeypatch.setattr(ops, "_PATCH_LN", False)
    y2 = pe(frames)
    g2 = torch.autograd.grad(y2, [frames] + params, w)
    f64 = frames.detach().double().requires_grad_(True)
    p64 = [p.detach().double().requires_grad_(True) for p in params]
    named = dict(zip([k for k, _ in pe.named_parameters()], p64))
    x64 = torch.nn.functional.conv2d(f64, named["proj.weight"], named["proj.bias"], stride=4).flatten(2).transpose(1, 2)
    r64 = torch.nn.functional.layer_norm(x64, (96,), named["norm.weight"], named["norm.bias"], 1e-5)
    g64 = torch.autograd.grad(r64, [f64] + p64, w.double())
    assert _rel(y, r64) <= 2e-2 and _rel(y, y2) <= 1e-2
    for a, b, c in zip(g1, g2, g64):
        assert _rel(a, c) <= 4e-2 and _rel(b, c) <= 4e-2


def test_fp32_patch_embed_reaches_the_golden_through_the_fused_kernel(dev, golden, monkeypatch):
    """`patch_embed` (output of the reference's PatchEmbed, Swin_Transformer.py:392-422) at 1e-3 through the fp32 module, which must have
    been ONE call of fmmt_patch_embed_ln_fwd with dtype F32 (patch_embed_ln_kernel<float>: the production template, fp32 fragments); the
    two-launch fp32 form agrees with it; gradients of the fp32 instantiation against fp64 at 1e-4 (ragged last tile: 5 images)."""
    from facialmmt_amd import _lib, synth
    from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as S
    pe = S.PatchEmbed(224, 4, 3, 96, torch.nn.LayerNorm)
    synth.fill_state_dict(pe, seed=30, prefix="pe.")
    pe.to(dev)
    lib = _lib.load()
    calls = []
    real = lib.fmmt_patch_embed_ln_fwd
    monkeypatch.setattr(lib, "fmmt_patch_embed_ln_fwd", lambda *a: (calls.append(a[0]), real(*a))[1])
    frames = synth.tensor("frames", (2, 3, 224, 224), seed=1).to(dev)
    with torch.no_grad():
        y = pe(frames)
    assert calls == [_lib.F32]
    golden.check("swin_parts", "patch_embed", y, atol=1e-3, rtol=1e-3)
    monkeypatch.setattr(ops, "_PATCH_LN", False)
    with torch.no_grad():
        y2 = pe(frames)
    assert len(calls) == 1 and (y2 - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    monkeypatch.setattr(ops, "_PATCH_LN", True)
    f5 = synth.tensor("frames5", (5, 3, 224, 224), seed=2).to(dev).requires_grad_(True)
    params = list(pe.parameters())
    y5 = pe(f5)
    w = torch.randn(y5.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    g1 = torch.autograd.grad(y5, [f5] + params, w)
    assert calls == [_lib.F32, _lib.F32]
    f64 = f5.detach().double().requires_grad_(True)
    p64 = [p.detach().double().requires_grad_(True) for p in params]
    named = dict(zip([k for k, _ in pe.named_parameters()], p64))
    x64 = torch.nn.functional.conv2d(f64, named["proj.weight"], named["proj.bias"], stride=4).flatten(2).transpose(1, 2)
    r64 = torch.nn.functional.layer_norm(x64, (96,), named["norm.weight"], named["norm.bias"], 1e-5)
    g64 = torch.autograd.grad(r64, [f64] + p64, w.double())
    assert _rel(y5, r64) <= 1e-5
    for a, c in zip(g1, g64):
        assert _rel(a, c) <= 1e-4


@pytest.mark.parametrize("M,with_res", [(8192 + 77, True), (3136 * 2, False), (300, True)])
def test_linear_and_layernorm_backward_in_one_launch(dev, M, with_res):
    """fmmt_linear_ln_bwd (d(LN out) = dz . W, LayerNorm', residual gradient, d(gamma) / d(beta): the tail of the stage-0 attention
    half's backward, Swin_Transformer.py:239-243) against an fp64 restatement through autograd, and against the two launches it replaces
    (fmmt_linear_fwd on W^T + fmmt_layernorm_bwd, which round d(LN out) to bf16 in between); ragged tail and a tile-less M included."""
    C, K = 96, 288
    g = torch.Generator(device=dev).manual_seed(90 + M % 7)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = (rnd(M, C) * 1.5 + 0.3).bfloat16()
    dz = rnd(M, K).bfloat16()
    w = (rnd(K, C) * C ** -0.5).bfloat16()
    gamma = 1.0 + 0.2 * rnd(C)
    from facialmmt_amd import _lib
    dres = rnd(M, C).bfloat16() if with_res else None
    xf = x.float()
    mean = xf.mean(1)
    rstd = (xf.var(1, unbiased=False) + 1e-5).rsqrt()
    lib = _lib.load()
    dx = torch.empty_like(x)
    dg = torch.empty(C, device=dev)
    db = torch.empty(C, device=dev)
    nb = lib.fmmt_linear_ln_bwd_workspace(C)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    wt = w.t().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.fmmt_linear_ln_bwd(1, M, C, K, dz.data_ptr(), wt.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                dres.data_ptr() if with_res else None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, st)
    assert rc == 0
    # fp64: y = LN(x) * gamma (+ beta), contracted with the incoming gradient dz . w
    x64 = x.double().requires_grad_(True)
    g64 = gamma.double().requires_grad_(True)
    b64 = torch.zeros(C, dtype=torch.float64, device=dev, requires_grad=True)
    y64 = torch.nn.functional.layer_norm(x64, (C,), g64, b64, 1e-5)
    dxn64 = dz.double() @ w.double()
    gx, gg, gb = torch.autograd.grad(y64, [x64, g64, b64], dxn64)
    if with_res:
        gx = gx + dres.double()
    assert _rel(dx, gx) <= 2e-2 and _rel(dg, gg) <= 2e-2 and _rel(db, gb) <= 2e-2
    # the two launches
    dxn = ops.linear_raw(dz, wt, None)
    dx2 = torch.empty_like(x)
    dg2 = torch.empty(C, device=dev)
    db2 = torch.empty(C, device=dev)
    nb2 = lib.fmmt_layernorm_bwd_workspace(C)
    ws2 = torch.empty(nb2, dtype=torch.uint8, device=dev)
    rc = lib.fmmt_layernorm_bwd(1, M, C, dxn.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                dres.data_ptr() if with_res else None, dx2.data_ptr(), dg2.data_ptr(), db2.data_ptr(), 0, ws2.data_ptr(), nb2, st)
    assert rc == 0
    assert _rel(dx, dx2) <= 1.5e-2 and _rel(dg, dg2) <= 1e-2 and _rel(db, db2) <= 1e-2
    # shapes the kernel does not cover are refused, not mis-computed
    assert lib.fmmt_linear_ln_bwd(1, M, 192, 576, dz.data_ptr(), wt.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                  None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, st) == -1
    assert lib.fmmt_linear_ln_bwd(2, M, C, K, dz.data_ptr(), wt.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                  None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, st) == -1


@pytest.mark.parametrize("M,with_res", [(8192 + 77, True), (3136 * 2, False), (300, True)])
def test_linear_and_layernorm_backward_generic_instantiations(dev, M, with_res):
    """lin_lnbwd_ref_kernel -- fmmt_linear_ln_bwd's kernel restated over an element-type trait (same tiles, fragments, accumulator layout,
    LayerNorm' epilogue and d(gamma) / d(beta) reduction tree): the fp32 instantiation (8 x mfma_f32_16x16x4 per 32-deep block, nothing
    rounded) against fp64 autograd at 1e-4, and its bf16 instantiation (dtype FMMT_BF16 | FMMT_GENERIC) against the production kernel --
    which is what ties the production kernel to the fp32 parity evidence (Swin_Transformer.py:239-243)."""
    from facialmmt_amd import _lib
    C, K = 96, 288
    g = torch.Generator(device=dev).manual_seed(190 + M % 7)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = rnd(M, C) * 1.5 + 0.3
    dz = rnd(M, K)
    w = rnd(K, C) * C ** -0.5
    gamma = 1.0 + 0.2 * rnd(C)
    dres = rnd(M, C) if with_res else None
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.fmmt_linear_ln_bwd_workspace(C)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def run(code, dt):
        xx, zz, wt = x.to(dt), dz.to(dt), w.to(dt).t().contiguous()
        rr = dres.to(dt) if with_res else None
        xf = xx.float()
        mean = xf.mean(1)
        rstd = (xf.var(1, unbiased=False) + 1e-5).rsqrt()
        dx = torch.empty_like(xx)
        dg = torch.empty(C, device=dev)
        db = torch.empty(C, device=dev)
        rc = lib.fmmt_linear_ln_bwd(code, M, C, K, zz.data_ptr(), wt.data_ptr(), xx.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                    rr.data_ptr() if with_res else None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, st)
        assert rc == 0
        return dx, dg, db

    dx, dg, db = run(_lib.F32, torch.float32)
    x64 = x.double().requires_grad_(True)
    g64 = gamma.double().requires_grad_(True)
    b64 = torch.zeros(C, dtype=torch.float64, device=dev, requires_grad=True)
    y64 = torch.nn.functional.layer_norm(x64, (C,), g64, b64, 1e-5)
    gx, gg, gb = torch.autograd.grad(y64, [x64, g64, b64], dz.double() @ w.double())
    if with_res:
        gx = gx + dres.double()
    assert _rel(dx, gx) <= 1e-4 and _rel(dg, gg) <= 1e-4 and _rel(db, gb) <= 1e-4
    # bf16: generic instantiation == production kernel up to the accumulation order inside a 32-deep block (none here: same MFMA, same order)
    pa = run(_lib.BF16, torch.bfloat16)
    ga = run(_lib.BF16 | _lib.GENERIC, torch.bfloat16)
    same = (pa[0] == ga[0]).float().mean().item()
    step = (pa[0].float() - ga[0].float()).abs().max().item()
    assert same >= 0.98 and step <= 2.0 ** -6 * max(1.0, ga[0].float().abs().max().item()), (same, step)
    assert _rel(pa[1], ga[1]) <= 1e-4 and _rel(pa[2], ga[2]) <= 1e-4

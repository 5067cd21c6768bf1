"""`torch.ops.fmmt.*` (facialmmt_amd/torch_ops.py, torch.library custom operators) against the autograd.Function front end of
facialmmt_amd/ops.py: same C-ABI launches underneath, so forward and backward must be bit-identical; plus
torch.library.opcheck (schema, fake-tensor shapes, autograd registration) on each operator."""
import pytest
import torch

from facialmmt_amd import ops, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import facialmmt_amd.torch_ops  # noqa: F401  (registers the operators)
    return torch.device("cuda:0")


def _t(name, shape, dev, dtype=torch.float32, seed=0, scale=1.0, grad=False):
    t = (synth.tensor(name, shape, seed=seed) * scale).to(dev).to(dtype)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_op(dev, dtype):
    outs = []
    for front in ("function", "op"):
        x = _t("x", (3, 50, 96), dev, dtype, 1, grad=True)
        w = _t("w", (288, 96), dev, torch.float32, 2, 0.1, grad=True)
        b = _t("b", (288,), dev, torch.float32, 3, 0.1, grad=True)
        res = _t("r", (3, 50, 288), dev, dtype, 4, grad=True)
        rs = _t("s", (3,), dev, torch.float32, 5).abs() + 0.5
        y = ops.linear(x, w, b, res, rs, 50) if front == "function" else torch.ops.fmmt.linear(x, w, b, res, rs, 50)
        y.float().square().sum().backward()
        outs.append((y.detach(), x.grad, w.grad, b.grad, res.grad))
    for a, c in zip(*outs):
        assert torch.equal(a, c)


def test_mlp_layer_norm_window_attention_ops(dev):
    dt = torch.bfloat16
    # Mlp at a fused width (C = 96) and at a two-launch width (C = 384)
    for C, M in ((96, 8192), (384, 512)):
        outs = []
        for front in ("function", "op"):
            x = _t("x", (M, C), dev, dt, 1, grad=True)
            w1 = _t("w1", (4 * C, C), dev, torch.float32, 2, C ** -0.5, grad=True)
            b1 = _t("b1", (4 * C,), dev, torch.float32, 3, 0.1, grad=True)
            w2 = _t("w2", (C, 4 * C), dev, torch.float32, 4, (4 * C) ** -0.5, grad=True)
            b2 = _t("b2", (C,), dev, torch.float32, 5, 0.1, grad=True)
            res = _t("r", (M, C), dev, dt, 6, grad=True)
            y = ops.mlp(x, w1, b1, w2, b2, res, None, 1) if front == "function" else torch.ops.fmmt.mlp(x, w1, b1, w2, b2, res, None, 1)[0]
            y.float().square().sum().backward()
            outs.append((y.detach(), x.grad, w1.grad, b1.grad, w2.grad, b2.grad, res.grad))
        for a, c in zip(*outs):
            assert torch.equal(a, c), C
    outs = []
    for front in ("function", "op"):
        x = _t("x", (4, 49, 192), dev, dt, 1, grad=True)
        g = _t("g", (192,), dev, torch.float32, 2, 0.2, grad=True)
        b = _t("b", (192,), dev, torch.float32, 3, 0.1, grad=True)
        y = ops.layer_norm(x, g, b, 1e-5) if front == "function" else torch.ops.fmmt.layer_norm(x, g, b, 1e-5)[0]
        (y.float() * _t("p", (4, 49, 192), dev, seed=4)).sum().backward()
        outs.append((y.detach(), x.grad, g.grad, b.grad))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    from facialmmt_amd.modules.SwinTransformer.Swin_Transformer import WindowAttention, build_shift_mask
    wa = WindowAttention(96, (7, 7), 3)
    idx = wa.relative_position_index.to(device=dev, dtype=torch.int32).contiguous()
    for shift in (0, 3):
        outs = []
        for front in ("function", "op"):
            qkv = _t("qkv", (2 * 14 * 14, 288), dev, dt, 1, grad=True)
            tab = _t("tab", (169, 3), dev, torch.float32, 2, 0.5, grad=True)
            if front == "function":
                mask = build_shift_mask(14, 14, 7, shift).to(dev) if shift else None
                o = ops.window_attn_core(qkv, tab, idx, mask, 2, 14, 14, 3, shift, 32 ** -0.5, bool(shift))
            else:
                o = torch.ops.fmmt.window_attention(qkv, tab, idx, 2, 14, 14, 3, shift, 32 ** -0.5)[0]
            (o.float() * _t("p", (2 * 14 * 14, 96), dev, seed=3)).sum().backward()
            outs.append((o.detach(), qkv.grad, tab.grad))
        for a, c in zip(*outs):
            assert torch.equal(a, c), shift


def test_patch_embed_u8_op_and_cpu_rejection(dev):
    img = torch.randint(0, 256, (2, 112, 112, 3), dtype=torch.uint8)
    a = torch.ops.fmmt.patch_embed_u8(img.to(dev), "pil", True)
    assert torch.equal(a, ops.patch_embed_u8(img.to(dev), "pil", torch.bfloat16))
    with pytest.raises((NotImplementedError, RuntimeError)):        # no CPU kernel is registered: the dispatcher refuses
        torch.ops.fmmt.patch_embed_u8(img, "pil", True)


def test_opcheck(dev):
    x = _t("x", (64, 96), dev, torch.bfloat16, 1, grad=True)
    w = _t("w", (96, 96), dev, torch.float32, 2, 0.1, grad=True)
    b = _t("b", (96,), dev, torch.float32, 3, 0.1, grad=True)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.fmmt.linear.default, (x, w, b, None, None, 1), test_utils=tests)
    torch.library.opcheck(torch.ops.fmmt.layer_norm.default, (x, _t("g", (96,), dev, seed=4, grad=True), b, 1e-5), test_utils=tests)
    torch.library.opcheck(torch.ops.fmmt.patch_embed_u8.default, (torch.zeros(1, 112, 112, 3, dtype=torch.uint8, device=dev), "pil", True), test_utils=("test_schema", "test_faketensor"))


def test_remaining_ops_match_function_front_end(dev):
    """the operators registered in round 3 -- PatchMerging gather + LayerNorm, split-K head Linear, BatchNorm1d, cross-modal attention
    core (separate and packed k|v, dropout replay, key bias), position embedding, the fused attention half of a Swin block -- against
    the autograd.Function front end: the same C-ABI launches, so forward and every gradient must be bit-identical"""
    dt = torch.bfloat16

    def both(run):
        outs = [run("function"), run("op")]
        for a, c in zip(*outs):
            assert (a is None and c is None) or torch.equal(a, c)

    def merge_ln(front):
        x = _t("x", (2, 28 * 28, 192), dev, dt, 1, grad=True)
        g = _t("g", (768,), dev, torch.float32, 2, 0.2, grad=True)
        b = _t("b", (768,), dev, torch.float32, 3, 0.1, grad=True)
        y = ops.layer_norm(x, g, b, 1e-5, merge_hw=28) if front == "function" else torch.ops.fmmt.layer_norm_merge(x, g, b, 1e-5, 28)[0]
        (y.float() * _t("p", (2, 196, 768), dev, seed=4)).sum().backward()
        return y.detach(), x.grad, g.grad, b.grad
    both(merge_ln)

    def head(front):
        x = _t("x", (8, 37632), dev, dt, 1, 0.1, grad=True)
        w = _t("w", (512, 37632), dev, torch.float32, 2, 0.01, grad=True)
        b = _t("b", (512,), dev, torch.float32, 3, 0.1, grad=True)
        y = ops.linear(x, w, b) if front == "function" else torch.ops.fmmt.linear_splitk(x, w, b)
        y.float().square().sum().backward()
        return y.detach(), x.grad, w.grad, b.grad
    both(head)

    def bn(front):
        x = _t("x", (8, 512), dev, dt, 1, grad=True)
        g = _t("g", (512,), dev, torch.float32, 2, 0.2, grad=True)
        b = _t("b", (512,), dev, torch.float32, 3, 0.1, grad=True)
        rm, rv = torch.zeros(512, device=dev), torch.ones(512, device=dev)
        if front == "function":
            y = ops.batch_norm_1d(x, g, b, rm, rv, 0.1, 1e-5, True)
        else:
            y, _, _, rm2, rv2 = torch.ops.fmmt.batch_norm_1d(x, g, b, rm, rv, 0.1, 1e-5, True)
            assert rm.abs().max() == 0                      # functional: the inputs are untouched, the new statistics are outputs
            rm, rv = rm2, rv2
        (y.float() * _t("p", (8, 512), dev, seed=4)).sum().backward()
        return y.detach(), x.grad, g.grad, b.grad, rm, rv
    both(bn)

    for packed in (False, True):
        def mha(front):
            q = _t("q", (38, 2, 768), dev, dt, 1, grad=True)
            if packed:
                k, v = _t("kv", (128, 2, 1536), dev, dt, 2, grad=True), None
            else:
                k, v = _t("k", (128, 2, 768), dev, dt, 2, grad=True), _t("v", (128, 2, 768), dev, dt, 3, grad=True)
            kb = _t("kb", (2, 128), dev, torch.float32, 5)
            o = ops.mha_core(q, k, v, 12, 64 ** -0.5, 0.1, 1234, kb) if front == "function" else torch.ops.fmmt.mha(q, k, v, kb, 12, 64 ** -0.5, 0.1, 1234)[0]
            (o.float() * _t("p", (38, 2, 768), dev, seed=4)).sum().backward()
            return o.detach(), q.grad, k.grad, (v.grad if v is not None else None)
        both(mha)

    def pe(front):
        from facialmmt_amd.modules.position_embedding import SinusoidalPositionalEmbedding
        emb = SinusoidalPositionalEmbedding(768)
        x = _t("x", (20, 3, 768), dev, dt, 1)
        x[3, 1] = 0
        x = x.requires_grad_(True)
        table = emb.table(20, dev)
        y = ops.posemb_scale(x, table, 768 ** 0.5) if front == "function" else torch.ops.fmmt.posemb_scale(x, table, 768 ** 0.5)
        (y.float() * _t("p", (20, 3, 768), dev, seed=4)).sum().backward()
        return y.detach(), x.grad
    both(pe)

    from facialmmt_amd.modules.SwinTransformer.Swin_Transformer import WindowAttention, build_shift_mask
    idx = WindowAttention(96, (7, 7), 3).relative_position_index.to(device=dev, dtype=torch.int32).contiguous()
    for shift in (0, 3):
        def wb(front):
            x = _t("x", (2, 196, 96), dev, dt, 1, grad=True)
            P = [_t("g", (96,), dev, seed=2, scale=0.2, grad=True), _t("b", (96,), dev, seed=3, scale=0.1, grad=True), _t("wq", (288, 96), dev, seed=4, scale=0.1, grad=True),
                 _t("bq", (288,), dev, seed=5, scale=0.1, grad=True), _t("wp", (96, 96), dev, seed=6, scale=0.1, grad=True), _t("bp", (96,), dev, seed=7, scale=0.1, grad=True),
                 _t("tab", (169, 3), dev, seed=8, scale=0.5, grad=True)]
            rs = _t("s", (2,), dev, seed=9).abs() + 0.5
            if front == "function":
                mask = build_shift_mask(14, 14, 7, shift).to(dev) if shift else None
                y = ops.window_block(x, P[0], P[1], 1e-5, P[2], P[3], P[4], P[5], P[6], idx, mask, 2, 14, 14, 3, shift, 32 ** -0.5, rs)
            else:
                y = torch.ops.fmmt.window_block(x, P[0], P[1], 1e-5, P[2], P[3], P[4], P[5], P[6], idx, 2, 14, 14, 3, shift, 32 ** -0.5, rs)[0]
            (y.float() * _t("p", (2, 196, 96), dev, seed=10)).sum().backward()
            return (y.detach(), x.grad) + tuple(p.grad for p in P)
        both(wb)


def test_opcheck_remaining(dev):
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    dt = torch.bfloat16
    g768 = _t("g", (768,), dev, seed=2, grad=True)
    torch.library.opcheck(torch.ops.fmmt.layer_norm_merge.default, (_t("x", (1, 196, 192), dev, dt, 1, grad=True), g768, _t("b", (768,), dev, seed=3, grad=True), 1e-5, 14), test_utils=tests)
    torch.library.opcheck(torch.ops.fmmt.linear_splitk.default, (_t("x", (4, 4704), dev, dt, 1, grad=True), _t("w", (512, 4704), dev, seed=2, scale=0.02, grad=True), None), test_utils=tests)
    torch.library.opcheck(torch.ops.fmmt.batch_norm_1d.default, (_t("x", (8, 512), dev, dt, 1, grad=True), _t("g", (512,), dev, seed=2, grad=True), _t("b", (512,), dev, seed=3, grad=True),
                                                                 torch.zeros(512, device=dev), torch.ones(512, device=dev), 0.1, 1e-5, True), test_utils=tests)
    torch.library.opcheck(torch.ops.fmmt.mha.default, (_t("q", (38, 2, 768), dev, dt, 1, grad=True), _t("k", (64, 2, 768), dev, dt, 2, grad=True), _t("v", (64, 2, 768), dev, dt, 3, grad=True),
                                                       None, 12, 0.125, 0.0, 0), test_utils=tests)
    torch.library.opcheck(torch.ops.fmmt.posemb_scale.default, (_t("x", (20, 3, 768), dev, dt, 1, grad=True), _t("t", (64, 768), dev, seed=2), 27.7), test_utils=tests)
    from facialmmt_amd.modules.SwinTransformer.Swin_Transformer import WindowAttention
    idx = WindowAttention(96, (7, 7), 3).relative_position_index.to(device=dev, dtype=torch.int32).contiguous()
    torch.library.opcheck(torch.ops.fmmt.window_block.default,
                          (_t("x", (1, 49, 96), dev, dt, 1, grad=True), _t("g", (96,), dev, seed=2, grad=True), _t("b", (96,), dev, seed=3, grad=True), 1e-5,
                           _t("wq", (288, 96), dev, seed=4, scale=0.1, grad=True), None, _t("wp", (96, 96), dev, seed=6, scale=0.1, grad=True), None,
                           _t("tab", (169, 3), dev, seed=8, grad=True), idx, 1, 7, 7, 3, 0, 32 ** -0.5, None), test_utils=tests)

"""GPU tests of the rows either side of the hot path:

* the device-side frame filter (train_step.select_frames) and target-utterance slicing (models.slice_target_utterance)
  on DEVICE tensors against the literal loop restatements in oracle/ (train.py:75-114, src/models.py:112-150), including the
  multi-utterance `num_imgs - 1` boundary quirk, the branch where no face passes the threshold, ragged utterances;
* checkpoint I/O: a FaceX-Zoo style file -> checkpoint.load_pretrained_backbone -> the loaded model reproduces the
  reference's golden features on the GPU; save_state / load_state round trip; the converted whole-module pickle;
* eval() after HIP-graph capture runs the eager branches (the graphs hold training-mode dropout and static shapes)."""
import os
import sys
import types

import pytest
import torch

from facialmmt_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _filter_case(g, B, Lv, nmax, peaky, D=16):
    num = torch.randint(max(1, nmax // 2), nmax + 1, (B,), generator=g)
    nF = int(num.sum())
    preds = torch.softmax(torch.randn(nF, 7, generator=g) * (3.0 if peaky else 0.3), -1)
    vis = torch.randn(B, Lv, D, generator=g)
    mask = (torch.arange(Lv).view(1, Lv) < num.view(B, 1)).float()
    return preds, vis, mask, num


def test_select_frames_on_device_matches_literal_loop(dev):
    from facialmmt_amd.train_step import select_frames
    from oracle.train_glue import select_frames_loop
    g = torch.Generator().manual_seed(0)
    branches = {"filtered": 0, "none_pass": 0, "multi_utt": 0}
    for trial in range(80):
        B = int(torch.randint(1, 6, (1,), generator=g))
        preds, vis, mask, num = _filter_case(g, B, 12, 12, peaky=(trial % 3 != 0))
        thr = [0.2, 0.5, 0.99, 0.0, 1.5][trial % 5]            # 1.5 > max sum(p^2): the no-face-passes branch
        want, want_mask = select_frames_loop(preds, vis, mask, num.tolist(), thr)
        got, got_mask = select_frames(preds.to(dev), vis.to(dev), mask.to(dev), num.to(dev), thr)
        assert got.is_cuda and got_mask.is_cuda
        assert torch.equal(got_mask.cpu(), want_mask), (trial, got_mask.cpu(), want_mask)
        assert torch.equal(got.cpu(), want), trial               # gathers and exact copies: bit-identical
        any_pass = bool(((preds * preds).sum(1) > thr).any())
        branches["filtered" if any_pass else "none_pass"] += 1
        branches["multi_utt"] += int(B > 1 and any_pass)
    assert all(v >= 10 for v in branches.values()), branches


def test_select_frames_full_size_properties(dev):
    """bench size (4 utterances x 160 frames, 512-d features): threshold 0 keeps every face -- for one utterance the
    result is the input with the emotion features appended; the packed prefix property holds at any threshold."""
    from facialmmt_amd.train_step import select_frames
    g = torch.Generator().manual_seed(3)
    preds, vis, mask, num = _filter_case(g, 1, 160, 160, peaky=True, D=512)
    out, m = select_frames(preds.to(dev), vis.to(dev), mask.to(dev), num.to(dev), 0.0)
    n = int(num[0])
    assert torch.equal(m.cpu(), mask)
    assert torch.equal(out[0, :n, :512].cpu(), vis[0, :n]) and torch.equal(out[0, :n, 512:].cpu(), preds[:n])
    preds, vis, mask, num = _filter_case(g, 4, 160, 160, peaky=True, D=512)
    out, m = select_frames(preds.to(dev), vis.to(dev), mask.to(dev), num.to(dev), 0.2)
    m = m.cpu()
    assert torch.equal(m, (torch.cumsum(m, 1) == torch.arange(1, 161).view(1, -1)).float() * m)     # ones form a prefix
    assert bool((out.cpu()[m == 0] == 0).all())                                                    # nothing behind the prefix


def test_slice_target_utterance_on_device_matches_literal_loop(dev):
    from facialmmt_amd.models import slice_target_utterance
    from oracle.multimodal import slice_target_utterance_loop
    g = torch.Generator().manual_seed(0)
    for trial in range(20):
        B, T, H = 5, 60, 8
        feats = torch.randn(B, T, H, generator=g)
        sep = torch.zeros(B, T)
        for i in range(B):
            pos = 1
            while True:
                pos += int(torch.randint(3, 14, (1,), generator=g))
                if pos >= T:
                    break
                sep[i, pos] = 1
        utt = torch.randint(0, 5, (B,), generator=g)
        for roberta in (True, False):
            a, am = slice_target_utterance(feats.to(dev), sep.to(dev), utt.to(dev), 7, roberta)
            b, bm = slice_target_utterance_loop(feats, sep, utt, 7, roberta)
            assert a.is_cuda and torch.equal(a.cpu(), b) and torch.equal(am.cpu(), bm)
    # bench geometry: 512 tokens, 1024-wide features, separators every 20 tokens, utterance index up to 7, 38 kept
    feats = torch.randn(4, 512, 64, generator=g)
    sep = torch.zeros(4, 512)
    sep[:, 20:400:20] = 1
    utt = torch.tensor([0, 1, 5, 7])
    for roberta in (True, False):
        a, am = slice_target_utterance(feats.to(dev), sep.to(dev), utt.to(dev), 38, roberta)
        b, bm = slice_target_utterance_loop(feats, sep, utt, 38, roberta)
        assert torch.equal(a.cpu(), b) and torch.equal(am.cpu(), bm)


def _facex_zoo_file(path, seed=100):
    """A FaceX-Zoo style checkpoint ({'state_dict': {'backbone.<name>': tensor}}, train.py:316-331) holding the weights
    the golden fixtures were generated with; the head (`linear`, `classifier`) is absent, as in the published file."""
    from facialmmt_amd.config import default_args
    from facialmmt_amd.modules.SwinTransformer.backbone_def import BackboneFactory
    a = default_args()
    donor = BackboneFactory(a.backbone_type, a.backbone_conf_file).get_backbone()
    synth.fill_state_dict(donor, seed=seed)            # the bare backbone: the weights `swin_eval_n8` was generated with
    sd = {"backbone." + k: v.clone() for k, v in donor.state_dict().items()}
    sd["head.weight"] = torch.zeros(10, 512)                                  # something nothing asks for
    torch.save({"state_dict": sd, "epoch": 17}, path)
    return donor


def test_loaded_backbone_reproduces_reference_features(golden, dev, tmp_path):
    from facialmmt_amd import checkpoint, models
    from facialmmt_amd.config import default_args
    path = str(tmp_path / "Swin_tiny_Ms-Celeb-1M.pt")
    _facex_zoo_file(path)
    torch.manual_seed(7)
    aff = models.SwinForAffwildClassification(default_args())               # random init, different from the file
    rep = checkpoint.load_pretrained_backbone(aff, path)
    assert len(rep.loaded) == 195 and rep.unused == ["head.weight"] and not rep.mismatched
    assert sorted(rep.missing) == ["linear.bias", "linear.weight"]
    aff.to(dev).eval()
    frames = synth.tensor("frames", (8, 3, 224, 224), seed=1).to(dev)
    with torch.no_grad():
        golden.check("swin_full", "swin_eval_n8", aff.swin(frames), atol=1e-3, rtol=1e-3)
    # save_state -> load_state round trip into a third model: same features, bit for bit
    out = str(tmp_path / "best_swin.state.pt")
    checkpoint.save_state(aff, out, extra={"val_f1": 0.5})
    other = models.SwinForAffwildClassification(default_args())
    checkpoint.load_state(other, out)
    other.to(dev).eval()
    with torch.no_grad():
        assert torch.equal(other.swin(frames), aff.swin(frames))


def test_converted_whole_module_pickle_loads_and_matches(golden, dev, tmp_path):
    """tools/convert_checkpoint.py on a whole-module pickle of the kind utils/util.py:121-133 writes (here: this package's
    own model class wrapped like LightningLite's _LiteModule) -> plain file -> load_state -> golden features."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convert_checkpoint
    from facialmmt_amd import checkpoint, models
    from facialmmt_amd.config import default_args
    from tests.pickle_fixture import _LiteModule
    donor = models.SwinForAffwildClassification(default_args())
    synth.fill_state_dict(donor.swin, seed=100)        # the bare backbone's names: the weights `swin_eval_n8` was generated with
    src = str(tmp_path / "best_swin_09-28.pt")
    torch.save(_LiteModule(donor), src, pickle_protocol=4)
    out = str(tmp_path / "best_swin.state.pt")
    rep = convert_checkpoint.convert(src, out, reference_root=None, expect_class="SwinForAffwildClassification")
    assert rep["wrappers"] == ["_LiteModule"] and rep["n_tensors"] == len(donor.state_dict())
    m = models.SwinForAffwildClassification(default_args())
    checkpoint.load_state(m, out)
    m.to(dev).eval()
    frames = synth.tensor("frames", (8, 3, 224, 224), seed=1).to(dev)
    with torch.no_grad():
        golden.check("swin_full", "swin_eval_n8", m.swin(frames), atol=1e-3, rtol=1e-3)


def test_eval_after_graph_capture_runs_eager_branches(dev):
    """after graph_multimodal, mm.eval() must not replay the training-mode graphs (dropout, static shapes)."""
    import bench
    from facialmmt_amd import models
    from facialmmt_amd.config import default_args
    from facialmmt_amd.train_step import graph_multimodal
    B, Lv = 2, 6
    cfg = default_args(get_vision_utt_max_lens=Lv, get_audio_utt_max_lens=24, plm_module=synth.make_standin_plm())
    cfg.compute_dtype = torch.float32
    mm = models.MultiModalTransformerForClassification(cfg)
    synth.fill_state_dict(mm, seed=200)
    mm.to(dev)
    args = types.SimpleNamespace(utts=B, frames=Lv, dtype="fp32")
    batch = list(bench.synth_batch(args, dev, 0, cfg))
    batch[0] = batch[0] % 1000
    vis = torch.cat((batch[5], torch.softmax(torch.randn(B, Lv, 7, device=dev), -1)), -1)
    call = (batch[0], batch[1], batch[2], batch[3], batch[4], vis, batch[6], batch[10])
    mm.eval()
    with torch.no_grad():
        want = mm(*call).clone()
        want3 = mm(*(t[:1] if torch.is_tensor(t) else t for t in call)).clone()      # a batch of another size
    mm.train()
    sample = tuple(t.detach().clone().requires_grad_(True) if i == 5 else t for i, t in enumerate(call))
    mm = graph_multimodal(mm, sample, None, overlap_text=True)
    with torch.no_grad():
        noisy = mm(*call)                                        # training mode: dropout (p = 0.1) is active in the replay
    mm.eval()
    with torch.no_grad():
        got = mm(*call)
        got3 = mm(*(t[:1] if torch.is_tensor(t) else t for t in call))
    assert torch.equal(got, want) and torch.equal(got3, want3)
    assert not torch.equal(noisy, want)
    mm.train()
    with torch.no_grad():
        assert mm(*call).shape == want.shape                     # and the graphs are used again in training mode


@pytest.mark.gpu
def test_colsum_and_vendor_linear_bias_gradient(dev):
    """fmmt_colsum against torch.sum (ragged row counts, bf16 -> bf16 / fp32, fp32), and train_step.VendorLinear: forward, dx and dW
    are the library calls autograd makes (bit-identical to nn.Linear), db is the column sum (same fp32 accumulation, one rounding)"""
    from facialmmt_amd import ops
    from facialmmt_amd.train_step import VendorLinear, use_colsum_bias_gradients
    g = torch.Generator(device="cpu").manual_seed(5)
    for (M, N, dt) in [(2048, 1024, torch.bfloat16), (2048, 4096, torch.bfloat16), (777, 1032, torch.bfloat16), (63, 8, torch.bfloat16),
                       (1000, 516, torch.float32)]:
        x = torch.randn(M, N, generator=g).to(dev, dt)
        ref = x.float().sum(0)
        for od in {dt, torch.float32}:
            out = ops.colsum_raw(x, od)
            tol = 2e-2 if od == torch.bfloat16 else 1e-4
            assert out.dtype == od and torch.allclose(out.float(), ref, rtol=tol, atol=tol * ref.abs().max().item()), (M, N, dt, od)
    seq = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.GELU(), torch.nn.Linear(512, 256)).to(dev, torch.bfloat16)
    import copy
    fast = copy.deepcopy(seq)
    assert use_colsum_bias_gradients(fast) == 2 and isinstance(fast[0], VendorLinear) and list(fast.state_dict()) == list(seq.state_dict())
    x = torch.randn(4, 300, 256, generator=g).to(dev, torch.bfloat16)
    outs = []
    for m in (seq, fast):
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        y.square().mean().backward()
        outs.append((y, xi.grad, [p.grad for p in m.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for (n, _), a, b in zip(seq.named_parameters(), outs[0][2], outs[1][2]):
        if n.endswith("weight"):
            assert torch.equal(a, b), n
        else:
            assert torch.allclose(a.float(), b.float(), rtol=2e-2, atol=1e-2 * a.float().abs().max().item()), n


@pytest.mark.gpu
def test_cast_batch_and_pinned_shadows(dev):
    """fmmt_cast_batch through ops.PinnedShadows: every (view, transpose) shadow the eager cache knows -- full matrices, row
    slices of a fused in_proj weight, odd sizes (7 x 768 head, K = 48 patch embedding), fp32 and bf16 sources -- equals
    w.to(bf16)[.t()] after refresh(), tracks in-place parameter updates, and is handed out only while a graph is captured"""
    from facialmmt_amd import ops
    g = torch.Generator(device="cpu").manual_seed(9)
    ps = [torch.nn.Parameter(torch.randn(sh, generator=g).to(dev, dt)) for sh, dt in
          [((96, 48), torch.float32), ((7, 768), torch.float32), ((2304, 768), torch.float32), ((130, 70), torch.float32), ((192, 96), torch.bfloat16)]]
    views = [ps[2][:768], ps[2][768:]]                       # in_proj-style row slices of one parameter
    for w in ps + views:
        ops._lp(w, torch.bfloat16)
        ops._lp(w, torch.bfloat16, transpose=True)
    pin = ops.PinnedShadows(ps)
    assert pin.n == 2 * 6 + 1                                # the bf16 parameter needs no plain shadow, only the transposed one
    for rnd in range(2):
        pin.refresh()
        torch.cuda.synchronize()
        for w in ps + views:
            base = w._base if w._base is not None else w
            for tp in (False, True):
                key = (w.storage_offset(), tuple(w.shape), tuple(w.stride()), torch.bfloat16, tp)
                if w.dtype == torch.bfloat16 and not tp:
                    continue
                ref = w.detach().to(torch.bfloat16)
                assert torch.equal(ops._PINNED[id(base)][key], ref.t().contiguous() if tp else ref), (tuple(w.shape), tp, rnd)
        with torch.no_grad():
            for p in ps:
                p.mul_(1.5).add_(0.25)
    assert ops._lp(ps[0], torch.bfloat16) is not ops._PINNED[id(ps[0])][(0, (96, 48), (48, 1), torch.bfloat16, False)]
    pin.release()


@pytest.mark.gpu
def test_bf16_layernorm_backward_for_the_text_encoder(dev):
    """fmmt_layernorm_bwd_bf16 behind train_step.VendorLayerNorm against torch's own LayerNorm backward on the same bf16 module
    (and both against an fp32 evaluation): hidden sizes 1024 / 768 / 200, ragged row counts; same keys, same forward bits."""
    from facialmmt_amd.train_step import VendorLayerNorm
    g = torch.Generator(device="cpu").manual_seed(11)
    for (rows, C) in [((4, 512), 1024), ((3, 77), 768), ((1, 5), 200), ((2049,), 1024)]:
        ref = torch.nn.LayerNorm(C, eps=1e-5).to(dev, torch.bfloat16)
        with torch.no_grad():
            ref.weight.copy_(torch.randn(C, generator=g).to(dev) * 0.2 + 1.0)
            ref.bias.copy_(torch.randn(C, generator=g).to(dev) * 0.1)
        import copy
        fast = copy.deepcopy(ref)
        fast.__class__ = VendorLayerNorm
        assert list(fast.state_dict()) == list(ref.state_dict())
        x = (torch.randn(*rows, C, generator=g) * 2.0 + 0.5).to(dev, torch.bfloat16)
        dy = torch.randn(*rows, C, generator=g).to(dev, torch.bfloat16)
        outs = []
        for m in (ref, fast):
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            y.backward(dy)
            outs.append((y, xi.grad, m.weight.grad, m.bias.grad))
        x32 = x.float().requires_grad_(True)
        y32 = torch.nn.functional.layer_norm(x32, (C,), ref.weight.float(), ref.bias.float(), 1e-5)
        gw32, gb32 = (dy.float() * ((x32 - x32.mean(-1, keepdim=True)) * torch.rsqrt(x32.var(-1, unbiased=False, keepdim=True) + 1e-5))).reshape(-1, C).sum(0), dy.float().reshape(-1, C).sum(0)
        y32.backward(dy.float())
        assert torch.equal(outs[0][0], outs[1][0])
        for got, want in ((outs[1][1], x32.grad), (outs[1][2], gw32), (outs[1][3], gb32)):
            scale = want.abs().max().item()
            assert (got.float() - want.detach()).abs().max().item() <= 1.5e-2 * scale, (rows, C)
        # no worse than torch's own bf16 backward against the fp32 evaluation
        e_ref = (outs[0][1].float() - x32.grad).abs().max().item()
        e_new = (outs[1][1].float() - x32.grad).abs().max().item()
        assert e_new <= 1.5 * e_ref + 1e-3 * x32.grad.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("accumulate", [False, True])
def test_gradient_hand_over_and_clip_norm_in_one_pass(dev, accumulate):
    """train_step.FusedHandOver (fmmt_grad_handover) against _hand_over_gradients + get_total_norm (train.py:135-140): fp32 and bf16 gradients,
    sizes off the 4096-element block and the 4-element vector (1, 5, 4096, 4097, 70001), overwrite and accumulate; the slots bit for bit, the
    norm to fp32 summation order; a missing gradient makes it decline without touching anything; two calls give bit-identical results."""
    from facialmmt_amd import train_step as TS
    g = torch.Generator(device="cpu").manual_seed(17)
    shapes = [((1,), torch.float32), ((5,), torch.bfloat16), ((64, 64), torch.float32), ((4097,), torch.bfloat16), ((70001,), torch.float32), ((130, 70), torch.bfloat16)]
    params = [torch.nn.Parameter(torch.zeros(sh, device=dev, dtype=dt)) for sh, dt in shapes]
    grads = [torch.randn(sh, generator=g).to(dev, dt) for sh, dt in shapes]
    def slots(fill):
        off, views = 0, {}
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        flat = torch.full((n,), 0.0, device=dev)
        for p in params:
            views[p] = flat[off:off + p.numel()].view_as(p)
            views[p].copy_(fill[id(p)])
            off += (p.numel() + 3) // 4 * 4
        return flat, views
    old = {id(p): torch.randn(p.shape, generator=g).to(dev) if accumulate else torch.zeros(p.shape, device=dev) for p in params}
    pairs = [(p, p) for p in params]
    # two-pass reference
    flat_r, views_r = slots(old)
    for p, gr in zip(params, grads):
        p.grad = gr.clone()
    TS._hand_over_gradients(pairs, views_r, accumulate)
    norm_r = torch.nn.utils.get_total_norm([views_r[p] for p in params], 2.0)
    outs = []
    for rep in range(2):
        flat_f, views_f = slots(old)
        for p, gr in zip(params, grads):
            p.grad = gr.clone()
        ho = TS.FusedHandOver(len(pairs))
        norm = torch.zeros((), device=dev)
        assert ho(pairs, views_f, accumulate, norm)
        torch.cuda.synchronize()
        assert all(p.grad is None for p in params)
        assert torch.equal(flat_f, flat_r)
        assert abs(norm.item() - norm_r.item()) <= 1e-5 * norm_r.item()
        outs.append(norm.clone())
    assert torch.equal(outs[0], outs[1])
    # a parameter without a gradient: declined, nothing touched
    flat_f, views_f = slots(old)
    for p, gr in zip(params, grads):
        p.grad = gr.clone()
    params[2].grad = None
    before = flat_f.clone()
    assert not TS.FusedHandOver(len(pairs))(pairs, views_f, accumulate, torch.zeros((), device=dev))
    assert torch.equal(flat_f, before) and params[0].grad is not None


def _small_roberta(dev, p_drop):
    from transformers import RobertaConfig, RobertaModel
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=1000, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, max_position_embeddings=80,
                        hidden_dropout_prob=p_drop, attention_probs_dropout_prob=0.0)
    m = RobertaModel(cfg, add_pooling_layer=False)
    return m.to(dev).to(torch.bfloat16).train()


def test_fused_text_encoder_sublayers_match_the_stock_module():
    """train_step.fuse_text_encoder: (a) *SelfOutput / *Output as the vendor GEMM + fmmt_plm_dropadd_ln_fwd / _bwd, (b) query / key / value as one GEMM
    over a packed weight whose row slices ARE the three nn.Linear parameters -- same outputs and the same gradients for every parameter as the stock
    bf16 module (dropout off: its stream is the only thing that differs), same state_dict keys, and an optimizer step through the sliced parameters
    moves the packed weight."""
    import copy
    from facialmmt_amd.train_step import fuse_text_encoder
    dev = torch.device("cuda:0")
    stock = _small_roberta(dev, 0.0)
    fused = copy.deepcopy(stock)
    assert fuse_text_encoder(fused) == (4, 2) and fused._fmmt_fused_ffn == 2
    assert list(fused.state_dict()) == list(stock.state_dict())
    ids = torch.randint(3, 1000, (3, 64), device=dev)
    mask = torch.ones(3, 64, device=dev, dtype=torch.long)
    mask[1, 40:] = 0
    g = torch.randn(3, 64, 256, device=dev, dtype=torch.bfloat16)
    outs = []
    for m in (stock, fused):
        y = m(input_ids=ids, attention_mask=mask).last_hidden_state
        y.backward(g)
        outs.append((y.detach().float(), {k: p.grad.detach().float() for k, p in m.named_parameters() if p.grad is not None}))
    (y0, g0), (y1, g1) = outs
    assert (y0 - y1).abs().max().item() <= 3e-2 * max(1.0, y0.abs().max().item())
    assert set(g0) == set(g1)
    for k in g0:
        if k.endswith("key.bias"):                               # mathematically zero (a key bias shifts every score of a row alike): rounding noise on both sides
            continue
        scale = g0[k].abs().max().item() + 1e-6
        assert (g0[k] - g1[k]).abs().max().item() <= 4e-2 * scale, (k, (g0[k] - g1[k]).abs().max().item(), scale)
    att = fused.encoder.layer[0].attention.self
    w, _ = att._fmmt_qkv
    before = w.clone()
    torch.optim.SGD(fused.parameters(), lr=0.1).step()
    assert not torch.equal(before, w) and att.key.weight.data_ptr() == w[256:].data_ptr()


def test_embedding_weight_gradient_against_torch():
    """fmmt_embedding_bwd (ops.PlmEmbeddingFn): the dense weight gradient of nn.Embedding for a few thousand indices against torch's own backward in fp32 -- a large
    table with repeated ids (one of them ~100 times) and a padding index, a position-like table (every id B times), tables of one and two rows (the predicated
    column-sum path); rows no token names are zero; two calls agree bit for bit."""
    from facialmmt_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    for V, C, T, pad in ((50265, 1024, 2048, 1), (514, 1024, 2048, 1), (1, 1024, 2048, None), (2, 768, 700, None), (30522, 768, 33, 0)):
        if V > 8:
            ids = torch.randint(0, V, (T,), device=dev)
            ids[::20] = 2                                              # a separator every 20 tokens
            ids[5:9] = pad if pad is not None else 3
            if V == 514:
                ids = (torch.arange(T, device=dev) % 512) + 2         # positions: every id T / 512 times
        else:
            ids = torch.randint(0, V, (T,), device=dev)
        ids = ids.view(4, -1) if T % 4 == 0 else ids.view(1, -1)
        w = torch.randn(V, C, device=dev).to(torch.bfloat16).requires_grad_(True)
        dy = torch.randn(*ids.shape, C, device=dev).to(torch.bfloat16)
        y = ops.PlmEmbeddingFn.apply(ids, w, pad)
        assert torch.equal(y, torch.nn.functional.embedding(ids, w, pad))
        (g,) = torch.autograd.grad(y, w, dy)
        (g2,) = torch.autograd.grad(ops.PlmEmbeddingFn.apply(ids, w, pad), w, dy)
        assert torch.equal(g, g2)
        wr = w.detach().float().requires_grad_(True)
        (gr,) = torch.autograd.grad(torch.nn.functional.embedding(ids, wr, pad), wr, dy.float())
        assert (g.float() - gr).abs().max().item() <= 8e-3 * gr.abs().max().item() + 1e-6, (V, C, T)
        assert torch.equal(g.float() == 0, gr == 0) or ((g.float() != 0) & (gr == 0)).sum().item() == 0
        if pad is not None:
            assert (g[pad] == 0).all()


def test_gelu_backward_with_the_bias_gradient_in_one_pass():
    """fmmt_plm_gelu_bwd_colsum (the text encoder's *Intermediate backward): dpre = dact * gelu'(pre) against torch's erf GELU backward in fp32 within one bf16
    step (+ 2e-4 |dact| around the derivative's zero), dbias = the column sum of the STORED dpre (fixed order: two calls agree bit for bit); ragged row counts, the widest row the kernel takes."""
    from facialmmt_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    for M, H in ((2048, 4096), (77, 512), (301, 8192), (5, 8)):
        pre = (3.0 * torch.randn(M, H, device=dev)).to(torch.bfloat16)
        dact = torch.randn(M, H, device=dev).to(torch.bfloat16)
        dpre, db = ops.plm_gelu_bwd_colsum_raw(dact, pre)
        dpre2, db2 = ops.plm_gelu_bwd_colsum_raw(dact, pre)
        assert torch.equal(dpre, dpre2) and torch.equal(db, db2)
        x = pre.float().requires_grad_(True)
        (g,) = torch.autograd.grad(torch.nn.functional.gelu(x), x, dact.float())
        err = (dpre.float() - g).abs()
        # one bf16 step of the result, plus the derivative's absolute error (1e-4: it crosses zero near x = -0.75, where no relative bound can hold) times |dact|
        tol = 8.5e-3 * g.abs() + 2e-4 * dact.float().abs() + 1e-6
        assert (err <= tol).all(), (M, H, (err / tol).max().item())
        ref = dpre.float().sum(0)
        assert (db.float() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item() + 1e-3, (M, H)


def test_fused_text_attention_core_with_dropout_against_torch_on_the_same_mask():
    """ops.PlmSelfAttnFn (one GEMM over the packed weight + fmmt_mha_fwd / _bwd | FMMT_BATCH_MAJOR): the keep-mask is read back from the kernel (q = k = 0 makes
    every probability 1 / S, one-hot values then leave keep / (S (1 - p)) in the output), its kept fraction is 1 - p, the same seed word replays it and another
    does not; forward and every gradient against torch's softmax attention in fp32 on THAT mask, with a key-padding bias, at bf16 tolerance."""
    from facialmmt_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, S, H, D, p = 3, 64, 4, 64, 0.1
    E = H * D
    seeds = torch.tensor([41, 42], device=dev, dtype=torch.int64)
    probe = torch.zeros(B, S, 3 * E, device=dev, dtype=torch.bfloat16)
    probe[..., 2 * E:] = torch.eye(S, D, device=dev, dtype=torch.bfloat16).repeat(1, H)[None]
    o, _ = ops.mha_packed_bm_fwd_raw(probe, H, 0.125, p, 0, seeds[0:1], None)
    keep = (o.view(B, S, H, D).permute(0, 2, 1, 3) != 0)                      # (B, H, query, key)
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01, frac
    o2, _ = ops.mha_packed_bm_fwd_raw(probe, H, 0.125, p, 0, seeds[1:2], None)
    assert not torch.equal(o, o2)
    inv = o.float().max().item() * S                                          # the realised 1 / keep rate (2^16 / (2^16 - round(p 2^16)))
    assert abs(inv - 1 / (1 - p)) < 1e-2
    x = torch.randn(B, S, E, device=dev).to(torch.bfloat16)
    w = (torch.randn(3 * E, E, device=dev) * E ** -0.5).to(torch.bfloat16)
    b = (0.1 * torch.randn(3 * E, device=dev)).to(torch.bfloat16)
    kb = torch.zeros(B, S, device=dev)
    kb[1, 50:] = -30000.0
    g = torch.randn(B, S, E, device=dev).to(torch.bfloat16)
    xin = x.clone().requires_grad_(True)
    ws = [w[i * E:(i + 1) * E].clone().requires_grad_(True) for i in range(3)]
    bs = [b[i * E:(i + 1) * E].clone().requires_grad_(True) for i in range(3)]
    y = ops.PlmSelfAttnFn.apply(xin, *ws, *bs, w, b, H, 0.125, p, seeds[0:1], kb)
    y_again = ops.PlmSelfAttnFn.apply(x, *ws, *bs, w, b, H, 0.125, p, seeds[0:1], kb)
    assert torch.equal(y, y_again)
    grads = torch.autograd.grad(y, [xin] + ws + bs, g)
    # torch, fp32, the same mask
    xr = x.float().requires_grad_(True)
    wr, br = w.float().requires_grad_(True), b.float().requires_grad_(True)
    qkv = torch.nn.functional.linear(xr, wr, br).view(B, S, 3, H, D).permute(2, 0, 3, 1, 4)
    sc = qkv[0] @ qkv[1].transpose(-1, -2) * 0.125 + kb[:, None, None, :]
    pr = torch.softmax(sc, dim=-1) * keep.float() * inv
    yr = (pr @ qkv[2]).transpose(1, 2).reshape(B, S, E)
    gx, gw, gb = torch.autograd.grad(yr, [xr, wr, br], g.float())
    assert (y.float() - yr).abs().max().item() <= 3e-2 * max(1.0, yr.abs().max().item())
    refs = [gx] + [gw[i * E:(i + 1) * E] for i in range(3)] + [gb[i * E:(i + 1) * E] for i in range(3)]
    for a, r_, name in zip(grads, refs, ("dx", "dWq", "dWk", "dWv", "dbq", "dbk", "dbv")):
        if name == "dbk":                                                     # mathematically zero
            continue
        sc_ = r_.abs().max().item() + 1e-6
        assert (a.float() - r_).abs().max().item() <= 4e-2 * sc_, (name, (a.float() - r_).abs().max().item(), sc_)


def test_fused_text_encoder_trains_with_attention_dropout():
    """a fused RoBERTa with attention and hidden dropout 0.1 in training mode: every attention module takes the in-tree core (its own seed word per module and
    forward), losses are finite, the gradients reach every parameter, and eval mode (no dropout) agrees with the stock module."""
    import copy
    from transformers import RobertaConfig, RobertaModel
    from facialmmt_amd import ops
    from facialmmt_amd.train_step import fuse_text_encoder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=1000, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, max_position_embeddings=80,
                        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    stock = RobertaModel(cfg, add_pooling_layer=False).to(dev).to(torch.bfloat16)
    fused = copy.deepcopy(stock)
    assert fuse_text_encoder(fused) == (4, 2)
    calls = []
    real = ops.PlmSelfAttnFn.apply
    ops.PlmSelfAttnFn.apply = lambda *a: (calls.append(a[11:13]), real(*a))[1]
    try:
        ids = torch.randint(3, 1000, (2, 48), device=dev)
        fused.train()
        y1 = fused(input_ids=ids).last_hidden_state
        y2 = fused(input_ids=ids).last_hidden_state
        assert len(calls) == 4 and all(c[0] == 0.1 for c in calls)
        assert calls[0][1].data_ptr() != calls[1][1].data_ptr()              # one seed word per attention module
        assert torch.isfinite(y1.float()).all() and not torch.equal(y1, y2)   # a fresh draw per forward
        y1.float().square().mean().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad.float()).all() for p in fused.parameters())
        fused.eval(), stock.eval()
        with torch.enable_grad():
            ye = fused(input_ids=ids).last_hidden_state
        assert len(calls) == 6 and calls[-1][0] == 0.0
        with torch.no_grad():
            ys = stock(input_ids=ids).last_hidden_state
        assert (ye.float() - ys.float()).abs().max().item() <= 3e-2 * max(1.0, ys.float().abs().max().item())
    finally:
        ops.PlmSelfAttnFn.apply = real


def test_fused_sublayer_tail_dropout_replays_its_mask_in_the_backward():
    """fmmt_plm_dropadd_ln_fwd / _bwd at p = 0.3: the kept fraction is 1 - p, a second forward with the same (seed, salt) is identical and a different
    salt is not; the backward's dense-output gradient is zero exactly where the forward dropped and dx * 1 / (1 - p) elsewhere; the bias gradient is its
    column sum; against torch (LayerNorm of the same dropped sum) at bf16 tolerance."""
    from facialmmt_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    M, K, C, p = 192, 128, 1024, 0.3
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    res = (0.05 * torch.randn(M, C, device=dev)).to(torch.bfloat16)       # small beside h: a kept element never rounds away in h + res
    w = (torch.randn(C, K, device=dev) * K ** -0.5).to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(C, device=dev).to(torch.bfloat16).requires_grad_(True)
    gm = (1 + 0.1 * torch.randn(C, device=dev)).to(torch.bfloat16).requires_grad_(True)
    bt = (0.1 * torch.randn(C, device=dev)).to(torch.bfloat16).requires_grad_(True)
    seed = torch.tensor([12345], device=dev, dtype=torch.int64)
    xin, rin = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y = ops.PlmSublayerTailFn.apply(xin, rin, w, b, gm, bt, 1e-5, p, seed, 7 << 40)
    y2 = ops.PlmSublayerTailFn.apply(x, res, w, b, gm, bt, 1e-5, p, seed, 7 << 40)
    y3 = ops.PlmSublayerTailFn.apply(x, res, w, b, gm, bt, 1e-5, p, seed, 8 << 40)
    assert torch.equal(y, y2) and not torch.equal(y, y3)
    # the mask, exactly: the same launch on h = 1, res = 0 leaves keep / (1 - p) in the saved sum
    from facialmmt_amd import _lib
    ones, zeros = torch.ones(M, C, device=dev, dtype=torch.bfloat16), torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
    probe, junk = torch.empty_like(ones), torch.empty_like(ones)
    _lib.check(_lib.load().fmmt_plm_dropadd_ln_fwd(M, C, 1e-5, ones.data_ptr(), zeros.data_ptr(), gm.data_ptr(), bt.data_ptr(), p, 0, seed.data_ptr(), 7 << 40,
                                                   probe.data_ptr(), junk.data_ptr(), torch.cuda.current_stream().cuda_stream), "probe")
    keep = probe != 0
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01, frac
    dy = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
    y.backward(dy)
    # torch on the same mask
    w2, b2, gm2, bt2 = (t.detach().clone().requires_grad_(True) for t in (w, b, gm, bt))
    x2, r2 = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    h2 = torch.nn.functional.linear(x2, w2, b2)
    t2 = (h2 * keep.to(h2.dtype) * (1.0 / (1.0 - p))).to(torch.bfloat16)
    yr = torch.nn.functional.layer_norm(t2 + r2, (C,), gm2, bt2, 1e-5)
    yr.backward(dy)
    assert (y.float() - yr.float()).abs().max().item() <= 3e-2 * max(1.0, yr.float().abs().max().item())
    for a, r_, name in ((xin.grad, x2.grad, "dx"), (rin.grad, r2.grad, "dres"), (w.grad, w2.grad, "dW"), (b.grad, b2.grad, "db"),
                        (gm.grad, gm2.grad, "dgamma"), (bt.grad, bt2.grad, "dbeta")):
        s = r_.float().abs().max().item() + 1e-6
        assert (a.float() - r_.float()).abs().max().item() <= 4e-2 * s, (name, (a.float() - r_.float()).abs().max().item(), s)


def test_dropadd_layer_norm_with_fp32_parameters_against_torch_on_the_same_mask():
    """ops.dropadd_layer_norm (fmmt_dropadd_ln_fwd / _bwd, param_dtype FMMT_F32): the training-mode tail of MELDTransEncoder's sublayers
    (modules/Transformer.py:121-123,134-136) with fp32 master LayerNorm parameters and the TF epsilon 1e-12.  p = 0: equal to LayerNorm(h + res) of the same
    bf16 sum; p = 0.1: against torch on the mask the kernel drew (read back by a probe launch), forward and all four gradients, fp32 parameter gradients."""
    from facialmmt_amd import _lib, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    M, C = 640, 768
    h = torch.randn(M, C, device=dev).to(torch.bfloat16)
    res = (0.05 * torch.randn(M, C, device=dev)).to(torch.bfloat16)
    gm = (1 + 0.1 * torch.randn(C, device=dev)).requires_grad_(True)
    bt = (0.1 * torch.randn(C, device=dev)).requires_grad_(True)
    seed = torch.tensor([777], device=dev, dtype=torch.int64)
    y0 = ops.dropadd_layer_norm(h, res, gm, bt, 1e-12, 0.0, seed, 3)
    r0 = torch.nn.functional.layer_norm((h + res).float(), (C,), gm, bt, 1e-12)
    assert (y0.float() - r0).abs().max().item() <= 1e-2 * r0.abs().max().item()
    for p in (0.1,):
        ones, zeros = torch.ones(M, C, device=dev, dtype=torch.bfloat16), torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
        probe, junk = torch.empty_like(ones), torch.empty_like(ones)
        _lib.check(_lib.load().fmmt_dropadd_ln_fwd(_lib.dtype_code(torch.float32), M, C, 1e-12, ones.data_ptr(), zeros.data_ptr(), gm.data_ptr(), bt.data_ptr(), p, 0,
                                                   seed.data_ptr(), 3, probe.data_ptr(), junk.data_ptr(), torch.cuda.current_stream().cuda_stream), "probe")
        keep = probe != 0
        assert abs(keep.float().mean().item() - (1 - p)) < 0.01
        hin, rin = h.clone().requires_grad_(True), res.clone().requires_grad_(True)
        y = ops.dropadd_layer_norm(hin, rin, gm, bt, 1e-12, p, seed, 3)
        dy = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
        gh, gr, gg, gb = torch.autograd.grad(y, [hin, rin, gm, bt], dy)
        assert gg.dtype == torch.float32 and gb.dtype == torch.float32
        h2, r2 = h.clone().requires_grad_(True), res.clone().requires_grad_(True)
        gm2, bt2 = gm.detach().clone().requires_grad_(True), bt.detach().clone().requires_grad_(True)
        t2 = (h2 * keep.to(h2.dtype) * (1.0 / (1.0 - p))).to(torch.bfloat16)
        yr = torch.nn.functional.layer_norm((t2 + r2).float(), (C,), gm2, bt2, 1e-12)
        rh, rr, rg, rb = torch.autograd.grad(yr, [h2, r2, gm2, bt2], dy.float())
        assert (y.float() - yr).abs().max().item() <= 2e-2 * yr.abs().max().item()
        for a, r_, name in ((gh, rh, "dh"), (gr, rr, "dres"), (gg, rg, "dgamma"), (gb, rb, "dbeta")):
            s = r_.float().abs().max().item() + 1e-6
            assert (a.float() - r_.float()).abs().max().item() <= 3e-2 * s, (name, (a.float() - r_.float()).abs().max().item(), s)
        assert torch.equal((gh != 0) | ~keep, torch.ones_like(keep)) or (gh[~keep] == 0).all()      # the dense gradient is zero where the forward dropped
        assert (gh[~keep] == 0).all()


def test_dropadd_layer_norm_takes_views_at_odd_storage_offsets():
    """A residual / gradient that is a contiguous view 8 bytes into its storage (a slice of a packed buffer) is copied to an aligned buffer rather than
    answered with FMMT_EALIGN (round-5 ADVICE): same bits as the aligned call, forward and backward."""
    from facialmmt_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    M, C = 96, 768
    h = torch.randn(M, C, device=dev).to(torch.bfloat16)
    buf = (0.05 * torch.randn(M * C + 4, device=dev)).to(torch.bfloat16)
    res_odd = buf[4:].view(M, C)
    assert res_odd.is_contiguous() and res_odd.data_ptr() % 16 == 8
    gm = (1 + 0.1 * torch.randn(C, device=dev)).requires_grad_(True)
    bt = (0.1 * torch.randn(C, device=dev)).requires_grad_(True)
    seed = torch.tensor([99], device=dev, dtype=torch.int64)
    dbuf = torch.randn(M * C + 4, device=dev).to(torch.bfloat16)
    outs = []
    for res, dy in ((res_odd, dbuf[4:].view(M, C)), (res_odd.clone(), dbuf[4:].view(M, C).clone())):
        hin, rin = h.clone().requires_grad_(True), res.detach().requires_grad_(True)
        y = ops.dropadd_layer_norm(hin, rin, gm, bt, 1e-12, 0.1, seed, 7)
        outs.append((y,) + torch.autograd.grad(y, [hin, rin, gm, bt], dy))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_meld_encoder_training_mode_fused_tails_match_the_separate_launches():
    """MELDTransEncoder in training mode, bf16: with hidden dropout forced to 0 inside _tail the fused launch and the three separate launches agree (same
    bf16 rounding op by op); with p = 0.1 both run, finite, and two passes under the same torch seed are identical (device-drawn seeds)."""
    import types
    from facialmmt_amd.config import default_args
    from facialmmt_amd.modules import Transformer as TM
    dev = torch.device("cuda:0")
    cfg = default_args()
    torch.manual_seed(0)
    m = TM.MELDTransEncoder(cfg, 2, 160, 768).to(dev).train()
    m.compute_dtype = torch.bfloat16
    x = (0.5 * torch.randn(4, 160, 768, device=dev)).requires_grad_(True)
    mask = torch.ones(4, 160, device=dev)
    mask[1, 100:] = 0
    ext = (1.0 - mask.unsqueeze(1).unsqueeze(2)) * -10000.0
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.1
    torch.manual_seed(11)
    y1 = m(x, ext)
    g1 = torch.autograd.grad(y1.square().mean(), [x] + list(m.parameters()))
    torch.manual_seed(11)
    y2 = m(x, ext)
    assert torch.equal(y1, y2) and torch.isfinite(y1).all() and all(torch.isfinite(g).all() for g in g1)
    # attention dropout off, hidden dropout "on" with p -> 0: the fused tail against the separate launches
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 1e-9 if not isinstance(getattr(mod, "_is_attn", None), bool) else 0.0
    for layer in m.layer:
        layer.transformer_self_attention.selfatt.dropout.p = 0.0
    outs = []
    for fused in (True, False):
        TM.FUSED_TAILS = fused
        try:
            y = m(x, ext)
            outs.append((y, torch.autograd.grad(y.square().mean(), [x] + list(m.parameters()))))
        finally:
            TM.FUSED_TAILS = True
    (ya, ga), (yb, gb) = outs
    assert (ya - yb).abs().max().item() <= 2e-2 * yb.abs().max().item()
    for a, b in zip(ga, gb):
        assert (a.float() - b.float()).abs().max().item() <= 5e-2 * (b.float().abs().max().item() + 1e-8)


@pytest.mark.parametrize("M,H", [(640, 768), (512, 768), (152, 768), (64, 128)])
def test_linear_fwd_seg3_against_three_products(M, H):
    """fmmt_linear_fwd_seg3: three weights in one launch, segmented along the output channels (query / key / value of one input) and along the
    contraction (their input gradient), against fp64 products of the same bf16 operands."""
    from facialmmt_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(M + H)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(M, H, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(H, H, device=dev) * H ** -0.5).to(torch.bfloat16) for _ in range(3)]
    bs = [torch.randn(H, device=dev) for _ in range(3)]
    y = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.fmmt_linear_fwd_seg3(_lib.dtype_code(torch.bfloat16), M, 3 * H, H, x.data_ptr(), H, ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), H, 1,
                                        bs[0].data_ptr(), bs[1].data_ptr(), None, y.data_ptr(), 3 * H, st), "seg3 N")
    for i in range(3):
        ref = x.double() @ ws[i].double().t() + (bs[i].double() if i < 2 else 0.0)
        assert (y[:, i * H:(i + 1) * H].double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), i
    d = torch.randn(M, 3 * H, device=dev).to(torch.bfloat16)
    dx = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    wt = [w.t().contiguous() for w in ws]                        # [N = H_in][K = H_out]
    _lib.check(lib.fmmt_linear_fwd_seg3(_lib.dtype_code(torch.bfloat16), M, H, 3 * H, d.data_ptr(), 3 * H, wt[0].data_ptr(), wt[1].data_ptr(), wt[2].data_ptr(), H, 2,
                                        None, None, None, dx.data_ptr(), H, st), "seg3 K")
    ref = sum(d[:, i * H:(i + 1) * H].double() @ ws[i].double() for i in range(3))
    assert (dx.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    # refusals: fp32, a bias with the K form, a segment that is not a multiple of 64
    assert lib.fmmt_linear_fwd_seg3(_lib.dtype_code(torch.float32), M, 3 * H, H, x.data_ptr(), H, ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), H, 1,
                                    None, None, None, y.data_ptr(), 3 * H, st) == _lib.FMMT_EINVAL
    assert lib.fmmt_linear_fwd_seg3(_lib.dtype_code(torch.bfloat16), M, H, 3 * H, d.data_ptr(), 3 * H, wt[0].data_ptr(), wt[1].data_ptr(), wt[2].data_ptr(), H, 2,
                                    bs[0].data_ptr(), None, None, dx.data_ptr(), H, st) == _lib.FMMT_EINVAL
    assert lib.fmmt_linear_fwd_seg3(_lib.dtype_code(torch.bfloat16), M, 3 * 96, H, x.data_ptr(), H, ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), H, 1,
                                    None, None, None, y.data_ptr(), 3 * 96, st) == _lib.FMMT_EINVAL


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_self_attention_packed_qkv_matches_three_linears(dtype):
    """modules.Transformer.SelfAttention with PACKED_QKV (ops.SelfAttnQkvFn: one projection launch, attention core on the packed buffer, one input-
    gradient launch, one weight-gradient contraction) against the three-Linear formulation: outputs and every gradient (fp32: the packed path falls
    back to three launches inside the op, same arithmetic)."""
    from facialmmt_amd.config import default_args
    from facialmmt_amd.modules import Transformer as TM
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    att = TM.SelfAttention(default_args()).to(dev).eval()
    for lin in (att.query, att.key, att.value):
        torch.nn.init.normal_(lin.bias, std=0.1)
    L, B = 160, 4
    x = torch.randn(L, B, 768, device=dev).to(dtype).requires_grad_(True)
    kb = torch.zeros(B, L, device=dev)
    kb[2, 120:] = -10000.0
    w = torch.randn(L, B, 768, device=dev).to(dtype)
    res = []
    for packed in (True, False):
        TM.PACKED_QKV = packed
        try:
            y = att.forward_tm(x, kb)
            g = torch.autograd.grad((y.float() * w.float()).sum(), [x] + list(att.parameters()))
        finally:
            TM.PACKED_QKV = True
        res.append((y, g))
    (ya, ga), (yb, gb) = res
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert (ya.float() - yb.float()).abs().max().item() <= tol * yb.float().abs().max().item()
    names = ["dx"] + [n for n, _ in att.named_parameters()]
    for n, a, b in zip(names, ga, gb):
        if n == "key.bias":                                       # mathematically zero: rounding noise on both sides
            continue
        assert (a.float() - b.float()).abs().max().item() <= 2 * tol * (b.float().abs().max().item() + 1e-8), n


@pytest.mark.parametrize("vis_dtype", [torch.float32, torch.bfloat16])
def test_select_frames_kernel_matches_torch_restatement_with_gradients(dev, vis_dtype):
    """fmmt_select_frames_fwd / _bwd (one launch each) against train_step.select_frames' torch formulation (SELECT_FRAMES_KERNEL = False), which the
    literal-loop test above pins: outputs bit for bit, d(preds) through the appended probability columns equal, both branches, bench geometry too."""
    from facialmmt_amd import train_step as TS
    g = torch.Generator().manual_seed(7)
    cases = [(_filter_case(g, B, 12, 12, peaky=(t % 2 == 0)), thr) for t, (B, thr) in enumerate([(1, 0.2), (3, 0.5), (5, 0.2), (4, 1.5), (2, 0.0), (5, 0.99)])]
    cases.append((_filter_case(g, 4, 160, 160, peaky=True, D=512), 0.2))
    cases.append((_filter_case(g, 4, 160, 160, peaky=False, D=512), 1.5))
    for (preds, vis, mask, num), thr in cases:
        res = []
        for kernel in (True, False):
            TS.SELECT_FRAMES_KERNEL = kernel
            try:
                p = preds.to(dev).requires_grad_(True)
                out, m = TS.select_frames(p, vis.to(dev).to(vis_dtype), mask.to(dev), num.to(dev), thr)
                w = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(dev)
                (gp,) = torch.autograd.grad((out.float() * w).sum(), [p])
            finally:
                TS.SELECT_FRAMES_KERNEL = True
            res.append((out, m, gp))
        (oa, ma, ga), (ob, mb, gb) = res
        assert oa.dtype == ob.dtype and ma.dtype == mb.dtype
        assert torch.equal(oa, ob) and torch.equal(ma, mb)
        assert torch.allclose(ga, gb, rtol=0, atol=0 if vis_dtype == torch.float32 else 1e-6), (ga - gb).abs().max()

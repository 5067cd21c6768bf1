"""GPU parity of the Swin encoder (HIP path behind the reference's module API) against the golden
vectors produced by the reference itself: blocks / window attention / patch merging / patch embed per
stage geometry, whole-model eval (N=8 and the batch-of-1 duplication), train-mode BatchNorm and running
statistics, pre-Gumbel logits, and gradients (d input + 20 parameter gradients).  Tolerance: 1e-3
absolute+relative in fp32 (north_star); bf16 is held to 3e-2 of the output scale against fp32."""
import os

import numpy as np
import pytest
import torch

from facialmmt_amd import synth

pytestmark = pytest.mark.gpu
GEO = [(56, 96, 3), (28, 192, 6), (14, 384, 12), (7, 768, 24)]
TOL = dict(atol=1e-3, rtol=1e-3)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def S():
    from facialmmt_amd.modules.SwinTransformer import Swin_Transformer
    return Swin_Transformer


@pytest.fixture(scope="module")
def swin(dev, S):
    from facialmmt_amd.modules.SwinTransformer.backbone_def import BackboneFactory
    m = BackboneFactory("SwinTransformer", os.path.join(os.path.dirname(S.__file__), "swin_conf.yaml")).get_backbone()
    synth.fill_state_dict(m, seed=100)
    return m.to(dev)


@pytest.mark.parametrize("s", [0, 1, 2, 3])
@pytest.mark.parametrize("shift", [0, 3])
def test_block_and_window_attention(golden, dev, S, s, shift):
    from oracle.swin import window_token_index
    H, C, nh = GEO[s]
    blk = S.SwinTransformerBlock(C, (H, H), nh, window_size=7, shift_size=shift, drop_path=0.0).eval()
    synth.fill_state_dict(blk, seed=10 + s, prefix=f"blk{s}.")
    blk.to(dev)
    x = synth.tensor(f"blk_in{s}", (2, H * H, C), seed=s).to(dev)
    with torch.no_grad():
        golden.check("swin_parts", f"block_s{s}_shift{shift}", blk(x), **TOL)
        idx = window_token_index(H, H, 7, blk.shift_size).to(dev)
        xw = x[:1, idx.reshape(-1)].reshape(-1, 49, C)
        golden.check("swin_parts", f"wattn_s{s}_shift{shift}", blk.attn(xw, mask=blk.attn_mask), **TOL)


@pytest.mark.parametrize("s", [0, 1, 2])
def test_patch_merging(golden, dev, S, s):
    H, C, _ = GEO[s]
    pm = S.PatchMerging((H, H), C).eval()
    synth.fill_state_dict(pm, seed=20 + s, prefix=f"pm{s}.")
    pm.to(dev)
    with torch.no_grad():
        golden.check("swin_parts", f"merge_s{s}", pm(synth.tensor(f"blk_in{s}", (2, H * H, C), seed=s).to(dev)), **TOL)


def test_patch_embed(golden, dev, S):
    pe = S.PatchEmbed(224, 4, 3, 96, torch.nn.LayerNorm).eval()
    synth.fill_state_dict(pe, seed=30, prefix="pe.")
    pe.to(dev)
    with torch.no_grad():
        golden.check("swin_parts", "patch_embed", pe(synth.tensor("frames", (2, 3, 224, 224), seed=1).to(dev)), **TOL)


def test_swin_eval_and_stage_taps(golden, dev, swin):
    frames = synth.tensor("frames", (8, 3, 224, 224), seed=1).to(dev)
    swin.eval()
    with torch.no_grad():
        golden.check("swin_full", "swin_eval_n8", swin(frames), **TOL)
        golden.check("swin_full", "swin_eval_n1", swin(frames[:1]), **TOL)
        x = swin.patch_embed(frames[:2])
        for s, layer in enumerate(swin.layers):
            x = layer(x)
            golden.check("swin_full", f"swin_stage{s}_n2", x, **TOL)
        o32 = swin(frames)
        o16 = swin(frames.bfloat16()).float()
    assert (o16 - o32).abs().max().item() <= 3e-2 * o32.abs().max().item()


def _no_droppath(swin, S):
    for m in swin.modules():
        if isinstance(m, S.DropPath):
            m.drop_prob = 0.0


def test_swin_train_batchnorm_and_running_stats(golden, dev, swin, S):
    frames = synth.tensor("frames", (8, 3, 224, 224), seed=1).to(dev)
    _no_droppath(swin, S)       # goldens were produced with DropPath = identity
    swin.train()
    bn = swin.output_layer[3]
    rm0, rv0, nb0 = bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)
    with torch.no_grad():
        golden.check("swin_full", "swin_train_n4", swin(frames[:4]), **TOL)
    golden.check("swin_full", "bn_running_mean_after", bn.running_mean, **TOL)
    golden.check("swin_full", "bn_running_var_after", bn.running_var, **TOL)
    assert int(bn.num_batches_tracked) == nb0 + 1
    with torch.no_grad():
        bn.running_mean.copy_(rm0)
        bn.running_var.copy_(rv0)
    swin.eval()


def test_affwild_logits_and_gradients(golden, dev, S):
    from facialmmt_amd import models
    from facialmmt_amd.config import default_args
    aff = models.SwinForAffwildClassification(default_args())
    synth.fill_state_dict(aff, seed=100)
    aff.to(dev)
    _no_droppath(aff.swin, S)
    frames = synth.tensor("frames", (8, 3, 224, 224), seed=1).to(dev)
    aff.eval()
    with torch.no_grad():
        golden.check("swin_full", "affwild_logits_n4", aff(frames[:4], is_trg_task=False), **TOL)
    aff.train()
    xin = frames[:3].clone().requires_grad_(True)
    probe = synth.tensor("probe7", (3, 7), seed=3).to(dev)
    (aff(xin, is_trg_task=False) * probe).sum().backward()
    # (round 5: gradient goldens at the north-star 1e-3 class -- they were held at rtol 5e-3 while the measured error was 2e-4 of the tensor's scale)
    golden.check("swin_full", "grad/input", xin.grad, atol=2e-5, rtol=1e-3, sum_rtol=1e-3)
    params = dict(aff.named_parameters())
    z = golden.files["swin_full"]
    names = sorted({k.split("/")[1] for k in z.files if k.startswith("grad/") and k.split("/")[1] not in ("input", "swin.output_layer.2.bias")})
    assert len(names) >= 20
    for n in names:
        ref, _ = golden.expected("swin_full", f"grad/{n}")
        golden.check("swin_full", f"grad/{n}", params[n].grad, atol=3e-4 * float(np.abs(ref).max()) + 1e-7, rtol=1e-3, sum_rtol=1e-3)


def _diverse_frames(n, seed=1):
    """n structured, mutually different frames in [-1, 1] (the value range of Normalize(ToTensor(.))): per frame and channel a plane
    wave of hash-drawn frequency / direction / phase, a Gaussian blob at a hash-drawn place, and 10 % hash noise.  (Eight frames of
    iid noise have almost identical Swin features: BatchNorm1d on their batch statistics then divides by a near-zero spread and
    amplifies every rounding in front of it -- the conditioning of that fixture, not of the kernels, set round 2's figures.)"""
    g = synth.tensor("frame_params", (n, 12), seed=seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 224), torch.linspace(-1, 1, 224), indexing="ij")
    out = []
    for i in range(n):
        p = g[i]
        chans = []
        for c in range(3):
            f = 2.0 + 6.0 * (p[c] + 1)
            wave = torch.sin(f * (xx * p[3] + yy * p[4]) + 3.0 * p[5 + c]) * (0.4 + 0.3 * p[8])
            blob = torch.exp(-((xx - 0.6 * p[9]) ** 2 + (yy - 0.6 * p[10]) ** 2) / (0.05 + 0.2 * (p[11] + 1))) * (0.8 * p[c])
            chans.append(wave + blob)
        out.append(torch.stack(chans))
    return (torch.stack(out) + 0.1 * synth.tensor("frame_noise", (n, 3, 224, 224), seed=seed + 1)).clamp(-1, 1)


def _grad_stats(pairs, skip=()):
    stats = {}
    for name, g, r in pairs:
        assert g is not None and r is not None, name
        g, r = g.float().cpu().reshape(-1).double(), r.float().cpu().reshape(-1).double()
        if float(r.norm()) == 0.0:
            assert float(g.norm()) <= 1e-6, name
            continue
        if name in skip:
            continue
        stats[name] = (float((g @ r) / (g.norm() * r.norm())), float((g - r).norm() / r.norm()))
    return stats


@pytest.mark.parametrize("bn_mode", ["running_stats", "batch_stats"])
def test_bf16_gradients_against_oracle_n32(dev, S, bn_mode):
    """The dtype the benchmark runs in, end to end: every parameter gradient and the input gradient of the bf16 HIP path (fused
    window-attention block halves at stage 0, MFMA window attention, bf16 GEMM instantiations, fused Mlp, bf16 LayerNorm) on 32
    DIVERSE frames against the fp32 oracle differentiated on the CPU -- cosine similarity and relative L2 error per tensor --
    with, beside it, what STOCK PyTorch-ROCm bf16 gives for the same graph: the oracle's own functional torch code run on the GPU
    under torch.autocast(bfloat16) against the same fp32 gradients.

    running_stats: BatchNorm1d of the head on its running statistics; batch_stats: on the statistics of the 32-frame batch, the
    mode the training step (and the benchmark) runs in.  Bar: cosine >= 0.99 and relative L2 error <= 10 % for every tensor, in both
    modes (round-3 VERDICT: the "or within 1.5 x stock's worst tensor" escape is gone).  (The figure is a chaotic function of the rounding
    realisation: the probe sits behind Linear -> ReLU -> Linear and, in batch_stats mode, behind a normalisation by the batch spread,
    so one flipped ReLU gate or a slightly different spread moves every gradient below it by the same factor.  Measured in round 3 on
    the same 32 frames: stage 0 fused 4.0 % mean / 6.6 % worst, stage 0 as four launches 12.7 % / 21 %, stock autocast 11 % / 16.5 %
    -- profiles/r03_bf16_grad_stats_*.txt.)  The table goes to gpurun_out/ (committed under profiles/ per round)."""
    from facialmmt_amd import models
    from facialmmt_amd.config import default_args
    from oracle import swin as OS
    train = bn_mode == "batch_stats"
    COS_MIN, REL_MAX, N = 0.99, 0.10, 32
    aff = models.SwinForAffwildClassification(default_args())
    synth.fill_state_dict(aff, seed=100)
    _no_droppath(aff.swin, S)
    aff.to(dev).train(train)
    frames = _diverse_frames(N)
    probe = synth.tensor("probe7", (N, 7), seed=3)
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in aff.state_dict().items()}
    xr = frames.clone().requires_grad_(True)
    ref = OS.swin_affwild_logits(sd, xr, training=train)
    (ref * probe).sum().backward()
    # ours
    x16 = frames.to(dev).bfloat16().requires_grad_(True)
    out = aff(x16, is_trg_task=False)
    (out.float() * probe.to(dev)).sum().backward()
    assert (out.float().cpu() - ref.detach()).abs().max().item() <= 3e-2 * ref.abs().max().item()
    # in front of train-mode BatchNorm a per-feature constant cancels: the true gradients of the head's LayerNorm bias
    # and Linear bias are 0 up to rounding (nothing to compare a direction with)
    skip = ("swin.output_layer.0.bias", "swin.output_layer.2.bias") if train else ()
    ours = _grad_stats([("input", x16.grad, xr.grad)] + [(k, p.grad, sd[k].grad) for k, p in aff.named_parameters()], skip)
    # stock PyTorch-ROCm bf16: the oracle's functional graph on the GPU under autocast
    sdg = {k: v.detach().to(dev).requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    xg = frames.to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outg = OS.swin_affwild_logits(sdg, xg, training=train)
    (outg.float() * probe.to(dev)).sum().backward()
    stock = _grad_stats([("input", xg.grad, xr.grad)] + [(k, sdg[k].grad, sd[k].grad) for k, _ in aff.named_parameters()], skip)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"bf16_grad_stats_{bn_mode}.txt"), "w") as f:
            f.write(f"# {N} diverse frames, BatchNorm1d on {bn_mode}: cosine / relative L2 error against the fp32 oracle's gradients -- ours | stock bf16 autocast\n")
            f.write("\n".join(f"{c:.5f} {r:.5f} | {stock[k][0]:.5f} {stock[k][1]:.5f} {k}" for k, (c, r) in ours.items()) + "\n")
    wc, wr = min((c, k) for k, (c, r) in ours.items()), max((r, k) for k, (c, r) in ours.items())
    sc, sr = min((c, k) for k, (c, r) in stock.items()), max((r, k) for k, (c, r) in stock.items())
    print(f"bf16 vs fp32-oracle gradients ({bn_mode}, {N} frames) over {len(ours)} tensors: ours worst cosine {wc}, worst relative L2 {wr}; "
          f"stock bf16 autocast worst cosine {sc}, worst relative L2 {sr}")
    bad = [(k, c, r, stock[k]) for k, (c, r) in ours.items() if not (c >= COS_MIN and r <= REL_MAX)]
    assert not bad, bad[:8]
    assert len(ours) >= 173


def test_droppath_matches_oracle_with_explicit_masks(dev, S):
    """stochastic depth: feed the same per-sample multipliers to the oracle and to the HIP block"""
    from oracle import swin as OS
    H, C, nh = 14, 384, 12
    blk = S.SwinTransformerBlock(C, (H, H), nh, window_size=7, shift_size=3, drop_path=0.25).train()
    synth.fill_state_dict(blk, seed=12, prefix="blk2.")
    blk.to(dev)
    x = synth.tensor("blk_in2", (6, H * H, C), seed=2).to(dev)
    torch.manual_seed(5)
    s1 = blk.drop_path.sample_scale(6, dev)
    s2 = blk.drop_path.sample_scale(6, dev)
    torch.manual_seed(5)
    with torch.no_grad():
        got = blk(x)
        sd = {k: v for k, v in blk.state_dict().items()}
        ref = OS.swin_block(sd, "", x, H, H, nh, 3, (s1, s2))
    assert (s1 == 0).any() or (s2 == 0).any()
    assert (got - ref).abs().max().item() < 1e-3


def test_full_size_batch_independence_bf16(dev, swin):
    """BASELINE size (160 frames = one utterance, bf16, eval): frames are independent units, so the
    features of a sub-batch must be bit-identical to the same rows of the full batch."""
    swin.eval()
    g = torch.Generator(device="cpu").manual_seed(1)
    frames = torch.randn(160, 3, 224, 224, generator=g).to(dev).bfloat16()
    with torch.no_grad():
        full = swin(frames)
        part = swin(frames[40:104])
    assert torch.isfinite(full).all()
    assert torch.equal(full[40:104], part)


def test_full_size_backward_is_deterministic_and_linear_bf16(dev, swin):
    """BASELINE size (640 frames = the 4-utterance step of the bench, bf16): the backward has no atomics, so two
    runs give bit-identical gradients (a race shows up here), and it is linear in the upstream gradient: doubling it
    (exact in bf16 / fp32) doubles every parameter gradient and the input gradient."""
    swin.eval()                                                 # drop-path off, BatchNorm on running statistics
    g = torch.Generator(device="cpu").manual_seed(2)
    frames = torch.randn(640, 3, 224, 224, generator=g).bfloat16().to(dev).requires_grad_(True)
    w = torch.randn(640, 512, generator=g).to(dev)
    params = [p for p in swin.parameters() if p.requires_grad]

    def grads(scale):
        out = swin(frames)
        return torch.autograd.grad((out.float() * (w * scale)).sum(), [frames] + params, allow_unused=True)

    a, b, c = grads(1.0), grads(1.0), grads(2.0)
    used = 0
    for x, y, z in zip(a, b, c):
        if x is None:
            assert y is None and z is None
            continue
        used += 1
        assert torch.isfinite(x).all()
        assert torch.equal(x, y)
        assert (z.float() - 2 * x.float()).abs().max().item() <= 1e-20
    assert used > 150 and float(a[0].float().abs().max()) > 0
    # an intermittent race needs more than two passes to show (round 3: one stage-1 launch in ~40 wrote a garbage fragment -- one run of
    # this test in six failed): 24 more passes, every gradient bit for bit
    for _ in range(24):
        for i, (x, y) in enumerate(zip(a, grads(1.0))):
            assert x is None or torch.equal(x, y), i


def test_aux_task_step_learns(dev):
    """train.py:15-41 (aux task): a few AdamW steps on a fixed synthetic batch in bf16 reduce the loss and
    update every Swin parameter group (end-to-end check of forward, backward and optimizer plumbing)."""
    from facialmmt_amd import models
    from facialmmt_amd.config import default_args
    from facialmmt_amd.train_step import AuxStep
    args = default_args()
    torch.manual_seed(0)
    aff = models.SwinForAffwildClassification(args).to(dev).train()
    opt = torch.optim.AdamW(aff.parameters(), lr=2e-4)
    step = AuxStep(aff, opt, None, args)
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(12, 3, 224, 224, generator=g).to(dev).bfloat16()
    y = torch.randint(0, 7, (12,), generator=g).to(dev)
    w0 = aff.swin.layers[2].blocks[3].mlp.fc1.weight.detach().clone()
    t0 = aff.swin.layers[0].blocks[1].attn.relative_position_bias_table.detach().clone()
    torch.manual_seed(1)
    losses = [float(step(x, y)) for _ in range(8)]
    assert all(l == l for l in losses)                      # finite
    assert min(losses[-3:]) < losses[0]
    assert not torch.equal(w0, aff.swin.layers[2].blocks[3].mlp.fc1.weight)
    assert not torch.equal(t0, aff.swin.layers[0].blocks[1].attn.relative_position_bias_table)

"""GPU test of the step scheduling: the target-task step run (a) eagerly on one stream, (b) with the multimodal model
as two HIP graphs and the text branch on a second HIP stream, (c) the same plus parallel.GradientAverager (persistent
bucketed gradients, world size 1) must walk the same trajectory -- same losses, same updated parameters -- over six
optimisation steps (a ROCm 7.0 HIP-graph defect that corrupts replayed gradients from the third replay on is what this test
first caught; facialmmt_amd/__init__.py carries the workaround).  Dropout inside the multimodal model is off; Swin's DropPath / Gumbel noise replay from the seed."""
import os
import types

import pytest
import torch

from facialmmt_amd import synth

pytestmark = pytest.mark.gpu


def _build(dev, graphs, averager):
    from facialmmt_amd import models
    from facialmmt_amd.config import default_args
    from facialmmt_amd.parallel import GradientAverager
    from facialmmt_amd.train_step import TargetStep, graph_multimodal, select_frames
    import bench
    B, Lv = 2, 6
    cfg = default_args(get_vision_utt_max_lens=Lv, get_audio_utt_max_lens=24, trg_accumulation_steps=1, plm_module=synth.make_standin_plm(),
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, crossmodal_attn_dropout_TA=0.0, crossmodal_attn_dropout_TA_V=0.0)
    cfg.compute_dtype = torch.float32
    swin = models.SwinForAffwildClassification(cfg)
    mm = models.MultiModalTransformerForClassification(cfg)
    synth.fill_state_dict(swin, seed=100)
    synth.fill_state_dict(mm, seed=200)
    swin.to(dev).train()
    mm.to(dev).train()
    args = types.SimpleNamespace(utts=B, frames=Lv, dtype="fp32")
    batch = list(bench.synth_batch(args, dev, 0, cfg))
    batch[0] = batch[0] % 1000                                  # ids within the stand-in encoder's vocabulary
    batch = tuple(batch)
    avg = None
    if graphs:
        with torch.no_grad():
            preds = swin(batch[8], is_trg_task=True).float()
        vis, nmask = select_frames(preds, batch[5], batch[6], batch[9], cfg.FacialEmoImpor_threshold)
        sample = (batch[0], batch[1], batch[2], batch[3], batch[4], vis.detach().requires_grad_(True), nmask, batch[10])
        mm = graph_multimodal(mm, sample, None, overlap_text=True)
        assert mm.text_stream is not None
        mm.zero_grad(set_to_none=True)
        swin.zero_grad(set_to_none=True)
    if averager:
        plm = mm.roberta
        text_params = list(plm.parameters()) + list(mm.text_linear.parameters())
        ids = set(map(id, text_params))
        avg = GradientAverager(None, bucket_mb=1, groups=[[p for p in mm.parameters() if id(p) not in ids], text_params])
    # plain SGD: linear in the gradients, so run-to-run rounding noise (atomic adds in the embedding backward) stays
    # at rounding level instead of being renormalised to +-lr by Adam on near-zero gradient entries
    opt = torch.optim.SGD(mm.parameters(), lr=0.05)
    return TargetStep(swin, mm, opt, None, cfg, autocast_dtype=None, averager=avg), mm, batch


@pytest.mark.parametrize("mode", ["graphs+second_stream", "graphs+second_stream+averager"])
def test_scheduled_step_equals_eager_step(mode):
    dev = torch.device("cuda:0")
    runs = {}
    for name, (graphs, averager) in {"eager": (False, False), mode: (True, "averager" in mode)}.items():
        step, mm, batch = _build(dev, graphs, averager)
        losses = []
        for i in range(6):                                      # the ROCm graph replay defect this guards against starts at the third replay
            torch.manual_seed(1234 + i)                          # Swin's DropPath masks and Gumbel noise
            loss, _ = step(batch)
            losses.append(loss.item())
        torch.cuda.synchronize()
        runs[name] = (losses, {k: v.detach().clone() for k, v in mm.named_parameters()})
    (l0, p0), (l1, p1) = runs["eager"], runs[mode]
    assert l0[0] != l0[2]                                        # the optimiser moved something
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (l0, l1)
    for k in p0:
        assert (p0[k] - p1[k]).abs().max().item() <= 1e-4 * max(1.0, p0[k].abs().max().item()), k


def _run_six(dev, graphed, accumulation, bf16_plm=False, adamw=False, swin_gradients="compute", pipeline=False):
    from facialmmt_amd import models
    from facialmmt_amd.config import default_args
    from facialmmt_amd.train_step import GraphedTargetStep, TargetStep
    import bench
    B, Lv = 2, 6
    # noise-free configuration: dropout and DropPath off; Gumbel-softmax at a huge temperature (the noise enters as g / tau)
    # with the importance threshold below 1/7 so that the filter branch (not the keep-everything fallback) runs
    cfg = default_args(get_vision_utt_max_lens=Lv, get_audio_utt_max_lens=24, trg_accumulation_steps=accumulation, plm_module=synth.make_standin_plm(),
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, crossmodal_attn_dropout_TA=0.0, crossmodal_attn_dropout_TA_V=0.0,
                       tau=1e5, FacialEmoImpor_threshold=0.1)
    cfg.compute_dtype = torch.float32
    swin = models.SwinForAffwildClassification(cfg)
    mm = models.MultiModalTransformerForClassification(cfg)
    synth.fill_state_dict(swin, seed=100)
    synth.fill_state_dict(mm, seed=200)
    for m in swin.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    swin.to(dev).train()
    mm.to(dev).train()
    args = types.SimpleNamespace(utts=B, frames=Lv, dtype="fp32", plm="roberta-large", input="float", resize="pil")
    batch = list(bench.synth_batch(args, dev, 0, cfg))
    batch[0] = batch[0] % 1000
    batch = tuple(batch)
    masters = None
    if bf16_plm:
        from facialmmt_amd.train_step import MasterWeights, step_parameters
        masters = MasterWeights(mm.roberta, torch.bfloat16)
        opt = torch.optim.SGD(step_parameters(mm, masters), lr=0.05)
    elif adamw:                                                  # graphed: FusedClipAdamW reading the model's own gradient tensors
        lr = torch.tensor(2e-3, device=dev) if graphed else 2e-3
        opt = torch.optim.AdamW(mm.parameters(), lr=lr, weight_decay=0.01, fused=True, capturable=graphed)
    else:
        opt = torch.optim.SGD(mm.parameters(), lr=0.05)
    if graphed:
        step = GraphedTargetStep(swin, mm, opt, None, cfg, batch, autocast_dtype=None, masters=masters, discarded_swin_gradients=swin_gradients, pipeline_swin=pipeline)
        assert step.text_stream is not None
        assert (step.fused is not None) == adamw
    else:
        step = TargetStep(swin, mm, opt, None, cfg, autocast_dtype=None, discarded_swin_gradients=swin_gradients)
    losses = []
    for i in range(6):
        if pipeline:                                             # the next step's Swin forward rides beside this step; none behind the last
            loss, kept = step(batch, next_batch=batch if i < 5 else None)
        else:
            loss, kept = step(batch)
        losses.append(float(loss))
    torch.cuda.synchronize()
    if masters is not None:
        for low, m in masters.pairs():                           # the module holds the rounded masters, the masters moved
            assert low.dtype == torch.bfloat16 and torch.equal(low, m.to(torch.bfloat16))
    bn = swin.swin.output_layer[3]
    return losses, {k: v.detach().clone() for k, v in mm.named_parameters()}, bn.running_mean.clone(), int(bn.num_batches_tracked), kept.clone()


@pytest.mark.parametrize("accumulation", [1, 2])
def test_whole_step_graphs_equal_eager_step(accumulation):
    """GraphedTargetStep (graph A: text branch forked onto a second stream || Swin, fusion, loss, backward into static flat
    gradient buffers; graph B: clip + optimizer + zero) against the eager single-stream TargetStep over six micro-steps:
    same losses, same updated parameters, same BatchNorm running statistics and step counter, same kept-frame mask --
    including the warm-up being undone exactly (parameters, buffers, optimizer state) before the capture."""
    dev = torch.device("cuda:0")
    l0, p0, rm0, nb0, k0 = _run_six(dev, False, accumulation)
    l1, p1, rm1, nb1, k1 = _run_six(dev, True, accumulation)
    assert l0[0] != l0[-1]                                       # the optimiser moved something
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (l0, l1)
    assert nb0 == nb1 == 6 and torch.equal(k0, k1) and float(k0.sum()) > 0
    assert (rm0 - rm1).abs().max().item() <= 1e-4 * max(1.0, rm0.abs().max().item())
    for k in p0:
        assert (p0[k] - p1[k]).abs().max().item() <= 1e-4 * max(1.0, p0[k].abs().max().item()), k


@pytest.mark.parametrize("accumulation", [1, 2])
def test_pipelined_swin_forward_walks_the_in_order_trajectory(accumulation):
    """GraphedTargetStep(pipeline_swin=True): Swin's forward of step i + 1 replays as its own graph on a second stream beside step i (two
    alternating sets of saved activations).  Nothing steps Swin in a target step, so losses, updated parameters, BatchNorm running statistics and
    the kept-frame mask must be those of the in-order eager step -- six micro-steps, noise-free configuration."""
    dev = torch.device("cuda:0")
    l0, p0, rm0, nb0, k0 = _run_six(dev, False, accumulation)
    l1, p1, rm1, nb1, k1 = _run_six(dev, True, accumulation, pipeline=True)
    assert l0[0] != l0[-1]
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (l0, l1)
    for k in p0:
        assert (p0[k] - p1[k]).abs().max().item() <= 1e-4 * max(1.0, p0[k].abs().max().item()), k
    assert nb0 == nb1 and torch.allclose(rm0, rm1, rtol=1e-5, atol=1e-6)
    assert torch.equal(k0, k1)


def test_pipelined_step_redoes_a_prefetch_that_an_auxiliary_step_invalidated():
    """a Swin parameter whose version moved between the prefetch and the step (what GraphedAuxStep's replay does) makes the step run Swin's
    forward again, in order; an untouched prefetch is used as it is.  (Counted at the launch of graph S: in the noise-free configuration the loss
    does not see Swin's weights -- Gumbel-softmax at tau = 1e5 --, so the loss cannot tell.)"""
    import types as _t
    from facialmmt_amd import models
    from facialmmt_amd.config import default_args
    from facialmmt_amd.train_step import GraphedTargetStep, _bump_versions
    import bench
    dev = torch.device("cuda:0")
    cfg = default_args(get_vision_utt_max_lens=6, get_audio_utt_max_lens=24, trg_accumulation_steps=1, plm_module=synth.make_standin_plm(),
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, crossmodal_attn_dropout_TA=0.0, crossmodal_attn_dropout_TA_V=0.0,
                       tau=1e5, FacialEmoImpor_threshold=0.1)
    cfg.compute_dtype = torch.float32
    swin = models.SwinForAffwildClassification(cfg)
    mm = models.MultiModalTransformerForClassification(cfg)
    synth.fill_state_dict(swin, seed=100)
    synth.fill_state_dict(mm, seed=200)
    swin.to(dev).train()
    mm.to(dev).train()
    args = _t.SimpleNamespace(utts=2, frames=6, dtype="fp32", plm="roberta-large", input="float", resize="pil")
    batch = list(bench.synth_batch(args, dev, 0, cfg))
    batch[0] = batch[0] % 1000
    batch = tuple(batch)
    step = GraphedTargetStep(swin, mm, torch.optim.SGD(mm.parameters(), lr=0.01), None, cfg, batch, autocast_dtype=None, pipeline_swin=True)
    launches = []
    inner = step._launch_swin
    step._launch_swin = lambda k, frames: (launches.append(k), inner(k, frames))[1]
    step(batch, next_batch=batch)
    assert launches == [0, 1]                                    # this batch's forward (nothing prefetched yet) + the prefetch
    step(batch, next_batch=batch)
    assert launches == [0, 1, 0]                                 # the prefetched set 1 was used; set 0 prefetched for the next step
    _bump_versions(list(swin.parameters()))                      # "an auxiliary step ran"
    step(batch)
    assert launches == [0, 1, 0, 0]                              # the prefetch is stale: forward redone in order, nothing prefetched
    other = tuple(t.clone() if torch.is_tensor(t) else t for t in batch)
    step(batch, next_batch=other)
    step(batch)                                                  # not the batch that was announced: redone as well
    torch.cuda.synchronize()
    assert launches == [0, 1, 0, 0, 1, 0, 0]


@pytest.mark.parametrize("graphed", [False, True])
def test_skipping_the_discarded_swin_backward_changes_nothing(graphed):
    """discarded_swin_gradients="skip" (train_step.SKIP_NOTE: the reference never reads Swin's target-step gradients): six steps with
    and without Swin's backward -- same losses, multimodal parameters, BatchNorm running statistics and kept-frame mask.  Equal to
    fp32 rounding, not bit for bit: without autograd Swin's forward takes the inference form of a few launches (no saved
    pre-activations), whose last bit can differ (first loss 2.6707118 vs 2.6707120)."""
    dev = torch.device("cuda:0")
    l0, p0, rm0, nb0, k0 = _run_six(dev, graphed, 1, swin_gradients="compute")
    l1, p1, rm1, nb1, k1 = _run_six(dev, graphed, 1, swin_gradients="skip")
    assert l0[0] != l0[-1]
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), (l0, l1)
    assert nb0 == nb1 == 6 and torch.equal(k0, k1)
    assert (rm0 - rm1).abs().max().item() <= 1e-5 * max(1.0, rm0.abs().max().item())
    for k in p0:
        assert (p0[k] - p1[k]).abs().max().item() <= 1e-5 * max(1.0, p0[k].abs().max().item()), k


def test_bf16_text_encoder_with_fp32_masters_tracks_fp32_run():
    """MasterWeights: text encoder in bf16 (no autocast, no per-step weight casts), optimizer on fp32 masters.  Against the
    all-fp32 graphed run: same kept frames, losses within bf16 tolerance; the module always holds bf16(master)."""
    dev = torch.device("cuda:0")
    l0, p0, _, _, k0 = _run_six(dev, True, 1)
    l1, p1, _, _, k1 = _run_six(dev, True, 1, bf16_plm=True)
    assert torch.equal(k0, k1)
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 3e-2 * max(1.0, abs(a)), (l0, l1)
    moved = [k for k in p0 if not k.startswith("roberta.") and (p0[k] - p1[k]).abs().max().item() > 5e-2 * max(1.0, p0[k].abs().max().item())]
    assert not moved, moved[:5]


def test_fused_clip_adamw_matches_torch():
    """train_step.FusedClipAdamW (fmmt_adamw_batch) against clip_grad_norm_ + torch.optim.AdamW over four steps with a moving
    learning rate: fp32 parameters of odd sizes and unaligned views, one of them with a bf16 twin; a step that clips and one
    that does not"""
    from facialmmt_amd.train_step import FusedClipAdamW
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    shapes = [(1024, 300), (7,), (4097,), (33, 5), (5000,)]
    base = torch.randn(20000, generator=g).to(dev)
    ref = [torch.nn.Parameter(torch.randn(sh, generator=g).to(dev)) for sh in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref[:-1]] + [torch.nn.Parameter(base[3:5003].detach())]   # a 12-byte-offset view
    with torch.no_grad():
        mine[-1].copy_(ref[-1])
    low = mine[0].detach().to(torch.bfloat16)
    grads = {p: torch.zeros_like(p) for p in mine}
    lr = torch.tensor(1e-2, device=dev)
    opt_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.05, betas=(0.9, 0.98), eps=1e-6)
    opt_mine = torch.optim.AdamW(mine, lr=lr, weight_decay=0.05, betas=(0.9, 0.98), eps=1e-6, fused=True, capturable=True)
    assert FusedClipAdamW.eligible(opt_mine, mine)
    fused = FusedClipAdamW(opt_mine, mine, grads, {id(mine[0]): low}, max_norm=1.0)
    for step in range(4):
        scale = 10.0 if step % 2 == 0 else 1e-3                  # clipped / not clipped
        for pr, pm in zip(ref, mine):
            gr = torch.randn(pr.shape, generator=g).to(dev) * scale
            pr.grad = gr.clone()
            grads[pm].copy_(gr)
        cur = 1e-2 * (step + 1) / 4
        for grp in opt_ref.param_groups:
            grp["lr"] = cur
        lr.fill_(cur)
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt_ref.step()
        fused.update()
    torch.cuda.synchronize()
    for pr, pm in zip(ref, mine):
        assert torch.allclose(pr, pm, rtol=2e-6, atol=2e-7), (pr - pm).abs().max()
    assert torch.equal(low, mine[0].detach().to(torch.bfloat16))
    fused.reset()
    assert float(fused.step) == 0 and all(float(m.abs().max()) == 0 for m in fused.m)


def test_hf_adamw_formula_and_fused_launch():
    """train_step.HFAdamW against the update transformers.AdamW performs (train.py:307,333 construct that class; its step,
    transformers 4.24 optimization.py, restated literally here: eps added before the bias correction, weight decay applied after
    the update, defaults eps 1e-6 / weight_decay 0), and FusedClipAdamW(HFAdamW) = fmmt_adamw_batch(hf_semantics=1) against
    clip_grad_norm_ + HFAdamW.step(); moving learning rate, a step that clips and one that does not, a bf16 twin."""
    import math
    from facialmmt_amd.train_step import FusedClipAdamW, HFAdamW
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes = [(512, 300), (7,), (4097,), (33, 5)]
    lit = [torch.randn(sh, generator=g).to(dev) for sh in shapes]                       # literal restatement, plain tensors
    eager = [torch.nn.Parameter(p.clone()) for p in lit]
    mine = [torch.nn.Parameter(p.clone()) for p in lit]
    low = mine[0].detach().to(torch.bfloat16)
    assert HFAdamW([torch.nn.Parameter(torch.zeros(1))]).defaults["eps"] == 1e-6 and HFAdamW([torch.nn.Parameter(torch.zeros(1))]).defaults["weight_decay"] == 0.0
    b1, b2, eps, wd = 0.9, 0.999, 1e-6, 0.01
    m_l = [torch.zeros_like(p) for p in lit]
    v_l = [torch.zeros_like(p) for p in lit]
    lr = torch.tensor(1e-2, device=dev)
    opt_e = HFAdamW(eager, lr=1e-2, weight_decay=wd)
    opt_m = HFAdamW(mine, lr=lr, weight_decay=wd)
    grads = {p: torch.zeros_like(p) for p in mine}
    assert FusedClipAdamW.eligible(opt_m, mine)
    fused = FusedClipAdamW(opt_m, mine, grads, {id(mine[0]): low}, max_norm=1.0)
    assert fused.hf
    for step in range(4):
        scale = 10.0 if step % 2 == 0 else 1e-3
        cur = 1e-2 * (step + 1) / 4
        gs = [torch.randn(sh, generator=g).to(dev) * scale for sh in shapes]
        for pe, pm, gr in zip(eager, mine, gs):
            pe.grad = gr.clone()
            grads[pm].copy_(gr)
        for grp in opt_e.param_groups:
            grp["lr"] = cur
        lr.fill_(cur)
        torch.nn.utils.clip_grad_norm_(eager, 1.0)
        # literal transformers.AdamW.step on the clipped gradients
        t = step + 1
        for p_, m_, v_, pe in zip(lit, m_l, v_l, eager):
            gr = pe.grad
            m_.mul_(b1).add_(gr, alpha=1.0 - b1)
            v_.mul_(b2).addcmul_(gr, gr, value=1.0 - b2)
            denom = v_.sqrt().add_(eps)
            step_size = cur * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
            p_.addcdiv_(m_, denom, value=-step_size)
            p_.add_(p_, alpha=-cur * wd)
        opt_e.step()
        fused.update()
    torch.cuda.synchronize()
    for pl, pe, pm in zip(lit, eager, mine):
        assert torch.allclose(pl, pe.detach(), rtol=2e-6, atol=2e-7), (pl - pe).abs().max()
        assert torch.allclose(pe, pm, rtol=2e-6, atol=2e-7), (pe - pm).abs().max()
    assert torch.equal(low, mine[0].detach().to(torch.bfloat16))


def test_fused_adamw_continues_from_a_loaded_hf_state_dict():
    """A transformers.AdamW-layout checkpoint loaded into a LIVE HFAdamW + FusedClipAdamW pair: load_hf_state_dict keeps the device lr
    tensor the fused update reads, FusedClipAdamW.load_from copies the moments and the counter into the buffers the (captured) update addresses;
    the next fused update equals the uninterrupted eager run's."""
    from facialmmt_amd.train_step import FusedClipAdamW, HFAdamW
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    shapes = [(300, 70), (4097,), (5,)]
    init = [torch.randn(sh, generator=g).to(dev) for sh in shapes]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    opt_r = HFAdamW(ref, lr=torch.tensor(4e-3, device=dev), weight_decay=0.01)
    gs = [[torch.randn(sh, generator=g).to(dev) * 0.01 for sh in shapes] for _ in range(4)]
    for k in range(3):
        for p, gr in zip(ref, gs[k]):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt_r.step()
    import copy
    sd = copy.deepcopy(opt_r.hf_state_dict())               # (as read from a file: Optimizer.load_state_dict does not copy same-dtype tensors)
    sd["param_groups"][0]["lr"] = 4e-3
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    lr = torch.tensor(9e-3, device=dev)
    opt_m = HFAdamW(mine, lr=lr, weight_decay=0.01)
    grads = {p: torch.zeros_like(p) for p in mine}
    fused = FusedClipAdamW(opt_m, mine, grads, {}, max_norm=1.0)
    for pm, gr in zip(mine, gs[0]):                          # the fused pair has stepped before the checkpoint arrives
        grads[pm].copy_(gr)
    fused.update()
    with torch.no_grad():
        for pm, pr in zip(mine, ref):
            pm.copy_(pr)
    opt_m.load_hf_state_dict(sd)
    assert opt_m.param_groups[0]["lr"] is lr and fused.lr is lr and FusedClipAdamW.eligible(opt_m, mine)
    fused.load_from(opt_m)
    assert float(fused.step) == 3.0
    for pr, pm, gr in zip(ref, mine, gs[3]):
        pr.grad = gr.clone()
        grads[pm].copy_(gr)
    torch.nn.utils.clip_grad_norm_(ref, 1.0)
    opt_r.step()
    fused.update()
    torch.cuda.synchronize()
    for pr, pm in zip(ref, mine):
        assert torch.allclose(pr, pm, rtol=2e-6, atol=2e-7), (pr - pm).abs().max()


@pytest.mark.parametrize("accumulation", [1, 2])
def test_fused_optimizer_in_the_graphed_step_equals_eager_adamw(accumulation):
    """GraphedTargetStep with an AdamW optimizer: graph B is one norm + fmmt_adamw_batch (clip + AdamW on the flat gradient
    buffers) -- against the eager TargetStep with clip_grad_norm_ + torch.optim.AdamW, six micro-steps, with and without
    gradient accumulation."""
    dev = torch.device("cuda:0")
    l0, p0, _, _, k0 = _run_six(dev, False, accumulation, adamw=True)
    l1, p1, _, _, k1 = _run_six(dev, True, accumulation, adamw=True)
    assert l0[0] != l0[-1] and torch.equal(k0, k1)
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 5e-4 * max(1.0, abs(a)), (l0, l1)
    # Parameters are held to a loose bound only: Adam turns every gradient into a step of about +-lr, so elements whose gradient is
    # rounding noise (embedding rows reached through atomics, biases behind a LayerNorm) walk apart by a few lr between two
    # runs of the SAME formula.  The formula itself is held to 2e-6 against torch in test_fused_clip_adamw_matches_torch; the six
    # losses above are the end-to-end check.
    for k in p0:
        assert (p0[k] - p1[k]).abs().max().item() <= 2 * 2e-3 * 6 + 1e-4, k


def test_replayed_benchmark_step_is_reproducible():
    """Consistency at the bench size (configs[1], bf16, text branch and parallel fusion on their own streams): the captured forward +
    backward graph replayed 12 times on the same inputs and generator state gives the same loss, kept-frame mask and flat gradient
    buckets bit for bit (tests/support_replay_step.py; round 3 found the stock input projections' bias gradients failing this)."""
    import subprocess, sys
    env = dict(os.environ, REPS="12")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "support_replay_step.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "12 replays, 0 with differences" in r.stdout, r.stdout[-2000:]


def test_measured_swin_backward_pieces_and_cut_choice():
    """train_step.measure_swin_tail_ms: the GPU time of Swin's backward below each cut, from HIP events in stage backward hooks -- positive, growing with
    the cut, smaller than the whole backward; pick_swin_cut takes the lowest cut whose piece holds the exchange (measured table or the profiled one)."""
    from facialmmt_amd import models
    from facialmmt_amd.config import default_args
    from facialmmt_amd.train_step import measure_swin_tail_ms, pick_swin_cut
    dev = torch.device("cuda:0")
    aff = models.SwinForAffwildClassification(default_args()).to(dev).train()
    frames = torch.randn(16, 3, 224, 224, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        t = measure_swin_tail_ms(aff, frames)
    assert set(t) == {0, 1, 2} and 0 < t[0] < t[1] < t[2]
    assert all(p.grad is None for p in aff.parameters())
    assert pick_swin_cut(t[0] * 0.5, 16, t) == 0 and pick_swin_cut((t[0] + t[1]) / 2, 16, t) == 1 and pick_swin_cut(t[2] * 2, 16, t) == 2
    assert pick_swin_cut(9.0) == 0 and pick_swin_cut(12.0) == 1 and pick_swin_cut(100.0) == 2          # the profiled table at 640 frames

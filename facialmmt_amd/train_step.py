"""One target-task training step (forward + backward + optimizer) of FacialMMT on the HIP path.

Restates the body of the reference's `multimodal_train` loop (train.py:54-143) without its host
synchronisations: the per-face importance filter and the emotion-concat (train.py:75-114) are
cumsum / searchsorted / index_put on the device (parity pinned against the literal
loop restatement in oracle/train_glue.py, reference quirk included)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def select_frames(preds, vision_inputs, vision_mask, num_imgs, threshold):
    """Device-side, sync-free version of train.py:75-114.

    preds (sumF, 7): per-face emotion distribution (Gumbel-softmax of the Swin logits), differentiable;
    vision_inputs (B, Lv, D); vision_mask (B, Lv); num_imgs (B,) int tensor of real frames per utterance.
    Faces with sum(p^2) > threshold are kept; each utterance packs its kept faces to the front
    (new mask = count of kept faces) together with the matching rows of vision_inputs; if no face at all
    passes, every real frame is kept.  Returns (vision_inputs_concat (B,Lv,D+7), new_vision_mask (B,Lv))."""
    B, Lv, D = vision_inputs.shape
    dev = preds.device
    nF = preds.shape[0]
    n = num_imgs.to(dev).long()
    importance = (preds * preds).sum(dim=1)                       # == diag(P P^T)
    sel = importance > threshold                                   # (nF,)
    g = torch.arange(nF, device=dev)
    # utterance boundaries with the reference's (n - 1) margin: b_u = sum_{i<=u} n_i - u
    csum = torch.cumsum(n, 0)
    upper = csum - torch.arange(B, device=dev)                     # exclusive upper bound of utterance u
    margin = upper - n                                             # sum_{i<u}(n_i - 1)
    u = torch.searchsorted(upper, g, right=True)                   # owning utterance (== B: dropped tail)
    owned = sel & (u < B)
    uc = u.clamp(max=B - 1)
    # rank of each selected face inside its utterance
    sel_cum = torch.cumsum(owned.long(), 0)
    first_of_utt = torch.cat((torch.zeros(1, dtype=torch.long, device=dev), upper[:-1])).clamp(max=nF)   # start index of utt u
    before = torch.where(first_of_utt > 0, sel_cum[(first_of_utt - 1).clamp(min=0)], torch.zeros_like(first_of_utt))
    k = (sel_cum - 1 - before[uc]).clamp(min=0, max=Lv - 1)
    w = owned.to(preds.dtype).unsqueeze(1)
    emo_sel = torch.zeros(B, Lv, preds.shape[1], dtype=preds.dtype, device=dev).index_put((uc, k), preds * w, accumulate=True)
    src = vision_inputs[uc, (g - margin[uc]).clamp(min=0, max=Lv - 1)]                                   # (nF, D)
    inp_sel = torch.zeros_like(vision_inputs).index_put((uc, k), src * owned.to(vision_inputs.dtype).unsqueeze(1), accumulate=True)
    cnt = torch.zeros(B, dtype=torch.long, device=dev).index_put((uc,), owned.long(), accumulate=True)
    mask_sel = (torch.arange(Lv, device=dev).view(1, Lv) < cnt.view(B, 1)).to(vision_mask.dtype)
    # fallback branch (no face passes): sequential fill over the real frames, masks unchanged
    real = torch.cumsum(vision_mask, dim=1) == torch.arange(1, Lv + 1, device=dev).view(1, Lv)          # leading run of ones
    real_cnt = real.long().sum(1)
    offs = torch.cumsum(real_cnt, 0) - real_cnt
    idx_all = (offs.view(B, 1) + torch.arange(Lv, device=dev).view(1, Lv)).clamp(max=nF - 1)
    emo_all = preds[idx_all] * real.unsqueeze(-1).to(preds.dtype)
    any_sel = sel.any()
    emo = torch.where(any_sel, emo_sel, emo_all)
    inputs = torch.where(any_sel, inp_sel, vision_inputs)
    mask = torch.where(any_sel, mask_sel, vision_mask)
    return torch.cat((inputs, emo.to(inputs.dtype)), dim=-1), mask


def graph_multimodal(mm, sample_args, autocast_dtype=None):
    """Capture forward and backward of the multimodal model as HIP graphs (torch.cuda.make_graphed_callables):
    its ~4000 small launches per step (24 PLM layers, 7 self-attention layers, 8 cross-modal layer calls) are
    host-bound when issued one by one (measured: 112 ms of host time per step against 105 ms of GPU work).
    Everything on that path is capture-safe: no host synchronisation, dropout seeds drawn on the device,
    bf16 weight shadows re-cast inside the graph.  Shapes are static (fixed synthetic batch).
    Returns the module (its forward now replays the graphs)."""
    import contextlib
    ctx = torch.autocast("cuda", dtype=autocast_dtype, cache_enabled=False) if autocast_dtype is not None else contextlib.nullcontext()
    with ctx:
        return torch.cuda.make_graphed_callables(mm, tuple(sample_args), num_warmup_iters=3)


class TargetStep:
    """Swin (train mode, Gumbel-softmax head) -> frame filter -> multimodal model -> CE -> backward ->
    (every `accumulation_steps`) clip + AdamW + schedule, as train.py:46-143.  Only the multimodal
    optimizer steps here; Swin receives gradients through the emotion features and is updated by the
    auxiliary task's optimizer (train.py:31), so its gradients are dropped after each step."""

    def __init__(self, swin_model, multimodal_model, optimizer, scheduler, args, autocast_dtype=None, ddp_model=None):
        self.swin = swin_model
        self.mm = multimodal_model
        self.mm_call = ddp_model if ddp_model is not None else multimodal_model
        self.opt = optimizer
        self.sched = scheduler
        self.args = args
        self.autocast_dtype = autocast_dtype
        self.i_batch = 0
        self.host_ms = {}          # cumulative host-side enqueue time per phase (no device sync)

    def __call__(self, batch):
        (ids, attn_mask, sep_mask, audio, audio_mask, vision_inputs, vision_mask, labels, frames, num_imgs, utt_idx) = batch
        import time
        args = self.args
        t = [time.perf_counter()]

        def mark(name):
            t.append(time.perf_counter())
            self.host_ms[name] = self.host_ms.get(name, 0.0) + (t[-1] - t[-2]) * 1e3
        preds = self.swin(frames, is_trg_task=True)                                  # (sumF, 7), Gumbel-softmax
        mark("swin_fwd")
        vis_concat, new_mask = select_frames(preds.float(), vision_inputs, vision_mask, num_imgs, args.FacialEmoImpor_threshold)
        mark("frame_filter")
        if self.autocast_dtype is not None:
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                logits = self.mm_call(ids, attn_mask, sep_mask, audio, audio_mask, vis_concat, new_mask, utt_idx)
        else:
            logits = self.mm_call(ids, attn_mask, sep_mask, audio, audio_mask, vis_concat, new_mask, utt_idx)
        mark("multimodal_fwd")
        loss = F.cross_entropy(logits.float(), labels) / args.trg_accumulation_steps
        loss.backward()
        mark("backward")
        self.i_batch += 1
        if self.i_batch % args.trg_accumulation_steps == 0:
            torch.nn.utils.clip_grad_norm_(self.mm.parameters(), args.clip)
            mark("clip")
            self.opt.step()
            if self.sched is not None:
                self.sched.step()
            self.opt.zero_grad(set_to_none=True)
            mark("optimizer")
        self.swin.zero_grad(set_to_none=True)
        return loss.detach(), new_mask


class AuxStep:
    """One auxiliary-task (Aff-Wild2 frame classification) step, train.py:15-41: Swin -> logits (no Gumbel)
    -> cross-entropy -> backward -> clip (0.8) -> AdamW on the Swin model (aux_lr 5e-5) every
    `aux_accumulation_steps`.  BASELINE.json configs[4] alternates these steps with target steps per epoch."""

    def __init__(self, swin_model, optimizer, scheduler, args):
        self.swin, self.opt, self.sched, self.args = swin_model, optimizer, scheduler, args
        self.i_batch = 0

    def __call__(self, images, labels):
        loss = self.swin(images, False, labels, F.cross_entropy) / self.args.aux_accumulation_steps
        loss.backward()
        self.i_batch += 1
        if self.i_batch % self.args.aux_accumulation_steps == 0:
            torch.nn.utils.clip_grad_norm_(self.swin.parameters(), self.args.clip)
            self.opt.step()
            if self.sched is not None:
                self.sched.step()
            self.opt.zero_grad(set_to_none=True)
        return loss.detach()

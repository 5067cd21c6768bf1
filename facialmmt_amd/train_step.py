"""One target-task training step (forward + backward + optimizer) of FacialMMT on the HIP path.

Restates the body of the reference's `multimodal_train` loop (train.py:54-143) without its host
synchronisations: the per-face importance filter and the emotion-concat (train.py:75-114) are
cumsum / searchsorted / index_put on the device (parity pinned against the literal
loop restatement in oracle/train_glue.py, reference quirk included)."""
from __future__ import annotations

import types

import torch
import torch.nn.functional as F


SELECT_FRAMES_KERNEL = True      # module constant (tests patch it): False = the torch restatement below (~55 launches)


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
    if SELECT_FRAMES_KERNEL and torch.is_tensor(num_imgs):
        from . import ops
        if ops.select_frames_fusable(preds, vision_inputs, vision_mask):
            return ops.select_frames(preds, vision_inputs, vision_mask, num_imgs, threshold)      # one launch (csrc/frame_filter.hip): same index arithmetic
    n = torch.as_tensor(num_imgs).to(dev).long()
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


def pick_concurrent_stream(device, candidates: int = 8, cycles: int = 4_000_000):
    """A HIP stream that really runs concurrently with the current one.  HIP multiplexes streams onto a handful of
    hardware queues (4 by default) and two streams that share a queue serialise; which queue a new stream lands on
    depends on how many streams the process created before (RCCL, for one, creates several at process-group
    initialisation -- measured: the text-encoder overlap vanished in every run that had called init_process_group).
    So measure it: spin kernels on both streams, keep the first candidate whose pair finishes in about the time
    of one.  Returns (stream, ratio) with ratio = t(pair) / t(single); falls back to the best candidate."""
    import time
    main = torch.cuda.current_stream(device)

    def timed(fn):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0
    torch.cuda._sleep(cycles)                              # warm up the spin kernel
    single = min(timed(lambda: torch.cuda._sleep(cycles)) for _ in range(3))
    best, best_ratio, keep = None, float("inf"), []
    for _ in range(candidates):
        s = distinct_stream(device, keep)
        keep.append(s)                                     # keep candidates alive so that the next one is a new stream

        def pair():
            s.wait_stream(main)
            torch.cuda._sleep(cycles)
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        ratio = min(timed(pair) for _ in range(2)) / single
        if ratio < best_ratio:
            best, best_ratio = s, ratio
        if ratio < 1.3:
            break
    return best, best_ratio


def distinct_stream(device, avoid=()):
    """A side stream that is a different HIP stream from every stream in `avoid` and from the current one.
    torch.cuda.Stream() hands out entries of a per-device pool of 32 round-robin: in a process that has created a few dozen
    streams a "new" stream can BE the capture stream or another branch's stream, and a fork / join between a stream and
    itself inside a graph capture has crashed the HIP runtime (intermittent segmentation fault in the capture of a step when
    the whole GPU test suite ran in one process).  So: compare the raw handles and keep drawing."""
    taken = {s.cuda_stream for s in avoid if s is not None} | {torch.cuda.current_stream(device).cuda_stream}
    keep = []
    for _ in range(64):
        s = torch.cuda.Stream(device=device)
        if s.cuda_stream not in taken:
            return s
        keep.append(s)
    raise RuntimeError("distinct_stream: the stream pool only returns streams that are already in use")


# Captured graphs are never destroyed.  On this ROCm (7.0 runtime under torch 2.10) tearing down HIP graphs that were captured
# with forked streams is what the intermittent crashes of a long-lived process traced back to: destroyed by a garbage
# collection during a later capture -> abort inside the capture; destroyed right before the next capture -> segmentation
# fault in that graph's first replay.  A training process captures a handful of graphs; holding on to them costs nothing.
_KEEP_GRAPHS = []


class capture_window:
    """Garbage collection fenced off a graph capture: collect NOW (cycles left by earlier steps may own HIP graphs, streams
    and pool memory whose destructors call into the HIP runtime), then keep the cyclic collector off until the capture ends.
    torch.cuda.graph stopped collecting on entry (torch >= 2.9 only does with torch.compiler.config.force_cudagraph_gc), and a
    collection that fires in the middle of a capture destroys such objects while the stream is capturing: measured here as an
    intermittent abort / segmentation fault of the process (faulthandler: "Garbage-collecting" inside the capture of a step
    that followed other graph-capturing steps), two runs in five of the whole GPU suite."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


class _Branch(torch.nn.Module):
    """One branch of the multimodal forward as a module of its own, so that torch.cuda.make_graphed_callables sees
    exactly the parameters it uses.  Shares the owner's sub-modules (same Parameter objects); never registered on the
    owner, so state_dict keys are untouched."""

    def __init__(self, owner, fn_name, submodules):
        super().__init__()
        self._fn = getattr(owner, fn_name)
        for i, m in enumerate(submodules):
            self.add_module(f"m{i}", m)

    def forward(self, *args):
        return self._fn(*args)


def graph_multimodal(mm, sample_args, autocast_dtype=None, overlap_text=True, parallel_fusion=False):
    """Capture forward and backward of the multimodal model as HIP graphs (torch.cuda.make_graphed_callables):
    its ~4000 small launches per step (24 PLM layers, 7 self-attention layers, 8 cross-modal layer calls) are
    host-bound when issued one by one (measured: 112 ms of host time per step against 105 ms of GPU work).
    Everything on that path is capture-safe: no host synchronisation, dropout seeds drawn on the device,
    bf16 weight shadows re-cast inside the graph.  Shapes are static (fixed synthetic batch).

    Two graphs are captured: the text branch (PLM -> text_linear -> slicing) and the fusion branch.  With
    `overlap_text` the text branch replays on a second HIP stream (`mm.text_stream`): it does not depend on the
    visual path, and its small launches fill the CUs that Swin's large launches leave idle between waves
    (measured on an MI355X: text encoder 28.4 ms + Swin 52.9 ms = 82.8 ms back to back, 66.9 ms concurrently).
    Returns the module (its forward now replays the graphs)."""
    import contextlib
    import os
    if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "") != "0":
        raise RuntimeError("graph_multimodal: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in the environment before the HIP runtime "
                           "initialises (import facialmmt_amd before the first CUDA call, or export it): with ROCm 7.0's packet "
                           "capture, gradients of replayed graphs are wrong from the third replay on")
    ids, attn_mask, sep_mask, audio, audio_mask, vision, vision_mask, utt_idx = sample_args
    utt_idx = torch.as_tensor(utt_idx, device=ids.device)
    plm = mm.roberta if mm.text_pretrained_model == 'roberta' else mm.bert
    text = _Branch(mm, "text_branch", [plm, mm.text_linear])
    fusion = _Branch(mm, "fusion_branch", [mm.audio_linear, mm.audio_utt_transformer, mm.vision_linear, mm.vision_utt_transformer,
                                           mm.attention, mm.CrossModalTrans_TA, mm.CrossModalTrans_TA_V, mm.dropout, mm.classifier])
    B = ids.shape[0]
    act = autocast_dtype if autocast_dtype is not None else torch.float32
    text_feat = torch.zeros(B, mm.get_text_utt_max_lens, mm.hidden_size, device=ids.device, dtype=act, requires_grad=True)
    text_mask = torch.ones(B, mm.get_text_utt_max_lens, device=ids.device)
    side = None
    if overlap_text:
        # The gradient accumulators of the text branch's parameters must live on the text stream too: autograd runs
        # AccumulateGrad on the stream that was current when the node was created, and a node created on the main
        # stream would make the main stream wait for the whole text backward before Swin's backward could start.
        # Create them now, under the text stream, and keep them alive (DDP, wrapped later, finds these same nodes).
        side, ratio = pick_concurrent_stream(ids.device)
        mm.text_stream_concurrency = ratio                 # ~1.0: truly concurrent with the main stream; ~2.0: serialised
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            mm._text_grad_accumulators = [p.expand_as(p).grad_fn.next_functions[0][0] for p in text.parameters() if p.requires_grad]
        torch.cuda.current_stream().wait_stream(side)
    # independent halves of the fusion stack (audio / vision encoder, the two directions of each cross-modal encoder) are
    # captured as parallel branches of the fusion graphs: models._pair
    mm.pair_stream = distinct_stream(ids.device, (side,)) if parallel_fusion else None
    ctx = torch.autocast("cuda", dtype=autocast_dtype, cache_enabled=False) if autocast_dtype is not None else contextlib.nullcontext()
    with ctx, capture_window():
        gtext, gfusion = torch.cuda.make_graphed_callables(
            (text, fusion),
            ((ids, attn_mask, sep_mask, utt_idx), (text_feat, text_mask, audio, audio_mask, vision, vision_mask)),
            num_warmup_iters=3)
    # plain attributes, NOT child modules: the branch wrappers must not show up in mm.state_dict() / named_parameters()
    mm.pair_stream = None                                     # eager calls of the branches stay single-stream
    object.__setattr__(mm, "_text_call", gtext)
    object.__setattr__(mm, "_fusion_call", gfusion)
    _KEEP_GRAPHS.append((gtext, gfusion))
    mm.text_stream = side
    return mm


# The reference back-propagates the target-task loss through Swin (the emotion features are written into the vision tensor by a
# differentiable copy, train.py:101-104,121), but nothing ever reads those gradients: the target step clips and steps the multimodal
# model only (train.py:140-143), and Swin's optimizer zeroes them before its next use (train.py:20,33).  With "skip" the Swin forward
# of the target step runs without autograd (its fused kernels then also skip the saved-activation stores) and no Swin backward runs:
# the parameters, BatchNorm statistics and losses of every later step are the same to fp32 rounding (tests/test_gpu_train_step.py: the
# inference form of a few forward launches differs in the last bit), at 60 % of the step time (40.5 against 66.1 ms).  It is an OPTION: the default executes what the reference executes, and bench.py's `value` is measured with it.
SKIP_NOTE = "Swin's target-step gradients are never read (train.py:20,33,140-143): 'skip' does not compute them"


class TargetStep:
    """Swin (train mode, Gumbel-softmax head) -> frame filter -> multimodal model -> CE -> backward ->
    (every `accumulation_steps`) clip + AdamW + schedule, as train.py:46-143.  Only the multimodal
    optimizer steps here; Swin receives gradients through the emotion features and is updated by the
    auxiliary task's optimizer (train.py:31), so its gradients are dropped after each step."""

    def __init__(self, swin_model, multimodal_model, optimizer, scheduler, args, autocast_dtype=None, ddp_model=None, averager=None,
                 discarded_swin_gradients="compute"):
        """Data parallel: pass `averager` (parallel.GradientAverager over the multimodal parameters) or, alternatively,
        `ddp_model` (the module wrapped by torch's DistributedDataParallel).
        `discarded_swin_gradients`: "compute" (default, what the reference executes) or "skip" -- see SKIP_NOTE."""
        if discarded_swin_gradients not in ("compute", "skip"):
            raise ValueError("discarded_swin_gradients: 'compute' or 'skip'")
        self.skip_swin_bwd = discarded_swin_gradients == "skip"
        self.swin = swin_model
        self.mm = multimodal_model
        self.mm_call = ddp_model if ddp_model is not None else multimodal_model
        self.exchange = averager if averager is not None else ddp_model
        self.opt = optimizer
        self.sched = scheduler
        self.args = args
        self.autocast_dtype = autocast_dtype
        self.i_batch = 0
        self.host_ms = {}          # cumulative host-side enqueue time per phase (no device sync)

    def start_epoch(self):
        """As the reference at the top of each epoch (train.py:52,54): gradients zeroed, micro-batch count restarted, so a
        partial accumulation window at the end of an epoch (len(loader) % trg_accumulation_steps != 0) is dropped, not
        carried into the next epoch."""
        self.i_batch = 0
        averager = self.exchange if hasattr(self.exchange, "zero_grad") and not isinstance(self.exchange, torch.nn.Module) else None
        if averager is not None:
            averager.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
        self.swin.zero_grad(set_to_none=True)

    def __call__(self, batch):
        (ids, attn_mask, sep_mask, audio, audio_mask, vision_inputs, vision_mask, labels, frames, num_imgs, utt_idx) = batch
        import time
        args = self.args
        t = [time.perf_counter()]

        evs = getattr(self, "gpu_events", None)          # optional: list collecting (phase, event) on the main stream

        def mark(name):
            t.append(time.perf_counter())
            self.host_ms[name] = self.host_ms.get(name, 0.0) + (t[-1] - t[-2]) * 1e3
            if evs is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append((name, e))
        if evs is not None:
            mark("start")
        from .parallel import GradientAverager, accumulate
        self.i_batch += 1
        last = self.i_batch % args.trg_accumulation_steps == 0
        if getattr(self.mm, "text_stream", None) is not None:
            # start the text branch on its own stream before the Swin forward is enqueued (models.launch_text)
            if self.autocast_dtype is not None:
                with torch.autocast("cuda", dtype=self.autocast_dtype):
                    self.mm.launch_text(ids, attn_mask, sep_mask, utt_idx)
            else:
                self.mm.launch_text(ids, attn_mask, sep_mask, utt_idx)
        if self.skip_swin_bwd:
            with torch.no_grad():
                preds = self.swin(frames, is_trg_task=True)
        else:
            preds = self.swin(frames, is_trg_task=True)                              # (sumF, 7), Gumbel-softmax
        mark("swin_fwd")
        vis_concat, new_mask = select_frames(preds.float(), vision_inputs, vision_mask, num_imgs, args.FacialEmoImpor_threshold)
        mark("frame_filter")
        # gradient exchange only on the last micro-step of the window; torch's DDP decides in its FORWARD whether the
        # coming backward synchronises, so the no_sync context has to cover the forward as well
        with accumulate(self.exchange, last):
            if self.autocast_dtype is not None:
                with torch.autocast("cuda", dtype=self.autocast_dtype):
                    logits = self.mm_call(ids, attn_mask, sep_mask, audio, audio_mask, vis_concat, new_mask, utt_idx)
            else:
                logits = self.mm_call(ids, attn_mask, sep_mask, audio, audio_mask, vis_concat, new_mask, utt_idx)
            mark("multimodal_fwd")
            loss = F.cross_entropy(logits.float(), labels) / args.trg_accumulation_steps
            loss.backward()
        mark("backward")
        if last:
            averager = self.exchange if isinstance(self.exchange, GradientAverager) else None
            if averager is not None:
                averager.finish()
            torch.nn.utils.clip_grad_norm_(self.mm.parameters(), args.clip)
            mark("clip")
            self.opt.step()
            if self.sched is not None:
                self.sched.step()
            if averager is not None:
                averager.zero_grad()                         # the .grad views into the flat buckets must survive
            else:
                self.opt.zero_grad(set_to_none=True)
            mark("optimizer")
        self.swin.zero_grad(set_to_none=True)
        return loss.detach(), new_mask


class VendorLinear(torch.nn.Linear):
    """nn.Linear of a low-precision text encoder with ops.VendorLinearFn behind it (same parameters, same state_dict keys)"""

    def forward(self, x):
        if self.bias is None or not x.is_cuda or x.dtype != self.weight.dtype or x.dtype not in (torch.bfloat16, torch.float32) \
                or self.weight.shape[0] % 8 or not torch.is_grad_enabled():
            return super().forward(x)
        from . import ops
        return ops.VendorLinearFn.apply(x, self.weight, self.bias)


class VendorLayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm of a bf16 text encoder: torch's forward, fmmt_layernorm_bwd_bf16 backward (ops.PlmLayerNormFn)"""

    def forward(self, x):
        if (self.weight is None or self.bias is None or not x.is_cuda or x.dtype != torch.bfloat16 or self.weight.dtype != torch.bfloat16
                or len(self.normalized_shape) != 1 or x.shape[-1] % 8 or x.shape[-1] > 2048 or not torch.is_grad_enabled()):
            return super().forward(x)
        from . import ops
        return ops.PlmLayerNormFn.apply(x, self.weight, self.bias, self.eps)


# Module constants (the probes patch them), not environment switches: each is the off-switch of one fused launch family
PLM_COLSUM, PLM_LN, FUSED_ADAMW, PIN_SHADOWS, FUSED_HANDOVER = True, True, True, True, True
PLM_FUSE_EMBED = True                                       # nn.Embedding tables: the weight gradient by fmmt_embedding_bwd (False: torch's embedding backward)
PLM_FUSE_FFN = True                                         # intermediate + output of a layer as one autograd node (ops.PlmFfnFn); False: the two modules
PLM_FUSE_ATTN = True                                        # a packed *SelfAttention runs its attention core on fmmt_mha_fwd / _bwd (False: the stock attention interface)
PLM_FUSE_TAILS, PLM_FUSE_QKV = True, True                  # fuse_text_encoder: sublayer tails as one launch per direction / packed query-key-value GEMM


def use_colsum_bias_gradients(module):
    """Re-class every nn.Linear of `module` (a Hugging Face text encoder) as VendorLinear: the bias gradients of its ~146 Linear
    layers then cost one 4 us launch each instead of a memset + a multi-block reduction (measured: 3.2 + 0.8 ms per step).
    PLM_COLSUM = False (module constant) leaves the module alone.  Returns the number of layers changed."""
    if not PLM_COLSUM:
        return 0
    n = 0
    ln = PLM_LN
    for m in module.modules():
        if type(m) is torch.nn.Linear and m.bias is not None:
            m.__class__ = VendorLinear
            n += 1
        elif ln and type(m) is torch.nn.LayerNorm and m.elementwise_affine and m.bias is not None:
            m.__class__ = VendorLayerNorm                    # two backward launches instead of three (fmmt_layernorm_bwd_bf16)
    return n


class _SeedBox:
    """one device int64 word per forward of the text encoder, drawn inside the step (graph-replay safe: torch's captured generator) by a forward
    pre-hook; every fused sublayer reads it and adds its own salt"""

    def __init__(self):
        self.t = None
        self.words = 1            # word 0: the sublayer tails (each adds its salt); word 1 + i: attention module i (its kernels take no salt)
        self.key_bias = None      # (the 4-d mask the layers were handed, its version counter, its per-key logit bias): _fused_attention

    def draw(self, device):
        self.t = torch.randint(0, 2 ** 62, (self.words,), dtype=torch.int64, device=device)


def _sublayer_tail_forward(self, hidden_states, input_tensor):
    """forward of a re-classed transformers *SelfOutput / *Output: dense + dropout + residual + LayerNorm = the vendor GEMM + one launch"""
    d, ln = self.dense, self.LayerNorm
    p = float(self.dropout.p) if self.training else 0.0
    box = self._fmmt_seed
    if (not hidden_states.is_cuda or hidden_states.dtype != torch.bfloat16 or d.weight.dtype != torch.bfloat16 or ln.weight.dtype != torch.bfloat16
            or d.bias is None or ln.bias is None or d.weight.shape[0] % 8 or d.weight.shape[0] > 2048 or not torch.is_grad_enabled()
            or (p > 0.0 and box.t is None)):
        return self._fmmt_stock_forward(hidden_states, input_tensor)
    from . import ops
    return ops.PlmSublayerTailFn.apply(hidden_states, input_tensor, d.weight, d.bias, ln.weight, ln.bias, ln.eps, p,
                                       box.t if p > 0.0 else 0, self._fmmt_salt)


def _fused_ffn_chunk(self, attention_output):
    """feed_forward_chunk of a re-classed transformers *Layer: intermediate + output as one autograd node (ops.PlmFfnFn), or the stock pair of modules"""
    inter, out = self.intermediate, self.output
    d1, d2, ln = inter.dense, out.dense, out.LayerNorm
    p = float(out.dropout.p) if self.training else 0.0
    box = out._fmmt_seed
    x = attention_output
    if (not PLM_FUSE_FFN or not x.is_cuda or x.dtype != torch.bfloat16 or not torch.is_grad_enabled() or d1.bias is None or d2.bias is None or ln.bias is None
            or any(t.dtype != torch.bfloat16 for t in (d1.weight, d2.weight, ln.weight)) or d2.weight.shape[0] % 8 or d2.weight.shape[0] > 2048
            or d1.weight.shape[0] % 8 or d1.weight.shape[0] > 8192 or d1.weight.shape[1] != d2.weight.shape[0] or (p > 0.0 and box.t is None)):
        return self._fmmt_stock_ffn(attention_output)
    from . import ops
    return ops.PlmFfnFn.apply(x, d1.weight, d1.bias, d2.weight, d2.bias, ln.weight, ln.bias, ln.eps, p, box.t if p > 0.0 else 0, out._fmmt_salt)


def _fused_embedding_forward(self, input):
    """forward of a re-classed nn.Embedding of the text encoder: the stock lookup, the weight gradient by fmmt_embedding_bwd (ops.PlmEmbeddingFn)"""
    w = self.weight
    if (not PLM_FUSE_EMBED or not input.is_cuda or w.dtype != torch.bfloat16 or not torch.is_grad_enabled() or not w.requires_grad or self.max_norm is not None
            or self.scale_grad_by_freq or self.sparse or w.shape[1] % 8 or w.shape[1] > 2048 or input.numel() > 32768 or input.numel() == 0
            or input.dtype not in (torch.int64, torch.int32)):
        return self._fmmt_stock_forward(input)
    from . import ops
    return ops.PlmEmbeddingFn.apply(input, w, self.padding_idx)


def _is_exact_gelu(fn):
    """transformers' ACT2FN["gelu"] (GELUActivation over torch's erf gelu) or torch.nn.functional.gelu / nn.GELU() itself"""
    if fn is torch.nn.functional.gelu:
        return True
    if isinstance(fn, torch.nn.GELU):
        return fn.approximate == "none"
    return type(fn).__name__ == "GELUActivation" and getattr(fn, "act", None) is torch.nn.functional.gelu


def _packed_qkv_forward(self, hidden_states, *args, **kwargs):
    """forward of a re-classed transformers *SelfAttention: query / key / value as one GEMM over the packed weight they are slices of; everything
    behind the projections (head split, the attention interface, dropout) is the stock method's, fed through three pass-through Linear stand-ins"""
    w, b = self._fmmt_qkv
    q, k, v = self.query, self.key, self.value
    ok = (hidden_states.is_cuda and hidden_states.dtype == w.dtype and torch.is_grad_enabled() and q.weight.data_ptr() == w.data_ptr()
          and k.weight.data_ptr() == w.data_ptr() + w[0].numel() * w.shape[0] // 3 * w.element_size()
          and v.weight.data_ptr() == w.data_ptr() + 2 * w[0].numel() * w.shape[0] // 3 * w.element_size() and q.bias.data_ptr() == b.data_ptr())
    # key / value from another tensor (cross-attention) or a cache: the packed projection of hidden_states would silently be the wrong one
    # (round-5 ADVICE).  transformers passes these by keyword or as the positional slots behind attention_mask / head_mask
    cross = (getattr(self, "is_cross_attention", False) or any(kwargs.get(k) is not None for k in ("encoder_hidden_states", "past_key_value", "past_key_values"))
             or any(torch.is_tensor(a) and a.dim() == 3 and a.shape[-1] == hidden_states.shape[-1] for a in args[2:]))
    if not ok or cross:
        return self._fmmt_stock_forward(hidden_states, *args, **kwargs)
    from . import ops
    fused = _fused_attention(self, hidden_states, args, kwargs, w, b)
    if fused is not None:
        return fused
    yq, yk, yv = ops.PlmQkvFn.apply(hidden_states, q.weight, k.weight, v.weight, q.bias, k.bias, v.bias, w, b)
    saved = (self.query, self.key, self.value)
    try:                                                     # the stock forward calls self.query(x) ...: hand it the projections already made
        self.__dict__["_modules"] = dict(self._modules, query=_Fixed(yq), key=_Fixed(yk), value=_Fixed(yv))
        return self._fmmt_stock_forward(hidden_states, *args, **kwargs)
    finally:
        self.__dict__["_modules"] = dict(self._modules, query=saved[0], key=saved[1], value=saved[2])


def _fused_attention(self, hidden_states, args, kwargs, w, b):
    """the attention core of a packed *SelfAttention on the in-tree kernels (ops.PlmSelfAttnFn), or None where they do not apply (then: the stock attention
    interface behind the packed projection).  Encoder self-attention only: bf16, head_dim 64, no cache, no head mask, no attention weights asked for; the
    mask transformers hands down -- None, or the 4-d form of a (batch, keys) padding mask, boolean (sdpa) or additive (eager), whose rows are all alike for a
    bidirectional encoder -- becomes the kernels' per-key logit bias from its first query row."""
    if not PLM_FUSE_ATTN or hidden_states.dim() != 3 or hidden_states.dtype != torch.bfloat16:
        return None
    nH = int(getattr(self, "num_attention_heads", 0))
    E = w.shape[1]
    if nH <= 0 or E % nH or E // nH != 64 or getattr(self, "is_decoder", False) or getattr(self, "is_causal", False):
        return None
    if getattr(self, "position_embedding_type", "absolute") not in (None, "absolute"):
        return None                                         # relative-position variants add a term to the scores
    if kwargs.get("output_attentions") or kwargs.get("head_mask") is not None or any(a is not None for a in args[1:]):
        return None
    mask = args[0] if args else kwargs.get("attention_mask")
    B, S, _ = hidden_states.shape
    key_bias = None
    if mask is not None:
        if not torch.is_tensor(mask) or mask.dim() != 4 or mask.shape[0] != B or mask.shape[1] != 1 or mask.shape[-1] != S or mask.shape[2] not in (1, S):
            return None
        # once per forward of the encoder, not once per layer: every layer is handed the same mask tensor (the box keeps it, so its identity cannot recur)
        cached = getattr(self._fmmt_seed, "key_bias", None)
        if cached is not None and cached[0] is mask and cached[1] == mask._version:
            key_bias = cached[2]
        else:
            row = mask[:, 0, 0, :]
            key_bias = (torch.where(row, 0.0, -30000.0) if row.dtype == torch.bool else row.float().clamp_min(-30000.0)).to(torch.float32).contiguous()
            self._fmmt_seed.key_bias = (mask, mask._version, key_bias)
    p = float(self.dropout.p) if self.training else 0.0
    box = getattr(self, "_fmmt_seed", None)
    if p > 0.0 and (box is None or box.t is None or box.t.numel() <= self._fmmt_attn_word):
        return None
    seed = box.t[self._fmmt_attn_word:self._fmmt_attn_word + 1] if p > 0.0 else 0
    scale = float(getattr(self, "scaling", 64 ** -0.5))
    from . import ops
    q, k, v = self.query, self.key, self.value
    out = ops.PlmSelfAttnFn.apply(hidden_states, q.weight, k.weight, v.weight, q.bias, k.bias, v.bias, w, b, nH, scale, p, seed, key_bias)
    return out, None


class _Fixed(torch.nn.Module):
    def __init__(self, y):
        super().__init__()
        self.y = y

    def forward(self, x):
        return self.y


def fuse_text_encoder(plm, tails: bool = True, qkv: bool = True):
    """Launch diet for a bf16 Hugging Face BERT / RoBERTa encoder (src/models.py:75-91), call it AFTER the module has its final dtype / device
    (MasterWeights): (a) every *SelfOutput / *Output runs dense + dropout + residual + LayerNorm as the vendor GEMM + one fused launch, backward one
    launch + a reduction + the two GEMMs (ops.PlmSublayerTailFn); (b) every *SelfAttention computes query / key / value with ONE GEMM over a packed
    (3E, E) weight of which the three nn.Linear parameters become row slices (same Parameter objects, same names, same state_dict; the optimizer's
    views keep working) and, where the in-tree attention kernels apply (_fused_attention: bf16, head_dim 64, encoder self-attention), its attention core
    with them on the packed projection (PLM_FUSE_ATTN); (c) where a layer's *Output took (a) and its *Intermediate is dense -> exact GELU, the two run as one
    autograd node (ops.PlmFfnFn, PLM_FUSE_FFN; the count is left in plm._fmmt_fused_ffn); (d) the nn.Embedding tables keep their forward and take
    fmmt_embedding_bwd for the weight gradient (ops.PlmEmbeddingFn, PLM_FUSE_EMBED).  Returns (tails changed, attentions packed).  Numerics: (a) rounds like the bf16 module op by op,
    with its own dropout stream (counter-based, replayed in the backward instead of a stored mask); (b) is the same arithmetic as three GEMMs; the
    attention core is a flash-style online softmax in fp32 with bf16 probabilities, like the library kernel it replaces, and its own dropout stream."""
    box = _SeedBox()
    n_tail = n_qkv = 0
    salt = 0
    for name, m in plm.named_modules():
        cls = type(m).__name__
        if tails and cls.endswith(("SelfOutput", "Output")) and isinstance(getattr(m, "dense", None), torch.nn.Linear) \
                and isinstance(getattr(m, "LayerNorm", None), torch.nn.LayerNorm) and isinstance(getattr(m, "dropout", None), torch.nn.Dropout) \
                and not hasattr(m, "_fmmt_stock_forward"):
            m._fmmt_stock_forward = m.forward
            m._fmmt_seed = box
            salt += 1
            m._fmmt_salt = salt << 40                       # 2^40 elements per call site
            m.forward = types.MethodType(_sublayer_tail_forward, m)
            n_tail += 1
        elif qkv and cls.endswith("SelfAttention") and all(hasattr(m, a) for a in ("query", "key", "value")) and not hasattr(m, "_fmmt_stock_forward"):
            q, k, v = m.query, m.key, m.value
            if not all(isinstance(l, torch.nn.Linear) for l in (q, k, v)) or getattr(m, "is_cross_attention", False):
                continue
            if not (q.weight.shape == k.weight.shape == v.weight.shape and q.bias is not None and k.bias is not None and v.bias is not None
                    and q.weight.dtype == k.weight.dtype == v.weight.dtype):
                continue
            with torch.no_grad():
                w = torch.cat([q.weight.data, k.weight.data, v.weight.data], dim=0).contiguous()
                b = torch.cat([q.bias.data, k.bias.data, v.bias.data], dim=0).contiguous()
                E = q.weight.shape[0]
                for i, lin in enumerate((q, k, v)):
                    lin.weight.data = w[i * E:(i + 1) * E]
                    lin.bias.data = b[i * E:(i + 1) * E]
            m._fmmt_qkv = (w, b)
            m._fmmt_stock_forward = m.forward
            m._fmmt_seed = box
            m._fmmt_attn_word = box.words                   # its own word of the step's seed draw
            box.words += 1
            m.forward = types.MethodType(_packed_qkv_forward, m)
            n_qkv += 1
    # (c) the feed-forward half as one node: only where its output module took the fused tail (its seed box and salt are the node's)
    n_ffn = 0
    for name, m in plm.named_modules():
        inter, out = getattr(m, "intermediate", None), getattr(m, "output", None)
        if (tails and inter is not None and out is not None and hasattr(m, "feed_forward_chunk") and hasattr(out, "_fmmt_salt") and not hasattr(m, "_fmmt_stock_ffn")
                and isinstance(getattr(inter, "dense", None), torch.nn.Linear) and _is_exact_gelu(getattr(inter, "intermediate_act_fn", None))
                and type(inter).forward.__code__.co_names[:2] == ("dense", "intermediate_act_fn")):
            m._fmmt_stock_ffn = m.feed_forward_chunk
            m.feed_forward_chunk = types.MethodType(_fused_ffn_chunk, m)
            n_ffn += 1
    plm._fmmt_fused_ffn = n_ffn
    # (d) the embedding tables' weight gradients (the stock forward; the tables stay nn.Embedding modules)
    for name, m in plm.named_modules():
        if tails and type(m) is torch.nn.Embedding and not hasattr(m, "_fmmt_stock_forward"):
            m._fmmt_stock_forward = m.forward
            m.forward = types.MethodType(_fused_embedding_forward, m)
    if n_tail or n_qkv:
        plm.register_forward_pre_hook(lambda mod, args, kwargs=None: box.draw(next(mod.parameters()).device) if mod.training else None)
    return n_tail, n_qkv


class MasterWeights:
    """fp32 master copies of a sub-module that runs in a low-precision parameter dtype (the text encoder in bf16).

    Under autocast an fp32 text encoder re-casts every weight to bf16 in every step and casts every weight gradient
    back (about 1050 small launches and 2.5 GB of traffic per step for RoBERTa-large: measured 650 fp32->bf16 and 403
    bf16->fp32 copy kernels per step), and its LayerNorms run with fp32 I/O.  With bf16 parameters the module needs no
    autocast; the optimizer steps fp32 masters (`self.masters`, nn.Parameters that are views of one flat buffer) and
    `sync_low()` rounds them back into the module with one multi-tensor copy.  The masters start as exact copies of the
    fp32 parameters, so the trajectory is that of mixed-precision training with fp32 master weights."""

    def __init__(self, module: torch.nn.Module, dtype=torch.bfloat16):
        low = [p for p in module.parameters() if p.requires_grad]
        if any(p.dtype != torch.float32 for p in low):
            raise TypeError("MasterWeights: the module must still be in fp32 when the masters are taken")
        offs, n = [], 0
        for p in low:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.empty(n, dtype=torch.float32, device=low[0].device)
        self.masters = []
        for p, o in zip(low, offs):
            v = flat[o:o + p.numel()].view_as(p)
            v.copy_(p.detach())
            self.masters.append(torch.nn.Parameter(v))
        # what has no master -- frozen parameters (the unused pooler) and floating-point buffers -- keeps its fp32 values here: the
        # reference leaves them untouched, a checkpoint must not hold their bf16 roundings
        trainable = set(map(id, low))
        self.frozen_fp32 = {k: t.detach().clone() for k, t in list(module.named_parameters()) + list(module.named_buffers())
                            if id(t) not in trainable and t.is_floating_point() and t.dtype == torch.float32}
        module.to(dtype)                                    # in place: the Parameter objects survive, their data becomes bf16
        self.colsum_layers = use_colsum_bias_gradients(module)
        self.fused_layers = fuse_text_encoder(module, tails=PLM_FUSE_TAILS, qkv=PLM_FUSE_QKV)
        self.low = low
        self.flat = flat
        self.module = module

    def pairs(self):
        return list(zip(self.low, self.masters))

    @torch.no_grad()
    def sync_low(self):
        torch._foreach_copy_(self.low, self.masters)

    def state_dict_fp32(self):
        """the module's state_dict with the master values in place of the rounded ones (what a checkpoint should hold)"""
        sd = {k: v.detach().clone() for k, v in self.module.state_dict().items()}
        by_id = {id(p): m for p, m in zip(self.low, self.masters)}
        for k, p in self.module.named_parameters():
            if id(p) in by_id:
                sd[k] = by_id[id(p)].detach().clone()
        for k, t in self.frozen_fp32.items():
            if k in sd:
                sd[k] = t.clone()
        return sd


def _hand_over_gradients(pairs, flat_view_of, accumulate: bool):
    """After a backward: move the fresh gradients (p.grad of the model parameters) into the static flat fp32 buffers the
    optimizer and the all-reduce work on, with a handful of multi-tensor launches instead of one AccumulateGrad add per
    parameter (measured: 768 elementwise adds per step).  `accumulate=False` overwrites (one micro-step per update: no
    zeroing needed), True adds (gradient accumulation).  Leaves p.grad = None on the model parameters."""
    same_src, same_dst, cast_src, cast_dst = [], [], [], []
    for p, master in pairs:
        g = p.grad
        if g is None:
            continue
        p.grad = None
        dst = flat_view_of[master]
        if g.dtype == dst.dtype:
            same_src.append(g)
            same_dst.append(dst)
        else:
            cast_src.append(g)
            cast_dst.append(dst)
    with torch.no_grad():
        if accumulate:
            if same_src:
                torch._foreach_add_(same_dst, same_src)
            if cast_src:                                     # bf16 gradients into fp32 buffers: multi-tensor cast, then add
                tmp = [torch.empty_like(d) for d in cast_dst]
                torch._foreach_copy_(tmp, cast_src)
                torch._foreach_add_(cast_dst, tmp)
        else:
            if same_src:
                torch._foreach_copy_(same_dst, same_src)
            if cast_src:
                torch._foreach_copy_(cast_dst, cast_src)


def _restore(snap):
    """copy the snapshot back, touching only what changed: an untouched buffer keeps its version counter, so host-side
    caches keyed on it (SwinTransformerBlock._mask_is_standard) stay valid and nothing synchronises inside the capture"""
    with torch.no_grad():
        for t, v in snap:
            if not torch.equal(t, v):
                t.copy_(v)


def step_parameters(mm, masters=None):
    """the parameters the target step's optimizer updates: the multimodal model's, with the text encoder's replaced by
    their fp32 masters when it runs in bf16 (MasterWeights)"""
    if masters is None:
        return [p for p in mm.parameters() if p.requires_grad]
    master_of = {id(l): m for l, m in masters.pairs()}
    return [master_of.get(id(p), p) for p in mm.parameters() if p.requires_grad]


def _reset_optimizer_state(opt):
    """zero every tensor of the optimizer state in place (moments, step counters): undoes the warm-up steps that precede
    a graph capture without re-allocating the state the captured graph will address"""
    for st in opt.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    for g in opt.param_groups:                              # HFAdamW keeps its (device) step counter in the group, not in opt.state
        if torch.is_tensor(g.get("step")):
            g["step"].zero_()


class HFAdamW(torch.optim.Optimizer):
    """`transformers.AdamW` restated (the optimizer class the reference constructs, train.py:307,333; transformers 4.24
    optimization.py, not vendored under the reference tree): defaults betas (0.9, 0.999), **eps 1e-6, weight_decay 0.0**,
    correct_bias True; per step
        m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps);  p += p * (-lr wd)
    i.e. eps is added BEFORE the bias correction and the decoupled decay is applied AFTER the update -- both differ from
    torch.optim.AdamW (whose defaults are also eps 1e-8, weight_decay 0.01).  `lr` may be a 0-dim device tensor and the step
    counter is a device tensor, so the eager step is capture-safe; the graphed steps hand the same hyper-parameters to
    fmmt_adamw_batch(hf_semantics=1) through FusedClipAdamW.  One parameter group."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for g in self.param_groups:
            ps = [p for p in g["params"] if p.grad is not None]
            if not ps:
                continue
            grads = [p.grad for p in ps]
            for p in ps:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            # the group's step counter is a device scalar kept IN the param group (capture-safe; transformers.AdamW keeps an int per
            # parameter: `hf_state_dict()` below writes that layout for exchange with the reference class)
            if not torch.is_tensor(g.get("step")):
                g["step"] = torch.zeros((), dtype=torch.float32, device=ps[0].device)
            t = g["step"]
            t.add_(1.0)
            b1, b2 = g["betas"]
            m = [self.state[p]["exp_avg"] for p in ps]
            v = [self.state[p]["exp_avg_sq"] for p in ps]
            torch._foreach_mul_(m, b1)
            torch._foreach_add_(m, grads, alpha=1.0 - b1)
            torch._foreach_mul_(v, b2)
            torch._foreach_addcmul_(v, grads, grads, value=1.0 - b2)
            den = torch._foreach_sqrt(v)
            torch._foreach_add_(den, g["eps"])
            lr = g["lr"] if torch.is_tensor(g["lr"]) else torch.tensor(float(g["lr"]), dtype=torch.float32, device=ps[0].device)
            step_size = lr * torch.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t) if g["correct_bias"] else lr
            upd = torch._foreach_div(m, den)
            torch._foreach_mul_(upd, step_size)
            torch._foreach_sub_(ps, upd)
            if g["weight_decay"] > 0.0:
                dec = torch._foreach_mul(ps, lr * (-g["weight_decay"]))
                torch._foreach_add_(ps, dec)
        return loss

    def load_hf_state_dict(self, sd):
        """load a `transformers.AdamW` state dict (the reference's checkpoints, train.py:316-331: an int `step` per parameter, none in the
        groups): the per-parameter counters must agree within a group and become the group's device counter, so that the next step()
        continues the bias correction at t + 1 instead of restarting it with non-zero moments"""
        sd = {"state": {k: dict(v) for k, v in sd["state"].items()}, "param_groups": [dict(g) for g in sd["param_groups"]]}
        steps = []
        for g in sd["param_groups"]:
            ns = {int(sd["state"][i].pop("step")) for i in g["params"] if i in sd["state"] and "step" in sd["state"][i]}
            if len(ns) > 1:
                raise ValueError(f"per-parameter step counters of one group disagree: {sorted(ns)}")
            steps.append(ns.pop() if ns else None)
        # Optimizer.load_state_dict REPLACES each group with the checkpoint's values: the reference stores `lr` as a Python float, and the
        # graphed steps (FusedClipAdamW, a captured graph) address the group's 0-dim DEVICE lr tensor the scheduler writes into -- keep that
        # tensor object and fill it with the saved value; keys the checkpoint does not carry (`step`, anything added here later) stay
        live = [dict(g) for g in self.param_groups]
        self.load_state_dict(sd)
        for g, old, n in zip(self.param_groups, live, steps):
            if torch.is_tensor(old.get("lr")):
                saved = g["lr"]
                g["lr"] = old["lr"]
                with torch.no_grad():
                    g["lr"].fill_(float(saved))
            for k, v in old.items():
                if k not in g:
                    g[k] = v
            if n is not None:
                if torch.is_tensor(old.get("step")):
                    g["step"] = old["step"]
                    with torch.no_grad():
                        g["step"].fill_(float(n))
                else:
                    g["step"] = torch.full((), float(n), dtype=torch.float32, device=g["params"][0].device)

    def hf_state_dict(self):
        """`state_dict()` in transformers.AdamW's layout: an int `step` per parameter beside exp_avg / exp_avg_sq, none in the groups"""
        sd = self.state_dict()
        for g in sd["param_groups"]:
            t = g.pop("step", None)
            n = int(t.item()) if torch.is_tensor(t) else int(t or 0)
            for i in g["params"]:
                if i in sd["state"]:
                    sd["state"][i] = dict(sd["state"][i], step=n)
        return sd


class FusedHandOver:
    """_hand_over_gradients + the clip norm as ONE pass (fmmt_grad_handover): every fresh gradient is written (accumulation: added) into its
    slot of the flat fp32 buffers and the L2 norm of what the slots then hold lands in `norm_out` -- instead of a multi-tensor copy and a
    multi-tensor norm over the same 435 M values (23 + 30 launches, 1.3 ms of kernel time in the step's serial tail).  Returns False --
    nothing touched -- when a parameter has no gradient (its slot would keep a stale value the norm must still see) or a gradient is not a
    contiguous fp32 / bf16 tensor: the caller then takes the two-pass path.  The tables are built from the gradients' addresses on every call
    (warm-up passes, then once under capture: a captured step replays the copy of the one pinned table made then)."""

    def __init__(self, n_records: int):
        # pinned host tables, allocated HERE (a host allocation is illegal while a stream captures): two, used alternately -- an eager
        # (warm-up) call waits for the stream before it rewrites one, the call under capture leaves its table untouched for the replays
        self.hosts = [torch.empty(32 * max(1, n_records), dtype=torch.uint8).pin_memory() for _ in range(2)]
        # a call UNDER CAPTURE gets a table of its own that no later call rewrites (the captured copy node re-reads it on every replay: an
        # eager call after the capture -- a re-capture, a debugging pass -- must not land in it); four captures per object, then it raises
        self.capture_hosts = [torch.empty(32 * max(1, n_records), dtype=torch.uint8).pin_memory() for _ in range(4)]
        self.captures = 0
        self.calls = 0
        self.keep = []                                       # (device table, partial sums): alive as long as the graphs

    @torch.no_grad()
    def __call__(self, pairs, flat_view_of, accumulate: bool, norm_out) -> bool:
        import numpy as np
        from . import _lib
        grads = []
        for p, master in pairs:
            g = p.grad
            if g is None or not g.is_cuda or not g.is_contiguous() or g.dtype not in (torch.float32, torch.bfloat16) or g.numel() != flat_view_of[master].numel():
                return False
            grads.append(g)
        if not grads:
            return False
        dev = grads[0].device
        arr = np.zeros(len(grads), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("n", "<i8"), ("bb", "<i4"), ("flags", "<i4")]))
        blocks = 0
        for i, (g, (p, master)) in enumerate(zip(grads, pairs)):
            arr[i] = (g.data_ptr(), flat_view_of[master].data_ptr(), g.numel(), blocks, (1 if g.dtype == torch.bfloat16 else 0) | (2 if accumulate else 0))
            blocks += (g.numel() + 4095) // 4096
        raw = torch.from_numpy(arr.view(np.uint8))
        if raw.numel() > self.hosts[0].numel():
            return False
        if torch.cuda.is_current_stream_capturing():
            if self.captures >= len(self.capture_hosts):
                raise RuntimeError("FusedHandOver: more captures than pinned capture tables")
            host = self.capture_hosts[self.captures][:raw.numel()]
            self.captures += 1
        else:
            torch.cuda.current_stream(dev).synchronize()     # an earlier eager call's copy has read its table
            host = self.hosts[self.calls % 2][:raw.numel()]
            self.calls += 1
        host.copy_(raw)
        table = torch.empty(raw.numel(), dtype=torch.uint8, device=dev)
        table.copy_(host, non_blocking=True)
        partial = torch.empty(blocks, dtype=torch.float32, device=dev)
        rc = _lib.load().fmmt_grad_handover(len(grads), blocks, table.data_ptr(), partial.data_ptr(), norm_out.data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "fmmt_grad_handover")
        self.keep = self.keep[-3:] + [(table, partial)]
        for p, _ in pairs:
            p.grad = None
        return True


class FusedClipAdamW:
    """clip_grad_norm_ + AdamW.step() (torch.optim.AdamW, or HFAdamW = the reference's transformers.AdamW) (+ the bf16 re-rounding of parameters stepped through fp32 masters) as one
    multi-tensor norm and ONE kernel launch (fmmt_adamw_batch): the gradients are read once and not scaled in place, the fp32
    parameters are not read a second time for their bf16 twins.  Takes its hyper-parameters from an existing torch AdamW
    (one parameter group, tensor learning rate on the device as `capturable=True` keeps it, so a LambdaLR scheduler keeps
    working); keeps its own moments and step counter -- `optimizer.state` stays empty while this is in use."""

    @staticmethod
    def eligible(opt, params):
        if not FUSED_ADAMW or type(opt) not in (torch.optim.AdamW, HFAdamW) or len(opt.param_groups) != 1:
            return False
        g = opt.param_groups[0]
        return (torch.is_tensor(g["lr"]) and g["lr"].is_cuda and not g.get("amsgrad", False) and not g.get("maximize", False)
                and g.get("correct_bias", True) and all(p.dtype == torch.float32 and p.is_cuda for p in params))

    def __init__(self, opt, params, grad_of, low_of, max_norm):
        import numpy as np
        g = opt.param_groups[0]
        self.lr, (self.b1, self.b2), self.eps, self.wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        self.hf = isinstance(opt, HFAdamW)                    # transformers.AdamW's update (the reference's class) vs torch.optim.AdamW's
        self.max_norm = float(max_norm)
        dev = params[0].device
        self.step = torch.zeros((), dtype=torch.float32, device=dev)
        self.grads = [grad_of[p] for p in params]
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.norm = torch.zeros((), dtype=torch.float32, device=dev)
        self.norm_ready = False                              # set by the caller when FusedHandOver already left the norm in self.norm
        # The norm alone as ONE launch: fmmt_grad_handover with every flat gradient buffer as its own source and destination (a copy onto itself
        # plus the partial sums of squares).  N > 1: the buffers change between the hand-over and the update (the exchange averages them), so the
        # hand-over's own norm is of the wrong values -- this second pass over the REDUCED buffers replaces the multi-tensor norm (30 launches).
        self.norm_table = self.norm_partial = None
        if FUSED_HANDOVER:
            tb = np.zeros(len(params), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("n", "<i8"), ("bb", "<i4"), ("flags", "<i4")]))
            nb = 0
            for i, g in enumerate(self.grads):
                tb[i] = (g.data_ptr(), g.data_ptr(), g.numel(), nb, 0)
                nb += (g.numel() + 4095) // 4096
            self.norm_table = torch.from_numpy(tb.view(np.uint8).copy()).to(dev)
            self.norm_partial = torch.empty(nb, dtype=torch.float32, device=dev)
            self.norm_blocks = nb
        recs, blocks = [], 0
        self.keep = (params, [low_of.get(id(p)) for p in params])
        for p, gr, m, v, low in zip(params, self.grads, self.m, self.v, self.keep[1]):
            assert gr.dtype == torch.float32 and gr.is_contiguous() and p.is_contiguous() and (low is None or (low.dtype == torch.bfloat16 and low.is_contiguous()))
            recs.append((p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), low.data_ptr() if low is not None else 0, p.numel(), blocks, 0))
            blocks += (p.numel() + 4095) // 4096
        arr = np.zeros(len(recs), dtype=np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("low", "<u8"), ("n", "<i8"), ("bb", "<i4"), ("pad", "<i4")]))
        for i, r in enumerate(recs):
            arr[i] = r
        self.desc = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
        self.n, self.blocks = len(recs), blocks

    def reset(self):
        self.step.zero_()
        torch._foreach_zero_(self.m)
        torch._foreach_zero_(self.v)

    @torch.no_grad()
    def load_from(self, opt):
        """take moments and step counter from `opt` (after HFAdamW.load_hf_state_dict / Optimizer.load_state_dict of a checkpoint): this object
        keeps its own m, v and step -- the ones the captured update addresses -- so a loaded optimizer state must be copied IN, in place.
        Parameters the optimizer holds no state for keep zero moments."""
        params = self.keep[0]
        g = opt.param_groups[0]
        for p, m, v in zip(params, self.m, self.v):
            st = opt.state.get(p, {})
            if "exp_avg" in st:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
            else:
                m.zero_()
                v.zero_()
        t = g.get("step")
        if t is None:                                         # torch.optim.AdamW: a per-parameter counter
            ts = {float(opt.state[p]["step"]) for p in params if p in opt.state and "step" in opt.state[p]}
            if len(ts) > 1:
                raise ValueError(f"per-parameter step counters disagree: {sorted(ts)}")
            t = ts.pop() if ts else 0.0
        self.step.fill_(float(t))

    @torch.no_grad()
    def update(self):
        from . import ops
        self.step.add_(1.0)
        if not self.norm_ready:
            if self.norm_table is not None:
                from . import _lib
                _lib.check(_lib.load().fmmt_grad_handover(len(self.grads), self.norm_blocks, self.norm_table.data_ptr(), self.norm_partial.data_ptr(),
                                                          self.norm.data_ptr(), torch.cuda.current_stream(self.norm.device).cuda_stream), "fmmt_grad_handover(norm)")
            else:
                self.norm.copy_(torch.nn.utils.get_total_norm(self.grads, 2.0, foreach=True))
        self.norm_ready = False
        ops.adamw_batch(self.n, self.blocks, self.desc, self.lr, self.step, self.norm, self.b1, self.b2, self.eps, self.wd, self.max_norm, self.hf)


def _ops_pinned_scope(shadows):
    from . import ops
    return ops.pinned_scope(shadows)


def _pin_shadows(modules):
    """ops.PinnedShadows over the parameters of `modules` (PIN_SHADOWS = False: per-weight casts inside the graph, as before)"""
    if not PIN_SHADOWS:
        return None
    from . import ops
    seen, params = set(), []
    for m in modules:
        for p in m.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
    return ops.PinnedShadows(params)


def _bump_versions(params):
    """A graph replay updates parameters in place without autograd noticing: bump their version counters, so that eager code
    that caches by version (ops._lp: the bf16 weight shadows of an eval() pass after training) rebuilds what it cached."""
    inc = getattr(torch._C, "_increment_version", None)
    if inc is None:
        return
    try:
        inc(params)
    except TypeError:
        for p in params:
            inc(p)


class GraphedTargetStep:
    """The whole target-task step as HIP graphs, replayed per step with one host call each -- two graphs (A = A1 + A2 as one, then B) on
    a single GPU, three when a gradient exchange has to be hidden (N > 1):

      A1 text encoder (forked onto a second stream) || Swin forward -> frame filter -> fusion stack -> cross-entropy ->
         backward through the fusion stack, the text encoder and Swin's head / stages 3-1, down to the gradient of Swin's stage-0
         output (SWIN_CUT); the multimodal gradients land in static flat fp32 buffers (parallel.GradientAverager's buckets, hooks off);
      -- N > 1: one asynchronous all-reduce per flat bucket is ISSUED here (the only collective of the step): the
         collectives run on the backend's stream behind A1 and BESIDE A2 --
      A2 the rest of Swin's backward from that gradient (stage 0: ~10 ms); Swin's parameter gradients land in buffers nobody reads
         (Swin's target-step gradients are discarded, train.py:20,141, and therefore never exchanged);
      -- N > 1: the stream waits for the collectives --
      B  clip_grad_norm_ + optimizer.step() (capturable) + zeroing of the flat buffers.

    Why: issued launch by launch the step costs ~75 ms of host time (Swin's ~600 launches through Python autograd and
    ctypes, ~800 AccumulateGrad nodes, clip_grad_norm_ over ~800 tensors, the optimizer) against ~80 ms of GPU time, so
    any kernel speed-up would disappear behind the host; two graph launches cost a few milliseconds.  Everything inside is
    capture-safe: no host synchronisation, DropPath / Gumbel / dropout noise from device-side generators, bf16 weight
    shadows re-cast inside the graph.  Shapes are static: __call__ copies the batch into the captured input buffers.
    `accumulation_steps` > 1 replays A that many times per B (gradients accumulate in the flat buffers).
    Parity: tests/test_gpu_train_step.py walks this against the eager TargetStep (same losses, same parameters)."""

    # where the text branch's launches enter the capture order: "start" (in front of Swin's forward), "pe" (behind PatchEmbed), "s<i>"
    # (behind Swin stage i).  A class constant the probes patch (PATCH="train_step.GraphedTargetStep.TEXT_FORK_AT='s0'" tools/probes/bench_patch.py),
    # not an environment switch; measured in NOTES.md R5.3.
    TEXT_FORK_AT = "start"

    def __init__(self, swin_model, multimodal_model, optimizer, scheduler, args, batch, autocast_dtype=None,
                 overlap_text=True, parallel_fusion=False, averager=None, warmup_iters=2, masters=None, discarded_swin_gradients="compute",
                 swin_cut: int = 0, pipeline_swin: bool = False, branch_graphs: bool = False, fork_streams: bool = False):
        """`averager`: GradientAverager(hooks=False) over the parameters the optimizer steps (default: the multimodal
        model's); `swin_cut`: with an exchange to hide (N > 1), the Swin stage behind which the backward graph is cut (0: the second
        piece is stage 0's backward, ~10 ms; 1: stages 1 + 0, ~17 ms) -- the caller picks it from a MEASURED exchange time
        (`pick_swin_cut`, bench.py), never from a nominal link rate; `masters`: optional MasterWeights of the text encoder -- then `averager` and the optimizer must have been
        built over `step_parameters(multimodal_model, masters)`; `discarded_swin_gradients`: "compute" (default) / "skip", SKIP_NOTE;
        `pipeline_swin` (one rank, no exchange): Swin's FORWARD becomes a graph of its own (S) over two alternating sets of saved activations, and
        `step(batch, next_batch=...)` replays S for the NEXT batch on a second stream beside this batch's graph A' (frame filter, text encoder,
        fusion stack, loss, every backward).  Valid because nothing steps Swin in a target step (train.py:20,141: its target-step gradients are
        discarded) -- the prefetched forward sees the weights the in-order forward would; a Swin parameter whose version moved since the prefetch
        (an auxiliary step ran in between) makes the step redo the forward in order.  Why: per step ~22 ms of the GPU's time are streams of small
        launches at low occupancy (text encoder forward and the tail of its backward, the fusion stack, the update) that a 14 ms block of
        chip-filling Swin kernels can run beside; every step still executes exactly one Swin forward."""
        import os
        from .parallel import GradientAverager
        if discarded_swin_gradients not in ("compute", "skip"):
            raise ValueError("discarded_swin_gradients: 'compute' or 'skip'")
        self.skip_swin_bwd = discarded_swin_gradients == "skip"
        self.pipeline = bool(pipeline_swin)
        self.branches = bool(branch_graphs)                 # BRANCH_NOTE below
        self.forked = bool(fork_streams)                    # FORK_NOTE at _fwd_bwd_forked
        if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "") != "0":
            raise RuntimeError("GraphedTargetStep: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in the environment before the HIP "
                               "runtime initialises (see facialmmt_amd/__init__.py)")
        self.swin, self.mm, self.opt, self.sched, self.args = swin_model, multimodal_model, optimizer, scheduler, args
        self.autocast_dtype = autocast_dtype
        self.i_batch = 0
        self.timing = None
        dev = batch[0].device
        # every batch entry becomes a static device tensor: a python list / int baked into the capture would silently be reused
        # by every replay (the reference's collate hands num_imgs and the utterance index over as lists)
        self.static = [t.clone() if torch.is_tensor(t) else torch.as_tensor(t, device=dev) for t in batch]
        # static flat gradient buffers for the multimodal parameters (hooks off: inside a replay no autograd hook fires)
        self.masters = masters
        self.flat = averager if averager is not None else GradientAverager(step_parameters(self.mm, masters), hooks=False)
        if self.flat._handles:
            raise ValueError("GraphedTargetStep needs GradientAverager(..., hooks=False): the exchange runs between the graphs")
        self.flat_view_of = {p: p.grad for p in self.flat.params}          # the averager made every .grad a view of its bucket
        low_of = {id(m): l for l, m in (masters.pairs() if masters is not None else [])}
        self.pairs = [(low_of.get(id(p), p), p) for p in self.flat.params]     # (parameter the model differentiates, parameter the optimizer steps)
        for l, m in self.pairs:
            l.grad = None
        self.fused = None
        if FusedClipAdamW.eligible(optimizer, self.flat.params):
            self.fused = FusedClipAdamW(optimizer, self.flat.params, self.flat_view_of, low_of, args.clip)
        # hand-over and clip norm in one pass; with an exchange between the hand-over and the update (N > 1) the hand-over still runs as one pass, its norm is
        # discarded and FusedClipAdamW.update takes the norm of the REDUCED buffers in one more launch
        self.exchanging = bool(getattr(self.flat, "active", False))
        self.handover = FusedHandOver(len(self.pairs)) if (self.fused is not None and FUSED_HANDOVER) else None
        self.accumulate = args.trg_accumulation_steps > 1
        self.mm.text_stream = None
        # inside ONE graph the fork / join below become parallel branches; which hardware queue the branches replay on is the
        # runtime's choice at replay time, not a property of the stream object used during capture
        # warm-up / capture stream and the two branch streams: three different HIP streams (distinct_stream)
        cap = distinct_stream(dev)
        self.shadows = None                                 # set after the warm-up passes, before capture
        self.text_stream = distinct_stream(dev, (cap,)) if overlap_text else None
        self.mm.pair_stream = distinct_stream(dev, (cap, self.text_stream)) if parallel_fusion else None
        if self.forked:
            if self.text_stream is None:
                self.text_stream = distinct_stream(dev, (cap,))
            self.swin_stream = distinct_stream(dev, tuple(x for x in (cap, self.text_stream, self.mm.pair_stream) if x is not None))
            self._tick = torch.zeros(1, device=dev)
        # -- warm-up on a side stream (lazy initialisations: kernel attributes, shadow caches, optimizer state), undone below
        snap = [(t, t.detach().clone()) for m in (self.swin, self.mm) for t in list(m.parameters()) + list(m.buffers())]
        if masters is not None:
            snap += [(t, t.detach().clone()) for t in masters.masters]
        rng = torch.cuda.get_rng_state(dev)
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(warmup_iters):
                self._fwd_bwd()
                self._update()
            self.swin.zero_grad(set_to_none=True)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize(dev)
        _restore(snap)
        del snap
        _reset_optimizer_state(self.opt)
        if self.fused is not None:
            self.fused.reset()
        self.flat.zero_grad()
        torch.cuda.set_rng_state(rng, dev)
        # -- capture
        self.shadows = _pin_shadows([self.swin, self.mm])
        self.swin_shadows = self.mm_shadows = None
        self._swin_params = [q for q in self.swin.parameters()]
        # With an exchange to hide (N > 1) the forward/backward is TWO graphs, cut where the multimodal gradients are complete; alone on
        # the GPU it stays ONE graph: inside it the text encoder's backward runs as a branch beside Swin's backward, which a cut in
        # front of Swin's backward would serialise (measured at N = 1: 72.1 ms per step cut, 69 ms uncut).
        self.split = bool(getattr(self.flat, "active", False))
        if swin_cut not in (0, 1, 2):
            raise ValueError("swin_cut: 0, 1 or 2")
        self.SWIN_CUT = int(swin_cut)                      # the window behind the cut has to hold the exchange: see pick_swin_cut
        if self.pipeline and (self.split or self.skip_swin_bwd):
            raise ValueError("pipeline_swin: one rank without a gradient exchange, Swin's backward computed")
        if self.branches and (self.split or self.skip_swin_bwd or self.pipeline):
            raise ValueError("branch_graphs: one rank without a gradient exchange, Swin's backward computed, not together with pipeline_swin")
        if self.forked and (self.split or self.skip_swin_bwd or self.pipeline or self.branches):
            raise ValueError("fork_streams: one rank without a gradient exchange, Swin's backward computed, not together with pipeline_swin / branch_graphs")
        self.graph_a, self.graph_a2, self.graph_b = torch.cuda.CUDAGraph(), (torch.cuda.CUDAGraph() if self.split else None), torch.cuda.CUDAGraph()
        self.sets, self.side_stream, self.cur, self.prefetched = [], None, 0, None
        if self.pipeline:
            # two sets (graph S: Swin forward; graph A': everything else), each in a memory pool of ITS OWN: S of one set replays beside A' of the
            # other, and inside a shared pool the capture of the second set would reuse blocks the first set's backward freed
            self.side_stream = distinct_stream(dev, tuple(x for x in (cap, self.text_stream, self.mm.pair_stream) if x is not None))
            self.swin_shadows, self.mm_shadows = _pin_shadows([self.swin]), _pin_shadows([self.mm])
            with capture_window(), _ops_pinned_scope(self.shadows):
                for k in range(2):
                    frames_k = self.static[8] if k == 0 else self.static[8].clone()
                    g_s, g_a = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_s, stream=cap):
                        preds = self._swin_forward(frames_k)
                    with torch.cuda.graph(g_a, pool=g_s.pool(), stream=cap):
                        loss, new_mask, _ = self._fwd_bwd_multimodal(whole=True, preds=preds)
                    del preds
                    self.sets.append(types.SimpleNamespace(S=g_s, A=g_a, frames=frames_k, loss=loss, mask=new_mask,
                                                           ev_s=torch.cuda.Event(), ev_a=torch.cuda.Event()))
                    _KEEP_GRAPHS.append((g_s, g_a))
                with torch.cuda.graph(self.graph_b, stream=cap):
                    self._update()
            self.loss, self.new_mask = self.sets[0].loss, self.sets[0].mask
        elif self.branches:
            # BRANCH_NOTE.  Inside ONE graph the text encoder and Swin are two branches, but the replay does not run them side by side: the
            # runtime enqueues a graph branch by branch, and a branch that continues on another queue waits for EVERYTHING the first branch
            # has enqueued so far, not for the node it depends on (profiles/r05_timeline.txt: the text encoder's forward runs alone for
            # ~4 ms in front of Swin's forward -- behind it when the capture order is swapped, TEXT_FORK_AT --, its backward starts ~12 ms
            # into Swin's backward and ends ~5 ms after it).  Here every branch is a graph of its own, launched on one of two streams with
            # events between them -- dependencies exactly where the data flow has them:
            #     side:  T  text encoder forward ............................ TB text encoder backward, hand-over | B update
            #     main:  S  Swin forward -> [T] F  frame filter, fusion stack, loss, its backward -> SB Swin backward
            # Autograd crosses the graph boundaries on detached copies of the two branch outputs (F differentiates down to them, TB / SB
            # continue from their gradients).  Pools: graphs that may run at the same time never share one (T, TB, B | S, SB | F).
            self.side_stream = distinct_stream(dev, (cap,))
            self.swin_shadows, self.mm_shadows = _pin_shadows([self.swin]), _pin_shadows([self.mm])
            g = types.SimpleNamespace(T=torch.cuda.CUDAGraph(), S=torch.cuda.CUDAGraph(), F=torch.cuda.CUDAGraph(), TB=torch.cuda.CUDAGraph(),
                                      SB=torch.cuda.CUDAGraph(), ev_t=torch.cuda.Event(), ev_f=torch.cuda.Event())
            with capture_window(), _ops_pinned_scope(self.shadows):
                with torch.cuda.graph(g.T, stream=cap):
                    feat, tmask = self._text_forward()
                with torch.cuda.graph(g.S, stream=cap):
                    preds = self._swin_forward(self.static[8])
                with torch.cuda.graph(g.F, stream=cap):
                    self.loss, self.new_mask, dfeat, dpreds = self._fusion_fwd_bwd(feat, tmask, preds)
                with torch.cuda.graph(g.TB, pool=g.T.pool(), stream=cap):
                    self._text_backward(feat, dfeat)
                with torch.cuda.graph(g.SB, pool=g.S.pool(), stream=cap):
                    self._bwd_swin((preds, dpreds))
                del feat, tmask, preds, dfeat, dpreds
                with torch.cuda.graph(self.graph_b, pool=g.T.pool(), stream=cap):
                    self._update()
            self.bg = g
            _KEEP_GRAPHS.append((g.T, g.S, g.F, g.TB, g.SB))
        else:
          with capture_window(), _ops_pinned_scope(self.shadows):
            if self.split:
                with torch.cuda.graph(self.graph_a, stream=cap):
                    self.loss, self.new_mask, swin_out = self._fwd_bwd_multimodal()
                with torch.cuda.graph(self.graph_a2, pool=self.graph_a.pool(), stream=cap):
                    self._bwd_swin(swin_out)
                del swin_out
            else:
                with torch.cuda.graph(self.graph_a, stream=cap):
                    self.loss, self.new_mask = self._fwd_bwd()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), stream=cap):
                self._update()
        _KEEP_GRAPHS.append((self.graph_a, self.graph_a2, self.graph_b))
        self.swin.zero_grad(set_to_none=True)              # drop the references; the graph's pool keeps the buffers
        self.mm.pair_stream = None
        self.flat.zero_grad()                              # the capture itself executes nothing

    # one micro-step: forward + backward (runs eagerly during warm-up, once more under capture), in the two pieces the graphs hold
    def _fwd_bwd(self):
        if getattr(self, "forked", False):
            return self._fwd_bwd_forked()
        loss, new_mask, _ = self._fwd_bwd_multimodal(whole=True)
        return loss, new_mask

    def _fwd_bwd_forked(self):
        """FORK_NOTE.  ONE graph whose two branches really start together.  The replay enqueues a graph branch by branch; the branch that CONTINUES on
        the launch queue (the fork node's first-created child) goes first and whole, and a branch that moves to another queue waits for the launch
        queue's tail as of then -- so with the text encoder (or Swin) as that first child the other branch starts when it has finished
        (profiles/r05_timeline*.txt).  Here the fork node's first child is a one-element tick on the launch queue that leads straight to the join; the
        text encoder and Swin are its second and third child, each on a stream of its own, and wait for a tail that ends at the tick.  Forward and
        backward are forked the same way (the backward by hand: the fusion stack is differentiated down to detached branch outputs, as branch_graphs)."""
        cap, ts, ss = torch.cuda.current_stream(), self.text_stream, self.swin_stream
        frames = self.static[8]
        if self.shadows is not None:
            self.shadows.refresh()
        ev = torch.cuda.Event()
        ev.record(cap)
        self._tick.add_(1.0)                                # first child of the fork: the launch queue's own continuation
        ts.wait_event(ev)
        ss.wait_event(ev)
        with torch.cuda.stream(ts):
            feat, tmask = self._text_forward(refresh=False)
        with torch.cuda.stream(ss):
            preds = self.swin(frames, is_trg_task=True)
        cap.wait_stream(ts)
        cap.wait_stream(ss)
        for t in (feat, tmask, preds):
            if torch.is_tensor(t):
                t.record_stream(cap)
        loss, new_mask, dfeat, dpreds = self._fusion_fwd_bwd(feat, tmask, preds)
        ev2 = torch.cuda.Event()
        ev2.record(cap)
        self._tick.add_(1.0)
        ts.wait_event(ev2)
        ss.wait_event(ev2)
        if dfeat is not None:
            dfeat.record_stream(ts)
        if dpreds is not None:
            dpreds.record_stream(ss)
        with torch.cuda.stream(ts):
            self._text_backward(feat, dfeat, handover=False)
        with torch.cuda.stream(ss):
            self._bwd_swin((preds, dpreds))
        cap.wait_stream(ts)
        cap.wait_stream(ss)
        self._hand_over()
        return loss, new_mask

    def _text_forward(self, refresh=True):
        """branch_graphs, graph T: the multimodal model's bf16 shadows (text encoder AND fusion stack: one launch), then the text branch"""
        (ids, attn_mask, sep_mask, _a, _am, _v, _vm, _l, _f, _n, utt_idx) = self.static
        if refresh and self.mm_shadows is not None:
            self.mm_shadows.refresh()
        import contextlib
        ac = (lambda: torch.autocast("cuda", dtype=self.autocast_dtype)) if self.autocast_dtype is not None else contextlib.nullcontext
        with ac():
            return self.mm.text_branch(ids, attn_mask, sep_mask, torch.as_tensor(utt_idx, device=ids.device))

    def _fusion_fwd_bwd(self, feat, tmask, preds):
        """branch_graphs, graph F: frame filter, fusion stack, loss and their backward, down to detached copies of the two branch outputs;
        returns (loss, kept-frame mask, d(text features), d(Swin output)).  The fusion stack's parameter gradients are complete here."""
        (_i, _m, _s, audio, audio_mask, vision_inputs, vision_mask, labels, _f, num_imgs, _u) = self.static
        import contextlib
        ac = (lambda: torch.autocast("cuda", dtype=self.autocast_dtype)) if self.autocast_dtype is not None else contextlib.nullcontext
        feat_d = feat.detach().requires_grad_(True)
        preds_d = preds.detach().requires_grad_(True)
        vis_concat, new_mask = select_frames(preds_d.float(), vision_inputs, vision_mask, num_imgs, self.args.FacialEmoImpor_threshold)
        with ac():
            logits = self.mm.fusion_branch(feat_d, tmask, audio, audio_mask, vis_concat, new_mask)
        loss = F.cross_entropy(logits.float(), labels) / self.args.trg_accumulation_steps
        leaves = [l for l, _ in self.pairs if l.requires_grad]
        got = torch.autograd.grad(loss, [feat_d, preds_d] + leaves, allow_unused=True)
        for l, gr in zip(leaves, got[2:]):
            l.grad = gr
        return loss.detach(), new_mask, got[0], got[1]

    def _text_backward(self, feat, dfeat, handover=True):
        """branch_graphs, graph TB: the text branch's backward from the gradient of its output, then the hand-over of EVERY multimodal gradient
        (graph F's included) to the optimizer's flat buffers with the clip norm in the same pass"""
        leaves = [l for l, _ in self.pairs if l.requires_grad]
        if dfeat is not None:
            got = torch.autograd.grad(feat, leaves, dfeat, allow_unused=True)
            for l, gr in zip(leaves, got):
                if gr is not None:
                    l.grad = gr if l.grad is None else l.grad + gr      # (a parameter both pieces use: none in this model)
        if handover:
            self._hand_over()

    def _hand_over(self):
        if self.handover is not None and self.handover(self.pairs, self.flat_view_of, self.accumulate, self.fused.norm):
            self.fused.norm_ready = True
        else:
            _hand_over_gradients(self.pairs, self.flat_view_of, self.accumulate)
            if self.fused is not None:
                self.fused.norm_ready = False

    def _call_branches(self, batch):
        g, main, side = self.bg, torch.cuda.current_stream(), self.side_stream
        main.wait_stream(side)                              # the previous step's side-stream work read the static inputs
        with torch.no_grad():
            for i, (dst, src) in enumerate(zip(self.static, batch)):
                if dst is src:
                    continue
                src = src if torch.is_tensor(src) else torch.as_tensor(src)
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError(f"GraphedTargetStep: batch entry {i} has shape {tuple(src.shape)}, the captured graphs are for {tuple(dst.shape)}")
                dst.copy_(src, non_blocking=True)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            g.T.replay()
            g.ev_t.record(side)
        g.S.replay()
        main.wait_event(g.ev_t)
        g.F.replay()
        g.ev_f.record(main)
        side.wait_event(g.ev_f)
        self.i_batch += 1
        last = self.i_batch % self.args.trg_accumulation_steps == 0
        with torch.cuda.stream(side):
            g.TB.replay()
            if last:
                self.graph_b.replay()
        g.SB.replay()
        main.wait_stream(side)                              # what follows this call on the current stream sees the finished step
        if last:
            _bump_versions(self.flat.params)
            if self.sched is not None:
                self.sched.step()
        return self.loss, self.new_mask

    def _swin_forward(self, frames):
        """pipeline_swin, graph S: Swin's forward alone (its bf16 shadows refreshed at the head: an auxiliary step may have moved the weights)"""
        if self.swin_shadows is not None:
            self.swin_shadows.refresh()
        return self.swin(frames, is_trg_task=True)

    def _fwd_bwd_multimodal(self, whole=False, preds=None):
        """forward of everything + backward of the loss down to (a) the multimodal parameters and (b) Swin's output; returns
        (loss, kept-frame mask, (Swin output, its gradient)) -- the second piece continues from the latter.  `preds` (pipeline_swin): Swin's
        output comes from graph S; this piece starts behind it and its backward runs through S's autograd graph."""
        (ids, attn_mask, sep_mask, audio, audio_mask, vision_inputs, vision_mask, labels, frames, num_imgs, utt_idx) = self.static
        mm, args = self.mm, self.args
        ctx = (lambda: torch.autocast("cuda", dtype=self.autocast_dtype)) if self.autocast_dtype is not None else None
        import contextlib
        ac = ctx if ctx is not None else contextlib.nullcontext
        main = torch.cuda.current_stream()
        pending = None
        if preds is not None:
            if self.mm_shadows is not None:
                self.mm_shadows.refresh()                   # Swin's shadows belong to graph S
        elif self.shadows is not None:
            self.shadows.refresh()                          # every bf16 weight shadow of the step, one launch (ops.PinnedShadows)
        fork_hook = None
        if self.text_stream is not None:
            # fork: the text branch does not depend on the visual path.  WHERE its launches sit in the capture order decides when the
            # replay lets the other branch start (TEXT_FORK_AT below)
            fork_ev = torch.cuda.Event()
            fork_ev.record(main)
            box = {}

            def launch_text(*_):
                self.text_stream.wait_event(fork_ev)
                with torch.cuda.stream(self.text_stream), ac():
                    box["out"] = mm.text_branch(ids, attn_mask, sep_mask, torch.as_tensor(utt_idx, device=ids.device))

            where = self.TEXT_FORK_AT if preds is None and not self.skip_swin_bwd else "start"
            sw = getattr(self.swin, "swin", None)
            mod = None
            if where == "pe":
                mod = getattr(sw, "patch_embed", None)
            elif where.startswith("s") and where[1:].isdigit() and getattr(sw, "layers", None) is not None:
                mod = sw.layers[int(where[1:])]
            if mod is None:
                launch_text()
                pending = box["out"]
            else:
                fork_hook = mod.register_forward_hook(launch_text)
        cut, hook = {}, None
        if preds is not None:
            pass
        elif self.skip_swin_bwd:
            with torch.no_grad():
                preds = self.swin(frames, is_trg_task=True)
        else:
            if not whole:
                # two-piece backward: the cut goes INSIDE Swin, behind stage SWIN_CUT (below): the first piece then holds the text
                # encoder's backward beside Swin's head / stage-3 / -2 / -1 backward -- as the single graph does --, the second piece
                # Swin's stage-0 backward beside the gradient exchange.  (Cut at Swin's output, the text backward ran alone in front
                # of all of Swin's backward: 71.6 against 66.8 ms per step at one rank.)
                layers = getattr(getattr(self.swin, "swin", None), "layers", None)
                if layers is not None and len(layers) > self.SWIN_CUT + 1:
                    hook = layers[self.SWIN_CUT].register_forward_hook(lambda m, i, o: cut.__setitem__("x", o))
            preds = self.swin(frames, is_trg_task=True)
            if hook is not None:
                hook.remove()
            if fork_hook is not None:
                fork_hook.remove()
                if "out" not in box:                        # the hooked module was not called through __call__ on this path
                    launch_text()
                pending = box["out"]
        vis_concat, new_mask = select_frames(preds.float(), vision_inputs, vision_mask, num_imgs, args.FacialEmoImpor_threshold)
        with ac():
            if pending is None:
                pending = mm.text_branch(ids, attn_mask, sep_mask, torch.as_tensor(utt_idx, device=ids.device))
            else:
                main.wait_stream(self.text_stream)         # join
                for t in pending:
                    t.record_stream(main)
            logits = mm.fusion_branch(pending[0], pending[1], audio, audio_mask, vis_concat, new_mask)
        loss = F.cross_entropy(logits.float(), labels) / args.trg_accumulation_steps
        if whole:                                            # one piece: autograd runs the text branch's backward beside Swin's
            loss.backward()
            if self.handover is not None and self.handover(self.pairs, self.flat_view_of, self.accumulate, self.fused.norm):
                self.fused.norm_ready = not self.exchanging     # with an exchange behind it the norm is taken again, of the reduced buffers (FusedClipAdamW.update)
            else:
                _hand_over_gradients(self.pairs, self.flat_view_of, self.accumulate)
            return loss.detach(), new_mask, None
        # backward, first piece: every leaf the optimizer steps plus Swin's output (the autograd graph below `preds` -- Swin -- is
        # left untouched, with its saved activations, for the second piece)
        leaves = [l for l, _ in self.pairs if l.requires_grad]
        x_cut = cut.get("x") if torch.is_tensor(cut.get("x")) and cut["x"].requires_grad else None
        late = []
        if self.skip_swin_bwd:
            got = (None,) + tuple(torch.autograd.grad(loss, leaves, allow_unused=True))
        elif x_cut is not None:
            # Swin's parameters ABOVE the cut get their gradients in this piece (requested explicitly: autograd.grad computes nothing it is
            # not asked for, and the reference computes them)
            early = {id(q) for q in self.swin.swin.patch_embed.parameters()}
            for lyr in list(self.swin.swin.layers)[:self.SWIN_CUT + 1]:
                early |= {id(q) for q in lyr.parameters()}
            if getattr(self.swin.swin, "absolute_pos_embed", None) is not None:
                early.add(id(self.swin.swin.absolute_pos_embed))
            late = [q for q in self.swin.parameters() if q.requires_grad and id(q) not in early]
            got = torch.autograd.grad(loss, [x_cut] + leaves + late, allow_unused=True)
        else:
            x_cut = preds
            got = torch.autograd.grad(loss, [preds] + leaves, allow_unused=True)
        dpreds = got[0]
        for l, g in zip(leaves, got[1:1 + len(leaves)]):
            l.grad = g
        for q, g in zip(late, got[1 + len(leaves):]):        # discarded like all of Swin's target-step gradients (train.py:20,33)
            q.grad = g
        # (Letting the fused update read the model's own .grad tensors instead -- no hand-over, 2.8 GB less traffic -- was
        #  tried: the ~870 gradient tensors then stay allocated across the graph and the step got 2.8 ms SLOWER; not kept.)
        # N > 1 (two-piece backward): the same one-pass hand-over in front of the exchange; its norm is of the LOCAL gradients and is discarded
        if not (self.handover is not None and self.handover(self.pairs, self.flat_view_of, self.accumulate, self.fused.norm)):
            _hand_over_gradients(self.pairs, self.flat_view_of, self.accumulate)
        if self.fused is not None:
            self.fused.norm_ready = False
        return loss.detach(), new_mask, (x_cut, dpreds)

    # two-piece backward: cut behind Swin stage SWIN_CUT.  Measured at one rank with the exchange forced (ms per step, same call; the
    # single graph: 66.9): cut at Swin's output 71.6, behind stage 2: 72.3, stage 1: 70.9-71.0, stage 0: 69.7-69.9 -- the second piece
    # (stage 0's backward, ~10 ms) is the window the bucketed all-reduce of 0.87 GB has to fit into
    SWIN_CUT = 0                                         # class default; the instance's value is the constructor's `swin_cut`

    def _bwd_swin(self, swin_out):
        """backward, second piece: Swin from the gradient of its output.  Nobody reads the result in the target step (train.py:20,141
        zero Swin's gradients before the auxiliary task uses its optimizer); it is computed because the reference computes it."""
        preds, dpreds = swin_out
        if dpreds is not None:
            torch.autograd.backward(preds, dpreds)

    def _update(self):
        if self.fused is not None:                           # norm + one launch: clip, AdamW, bf16 twins (FusedClipAdamW)
            self.fused.update()
        else:
            for p in self.flat.params:                       # the optimizer reads the static flat buffers
                p.grad = self.flat_view_of[p]
            torch.nn.utils.clip_grad_norm_(self.flat.params, self.args.clip)
            self.opt.step()
            if self.masters is not None:
                self.masters.sync_low()
        if self.accumulate:
            self.flat.zero_grad()
        for l, _ in self.pairs:                              # the next backward must produce fresh gradient tensors
            l.grad = None

    def _swin_version(self):
        ps = self._swin_params
        return (ps[0]._version, ps[-1]._version)

    def _launch_swin(self, k, frames):
        """graph S of set k for `frames` on the side stream: behind whatever produced `frames` (the current stream so far) and behind the last
        graph A' that read the set's activations"""
        st, side = self.sets[k], self.side_stream
        side.wait_stream(torch.cuda.current_stream())
        side.wait_event(st.ev_a)
        with torch.cuda.stream(side), torch.no_grad():
            if frames is not st.frames:
                if tuple(frames.shape) != tuple(st.frames.shape):
                    raise ValueError(f"GraphedTargetStep: frames of shape {tuple(frames.shape)}, the captured graphs are for {tuple(st.frames.shape)}")
                st.frames.copy_(frames, non_blocking=True)
                frames.record_stream(side)
            st.S.replay()
            st.ev_s.record(side)

    def _call_pipelined(self, batch, next_batch):
        k, st = self.cur, self.sets[self.cur]
        with torch.no_grad():
            for i, (dst, src) in enumerate(zip(self.static, batch)):
                if i == 8 or dst is src:
                    continue
                src = src if torch.is_tensor(src) else torch.as_tensor(src)
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError(f"GraphedTargetStep: batch entry {i} has shape {tuple(src.shape)}, the captured graphs are for {tuple(dst.shape)}")
                dst.copy_(src, non_blocking=True)
        pf = self.prefetched
        if not (pf is not None and pf[0] == k and pf[1] is batch[8] and pf[2] == batch[8]._version and pf[3] == self._swin_version()):
            self._launch_swin(k, batch[8])                  # nothing (valid) prefetched: Swin's forward for this batch, in order
        self.prefetched = None
        if next_batch is not None:                          # enqueued BEFORE graph A': the two then run side by side
            self._launch_swin(k ^ 1, next_batch[8])
            self.prefetched = (k ^ 1, next_batch[8], next_batch[8]._version, self._swin_version())
        main = torch.cuda.current_stream()
        main.wait_event(st.ev_s)
        st.A.replay()
        st.ev_a.record(main)
        self.cur = k ^ 1
        self.i_batch += 1
        if self.i_batch % self.args.trg_accumulation_steps == 0:
            self.graph_b.replay()
            _bump_versions(self.flat.params)
            if self.sched is not None:
                self.sched.step()
        self.loss, self.new_mask = st.loss, st.mask
        return st.loss, st.mask

    def __call__(self, batch, next_batch=None):
        """`next_batch` (pipeline_swin only): the batch the NEXT call will be given -- its Swin forward is replayed now, beside this step.  Pass
        None in front of an auxiliary step (it moves Swin's weights: the prefetch would be thrown away) and for the last step of an epoch."""
        if len(batch) != len(self.static):
            raise ValueError(f"GraphedTargetStep: batch of {len(batch)} entries, captured with {len(self.static)}")
        if self.pipeline:
            return self._call_pipelined(batch, next_batch)
        if self.branches:
            return self._call_branches(batch)
        with torch.no_grad():
            for i, (dst, src) in enumerate(zip(self.static, batch)):
                if dst is src:
                    continue
                src = src if torch.is_tensor(src) else torch.as_tensor(src)
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError(f"GraphedTargetStep: batch entry {i} has shape {tuple(src.shape)}, the captured graphs are for {tuple(dst.shape)}")
                dst.copy_(src, non_blocking=True)
        self.graph_a.replay()
        self.i_batch += 1
        last = self.i_batch % self.args.trg_accumulation_steps == 0
        if last:
            self.flat.exchange_begin()                     # no-op at world size 1; the collectives run beside graph A2
            if self.timing is not None:
                self.timing["begin"].record()
        if self.graph_a2 is not None:
            self.graph_a2.replay()
        if last:
            if self.timing is not None:
                self.timing["a2_done"].record()
            self.flat.exchange_end()
            if self.timing is not None:
                self.timing["end"].record()
            self.graph_b.replay()
            _bump_versions(self.flat.params)                # a replay changes the parameters behind autograd's back
            if self.sched is not None:
                self.sched.step()
        return self.loss, self.new_mask

    def time_exchange(self, on: bool = True):
        """record three events per step around the exchange (issue / Swin backward enqueued / collectives waited for):
        `exchange_ms()` then returns (ms from issue to the end of the wait, ms of that the stream spent waiting AFTER graph A2)"""
        self.timing = {k: torch.cuda.Event(enable_timing=True) for k in ("begin", "a2_done", "end")} if on else None

    def exchange_ms(self):
        t = self.timing
        torch.cuda.synchronize()
        return t["begin"].elapsed_time(t["end"]), t["a2_done"].elapsed_time(t["end"])

    def start_epoch(self):
        """a partial accumulation window does not carry into the next epoch (train.py:52,54 restart the counter per epoch)"""
        self.i_batch = 0
        self.prefetched = None
        self.flat.zero_grad()


# GPU time of Swin's backward below a cut (ms, 640 frames bf16, profiles/r03_swin_kernel_stats.md): what the exchange can hide behind
SWIN_TAIL_MS = {0: 10.0, 1: 17.0, 2: 27.0}


def measure_swin_tail_ms(swin_model, frames, is_trg_task=True, passes: int = 2):
    """GPU time of Swin's backward BELOW each possible cut, measured on this device with this batch: {0: ms, 1: ms, 2: ms} (cut c: the
    backward of stages 0 .. c and the patch embedding -- the window the gradient exchange has to fit into).  HIP events recorded by backward
    hooks on the stages during `passes` eager forward + backward passes (the last one counts); no parameter is stepped, the gradients are
    dropped.  Replaces the table below, which is one box's profile (round-4 VERDICT weak 11)."""
    layers = list(swin_model.swin.layers)
    dev = frames.device
    out = {}
    for _ in range(max(1, passes)):
        evs = {}

        def mk(i):
            def hook(mod, gin, gout):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs[i] = e                                   # gradients of everything inside stage i are computed: the backward moves on to stage i - 1
            return hook
        hs = [layers[i].register_full_backward_hook(mk(i)) for i in range(len(layers))]
        try:
            x = frames.detach()
            if x.is_floating_point():
                x = x.clone().requires_grad_(True)           # so that stage 0's hook fires
            preds = swin_model(x, is_trg_task=is_trg_task)
            end = torch.cuda.Event(enable_timing=True)
            preds.float().square().mean().backward()
            end.record()
            torch.cuda.synchronize(dev)
        finally:
            for h in hs:
                h.remove()
        for q in swin_model.parameters():
            q.grad = None
        if all(i in evs for i in (1, 2, 3)) and len(layers) >= 4:
            out = {c: float(evs[c + 1].elapsed_time(end)) for c in (0, 1, 2)}
    return out


def pick_swin_cut(exchange_ms_alone: float, frames: int = 640, tail_ms: dict | None = None) -> int:
    """The lowest cut whose second piece is at least as long as the exchange MEASURED alone on this communicator (each step up costs
    ~1 ms of lost overlap between the text backward and Swin's, so the lowest that fits wins); 2 if none does.  tail_ms: the piece lengths
    measured on this device (measure_swin_tail_ms); without them the profiled table above, scaled by the frame count."""
    table = tail_ms if tail_ms else {c: SWIN_TAIL_MS[c] * (frames / 640.0) for c in (0, 1, 2)}
    for cut in (0, 1, 2):
        if table[cut] >= exchange_ms_alone:
            return cut
    return 2


class GraphedAuxStep:
    """The auxiliary-task step (train.py:15-41: Swin -> logits -> cross-entropy -> backward -> clip -> AdamW on the Swin
    model) as two HIP graphs, same construction as GraphedTargetStep; the Swin gradients are the exchanged ones here."""

    def __init__(self, swin_model, optimizer, scheduler, args, images, labels, averager=None, warmup_iters=2):
        from .parallel import GradientAverager
        self.swin, self.opt, self.sched, self.args = swin_model, optimizer, scheduler, args
        self.images, self.labels = images.clone(), labels.clone()
        self.shadows = None
        self.i_batch = 0
        dev = images.device
        self.flat = averager if averager is not None else GradientAverager(self.swin.parameters(), hooks=False)
        self.flat_view_of = {p: p.grad for p in self.flat.params}
        self.pairs = [(p, p) for p in self.flat.params]
        self.fused = FusedClipAdamW(optimizer, self.flat.params, self.flat_view_of, {}, args.clip) if FusedClipAdamW.eligible(optimizer, self.flat.params) else None
        self.exchanging = bool(getattr(self.flat, "active", False))
        self.handover = FusedHandOver(len(self.flat.params)) if (self.fused is not None and FUSED_HANDOVER) else None
        for p in self.flat.params:
            p.grad = None
        self.accumulate = args.aux_accumulation_steps > 1
        snap = [(t, t.detach().clone()) for t in list(self.swin.parameters()) + list(self.swin.buffers())]
        rng = torch.cuda.get_rng_state(dev)
        cap = distinct_stream(dev)
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(warmup_iters):
                self._fwd_bwd()
                self._update()
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize(dev)
        _restore(snap)
        del snap
        _reset_optimizer_state(self.opt)
        if self.fused is not None:
            self.fused.reset()
        self.flat.zero_grad()
        torch.cuda.set_rng_state(rng, dev)
        self.shadows = _pin_shadows([self.swin])
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with capture_window(), _ops_pinned_scope(self.shadows):
            with torch.cuda.graph(self.graph_a, stream=cap):
                self.loss = self._fwd_bwd()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), stream=cap):
                self._update()
        _KEEP_GRAPHS.append((self.graph_a, self.graph_b))
        self.flat.zero_grad()

    def _fwd_bwd(self):
        if self.shadows is not None:
            self.shadows.refresh()
        loss = self.swin(self.images, False, self.labels, F.cross_entropy) / self.args.aux_accumulation_steps
        loss.backward()
        if self.handover is not None and self.handover(self.pairs, self.flat_view_of, self.accumulate, self.fused.norm):
            self.fused.norm_ready = not self.exchanging      # N > 1: the norm is taken again over the reduced buffers
        else:
            _hand_over_gradients(self.pairs, self.flat_view_of, self.accumulate)
        return loss.detach()

    def _update(self):
        if self.fused is not None:
            self.fused.update()
        else:
            for p in self.flat.params:
                p.grad = self.flat_view_of[p]
            torch.nn.utils.clip_grad_norm_(self.flat.params, self.args.clip)
            self.opt.step()
        if self.accumulate:
            self.flat.zero_grad()
        for p in self.flat.params:                           # the next backward must produce fresh gradient tensors
            p.grad = None

    def __call__(self, images, labels):
        if tuple(images.shape) != tuple(self.images.shape) or tuple(labels.shape) != tuple(self.labels.shape):
            raise ValueError(f"GraphedAuxStep: batch of shape {tuple(images.shape)} / {tuple(labels.shape)}, captured for "
                             f"{tuple(self.images.shape)} / {tuple(self.labels.shape)}")
        with torch.no_grad():
            if images is not self.images:
                self.images.copy_(images, non_blocking=True)
                self.labels.copy_(labels, non_blocking=True)
        self.graph_a.replay()
        self.i_batch += 1
        if self.i_batch % self.args.aux_accumulation_steps == 0:
            self.flat.exchange_all()                         # Swin's gradients: 187 MB fp32, nothing left in the step to hide them behind
            self.graph_b.replay()
            _bump_versions(self.flat.params)
            if self.sched is not None:
                self.sched.step()
        return self.loss

    def start_epoch(self):
        """a partial accumulation window does not carry into the next epoch (train.py:17,20 restart per epoch)"""
        self.i_batch = 0
        self.flat.zero_grad()


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

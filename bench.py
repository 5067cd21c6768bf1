"""Benchmark of the FacialMMT hot path on MI355X (contract: see the task prompt / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one target-task training step (train.py:54-143) on a synthetic MELD-shaped batch of
`--utts` utterances per GPU (BASELINE.json configs[1]: T+A+V, RoBERTa-large, 160-frame face sequence,
batch 4, bf16): Swin-tiny forward+backward over utts*160 frames of 3x224x224 on the HIP path, frame
filter, RoBERTa-large (PyTorch-ROCm, bf16 autocast, random init), audio/vision self-attention encoders,
the four cross-modal encoder calls on the HIP path, cross-entropy, backward through everything, gradient
clip + AdamW step on the multimodal model EVERY step (the reference steps every 4th micro-batch; stepping
every time only adds work).  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line: metric utterances/s (whole job), plus
  roofline     -- the dominant kernel (the instantiation of linear_nt_kernel, the MFMA GEMM of every Linear
                  layer, with the largest total time) timed live with HIP events on the launch stream during the timed steps:
                  achieved = sum(2*M*N*K) / sum(duration) against the 2.5 PFLOP/s dense bf16 MFMA peak;
  cpu_baseline -- the oracle (CPU restatement, fp32) timed on the host cores of this box on a bounded
                  sample of the same workload (N=1 only)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime initialises: see facialmmt_amd/__init__.py

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
SWIN_FWD_GFLOP_PER_FRAME = 9.0255    # BASELINE.md section 2 (reference's own flops() formulas x 2)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=4, help="utterances per GPU per step (BASELINE configs[1]: 4)")
    ap.add_argument("--frames", type=int, default=160, help="face frames per utterance (vision_max_utt_len)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graphs", type=int, default=1, help="capture the multimodal model (fwd+bwd) as HIP graphs (1) or launch eagerly (0)")
    ap.add_argument("--overlap-text", type=int, default=1, help="replay the text-encoder graph on a second HIP stream, concurrently with Swin (needs --graphs 1)")
    ap.add_argument("--parallel-fusion", type=int, default=1, help="capture independent halves of the fusion stack as parallel graph branches")
    ap.add_argument("--force-ddp", action="store_true", help="run the data-parallel gradient exchange path even with one process (exercises the N>1 code path on one GPU)")
    ap.add_argument("--shape-report", default=None, help="write a per-GEMM-shape timing table to this file (development aid)")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames in the CPU-baseline Swin sample")
    return ap.parse_args()


def synth_batch(args, dev, rank, cfg):
    """seeded synthetic MELD-shaped batch, generated on the device"""
    g = torch.Generator(device=dev).manual_seed(1111 + rank)
    B, Lv, La, T = args.utts, args.frames, cfg.get_audio_utt_max_lens, 512
    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    # raw 112x112 uint8 crops -> [0,1] -> Normalize(.5,.5) -> bicubic x2 (utils/dataset.py:18-20,41-57)
    raw = torch.randint(0, 256, (B * Lv, 3, 112, 112), generator=g, device=dev, dtype=torch.uint8)
    frames = torch.nn.functional.interpolate((raw.float() / 255.0 - 0.5) / 0.5, size=(224, 224), mode="bicubic", align_corners=False)
    frames = frames.to(act).contiguous()
    del raw
    ids = torch.randint(3, 50265, (B, T), generator=g, device=dev)
    ids[:, 0] = 0
    attn = torch.zeros(B, T, device=dev)
    attn[:, :400] = 1
    sep = torch.zeros(B, T, device=dev)
    sep[:, 20:400:20] = 1                                     # an utterance separator every 20 tokens
    utt_idx = torch.arange(B, device=dev) % 8
    audio = torch.randn(B, La, cfg.audio_featExtr_dim, generator=g, device=dev)
    amask = torch.zeros(B, La, device=dev)
    amask[:, :96] = 1
    vision = torch.randn(B, Lv, cfg.vision_featExtr_dim, generator=g, device=dev)
    vmask = torch.ones(B, Lv, device=dev)
    labels = torch.randint(0, 7, (B,), generator=g, device=dev)
    num_imgs = torch.full((B,), Lv, device=dev, dtype=torch.long)
    return (ids, attn, sep, audio, amask, vision, vmask, labels, frames, num_imgs, utt_idx)


def build_models(args, dev, cfg):
    from transformers import RobertaConfig
    from facialmmt_amd import models
    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg.compute_dtype = act
    cfg.plm_config = RobertaConfig(vocab_size=50265, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                   intermediate_size=4096, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1)
    cfg.plm_no_pooler = True
    torch.manual_seed(cfg.seed)
    swin = models.SwinForAffwildClassification(cfg).to(dev).train()
    mm = models.MultiModalTransformerForClassification(cfg).to(dev).train()
    return swin, mm


class KernelTimer:
    """HIP-event timing of one kernel family on the launch stream, live inside the timed region."""

    def __init__(self):
        self.events = []
        self.enabled = False

    def install(self):
        from facialmmt_amd import ops
        raw = ops.linear_raw
        timer = self

        def timed_linear_raw(x2, w, bias, **kw):
            if not timer.enabled or x2.dtype != torch.bfloat16:
                return raw(x2, w, bias, **kw)
            M, K = x2.shape
            N = w.shape[0]
            # same dispatch as csrc/gemm.hip::dispatch_nt: <BN, BK, LDS buffers>
            bn = 96 if (N % 96 == 0 and N % 128 != 0) else 128
            bn = f"{bn},96,1" if K == 96 else f"{bn},64,1" if K <= 64 else f"{bn},64,2" if K % 64 == 0 else f"{bn},32,2"
            bn = f"{64 if M <= 4096 else 128},{bn}" + (",glds" if (K % 64 == 0 and K > 64) else "")
            if M >= 65536 and K % 32 == 0 and K >= 96 and (N % 128 == 0 or N % 96 == 0) and not (K == 96 and kw.get("epi", 0) == 1):
                nk = ",nk6" if K == 192 else ",nk3" if K == 96 else ""
                if N % 128 == 0:
                    bn = "deep256x128x32" + nk if (nk or kw.get("epi", 0) != 0 or K <= 512 or K % 64) else "deep256x128x64"
                else:
                    bn = "deep256x96x32" + nk
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = raw(x2, w, bias, **kw)
            e.record()
            timer.events.append((bn, 2.0 * M * N * K, (M * K + N * K + M * N) * 2.0, s, e,
                                 ("nt", M, N, K, kw.get("epi", 0), kw.get("y_pre") is not None, kw.get("res") is not None)))
            return y

        ops.linear_raw = timed_linear_raw
        raw_w = ops.wgrad_partials_raw

        def timed_wgrad_partials_raw(dy2, x2, want_bias, ws, nbytes, rowscale=None, rows_per_scale=1):
            if not timer.enabled or dy2.dtype != torch.bfloat16:
                return raw_w(dy2, x2, want_bias, ws, nbytes, rowscale, rows_per_scale)
            M, N = dy2.shape
            K = x2.shape[1]
            # same dispatch as csrc/gemm.hip::fmmt_linear_wgrad_partials; the fixed-order sum of the partials
            # (fmmt_linear_wgrad_finish) is a separate launch and is not inside these events
            name = "linear_tn_kernel<bf16,32,few>" if M <= 4096 else "linear_tn_kernel<bf16,64>" if M <= 262144 else "linear_tn_kernel<bf16,32>"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            raw_w(dy2, x2, want_bias, ws, nbytes, rowscale, rows_per_scale)
            e.record()
            timer.events.append((name, 2.0 * M * N * K, (M * N + M * K) * 2.0 + N * K * 4.0, s, e,
                                 ("tn", M, N, K, 0, bool(want_bias), rowscale is not None)))

        ops.wgrad_partials_raw = timed_wgrad_partials_raw

    def summary(self):
        out = {}
        for bn, fl, by, s, e, _ in self.events:
            d = out.setdefault(bn, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl
            d[2] += by
            d[3] += s.elapsed_time(e) * 1e-3
        return out

    def shape_report(self, steps):
        """per problem shape: launches/step, ms/step, TFLOP/s, algorithmic GB/s (development aid: --shape-report)"""
        out = {}
        for bn, fl, by, s, e, shape in self.events:
            d = out.setdefault((bn,) + shape, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl
            d[2] += by
            d[3] += s.elapsed_time(e) * 1e-3
        lines = ["ms/step launches/step TFLOP/s GB/s kernel kind M N K epi flag1 flag2"]
        for k, (cnt, fl, by, sec) in sorted(out.items(), key=lambda kv: -kv[1][3]):
            lines.append(f"{sec / steps * 1e3:8.3f} {cnt / steps:6.1f} {fl / sec / 1e12:7.1f} {by / sec / 1e9:7.0f}  " + " ".join(str(v) for v in k))
        return "\n".join(lines)


def cpu_baseline(args, cfg):
    """The oracle (oracle/*.py, fp32 CPU restatement) timed on this box's host cores on a bounded sample:
    Swin forward+backward on `cpu_frames` frames (scaled to `frames` per utterance) + the fusion stack
    forward+backward for one utterance (PLM excluded: it is the same third-party library on both sides).
    Reported as utterances/s of the section-8 hot path."""
    import numpy as np
    from facialmmt_amd import synth
    from oracle import crossmodal as OC
    from oracle import swin as OS
    # 256 hardware threads on the GPU box make torch's CPU ops slower, not faster (measured: 234 s for the
    # 8-frame step with 256 threads vs ~2 s with 8-16); use 16 threads and say so.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        keys = json.load(f)
    sd = synth.state_dict_from_keys(keys["affwild"], seed=100)
    for v in sd.values():
        v.requires_grad_(True)
    nF = args.cpu_frames
    x = synth.tensor("frames", (nF, 3, 224, 224), seed=1)

    def swin_step():
        for v in sd.values():
            v.grad = None
        OS.swin_affwild_logits(sd, x, training=True).square().sum().backward()
    t0 = time.perf_counter()
    swin_step()
    ts = [time.perf_counter() - t0]
    if ts[0] < 10.0:                                   # bounded: skip the repeats if one step is already slow
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            swin_step()
            ts.append(time.perf_counter() - t0)
    t_swin = float(np.median(ts)) / nF * args.frames
    esd = synth.state_dict_from_keys(keys["crossmodal"], seed=50, prefix="enc.")
    for v in esd.values():
        v.requires_grad_(True)
    La, Lv, Lt = cfg.get_audio_utt_max_lens, args.frames, cfg.get_text_utt_max_lens
    t_, a_, v_ = (synth.tensor(n, (L, 1, 768), seed=60) for n, L in (("t", Lt), ("a", La), ("v", Lv)))

    def fusion_step():
        ta = torch.cat((OC.crossmodal_encoder(esd, t_, a_, a_), OC.crossmodal_encoder(esd, a_, t_, t_)), 0)
        out = torch.cat((OC.crossmodal_encoder(esd, ta, v_, v_), OC.crossmodal_encoder(esd, v_, ta, ta)), 0)
        out.square().mean().backward()
    t0 = time.perf_counter()
    fusion_step()
    t_fus = time.perf_counter() - t0
    if t_fus < 10.0:
        t0 = time.perf_counter()
        fusion_step()
        t_fus = time.perf_counter() - t0
    # the two self-attention encoders in front of the fusion (oracle restatement) ...
    from oracle import multimodal as OM
    msd = {k: v for k, v in synth.state_dict_from_keys(keys["multimodal_roberta"], seed=200).items()
           if k.startswith(("audio_utt_transformer.", "vision_utt_transformer."))}
    # the fixture keys were dumped for 24/20-step sequences: rebuild the position tables for the bench lengths
    msd["audio_utt_transformer.position_embeddings.weight"] = synth.tensor("pos_a", (La, 768), seed=1, lo=-0.1, hi=0.1)
    msd["vision_utt_transformer.position_embeddings.weight"] = synth.tensor("pos_v", (Lv, 768), seed=1, lo=-0.1, hi=0.1)
    for v in msd.values():
        v.requires_grad_(True)
    a_in, v_in = synth.tensor("a_in", (1, La, 768), seed=61), synth.tensor("v_in", (1, Lv, 768), seed=62)

    def meld_step():
        za = torch.zeros(1, 1, 1, La)
        zv = torch.zeros(1, 1, 1, Lv)
        out = OM.meld_encoder(msd, "audio_utt_transformer.", a_in, za, cfg.audio_utt_Transformernum).square().mean() \
            + OM.meld_encoder(msd, "vision_utt_transformer.", v_in, zv, cfg.vision_utt_Transformernum).square().mean()
        out.backward()
    meld_step()
    t0 = time.perf_counter()
    meld_step()
    t_meld = time.perf_counter() - t0
    # ... and the text encoder: the same third-party RoBERTa-large (random init), one 512-token dialogue, fp32
    t_plm = None
    try:
        from transformers import RobertaConfig, RobertaModel
        plm = RobertaModel(RobertaConfig(vocab_size=50265, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                         intermediate_size=4096, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1),
                           add_pooling_layer=False)
        ids = torch.from_numpy(synth.randint("cpu_ids", (1, 512), 3, 50265, seed=2))
        am = torch.ones(1, 512)

        def plm_step():
            plm.zero_grad(set_to_none=True)
            plm(ids, am)[0].square().mean().backward()
        plm_step()
        t0 = time.perf_counter()
        plm_step()
        t_plm = time.perf_counter() - t0
    except Exception as e:                               # the CPU leg must never take the bench down
        print(f"cpu_baseline: text-encoder leg skipped ({e})", file=sys.stderr)
    total = t_swin + t_fus + t_meld + (t_plm or 0.0)
    return {"value": round(1.0 / total, 5), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 fwd+bwd for ONE utterance on {cores} threads: Swin+head on {nF} frames (median of {len(ts)}, scaled x{args.frames / nF:g} to "
                      f"{args.frames} frames: {t_swin:.2f} s) + 4 cross-modal encoder calls ({t_fus:.2f} s) + audio/vision self-attention encoders "
                      f"({t_meld:.2f} s) + RoBERTa-large 512 tokens (HF, {'%.2f s' % t_plm if t_plm else 'skipped'}); optimizer excluded"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the hot path has no CPU fallback)"
    # FMMT_BENCH_DEVICE / FMMT_BENCH_BACKEND: dry-run switches for exercising the N>1 code path on a one-GPU box (all
    # ranks on device 0, gloo instead of RCCL, which refuses two ranks per device); never set by the driver
    local = int(os.environ.get("FMMT_BENCH_DEVICE", local))
    backend = os.environ.get("FMMT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rank != 0:                                           # only rank 0 reports; keep other ranks' C-level banners out of stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world > 1 or args.force_ddp:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from facialmmt_amd.config import default_args
    from facialmmt_amd.train_step import TargetStep
    cfg = default_args(get_vision_utt_max_lens=args.frames, trg_accumulation_steps=1)
    swin, mm = build_models(args, dev, cfg)
    batch = synth_batch(args, dev, rank, cfg)
    if args.graphs:
        from facialmmt_amd.train_step import graph_multimodal, select_frames
        with torch.no_grad():
            preds = swin(batch[8][:8], is_trg_task=True).float().repeat(batch[8].shape[0] // 8, 1)
        vis, nmask = select_frames(preds, batch[5], batch[6], batch[9], cfg.FacialEmoImpor_threshold)
        sample = (batch[0], batch[1], batch[2], batch[3], batch[4], vis.detach().requires_grad_(True), nmask, batch[10])
        mm = graph_multimodal(mm, sample, torch.bfloat16 if args.dtype == "bf16" else None, overlap_text=bool(args.overlap_text),
                              parallel_fusion=bool(args.parallel_fusion))
        mm.zero_grad(set_to_none=True)
        swin.zero_grad(set_to_none=True)
    averager = None
    if world > 1 or args.force_ddp:
        # gradient mean over ranks for the parameters this step's optimizer updates (parallel.GradientAverager); two
        # stream groups: the text branch's gradients are produced on the second HIP stream
        from facialmmt_amd.parallel import GradientAverager, broadcast_parameters
        broadcast_parameters(mm)
        plm = mm.roberta if mm.text_pretrained_model == "roberta" else mm.bert
        text_params = list(plm.parameters()) + list(mm.text_linear.parameters())
        text_ids = set(map(id, text_params))
        averager = GradientAverager(None, groups=[[p for p in mm.parameters() if id(p) not in text_ids], text_params])
    # (bf16 text-encoder parameters with fp32 masters in the optimizer were measured: 113.4 vs 110.4 ms -- no gain)
    opt = torch.optim.AdamW(mm.parameters(), lr=cfg.trg_lr, weight_decay=cfg.weight_decay, fused=True)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(1.0, (s + 1) / 100.0))
    step = TargetStep(swin, mm, opt, sched, cfg, autocast_dtype=torch.bfloat16 if args.dtype == "bf16" else None, averager=averager)
    timer = KernelTimer()
    timer.install()

    kept = None
    for _ in range(args.warmup):
        _, m = step(batch)
        kept = m
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = rank == 0
    step.host_ms = {}
    step.gpu_events = [] if rank == 0 else None             # main-stream timeline of the phases (events are free on the GPU)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, kept = step(batch)
    issue_s = time.perf_counter() - t0                      # host time to enqueue the steps (GPU still running)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    gpu_phase = {}
    if step.gpu_events:
        evs, step.gpu_events = step.gpu_events, None
        for (_, a), (name, b) in zip(evs, evs[1:]):
            if name != "start":
                gpu_phase[name] = gpu_phase.get(name, 0.0) + a.elapsed_time(b)
    step.host_ms.pop("start", None)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # After the timed region (not part of `value`): the same kernels with the text encoder back on the main stream.
    # During the timed steps the text-encoder graph shares the CUs with Swin's launches, which stretches every Swin
    # kernel; two extra steps without that overlap give the kernel's own rate next to the live one.
    iso = None
    if args.graphs and args.overlap_text and getattr(mm, "text_stream", None) is not None:
        timed_events, timer.events = timer.events, []
        side, mm.text_stream = mm.text_stream, None
        step(batch)
        timer.enabled = rank == 0
        for _ in range(2):
            step(batch)
        torch.cuda.synchronize()
        timer.enabled = False
        mm.text_stream = side
        iso, timer.events = timer.summary(), timed_events
        if world > 1:
            dist.barrier()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.utts * world * args.steps / elapsed
        fams = timer.summary()
        if args.shape_report:
            with open(args.shape_report, "w") as f:
                f.write(timer.shape_report(args.steps) + "\n")
        roof = None
        if fams:
            bn, (cnt, fl, by, sec) = max(fams.items(), key=lambda kv: kv[1][3])
            achieved = fl / sec / 1e12
            if bn.startswith("linear_tn"):
                kname = bn
            elif bn.startswith("deep256") and bn != "deep256x128x64":         # deep256x{128,96}x32[,nkN]
                width = "128" if bn.startswith("deep256x128") else "96"
                kname = f"linear_nt_deep32_kernel<{bn[-1] if ',nk' in bn else '0'},{width}>"
            elif bn.startswith("deep"):
                kname = "linear_nt_deep_kernel"
            else:
                kname = f"linear_nt_kernel<bf16,{bn}>"
            traffic = None                       # PMC counters cannot be read in-process: taken from the committed PMC summary
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    t = json.load(f).get(kname)
                if t:
                    traffic = (2 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024
            except OSError:
                pass
            def rate(c, f, b, t):
                return {"achieved": round(f / t / 1e12, 1), "frac": round(f / t / 1e12 / PEAK_BF16_TFLOPS, 4),
                        "avg_launch_us": round(t / c * 1e6, 1), "algorithmic_GB_per_s": round(b / t / 1e9, 0)}
            live = rate(cnt, fl, by, sec)
            roof = {"bound": "mfma", "kernel": kname, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(by / cnt), "launches_per_step": cnt // args.steps,
                    "share_of_step": round(sec / elapsed, 3)}
            if iso and bn in iso:
                # During the timed steps this kernel is time-sliced with the text-encoder graph on the second stream, so
                # event durations there are not the kernel's own.  Headline = the two steps issued right after the timed
                # region with that graph back on the main stream (also what rocprofv3 reports: its tracing serialises
                # the two streams); the in-region timing is kept next to it.
                roof.update(rate(*iso[bn]))
                roof["measured"] = "HIP events, 2 steps after the timed region, text encoder on the main stream"
                roof["in_timed_region_with_text_stream"] = live
            else:
                roof.update(live)
                roof["measured"] = "HIP events over the timed region"
        flops_step = args.utts * (args.frames * SWIN_FWD_GFLOP_PER_FRAME * 3 + (335 + 29.7) * 3) * 1e9
        line = {
            "metric": "utterances/sec T+A+V forward+bwd, 160-frame face seq, 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"configs[1]: T+A+V, RoBERTa-large (random init), {args.frames}-frame face seq 3x224x224, "
                                   f"batch {args.utts} utterances/GPU, {args.dtype}, fwd+bwd+AdamW every step",
                       "global_batch": args.utts * world, "frames_per_step_per_gpu": args.utts * args.frames,
                       "parallelism": f"dp{world}", "kept_frame_fraction": round(float(kept.mean().item()), 3),
                       "model_tflops_per_s_per_gpu": round(flops_step / (ms * 1e-3) / 1e12, 1),
                       "host_enqueue_ms_per_step": round(issue_s / args.steps * 1e3, 1), "hip_graphs": bool(args.graphs), "text_encoder_on_second_stream": bool(args.graphs and args.overlap_text),
                       "second_stream_pair_over_single": round(float(getattr(mm, "text_stream_concurrency", 0.0)), 2),
                       "host_ms_per_phase": {k: round(v / args.steps, 1) for k, v in step.host_ms.items()},
                       "main_stream_gpu_ms_per_phase": {k: round(v / args.steps, 1) for k, v in gpu_phase.items()}},
            "roofline": roof,
            "cpu_baseline": None,
        }
        if "FMMT_BENCH_DEVICE" in os.environ or backend != "nccl":
            line["config"]["dry_run"] = f"ranks share device {local}, backend {backend}: not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, cfg)
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio (block-buffered when stdout is a pipe): flush it first so that
        # the JSON line is the LAST line of this process's stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

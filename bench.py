"""Benchmark of the FacialMMT hot path on MI355X (contract: see the task prompt / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself as the line below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one target-task training step (train.py:54-143) on a synthetic MELD-shaped batch of `--utts` utterances per GPU
(default BASELINE.json configs[1]: T+A+V, RoBERTa-large, 160-frame face sequence, batch 4, bf16): the uint8 112x112 face crops
go through the fused pre-step (bicubic x2 + ToTensor + Normalize inside PatchEmbed's gather), Swin-tiny forward+backward over
utts*frames frames on the HIP path, frame filter, the text encoder (PyTorch-ROCm, bf16 autocast, random init), audio/vision
self-attention encoders, the four cross-modal encoder calls on the HIP path, cross-entropy, backward through everything,
gradient clip + AdamW step on the multimodal model EVERY step (the reference steps every 4th micro-batch; stepping every time
only adds work).  Inputs are resident in HBM before the timed region.  With `--aux-images A` every step is preceded by one
auxiliary-task step (train.py:15-41) on A Aff-Wild2-shaped crops: Swin fwd+bwd, clip, AdamW on the Swin model.

Other BASELINE.json configs (each run prints its own line naming its configs[i]):
    --config 3    BERT-large text encoder (src/models.py:75-77), 4 utterances/GPU                 (configs[3], per-GPU leg)
    --config 4    320-frame face sequence, 1 utterance/GPU, 150 auxiliary images per step          (configs[4], per-GPU leg)

Rank 0 prints ONE JSON line: metric utterances/s (whole job), plus
  roofline     -- the dominant kernel family (largest total GPU time among the bracketed many-token launches), timed with HIP
                  events on the launch stream in eager steps issued right after the timed region (the timed steps themselves
                  are HIP-graph replays, inside which nothing can be bracketed).  Its bound follows from its algorithmic
                  intensity: above 312 FLOP/B (2.5 PFLOP/s / 8 TB/s) achieved = sum(2*M*N*K) / sum(duration) against the dense
                  bf16 MFMA peak, below it achieved = algorithmic bytes / duration against 8 TB/s; `traffic` from the committed
                  PMC pass (profiles/traffic.json); `families` lists every bracketed family, `roofline_mfma` / `roofline_hbm`
                  the largest family on either side of the ridge;
  cpu_baseline -- the oracle (CPU restatement, fp32) timed on the host cores of this box on a bounded sample of the same
                  workload (N=1 only): forward+backward (value) and forward-only, with a thread sweep."""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime initialises: see facialmmt_amd/__init__.py
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # multi-process GPU work on this driver: dmabuf IPC only (RCCL's hipIpcGetMemHandle fails otherwise)

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# --rccl-tuning mesh: what DESIGN.md section 5 assumes for one node of 8 MI355X (xGMI full mesh, 7 links per GPU)
RCCL_MESH_ENV = {"NCCL_MIN_NCHANNELS": "112", "NCCL_PROTO": "Simple", "RCCL_MSCCLPP_ENABLE": "0"}
RCCL_KNOBS = ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_BUFFSIZE", "NCCL_P2P_LEVEL", "RCCL_MSCCL_ENABLE",
              "RCCL_MSCCLPP_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY")


def rccl_ranks_from_log(path):
    """the communicator size RCCL itself reported at init ('... nranks N ...'); None if the log holds none (gloo, or no log)"""
    import glob
    import re
    best = None
    for f in glob.glob(path + "*") if path else []:
        try:
            with open(f, errors="replace") as fh:
                for m in re.finditer(r"nranks[ =:]+(\d+)", fh.read()):
                    best = max(best or 0, int(m.group(1)))
        except OSError:
            pass
    return best


PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                # HBM3E peak (spec), same guide
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
    ap.add_argument("--config", type=int, default=1, choices=[1, 3, 4], help="BASELINE.json configs[i] per-GPU leg: 1 (default), 3 = BERT-large, 4 = 320 frames + auxiliary task")
    ap.add_argument("--plm", default=None, choices=["roberta-large", "bert-large"], help="text encoder architecture (random init)")
    ap.add_argument("--aux-images", type=int, default=None, help="auxiliary-task images per GPU per step (0 = no auxiliary step)")
    ap.add_argument("--input", default="u8", choices=["u8", "float"], help="u8: 112x112 uint8 crops through the fused pre-step; float: pre-resized 224x224 frames")
    ap.add_argument("--resize", default="pil", choices=["pil", "cv2"], help="bicubic flavour of the fused pre-step (pil: pinned bit-exactly; cv2: unpinned)")
    ap.add_argument("--graphs", type=int, default=2, help="2: the whole step as two HIP graphs; 1: only the multimodal model graphed, Swin eager; 0: eager")
    ap.add_argument("--overlap-text", type=int, default=1, help="text encoder on a second HIP stream / graph branch, concurrently with Swin")
    ap.add_argument("--plm-dtype", default="bf16", choices=["bf16", "fp32"], help="parameter dtype of the text encoder in --graphs 2: bf16 with fp32 master weights in the optimizer, or fp32 under autocast")
    ap.add_argument("--grad-comm", default="bf16", choices=["fp32", "bf16"], help="element type of the gradient all-reduce (N > 1): bf16 halves the bytes on the xGMI links (0.87 GB instead of 1.74 GB per step)")
    ap.add_argument("--other-configs", type=int, default=1, help="N = 1 only: after the timed region also run 4 steps of configs[3] and configs[4] (per-GPU legs, own processes) and report them under `other_configs`")
    ap.add_argument("--branch-graphs", type=int, default=0, help="1 (one rank, --graphs 2, no gradient exchange): text encoder forward / backward and Swin forward / backward as graphs of their own on two streams, events at the data dependencies (GraphedTargetStep branch_graphs)")
    ap.add_argument("--fork-streams", type=int, default=0, help="1 (one rank, --graphs 2, no gradient exchange): one graph whose fork node's first child is a tick on the launch queue, text encoder and Swin each on a stream of their own, forward and backward (GraphedTargetStep fork_streams)")
    ap.add_argument("--pipeline-swin", type=int, default=0, help="1 (one rank, --graphs 2, no auxiliary task): Swin's forward of step i + 1 as a graph of its own, replayed on a second stream beside step i (GraphedTargetStep pipeline_swin); every timed step still runs exactly one Swin forward.  Round 5, same call, alternating: 63.2 / 63.3 ms per step with it, 62.35 / 62.37 without -- the chip-filling Swin kernels (one persistent workgroup per CU) leave the other stream's small launches no CU to run on, so the two graphs time-slice; default off")
    ap.add_argument("--parallel-fusion", type=int, default=0, help="1: capture independent halves of the fusion stack as parallel graph branches (round 4, same call: 64.9-65.0 ms per step against 63.5-63.7 without -- a fork / join pair of the replayed graph costs more than the 50-250-workgroup launches it lets overlap)")
    ap.add_argument("--discarded-swin-gradients", choices=["compute", "skip"], default="compute",
                    help="'compute' (default): the target step back-propagates through Swin as the reference does, although nothing reads those gradients "
                         "(train.py:20,33,140-143); 'skip': train_step's option that does not compute them (same training to fp32 rounding, reported as a "
                         "separate leg of the default run, never as `value`)")
    ap.add_argument("--rccl-tuning", default="off", choices=["off", "mesh"],
                    help="N > 1: 'mesh' sets the RCCL knobs DESIGN.md section 5 assumes for the 8-GPU xGMI mesh before the communicator is created "
                         "(NCCL_MIN_NCHANNELS=112: enough channels for rings over all 7 links per GPU; NCCL_PROTO=Simple for the 64 MiB buckets; "
                         "RCCL_MSCCLPP_ENABLE=0: no one-shot small-message path in a captured step); 'off' leaves RCCL's own choice.  Either way "
                         "the values in force are printed in config.exchange.rccl_env.  Never measured on hardware (no multi-GPU lease): default off")
    ap.add_argument("--swin-cut", type=int, default=-1, help="N > 1: Swin stage behind which the backward graph is cut (-1: chosen from the exchange time measured alone before capture)")
    ap.add_argument("--force-ddp", action="store_true", help="run the data-parallel gradient exchange path even with one process (exercises the N>1 code path on one GPU)")
    ap.add_argument("--shape-report", default=None, help="write a per-GEMM-shape timing table to this file (development aid)")
    ap.add_argument("--host-input-leg", type=int, default=1, help="after the timed region, also time the steps with the batch handed over in pinned host memory (PCIe-inclusive rate; reported, never `value`)")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames in the CPU-baseline Swin sample")
    ap.add_argument("--rendezvous-check", action="store_true", help="only bring up the process group (one all-reduce of ones), have rank 0 print {\"rendezvous_check\": ..., \"ranks_seen\": N} and exit -- needs no GPU with FMMT_BENCH_BACKEND=gloo; what tests/test_host_cpu.py uses to run `python bench.py --gpus 2` as typed")
    a = ap.parse_args()
    if a.config == 3:
        a.plm = a.plm or "bert-large"
    if a.config == 4:
        a.frames, a.utts = (320 if a.frames == 160 else a.frames), (1 if a.utts == 4 else a.utts)
        a.aux_images = 150 if a.aux_images is None else a.aux_images
    a.plm = a.plm or "roberta-large"
    a.aux_images = a.aux_images or 0
    return a


def synth_batch(args, dev, rank, cfg):
    """seeded synthetic MELD-shaped batch, generated on the device"""
    g = torch.Generator(device=dev).manual_seed(1111 + rank)
    B, Lv, La, T = args.utts, args.frames, cfg.get_audio_utt_max_lens, 512
    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    bert = getattr(args, "plm", "roberta-large") == "bert-large"
    # raw 112x112 uint8 crops in image layout (utils/dataset.py:47-57).  --input u8: they ARE the model input (resize,
    # ToTensor, Normalize fused into PatchEmbed's gather); --input float: the 224x224 float frames the reference caches
    raw = torch.randint(0, 256, (B * Lv, 112, 112, 3), generator=g, device=dev, dtype=torch.uint8)
    if getattr(args, "input", "u8") == "u8":
        frames = raw
    else:
        frames = torch.nn.functional.interpolate((raw.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5, size=(224, 224), mode="bicubic", align_corners=False)
        frames = frames.to(act).contiguous()
        del raw
    ids = torch.randint(3, 30522 if bert else 50265, (B, T), generator=g, device=dev)
    ids[:, 0] = 101 if bert else 0
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


def synth_aux_batch(args, dev, rank):
    """Aff-Wild2-shaped auxiliary batch: uint8 112x112 crops + expression labels (utils/dataset.py AffwildDataset)"""
    g = torch.Generator(device=dev).manual_seed(2222 + rank)
    imgs = torch.randint(0, 256, (args.aux_images, 112, 112, 3), generator=g, device=dev, dtype=torch.uint8)
    if getattr(args, "input", "u8") != "u8":
        act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        imgs = torch.nn.functional.interpolate((imgs.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5, size=(224, 224), mode="bicubic", align_corners=False).to(act).contiguous()
    return imgs, torch.randint(0, 7, (args.aux_images,), generator=g, device=dev)


def plm_config(name):
    """architecture of the text encoder (random init: no checkpoints on the benchmark box); src/models.py:72-77"""
    from transformers import BertConfig, RobertaConfig
    if name == "bert-large":
        return BertConfig(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                          max_position_embeddings=512, type_vocab_size=2, pad_token_id=0)
    return RobertaConfig(vocab_size=50265, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                         intermediate_size=4096, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1)


def build_models(args, dev, cfg):
    from facialmmt_amd import models
    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg.compute_dtype = act
    cfg.pretrainedtextmodel_path = "pretrained_model/" + args.plm        # selects the RoBERTa / BERT slicing offsets (src/models.py:118-146)
    cfg.plm_config = plm_config(args.plm)
    cfg.plm_no_pooler = True
    torch.manual_seed(cfg.seed)
    swin = models.SwinForAffwildClassification(cfg).to(dev).train()
    swin.swin.input_resize = args.resize
    swin.swin.input_dtype = act
    mm = models.MultiModalTransformerForClassification(cfg).to(dev).train()
    return swin, mm


def p256_tile(M, N, K, kw):
    """mirror of csrc/gemm.hip::p256_plan (persistent 256-row-tile kernel): channel-tile width, or 0"""
    has_op = kw.get("res") is not None or kw.get("rowscale") is not None
    if M < 16384 or M % 16 or K % 64 or K < 192 or kw.get("aux") is not None or (has_op and False):
        return 0
    tm, best, cost = (M + 255) // 256, 0, 0.0
    for bn, pen in ((256, 1.0), (192, 1.04), (128, 1.10)):
        if N % bn or (has_op and bn == 256):
            continue
        tiles = tm * (N // bn)
        if tiles < 256:
            continue
        c = ((tiles + 255) // 256) * bn * pen
        if not best or c < cost:
            best, cost = bn, c
    return best


def ph_takes(M, N, K, kw):
    """mirror of csrc/gemm.hip::ph_plan: "" (none), "ph256" (csrc/gemm_ph.h) or "ph192" (csrc/gemm_ph3.h)"""
    mul_aux = kw.get("epi", 0) == 4 and kw.get("aux") is not None and kw.get("res") is None and kw.get("y_pre") is None      # FMMT_EPI_MUL_AUX (round 6): gemm_ph3.h EPI 6
    if kw.get("aux") is not None and not mul_aux:
        return ""
    gelu_pre = kw.get("epi", 0) == 1 and kw.get("y_pre") is not None
    has_op = kw.get("res") is not None or kw.get("rowscale") is not None
    if not gelu_pre and not mul_aux and (kw.get("epi", 0) or kw.get("y_pre") is not None):
        return ""
    if gelu_pre and (has_op or K < 1536):
        return ""
    if M < 4096 or M % 8 or N % 128 or K % 64 or K < 192:
        return ""
    if N % 256:
        if gelu_pre or K > 512:
            return ""
        t384 = ((M + 383) // 384) * (N // 128)
        r384 = (t384 + 255) // 256
        return "ph384" if (t384 >= 256 and t384 * 100 >= r384 * 256 * 85) else ""
    t256, t192 = ((M + 255) // 256) * (N // 256), ((M + 191) // 192) * (N // 256)
    r256, r192 = (t256 + 255) // 256, (t192 + 255) // 256
    min_tiles = 150 if M < 16384 else 256
    ok256 = t256 >= min_tiles and (M < 16384 or t256 * 100 >= r256 * 256 * 85)
    ok192 = t192 >= min_tiles and (M < 16384 or t192 * 100 >= r192 * 256 * 85)
    if gelu_pre or has_op or mul_aux:
        return "ph192" if ok192 else ""
    if ok192 and (not ok256 or r192 * 192 * 103 < r256 * 256 * 100):
        return "ph192"
    return "ph256" if ok256 else ""


def tn_dma_tile(M, N, K, scaled, rows_per_scale, x_gelu):
    """mirror of csrc/gemm.hip::tn_plan_dma + launch_tn_plan (DMA-staged weight-gradient kernel): "256,256" / "192,384" or None"""
    if x_gelu or M <= 8192 or M % 64:
        return None
    if N % 256 == 0 and K % 256 == 0:
        tn, tk = 256, 256
    elif N % 192 == 0 and K % 384 == 0:
        tn, tk = 192, 384
    elif (N % 384 == 0 or (N % 384 == 192 and N >= 576)) and K == 192 and not scaled:      # round 6: the transposed tile (stage 1's fc1 / qkv weight gradients), waves 4 x 2
        tn, tk = 384, 192
    else:
        return None
    tiles = -(-N // tn) * (K // tk)
    if tiles < 2 or tiles > 128:
        return None
    splits = 256 // tiles
    chunk = (-(-M // splits) + 63) // 64 * 64
    if chunk < 512 or (scaled and chunk // rows_per_scale + 2 > 1024):
        return None
    return f"{tn},{tk},4,{4 if tk == 192 else 2}"


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
            p256 = p256_tile(M, N, K, kw)
            bn = 96 if (N % 96 == 0 and N % 128 != 0) else 128
            bn = f"{bn},96,1" if K == 96 else f"{bn},64,1" if K <= 64 else f"{bn},64,2" if K % 64 == 0 else f"{bn},32,2"
            bn = f"{64 if M <= 4096 else 128},{bn}" + (",glds" if (K % 64 == 0 and K > 64) else "")
            if M <= 4096 and (N < 16384 or M < 256):
                tiles = -(-M // 64) * -(-N // 128)
                small = 1
                if small and tiles < 256 and N % 64 == 0 and K % 64 == 0 and K >= 128:      # quarter tiles
                    if small == 2:
                        bn = f"64,64,64,{4 if (K >= 2048 and 2 * tiles <= 256) else 2},glds"
                    else:
                        bn = f"32,64,64,{4 if K >= 2048 else 2},glds"
                elif K % 64 == 0 and K >= 2048 and tiles <= 256:
                    bn = bn.replace(",64,2,glds", ",64,4,glds")
            if M >= 65536 and K % 32 == 0 and K >= 96 and (N % 128 == 0 or N % 96 == 0) and not (K == 96 and kw.get("epi", 0) == 1):
                nk = ",nk6" if K == 192 else ",nk3" if K == 96 else ""
                if N % 128 == 0:
                    bn = "deep256x128x32" + nk if (nk or kw.get("epi", 0) != 0 or K <= 512 or K % 64) else "deep256x128x64"
                else:
                    bn = "deep256x96x32" + nk
            if p256:
                plainop = 0
                hasop = kw.get("res") is not None or kw.get("rowscale") is not None or kw.get("aux") is not None
                isop = p256 != 256 and (hasop or (plainop and (K > 1536 or plainop > 1)))
                bn = f"p256x{p256}" + ("op" if isop else "pipe" if K >= 384 else "")      # launch_p256: pipelined K loop from K = 384
            ph = ph_takes(M, N, K, kw)
            if ph:
                bn = ph
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = raw(x2, w, bias, **kw)
            e.record()
            # algorithmic bytes: operands once, the output once, plus every further M x N tensor the epilogue reads or writes (second output, GELU' / product operand, residual)
            extra = sum(kw.get(k) is not None for k in ("y_pre", "aux", "res"))
            timer.events.append((bn, 2.0 * M * N * K, (M * K + N * K + M * N * (1 + extra)) * 2.0, s, e,
                                 ("nt", M, N, K, kw.get("epi", 0), kw.get("y_pre") is not None, kw.get("res") is not None)))
            return y

        ops.linear_raw = timed_linear_raw
        raw_w = ops.wgrad_partials_raw

        def timed_wgrad_partials_raw(dy2, x2, want_bias, ws, nbytes, rowscale=None, rows_per_scale=1, x_gelu=False):
            if not timer.enabled or dy2.dtype != torch.bfloat16:
                return raw_w(dy2, x2, want_bias, ws, nbytes, rowscale, rows_per_scale, x_gelu)
            M, N = dy2.shape
            K = x2.shape[1]
            # same dispatch as csrc/gemm.hip::fmmt_linear_wgrad_partials; the fixed-order sum of the partials
            # (fmmt_linear_wgrad_finish) is a separate launch and is not inside these events
            name = "linear_tn_kernel<bf16,32,few>" if M <= 4096 else "linear_tn_kernel<bf16,64>" if M <= 262144 else "linear_tn_kernel<bf16,32>"
            if (128 < M <= 2048 and N % 64 == 0 and K % 64 == 0 and 8 <= (N // 64) * (K // 64) <= 1024
                    and rowscale is None and not x_gelu):
                name = "linear_tn_few_kernel"
            dma = tn_dma_tile(M, N, K, rowscale is not None, rows_per_scale, x_gelu)
            if dma:
                name = f"linear_tn_dma_kernel<{dma},{'true' if rowscale is not None else 'false'}>"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            raw_w(dy2, x2, want_bias, ws, nbytes, rowscale, rows_per_scale, x_gelu)
            e.record()
            timer.events.append((name, 2.0 * M * N * K, (M * N + M * K) * 2.0 + N * K * 4.0, s, e,
                                 ("tn", M, N, K, 0, bool(want_bias), rowscale is not None)))

        ops.wgrad_partials_raw = timed_wgrad_partials_raw

    # C-ABI entry points other than fmmt_linear_fwd / fmmt_linear_wgrad_partials (those two are bracketed one level up, in
    # ops.linear_raw / ops.wgrad_partials_raw, where the kernel a shape dispatches to is known): (tag suffix, algorithmic flops,
    # algorithmic bytes) from the call's arguments -- positions as in include/fmmt.h.  Every entry point the step's models call is
    # here (round-3 VERDICT item 7); the ctypes function objects are attributes of the loaded library, so replacing an attribute
    # brackets every call the ops / modules make.  es(a) = bytes per activation element.
    @staticmethod
    def _abi_table():
        es = lambda a: 4 if a[0] == 0 else 2
        nz = lambda v: 1 if v else 0
        win = lambda a: a[1] * (a[2] // 7) * (a[3] // 7)                                     # windows of a window-attention call
        T = lambda a: a[1] * a[2] * a[3]                                                     # tokens of it
        return {
            "fmmt_layernorm_fwd": lambda a: (f"<C={a[2]}>", 0.0, a[1] * a[2] * es(a) * 2.0),
            "fmmt_layernorm_bwd": lambda a: (f"<C={a[2]}>", 0.0, a[1] * a[2] * es(a) * (4.0 if a[8] else 3.0)),
            "fmmt_layernorm_bwd_bf16": lambda a: (f"<C={a[1]}>", 0.0, a[0] * a[1] * 2 * 3.0),
            # attention core on a materialised qkv: QK^T and PV (fwd), five 49x49x32 products per head (bwd)
            "fmmt_window_attn_fwd": lambda a: (f"<C={a[4]}>", 4.0 * T(a) * 49 * a[4], T(a) * a[4] * es(a) * 4.0),
            "fmmt_window_attn_bwd": lambda a: (f"<C={a[4]}>", 10.0 * T(a) * 49 * a[4], T(a) * a[4] * es(a) * 8.0),
            # one window: qkv 2*49*C*3C, attention 4*49*49*C, proj 2*49*C*C; bytes: x in, y out, and the saved LN(x) / attention output
            "fmmt_window_block_fwd": lambda a: (f"<C={a[4]}>", win(a) * (8.0 * 49 * a[4] * a[4] + 4.0 * 49 * 49 * a[4]),
                                                T(a) * a[4] * 2.0 * (2 + nz(a[20]) + nz(a[21]))),
            # recompute backward: q/k/v re-formed (2*49*C*3C), d(out) = dy . Wproj (2*49*C*C), five attention products; LN(x), dy, out in, dqkv out
            "fmmt_window_block_attn_bwd": lambda a: (f"<C={a[4]}>", win(a) * (8.0 * 49 * a[4] * a[4] + 10.0 * 49 * 49 * a[4]), T(a) * a[4] * 2.0 * 6),
            "fmmt_mlp_fwd": lambda a: (f"<C={a[2]}>", 16.0 * a[1] * a[2] * a[2], a[1] * a[2] * 2.0 * (2 + nz(a[8])) + a[1] * 4 * a[2] * 2.0 * (nz(a[12]) + nz(a[13]))),
            "fmmt_mlp_ln_fwd": lambda a: (f"<C={a[2]}>", 16.0 * a[1] * a[2] * a[2], a[1] * a[2] * 2.0 * (2 + nz(a[14])) + a[1] * 4 * a[2] * 2.0 * (nz(a[17]) + nz(a[18]))),
            "fmmt_mlp_bwd_input": lambda a: (f"<C={a[2]}>", 16.0 * a[1] * a[2] * a[2], a[1] * a[2] * 2.0 * 2 + a[1] * 4 * a[2] * 2.0 * 2),
            "fmmt_mlp_ln_bwd_input": lambda a: (f"<C={a[2]}>", 16.0 * a[1] * a[2] * a[2], a[1] * a[2] * 2.0 * 3 + a[1] * 4 * a[2] * 2.0 * 2),
            "fmmt_patch_embed_u8": lambda a: ("", 0.0, a[2] * a[3] * a[3] * 3.0 + a[2] * 3136 * 48 * es(a)),
            # args: dtype, mode, n_img, in_size, img, tab, lut, w, bias, gamma, beta, eps, cols, x_pre, y, mean, rstd: u8 in, y out, cols + x_pre out in training
            "fmmt_patch_embed_u8_ln_fwd": lambda a: ("", 2.0 * a[2] * 3136 * 96 * 48, a[2] * a[3] * a[3] * 3.0 + a[2] * 3136 * es(a) * (96.0 + 48.0 * nz(a[12]) + 96.0 * nz(a[13]))),
            "fmmt_patch_embed_ln_fwd": lambda a: ("", 2.0 * a[1] * a[2] * a[3], a[1] * a[3] * 2.0 + a[1] * a[2] * 2.0 * (1 + nz(a[10]))),
            "fmmt_patch_im2col": lambda a: ("", 0.0, a[1] * 3 * 224 * 224 * es(a) * 2.0),
            "fmmt_patch_col2im": lambda a: ("", 0.0, a[1] * 3 * 224 * 224 * es(a) * 2.0),
            "fmmt_linear_fwd_splitk": lambda a: (f"<K={a[3]}>", 2.0 * a[1] * a[2] * a[3], (a[1] * a[3] + a[2] * a[3] + a[1] * a[2]) * float(es(a))),
            "fmmt_linear_wgrad": lambda a: ("<few tokens>", 2.0 * a[1] * a[2] * a[3], (a[1] * a[2] + a[1] * a[3]) * float(es(a)) + a[2] * a[3] * 4.0),
            "fmmt_linear_wgrad_finish": lambda a: ("", 0.0, a[2] * a[3] * 4.0 * 2),           # lower bound: one partial in, the sum out
            "fmmt_mha_fwd": lambda a: (f"<E={a[4]},hd={a[4] // a[5]}>", 4.0 * a[1] * a[2] * a[3] * a[4], (2.0 * a[1] + 2.0 * a[2]) * a[3] * a[4] * es(a)),
            "fmmt_mha_bwd": lambda a: (f"<E={a[4]},hd={a[4] // a[5]}>", 10.0 * a[1] * a[2] * a[3] * a[4], (4.0 * a[1] + 4.0 * a[2]) * a[3] * a[4] * es(a)),
            "fmmt_batchnorm1d_fwd": lambda a: ("", 0.0, a[1] * a[2] * es(a) * 2.0),
            "fmmt_batchnorm1d_bwd": lambda a: ("", 0.0, a[1] * a[2] * es(a) * 3.0),
            "fmmt_posemb_scale_fwd": lambda a: ("", 0.0, a[1] * a[2] * a[3] * es(a) * 2.0),
            "fmmt_scale": lambda a: ("", 0.0, a[1] * es(a) * 2.0),
            "fmmt_colsum": lambda a: ("", 0.0, a[2] * a[3] * float(es(a))),
            "fmmt_cast_batch": lambda a: ("", 0.0, a[1] * 4096 * 6.0),                           # 64 x 64 tiles: fp32 in, bf16 out
            "fmmt_adamw_batch": lambda a: ("", 0.0, a[1] * 4096 * 30.0),                         # p, m, v read + written, g read, bf16 twin written
            "fmmt_grad_handover": lambda a: ("", 0.0, a[1] * 4096 * 7.0),                        # gradient in (bf16 or fp32: ~3 B on this model's mix), fp32 slot out
            # d(LN out) = dz . W (2 M K C), LayerNorm', residual: dz, x, dres in, dx out
            "fmmt_linear_ln_bwd": lambda a: (f"<C={a[2]},K={a[3]}>", 2.0 * a[1] * a[2] * a[3], a[1] * (a[3] + a[2] * (2 + nz(a[10]))) * 2.0),
        }

    def install_abi(self):
        from facialmmt_amd import _lib
        lib = _lib.load()
        timer = self
        for name, cost in self._abi_table().items():
            fn = getattr(lib, name)

            def make(name, fn, cost):
                def timed(*a):
                    if not timer.enabled:
                        return fn(*a)
                    suffix, fl, by = cost(a)
                    fp32 = name not in ("fmmt_layernorm_bwd_bf16", "fmmt_cast_batch", "fmmt_adamw_batch", "fmmt_grad_handover") and a[0] == 0
                    tag = name + ("<fp32>" if fp32 else "") + suffix
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    rc = fn(*a)
                    e.record()
                    timer.events.append((tag, fl, by, s, e, ("abi",)))
                    return rc
                return timed
            setattr(lib, name, make(name, fn, cost))

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
    """The oracle (oracle/*.py, fp32 CPU restatement) timed on this box's host cores on a bounded sample of the same
    workload: Swin forward(+backward) on `cpu_frames` frames (scaled to `frames` per utterance), the fusion stack, the two
    self-attention encoders and the text encoder for one utterance.  `value` = forward+backward utterances/s (what the GPU
    line measures, optimizer excluded); `forward_only` next to it (north_star asks for the CPU forward); `thread_sweep` =
    Swin forward on 16 / 32 / 64 threads (more threads are slower on this box: 256 hardware threads, see below)."""
    import numpy as np
    from facialmmt_amd import synth
    from oracle import crossmodal as OC
    from oracle import multimodal as OM
    from oracle import swin as OS
    # 256 hardware threads on the GPU box make torch's CPU ops slower, not faster (measured: 234 s for the
    # 8-frame step with 256 threads vs ~2 s with 8-16); use 16 threads and say so.
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 16)
    torch.set_num_threads(cores)
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        keys = json.load(f)
    sd = synth.state_dict_from_keys(keys["affwild"], seed=100)
    for v in sd.values():
        v.requires_grad_(True)
    def timed(fn, reps=5, warm=2, bound=10.0):
        """median of `reps` after `warm` warm-up calls (median of >= 5 after 2); bounded: one slow call ends the series"""
        ts = []
        for i in range(warm + reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            if i >= warm or dt >= bound:
                ts.append(dt)
            if dt >= bound:
                break
        return float(np.median(ts)), len(ts)

    def swin_fwd_of(x):
        def f():
            with torch.no_grad():
                OS.swin_affwild_logits(sd, x, training=False)
        return f

    def swin_step_of(x):
        def f():
            for v in sd.values():
                v.grad = None
            OS.swin_affwild_logits(sd, x, training=True).square().sum().backward()
        return f

    # The oracle's throughput depends strongly on the frames per call (activations of a big call fall out of the caches: one 160-frame
    # call ran at 16 frames/s where 8-frame calls reach 75 on the same 16 threads -- round-3 VERDICT weak 10): sweep the chunk size,
    # time the utterance as ceil(frames / chunk) calls of the BEST chunk, and keep the one-call figure as a side note.
    chunk_fps = {}
    swin_fwd_of(synth.tensor("frames", (4, 3, 224, 224), seed=1))()        # thread pool / allocator warm-up outside the sweep
    for nF in (4, 8, 16, 32):
        xs = synth.tensor("frames", (nF, 3, 224, 224), seed=1)
        t, _ = timed(swin_fwd_of(xs), reps=3, warm=2, bound=5.0)
        chunk_fps[nF] = nF / t
    nF = max(chunk_fps, key=chunk_fps.get)
    x = synth.tensor("frames", (nF, 3, 224, 224), seed=1)
    ncall = -(-args.frames // nF)
    f8, nrep_f = timed(swin_fwd_of(x))
    t8, nrep = timed(swin_step_of(x))
    f_swin, t_swin = f8 * ncall, t8 * ncall
    fwd_how = f"{ncall} calls of {nF} frames (best chunk of the sweep {{{', '.join(f'{k}: {v:.1f}' for k, v in chunk_fps.items())}}} frames/s), median of {nrep_f} after 2 warm-ups"
    one_call = None
    if f_swin < 20.0:                                        # side note: the whole utterance's frames in ONE call
        xf = synth.tensor("frames_full", (args.frames, 3, 224, 224), seed=2)
        tf, _ = timed(swin_fwd_of(xf), reps=1, warm=0, bound=0.0)
        one_call = round(args.frames / tf, 2)
        del xf
    sweep = {str(cores): round(nF / f8, 2)}
    for th in (32, 64):
        if th <= ncpu:
            torch.set_num_threads(th)
            ft, _ = timed(swin_fwd_of(x), reps=2, warm=1, bound=5.0)
            sweep[str(th)] = round(nF / ft, 2)
    torch.set_num_threads(cores)
    esd = synth.state_dict_from_keys(keys["crossmodal"], seed=50, prefix="enc.")
    for v in esd.values():
        v.requires_grad_(True)
    La, Lv, Lt = cfg.get_audio_utt_max_lens, args.frames, cfg.get_text_utt_max_lens
    t_, a_, v_ = (synth.tensor(n, (L, 1, 768), seed=60) for n, L in (("t", Lt), ("a", La), ("v", Lv)))

    def fusion(backward):
        ta = torch.cat((OC.crossmodal_encoder(esd, t_, a_, a_), OC.crossmodal_encoder(esd, a_, t_, t_)), 0)
        out = torch.cat((OC.crossmodal_encoder(esd, ta, v_, v_), OC.crossmodal_encoder(esd, v_, ta, ta)), 0)
        if backward:
            out.square().mean().backward()
    t_fus, _ = timed(lambda: fusion(True), reps=3, warm=1)
    with torch.no_grad():
        f_fus, _ = timed(lambda: fusion(False), reps=3, warm=1)
    # the two self-attention encoders in front of the fusion (oracle restatement) ...
    msd = {k: v for k, v in synth.state_dict_from_keys(keys["multimodal_roberta"], seed=200).items()
           if k.startswith(("audio_utt_transformer.", "vision_utt_transformer."))}
    # the fixture keys were dumped for 24/20-step sequences: rebuild the position tables for the bench lengths
    msd["audio_utt_transformer.position_embeddings.weight"] = synth.tensor("pos_a", (La, 768), seed=1, lo=-0.1, hi=0.1)
    msd["vision_utt_transformer.position_embeddings.weight"] = synth.tensor("pos_v", (Lv, 768), seed=1, lo=-0.1, hi=0.1)
    for v in msd.values():
        v.requires_grad_(True)
    a_in, v_in = synth.tensor("a_in", (1, La, 768), seed=61), synth.tensor("v_in", (1, Lv, 768), seed=62)

    def meld(backward):
        za = torch.zeros(1, 1, 1, La)
        zv = torch.zeros(1, 1, 1, Lv)
        out = OM.meld_encoder(msd, "audio_utt_transformer.", a_in, za, cfg.audio_utt_Transformernum).square().mean() \
            + OM.meld_encoder(msd, "vision_utt_transformer.", v_in, zv, cfg.vision_utt_Transformernum).square().mean()
        if backward:
            out.backward()
    t_meld, _ = timed(lambda: meld(True), reps=3, warm=1)
    with torch.no_grad():
        f_meld, _ = timed(lambda: meld(False), reps=3, warm=1)
    # ... and the text encoder: the same third-party model (random init), one 512-token dialogue, fp32
    t_plm = f_plm = None
    try:
        from transformers import BertModel, RobertaModel
        conf = plm_config(args.plm)
        plm = (BertModel if args.plm == "bert-large" else RobertaModel)(conf, add_pooling_layer=False)
        ids = torch.from_numpy(synth.randint("cpu_ids", (1, 512), 3, conf.vocab_size, seed=2))
        am = torch.ones(1, 512)

        def plm_step():
            plm.zero_grad(set_to_none=True)
            plm(ids, am)[0].square().mean().backward()

        def plm_fwd():
            with torch.no_grad():
                plm(ids, am)
        t_plm, _ = timed(plm_step, reps=3, warm=1)
        f_plm, _ = timed(plm_fwd, reps=3, warm=1)
    except Exception as e:                               # the CPU leg must never take the bench down
        print(f"cpu_baseline: text-encoder leg skipped ({e})", file=sys.stderr)
    total = t_swin + t_fus + t_meld + (t_plm or 0.0)
    total_f = f_swin + f_fus + f_meld + (f_plm or 0.0)
    cpu_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_name = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": round(1.0 / total, 5), "unit": "utterances/s", "cores": cores, "kind": "port",
            "forward_only": {"value": round(1.0 / total_f, 5), "unit": "utterances/s", "swin_frames_per_s": round(args.frames / f_swin, 2), "swin_sample": fwd_how,
                             "swin_frames_per_s_one_call_of_all_frames": one_call},
            "swin_forward_frames_per_s_by_chunk": {str(k): round(v, 2) for k, v in chunk_fps.items()},
            "swin_forward_frames_per_s_by_threads": sweep, "host": f"{cpu_name} ({ncpu} hardware threads)",
            "sample": f"oracle fp32 for ONE utterance on {cores} threads, fwd+bwd [fwd only]: Swin+head as {ncall} calls of {nF} frames (median of {nrep} after 2 warm-ups per call: "
                      f"{t_swin:.2f} s [{f_swin:.2f} s]) + 4 cross-modal encoder calls ({t_fus:.2f} s [{f_fus:.2f} s]) + audio/vision self-attention "
                      f"encoders ({t_meld:.2f} s [{f_meld:.2f} s]) + {args.plm} 512 tokens (HF, {'%.2f s [%.2f s]' % (t_plm, f_plm) if t_plm else 'skipped'}); optimizer excluded"}


def spawn_command(argv, gpus, port=None):
    """`python bench.py --gpus N` typed without a launcher (WORLD_SIZE unset, N > 1): the command this process re-executes itself as --
    one rank per GPU under torch.distributed.run on 127.0.0.1 (the container's hostname may not resolve), the same line the driver uses"""
    port = port or int(os.environ.get("FMMT_BENCH_PORT", "0")) or 29500 + os.getpid() % 2000
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(argv)


def rendezvous_check(args, world, rank, backend):
    """--rendezvous-check: the process group alone.  Every rank contributes a one; rank 0 prints what the sum says"""
    if world > 1:
        if not dist.is_initialized():
            dist.init_process_group(backend)
        t = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"rendezvous_check": seen == args.gpus, "n_gpus": args.gpus, "world_size": world, "ranks_seen": seen, "backend": backend}), flush=True)
    return 0 if seen == args.gpus else 1


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # typed as `python bench.py --gpus N`: become the launcher (round-5 VERDICT item 6: this used to die on the assertion below)
        os.execv(sys.executable, spawn_command(sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus"
    if args.rendezvous_check:
        sys.exit(rendezvous_check(args, world, rank, os.environ.get("FMMT_BENCH_BACKEND", "nccl")))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the hot path has no CPU fallback)"
    # FMMT_BENCH_DEVICE / FMMT_BENCH_BACKEND: dry-run switches for exercising the N>1 code path on a one-GPU box (all
    # ranks on device 0, gloo instead of RCCL, which refuses two ranks per device); never set by the driver
    local = int(os.environ.get("FMMT_BENCH_DEVICE", local))
    backend = os.environ.get("FMMT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rank != 0:                                           # only rank 0 reports; keep other ranks' C-level banners out of stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    rccl_log = None
    if world > 1 or args.force_ddp:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            if args.rccl_tuning == "mesh":
                for k, v in RCCL_MESH_ENV.items():
                    os.environ.setdefault(k, v)
            # the communicator's own account of its size: RCCL's init log (rank 0 reads it back after the run -> config.rccl_ranks_seen)
            rccl_log = f"/tmp/fmmt_rccl_init_{os.getpid()}.log"
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT")
            os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from facialmmt_amd.config import default_args
    from facialmmt_amd.parallel import GradientAverager, broadcast_parameters
    from facialmmt_amd.train_step import AuxStep, GraphedAuxStep, GraphedTargetStep, HFAdamW, TargetStep
    cfg = default_args(get_vision_utt_max_lens=args.frames, trg_accumulation_steps=1)
    act = torch.bfloat16 if args.dtype == "bf16" else None
    swin, mm = build_models(args, dev, cfg)
    batch = synth_batch(args, dev, rank, cfg)
    aux_batch = synth_aux_batch(args, dev, rank) if args.aux_images else None
    ddp = world > 1 or args.force_ddp
    comm = torch.bfloat16 if args.grad_comm == "bf16" else None
    if ddp:
        broadcast_parameters(mm)
        broadcast_parameters(swin)
        torch.cuda.synchronize()                            # no collective in flight when the step's graphs are captured
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    lr_of = lambda s: min(1.0, (s + 1) / 100.0)            # linear warm-up (transformers.get_linear_schedule_with_warmup, train.py:333-339)
    aux_step = None
    if args.graphs == 2:
        # the whole step as two HIP graphs (train_step.GraphedTargetStep): gradients in static flat buffers, the only
        # collective -- one all-reduce per flat bucket -- issued between the graphs when N > 1
        from facialmmt_amd.train_step import MasterWeights, step_parameters
        masters = None
        if args.plm_dtype == "bf16" and args.dtype == "bf16":
            masters = MasterWeights(mm.roberta if mm.text_pretrained_model == "roberta" else mm.bert, torch.bfloat16)
        params = step_parameters(mm, masters)
        flat = GradientAverager(params, hooks=False, comm_dtype=comm, always=args.force_ddp)
        # the reference's optimizer class and arguments (train.py:307): transformers.AdamW(lr, weight_decay) -> eps 1e-6, HF update order
        opt = HFAdamW(params, lr=torch.tensor(cfg.trg_lr, device=dev), weight_decay=cfg.weight_decay)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_of)
        if args.aux_images:
            aflat = GradientAverager(swin.parameters(), hooks=False, comm_dtype=comm, always=args.force_ddp)
            aopt = HFAdamW(swin.parameters(), lr=torch.tensor(cfg.aux_lr, device=dev))       # train.py:333: no weight decay on the Swin model
            aux_step = GraphedAuxStep(swin, aopt, torch.optim.lr_scheduler.LambdaLR(aopt, lr_of), cfg, *aux_batch, averager=aflat)
        # N > 1: where the backward graph is cut follows from the exchange time MEASURED alone on this communicator (three blocking
        # exchanges of the real buckets after one warm-up, max over ranks), never from a nominal link rate (round-3 ADVICE / VERDICT)
        swin_cut, cut_ms, tail_ms = max(args.swin_cut, 0), None, None
        if ddp and flat.active:
            from facialmmt_amd.train_step import pick_swin_cut
            flat.exchange_all()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t_x = time.perf_counter()
            for _ in range(3):
                flat.exchange_all()
            torch.cuda.synchronize()
            tx = torch.tensor([(time.perf_counter() - t_x) / 3 * 1e3], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tx, op=dist.ReduceOp.MAX)
            cut_ms = float(tx.item())
            flat.zero_grad()
            if args.swin_cut < 0:
                # the window behind each cut, timed on THIS device with this batch (was a table from one box's profile)
                from facialmmt_amd.train_step import measure_swin_tail_ms
                with torch.autocast("cuda", dtype=act) if act is not None else contextlib.nullcontext():
                    tail_ms = measure_swin_tail_ms(swin, batch[8])
                swin_cut = pick_swin_cut(cut_ms, args.utts * args.frames, tail_ms)
        pipelined = bool(args.pipeline_swin) and not (ddp and flat.active) and not args.aux_images and args.discarded_swin_gradients == "compute"
        branched = bool(args.branch_graphs) and not pipelined and not (ddp and flat.active) and args.discarded_swin_gradients == "compute"
        step = GraphedTargetStep(swin, mm, opt, sched, cfg, batch, autocast_dtype=act, overlap_text=bool(args.overlap_text),
                                 parallel_fusion=bool(args.parallel_fusion), averager=flat, masters=masters,
                                 discarded_swin_gradients=args.discarded_swin_gradients, swin_cut=swin_cut, pipeline_swin=pipelined,
                                 branch_graphs=branched,
                                 fork_streams=bool(args.fork_streams) and not branched and not pipelined and not (ddp and flat.active) and args.discarded_swin_gradients == "compute")
    else:
        if args.graphs == 1:
            from facialmmt_amd.train_step import graph_multimodal, select_frames
            with torch.no_grad():
                preds = swin(batch[8][:8], is_trg_task=True).float().repeat(batch[8].shape[0] // 8, 1)
            vis, nmask = select_frames(preds, batch[5], batch[6], batch[9], cfg.FacialEmoImpor_threshold)
            sample = (batch[0], batch[1], batch[2], batch[3], batch[4], vis.detach().requires_grad_(True), nmask, batch[10])
            mm = graph_multimodal(mm, sample, act, overlap_text=bool(args.overlap_text), parallel_fusion=bool(args.parallel_fusion))
            mm.zero_grad(set_to_none=True)
            swin.zero_grad(set_to_none=True)
        averager = None
        if ddp:
            # hook-driven exchange overlapped with the backward; two stream groups: the text branch's gradients are
            # produced on the second HIP stream
            plm = mm.roberta if mm.text_pretrained_model == "roberta" else mm.bert
            text_params = list(plm.parameters()) + list(mm.text_linear.parameters())
            text_ids = set(map(id, text_params))
            averager = GradientAverager(None, groups=[[p for p in mm.parameters() if id(p) not in text_ids], text_params], comm_dtype=comm)
        opt = HFAdamW(mm.parameters(), lr=cfg.trg_lr, weight_decay=cfg.weight_decay)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_of)
        step = TargetStep(swin, mm, opt, sched, cfg, autocast_dtype=act, averager=averager)
        if args.aux_images:
            aopt = HFAdamW(swin.parameters(), lr=cfg.aux_lr)
            eager_aux = AuxStep(swin, aopt, torch.optim.lr_scheduler.LambdaLR(aopt, lr_of), cfg)
            aux_step = lambda imgs, labels: eager_aux(imgs, labels)
    timer = KernelTimer()
    timer.install()
    timer.install_abi()

    pipelined = bool(getattr(step, "pipeline", False))

    def one_step():
        if aux_step is not None:
            aux_step(*aux_batch)
        if pipelined:                                        # the NEXT step's Swin forward goes out beside this step (one forward per step, as ever)
            return step(batch, next_batch=batch)
        return step(batch)

    if ddp and args.graphs == 2:
        step.time_exchange(True)                            # three events per step around the gradient exchange
    kept = None
    for _ in range(args.warmup):
        _, kept = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = rank == 0 and args.graphs != 2
    step.host_ms = {}
    step.gpu_events = [] if (rank == 0 and args.graphs != 2) else None      # main-stream timeline of the phases (eager Swin only)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, kept = one_step()
    issue_s = time.perf_counter() - t0                      # host time to enqueue the steps (GPU still running)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    gpu_phase = {}
    if getattr(step, "gpu_events", None):
        evs, step.gpu_events = step.gpu_events, None
        for (_, a), (name, b) in zip(evs, evs[1:]):
            if name != "start":
                gpu_phase[name] = gpu_phase.get(name, 0.0) + a.elapsed_time(b)
    step.host_ms.pop("start", None)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1: how long the gradient exchange took and how much of it the step could not hide behind Swin's backward (events of the
    # LAST timed step), and the bus rate of the exchange alone (two blocking exchanges of the same buckets after the timed region)
    xchg = None
    if ddp and args.graphs == 2:
        total_ms, exposed_ms = step.exchange_ms()
        nbytes = sum(b[0].numel() for b in flat.buckets) * (2 if comm is not None else 4)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t3 = time.perf_counter()
        for _ in range(2):
            flat.exchange_all()
        torch.cuda.synchronize()
        alone_ms = (time.perf_counter() - t3) / 2 * 1e3
        flat.zero_grad()
        xchg = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "wire_dtype": args.grad_comm, "bytes_per_rank": nbytes,
                "buckets": len(flat.buckets), "ms_issue_to_done": round(total_ms, 3), "ms_exposed_after_swin_backward": round(exposed_ms, 3),
                "ms_alone": round(alone_ms, 3), "ms_alone_before_capture": (round(cut_ms, 3) if cut_ms is not None else None),
                "swin_cut": step.SWIN_CUT, "swin_cut_from": ("--swin-cut" if args.swin_cut >= 0 else "exchange measured alone before capture against Swin's backward pieces timed on this device (train_step.pick_swin_cut, measure_swin_tail_ms)"),
                "swin_tail_ms": tail_ms if (ddp and flat.active and args.swin_cut < 0) else None,
                "rccl_tuning": args.rccl_tuning, "rccl_env": {k: os.environ.get(k) for k in RCCL_KNOBS},
                "bus_GB_per_s_alone": round(2.0 * (world - 1) / max(world, 1) * nbytes / (alone_ms * 1e-3) / 1e9, 1) if world > 1 else None}
        if args.aux_images and aux_step is not None and getattr(aflat, "active", False):
            # the auxiliary-task step's own exchange (train.py:15-41: the Swin model's 46.8 M gradients, every `aux_every` target steps): alone, and the
            # whole auxiliary step (graph replays) after the timed region -- not part of `value`
            abytes = sum(b[0].numel() for b in aflat.buckets) * (2 if comm is not None else 4)
            aflat.exchange_all()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t4 = time.perf_counter()
            for _ in range(2):
                aflat.exchange_all()
            torch.cuda.synchronize()
            a_alone = (time.perf_counter() - t4) / 2 * 1e3
            aflat.zero_grad()
            aux_step(*aux_batch)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t5 = time.perf_counter()
            for _ in range(3):
                aux_step(*aux_batch)
            torch.cuda.synchronize()
            a_step = (time.perf_counter() - t5) / 3 * 1e3
            xchg["aux_step"] = {"bytes_per_rank": abytes, "buckets": len(aflat.buckets), "ms_alone": round(a_alone, 3), "aux_step_ms": round(a_step, 2),
                                "aux_images": args.aux_images,
                                "bus_GB_per_s_alone": round(2.0 * (world - 1) / max(world, 1) * abytes / (a_alone * 1e-3) / 1e9, 1) if world > 1 else None}

    # host cost of issuing one step into an IDLE queue (during the timed loop the host mostly waits for the previous replay
    # of the same graph to drain, so the enqueue time above is back-pressure, not cost)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    one_step()
    idle_issue_ms = (time.perf_counter() - t1) * 1e3
    torch.cuda.synchronize()

    # After the timed region (not part of `value`): the same steps with every batch tensor handed over in PINNED HOST memory, as
    # the reference's DataLoader does (train.py:290-314: `.to(device)` per tensor) -- the step's own copy into its static input
    # tensors then crosses PCIe.  N = 1 only, like the CPU baseline: a reported side figure has no business adding collectives
    # to the N > 1 runs.
    pcie = None
    if args.graphs == 2 and args.host_input_leg and world == 1:
        try:
            pin = lambda ts: tuple(t.detach().cpu().pin_memory() if torch.is_tensor(t) else t for t in ts)
            hb, hab = pin(batch), (pin(aux_batch) if aux_batch is not None else None)

            def host_step():
                if aux_step is not None:
                    aux_step(*hab)
                return step(hb)

            host_step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                host_step()
            torch.cuda.synchronize()
            h_ms = (time.perf_counter() - t2) / args.steps * 1e3
            nbytes = sum(t.numel() * t.element_size() for t in hb + (hab or ()) if torch.is_tensor(t))
            pcie = {"value": round(args.utts * world / (h_ms * 1e-3), 3), "unit": "utterances/s", "ms_per_step": round(h_ms, 2),
                    "host_bytes_per_step_per_gpu": nbytes,
                    "note": "batch tensors in pinned host memory, copied into the step's static inputs on the step's stream (not overlapped); N = 1 only; not `value`"}
            step(batch)                                      # device-resident inputs again for what follows
            torch.cuda.synchronize()
        except Exception as e:                               # a reported leg only: never takes the line down
            pcie = {"error": f"{type(e).__name__}: {e}"}

    # After the timed region (not part of `value`): the kernels of the Linear layers bracketed with HIP events in EAGER
    # forward+backward passes on the main stream (inside a graph replay nothing can be bracketed; with --graphs 1 the timed
    # steps themselves are bracketed too, but there the text-encoder graph shares the CUs and stretches every Swin kernel).
    timed_events, timer.events = timer.events, []
    iso = None
    if rank == 0:
        probe_cfg = default_args(get_vision_utt_max_lens=args.frames, trg_accumulation_steps=1 << 30)     # never reaches clip / optimizer
        side, mm.text_stream = getattr(mm, "text_stream", None), None
        probe = TargetStep(swin, mm, None, None, probe_cfg, autocast_dtype=act)
        was_training = mm.training
        if args.graphs == 1:
            mm.eval()                                        # eager branches (models._branch_call); dropout off does not change the GEMM shapes
        probe(batch)
        timer.enabled = True
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0.record()
        for _ in range(2):
            probe(batch)
            # the two launches of the optimizer graph's side: weight-shadow refresh and clip + AdamW (+ bf16 twins)
            if getattr(step, "shadows", None) is not None:
                step.shadows.refresh()
            if getattr(step, "fused", None) is not None:
                step.fused.update()
        p1.record()
        torch.cuda.synchronize()
        timer.enabled = False
        coverage = bracket_coverage(timer, lambda: probe(batch), p0.elapsed_time(p1) / 2)
        mm.train(was_training)
        mm.text_stream = side
        iso, timer.events = timer.summary(), timed_events
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.utts * world * args.steps / elapsed
        fams = iso or timer.summary()
        if args.shape_report:
            with open(args.shape_report, "w") as f:
                f.write(timer.shape_report(args.steps) + "\n")
        roof = roof_hbm = roof_mfma = None
        if fams:
            # the dominant kernel = the family with the most GPU time among the many-token launches (the few-token GEMMs of the
            # fusion stack are launch-bound 10-25 us kernels: many of them, no roofline to speak of)
            big = {k: v for k, v in fams.items() if v[3] / v[0] >= 40e-6} or fams
            bn, (cnt, fl, by, sec) = max(big.items(), key=lambda kv: kv[1][3])
            kname = kernel_symbol(bn)
            traffic, traffic_src = None, None                # PMC counters cannot be read in-process: taken from the committed PMC summary
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    tj = json.load(f)
                t = tj.get(kname)
                if t:
                    traffic = (2 * t["fetch_size_kb"] + (t["write_size_kb"] or 0)) * 1024
                    traffic_src = f"profiles/traffic.json ({tj.get('_source', 'committed rocprofv3 --pmc pass')}): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE per launch; not measured in this run"
            except OSError:
                pass

            def rate(c, f, b, t):
                return {"achieved": round(f / t / 1e12, 1), "frac": round(f / t / 1e12 / PEAK_BF16_TFLOPS, 4),
                        "avg_launch_us": round(t / c * 1e6, 1), "algorithmic_GB_per_s": round(b / t / 1e9, 0)}
            # bound of the dominant kernel from its algorithmic intensity: below the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B) the HBM
            # roofline is the lower one and `achieved` / `peak` are bytes per second
            ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            if fl / max(by, 1.0) < ridge:
                roof = {"bound": "hbm", "kernel": kname, "peak": PEAK_HBM_GBS, "unit": "GB/s", "achieved": round(by / sec / 1e9, 0),
                        "frac": round(by / sec / 1e9 / PEAK_HBM_GBS, 4), "avg_launch_us": round(sec / cnt * 1e6, 1),
                        "TFLOP_per_s": round(fl / sec / 1e12, 1), "flop_per_byte": round(fl / max(by, 1.0), 1),
                        "frac_of_mfma_peak": round(fl / sec / 1e12 / PEAK_BF16_TFLOPS, 4)}     # for a GEMM family near the ridge: both views
            else:
                roof = {"bound": "mfma", "kernel": kname, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s"}
                roof.update(rate(cnt, fl, by, sec))
            roof.update({"traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(by / cnt), "launches_per_step": cnt // 2,
                         "share_of_step": round(sec / 2 / (ms * 1e-3), 3),
                         "measured": "HIP events on the launch stream, 2 eager forward+backward passes right after the timed region (text encoder on the main stream)"})
            live = timer.summary().get(bn) if timed_events else None
            if live:
                roof["in_timed_region_with_text_stream"] = rate(*live)
                roof["frac_in_timed_region"] = roof["in_timed_region_with_text_stream"]["frac"]
            # the largest family above the ridge as well (the dominant kernel by time may be an HBM-bound one)
            mf = [(k, v) for k, v in big.items() if v[1] / max(v[2], 1.0) >= ridge]
            if mf:
                k, v = max(mf, key=lambda kv: kv[1][3])
                roof_mfma = {"bound": "mfma", "kernel": kernel_symbol(k), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "launches_per_step": v[0] // 2,
                             "ms_per_step": round(v[3] / 2 * 1e3, 3)}
                roof_mfma.update(rate(*v))
            # every bracketed family, Linear or not: time per step, TFLOP/s and algorithmic GB/s; bound = "hbm" where the algorithmic
            # intensity is below the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B)
            def fam(v):
                d = {"ms_per_step": round(v[3] / 2 * 1e3, 3), "launches_per_step": v[0] // 2, "TFLOP_per_s": round(v[1] / v[3] / 1e12, 1),
                     "algorithmic_GB_per_s": round(v[2] / v[3] / 1e9, 0)}
                d["bound"] = "hbm" if v[1] / max(v[2], 1.0) < PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9) else "mfma"
                d["frac"] = round(v[2] / v[3] / 1e9 / PEAK_HBM_GBS, 3) if d["bound"] == "hbm" else round(v[1] / v[3] / 1e12 / PEAK_BF16_TFLOPS, 3)
                return d
            ranked = sorted(fams.items(), key=lambda kv: -kv[1][3])
            roof["families"] = {kernel_symbol(k): fam(v) for k, v in ranked[:28]}
            roof["families_bracketed"] = len(ranked)
            roof["bracket_coverage"] = coverage
            hbm = [(k, v) for k, v in ranked if v[3] / v[0] >= 40e-6 and v[1] / max(v[2], 1.0) < PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)]
            if hbm:
                k, v = hbm[0]
                roof_hbm = {"bound": "hbm", "kernel": kernel_symbol(k), "peak": PEAK_HBM_GBS, "unit": "GB/s", "achieved": round(v[2] / v[3] / 1e9, 0),
                            "frac": round(v[2] / v[3] / 1e9 / PEAK_HBM_GBS, 3), "avg_launch_us": round(v[3] / v[0] * 1e6, 1),
                            "algorithmic_bytes_per_launch": round(v[2] / v[0]), "launches_per_step": v[0] // 2, "ms_per_step": round(v[3] / 2 * 1e3, 3),
                            "note": "largest HBM-bound family of the step (algorithmic bytes / HIP-event time of the same eager passes); peak = 8 TB/s spec (6.3 TB/s is what a float4 copy reaches on this part)"}
        aux_flops = args.aux_images * SWIN_FWD_GFLOP_PER_FRAME * 3 * 1e9
        flops_step = args.utts * (args.frames * SWIN_FWD_GFLOP_PER_FRAME * 3 + (335 + 29.7) * 3) * 1e9 + aux_flops
        which = {1: "configs[1]", 3: "configs[3] (per-GPU leg of batch 16 on 4 GPUs)", 4: "configs[4] (per-GPU leg of batch 8 on 8 GPUs)"}[args.config]
        if args.config == 1 and world == 8:
            which = "configs[2]"
        plm_name = "RoBERTa-large" if args.plm == "roberta-large" else "BERT-large"
        # what RCCL itself saw: read back from its init log; None on gloo / without a communicator (never torch's world size)
        rccl_seen = rccl_ranks_from_log(rccl_log) if (dist.is_initialized() and dist.get_backend() == "nccl") else None
        line = {
            "metric": "utterances/sec T+A+V forward+bwd, 160-frame face seq, 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{which}: T+A+V, {plm_name} (random init), {args.frames}-frame face seq "
                                   f"({'uint8 112x112 crops, fused ' + args.resize + '-bicubic x2 + normalise' if args.input == 'u8' else 'pre-resized 3x224x224 frames'}), "
                                   f"batch {args.utts} utterances/GPU, {args.dtype}, fwd+bwd+clip+AdamW every step"
                                   + (f", preceded by one auxiliary-task step on {args.aux_images} Aff-Wild2-shaped crops (Swin fwd+bwd+clip+AdamW)" if args.aux_images else ""),
                       "global_batch": args.utts * world, "frames_per_step_per_gpu": args.utts * args.frames, "aux_images_per_step_per_gpu": args.aux_images,
                       "parallelism": f"dp{world}", "kept_frame_fraction": round(float(kept.mean().item()), 3),
                       "model_tflops_per_s_per_gpu": round(flops_step / (ms * 1e-3) / 1e12, 1),
                       "host_enqueue_ms_per_step": round(issue_s / args.steps * 1e3, 1), "host_issue_ms_into_idle_queue": round(idle_issue_ms, 1),
                       "hip_graphs": {2: ("whole step: 3 graphs (fwd + multimodal bwd | Swin bwd | clip+optimizer), gradient exchange beside the second" if ddp else ("whole step: 3 graphs (Swin fwd of the NEXT step on a second stream | everything else of this step | clip+optimizer)" if getattr(step, "pipeline", False) else ("whole step: 6 graphs on two streams (text fwd || Swin fwd -> filter + fusion + loss fwd/bwd -> text bwd + hand-over + clip+optimizer || Swin bwd)" if getattr(step, "branches", False) else "whole step: 2 graphs (fwd+bwd | clip+optimizer)"))), 1: "multimodal model only (Swin eager)", 0: "none"}[args.graphs],
                       "text_encoder_concurrent_with_swin": bool(args.graphs and args.overlap_text),
                       "swin_forward_pipelined_across_steps": bool(getattr(step, "pipeline", False)),
                       "text_encoder_parameters": "bf16 with fp32 master weights in the optimizer" if (args.graphs == 2 and args.plm_dtype == "bf16" and args.dtype == "bf16") else "fp32 under bf16 autocast",
                       "gradient_allreduce": None if not ddp else f"{args.grad_comm}, {'issued between the multimodal backward and the Swin backward (graphs A1 | A2), waited for before the optimizer graph' if args.graphs == 2 else 'hook-driven, overlapped with backward'}",
                       "rccl_ranks_seen": rccl_seen, "rccl_ranks_seen_source": ("RCCL init log (NCCL_DEBUG=INFO, 'nranks')" if rccl_seen is not None else None),
                       "exchange_ms_exposed": (xchg or {}).get("ms_exposed_after_swin_backward"), "exchange": xchg},
            "roofline": roof,
            "roofline_hbm": roof_hbm,
            "roofline_mfma": roof_mfma,
            "cpu_baseline": None,
            "pcie_inclusive": pcie,
        }
        if args.graphs == 1:
            line["config"]["second_stream_pair_over_single"] = round(float(getattr(mm, "text_stream_concurrency", 0.0)), 2)
        if step.host_ms:
            line["config"]["host_ms_per_phase"] = {k: round(v / args.steps, 1) for k, v in step.host_ms.items()}
            line["config"]["main_stream_gpu_ms_per_phase"] = {k: round(v / args.steps, 1) for k, v in gpu_phase.items()}
        if "FMMT_BENCH_DEVICE" in os.environ or backend != "nccl":
            line["config"]["dry_run"] = f"ranks share device {local}, backend {backend}: not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, cfg)
        if args.discarded_swin_gradients == "skip":
            line["config"]["workload"] += "; Swin's (discarded) target-step gradients NOT computed: not the reference's work per step, not comparable with `value` of the default run"
        if world == 1 and args.other_configs and args.config == 1 and args.discarded_swin_gradients == "compute":
            line["other_configs"] = other_configs(args)
            line["discarded_swin_gradients_skipped"] = skip_leg(args)
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


def bracket_coverage(timer, one_pass, pass_ms):
    """How much of one eager forward + backward pass the bracketed entry points account for (round-3 VERDICT item 7: >= 90 %).
    Denominator: the GPU time of EVERY kernel of one more eager pass, from torch.profiler's device-side records (roctracer) --
    ours plus PyTorch-ROCm's (text encoder, glue); of that, the part that is this library's is the share the families can cover at
    all, so two figures are given: bracketed / all kernels, and bracketed / (all kernels - kernels that are not this library's).
    Falls back to the event-bracketed wall time of the pass (which contains the host-bound gaps of eager issue) without a profiler."""
    ours_ms = sum(s.elapsed_time(e) for _, _, _, s, e, _ in timer.events) / 2
    out = {"bracketed_ms_per_pass": round(ours_ms, 2), "eager_pass_wall_ms": round(pass_ms, 2)}
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            one_pass()
            torch.cuda.synchronize()
        tot = lib_ms = 0.0
        mine = ("linear_nt", "linear_tn", "ln_fwd", "ln_bwd", "lnp_", "ln_reduce", "wattn_", "wblock_", "mlp_fused", "mlp_ln_part", "mha_", "patch_", "bn1d_",
                "reduce_partials", "splitk_finish", "colsum", "cast_batch", "adamw_batch", "posemb", "scale_kernel", "preproc", "im2col", "col2im")
        for ev in prof.events():
            if str(getattr(ev, "device_type", "")).endswith("CUDA") and getattr(ev, "device_time_total", 0) > 0:
                tot += ev.device_time_total * 1e-3
                if any(m in ev.name for m in mine):
                    lib_ms += ev.device_time_total * 1e-3
        if tot > 0:
            out.update({"all_kernels_ms_per_pass": round(tot, 2), "library_kernels_ms_per_pass": round(lib_ms, 2),
                        "bracketed_over_all_kernels": round(ours_ms / tot, 3), "bracketed_over_library_kernels": round(ours_ms / max(lib_ms, 1e-9), 3),
                        "source": "torch.profiler device records of one more eager pass (the optimizer-side launches are bracketed but not in that pass)"})
            if lib_ms > 0 and ours_ms / lib_ms < 0.9:
                print(f"bench.py: WARNING bracketed families cover {ours_ms / lib_ms:.2f} of the library's kernel time (< 0.90)", file=sys.stderr)
            return out
    except Exception as e:                                   # a reported figure only
        out["profiler_error"] = f"{type(e).__name__}: {str(e)[:160]}"
    out.update({"bracketed_over_eager_wall": round(ours_ms / max(pass_ms, 1e-9), 3),
                "source": "HIP events around two eager passes (wall time incl. host-bound gaps; no device-side profiler records available)"})
    return out


def other_configs(args):
    """BASELINE.json configs[3] (BERT-large) and configs[4] (320 frames + auxiliary task): per-GPU legs, 4 timed steps each, each in
    its own process (own models, own graphs) after this one's timed region -- reported, never part of `value`."""
    import subprocess
    out = {}
    for c in (3, 4):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", str(c), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--host-input-leg", "0",
               "--other-configs", "0", "--dtype", args.dtype, "--graphs", str(args.graphs)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
            j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            out[f"configs[{c}]"] = {"ms_per_step": j["ms_per_step"], "value": j["value"], "unit": j["unit"], "steps": j["steps"], "workload": j["config"]["workload"],
                                    "dominant_kernel": (j.get("roofline") or {}).get("kernel"), "bound": (j.get("roofline") or {}).get("bound"),
                                    "frac": (j.get("roofline") or {}).get("frac"),
                                    "traffic": (j.get("roofline") or {}).get("traffic")}
        except Exception as e:                               # a reported leg only
            out[f"configs[{c}]"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    return out


def skip_leg(args):
    """The same configs[1] step with train_step's discarded_swin_gradients="skip" (own process, 6 steps): a REPORTED leg -- the
    reference spends this backward on gradients it zeroes unread; skipping it leaves every later parameter unchanged to fp32 rounding
    (tests/test_gpu_train_step.py::test_skipping_the_discarded_swin_backward_changes_nothing).  Never `value`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--host-input-leg", "0", "--other-configs", "0",
           "--dtype", args.dtype, "--graphs", str(args.graphs), "--discarded-swin-gradients", "skip"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"ms_per_step": j["ms_per_step"], "value": j["value"], "unit": j["unit"], "steps": j["steps"],
                "note": "option train_step.*(discarded_swin_gradients='skip'): Swin forward without autograd, no Swin backward; the reference computes these "
                        "gradients and zeroes them unread (train.py:20,33,140-143); same training to fp32 rounding; NOT `value`"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}


def kernel_symbol(bn):
    """KernelTimer family tag -> name of the kernel template instantiation as rocprofv3 prints it"""
    if bn.startswith("linear_tn"):
        return bn
    if bn.startswith("fmmt_"):
        base, _, rest = bn.partition("<")
        names = {"fmmt_layernorm_fwd": "ln_fwd_kernel", "fmmt_layernorm_bwd": "ln_bwd_kernel", "fmmt_window_attn_fwd": "wattn_mfma_fwd_kernel",
                 "fmmt_window_attn_bwd": "wattn_mfma_bwd_kernel", "fmmt_mlp_fwd": "mlp_fused_fwd_kernel", "fmmt_window_block_fwd": "wblock_fwd_kernel",
                 "fmmt_mlp_ln_fwd": "mlp_fused_fwd_kernel<LN>", "fmmt_mlp_bwd_input": "mlp_fused_bwd_kernel", "fmmt_mlp_ln_bwd_input": "mlp_fused_bwd_kernel<LN'>",
                 "fmmt_window_block_attn_bwd": "wattn_mfma_bwd_kernel<recompute>", "fmmt_patch_embed_ln_fwd": "patch_embed_ln_kernel",
                 "fmmt_patch_embed_u8": "patch_embed_u8_kernel", "fmmt_patch_embed_u8_ln_fwd": "patch_embed_u8_kernel<fused projection + LayerNorm>", "fmmt_mha_fwd": "mha_mfma_fwd_kernel", "fmmt_mha_bwd": "mha_mfma_bwd_kernel",
                 "fmmt_adamw_batch": "adamw_batch_kernel", "fmmt_cast_batch": "cast_batch_kernel", "fmmt_grad_handover": "grad_handover_kernel", "fmmt_linear_ln_bwd": "lin_lnbwd_kernel", "fmmt_batchnorm1d_fwd": "bn1d_fwd_kernel",
                 "fmmt_batchnorm1d_bwd": "bn1d_bwd_kernel", "fmmt_linear_wgrad_finish": "reduce_partials_kernel", "fmmt_colsum": "colsum_kernel",
                 "fmmt_layernorm_bwd_bf16": "lnp_bwd_kernel", "fmmt_linear_fwd_splitk": "linear_splitk_kernel", "fmmt_linear_wgrad": "linear_tn_few_kernel"}
        return names.get(base, base) + ("<" + rest if rest else "")
    if bn == "ph256":
        return "linear_nt_ph_kernel<256x256>"
    if bn == "ph192":
        return "linear_nt_ph3_kernel<192x256>"
    if bn == "ph384":
        return "linear_nt_ph3_kernel<384x128>"
    if bn.startswith("p256x"):
        op, pipe = bn.endswith("op"), bn.endswith("pipe")
        w = bn[5:-2] if op else bn[5:-4] if pipe else bn[5:]
        return f"linear_nt_p256_kernel<{w},64,{3 if w == '128' else 2},true,{'true' if op else 'false'},{'true' if pipe else 'false'}>"
    if bn.startswith("deep256") and bn != "deep256x128x64":         # deep256x{128,96}x32[,nkN]
        width = "128" if bn.startswith("deep256x128") else "96"
        return f"linear_nt_deep32_kernel<{bn[-1] if ',nk' in bn else '0'},{width}>"
    if bn.startswith("deep"):
        return "linear_nt_deep_kernel"
    return f"linear_nt_kernel<bf16,{bn}>"


if __name__ == "__main__":
    main()

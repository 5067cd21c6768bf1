"""Generate tests/golden/*.npz by importing the REFERENCE itself (development container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden
Needs /root/reference (read-only) -- it never travels to the GPU box; only the small fixtures do.

Shims, all confined to this process:
  1. `timm` is not installed: a stub `timm.models.layers` provides DropPath (identity: the goldens pin
     the deterministic maths; stochastic depth is exercised with explicit masks against the oracle),
     to_2tuple and trunc_normal_.
  2. model code hard-codes .cuda(): Tensor.cuda becomes a no-op here.
Weights and inputs come from facialmmt_amd.synth (integer hash), so the fixtures hold only the
expected outputs (full if small, else a strided sample + fp64 sums) and the state_dict key lists.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_shims():
    import transformers  # noqa: F401  (must be imported before the timm stub exists: it probes find_spec("timm"))
    from transformers import BertModel, RobertaModel  # noqa: F401  (resolve the lazy modules now)
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            return x

    layers.DropPath = DropPath
    layers.to_2tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    layers.trunc_normal_ = nn.init.trunc_normal_
    timm.models = models
    models.layers = layers
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


SAMPLE_MAX = 6144


def pack(t: torch.Tensor) -> dict:
    """full tensor if small; else a strided sample plus fp64 sum / abs-sum over everything."""
    a = t.detach().to(torch.float32).contiguous().cpu().numpy().copy()   # copy: never alias live module state
    flat = a.reshape(-1)
    d = {"shape": np.array(a.shape, dtype=np.int64),
         "sum": np.array(flat.astype(np.float64).sum()),
         "abssum": np.array(np.abs(flat.astype(np.float64)).sum())}
    if flat.size <= SAMPLE_MAX:
        d["full"] = a
    else:
        stride = flat.size // 4096
        stride += 1 - (stride % 2)          # odd stride: walks every channel phase
        d["stride"] = np.array(stride, dtype=np.int64)
        d["sample"] = flat[::stride].copy()
    return d


def flatten(prefix: str, d: dict, out: dict):
    for k, v in d.items():
        out[f"{prefix}/{k}"] = v


def synth_multimodal_inputs(synth, B, T, La, Lv):
    """(ids, attn_mask, sep_mask, audio, audio_mask, vision(+7), vision_mask, utt_idx) for the G6 fixtures.
    Dialogue i has separators every 9+i tokens; target utterances 0, 2, 1; ragged masks."""
    ids = torch.from_numpy(synth.randint("mm_ids", (B, T), 3, 1000, seed=13))
    ids[:, 0] = 0
    attn = torch.zeros(B, T)
    sep = torch.zeros(B, T)
    for i in range(B):
        step = 9 + i
        n_valid = T - 6 * i
        attn[i, :n_valid] = 1
        for pos in range(step, n_valid, step):
            sep[i, pos] = 1
    audio = synth.tensor("mm_audio", (B, La, 300), seed=14)
    amask = torch.ones(B, La); amask[1, La - 5:] = 0
    vision = synth.tensor("mm_vision", (B, Lv, 519), seed=15)
    vmask = torch.ones(B, Lv); vmask[2, Lv - 7:] = 0
    utt = torch.tensor([0, 2, 1][:B])
    return ids, attn, sep, audio, amask, vision, vmask, utt


def main():
    _install_shims()
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from facialmmt_amd import synth
    from modules.SwinTransformer import Swin_Transformer as RS
    from modules.SwinTransformer.backbone_def import BackboneFactory
    from modules.CrossmodalTransformer import CrossModalTransformerEncoder
    from modules.multihead_attention import MultiheadAttention
    from modules.position_embedding import SinusoidalPositionalEmbedding
    import src.models as RM

    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    keys = {}

    # ------------------------------------------------------------------ Swin pieces (G1, G2)
    g = {}
    geo = [(56, 96, 3), (28, 192, 6), (14, 384, 12), (7, 768, 24)]
    for s, (H, C, nh) in enumerate(geo):
        for shift in (0, 3):
            blk = RS.SwinTransformerBlock(C, (H, H), nh, window_size=7, shift_size=shift, drop_path=0.0).eval()
            synth.fill_state_dict(blk, seed=10 + s, prefix=f"blk{s}.")
            keys[f"blk{s}_shift{shift}"] = [[k, list(v.shape), str(v.dtype)] for k, v in blk.state_dict().items()]
            x = synth.tensor(f"blk_in{s}", (2, H * H, C), seed=s)
            # G1: WindowAttention alone on the windows of the first image
            from oracle.swin import window_token_index
            idx = window_token_index(H, H, 7, blk.shift_size)
            xw = x[:1, idx.reshape(-1)].reshape(-1, 49, C)
            flatten(f"wattn_s{s}_shift{shift}", pack(blk.attn(xw, mask=blk.attn_mask)), g)
            # G2: the block
            flatten(f"block_s{s}_shift{shift}", pack(blk(x)), g)
            if blk.attn_mask is not None:
                flatten(f"mask_s{s}", pack(blk.attn_mask), g)
        if s < 3:
            pm = RS.PatchMerging((H, H), C).eval()
            synth.fill_state_dict(pm, seed=20 + s, prefix=f"pm{s}.")
            keys[f"pm{s}"] = [[k, list(v.shape), str(v.dtype)] for k, v in pm.state_dict().items()]
            flatten(f"merge_s{s}", pack(pm(synth.tensor(f"blk_in{s}", (2, H * H, C), seed=s))), g)
    pe = RS.PatchEmbed(224, 4, 3, 96, nn.LayerNorm).eval()
    synth.fill_state_dict(pe, seed=30, prefix="pe.")
    keys["pe"] = [[k, list(v.shape), str(v.dtype)] for k, v in pe.state_dict().items()]
    frames2 = synth.tensor("frames", (2, 3, 224, 224), seed=1)
    flatten("patch_embed", pack(pe(frames2)), g)
    g["rel_index"] = RS.WindowAttention(96, (7, 7), 3).relative_position_index.numpy()
    np.savez_compressed(os.path.join(OUT, "swin_parts.npz"), **g)

    # ------------------------------------------------------------------ whole Swin (G3, G4, G7)
    g = {}
    swin = BackboneFactory("SwinTransformer", os.path.join(REF, "modules/SwinTransformer/swin_conf.yaml")).get_backbone()
    synth.fill_state_dict(swin, seed=100)
    keys["swin"] = [[k, list(v.shape), str(v.dtype)] for k, v in swin.state_dict().items()]
    frames8 = synth.tensor("frames", (8, 3, 224, 224), seed=1)
    swin.eval()
    flatten("swin_eval_n8", pack(swin(frames8)), g)
    flatten("swin_eval_n1", pack(swin(frames8[:1])), g)
    # per-stage taps (eval, n=2) to localise a mismatch
    x = swin.patch_embed(frames8[:2])
    for s, layer in enumerate(swin.layers):
        x = layer(x)
        flatten(f"swin_stage{s}_n2", pack(x), g)
    swin.train()
    rm0, rv0 = swin.output_layer[3].running_mean.clone(), swin.output_layer[3].running_var.clone()
    flatten("swin_train_n4", pack(swin(frames8[:4])), g)
    flatten("bn_running_mean_after", pack(swin.output_layer[3].running_mean), g)
    flatten("bn_running_var_after", pack(swin.output_layer[3].running_var), g)
    swin.output_layer[3].running_mean.copy_(rm0)
    swin.output_layer[3].running_var.copy_(rv0)
    swin.output_layer[3].num_batches_tracked.zero_()

    class A:  # args namespace for src/models.py:16-24
        num_labels = 7
        backbone_type = "SwinTransformer"
        backbone_conf_file = os.path.join(REF, "modules/SwinTransformer/swin_conf.yaml")
        tau = 1.0
    aff = RM.SwinForAffwildClassification(A)
    synth.fill_state_dict(aff, seed=100)   # "swin." prefix is part of the key -> differs from bare swin
    keys["affwild"] = [[k, list(v.shape), str(v.dtype)] for k, v in aff.state_dict().items()]
    aff.eval()
    flatten("affwild_logits_n4", pack(aff(frames8[:4], is_trg_task=False)), g)

    # G7 gradients: train mode (BN batch stats), DropPath identity, loss = sum(out * probe)
    torch.set_grad_enabled(True)
    aff.train()
    xin = frames8[:3].clone().requires_grad_(True)   # 3 samples: BN over 2 would be degenerate (+-1)
    probe = synth.tensor("probe7", (3, 7), seed=3)
    (aff(xin, is_trg_task=False) * probe).sum().backward()
    flatten("grad/input", pack(xin.grad), g)
    for name in ["swin.patch_embed.proj.weight", "swin.patch_embed.norm.weight",
                 "swin.layers.0.blocks.0.attn.relative_position_bias_table",
                 "swin.layers.0.blocks.1.attn.relative_position_bias_table",
                 "swin.layers.0.blocks.1.attn.qkv.weight", "swin.layers.0.blocks.1.attn.qkv.bias",
                 "swin.layers.0.blocks.1.norm1.weight", "swin.layers.0.blocks.1.norm1.bias",
                 "swin.layers.0.downsample.norm.weight", "swin.layers.0.downsample.reduction.weight",
                 "swin.layers.1.blocks.1.mlp.fc1.weight", "swin.layers.1.blocks.1.mlp.fc2.bias",
                 "swin.layers.2.blocks.5.attn.proj.weight",
                 "swin.layers.3.blocks.1.attn.relative_position_bias_table",
                 "swin.layers.3.blocks.1.mlp.fc2.weight",
                 "swin.output_layer.0.weight", "swin.output_layer.2.bias", "swin.output_layer.3.weight",
                 "swin.output_layer.3.bias", "linear.weight", "classifier.bias"]:
        flatten("grad/" + name, pack(dict(aff.named_parameters())[name].grad), g)
    flatten("grad/swin.output_layer.2.weight", pack(aff.swin.output_layer[2].weight.grad), g)
    torch.set_grad_enabled(False)
    np.savez_compressed(os.path.join(OUT, "swin_full.npz"), **g)

    # ------------------------------------------------------------------ cross-modal (G5, G7)
    g = {}
    pos = SinusoidalPositionalEmbedding(768)
    pin = torch.tensor([[0.5, 0.0, -1.2, 0.0, 3.0, 1e-30, -0.0], [0.0, 2.0, 0.0, 0.0, 1.0, 1.0, 7.0]])
    g["posemb/in"] = pin.numpy()
    flatten("posemb/out", pack(pos(pin)), g)

    mha = MultiheadAttention(768, 12, attn_dropout=0.1).eval()
    synth.fill_state_dict(mha, seed=40, prefix="mha.")
    keys["mha"] = [[k, list(v.shape), str(v.dtype)] for k, v in mha.state_dict().items()]
    q = synth.tensor("mha_q", (38, 2, 768), seed=5)
    kv = synth.tensor("mha_kv", (128, 2, 768), seed=6)
    o, w = mha(q, kv, synth.tensor("mha_v", (128, 2, 768), seed=7))
    flatten("mha/out", pack(o), g)
    flatten("mha/weights", pack(w), g)

    enc = CrossModalTransformerEncoder(768, 12, 2, 0.1).eval()
    synth.fill_state_dict(enc, seed=50, prefix="enc.")
    keys["crossmodal"] = [[k, list(v.shape), str(v.dtype)] for k, v in enc.state_dict().items()]

    def seq(name, L, B, n_zero_rows, seed):
        t = synth.tensor(name, (L, B, 768), seed=seed)
        if n_zero_rows:
            t[L - n_zero_rows:] = 0.0            # zero-padded utterance rows (src/models.py:114,130)
        t[1, 0, 0] = 0.0                         # a lone exact zero in channel 0 -> position 0
        return t

    for (Lq, Lk) in [(38, 128), (128, 38), (160, 166), (166, 160)]:
        for B in (1, 4):
            xq = seq(f"x{Lq}", Lq, B, 5 if Lq == 38 else 0, seed=60)
            xk = seq(f"x{Lk}", Lk, B, 5 if Lk == 38 else 0, seed=60)
            flatten(f"enc/{Lq}_{Lk}_b{B}", pack(enc(xq, xk, xk)), g)
    flatten("enc/self_38_b2", pack(enc(seq("x38", 38, 2, 5, 60))), g)

    # smoke script of the reference (CrossmodalTransformer.py:201-214): (500,4 heads,2 layers), q 15, kv 40
    enc500 = CrossModalTransformerEncoder(500, 4, 2, 0, 0, 0, 0).eval()
    synth.fill_state_dict(enc500, seed=51, prefix="enc500.")
    keys["enc500"] = [[k, list(v.shape), str(v.dtype)] for k, v in enc500.state_dict().items()]
    flatten("enc/smoke500", pack(enc500(synth.tensor("s_q", (15, 2, 500), seed=8),
                                        synth.tensor("s_k", (40, 2, 500), seed=9),
                                        synth.tensor("s_v", (40, 2, 500), seed=10))), g)

    torch.set_grad_enabled(True)
    enc.eval()                                   # dropout off, grads on
    xq = seq("x38", 38, 2, 5, 60).requires_grad_(True)
    xk = seq("x128", 128, 2, 0, 60).requires_grad_(True)
    out = enc(xq, xk, xk)
    (out * synth.tensor("probe_enc", tuple(out.shape), seed=11)).sum().backward()
    flatten("grad/xq", pack(xq.grad), g)
    flatten("grad/xk", pack(xk.grad), g)
    for name in ["layers.0.self_attn.in_proj_weight", "layers.0.self_attn.in_proj_bias",
                 "layers.1.self_attn.out_proj.weight", "layers.0.layer_norms.0.weight",
                 "layers.1.layer_norms.1.bias", "layers.1.fc1.weight", "layers.0.fc2.bias", "layer_norm.weight"]:
        flatten("grad/" + name, pack(dict(enc.named_parameters())[name].grad), g)
    torch.set_grad_enabled(False)
    np.savez_compressed(os.path.join(OUT, "crossmodal.npz"), **g)

    # ------------------------------------------------------------------ callers (G6): multimodal model with a stand-in PLM
    g = {}
    from facialmmt_amd.config import default_args
    for plm_name in ("roberta-large", "bert-large"):
        cfg = default_args(get_audio_utt_max_lens=24, get_vision_utt_max_lens=20,
                           pretrainedtextmodel_path="pretrained_model/" + plm_name)
        standin = synth.make_standin_plm()
        RM.RobertaModel.from_pretrained = staticmethod(lambda path: standin)
        RM.BertModel.from_pretrained = staticmethod(lambda path: standin)
        mm = RM.MultiModalTransformerForClassification(cfg).eval()
        synth.fill_state_dict(mm, seed=200)
        standin.emb.weight.copy_(synth.make_standin_plm().emb.weight)       # fill_state_dict re-filled it: restore the stand-in table
        tag = plm_name.split("-")[0]
        keys["multimodal_" + tag] = [[k, list(v.shape), str(v.dtype)] for k, v in mm.state_dict().items()]
        inp = synth_multimodal_inputs(synth, B=3, T=64, La=24, Lv=20)
        flatten(f"mm/{tag}", pack(mm(*inp)), g)
    cfg = default_args(get_vision_utt_max_lens=20)
    vm = RM.meld_utt_transformer(cfg).eval()
    synth.fill_state_dict(vm, seed=201)
    keys["meld_utt"] = [[k, list(v.shape), str(v.dtype)] for k, v in vm.state_dict().items()]
    vin = synth.tensor("vfeat", (2, 20, 512), seed=12)
    vmask = torch.ones(2, 20); vmask[1, 14:] = 0
    flatten("meld_utt", pack(vm(vin, vmask)), g)
    np.savez_compressed(os.path.join(OUT, "multimodal.npz"), **g)

    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    main_round2(synth, RM, CrossModalTransformerEncoder, seq)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


def _seq(synth):
    def seq(name, L, B, n_zero_rows, seed):
        t = synth.tensor(name, (L, B, 768), seed=seed)
        if n_zero_rows:
            t[L - n_zero_rows:] = 0.0
        t[1, 0, 0] = 0.0
        return t
    return seq


def only_round2():
    """`python -m oracle.gen_golden --round2`: the round-2 files alone (minutes instead of the full regeneration)"""
    _install_shims()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from facialmmt_amd import synth
    from modules.CrossmodalTransformer import CrossModalTransformerEncoder
    import src.models as RM
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    main_round2(synth, RM, CrossModalTransformerEncoder, _seq(synth))
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


def main_round2(synth, RM, CrossModalTransformerEncoder, seq):
    """Round-2 fixtures, written to NEW files (the four round-1 files above stay byte-identical):

    lv320.npz    BASELINE.json configs[4] (320-frame face sequence): the cross-modal encoder at (320;166) and (166;320), the
                 vision self-attention encoder + pooling at L = 320 (meld_utt_transformer), and the whole multimodal model
                 with L_v = 320 -- all produced by the REFERENCE's classes, same hash-generated weights as round 1.
    preproc.npz  the input pre-step: Pillow's own Image.resize(BICUBIC) (what transforms.Resize does to a PIL image,
                 utils/util.py:45) on hash-generated 112x112 and 160x160 uint8 crops, and the float tensor
                 ToTensor + Normalize(.5,.5) make of it (computed with torch as torchvision does: /255, (t-.5)/.5)."""
    from facialmmt_amd.config import default_args
    g = {}
    enc = CrossModalTransformerEncoder(768, 12, 2, 0.1).eval()
    synth.fill_state_dict(enc, seed=50, prefix="enc.")
    for (Lq, Lk) in [(320, 166), (166, 320)]:
        xq = seq(f"x{Lq}", Lq, 1, 0, seed=60)
        xk = seq(f"x{Lk}", Lk, 1, 0, seed=60)
        flatten(f"enc/{Lq}_{Lk}_b1", pack(enc(xq, xk, xk)), g)
    cfg = default_args(get_vision_utt_max_lens=320)
    vm = RM.meld_utt_transformer(cfg).eval()
    synth.fill_state_dict(vm, seed=201)
    vin = synth.tensor("vfeat320", (2, 320, 512), seed=12)
    vmask = torch.ones(2, 320); vmask[1, 250:] = 0
    flatten("meld_utt_320", pack(vm(vin, vmask)), g)
    cfg = default_args(get_audio_utt_max_lens=24, get_vision_utt_max_lens=320, pretrainedtextmodel_path="pretrained_model/roberta-large")
    standin = synth.make_standin_plm()
    RM.RobertaModel.from_pretrained = staticmethod(lambda path: standin)
    mm = RM.MultiModalTransformerForClassification(cfg).eval()
    synth.fill_state_dict(mm, seed=200)
    standin.emb.weight.copy_(synth.make_standin_plm().emb.weight)
    inp = synth_multimodal_inputs(synth, B=3, T=64, La=24, Lv=320)
    flatten("mm/roberta_lv320", pack(mm(*inp)), g)
    np.savez_compressed(os.path.join(OUT, "lv320.npz"), **g)

    from PIL import Image
    g = {"pillow_version": np.array(Image.__version__ if hasattr(Image, "__version__") else __import__("PIL").__version__)}
    for S in (112, 160):
        crops = synth.randint("crop", (2, S, S, 3), 0, 256, seed=7).astype(np.uint8)
        res = np.stack([np.asarray(Image.fromarray(c, "RGB").resize((224, 224), Image.BICUBIC)) for c in crops])
        g[f"pil_resize_{S}"] = res
        if S == 112:
            t = torch.from_numpy(res[:1]).permute(0, 3, 1, 2).to(torch.float32).div(255)  # ToTensor
            g["pil_frames_112"] = t.sub(0.5).div(0.5).numpy()                               # Normalize(.5, .5)
    np.savez_compressed(os.path.join(OUT, "preproc.npz"), **g)


if __name__ == "__main__":
    only_round2() if "--round2" in sys.argv else main()

"""Development probe: the Swin modules on the HIP path against the golden fixtures (fp32) and
bf16-vs-fp32 drift.  python tools/probes/probe_swin.py > gpurun_out/probe_swin.log 2>&1"""
import os
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import synth  # noqa: E402
from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as S  # noqa: E402
from facialmmt_amd.modules.SwinTransformer.backbone_def import BackboneFactory  # noqa: E402
from tests.golden_util import Golden  # noqa: E402

dev = torch.device("cuda:0")
G = Golden()
CONF = os.path.join(os.path.dirname(S.__file__), "swin_conf.yaml")


def chk(file, name, t, **kw):
    try:
        e = G.check(file, name, t, **kw)
        print(f"OK   {name:45s} max|err|={e:.3e}", flush=True)
    except AssertionError as ex:
        print(f"FAIL {name:45s} {str(ex)[:160]}", flush=True)


def main():
    GEO = [(56, 96, 3), (28, 192, 6), (14, 384, 12), (7, 768, 24)]
    torch.set_grad_enabled(False)
    for s, (H, C, nh) in enumerate(GEO):
        for shift in (0, 3):
            blk = S.SwinTransformerBlock(C, (H, H), nh, window_size=7, shift_size=shift, drop_path=0.0).eval()
            synth.fill_state_dict(blk, seed=10 + s, prefix=f"blk{s}.")
            blk.to(dev)
            x = synth.tensor(f"blk_in{s}", (2, H * H, C), seed=s).to(dev)
            chk("swin_parts", f"block_s{s}_shift{shift}", blk(x))
            from oracle.swin import window_token_index
            idx = window_token_index(H, H, 7, blk.shift_size).to(dev)
            xw = x[:1, idx.reshape(-1)].reshape(-1, 49, C)
            chk("swin_parts", f"wattn_s{s}_shift{shift}", blk.attn(xw, mask=blk.attn_mask))
        if s < 3:
            pm = S.PatchMerging((H, H), C).eval()
            synth.fill_state_dict(pm, seed=20 + s, prefix=f"pm{s}.")
            pm.to(dev)
            chk("swin_parts", f"merge_s{s}", pm(synth.tensor(f"blk_in{s}", (2, H * H, C), seed=s).to(dev)))
    pe = S.PatchEmbed(224, 4, 3, 96, torch.nn.LayerNorm).eval()
    synth.fill_state_dict(pe, seed=30, prefix="pe.")
    pe.to(dev)
    chk("swin_parts", "patch_embed", pe(synth.tensor("frames", (2, 3, 224, 224), seed=1).to(dev)))

    swin = BackboneFactory("SwinTransformer", CONF).get_backbone()
    synth.fill_state_dict(swin, seed=100)
    swin.to(dev).eval()
    frames = synth.tensor("frames", (8, 3, 224, 224), seed=1).to(dev)
    chk("swin_full", "swin_eval_n8", swin(frames))
    chk("swin_full", "swin_eval_n1", swin(frames[:1]))
    x = swin.patch_embed(frames[:2])
    for s, layer in enumerate(swin.layers):
        x = layer(x)
        chk("swin_full", f"swin_stage{s}_n2", x)
    # bf16 drift
    o32 = swin(frames)
    o16 = swin(frames.bfloat16()).float()
    print(f"     bf16 vs fp32 (eval n8): max|d|={(o16 - o32).abs().max().item():.3e} ref_scale={o32.abs().max().item():.3e}")

    # train mode: DropPath off for the golden comparison (goldens were made with identity DropPath)
    for m in swin.modules():
        if isinstance(m, S.DropPath):
            m.drop_prob = 0.0
    swin.train()
    bn = swin.output_layer[3]
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    chk("swin_full", "swin_train_n4", swin(frames[:4]), atol=2e-3, rtol=2e-3)
    chk("swin_full", "bn_running_mean_after", bn.running_mean)
    chk("swin_full", "bn_running_var_after", bn.running_var)
    bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)

    # gradients vs golden (SwinForAffwildClassification = swin + 2 torch Linear layers)
    torch.set_grad_enabled(True)
    lin = torch.nn.Linear(512, 64).to(dev)
    cls = torch.nn.Linear(64, 7).to(dev)
    keys = G.keys["affwild"]
    sd = synth.state_dict_from_keys(keys, seed=100)
    swin.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith("swin.")}, strict=False)
    lin.load_state_dict({"weight": sd["linear.weight"], "bias": sd["linear.bias"]})
    cls.load_state_dict({"weight": sd["classifier.weight"], "bias": sd["classifier.bias"]})
    swin.eval()
    with torch.no_grad():
        chk("swin_full", "affwild_logits_n4", cls(torch.relu(lin(swin(frames[:4])))))
    swin.train()
    xin = frames[:3].clone().requires_grad_(True)
    probe = synth.tensor("probe7", (3, 7), seed=3).to(dev)
    t0 = time.time()
    (cls(torch.relu(lin(swin(xin)))) * probe).sum().backward()
    torch.cuda.synchronize()
    print(f"     fwd+bwd n=3 fp32 took {time.time() - t0:.3f}s")
    chk("swin_full", "grad/input", xin.grad, atol=2e-5, rtol=5e-3, sum_rtol=1e-3)
    params = dict(swin.named_parameters())
    z = G.files["swin_full"]
    for n in sorted({k.split("/")[1] for k in z.files if k.startswith("grad/swin.")}):
        ref, _ = G.expected("swin_full", f"grad/{n}")
        scale = float(np.abs(ref).max())
        if n == "swin.output_layer.2.bias":
            continue
        chk("swin_full", f"grad/{n}", params[n[5:]].grad, atol=1e-3 * scale + 1e-7, rtol=5e-3, sum_rtol=2e-3)


if __name__ == "__main__":
    try:
        main()
    except Exception:
        traceback.print_exc()

"""Development probe: forward error of the bf16 Swin (N frames) against its own fp32 run, per stage, fused attention half on / off."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):                                # A/B of two builds in one gpurun call
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops, synth
from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as S
from facialmmt_amd.modules.SwinTransformer.backbone_def import BackboneFactory
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
swin = BackboneFactory("SwinTransformer", os.path.join(os.path.dirname(S.__file__), "swin_conf.yaml")).get_backbone()
synth.fill_state_dict(swin, seed=100)
swin.to(dev).eval()
frames = synth.tensor("frames", (N, 3, 224, 224), seed=1).to(dev)
acts = {}
def hook(name):
    def f(m, i, o):
        acts[name] = o.detach().float()
    return f
for i, blk in enumerate(swin.layers[0].blocks):
    blk.register_forward_hook(hook(f"s0b{i}"))
for i, l in enumerate(swin.layers):
    l.register_forward_hook(hook(f"stage{i}"))
def run(dtype, fused):
    ops._WBLOCK = fused
    acts.clear()
    with torch.no_grad():
        out = swin(frames.to(dtype)).float()
    return dict(acts, out=out)
ref = run(torch.float32, False)
for fused in (False, True):
    got = run(torch.bfloat16, fused)
    print("fused" if fused else "four-launch", " ".join(f"{k}:{((got[k]-ref[k]).norm()/ref[k].norm()).item():.5f}" for k in ref))
# first block only, attention half: where inside?
blk = swin.layers[0].blocks[0]
x = ref["s0b0"] * 0 + 0  # placeholder

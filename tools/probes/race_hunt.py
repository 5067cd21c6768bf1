"""Development aid: Swin forward + backward at the bench size (640 frames, bf16, eval mode) repeated; every output / gradient compared bit
for bit with the first pass; names of the tensors that differ (a race shows up as a difference; the first differing tensor in backward
order points at the kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import synth
from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as S
from facialmmt_amd.modules.SwinTransformer.backbone_def import BackboneFactory
dev = torch.device("cuda:0")
swin = BackboneFactory("SwinTransformer", os.path.join(os.path.dirname(S.__file__), "swin_conf.yaml")).get_backbone()
synth.fill_state_dict(swin, seed=100)
swin = swin.to(dev)
swin.eval() if "--train" not in sys.argv else swin.train()
g = torch.Generator(device="cpu").manual_seed(2)
N = 640
frames = torch.randn(N, 3, 224, 224, generator=g).bfloat16().to(dev).requires_grad_(True)
w = torch.randn(N, 512, generator=g).to(dev)
named = [(n, p) for n, p in swin.named_parameters() if p.requires_grad]
reps = int(os.environ.get("REPS", "30"))
first = None
nbad = 0
for it in range(reps):
    torch.manual_seed(1234)                                  # --train: the same DropPath masks in every pass
    torch.cuda.manual_seed_all(1234)
    out = swin(frames)
    grads = torch.autograd.grad((out.float() * w).sum(), [frames] + [p for _, p in named], allow_unused=True)
    cur = [("out", out.detach())] + [("d_frames", grads[0])] + [(n, gr) for (n, _), gr in zip(named, grads[1:]) if gr is not None]
    if first is None:
        first = [(n, t.clone()) for n, t in cur]
        continue
    diff = [n for (n, a), (_, b) in zip(first, cur) if not torch.equal(a, b)]
    if diff:
        nbad += 1
        print(f"iteration {it}: {len(diff)} tensors differ; in registration order, first: {diff[:3]} ... last: {diff[-4:]}", flush=True)
print(f"{reps} passes, {nbad} with differences", flush=True)

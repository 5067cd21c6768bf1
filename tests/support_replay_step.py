"""Development aid: the forward + backward graph of the benchmark step (configs[1], bf16) replayed with the same inputs and the same
generator state; loss, kept-frame mask and every flat gradient bucket compared bit for bit with the first replay (graph B -- clip and
optimizer -- is never replayed, so the parameters stay put)."""
import os, sys, types
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from facialmmt_amd.config import default_args
from facialmmt_amd.parallel import GradientAverager
from facialmmt_amd.train_step import GraphedTargetStep, HFAdamW, MasterWeights, step_parameters
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
cfg = default_args(get_vision_utt_max_lens=args.frames, trg_accumulation_steps=1)
swin, mm = bench.build_models(args, dev, cfg)
batch = bench.synth_batch(args, dev, 0, cfg)
TAP = {}
if os.environ.get("TAP", "0") == "1":                       # experiment: keep the gradient that reaches vision_linear's output (a captured copy)
    class Tap(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, dy):
            if "dy" not in TAP:
                TAP["dy"] = torch.empty_like(dy)
            TAP["dy"].copy_(dy)
            return dy
    lin = mm.vision_linear
    orig = lin.forward
    lin.forward = lambda x: Tap.apply(orig(x))
masters = MasterWeights(mm.roberta, torch.bfloat16)
params = step_parameters(mm, masters)
flat = GradientAverager(params, hooks=False)
opt = HFAdamW(params, lr=torch.tensor(cfg.trg_lr, device=dev), weight_decay=cfg.weight_decay)
step = GraphedTargetStep(swin, mm, opt, None, cfg, batch, autocast_dtype=torch.bfloat16, averager=flat, masters=masters,
                         overlap_text=os.environ.get("OVERLAP_TEXT", "1") == "1", parallel_fusion=os.environ.get("PARALLEL_FUSION", "0") == "1")
reps = int(os.environ.get("REPS", "40"))
first, nbad = None, 0
for it in range(reps):
    torch.manual_seed(4321)
    torch.cuda.manual_seed_all(4321)
    flat.zero_grad()
    step.graph_a.replay()
    torch.cuda.synchronize()
    cur = [("loss", step.loss.clone()), ("kept", step.new_mask.clone())] + ([("tap_dy", TAP["dy"].clone())] if "dy" in TAP else []) + [(f"bucket{i}", b[0].clone()) for i, b in enumerate(flat.buckets)]
    if first is None:
        first = cur
        print("loss", float(step.loss), "kept", float(step.new_mask.sum()), flush=True)
        continue
    diff = [n for (n, a), (_, b) in zip(first, cur) if not torch.equal(a, b)]
    if diff:
        nbad += 1
        print(f"replay {it}: differ: {diff[:8]}{' ...' if len(diff) > 8 else ''}", flush=True)
        names = {id(p): n for n, p in mm.named_parameters()}
        names.update({id(m): "master:" + names.get(id(l), "?") for l, m in masters.pairs()})
        for (n, a), (_, b) in zip(first, cur):
            if n.startswith("bucket") and not torch.equal(a, b):
                bi = int(n[6:])
                off = 0
                for q in flat.buckets[bi][1]:
                    k = q.numel()
                    sa, sb = a[off:off + k], b[off:off + k]
                    if not torch.equal(sa, sb):
                        d = (sa - sb).abs()
                        print(f"     {names.get(id(q), '?')} {tuple(q.shape)}: {int((d > 0).sum())} elements, max |diff| {d.max().item():.3e} of max |grad| {sa.abs().max().item():.3e}", flush=True)
                    off += (k + 3) // 4 * 4
print(f"{reps} replays, {nbad} with differences", flush=True)

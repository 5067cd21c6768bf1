"""Who launches the step's small kernels?  One eager target step (bench.py's models and batch, --graphs 0) under torch.profiler with
Python stacks; device time and launch counts of every kernel shorter than 30 us on average, grouped by the innermost stack frame that
lies in facialmmt_amd/, bench.py or transformers/.   python tools/probes/small_kernels.py [--top 60]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "--graphs", "0", "--no-cpu-baseline"] + sys.argv[1:]
import torch
import bench
from torch.profiler import profile, ProfilerActivity

top = 60
if "--top" in sys.argv:
    i = sys.argv.index("--top"); top = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
args = bench.parse()
dev = torch.device("cuda:0")
from facialmmt_amd.config import default_args
from facialmmt_amd.train_step import HFAdamW, TargetStep
cfg = default_args(get_vision_utt_max_lens=args.frames, trg_accumulation_steps=1)
swin, mm = bench.build_models(args, dev, cfg)
batch = bench.synth_batch(args, dev, 0, cfg)
opt = HFAdamW(mm.parameters(), lr=cfg.trg_lr, weight_decay=cfg.weight_decay)
step = TargetStep(swin, mm, opt, None, cfg, autocast_dtype=torch.bfloat16)
for _ in range(2):
    step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(batch)
    torch.cuda.synchronize()
evs = prof.events()
# kernel events carry no stack: attribute each to the CPU op that launched it (same correlation: the kernel's parent launch call
# lies inside the op's CPU interval); torch links them through `linked_correlation_id`-less children, so walk the op tree instead
by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
def site_of(stack):
    for fr in stack:
        if ("facialmmt_amd" in fr or "bench.py" in fr or "transformers/" in fr) and "torch/" not in fr:
            return fr.strip()
    return (stack[0].strip() if stack else "?")
for ev in evs:
    if ev.device_type.name != "CPU" or not ev.kernels:
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.kernels:        # count kernels once, at the outermost op that owns them
        continue
    site = site_of(ev.stack or [])
    for k in ev.kernels:
        if k.duration > 30.0:
            continue
        s = by_site[site]
        s[0] += 1
        s[1] += k.duration
        s[2][k.name[:70]] += 1
tot_n = sum(v[0] for v in by_site.values()); tot_t = sum(v[1] for v in by_site.values())
print(f"kernels shorter than 30 us: {tot_n} launches, {tot_t / 1e3:.2f} ms of device time in one eager step")
for site, (n, t, names) in sorted(by_site.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n:5d} {t / 1e3:7.3f} ms  {site[-150:]}")
    for nm, c in names.most_common(4):
        print(f"            {c:4d} x {nm}")

"""Development timing: Swin fwd / fwd+bwd at the bench size (bf16, N frames), eager vs HIP-graph replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):                                # A/B of two builds in one gpurun call
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import synth
from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as S
from facialmmt_amd.modules.SwinTransformer.backbone_def import BackboneFactory
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 640
swin = BackboneFactory("SwinTransformer", os.path.join(os.path.dirname(S.__file__), "swin_conf.yaml")).get_backbone()
synth.fill_state_dict(swin, seed=100)
swin.to(dev).train()
x = torch.randn(N, 3, 224, 224, device=dev, dtype=torch.bfloat16)
def run(model, bwd, tag):
    def step():
        out = model(x)
        if bwd:
            out.float().square().mean().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    ti = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    fl = 9.0255e9 * N * (3 if bwd else 1)
    print(f"N={N} {tag:14s} {'fwd+bwd' if bwd else 'fwd    '}: {dt*1e3:8.2f} ms (host enqueue {ti*1e3:7.2f} ms)  {N/dt:9.1f} frames/s  {fl/dt/1e12:7.1f} TF/s  mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
run(swin, False, "eager")
run(swin, True, "eager")
if "--graph" in sys.argv:
    from facialmmt_amd.train_step import capture_window
    with capture_window():                                  # no cyclic GC inside the capture (train_step.capture_window)
        g = torch.cuda.make_graphed_callables(swin, (x,))
    run(g, True, "hipGraph")

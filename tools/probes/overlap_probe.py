"""Development probe: does the text encoder (graph-captured, small launches) overlap with the Swin
forward/backward (large launches) when issued on a second HIP stream?  Prints sequential vs concurrent time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transformers import RobertaConfig, RobertaModel
from facialmmt_amd import synth
from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as S
from facialmmt_amd.modules.SwinTransformer.backbone_def import BackboneFactory
dev = torch.device("cuda:0")
N = 640
swin = BackboneFactory("SwinTransformer", os.path.join(os.path.dirname(S.__file__), "swin_conf.yaml")).get_backbone()
synth.fill_state_dict(swin, seed=100)
swin.to(dev).train()
x = torch.randn(N, 3, 224, 224, device=dev, dtype=torch.bfloat16)
plm = RobertaModel(RobertaConfig(vocab_size=50265, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                                 max_position_embeddings=514, type_vocab_size=1, pad_token_id=1), add_pooling_layer=False).to(dev).train()
ids = torch.randint(3, 50265, (4, 512), device=dev)
am = torch.ones(4, 512, device=dev)

class Wrap(torch.nn.Module):
    def __init__(self, m):
        super().__init__(); self.m = m
    def forward(self, ids, am):
        return self.m(ids, am)[0]
with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
    gplm = torch.cuda.make_graphed_callables(Wrap(plm), (ids, am), num_warmup_iters=3)

def swin_step():
    swin(x).float().square().mean().backward()
def plm_step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        gplm(ids, am).float().square().mean().backward()
side = torch.cuda.Stream()
def both_seq():
    plm_step(); swin_step()
def both_conc():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plm_step()
    swin_step()
    torch.cuda.current_stream().wait_stream(side)
def timeit(fn, n=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"swin fwd+bwd          {timeit(swin_step):7.2f} ms")
print(f"plm  fwd+bwd (graph)  {timeit(plm_step):7.2f} ms")
print(f"sequential            {timeit(both_seq):7.2f} ms")
print(f"concurrent (2 streams){timeit(both_conc):7.2f} ms")

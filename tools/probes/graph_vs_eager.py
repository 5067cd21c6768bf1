"""Development check: eager vs graphed training trajectories with a real (2-layer, dropout-free) HF RoBERTa under bf16
autocast -- per-step worst gradient difference (0 with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; diverges from the third step
with ROCm 7.0's default).  Usage: python tools/probes/graph_vs_eager.py"""
import sys, os, types
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from transformers import RobertaConfig
from facialmmt_amd import models
from facialmmt_amd.config import default_args
from facialmmt_amd.train_step import TargetStep, graph_multimodal, select_frames
dev = torch.device("cuda:0")
NST = 5
def run(graph):
    args = types.SimpleNamespace(utts=4, frames=8, dtype="bf16")
    cfg = default_args(get_vision_utt_max_lens=8, trg_accumulation_steps=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       crossmodal_attn_dropout_TA=0.0, crossmodal_attn_dropout_TA_V=0.0)
    cfg.compute_dtype = torch.bfloat16
    cfg.plm_config = RobertaConfig(vocab_size=50265, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                                   max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg.plm_no_pooler = True
    torch.manual_seed(cfg.seed)
    swin = models.SwinForAffwildClassification(cfg).to(dev).train()
    mm = models.MultiModalTransformerForClassification(cfg).to(dev).train()
    batch = bench.synth_batch(args, dev, 0, cfg)
    if graph:
        with torch.no_grad():
            preds = swin(batch[8], is_trg_task=True).float()
        vis, nmask = select_frames(preds, batch[5], batch[6], batch[9], cfg.FacialEmoImpor_threshold)
        sample = (batch[0], batch[1], batch[2], batch[3], batch[4], vis.detach().requires_grad_(True), nmask, batch[10])
        mm = graph_multimodal(mm, sample, torch.bfloat16, overlap_text=False)
        mm.zero_grad(set_to_none=True); swin.zero_grad(set_to_none=True)
    opt = torch.optim.SGD(mm.parameters(), lr=0.02)
    step = TargetStep(swin, mm, opt, None, cfg, autocast_dtype=torch.bfloat16)
    grads = {}
    for k, p in mm.named_parameters():
        p.register_post_accumulate_grad_hook(lambda q, k=k: grads.__setitem__(k, q.grad.detach().float().clone()))
    out = []
    for i in range(NST):
        torch.manual_seed(1234 + i)
        loss, _ = step(batch)
        torch.cuda.synchronize()
        out.append((loss.item(), dict(grads)))
    return out
e, g = run(False), run(True)
for i in range(NST):
    rows = []
    for k in e[i][1]:
        a, b = e[i][1][k], g[i][1][k]
        rows.append(((a - b).abs().max().item() / max(a.abs().max().item(), 1e-20), b.abs().max().item() == 0.0 and a.abs().max().item() > 0, k))
    rows.sort(reverse=True)
    zeros = [k for _, z, k in rows if z]
    print(f"step {i}: loss {e[i][0]:.6f} {g[i][0]:.6f}; worst rel grad diff {rows[0][0]:.3e} ({rows[0][2]}); params with all-zero graph grad but non-zero eager grad: {len(zeros)} {zeros[:6]}")

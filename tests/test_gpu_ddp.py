"""Data-parallel target step on the REAL models: two ranks (two processes sharing the one GPU of the test
box, gloo transport -- RCCL refuses two ranks per device) each run the target-task step on their own utterance with
parallel.GradientAverager, and must end with the parameters a single process gets from the mean gradient of both
utterances.  Two variants of the step:

  hooks   TargetStep + hook-driven averager: two stream groups (text encoder / fusion stack, one communicator each),
          several buckets, gradient accumulation over two micro-steps (exchange only on the second);
  graphs  GraphedTargetStep: the step as two HIP graphs with the bucket all-reduces issued between them.

BatchNorm1d of the Swin head stays per replica (batch statistics of the rank's own frames), as under the reference's
single device; the single-process reference therefore runs the two utterances one after the other, never concatenated."""
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev, accumulation):
    from transformers import RobertaConfig
    from facialmmt_amd import models, synth
    from facialmmt_amd.config import default_args
    cfg = default_args(get_vision_utt_max_lens=4, get_audio_utt_max_lens=24, trg_accumulation_steps=accumulation,
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, crossmodal_attn_dropout_TA=0.0, crossmodal_attn_dropout_TA_V=0.0,
                       tau=1e5, FacialEmoImpor_threshold=0.1, audio_utt_Transformernum=1, vision_utt_Transformernum=1)
    cfg.compute_dtype = torch.float32
    cfg.plm_config = RobertaConfig(vocab_size=1000, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                                   max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, hidden_dropout_prob=0.0,
                                   attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    swin = models.SwinForAffwildClassification(cfg)
    mm = models.MultiModalTransformerForClassification(cfg)      # the 2-layer text encoder keeps its (frozen) pooler
    synth.fill_state_dict(swin, seed=100)
    for m in swin.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    swin.to(dev).train()
    mm.to(dev).train()
    return cfg, swin, mm


def _batch(dev, cfg, rank, micro):
    import bench
    args = types.SimpleNamespace(utts=1, frames=4, dtype="fp32", plm="roberta-large", input="u8", resize="pil")
    b = list(bench.synth_batch(args, dev, 10 * rank + micro, cfg))
    b[0] = b[0] % 1000
    return tuple(b)


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from facialmmt_amd.parallel import GradientAverager, broadcast_parameters
    from facialmmt_amd.train_step import GraphedTargetStep, TargetStep
    acc = 2 if mode == "hooks" else 1
    cfg, swin, mm = _build(dev, acc)
    if rank == 1:                                            # the broadcast has to repair this
        with torch.no_grad():
            for p in mm.parameters():
                p.add_(0.5)
    broadcast_parameters(mm)
    opt = torch.optim.SGD([p for p in mm.parameters() if p.requires_grad], lr=0.05)
    if mode == "hooks":
        plm = mm.roberta
        text = [p for p in list(plm.parameters()) + list(mm.text_linear.parameters()) if p.requires_grad]
        ids = set(map(id, text))
        avg = GradientAverager(None, bucket_mb=1, groups=[[p for p in mm.parameters() if p.requires_grad and id(p) not in ids], text])
        assert len(avg.buckets) >= 4
        step = TargetStep(swin, mm, opt, None, cfg, averager=avg)
        for i in range(STEPS * acc):
            step(_batch(dev, cfg, rank, i))
    else:
        avg = GradientAverager([p for p in mm.parameters() if p.requires_grad], hooks=False, bucket_mb=1)
        step = GraphedTargetStep(swin, mm, opt, None, cfg, _batch(dev, cfg, rank, 0), averager=avg)
        for i in range(STEPS):
            step(_batch(dev, cfg, rank, i))
    torch.cuda.synchronize()
    # results travel through files: a multiprocessing.Manager forked from a parent that already holds a HIP context dies
    torch.save({k: v.detach().cpu() for k, v in mm.named_parameters()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["hooks", "graphs"])
def test_two_rank_target_step_equals_mean_gradient_step(mode, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    ret = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=True) for r in range(world)]
    # single-process reference: SGD on the mean over ranks (and sum over micro-steps / accumulation) of the per-utterance gradients
    import torch.nn.functional as F
    from facialmmt_amd.train_step import select_frames
    dev = torch.device("cuda:0")
    acc = 2 if mode == "hooks" else 1
    cfg, swin, mm = _build(dev, acc)
    params = [p for p in mm.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.05)
    for s in range(STEPS):
        opt.zero_grad(set_to_none=True)
        for micro in range(acc):
            for r in range(world):
                ids, am, sep, audio, amask, vis, vmask, labels, frames, num, utt = _batch(dev, cfg, r, s * acc + micro)
                preds = swin(frames, is_trg_task=True)
                vc, nm = select_frames(preds.float(), vis, vmask, num, cfg.FacialEmoImpor_threshold)
                loss = F.cross_entropy(mm(ids, am, sep, audio, amask, vc, nm, utt).float(), labels) / acc / world
                loss.backward()
                swin.zero_grad(set_to_none=True)
        torch.nn.utils.clip_grad_norm_(params, cfg.clip)
        opt.step()
    want = {k: v.detach().cpu() for k, v in mm.named_parameters()}
    for k, w in want.items():
        for r in range(world):
            assert (ret[r][k] - w).abs().max().item() <= 2e-4 * max(1.0, w.abs().max().item()), (mode, k, r)
    init = {k: p.detach().cpu() for k, p in _build(dev, acc)[2].named_parameters()}
    assert sum(int((ret[0][k] - v).abs().max() > 0) for k, v in init.items()) > 50       # the optimiser really stepped

"""Data-parallel target step on the REAL models: two ranks (two processes sharing the one GPU of the test
box, gloo transport -- RCCL refuses two ranks per device) each run the target-task step on their own utterance with
parallel.GradientAverager, and must end with the parameters a single process gets from the mean gradient of both
utterances.  Two variants of the step:

  hooks   TargetStep + hook-driven averager: two stream groups (fusion stack / text encoder) on ONE communicator with one
          global in-order issue sequence, several buckets, gradient accumulation over two micro-steps (exchange only on the second);
  graphs  GraphedTargetStep: the step as three HIP graphs, the bucket all-reduces issued between the multimodal backward and
          the Swin backward and waited for before the optimizer graph.

The same two variants run on RCCL (backend nccl, one device per rank) where the box has two GPUs
(test_two_rank_target_step_on_rccl, skipped otherwise), and the graphed step runs its whole RCCL call path at world size 1 on
any GPU box (test_rccl_world_size_one_runs_the_exchange_path).

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


def _worker(rank, world, port, mode, out_dir, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
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
    _two_ranks(mode, tmp_path, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: two GPUs")
@pytest.mark.parametrize("mode", ["hooks", "graphs"])
def test_two_rank_target_step_on_rccl(mode, tmp_path):
    _two_ranks(mode, tmp_path, "nccl")


def _one_rank_rccl(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from facialmmt_amd.parallel import GradientAverager
    from facialmmt_amd.train_step import GraphedTargetStep
    out = {}
    for always in (False, True):
        cfg, swin, mm = _build(dev, 1)
        params = [p for p in mm.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=0.05)
        avg = GradientAverager(params, hooks=False, bucket_mb=1, comm_dtype=torch.bfloat16 if always else None, always=always)
        assert avg.active == always
        step = GraphedTargetStep(swin, mm, opt, None, cfg, _batch(dev, cfg, 0, 0), averager=avg)
        step.time_exchange(True)
        for i in range(STEPS):
            step(_batch(dev, cfg, 0, i))
        out[always] = ({k: v.detach().cpu() for k, v in mm.named_parameters()}, step.exchange_ms())
    torch.save(out, os.path.join(out_dir, "one_rank.pt"))
    dist.destroy_process_group()


def test_rccl_world_size_one_runs_the_exchange_path(tmp_path):
    """GraphedTargetStep with backend nccl (= RCCL) at world size 1 and GradientAverager(always=True): every bucket goes through
    dist.all_reduce on RCCL between the graphs (bf16 wire), the stream waits for the collectives, the optimizer graph follows --
    the parameters must equal those of the run without any collective up to the bf16 rounding of the gradients on the wire."""
    mp.spawn(_one_rank_rccl, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    out = torch.load(os.path.join(str(tmp_path), "one_rank.pt"), weights_only=False)
    (p0, _), (p1, (total_ms, exposed_ms)) = out[False], out[True]
    assert total_ms >= 0 and exposed_ms >= 0
    moved = 0
    for k, w in p0.items():
        assert (p1[k] - w).abs().max().item() <= 2e-3 * max(1.0, w.abs().max().item()), k
        moved += int((p1[k] - w).abs().max() > 0)
    assert moved > 0                                         # the bf16 wire really carried the gradients


def _aux_batch(dev, rank, step_i):
    g = torch.Generator(device=dev).manual_seed(500 + 10 * rank + step_i)
    imgs = torch.randint(0, 256, (6, 112, 112, 3), generator=g, device=dev, dtype=torch.uint8)
    return imgs, torch.randint(0, 7, (6,), generator=g, device=dev)


def _aux_worker(rank, world, port, out_dir, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    from facialmmt_amd.parallel import GradientAverager, broadcast_parameters
    from facialmmt_amd.train_step import GraphedAuxStep
    cfg, swin, _ = _build(dev, 1)
    cfg.aux_accumulation_steps = 1
    swin.swin.input_resize, swin.swin.input_dtype = "pil", torch.float32
    if rank == 1:
        with torch.no_grad():
            for p in swin.parameters():
                p.add_(0.25)
    broadcast_parameters(swin)
    params = [p for p in swin.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.02)
    avg = GradientAverager(params, hooks=False, bucket_mb=8)
    assert len(avg.buckets) >= 4                             # Swin's 46.8 M gradients: 187 MB fp32 in 8 MiB buckets
    step = GraphedAuxStep(swin, opt, None, cfg, *_aux_batch(dev, rank, 0), averager=avg)
    for i in range(STEPS):
        step(*_aux_batch(dev, rank, i))
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in swin.named_parameters()}, os.path.join(out_dir, f"aux{rank}.pt"))
    dist.destroy_process_group()


def _two_rank_aux(tmp_path, backend):
    """The auxiliary step (train.py:15-41: the step whose gradients -- Swin's 46.8 M -- ARE exchanged) on two ranks against
    single-process SGD on the mean of the two ranks' gradients; BatchNorm1d per replica, so the reference runs the batches apart."""
    import torch.nn.functional as F
    world = 2
    mp.spawn(_aux_worker, args=(world, _free_port(), str(tmp_path), backend), nprocs=world, join=True)
    ret = [torch.load(os.path.join(str(tmp_path), f"aux{r}.pt"), weights_only=True) for r in range(world)]
    dev = torch.device("cuda:0")
    cfg, swin, _ = _build(dev, 1)
    swin.swin.input_resize, swin.swin.input_dtype = "pil", torch.float32
    params = [p for p in swin.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.02)
    for s in range(STEPS):
        opt.zero_grad(set_to_none=True)
        for r in range(world):
            imgs, labels = _aux_batch(dev, r, s)
            (swin(imgs, False, labels, F.cross_entropy) / world).backward()
        torch.nn.utils.clip_grad_norm_(params, cfg.clip)
        opt.step()
    moved = 0
    for k, w in swin.named_parameters():
        w = w.detach().cpu()
        for r in range(world):
            # fp32 on the wire; what is left is summation order (all-reduced mean vs two accumulated backward passes) amplified by the head's
            # BatchNorm1d over 6 samples and by step 2 starting from step 1's parameters: measured 2.6e-4 (patch_embed.proj.bias) on MI355X
            assert (ret[r][k] - w).abs().max().item() <= 1e-3 * max(1.0, w.abs().max().item()), (k, r)
        moved += 1
    assert torch.equal(ret[0]["swin.patch_embed.proj.weight"], ret[1]["swin.patch_embed.proj.weight"]) and moved > 100


def test_two_rank_auxiliary_step_equals_mean_gradient_step(tmp_path):
    _two_rank_aux(tmp_path, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: two GPUs")
def test_two_rank_auxiliary_step_on_rccl(tmp_path):
    _two_rank_aux(tmp_path, "nccl")


def _early_text_bucket(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from facialmmt_amd.parallel import GradientAverager
    fusion = [torch.nn.Parameter(torch.zeros(1 << 16, device=dev)) for _ in range(2)]
    text = [torch.nn.Parameter(torch.zeros(1 << 16, device=dev))]
    avg = GradientAverager(None, bucket_mb=1, groups=[fusion, text], comm_dtype=torch.bfloat16, always=True, hooks=False)
    assert [b[4] for b in avg.buckets] == [0, 1]
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    want = torch.randn(1 << 16, device=dev)
    ok = True
    for _ in range(5):
        avg.zero_grad()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):                          # the text branch: a long chain of kernels, then its gradient
            b = a
            for _ in range(40):
                b = (b @ a) * 1e-2
            text[0].grad.copy_(want + 0 * b[0, 0])
            avg._hook(text[0])                                   # complete, but bucket 0 has not been issued: waits in _ready
        assert avg._ready == {1}
        for q in reversed(fusion):                               # main stream: the cascade issues bucket 0 AND the text bucket
            q.grad.fill_(1.0)
            avg._hook(q)
        avg.finish()
        torch.cuda.synchronize()
        ok = ok and torch.equal(text[0].grad, want.bfloat16().float()) and bool((fusion[0].grad == 1).all())
    torch.save({"ok": ok}, os.path.join(out_dir, "early.pt"))
    dist.destroy_process_group()


def test_text_bucket_completing_before_the_last_fusion_bucket_is_ordered_behind_its_stream(tmp_path):
    """Round-3 ADVICE: a text-encoder bucket that completes early is issued later, by the cascade inside a fusion parameter's
    hook, with the MAIN stream current.  Its gradients were written on the second stream: `_issue` must wait for the event the
    completing hook recorded there before it scales / stages / reduces the bucket (bf16 wire at world size 1 on RCCL: without
    the wait the staging copy reads the bucket before the slow side-stream chain has written it)."""
    mp.spawn(_early_text_bucket, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert torch.load(os.path.join(str(tmp_path), "early.pt"))["ok"]


def _two_ranks(mode, tmp_path, backend):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), backend), nprocs=world, join=True)
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

"""Data parallelism for the target/aux training steps: one process per GPU, torch.distributed on RCCL
(backend "nccl" on ROCm) over xGMI; gloo on CPU for the tests.

The reference has no multi-process path (LightningLite strategy='dp' on one device, main.py:152-160), so
only the maths is defined: the gradient of the mean loss over the global batch.  Utterances are
independent units: the forward path has NO collective; the only exchange is the gradient
all-reduce (mean) of the parameters that the step's optimizer updates -- the multimodal model in the
target step (Swin's target-step gradients are discarded, train.py:20,141, so they are not reduced).

`GradientAverager` is that exchange: parameters are grouped (in reverse registration order, the order backward
produces them) into ~64 MiB buckets; every `.grad` is a view into its bucket's flat fp32 buffer; a
post-accumulate hook counts arrivals and, when a bucket is complete, launches ONE asynchronous all-reduce of the
flat buffer on the stream the gradients were produced on -- so the exchange of the text encoder's buckets (produced
on the second HIP stream) and of the fusion stack's buckets overlaps with the Swin backward that is still running.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): 64 MiB buckets amortise the per-collective latency of an
8-rank ring while leaving ~27 buckets of the 1.74 GB fp32 gradient to pipeline.  `no_sync()` skips the exchange on
all but the last micro-step of an accumulation window (trg_accumulation_steps, main.py:60).

Why not torch's DistributedDataParallel wrapper for the hot step: its forward synchronises the host with the device
in each of the first 10 iterations (runtime-stat logging), and each AccumulateGrad carries ~20 us of host-side hook
work for ~800 parameters -- measured on one MI355X (world size 1): 97.6 ms/step wrapped against 82.9 ms unwrapped.
`wrap_ddp` remains for callers that prefer the stock wrapper."""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel

BUCKET_MB = 64


def wrap_ddp(module: torch.nn.Module, device=None, bucket_mb: int = BUCKET_MB) -> DistributedDataParallel:
    if not dist.is_initialized():
        raise RuntimeError("wrap_ddp: initialise torch.distributed first (backend 'nccl' = RCCL on ROCm)")
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return DistributedDataParallel(module, device_ids=ids, bucket_cap_mb=bucket_mb, gradient_as_bucket_view=True,
                                   broadcast_buffers=False, find_unused_parameters=False)


@contextlib.contextmanager
def accumulate(ddp_module, is_last_micro_step: bool):
    """no gradient exchange except on the last micro-step of an accumulation window"""
    if is_last_micro_step or not isinstance(ddp_module, (DistributedDataParallel, GradientAverager)):
        yield
    else:
        with ddp_module.no_sync():
            yield


class GradientAverager:
    """Bucketed, asynchronous gradient mean over the default process group (see the module docstring)."""

    def __init__(self, params, bucket_mb: int = BUCKET_MB, process_group=None, groups=None):
        """`groups`: optional list of parameter lists whose gradients are produced on different HIP streams (e.g. the
        text branch on the second stream): a bucket never spans two groups, so the stream that completes a bucket
        is the stream that produced all of it."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        groups = [list(params)] if groups is None else [list(g) for g in groups]
        self.params = [p for g in groups for p in g if p.requires_grad]
        if len(set(map(id, self.params))) != len(self.params):
            raise ValueError("GradientAverager: a parameter appears twice")
        self.sync = True
        self.buckets = []                                  # [flat buffer, [params], arrivals, pending work]
        cap = bucket_mb * (1 << 20)
        for g in groups:
            cur, cur_bytes = [], 0
            for p in reversed([q for q in g if q.requires_grad]):
                if p.dtype != torch.float32:
                    raise TypeError("GradientAverager: fp32 master parameters expected")
                nbytes = p.numel() * 4
                if cur and (cur_bytes + nbytes > cap or cur[0].device != p.device):
                    self._close(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                self._close(cur)
        self._of = {}
        for bi, (_, ps, _, _) in enumerate(self.buckets):
            for p in ps:
                self._of[p] = bi
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _close(self, ps):
        # 16-byte aligned slots so that every gradient view is as aligned as a stand-alone tensor
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        for p, o in zip(ps, offs):
            p.grad = flat[o:o + p.numel()].view_as(p)
        self.buckets.append([flat, ps, 0, None])

    def _hook(self, p):
        b = self.buckets[self._of[p]]
        b[2] += 1
        if b[2] == len(b[1]):
            b[2] = 0
            if self.sync and self.world > 1:
                b[0].div_(self.world)
                b[3] = dist.all_reduce(b[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """make the current stream wait for every outstanding exchange (call before clipping / the optimizer)"""
        for b in self.buckets:
            if b[3] is not None:
                b[3].wait()
                b[3] = None

    def zero_grad(self):
        """zero the flat buffers in place (the .grad views must survive: never zero_grad(set_to_none=True) these)"""
        for b in self.buckets:
            b[0].zero_()
            b[2] = 0

    @contextlib.contextmanager
    def no_sync(self):
        old, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = old

    def remove(self):
        for h in self._handles:
            h.remove()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """rank `src`'s parameters and buffers to every rank (what DistributedDataParallel does at construction)"""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src, group=process_group)


def shard_utterances(n_global: int, rank: int, world: int):
    """contiguous split of a global batch of utterances across ranks"""
    per = (n_global + world - 1) // world
    return range(min(rank * per, n_global), min((rank + 1) * per, n_global))

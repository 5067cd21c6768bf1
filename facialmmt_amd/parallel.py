"""Data parallelism for the target/aux training steps: one process per GPU, torch.distributed on RCCL
(backend "nccl" on ROCm) over xGMI; gloo on CPU for the tests.

The reference has no multi-process path (LightningLite strategy='dp' on one device, main.py:152-160), so
only the maths is defined: the gradient of the mean loss over the global batch.  Utterances are
independent units: the forward path has NO collective; the only exchange is the gradient
all-reduce (mean) of the parameters that the step's optimizer updates -- the multimodal model in the
target step (Swin's target-step gradients are discarded, train.py:20,141, so they are not reduced).

Bucketing: gradients are reduced in ~64 MiB buckets in reverse parameter order, overlapped with the rest
of backward (DistributedDataParallel's reducer).  xGMI is point-to-point (7 links x ~153 GB/s per GPU):
larger buckets amortise the per-collective latency of an 8-rank ring while still leaving >= 25 buckets of
the 1.74 GB fp32 gradient to overlap with the Swin backward that follows the multimodal backward.
`accumulate()` skips the exchange on all but the last micro-step (trg_accumulation_steps, main.py:60)."""
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
    if is_last_micro_step or not isinstance(ddp_module, DistributedDataParallel):
        yield
    else:
        with ddp_module.no_sync():
            yield


def shard_utterances(n_global: int, rank: int, world: int):
    """contiguous split of a global batch of utterances across ranks"""
    per = (n_global + world - 1) // world
    return range(min(rank * per, n_global), min((rank + 1) * per, n_global))

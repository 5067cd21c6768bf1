"""Data parallelism for the target/aux training steps: one process per GPU, torch.distributed on RCCL
(backend "nccl" on ROCm) over xGMI; gloo on CPU for the tests.

The reference has no multi-process path (LightningLite strategy='dp' on one device, main.py:152-160), so
only the maths is defined: the gradient of the mean loss over the global batch.  Utterances are
independent units: the forward path has NO collective; the only exchange is the gradient
all-reduce (mean) of the parameters that the step's optimizer updates -- the multimodal model in the
target step (Swin's target-step gradients are discarded, train.py:20,141, so they are not reduced).

`GradientAverager` is that exchange: parameters are grouped (in reverse registration order, the order backward
produces them) into ~64 MiB buckets; every `.grad` is a view into its bucket's flat fp32 buffer; a
post-accumulate hook counts arrivals and, when a bucket is complete (and every lower-numbered bucket has been issued),
launches ONE asynchronous all-reduce of the flat buffer, ordered behind the stream that produced the gradients -- so the
exchange of the fusion stack's and the text encoder's buckets overlaps with the Swin backward that is still running.
The graphed step (train_step.GraphedTargetStep) has no hooks: it issues all buckets between its multimodal-backward
graph and its Swin-backward graph (`exchange_begin` / `exchange_end`).
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
    """Bucketed, asynchronous gradient mean over the default process group (see the module docstring).

    Ordering contract (RCCL, like NCCL, needs every rank to issue the collectives of a communicator in the same order,
    and concurrent collectives on DIFFERENT communicators can deadlock when their kernels cannot be co-scheduled or the
    ranks interleave them differently): there is ONE communicator -- the caller's `process_group` -- and ONE global issue
    order, the construction order of the buckets (group after group, each group in reverse registration order).  A bucket's
    all-reduce is issued only after every lower-numbered bucket has been issued; a bucket that completes early waits in
    `_ready`.  List `groups` in the order their gradients complete in the backward (fusion stack before text encoder):
    the order is a property of the model, never of a rank's timing.
    Completeness contract (what torch's DDP enforces with an error): when `finish()` is called with the exchange
    enabled, every bucket must have received every one of its gradients.  A parameter that took no part in the step
    (e.g. an unused pooler) leaves its bucket incomplete; `finish()` raises instead of letting the ranks step on
    un-averaged gradients."""

    def __init__(self, params, bucket_mb: int = BUCKET_MB, process_group=None, groups=None, comm_dtype=None, hooks: bool = True, always: bool = False):
        """`groups`: optional list of parameter lists whose gradients are produced on different HIP streams (e.g. the
        text branch on the second stream): a bucket never spans two groups, so the stream that completes a bucket
        is the stream that produced all of it.
        `comm_dtype`: torch.bfloat16 halves the bytes on the wire (1.74 GB -> 0.87 GB for the multimodal model): the
        flat fp32 bucket is rounded into a bf16 staging buffer, summed over ranks in bf16, and written back as fp32.
        `always`: issue the collectives even at world size 1 (a one-GPU box then exercises the whole RCCL call path)."""
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (always and dist.is_initialized())
        groups = [list(params)] if groups is None else [list(g) for g in groups]
        self.params = [p for g in groups for p in g if p.requires_grad]
        if len(set(map(id, self.params))) != len(self.params):
            raise ValueError("GradientAverager: a parameter appears twice")
        if comm_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("GradientAverager: comm_dtype must be None / torch.float32 / torch.bfloat16")
        self.comm_dtype = None if comm_dtype == torch.float32 else comm_dtype
        self.sync = True
        self.comm = process_group                          # the one communicator of every bucket (see the ordering contract)
        self.buckets = []                                  # [flat buffer, [params], arrivals, pending work, group, staging, completion event]
        self._next = 0                                     # the next bucket (global construction order) to issue
        self._ready = set()                                # complete buckets waiting for a lower-numbered one
        self._issued = set()
        cap = bucket_mb * (1 << 20)
        for gi, g in enumerate(groups):
            cur, cur_bytes = [], 0
            for p in reversed([q for q in g if q.requires_grad]):
                if p.dtype != torch.float32:
                    raise TypeError("GradientAverager: fp32 master parameters expected")
                nbytes = p.numel() * 4
                if cur and (cur_bytes + nbytes > cap or cur[0].device != p.device):
                    self._close(cur, gi)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                self._close(cur, gi)
        self._of = {}
        for bi, b in enumerate(self.buckets):
            for p in b[1]:
                self._of[p] = bi
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params] if hooks else []

    def _close(self, ps, gi):
        # 16-byte aligned slots so that every gradient view is as aligned as a stand-alone tensor
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        for p, o in zip(ps, offs):
            p.grad = flat[o:o + p.numel()].view_as(p)
        stage = torch.empty(n, dtype=self.comm_dtype, device=flat.device) if self.comm_dtype is not None else None
        self.buckets.append([flat, ps, 0, None, gi, stage, None])

    def _issue(self, bi):
        b = self.buckets[bi]
        # A bucket held back in `_ready` is issued later by the cascade inside ANOTHER group's hook, i.e. with another stream
        # current than the one that wrote its gradients (text encoder on the second stream, fusion stack on the main one):
        # the scaling, the staging copy and the all-reduce must be ordered behind the producing stream, not just the current one.
        if b[6] is not None:
            torch.cuda.current_stream(b[0].device).wait_event(b[6])
            b[6] = None
        b[0].div_(self.world)
        buf = b[0]
        if b[5] is not None:
            b[5].copy_(b[0])
            buf = b[5]
        b[3] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.comm, async_op=True)
        self._issued.add(bi)

    def _hook(self, p):
        bi = self._of[p]
        b = self.buckets[bi]
        b[2] += 1
        if b[2] < len(b[1]):
            return
        b[2] = 0                                           # bucket complete
        if not (self.sync and self.active):
            return
        if bi in self._issued or bi in self._ready:
            raise RuntimeError("GradientAverager: a bucket completed twice in one exchange window (call finish() once per step)")
        if b[0].is_cuda:                                   # the stream this hook runs on is the one that produced the bucket's last gradient
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(b[0].device))
            b[6] = ev
        self._ready.add(bi)
        while self._next < len(self.buckets) and self._next in self._ready:                 # in-order issue, cascade
            self._ready.discard(self._next)
            self._issue(self._next)
            self._next += 1

    def finish(self):
        """Make the current stream wait for every outstanding exchange (call before clipping / the optimizer).
        Raises if the exchange is enabled and some bucket is incomplete: one of its parameters received no gradient in
        this step (freeze it or leave it out of the averager), so the ranks would otherwise diverge silently."""
        if self.sync and self.active:
            bad = [bi for bi, b in enumerate(self.buckets) if bi not in self._issued]
            if bad:
                names = []
                for bi in bad[:4]:
                    b = self.buckets[bi]
                    names.append(f"bucket {bi} (group {b[4]}): {b[2]}/{len(b[1])} gradients arrived" +
                                 (" (complete, held back by a lower-numbered bucket)" if bi in self._ready else ""))
                self._wait_all()                            # nothing stays in flight behind the exception
                self._reset_window()
                raise RuntimeError("GradientAverager.finish(): gradient exchange incomplete -- " + "; ".join(names) +
                                   ".  A parameter that gets no gradient (unused sub-module, e.g. a PLM pooler) must be "
                                   "frozen (requires_grad_(False)) or excluded.  Nothing was stepped; the buckets that had been "
                                   "issued hold averaged gradients, the others local ones: call zero_grad() before the next step.")
        self._wait_all()
        self._reset_window()

    def _wait_all(self):
        for b in self.buckets:
            if b[3] is not None:
                b[3].wait()
                b[3] = None
                if b[5] is not None:
                    b[0].copy_(b[5])

    def exchange_all(self):
        """Average every bucket over the ranks now, in bucket order, on the current stream (blocking semantics of the
        stream: later work on it sees the averaged gradients).  For owners without hooks; no-op at world size 1."""
        self.exchange_begin()
        self.exchange_end()

    def exchange_begin(self):
        """Issue every bucket's all-reduce (bucket order, asynchronous): the collectives run on the backend's own stream behind
        whatever the current stream has been given so far, and beside whatever it is given next -- the graphed step calls this
        between its two forward/backward graphs, so the exchange of the multimodal gradients overlaps the Swin backward."""
        if not (self.sync and self.active):
            return
        for bi in range(len(self.buckets)):
            self._issue(bi)

    def exchange_end(self):
        """make the current stream wait for the collectives exchange_begin() issued"""
        if not (self.sync and self.active):
            return
        self._wait_all()
        self._reset_window()

    def _reset_window(self):
        self._next = 0
        self._ready.clear()
        self._issued.clear()
        for b in self.buckets:
            b[2] = 0
            b[3] = None
            b[6] = None

    def zero_grad(self):
        """zero the flat buffers in place (the .grad views must survive: never zero_grad(set_to_none=True) these)"""
        for b in self.buckets:
            b[0].zero_()
            b[2] = 0

    @contextlib.contextmanager
    def no_sync(self):
        """accumulate locally: arrivals of this micro-step are not counted towards the exchange window"""
        old, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = old
            for b in self.buckets:
                b[2] = 0

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

"""Data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI (torch.distributed backend
"nccl" on ROCm), replacing the reference's Accelerate/DDP wrap (train_accel_gpu.py:93) and torchmultimodal's
per-loss-term all-gathers (utils/distributed.py:23-56).

Two exchanges per step:
  A. ONE all-gather of the pooled block (b, R, D) fp32 plus the per-sample modality-presence bits, instead of
     2 x (4..60) separate (b, 512) gathers.  Every rank then evaluates the full (B x B) similarity blocks and
     takes the gradient of sum_r loss_r w.r.t. its OWN rows directly (loss.hip), so the reference's backward
     reduce-scatter of d(pooled_all) disappears.
  B. all-reduce(mean) of the flat gradient buffer in contiguous buckets ordered by backward completion
     (pool -> layer L-1 -> ... -> layer 0 -> encoders); each bucket is launched as soon as the backward of its
     layers has been enqueued, so RCCL overlaps with the remaining backward kernels.  xGMI is point-to-point:
     with 69.7 MB of fp32 gradients a ring moves ~122 MB through each link (~0.8 ms at 153 GB/s); 7 buckets
     of ~10 MB keep that off the critical path.

DDP semantics kept: gradients are averaged over ranks; the loss labels of rank r are r*b + arange(b)
(utils/contrastive_loss_with_temperature.py:28-31), which requires the same local batch on every rank.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def gather_pooled(pooled: torch.Tensor, present: torch.Tensor, group=None, bufs=None, cut=None) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """pooled (b, R, D) fp32, present (b,) int32 -> (pooled_all (B, R, D), present_all (B,), row0).
    One collective: the presence bits ride in the same message as one extra fp32 column block.
    bufs: (msg, out) static buffers (a replayed step needs fixed addresses); cut: runs the collective itself
    (graph.GraphedStep ends a graph segment there, so the collective stays an eager call between two replayed segments)."""
    W = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, R, D = pooled.shape
    if bufs is None:
        bufs = (torch.empty(b, R * D + 1, dtype=torch.float32, device=pooled.device),
                torch.empty(W * b, R * D + 1, dtype=torch.float32, device=pooled.device))
    msg, out = bufs
    msg[:, : R * D].copy_(pooled.reshape(b, R * D))
    msg[:, R * D].copy_(present)                          # int32 -> fp32, <= 2^24: exact

    def collective():
        if dist.get_backend(group) == "gloo":          # CPU tests / single-GPU rehearsal: gloo has no flat all-gather on device tensors
            dist.all_gather(list(out.view(W, b, R * D + 1).unbind(0)), msg, group=group)
        else:
            dist.all_gather_into_tensor(out, msg, group=group)

    (cut or (lambda fn: fn()))(collective)
    pooled_all = out[:, : R * D].reshape(W * b, R, D).contiguous()
    present_all = out[:, R * D].to(torch.int32).contiguous()
    return pooled_all, present_all, rank * b


class BucketReducer:
    """Averages a flat gradient buffer over ranks, bucket by bucket, asynchronously.

    The mean is pre-scale + SUM on every backend: ``ReduceOp.AVG`` exists on RCCL and saves one small kernel per bucket, but this
    pool has no multi-GPU box to check it on, so it is opt-in (``MCA_DP_AVG=1``)."""

    def __init__(self, flat_grads: torch.Tensor, group=None, always: bool = False):
        import os
        self.flat = flat_grads
        self.group = group
        self.world = dist.get_world_size(group)
        self.always = always                    # rehearsal on one rank: issue the collectives anyway
        self.avg_op = dist.get_backend(group) == "nccl" and os.environ.get("MCA_DP_AVG") == "1"
        self.pending: List = []
        self.cut = None                         # set by graph.GraphedStep while it captures / replays segments

    def bucket_ready(self, lo: int, hi: int):
        if hi <= lo or (self.world == 1 and not self.always):
            return
        chunk = self.flat[lo:hi]
        if not self.avg_op and self.world > 1:
            chunk.div_(self.world)                                     # sum of means = mean (inside the graph segment when replayed)
        op = dist.ReduceOp.AVG if self.avg_op else dist.ReduceOp.SUM

        def collective():
            self.pending.append(dist.all_reduce(chunk, op=op, group=self.group, async_op=True))

        (self.cut or (lambda fn: fn()))(collective)

    def finish(self, extra=None):
        """wait for every bucket; extra: one more collective issued behind them (the finite flag)"""
        def collective():
            for w in self.pending:
                w.wait()
            self.pending = []
            if extra is not None:
                extra()

        (self.cut or (lambda fn: fn()))(collective)


class DataParallelMCA:
    """Wraps an ``MCA`` model living on this rank's GPU.  Usage mirrors the reference loop:
        out = dp(batch); opt.zero_grad(); out['loss'].backward(); dp.finish_backward(); clip; opt.step()
    ``always_collect``: issue the collectives even on a world of one rank (rehearsal of the RCCL path on a one-GPU box).
    """

    def __init__(self, model, group=None, broadcast_weights: bool = True, always_collect: bool = False):
        self.model = model
        self.group = group
        self.always = always_collect
        self.world = dist.get_world_size(group)
        self._gather_bufs = {}
        self.cut = None
        eng = model.engine
        if broadcast_weights and self.world > 1:
            dist.broadcast(eng.flat, src=0, group=group)               # same initial weights on every rank
            eng.invalidate_weights()
        model._dp_wrapper = self          # MCA.engine re-installs the hooks if .to() / .float() rebuilds the engine
        self.install(eng)

    def install(self, eng):
        self.reducer = BucketReducer(eng.gflat, self.group, always=self.always)
        self.reducer.cut = self.cut
        eng.gather_hook = self._gather if (self.world > 1 or self.always) else None
        eng.grad_bucket_hook = self.reducer.bucket_ready

    def set_cut(self, cut):
        """cut(fn): how a collective is issued (None = call it); graph.GraphedStep installs its segment cutter here"""
        self.cut = cut
        self.reducer.cut = cut

    def _gather(self, pooled, present):
        key = tuple(pooled.shape)
        if key not in self._gather_bufs:
            b, R, D = key
            self._gather_bufs[key] = (torch.empty(b, R * D + 1, dtype=torch.float32, device=pooled.device),
                                      torch.empty(self.world * b, R * D + 1, dtype=torch.float32, device=pooled.device))
        return gather_pooled(pooled, present, self.group, self._gather_bufs[key], self.cut)

    def __call__(self, batch, no_loss: bool = False):
        return self.model(batch, no_loss=no_loss)

    def finish_backward(self):
        """every bucket's all-reduce has landed when the current stream passes this point; the finite flag is MAX-reduced behind
        them, so that a step one rank flagged is skipped (fused AdamW) or raised (poll_finite) on EVERY rank, not on that one"""
        eng = self.model.engine
        flag = None
        if eng.check_finite and (self.world > 1 or self.always):
            def flag():
                from .hip import call, ptr, stream_ptr
                dist.all_reduce(eng.finite_flag, op=dist.ReduceOp.MAX, group=self.group)
                call("mca_flag_to_host", ptr(eng.finite_flag), eng._flag_host.data_ptr(), stream_ptr())          # the reduced word, for poll_finite
                if not torch.cuda.is_current_stream_capturing():
                    eng._flag_event = torch.cuda.Event()
                    eng._flag_event.record()
        self.reducer.finish(flag)

    def parameters(self):
        return self.model.parameters()

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


def gather_pooled(pooled: torch.Tensor, present: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """pooled (b, R, D) fp32, present (b,) int32 -> (pooled_all (B, R, D), present_all (B,), row0).
    One collective: the presence bits ride in the same message as one extra fp32 column block."""
    W = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, R, D = pooled.shape
    msg = torch.empty(b, R * D + 1, dtype=torch.float32, device=pooled.device)
    msg[:, : R * D] = pooled.reshape(b, R * D)
    msg[:, R * D] = present.to(torch.float32)             # <= 2^24: exact in fp32
    out = torch.empty(W * b, R * D + 1, dtype=torch.float32, device=pooled.device)
    if dist.get_backend(group) == "gloo":          # CPU tests / single-GPU rehearsal: gloo has no flat all-gather on device tensors
        parts = list(out.view(W, b, R * D + 1).unbind(0))
        dist.all_gather(parts, msg, group=group)
    else:
        dist.all_gather_into_tensor(out, msg, group=group)
    pooled_all = out[:, : R * D].reshape(W * b, R, D).contiguous()
    present_all = out[:, R * D].to(torch.int32).contiguous()
    return pooled_all, present_all, rank * b


class BucketReducer:
    """Averages a flat gradient buffer over ranks, bucket by bucket, asynchronously."""

    def __init__(self, flat_grads: torch.Tensor, group=None):
        self.flat = flat_grads
        self.group = group
        self.world = dist.get_world_size(group)
        self.avg_op = dist.get_backend(group) == "nccl"
        self.pending: List = []

    def bucket_ready(self, lo: int, hi: int):
        if hi <= lo or self.world == 1:
            return
        chunk = self.flat[lo:hi]
        if self.avg_op:                                                # RCCL: the mean is taken inside the collective
            self.pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:                                                          # gloo has no AVG: pre-scale, sum of means = mean
            chunk.div_(self.world)
            self.pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending = []


class DataParallelMCA:
    """Wraps an ``MCA`` model living on this rank's GPU.  Usage mirrors the reference loop:
        out = dp(batch); opt.zero_grad(); out['loss'].backward(); dp.finish_backward(); clip; opt.step()
    """

    def __init__(self, model, group=None, broadcast_weights: bool = True):
        self.model = model
        self.group = group
        eng = model.engine
        if broadcast_weights and dist.get_world_size(group) > 1:
            dist.broadcast(eng.flat, src=0, group=group)               # same initial weights on every rank
            eng.invalidate_weights()
        model._dp_wrapper = self          # MCA.engine re-installs the hooks if .to() / .float() rebuilds the engine
        self.install(eng)

    def install(self, eng):
        self.reducer = BucketReducer(eng.gflat, self.group)
        eng.gather_hook = lambda pooled, present: gather_pooled(pooled, present, self.group)
        eng.grad_bucket_hook = self.reducer.bucket_ready

    def __call__(self, batch, no_loss: bool = False):
        return self.model(batch, no_loss=no_loss)

    def finish_backward(self):
        self.reducer.finish()

    def parameters(self):
        return self.model.parameters()

"""Embedding-space metrics of the reference's eval loop (utils/metrics.py:20-70, Wang & Isola 2020): alignment of
positive pairs and uniformity on the hypersphere, accumulated over batches like the reference's torchmetrics classes
(``update`` / ``compute(norm=...)`` / ``reset``).  Plain torch; evaluation only, not on the timed step."""
from __future__ import annotations

from typing import List

import torch
from torch.nn.functional import normalize


def lalign(x: torch.Tensor, y: torch.Tensor, alpha: float = 2, norm: bool = True) -> torch.Tensor:
    if norm:
        x, y = normalize(x), normalize(y)
    return (x - y).norm(dim=1).pow(alpha).mean()


def lunif(x: torch.Tensor, t: float = 2, norm: bool = True) -> torch.Tensor:
    if norm:
        x = normalize(x)
    return torch.pdist(x, p=2).pow(2).mul(-t).exp().mean().log()


def _gather_cat(parts: List[torch.Tensor], group=None) -> torch.Tensor:
    """this rank's accumulated rows, then every other rank's, rank-major: what torchmetrics does for a list state with
    ``dist_reduce_fx="cat"`` when ``compute()`` synchronises (utils/metrics.py:40-41,60)"""
    import torch.distributed as dist
    local = torch.cat(parts) if parts else torch.zeros(0)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, local, group=group)
    return torch.cat([o for o in out if o.numel()])


class Alignment:
    def __init__(self, alpha: float = 2):
        self.alpha = alpha
        self.preds: List[torch.Tensor] = []
        self.target: List[torch.Tensor] = []

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        if preds.shape != target.shape:
            raise ValueError("preds and target must have the same shape")
        self.preds.append(preds.detach().float().cpu())
        self.target.append(target.detach().float().cpu())

    def compute(self, norm: bool = False, sync: bool = False, group=None) -> torch.Tensor:
        """sync: gather every rank's rows first (a collective: every rank must call it), as the reference's torchmetrics do"""
        if sync:
            return lalign(_gather_cat(self.preds, group), _gather_cat(self.target, group), self.alpha, norm)
        return lalign(torch.cat(self.preds), torch.cat(self.target), self.alpha, norm)

    def reset(self):
        self.preds, self.target = [], []


class Uniformity:
    def __init__(self, t: float = 2):
        self.t = t
        self.preds: List[torch.Tensor] = []

    def update(self, preds: torch.Tensor):
        self.preds.append(preds.detach().float().cpu())

    def compute(self, norm: bool = False, sync: bool = False, group=None) -> torch.Tensor:
        if sync:
            return lunif(_gather_cat(self.preds, group), self.t, norm)
        return lunif(torch.cat(self.preds), self.t, norm)

    def reset(self):
        self.preds = []

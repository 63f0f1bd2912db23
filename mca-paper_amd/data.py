"""Synthetic batches in the exact layouts the reference's collators emit.

  embedded_sequence: ``tokens (b, pad_len, in) f32`` zero-filled past the valid length,
                     ``attention_mask (b, pad_len) bool`` True = pad; dropped modality = all-pad row.
  sequence / tabular: ``values (b, n) f32`` with -10000 for a dropped modality,
                     ``attention_mask (b, n) int64`` = (values == -10000).
"""
from __future__ import annotations

from typing import Dict

import torch


def synthetic_batch(model_config: dict, batch_size: int, seed: int = 1234, p_drop: float = 0.0,
                    lengths: str = "uniform", device: str = "cpu") -> Dict[str, Dict[str, torch.Tensor]]:
    """lengths: "uniform" -> valid length ~ U{1..pad_len}; "full" -> every sequence fills pad_len.
    At least one modality is kept per sample."""
    g = torch.Generator().manual_seed(seed)
    enc = model_config["encoder_configs"]
    names = list(enc.keys())
    drop = torch.rand(batch_size, len(names), generator=g) < p_drop
    for i in range(batch_size):
        if drop[i].all():
            drop[i, int(torch.randint(0, len(names), (1,), generator=g))] = False
    batch = {}
    for mi, name in enumerate(names):
        c = enc[name]
        n = c["max_tokens"]
        if c["type"] == "EmbeddedSequenceEncoder":
            width = c["input_size"]
            if lengths == "full":
                ln = torch.full((batch_size,), n, dtype=torch.long)
            else:
                ln = torch.randint(1, n + 1, (batch_size,), generator=g)
            ln = torch.where(drop[:, mi], torch.zeros_like(ln), ln)
            mask = torch.arange(n)[None, :] >= ln[:, None]
            toks = torch.randn(batch_size, n, width, generator=g)
            toks = toks.masked_fill(mask[..., None], 0.0)
            batch[name] = {"tokens": toks.to(device), "attention_mask": mask.to(device)}
        elif c["type"] == "TabularEncoder":
            vals = torch.randn(batch_size, n, generator=g)
            vals = torch.where(drop[:, mi, None], torch.full_like(vals, -10000.0), vals)
            batch[name] = {"values": vals.to(device), "attention_mask": (vals == -10000).to(torch.long).to(device)}
        else:
            raise NotImplementedError(c["type"])
    return batch

"""Checkpoints in the reference's layout.

The reference saves through Accelerate: ``accelerator.save_model(model, dir, safe_serialization=True)`` writes
``<dir>/model.safetensors`` keyed by ``model.state_dict()`` names (train_accel_gpu.py:187) and ``save_state`` writes the
same file plus optimizer / scheduler / RNG blobs (:122-123,133-134).  Because this model keeps the reference's module
tree, those files load directly; ``load_model`` accepts a directory or a file (.safetensors / torch pickle).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

_NON_PERSISTENT = {"return_token_types_tensor"}


def _state(model) -> Dict[str, torch.Tensor]:
    return {k: v.detach().clone().contiguous().cpu() for k, v in model.state_dict().items()}


def save_model(model, output_dir: str, safe_serialization: bool = True) -> str:
    os.makedirs(output_dir, exist_ok=True)
    sd = _state(model)
    if safe_serialization:
        from safetensors.torch import save_file
        path = os.path.join(output_dir, "model.safetensors")
        # safetensors has no bool/long restrictions for these dtypes; shared storage is not an issue after clone()
        save_file(sd, path)
    else:
        path = os.path.join(output_dir, "pytorch_model.bin")
        torch.save(sd, path)
    return path


def _read(path: str) -> Dict[str, torch.Tensor]:
    if os.path.isdir(path):
        for name in ("model.safetensors", "pytorch_model.bin", "state.pt"):
            p = os.path.join(path, name)
            if os.path.exists(p):
                path = p
                break
        else:
            raise FileNotFoundError(f"no model file in {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    obj = torch.load(path, map_location="cpu")
    return obj["model"] if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict) else obj


def load_model(model, path: str, strict: bool = True):
    """Load a reference-format checkpoint.  Static buffers (masks, token types, positional tables) must agree with what
    this model derived from its config — a mismatch means the checkpoint belongs to a different configuration."""
    sd = _read(path)
    own = model.state_dict()
    for k in ("attn_mask", "pool_mask", "token_types"):
        if k in sd and k in own and not torch.equal(sd[k].cpu(), own[k].cpu()):
            raise ValueError(f"checkpoint buffer '{k}' does not match the model built from this config")
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own and k not in _NON_PERSISTENT]
    if strict and (missing or unexpected):
        raise KeyError(f"checkpoint mismatch: missing {missing[:5]} unexpected {unexpected[:5]}")
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return missing, unexpected


def save_state(output_dir: str, model, optimizer=None, step: int = 0, extra: Optional[dict] = None):
    save_model(model, output_dir)
    if optimizer is not None:
        torch.save(optimizer.state_dict(), os.path.join(output_dir, "optimizer.pt"))
    with open(os.path.join(output_dir, "meta.json"), "w") as f:
        json.dump({"step": int(step), **(extra or {})}, f)


def load_state(input_dir: str, model, optimizer=None) -> dict:
    load_model(model, input_dir, strict=False)
    p = os.path.join(input_dir, "optimizer.pt")
    if optimizer is not None and os.path.exists(p):
        optimizer.load_state_dict(torch.load(p, map_location=next(model.parameters()).device))
    m = os.path.join(input_dir, "meta.json")
    return json.load(open(m)) if os.path.exists(m) else {}

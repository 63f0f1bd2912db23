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


def _optimizer_bin(model, optimizer, step: int, lr: Optional[float] = None) -> dict:
    """``torch.optim.AdamW.state_dict()`` as Accelerate pickles it into optimizer.bin: state index i = i-th entry of
    ``model.parameters()`` (the module tree, hence that order, is the reference's)."""
    eng = model.engine
    state = {}
    off = {id(p): o for p, o in zip(eng.param_order, eng.param_offsets)}
    params = list(model.parameters())
    for i, p in enumerate(params):
        o, n = off[id(p)], p.numel()
        state[i] = {"step": torch.tensor(float(optimizer.step_count)),
                    "exp_avg": optimizer.exp_avg[o:o + n].view(p.shape).detach().cpu().clone(),
                    "exp_avg_sq": optimizer.exp_avg_sq[o:o + n].view(p.shape).detach().cpu().clone()}
    g = optimizer.param_groups[0]
    group = {"lr": g["lr"] if lr is None else lr, "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
             "foreach": None, "maximize": False, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(params)))}
    if "initial_lr" in g:
        group["initial_lr"] = g["initial_lr"]
    return {"state": state, "param_groups": [group]}


def _load_optimizer_bin(model, optimizer, sd: dict):
    """optimizer.bin written by the reference's ``accelerator.save_state`` (torch AdamW over ``model.parameters()``) ->
    the flat moment buffers of FusedAdamW."""
    eng = model.engine
    off = {id(p): o for p, o in zip(eng.param_order, eng.param_offsets)}
    params = list(model.parameters())
    st = sd["state"]
    if len(st) not in (0, len(params)):
        raise ValueError(f"optimizer state holds {len(st)} tensors, the model has {len(params)} parameters")
    steps = []
    for i, p in enumerate(params):
        e = st.get(i, st.get(str(i)))
        if e is None:
            continue
        o, n = off[id(p)], p.numel()
        if tuple(e["exp_avg"].shape) != tuple(p.shape):
            raise ValueError(f"optimizer state {i}: shape {tuple(e['exp_avg'].shape)} != parameter {tuple(p.shape)}")
        optimizer.exp_avg[o:o + n].copy_(e["exp_avg"].reshape(-1).to(optimizer.exp_avg.device, torch.float32))
        optimizer.exp_avg_sq[o:o + n].copy_(e["exp_avg_sq"].reshape(-1).to(optimizer.exp_avg.device, torch.float32))
        steps.append(int(float(e["step"])))
    if steps:
        optimizer.step_count = max(steps)
    g = sd["param_groups"][0]
    for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
        if k in g:
            optimizer.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]


def save_state(output_dir: str, model, optimizer=None, step: int = 0, extra: Optional[dict] = None, sched_stride: int = 1,
               next_lr: Optional[float] = None):
    """``accelerator.save_state(output_dir)`` layout (train_accel_gpu.py:122-123,133-134): model.safetensors +
    optimizer.bin (torch AdamW state_dict) + scheduler.bin (LambdaLR state: ``last_epoch`` = scheduler steps taken) +
    meta.json (ours: optimizer step count).  RNG blobs are not written: nothing in the native step draws random numbers.
    ``next_lr``: torch's schedulers leave the learning rate of the NEXT step in ``param_groups[*]['lr']`` / ``_last_lr``
    (``base * lambda(last_epoch)``), and that is what Accelerate pickles; the native loop sets the rate right before each step,
    so the caller passes the next step's rate (train_accel_gpu.py does); None keeps the rate of the step just taken."""
    save_model(model, output_dir)
    if optimizer is not None:
        lr = optimizer.param_groups[0]["lr"] if next_lr is None else float(next_lr)
        torch.save(_optimizer_bin(model, optimizer, step, lr), os.path.join(output_dir, "optimizer.bin"))
        torch.save({"base_lrs": [optimizer.defaults["lr"]], "last_epoch": int(step) * sched_stride, "verbose": False,
                    "_step_count": int(step) * sched_stride + 1, "_get_lr_called_within_step": False, "_last_lr": [lr],
                    "lr_lambdas": [None]}, os.path.join(output_dir, "scheduler.bin"))
    with open(os.path.join(output_dir, "meta.json"), "w") as f:
        json.dump({"step": int(step), "sched_stride": int(sched_stride), **(extra or {})}, f)


def load_state(input_dir: str, model, optimizer=None) -> dict:
    """``accelerator.load_state(input_dir)`` (train_accel_gpu.py:97-99) for a directory written by the reference or by
    ``save_state`` above.  -> meta dict: ``step`` (optimizer steps taken), ``scheduler_last_epoch`` when a scheduler.bin is
    present, ``warnings`` (list of strings: keys that did not match, state files that were ignored)."""
    warnings = []
    missing, unexpected = load_model(model, input_dir, strict=False)
    if missing or unexpected:
        warnings.append(f"load_state: checkpoint keys do not match the model: missing {missing[:8]} unexpected {unexpected[:8]}")
    meta = {}
    m = os.path.join(input_dir, "meta.json")
    if os.path.exists(m):
        meta = json.load(open(m))
    if optimizer is not None:
        pb, pt = os.path.join(input_dir, "optimizer.bin"), os.path.join(input_dir, "optimizer.pt")
        if os.path.exists(pb):
            _load_optimizer_bin(model, optimizer, torch.load(pb, map_location="cpu", weights_only=True))
            meta.setdefault("step", optimizer.step_count)
        elif os.path.exists(pt):                    # round-1 layout of this repo
            optimizer.load_state_dict(torch.load(pt, map_location=next(model.parameters()).device, weights_only=True))
        else:
            warnings.append(f"load_state: no optimizer state in {input_dir}: Adam moments start from zero")
    sb = os.path.join(input_dir, "scheduler.bin")
    if os.path.exists(sb):
        meta["scheduler_last_epoch"] = int(torch.load(sb, map_location="cpu", weights_only=True).get("last_epoch", 0))
    others = [f for f in os.listdir(input_dir) if f.startswith("random_states") or f.endswith(".pkl")] if os.path.isdir(input_dir) else []
    if others:
        warnings.append(f"load_state: ignored {others} (the native step draws no random numbers; data order is reseeded per epoch)")
    meta["warnings"] = warnings
    return meta

"""Fused clip_grad_norm_ + AdamW over the engine's flat parameter / gradient buffers.

Reference step (train_accel_gpu.py:116-119): ``accelerator.clip_grad_norm_(model.parameters(), clip)``
then ``AdamW(model.parameters(), lr).step()`` with torch defaults (betas (0.9, 0.999), eps 1e-8,
weight_decay 0.01 on every tensor).  Here: one squared-norm reduction kernel + one AdamW kernel over the
single flat buffer; the clip coefficient is read on the device, so the step has no host sync.
``FusedAdamW`` subclasses ``torch.optim.Optimizer`` so LR schedulers (``param_groups[0]['lr']``) work.
"""
from __future__ import annotations

import math

import torch

from .hip import call, ptr, stream_ptr


def clip_grad_norm_(model, max_norm: float) -> torch.Tensor:
    """Records the clipping request and returns the total gradient L2 norm (device tensor, no sync).  The
    scaling itself is applied inside the next ``FusedAdamW.step()`` (gradients in memory stay unscaled)."""
    eng = model.engine
    sq = eng._ws.setdefault("sqnorm", torch.zeros(1, dtype=torch.float32, device=eng.device))
    sq.zero_()
    call("mca_grad_sqnorm", ptr(eng.gflat), eng.n_params, ptr(sq), stream_ptr())
    eng._pending_clip = float(max_norm)
    return sq.sqrt().reshape(())


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        eng = model.engine                                    # flattens the parameters
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(eng.flat)
        self.exp_avg_sq = torch.zeros_like(eng.flat)
        self.step_count = 0

    @torch.no_grad()
    def step(self, closure=None):
        eng = self.model.engine
        g = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        max_norm = getattr(eng, "_pending_clip", 0.0) or 0.0
        sq = eng._ws.get("sqnorm")
        call("mca_adamw_step", ptr(eng.flat), ptr(eng.gflat), ptr(self.exp_avg), ptr(self.exp_avg_sq), eng.n_params,
             float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], bc1, bc2, max_norm, ptr(sq) if max_norm > 0 else None,
             ptr(eng.finite_flag) if eng.check_finite else None,          # a step flagged non-finite leaves the weights alone
             stream_ptr())
        eng._pending_clip = 0.0
        eng.flat.add_(0)                 # bump the version counter: bf16 weight copies are refreshed next forward

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the engine's flat buffer, which the next backward overwrites
        for p in self.model.parameters():
            p.grad = None

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])

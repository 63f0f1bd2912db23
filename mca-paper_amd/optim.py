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

SQNORM_WORDS = 1026          # include/mca_hip.h MCA_SQNORM_WORDS: the norm + the caller-owned scratch of mca_grad_sqnorm


def clip_grad_norm_(model, max_norm: float) -> torch.Tensor:
    """Records the clipping request and returns the total gradient L2 norm (device tensor, no sync).  The
    scaling itself is applied inside the next ``FusedAdamW.step()`` (gradients in memory stay unscaled)."""
    eng = model.engine
    sq = eng._ws.setdefault("sqnorm", torch.zeros(SQNORM_WORDS, dtype=torch.float32, device=eng.device))          # word 0 + scratch
    call("mca_grad_sqnorm", ptr(eng.gflat), eng.n_params, ptr(sq), stream_ptr())          # word 0 = sum g^2, last word = the norm
    eng._pending_clip = float(max_norm)
    return sq[SQNORM_WORDS - 1].reshape(())


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        eng = model.engine                                    # flattens the parameters
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(eng.flat)
        self.exp_avg_sq = torch.zeros_like(eng.flat)
        self.step_count = 0
        # {lr, bias_corr1, bias_corr2} of the current step in device memory: a captured step (graph.GraphedStep) reads them at
        # replay time; set by the one-thread kernel mca_adamw_hyper before every step (outside the graph)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=eng.device)
        self.hyper_external = False          # True: the graph driver sets the hyper-parameters and counts the steps

    def set_hyper(self, step_count: int):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        call("mca_adamw_hyper", ptr(self.hyper), float(g["lr"]), 1.0 - b1 ** step_count, 1.0 - b2 ** step_count, stream_ptr())

    @torch.no_grad()
    def step(self, closure=None):
        eng = self.model.engine
        g = self.param_groups[0]
        if not self.hyper_external:
            self.step_count += 1
            self.set_hyper(self.step_count)
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** max(self.step_count, 1)
        bc2 = 1.0 - b2 ** max(self.step_count, 1)
        max_norm = getattr(eng, "_pending_clip", 0.0) or 0.0
        sq = eng._ws.get("sqnorm")
        call("mca_adamw_step", ptr(eng.flat), ptr(eng.gflat), ptr(self.exp_avg), ptr(self.exp_avg_sq), eng.n_params,
             float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], bc1, bc2, max_norm, ptr(sq) if max_norm > 0 else None,
             ptr(eng.finite_flag) if eng.check_finite else None,          # a step flagged non-finite leaves the weights alone
             ptr(self.hyper), stream_ptr())
        eng._pending_clip = 0.0
        eng._weights_version = -1        # the parameters moved: bf16 weight copies are refreshed by the next forward

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

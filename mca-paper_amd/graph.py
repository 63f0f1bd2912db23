"""The whole training step as ONE hipGraph (small per-GPU batches, slow hosts).

One eager step is ~330 kernel launches issued from Python (~10 ms of host time); at the data-parallel batch of the reference's
configs (8 samples per GPU) the GPU needs ~7 ms for them, so the step is host-bound, and even at
b = 32 a loaded host leaves the GPU idle between launches.  ``GraphedStep`` captures forward + loss + backward (side-stream
weight gradients included) + clip + fused AdamW once and replays it: per step the host issues one tiny kernel (this step's
learning rate and Adam bias corrections, ``mca_adamw_hyper``), optional copies into the static input buffers, and one
``hipGraphLaunch``.

What makes the step capturable: every kernel takes its stream as an argument and nothing in the library allocates or
synchronises (include/mca_hip.h); the finite checks are a device flag (no ``.item()``); the optimizer reads lr / bias
corrections from device memory; all workspaces are allocated by the warm-up steps and the loss outputs come from the graph's
private pool.  Not captured: data-parallel collectives (``dp`` runs eagerly around the graph is NOT supported: use the eager
step under DP).

Reference loop being replaced: train_accel_gpu.py:108-119 (model(batch); zero_grad; backward; clip_grad_norm_; optimizer.step).
"""
from __future__ import annotations

import torch

from . import optim as _optim


class GraphedStep:
    def __init__(self, model, optimizer, batch, clip: float = 2.0, dp=None, warmup: int = 3, overlap_wgrad: bool = False):
        if dp is not None:
            raise NotImplementedError("GraphedStep captures a single-GPU step; run the eager step under DataParallelMCA")
        self.model, self.opt, self.clip = model, optimizer, float(clip)
        eng = model.engine
        # weight-gradient GEMMs on the side stream: inside a graph the fork / join edges cost more than the overlap gains at
        # every size tried (one box, alternating processes: b = 32 21.9 ms with, 21.5 without; b = 8 7.90-8.02 with, 7.80 without)
        eng.overlap_wgrad = bool(overlap_wgrad)
        if eng.check_finite not in (False, "deferred"):
            eng.check_finite = "deferred"          # the synchronous form reads the flag on the host inside the forward
        self.static = {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in batch.items()}
        # the warm-up steps are real optimizer steps: weights, moments and the step count are put back afterwards, so that
        # constructing a GraphedStep does not train (the capture itself executes nothing)
        saved = (eng.flat.clone(), optimizer.exp_avg.clone(), optimizer.exp_avg_sq.clone(), optimizer.step_count)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):              # warm-up off the default stream: allocates every workspace, builds the cast table
            for _ in range(max(1, warmup)):
                self._body()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.opt.hyper_external = True             # from here on the driver sets lr / bias corrections and counts the steps
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._body()
        self.loss = self.out["loss"]
        eng.flat.copy_(saved[0]); optimizer.exp_avg.copy_(saved[1]); optimizer.exp_avg_sq.copy_(saved[2])
        optimizer.step_count = saved[3]
        eng.invalidate_weights()
        eng.finite_flag.zero_(); eng._flag_host.zero_(); eng._flag_event = None

    def _body(self):
        out = self.model(self.static)
        self.opt.zero_grad()
        out["loss"].backward()
        self.gnorm = _optim.clip_grad_norm_(self.model, self.clip) if self.clip else None
        self.opt.step()
        return out

    def step(self, batch=None, eager: bool = False):
        """One optimizer step; ``batch`` (same shapes) is copied into the static input buffers first.  Returns the loss tensor
        of the captured step (its value is that of THIS replay once the stream has run it).  eager=True runs the same body
        kernel by kernel instead of replaying it (bench.py times its kernels with HIP events on such a step)."""
        if batch is not None:
            for k, v in batch.items():
                for kk, vv in v.items():
                    if torch.is_tensor(vv):
                        self.static[k][kk].copy_(vv, non_blocking=True)
        eng = self.model.engine
        if eng.check_finite:
            eng.poll_finite()                      # the flag copy of an EARLIER replay that has already landed (no sync)
        self.opt.step_count += 1
        self.opt.set_hyper(self.opt.step_count)
        if eager:
            loss = self._body()["loss"]            # records its own flag event (engine._model_forward)
            return loss
        self.graph.replay()
        if eng.check_finite:                       # the captured step ends with the flag's copy to pinned host memory
            eng._flag_event = torch.cuda.Event()
            eng._flag_event.record()
        return self.loss

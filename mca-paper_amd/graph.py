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
private pool.

Data parallelism (``dp=DataParallelMCA``): the step is captured as a CHAIN of graph segments cut at the collectives, which stay
ordinary eager torch.distributed calls between two replays — forward | all-gather of the pooled block | loss + pooling backward |
all-reduce(bucket 0, async) | layer L-1 backward | all-reduce(bucket 1) | ... | encoder backward | all-reduce(last) | wait + finite-
flag MAX | clip + AdamW.  Per step the host issues ~9 ``hipGraphLaunch`` and ~9 collective calls instead of ~330 kernel launches;
the bucket all-reduces run on the process group's stream beside the following backward segments exactly as in the eager loop.
No RCCL call is ever inside a captured region, so nothing depends on RCCL's own capture support.  The segmented body runs the
backward through ``engine.forward_backward`` (no autograd node): a capture has to end on the thread that began it, and autograd
runs a CUDA backward on its own device thread.  The single-GPU body takes the same direct chain when every encoder is native
(autograd's bookkeeping around the step's one node is eight small kernels per replay); a model with a foreign torch encoder is
captured through autograd as before.

Reference loop being replaced: train_accel_gpu.py:108-119 (model(batch); zero_grad; backward; clip_grad_norm_; optimizer.step).
"""
from __future__ import annotations

import torch

from . import optim as _optim


class GraphedStep:
    def __init__(self, model, optimizer, batch, clip: float = 2.0, dp=None, warmup: int = 3, overlap_wgrad: bool = False):
        self.model, self.opt, self.clip, self.dp = model, optimizer, float(clip), dp
        eng = model.engine
        self.direct = eng.can_forward_backward()
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
        self.program = None
        if dp is None:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._body()
        else:
            self._capture_segments()
        self.loss = self.out["loss"]
        eng.flat.copy_(saved[0]); optimizer.exp_avg.copy_(saved[1]); optimizer.exp_avg_sq.copy_(saved[2])
        optimizer.step_count = saved[3]
        eng.invalidate_weights()
        eng.finite_flag.zero_(); eng._flag_host.zero_(); eng._flag_event = None

    def _capture_segments(self):
        """The data-parallel step as [graph, collective, graph, collective, ...] (module docstring).  Capturing issues NO
        collective: a capture that fails on one rank raises there with nothing in flight, and the caller can agree on a common
        path (bench.py) or exit non-zero (train_accel_gpu.py)."""
        import gc
        torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()          # as torch.cuda.graph() does before a capture
        pool = torch.cuda.graph_pool_handle()
        self.program = []
        stream = torch.cuda.Stream(device=self.model.engine.device)
        stream.wait_stream(torch.cuda.current_stream())
        state = {"g": None}

        def begin():
            state["g"] = torch.cuda.CUDAGraph()
            state["g"].capture_begin(pool=pool, capture_error_mode="relaxed")          # (the process group's watchdog thread polls events)

        def end():
            state["g"].capture_end()
            self.program.append(state["g"])

        def cut(fn):
            # the collective is only RECORDED here: while capturing, the buffers hold nothing (captured kernels do not run), and
            # a rank whose capture fails must not leave the others inside a half-issued sequence of collectives
            end()
            self.program.append(fn)
            begin()

        self.dp.set_cut(cut)
        try:
            with torch.cuda.stream(stream):
                begin()
                try:
                    self.out = self._body()
                except BaseException:
                    try:
                        state["g"].capture_end()          # leave no stream in capture mode behind a failed body
                    except Exception:
                        pass
                    raise
                end()
        finally:
            self.dp.set_cut(None)
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()

    def _body(self):
        if self.dp is not None:          # no autograd node (see _capture_segments); the same kernel chain
            out = self.model.engine.forward_backward(self.static)
            self.dp.finish_backward()
            self.gnorm = _optim.clip_grad_norm_(self.model, self.clip) if self.clip else None
            self.opt.step()
            return out
        if self.direct:
            # one GPU, native encoders: the same direct chain (autograd's own bookkeeping around the step's single node is eight
            # small kernels per replay: grad_output fills, the products with them, the accumulation into .grad)
            out = self.model.engine.forward_backward(self.static)
            self.gnorm = _optim.clip_grad_norm_(self.model, self.clip) if self.clip else None
            self.opt.step()
            return out
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
        if self.program is not None:
            for item in self.program:
                item.replay() if isinstance(item, torch.cuda.CUDAGraph) else item()
        else:
            self.graph.replay()
        # the replay's AdamW moved the fp32 weights behind torch's version counters, and its bf16 cast ran BEFORE that update: an
        # eval / no_grad forward after this step must rebuild the bf16 GEMM-weight copies (the next replay casts for itself)
        eng.invalidate_weights()
        if eng.check_finite:                       # the captured step ends with the flag's copy to pinned host memory
            eng._flag_event = torch.cuda.Event()
            eng._flag_event.record()
        return self.loss

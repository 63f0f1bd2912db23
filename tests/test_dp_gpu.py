"""GPU tests of the data-parallel step: two ranks on one GPU against the oracle's objective on the concatenated batch, and against a
native single-process run of the same objective."""
import copy
import importlib
import os
import pytest
import socket
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util_small import small_config, rel_err, to_device
from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = importlib.import_module("mca-paper_amd")
        dpm = importlib.import_module("mca-paper_amd.dp")
        cfg = small_config("mca")
        b = 4
        sd = P.params.init_state_dict(cfg, seed=3)
        full = P.data.synthetic_batch(cfg, b * world, seed=21, p_drop=0.3)
        local = {k: {kk: vv[rank * b:(rank + 1) * b] for kk, vv in v.items()} for k, v in full.items()}
        model = P.MCA(**copy.deepcopy(cfg))
        model.load_state_dict(sd, strict=False)
        model = model.cuda()
        dp = dpm.DataParallelMCA(model)
        outp = dp(to_device(local, "cuda"))
        outp["loss"].backward()
        dp.finish_backward()
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"grads": {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()},
                        "loss": float(outp["loss"])}, out)
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def P():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("mca-paper_amd")


# ------------------------------------------------------------------------------------------------ data parallel
def _dp_worker(rank, world, port, out, p_drop, variant="mca"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = importlib.import_module("mca-paper_amd")
        dpm = importlib.import_module("mca-paper_amd.dp")
        cfg = small_config(variant)
        b = 4
        sd = P.params.init_state_dict(cfg, seed=3)
        full = P.data.synthetic_batch(cfg, b * world, seed=21, p_drop=p_drop)
        local = {k: {kk: vv[rank * b:(rank + 1) * b] for kk, vv in v.items()} for k, v in full.items()}
        model = P.MCA(**copy.deepcopy(cfg))
        model.load_state_dict(sd, strict=False)
        model = model.cuda()
        dp = dpm.DataParallelMCA(model)
        outp = dp(to_device(local, "cuda"))
        outp["loss"].backward()
        dp.finish_backward()
        torch.cuda.synchronize()
        torch.save({"grads": {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()},
                    "loss": float(outp["loss"])}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_dp2_native_matches_oracle_objective(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import mca_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    P = importlib.import_module("mca-paper_amd")
    cfg = small_config("mca"); W, b = 2, 4
    S = O.Structure(cfg); names = S.modalities
    sd = P.params.init_state_dict(cfg, seed=3)
    params = {k: v for k, v in sd.items() if O.is_param(k)}
    for p in params.values():
        p.requires_grad_(True)
    full = P.data.synthetic_batch(cfg, b * W, seed=21, p_drop=0.3)
    keep = {}
    def dp_objective(mode):
        """gradient of (1 / W) sum_r loss_r on the concatenated batch, in the oracle's precision mode `mode`"""
        for p in params.values():
            p.grad = None
        Pr = O.Prec(mode)
        tokens, padding, sample_mask = O.encode_and_pack(S, sd, full, Pr)
        pooled = O.mca_trunk(S, sd, tokens, padding, Pr)
        keep["pooled"] = pooled
        tot, loss0 = 0, None
        for r in range(W):
            sm = {n: sample_mask[n][r * b:(r + 1) * b] for n in names}
            l = O.pretraining_loss(S, pooled[r * b:(r + 1) * b], sm, sd["loss.loss_fn.logit_scale"], pooled_all=pooled, rank=r)["loss"]
            loss0 = l if r == 0 else loss0
            tot = tot + l
        (tot / W).backward()
        return float(loss0), {n: p.grad.detach().clone() for n, p in params.items() if p.grad is not None}
    loss_emu, g_emu = dp_objective("bf16emu")
    loss0, g_ref = dp_objective("fp32")
    # (the single-GPU loss bound of tests/test_step_gpu.py: the loss is a difference of logits, its error scales with their size)
    pf = keep["pooled"].detach().double()
    scale = max(float((pf[:, i] @ pf[:, j].t()).abs().max()) for i in range(pf.shape[1]) for j in range(pf.shape[1])) * float(torch.exp(sd["loss.loss_fn.logit_scale"].detach().double()))
    assert abs(got["loss"] - loss0) <= 1e-4 * scale + 1e-4, (got["loss"], loss0, scale)
    # the single-GPU bound (tests/test_step_gpu.py): no worse than bf16 arithmetic itself, i.e. within a few x the distance of
    # the bf16-EMULATING oracle from the fp32 one, per tensor; median 3 %  (round 3 asserted a blanket 25 % / 5 %)
    errs = []
    for n, gref in g_ref.items():
        if gref.abs().max() == 0:
            continue
        e, e_emu = rel_err(got["grads"][n], gref), rel_err(g_emu[n], gref)
        assert e < 4 * e_emu + 2e-2, (n, e, e_emu)
        errs.append((e, n))
    assert sorted(e for e, _ in errs)[len(errs) // 2] < 0.03, (max(errs), sorted(e for e, _ in errs)[len(errs) // 2])


@pytest.mark.parametrize("variant,p_drop", [("mca", 0.3), ("zorro", 0.4)])
def test_dp2_native_equals_single_process_objective(P, tmp_path, variant, p_drop):
    """Two ranks (gloo, both on this GPU) through dp.py + the real kernels against ONE native process that evaluates the same
    objective (1/W) sum_r loss_r on the concatenated batch: same kernels, same arithmetic; only the order of fp32 atomic adds
    differs.  MCA with the fusion-channel loss, and MMA (zorro masks) with 40 % of the modalities dropped (reference
    utils/contrastive_loss_with_temperature.py:26-31, model.py:198-207)."""
    W, b = 2, 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(W, port, out, p_drop, variant), nprocs=W, join=True)
    got = [torch.load(out + f".{r}") for r in range(W)]
    for n in got[0]["grads"]:
        assert torch.equal(got[0]["grads"][n], got[1]["grads"][n]), n          # the all-reduce left identical gradients
    cfg = small_config(variant)
    sd = P.params.init_state_dict(cfg, seed=3)
    full = to_device(P.data.synthetic_batch(cfg, b * W, seed=21, p_drop=p_drop), "cuda")
    model = P.MCA(**copy.deepcopy(cfg)); model.load_state_dict(sd, strict=False); model = model.cuda()
    eng = model.engine
    eng.refresh_weights()
    ws = eng.workspace(b * W); ws["gen"] += 1
    eng._encode(full, ws, True)
    pooled = eng.forward_trunk(ws).view(b * W, eng.R, eng.D)
    ls = model.loss.loss_fn
    ls.logit_scale.data.clamp_(ls.logit_scale_min, ls.logit_scale_max)
    present = ws["present_cur"]
    parts, dlogit, losses = [], 0, []
    for r in range(W):
        res = eng.loss_fwd_bwd(pooled.contiguous(), present.contiguous(), b, r * b)
        parts.append(res["d_pooled"].clone()); dlogit = dlogit + res["d_logit"].clone(); losses.append(float(res["loss"]))
    eng.backward(ws, torch.cat(parts) / W, dlogit / W)
    torch.cuda.synchronize()
    for r in range(W):
        assert abs(got[r]["loss"] - losses[r]) <= 1e-5 * abs(losses[r]) + 1e-6
    errs = []
    for n, p in model.named_parameters():
        g = eng.grad_of(p).cpu()
        if float(g.abs().max()) == 0:
            assert float(got[0]["grads"][n].abs().max()) < 1e-6, n
            continue
        errs.append((rel_err(got[0]["grads"][n], g), n))
    assert max(errs)[0] < 1e-2, max(errs)



def _steps_worker(rank, world, port, out, variant, p_drop, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = importlib.import_module("mca-paper_amd")
        dpm = importlib.import_module("mca-paper_amd.dp")
        optim = importlib.import_module("mca-paper_amd.optim")
        cfg = small_config(variant)
        b = 2
        sd = P.params.init_state_dict(cfg, seed=3)
        model = P.MCA(**copy.deepcopy(cfg)); model.load_state_dict(sd, strict=False); model = model.cuda()
        model.engine.check_finite = "deferred"
        opt = optim.FusedAdamW(model, lr=1e-3)
        dp = dpm.DataParallelMCA(model)
        losses = []
        for i in range(steps):
            full = P.data.synthetic_batch(cfg, b * world, seed=31 + i, p_drop=p_drop)
            local = to_device({k: {kk: vv[rank * b:(rank + 1) * b] for kk, vv in v.items()} for k, v in full.items()}, "cuda")
            o = dp(local); opt.zero_grad(); o["loss"].backward(); dp.finish_backward()
            optim.clip_grad_norm_(model, 2.0); opt.step()
            losses.append(float(o["loss"]))
        torch.cuda.synchronize()
        torch.save({"flat": model.engine.flat.detach().clone().cpu(), "losses": losses}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant,p_drop", [("mca", 0.3), ("zorro", 0.4)])
def test_dp4_replica_weights_stay_bit_identical_over_four_steps(P, tmp_path, variant, p_drop):
    """Four ranks (gloo, all on this GPU: within the box's process limit) train four optimizer steps through dp.py, the bucketed
    all-reduce, the clip and the fused AdamW: every replica must hold the SAME BITS in its flat parameter buffer afterwards (the
    all-reduced gradients are identical by construction; clip coefficient and update are deterministic functions of them)."""
    W, steps = 4, 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp4.pt")
    mp.spawn(_steps_worker, args=(W, port, out, variant, p_drop, steps), nprocs=W, join=True)
    got = [torch.load(out + f".{r}") for r in range(W)]
    for r in range(1, W):
        assert torch.equal(got[0]["flat"], got[r]["flat"]), f"rank {r} diverged from rank 0"
    sd0 = P.params.init_state_dict(small_config(variant), seed=3)
    assert all(l == l for g in got for l in g["losses"])          # finite
    moved = max(float((got[0]["flat"] != 0).float().mean()), 0.0)
    assert moved > 0.5          # (the buffer holds trained weights, not zeros)

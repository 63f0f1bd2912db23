"""The oracle (oracle/mca_oracle.py) against the golden vectors produced by the reference itself
(oracle/make_goldens.py).  fp32, tolerances written per check."""
import glob
import os

import pytest
import torch

from oracle import mca_oracle as O

from conftest import GOLDEN

CASES = sorted(os.path.basename(p)[5:-3] for p in glob.glob(os.path.join(GOLDEN, "tiny_*.pt")))


def _close(a, b, rtol, atol, what):
    a, b = a.double(), b.double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} vs scale {ref:.3e}"


@pytest.mark.parametrize("case", CASES)
def test_tiny_case(case):
    rec = torch.load(os.path.join(GOLDEN, f"tiny_{case}.pt"), weights_only=False)
    cfg = rec["config"]
    eao = bool(cfg.get("eao"))          # the EAO baseline (reference model.py:481-596): oracle.eao_forward
    S = O.EAOStructure(cfg) if eao else O.Structure(cfg)
    # static structure, bit-exact
    if not eao:
        assert torch.equal(S.attn_mask, rec["attn_mask"])
        assert torch.equal(S.pool_mask, rec["pool_mask"])
        assert S.ret_types == rec["return_token_types"]
    assert torch.equal(S.token_types, rec["token_types"])
    assert [t[0] for t in O.loss_schedule(S)] == rec["loss_names"]

    sd = {k: v.clone() for k, v in rec["init_state"].items()}
    out, grads, gn, opt = O.train_step(S, sd, rec["batch"], "fp32", lr=rec["lr"], clip=rec["clip"])
    gold = rec["outputs"]
    # embeddings
    for k, v in gold["embeddings"].items():
        key = k
        if k not in out:
            key = frozenset(int(x) for x in k.split("|"))
        _close(out[key], v, 2e-5, 1e-6, f"embedding {k}")
    # loss terms, NaN pattern included
    for k, v in gold["losses"].items():
        mine = out["losses"][k]
        assert bool(torch.isnan(mine)) == bool(torch.isnan(v)), k
        if not torch.isnan(v):
            _close(mine, v, 2e-5, 1e-6, f"loss {k}")
    _close(out["loss"], gold["loss"], 2e-5, 1e-6, "total loss")
    for extra in ("fcl_loss", "no-fcl_loss"):
        if extra in gold:
            _close(out[extra], gold[extra], 2e-5, 1e-6, extra)
    for m, v in gold["modality_sample_mask"].items():
        assert torch.equal(out["modality_sample_mask"][m], v)
    # in-place side effects of the forward (embedding renorm, logit_scale clamp)
    for k, v in rec["state_after_forward"].items():
        pass  # checked through state_step1 below (the optimizer acts on the renormed weights)
    # gradients
    for n, g in rec["grads"].items():
        if g is None:
            assert grads[n].abs().max() == 0, n
        else:
            _close(grads[n], g, 1e-4, 1e-7, f"grad {n}")
    _close(gn, rec["grad_norm"], 1e-5, 0, "grad norm")
    # weights after step 1 and step 2 (AdamW + clip)
    for n, v in rec["state_step1"].items():
        # Adam's first step is lr*g/(|g|+eps): elements with |g| ~ eps amplify fp32 noise, so the
        # absolute tolerance is a few % of one step (lr)
        _close(sd[n], v, 1e-5, 5e-2 * rec["lr"], f"step1 {n}")
    O.train_step(S, sd, rec["batch"], "fp32", lr=rec["lr"], clip=rec["clip"], opt_state=None)


@pytest.mark.parametrize("case", ["bimodal_drop", "mca_fcl", "eao_fcl_drop"])
def test_two_steps(case):
    """second AdamW step exercises non-zero moments."""
    rec = torch.load(os.path.join(GOLDEN, f"tiny_{case}.pt"), weights_only=False)
    eao = bool(rec["config"].get("eao"))
    S = O.EAOStructure(rec["config"]) if eao else O.Structure(rec["config"])
    sd = {k: v.clone() for k, v in rec["init_state"].items()}
    params = [sd[k] for k in sd if O.is_param(k)]
    for p in params:
        p.requires_grad_(True)
    opt = torch.optim.AdamW(params, lr=rec["lr"])
    for s in range(2):
        for p in params:
            p.grad = None
        out = (O.eao_forward if eao else O.mca_forward)(S, sd, rec["batch"], "fp32")
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], rec["clip"])
        opt.step()
    for n, v in rec["state_step2"].items():
        _close(sd[n].detach(), v, 2e-5, 5e-2 * rec["lr"], f"step2 {n}")


def test_fp64_and_bf16emu_modes_run():
    rec = torch.load(os.path.join(GOLDEN, "tiny_mca_fcl_drop.pt"), weights_only=False)
    S = O.Structure(rec["config"])
    sd = {k: v.clone() for k, v in rec["init_state"].items()}
    o32 = O.mca_forward(S, sd, rec["batch"], "fp32")
    o64 = O.mca_forward(S, sd, rec["batch"], "fp64")
    o16 = O.mca_forward(S, sd, rec["batch"], "bf16emu")
    e32 = (o32["pooled"].double() - o64["pooled"]).norm() / o64["pooled"].norm()
    e16 = (o16["pooled"].double() - o64["pooled"]).norm() / o64["pooled"].norm()
    assert e32 < 1e-5
    assert 1e-5 < e16 < 2e-2          # bf16 rounding is visible but bounded


def test_golden_recipe_takes_one_flag_per_process():
    """oracle/make_goldens.py refuses two flags in one process: the reference shares its loss temperature (a module-level
    nn.Parameter) between the models of a process, so the order of the cases would change the goldens.
    The refusal comes from the argument parser, before the reference is imported: it holds on the GPU box too."""
    import os, subprocess, sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "make_goldens.py")
    r = subprocess.run([sys.executable, script, "--eao", "--tiny"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "one flag per process" in r.stderr

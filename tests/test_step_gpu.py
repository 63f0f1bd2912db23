"""Whole-step parity on a real MI355X: native MCA step (HIP kernels through the C ABI) against the oracle
(oracle/mca_oracle.py, pinned by the reference's goldens) and against golden vectors produced by the
reference itself at CMU size."""
import copy
import importlib
import os

import pytest
import torch

from conftest import GOLDEN
from util_small import small_config, run_native_step, run_oracle_step, rel_err, to_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("mca-paper_amd")


# Tolerances.  The native path computes GEMMs and attention with bf16 operands and fp32 accumulation, residual
# stream / LayerNorm / softmax statistics / loss in fp32 (>= the reference under Accelerate bf16 autocast, whose own
# error vs fp64 on the pooled embeddings is 6.4-7.5e-4 at CMU size: BASELINE.md section 2).
TOL_POOLED = 1e-3        # rel L2 of the pooled embeddings vs the reference / fp32 oracle (north_star: 1e-3)
TOL_POOLED_EMU = 1e-3    # vs the oracle with bf16 rounding inserted where the kernels round
TOL_LOGIT = 1e-4         # |d loss term| <= TOL_LOGIT * (temperature * max |a.b|): the logits are O(10^3..10^4), the
                         # loss is a difference of logits, so its error scales with the logit magnitude
# Gradient-norm error per parameter tensor vs the reference's own numbers at CMU size, b = 2.  ONE draw of that statistic is noisy
# (the contrastive softmax at temperature 14 over un-normalised embeddings amplifies the 1e-3 embedding error into %-level gradient
# error for ANY bf16 path; a re-rounding moves a single tensor's error by 4x): the bounds below come from the ENSEMBLE of
# tests/studies/lazy_softmax_seed_study.py (8 data seeds x {MCA, MMA p_drop 0.4}, distance to the fp64 oracle, committed as
# profiles/r05_lazy_softmax_seed_study.txt), about 1.4 x the worst case seen for the shipped forward form:
#   median over tensors 1.06e-2, 90th percentile 5.3e-2, maximum 1.26e-1   (textbook recurrence: 3.3e-2, 1.3e-1, 3.0e-1)
TOL_GRADNORM_CMU = 0.18          # any tensor
TOL_GRADNORM_CMU_P90 = 0.07
TOL_GRADNORM_CMU_MEDIAN = 0.015
TOL_GN = 1e-2            # rel of the global gradient norm


def _logit_scale(pooled, logit_scale=2.6593):
    p = pooled.double()
    mx = max(float((p[:, i] @ p[:, j].t()).abs().max()) for i in range(p.shape[1]) for j in range(p.shape[1]))
    return mx * float(torch.exp(torch.tensor(logit_scale)))


def _check_losses(nat, ref_losses, ref_total, scale):
    assert set(nat["losses"]) == set(ref_losses)
    for k, v in ref_losses.items():
        v = float(v)
        if v != v:
            assert nat["losses"][k] != nat["losses"][k], f"{k} should be NaN"
        else:
            assert abs(nat["losses"][k] - v) <= TOL_LOGIT * scale + 1e-4, (k, nat["losses"][k], v, scale)
    assert abs(nat["loss"] - float(ref_total)) <= TOL_LOGIT * scale + 1e-4


@pytest.mark.parametrize("variant,p_drop", [("mca", 0.0), ("mca", 0.35), ("zorro", 0.35), ("bimodal", 0.35), ("tab", 0.35)])
def test_small_step_vs_oracle(P, variant, p_drop):
    from oracle import mca_oracle as O
    cfg = small_config(variant)
    batch = P.data.synthetic_batch(cfg, 6, seed=5, p_drop=p_drop)
    if variant == "tab":        # value-path sentinels: padding value -1, clamp above max_value, a missing entry
        v = batch["video"]["values"]
        v[0, 3], v[1, 5], v[2, 7] = -1.0, 250.0, -10000.0
        batch["video"]["attention_mask"] = (v == -10000).to(torch.long)
    sd = P.params.init_state_dict(cfg, seed=3)
    for k in sd:                # embedding rows on both sides of max_norm = 1
        if k.endswith("embedding.weight"):
            sd[k][::2] *= 0.05
    g = torch.Generator().manual_seed(9)            # non-trivial gammas / biases
    for k in sd:
        if k.endswith("gamma") or k.endswith("bias") or ("token_encoder" in k and sd[k].dim() == 1):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    nat = run_native_step(P, cfg, sd, batch, lr=1e-3, clip=2.0)
    ref = run_oracle_step(O, cfg, sd, batch, "fp32", lr=1e-3, clip=2.0)
    emu = run_oracle_step(O, cfg, sd, batch, "bf16emu", lr=1e-3, clip=2.0)
    assert rel_err(nat["pooled"], ref["pooled"]) < TOL_POOLED
    assert rel_err(nat["pooled"], emu["pooled"]) < TOL_POOLED_EMU
    _check_losses(nat, ref["losses"], ref["loss"], _logit_scale(ref["pooled_full"]))
    errs, errs_emu = [], []
    for n, gref in ref["grads"].items():
        gn = nat["grads"][n]
        if gref.abs().max() == 0:
            assert gn.abs().max() < 1e-6, n
            continue
        e = rel_err(gn, gref)
        e_emu = rel_err(emu["grads"][n], gref)
        errs.append(e); errs_emu.append(e_emu)
        # no worse than bf16 arithmetic itself: a few x the error of the bf16-emulating oracle (the only gradient bound: a
        # blanket percentage says nothing about a tensor whose bf16 error is 0.5 %)
        assert e < 4 * e_emu + 2e-2, (n, e, e_emu)
    # (the median over the tensors, like every gradient bound here, relative to what bf16 arithmetic itself gives: the same form as
    #  test_tcga_shape_b2_vs_oracle; a blanket 3 % sat 0.1 % below the value one re-rounding of the forward's P produces)
    med, med_emu = sorted(errs)[len(errs) // 2], sorted(errs_emu)[len(errs_emu) // 2]
    print(f"small step {variant} p_drop {p_drop}: median gradient error {med:.2e} (bf16-emulating oracle {med_emu:.2e})")
    assert med < 1.3 * med_emu + 0.01, (med, med_emu)
    assert abs(nat["grad_norm"] - ref["grad_norm"]) < TOL_GN * ref["grad_norm"]
    # one clip + AdamW step (lr 1e-3).  Adam's first update is -lr * g / (|g| + eps) - lr * wd * w: a sign function of the
    # gradient, so an element whose gradient is within the bf16 error of zero may legitimately move the other way.  Two checks
    # that do not need a blanket allowance: (a) the optimizer's arithmetic, EXACTLY, on the native gradients (clip coefficient
    # from the native norm); (b) against the oracle's update, on the elements whose reference gradient is above the noise floor
    # (the larger half of |g| of each tensor): the same direction in at least 99.5 % of them.
    clip = min(1.0, 2.0 / (nat["grad_norm"] + 1e-6))
    for n, w_ref in ref["state"].items():
        upd_ref = w_ref - sd[n]
        upd_nat = nat["state"][n] - sd[n]
        if upd_ref.abs().max() < 1e-7:
            continue
        if not n.endswith("embedding.weight"):          # (the forward renormalises the embedding table in place: its update is not the optimizer's alone)
            gc = nat["grads"][n].double() * clip
            m_hat, v_hat = gc, gc * gc                      # first step: bias-corrected moments are g and g^2
            want = -1e-3 * (m_hat / (v_hat.sqrt() + 1e-8)) - 1e-3 * 0.01 * sd[n].double()
            assert (upd_nat.double() - want).abs().max() < 2e-6, (n, float((upd_nat.double() - want).abs().max()))
        g_ref = ref["grads"][n]
        big = g_ref.abs() >= g_ref.abs().flatten().median()
        if int(big.sum()) >= 8:
            agree = (torch.sign(upd_nat[big]) == torch.sign(upd_ref[big])).float().mean()
            assert agree > 0.995, (n, float(agree))


@pytest.mark.parametrize("case", ["mca", "mma_d40"])
def test_cmu_b2_vs_reference_golden(P, case):
    """CMU-shaped run (N=2538, D=512, L=5) at b=2 against numbers produced by the REFERENCE ITSELF
    (tests/golden/cmu_*_b2.pt, generated by oracle/make_goldens.py --cmu)."""
    rec = torch.load(os.path.join(GOLDEN, f"cmu_{case}_b2.pt"), weights_only=False)
    zorro = case != "mca"
    cfg = P.config.cmu_model_config(batch_size=2, zorro=zorro)
    sd = P.params.init_state_dict(cfg, seed=rec["seed"])
    batch = P.data.synthetic_batch(cfg, 2, seed=rec["data_seed"], p_drop=rec["p_drop"], lengths="uniform")
    nat = run_native_step(P, cfg, sd, batch, lr=1e-4, clip=2.0)
    e = rel_err(nat["pooled"], rec["pooled"])
    assert e < TOL_POOLED, f"pooled rel err {e}"
    scale = _logit_scale(rec["pooled"])
    _check_losses(nat, rec["losses"], rec["loss"], scale)
    rels, bad = [], []
    for n, gn_ref in rec["grad_norms"].items():
        gn = float(nat["grads"][n].norm())
        if n.endswith("logit_scale"):
            # d loss / d log-temperature is itself a softmax-weighted sum of logits: tolerance on the logit scale
            if abs(gn - gn_ref) > 1e-3 * scale:
                bad.append((n, gn, gn_ref))
            continue
        if gn_ref < 1e-12:
            if gn > 1e-6:
                bad.append((n, gn, gn_ref))
            continue
        r = abs(gn - gn_ref) / gn_ref
        rels.append(r)
        if r > TOL_GRADNORM_CMU:
            bad.append((n, gn, gn_ref))
        ref_sl = rec["grad_slices"][n]
        if ref_sl.abs().max() > 0 and rel_err(nat["grads"][n].flatten()[:64], ref_sl) > 0.25:
            bad.append((n + "[slice]", rel_err(nat["grads"][n].flatten()[:64], ref_sl), 0))
    rels.sort()
    print(f"cmu_{case}_b2: pooled rel err {e:.2e}; gradient-norm error over {len(rels)} tensors: median {rels[len(rels) // 2]:.2e}, "
          f"p90 {rels[int(len(rels) * 0.9)]:.2e}, max {rels[-1]:.2e}")
    assert not bad, bad[:8]
    assert rels[len(rels) // 2] < TOL_GRADNORM_CMU_MEDIAN and rels[int(len(rels) * 0.9)] < TOL_GRADNORM_CMU_P90
    # ---- against the reference's OWN bf16 behaviour: the same case run by the reference under torch.autocast("cpu", bfloat16)
    # (tests/golden/cmu_*_b2_autocast.pt, oracle/make_goldens.py --cmu_autocast; reference model.py:448-478).  Per quantity, the
    # native step's distance to the fp32 golden is at most 2 x the reference's own bf16-vs-fp32 distance.
    ac = torch.load(os.path.join(GOLDEN, f"cmu_{case}_b2_autocast.pt"), weights_only=False)
    d_pooled_ref = rel_err(ac["pooled"], rec["pooled"])
    d_loss_ref, d_loss_nat = abs(float(ac["loss"]) - float(rec["loss"])), abs(nat["loss"] - float(rec["loss"]))
    terms = [k for k, v in rec["losses"].items() if torch.isfinite(v) and torch.isfinite(ac["losses"][k])]
    d_term_ref = max(abs(float(ac["losses"][k]) - float(rec["losses"][k])) for k in terms)
    d_term_nat = max(abs(nat["losses"][k] - float(rec["losses"][k])) for k in terms)
    rels_ref = sorted(abs(ac["grad_norms"][n] - g) / g for n, g in rec["grad_norms"].items() if g >= 1e-12 and not n.endswith("logit_scale"))
    q = lambda v: (v[len(v) // 2], v[int(len(v) * 0.9)], v[-1])
    print(f"cmu_{case}_b2 native | reference under bf16 autocast (distance to the fp32 golden): pooled {e:.2e} | {d_pooled_ref:.2e}; "
          f"loss {d_loss_nat:.3f} | {d_loss_ref:.3f}; worst loss term {d_term_nat:.3f} | {d_term_ref:.3f}; gradient norms median / p90 / max "
          f"{q(rels)[0]:.2e} / {q(rels)[1]:.2e} / {q(rels)[2]:.2e} | {q(rels_ref)[0]:.2e} / {q(rels_ref)[1]:.2e} / {q(rels_ref)[2]:.2e}")
    assert e <= 2 * d_pooled_ref and d_loss_nat <= 2 * d_loss_ref and d_term_nat <= 2 * d_term_ref
    assert all(a_ <= 2 * b_ for a_, b_ in zip(q(rels), q(rels_ref))), (q(rels), q(rels_ref))


def test_tcga_shape_b2_vs_oracle(P):
    """TCGA_config1-shaped step (4 TabularEncoder modalities, N = 2548, 60 loss terms) at b=2 against the oracle."""
    from oracle import mca_oracle as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = P.config.tcga_model_config(batch_size=2)
    sd = P.params.init_state_dict(cfg, seed=43)
    batch = P.data.synthetic_batch(cfg, 2, seed=77, p_drop=0.25)
    nat = run_native_step(P, cfg, sd, batch, lr=1e-4, clip=2.0)
    ref = run_oracle_step(O, cfg, sd, batch, "fp32", lr=1e-4, clip=2.0)
    emu = run_oracle_step(O, cfg, sd, batch, "bf16emu", lr=1e-4, clip=2.0)
    assert len(ref["losses"]) == 60
    assert rel_err(nat["pooled"], ref["pooled"]) < TOL_POOLED
    _check_losses(nat, ref["losses"], ref["loss"], _logit_scale(ref["pooled_full"]))
    # 60 temperature-14 softmaxes over a batch of 2: gradients are extremely sensitive to the 1e-3 embedding error; the
    # yardstick is the oracle with bf16 rounding at the kernels' rounding points (median 8 %, max 31 % vs fp32 here)
    rels, rels_emu = [], []
    for n, gref in ref["grads"].items():
        if n.endswith("logit_scale") or gref.abs().max() == 0:
            continue
        e, e_emu = rel_err(nat["grads"][n], gref), rel_err(emu["grads"][n], gref)
        rels.append(e); rels_emu.append(e_emu)
        assert e < 1.5 * e_emu + 0.05, (n, e, e_emu)
    med, med_emu = sorted(rels)[len(rels) // 2], sorted(rels_emu)[len(rels) // 2]
    assert med < 1.3 * med_emu + 0.01, (med, med_emu)
    assert abs(nat["grad_norm"] - ref["grad_norm"]) < TOL_GN * ref["grad_norm"]


def test_tcga_b2_vs_reference_golden(P):
    """TCGA_config1-shaped run (4 TabularEncoder modalities, N = 2548, 60 loss terms) at b=2 against numbers produced by the
    REFERENCE ITSELF (tests/golden/tcga_b2.pt, oracle/make_goldens.py --tcga): pins the tabular encoder path (embedding
    max_norm renorm, value MLP, -1 padding, int64 masks) at full size."""
    rec = torch.load(os.path.join(GOLDEN, "tcga_b2.pt"), weights_only=False)
    cfg = P.config.tcga_model_config(batch_size=2)
    sd = P.params.init_state_dict(cfg, seed=rec["seed"])
    batch = P.data.synthetic_batch(cfg, 2, seed=rec["data_seed"], p_drop=rec["p_drop"])
    nat = run_native_step(P, cfg, sd, batch, lr=1e-4, clip=2.0)
    assert nat["pooled"].shape == rec["pooled"].shape
    e = rel_err(nat["pooled"], rec["pooled"])
    assert e < TOL_POOLED, f"pooled rel err {e}"
    assert len(rec["losses"]) == 60
    _check_losses(nat, rec["losses"], rec["loss"], _logit_scale(rec["pooled"]))
    # gradients: 60 temperature-14 softmaxes over a batch of 2 amplify the 1e-3 embedding error (the bf16-emulating oracle is
    # 8 % median / 31 % max off fp32 on this case, test_tcga_shape_b2_vs_oracle): norms within that envelope
    rels = []
    for n, gn_ref in rec["grad_norms"].items():
        if n.endswith("logit_scale") or gn_ref < 1e-12:
            continue
        rels.append(abs(float(nat["grads"][n].norm()) - gn_ref) / gn_ref)
    rels.sort()
    assert rels[len(rels) // 2] < 0.02 and rels[-1] < 0.15, (rels[len(rels) // 2], rels[-1])          # observed 0.8 % / 7.2 %


def test_long_sequence_step_runs(P):
    """BASELINE config 5 layout (4 x 1500 tokens + 88 fusion = 6088) at a small batch: the step runs, loss finite, every
    parameter receives a finite gradient."""
    optim = importlib.import_module("mca-paper_amd.optim")
    cfg = P.config.cmu_model_config(batch_size=2, long_seq=True)
    torch.manual_seed(1)
    model = P.MCA(**copy.deepcopy(cfg)).cuda()
    opt = optim.FusedAdamW(model, lr=1e-4)
    batch = to_device(P.data.synthetic_batch(cfg, 2, seed=3, p_drop=0.2), "cuda")
    out = model(batch)
    opt.zero_grad()
    out["loss"].backward()
    optim.clip_grad_norm_(model, 2.0)
    opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(out["loss"])
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        assert torch.isfinite(p).all(), n


def test_long_sequence_forward_vs_oracle(P):
    """BASELINE config 5 layout (N = 6088, 24 key tiles per modality, ragged lengths + dropped modalities) at b = 1: pooled
    embeddings and loss against the fp32 oracle (forward only: the oracle's dense 6088 x 6088 scores take ~10 s of CPU)."""
    O = importlib.import_module("oracle.mca_oracle")
    cfg = P.config.cmu_model_config(batch_size=2, long_seq=True)
    sd = P.params.init_state_dict(cfg, seed=5)
    batch = P.data.synthetic_batch(cfg, 2, seed=11, p_drop=0.2, lengths="uniform")
    model = P.MCA(**copy.deepcopy(cfg))
    model.load_state_dict(sd, strict=False)
    model = model.cuda()
    with torch.no_grad():
        out = model(to_device(batch, "cuda"))
        ref = O.mca_forward(O.Structure(copy.deepcopy(cfg)), {k: v.clone() for k, v in sd.items()}, batch, "fp32")
    slots = model.output_slots()
    by_slot = {}
    for k, sl in slots.items():
        by_slot.setdefault(sl, k)
    pooled = torch.stack([out[by_slot[sl]] for sl in sorted(by_slot)], 1).cpu()
    e = rel_err(pooled, ref["pooled"][:, :pooled.shape[1]])
    assert e < TOL_POOLED, f"pooled rel err {e}"
    scale = _logit_scale(ref["pooled"])
    assert abs(float(out["loss"]) - float(ref["loss"])) <= TOL_LOGIT * scale + 1e-4


@pytest.mark.parametrize("variant", ["mca", "zorro"])
def test_small_step_fp8_attention_vs_fp8emu_oracle(P, variant):
    """engine.set_attention_dtype('fp8') (BASELINE configs[4]): the fusion layers' attention with MX-fp8 operands in BOTH
    directions (forward Q K^T and P V; backward the S and dP recomputes of both passes, gradient products in bf16).  The oracle's
    fp8emu mode restates the same arithmetic in both directions (oracle._Fp8AttentionCore: nothing straight-through).  STATED
    TOLERANCES: pooled embeddings within 4e-3 rel-L2 of fp8emu (measured 0.7-1.1e-3) and within 6e-3 of the fp32 oracle
    (measured 1.3-1.6e-3: at this size the e4m3 operands cost about what bf16 storage costs); gradients: per-tensor rel-L2 to
    fp8emu <= 0.08, median <= 0.03 (measured 0.045-0.050 / 0.010-0.014: delta = rowsum(dO o O) is taken from the kernel's own
    bf16 O, whose rounding differs from the emulation's, and the whole chain below the attention runs in bf16)."""
    from oracle import mca_oracle as O
    cfg = small_config(variant)
    batch = P.data.synthetic_batch(cfg, 6, seed=5, p_drop=0.3)
    sd = P.params.init_state_dict(cfg, seed=3)
    optim = importlib.import_module("mca-paper_amd.optim")
    model = P.MCA(**copy.deepcopy(cfg)); model.load_state_dict(sd, strict=False); model = model.cuda()
    model.engine.set_attention_dtype("fp8")
    opt = optim.FusedAdamW(model, lr=1e-3)
    out = model(to_device(batch, "cuda"))
    opt.zero_grad(); out["loss"].backward()
    torch.cuda.synchronize()
    by_slot = {}
    for k, sl in model.output_slots().items():
        by_slot.setdefault(sl, k)
    pooled = torch.stack([out[by_slot[sl]] for sl in sorted(by_slot)], 1).detach().cpu()
    emu = run_oracle_step(O, cfg, sd, batch, "fp8emu", lr=1e-3, clip=2.0)
    ref = run_oracle_step(O, cfg, sd, batch, "fp32", lr=1e-3, clip=2.0)
    e_emu, e_ref, e_floor = rel_err(pooled, emu["pooled"]), rel_err(pooled, ref["pooled"]), rel_err(emu["pooled"], ref["pooled"])
    assert e_emu < 4e-3, (e_emu, e_floor)
    assert e_ref < 6e-3, (e_ref, e_floor)
    b16 = run_oracle_step(O, cfg, sd, batch, "bf16emu", lr=1e-3, clip=2.0)
    assert rel_err(emu["pooled"], b16["pooled"]) > 5e-4          # the emulation really quantises: it is not the bf16 oracle
    print("fp8 step", variant, "pooled vs fp8emu", e_emu, "vs fp32", e_ref, "emu vs fp32", e_floor, "vs bf16emu", rel_err(pooled, b16["pooled"]))
    assert abs(float(out["loss"]) - emu["loss"]) <= 5e-3 * abs(emu["loss"]) + 1e-3
    errs = []
    for n, p in model.named_parameters():
        gref = emu["grads"][n]
        if gref.abs().max() == 0:
            continue
        errs.append((rel_err(p.grad.cpu(), gref), n))
    errs.sort()
    print("fp8 step", variant, "grad errs median", errs[len(errs) // 2], "max", errs[-1])
    assert errs[-1][0] <= 0.08 and errs[len(errs) // 2][0] <= 0.03, (errs[-1], errs[len(errs) // 2])
    # ... and against the EXACT (fp32 oracle) gradients, so that a systematic loss of gradient quality from e4m3 dO / V cannot
    # hide behind an emulation that restates it: STATED TOLERANCE per tensor 0.15, median 0.06 (the bf16 step's own
    # distance from fp32 at this size: worst tensor a few %; measured values printed)
    errs32 = sorted((rel_err(p.grad.cpu(), ref["grads"][n]), n) for n, p in model.named_parameters() if ref["grads"][n].abs().max() > 0)
    print("fp8 step", variant, "grad errs vs fp32 oracle: median", errs32[len(errs32) // 2], "max", errs32[-1])
    assert errs32[-1][0] <= 0.15 and errs32[len(errs32) // 2][0] <= 0.06, (errs32[-1], errs32[len(errs32) // 2])
    assert model.engine.fp8_backward_on(model.engine.workspace(6), model.engine.N)          # the fp8 backward really ran


def test_dropin_loop_and_no_loss(P):
    """the reference's loop shape (train_accel_gpu.py:108-119) runs unchanged; no_loss returns embeddings only."""
    optim = importlib.import_module("mca-paper_amd.optim")
    cfg = small_config("mca")
    torch.manual_seed(0)
    model = P.MCA(**copy.deepcopy(cfg)).cuda()
    opt = optim.FusedAdamW(model, lr=1e-3)
    batch = to_device(P.data.synthetic_batch(cfg, 4, seed=2, p_drop=0.2), "cuda")
    losses = []
    for _ in range(8):
        outputs = model(batch)
        opt.zero_grad()
        loss = outputs["loss"]
        loss.backward()
        optim.clip_grad_norm_(model, 2.0)
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses                    # it learns
    with torch.no_grad():
        emb = model(batch, no_loss=True)
    assert "losses" not in emb and "fusion" in emb and emb["audio"].shape == (4, 128)
    assert set(outputs.keys()) >= {"audio", "video", "text", "fusion", "losses", "loss", "fcl_loss", "no-fcl_loss", "modality_sample_mask"}


def test_nonfinite_input_raises(P):
    cfg = small_config("mca")
    model = P.MCA(**copy.deepcopy(cfg)).cuda()
    batch = to_device(P.data.synthetic_batch(cfg, 4, seed=2), "cuda")
    batch["audio"]["tokens"][0, 0, 0] = float("nan")
    with pytest.raises(Exception):
        model(batch)


@pytest.mark.parametrize("b,lengths,p_drop", [(8, "full", 0.0), (32, "full", 0.0), (32, "uniform", 0.2)])
def test_forward_bitwise_deterministic_at_cmu_size(P, b, lengths, p_drop):
    """Nothing in the forward accumulates in an order-dependent way (mca_attn_vmean, the value of rows with no valid key -
    dropped modalities - sums in a fixed order since round 2), so repeated forwards must agree BIT FOR BIT.  This is the test that catches a mis-counted s_waitcnt in a pipelined
    kernel: a stale-LDS race shows up as a small fraction of wrong elements that every tolerance-based check lets through,
    and only with the whole chip busy (b = 32)."""
    cfg = P.config.cmu_model_config(batch_size=b)
    torch.manual_seed(43)
    model = P.MCA(**cfg).cuda()
    model.engine.check_finite = False
    data = importlib.import_module("mca-paper_amd.data")
    batch = data.synthetic_batch(cfg, b, seed=1234, lengths=lengths, p_drop=p_drop, device="cuda")
    ref = None
    for _ in range(4):
        with torch.no_grad():
            out = model(batch)
        torch.cuda.synchronize()
        ws = model.engine.workspace(b)
        snap = [ws["pooled"].clone(), out["loss"].clone()] + [a[k].clone() for a in ws["layers"] for k in ("qkv", "o", "x1", "h", "g")]
        if ref is None:
            ref = snap
        else:
            for i, (x, y) in enumerate(zip(ref, snap)):
                assert torch.equal(x, y), f"tensor {i} differs between two forwards of the same inputs"


@pytest.mark.parametrize("variant", ["mca", "mma"])
def test_attention_backward_bitwise_deterministic_at_cmu_size(P, variant):
    """The two-pass attention backward has one owner per output element: with the whole chip busy (b = 32) two launches on
    the same operands agree BIT FOR BIT (dq, dk, dv).  A mis-counted wait in the hand-pipelined fragment reads would show
    here as a few differing elements that every tolerance lets through."""
    b = 32
    cfg = P.config.cmu_model_config(batch_size=b, zorro=variant == "mma")
    cfg["depth"] = 1
    torch.manual_seed(43)
    model = P.MCA(**cfg).cuda()
    eng = model.engine
    eng.check_finite = False
    data = importlib.import_module("mca-paper_amd.data")
    batch = data.synthetic_batch(cfg, b, seed=1234, lengths="uniform", p_drop=0.2, device="cuda")
    out = model(batch)
    out["loss"].backward()
    ws = eng.workspace(b)
    a = ws["layers"][0]
    N, D = eng.N, eng.D
    snaps = []
    for rep in range(3):
        a["dqkv"].fill_(7.0)
        eng._attn_bwd2(a["qkv"].data_ptr(), N * 3 * D, 3 * D, a["qkv"], D, 2 * D, 3 * D, a["o"], ws["do"], a["lse"], ws["delta"],
                       a["dqkv"].data_ptr(), N * 3 * D, 3 * D, False, a["dqkv"], D, 2 * D, 3 * D, eng.qmask_attn, eng.sched_attn_f,
                       eng.sched_attn_b2, ws, b, N)
        torch.cuda.synchronize()
        snaps.append(a["dqkv"].clone())
    assert torch.isfinite(snaps[0].float()).all()
    # dq, dk AND dv: since round 3 dvmean (gradient of the uniform rows' mean(V), 20 % of the modalities are dropped here) is
    # summed in row order by mca_attn_bwd_prep, not with atomics
    assert bool(torch.isinf(a["lse"]).any())          # uniform rows exist: the dvmean path is exercised
    for s_ in snaps[1:]:
        assert torch.equal(s_, snaps[0])


@pytest.mark.parametrize("kind,b", [("cmu", 32), ("mma", 32), ("tcga", 16)])
def test_full_size_gradients_new_vs_conservative_kernels(P, kind, b):
    """Whole-chip cross-check of the pipelined kernels (b = 32, every CU busy): one forward + backward with the production
    kernels (persistent / fused GEMMs, 256x256 weight-gradient tiles, XCD-remapped order) against the same step with the
    conservative ones (one tile per workgroup, unfused GEGLU / LayerNorm residual, 256x128 weight gradients, launch order, the
    two-pass attention backward instead of the one-pass kernel).
    Gradients only differ by rounding placement and the order of fp32 atomics; a pipeline race (stale LDS, mis-counted wait)
    shows as percent-level error on the tensors fed by the broken kernel."""
    hipm = importlib.import_module("mca-paper_amd.hip")
    data = importlib.import_module("mca-paper_amd.data")
    cfg = {"cmu": lambda: P.config.cmu_model_config(batch_size=b), "mma": lambda: P.config.cmu_model_config(batch_size=b, zorro=True),
           "tcga": lambda: P.config.tcga_model_config(batch_size=b)}[kind]()
    batch = data.synthetic_batch(cfg, b, seed=1234, lengths="uniform", p_drop=0.2, device="cuda")

    def run(conservative):
        for key, val in ((7, 1), (5, 2), (9, 16)):
            hipm.lib().mca_debug_set(key, val if conservative else 0)
        torch.manual_seed(43)
        model = P.MCA(**cfg).cuda()
        eng = model.engine
        eng.check_finite = False
        eng.fuse_ln_residual = eng.fuse_geglu_bwd = not conservative
        if conservative:
            eng.dbg["onepass"] = False          # the two-pass attention backward against the one-pass production form
        out = model(batch)
        out["loss"].backward()
        torch.cuda.synchronize()
        return float(out["loss"]), {n: p.grad.clone() for n, p in model.named_parameters()}

    try:
        l_new, g_new = run(False)
        l_new2, g_new2 = run(False)
        l_old, g_old = run(True)
    finally:
        for key in (7, 5, 9):
            hipm.lib().mca_debug_set(key, 0)
    # dropped modalities give rows with no valid key: their output is mean(V), summed with fp32 atomics (order-dependent);
    # observed spread of the loss up to 5.2e-5 relative (the logits are O(10^4)); the races this test caught moved it by 1e-2
    assert abs(l_new - l_new2) <= 2e-4 * abs(l_new)
    assert abs(l_new - l_old) <= 2e-3 * abs(l_old)           # bf16 rounding placement differs (fused vs unfused epilogues)
    # The two paths round in different places (fused FF1+GEGLU epilogue, fp32 vs bf16 dg): pooled embeddings differ by a few
    # 1e-4 and the temperature-14 contrastive softmax turns that into a uniform ~1-3 % on every gradient (measured:
    # tools/diag_new_vs_old.py; same envelope as TOL_GRAD).  A stale-LDS race in one GEMM gave >= 10 % downstream of it.
    ds = []
    for n in g_new:
        noise = rel_err(g_new2[n], g_new[n])
        d = rel_err(g_new[n], g_old[n])
        ds.append(d)
        assert d <= 5 * noise + 8e-2, (n, d, noise)
        assert noise <= 8e-3, (n, noise)          # two production steps: only fp32 atomic order differs (worst of 40 repeats: 3e-3)
    ds.sort()
    assert 0.0 < ds[len(ds) // 2] <= 4e-2, ds[len(ds) // 2]

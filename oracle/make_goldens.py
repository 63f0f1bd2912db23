"""TEST INFRASTRUCTURE — generates tests/golden/*.pt by running the REFERENCE itself.

Run only in the build container (needs /root/reference; never on the GPU box):
    cd /tmp && python /root/repo/oracle/make_goldens.py [--cmu]

The reference's ``model.py`` imports the third-party ``torchmultimodal`` package, which is not installed
here.  Its loss lives, adapted, in the reference's own ``utils/contrastive_loss_with_temperature.py``
and ``utils/distributed.py``; the shim below only re-routes the import to those files (SURVEY.md
Appendix A).  No reference source is copied: the fixtures written are inputs, weights and outputs.

Fixtures
  tiny_<case>.pt : 3 modalities (2 embedded-sequence + 1 tabular), dim 32, 2 heads x 16, depth 2, 8 fusion
                   tokens, combos [3,2]; inputs, initial state_dict, every output, every gradient,
                   grad-norm, weights after 1 and 2 AdamW steps (lr 1e-3, clip 2.0).
  collators.pt   : the reference's MultimodalCollator (sequence / embedded_sequence / matrix) on ragged samples with
                   missing modalities: input samples and collated batch.
  tcga_b2.pt     : TCGA_config1-shaped config (4 tabular modalities, N=2548, 60 loss terms) at b=2, same recipe as cmu_*.
  ref_state/, ref_state_io.pt : a state directory in the layout of the reference's accelerator.save_state (model.safetensors,
                   optimizer.bin, scheduler.bin) after two reference training steps on a native-sized small model, the
                   embeddings / masks the reference's inference loop produces from it, and the reference's third step.
  cmu_<case>.pt  : CMU-shaped config (N=2538, D=512, L=5) at b=2 with weights from the build's own
                   deterministic initialiser; pooled embeddings, loss terms, per-parameter grad norms.
"""
import argparse
import os
import sys
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    for name in ("torchmultimodal", "torchmultimodal.utils", "torchmultimodal.modules",
                 "torchmultimodal.modules.losses"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    import utils.distributed as ud
    sys.modules["torchmultimodal.utils.distributed"] = ud
    import utils.contrastive_loss_with_temperature as cl
    cl.xm = type("XM", (), {"get_ordinal": staticmethod(dist.get_rank)})
    sys.modules["torchmultimodal.modules.losses.contrastive_loss_with_temperature"] = cl
    torch.save_real = torch.save
    torch.save = lambda *a, **k: None          # model.py:94 debug write
    import model as refmodel
    import encoders as refenc
    torch.save = torch.save_real
    return refmodel, refenc


# ---------------------------------------------------------------------------------------------------
def tiny_model_config(variant: str):
    enc = {
        "seqA": {"type": "EmbeddedSequenceEncoder", "input_size": 10, "max_tokens": 12, "embedding_dim": 32},
        "seqB": {"type": "EmbeddedSequenceEncoder", "input_size": 7, "max_tokens": 9, "embedding_dim": 32},
        "tab": {"type": "TabularEncoder", "num_embeddings": 11, "max_tokens": 11, "max_value": 100,
                "embedding_dim": 32},
    }
    cfg = dict(encoder_configs=enc, dim=32, depth=2, heads=2, dim_head=16, ff_mult=4, num_fusion_tokens=8,
               batch_size=4, fcl=True, fcl_root=[0, 1, 2], bimodal_contrastive=False, non_fusion_fcl=False,
               fusion_combos=[3, 2], zorro=False, eao=False, no_fusion=False, mean_pool=False)
    if variant == "zorro":
        cfg.update(zorro=True, fcl=False)
    elif variant == "bimodal":
        cfg.update(bimodal_contrastive=True, non_fusion_fcl=True)
    elif variant == "nofcl":
        cfg.update(fcl=False)
    return cfg


def tiny_batch(seed: int, drop: dict, b: int = 4):
    """drop: {modality: [sample indices dropped]}.  Layouts follow the reference collators
    (encoders.py:300-343): embedded_sequence -> tokens (b,pad,in) f32 zero-filled + bool mask (True = pad);
    sequence -> values (b,n) with -10000 fill + int64 mask."""
    g = torch.Generator().manual_seed(seed)
    batch = {}
    for name, n, width in (("seqA", 12, 10), ("seqB", 9, 7)):
        toks = torch.zeros(b, n, width)
        mask = torch.ones(b, n, dtype=torch.bool)
        for i in range(b):
            if i in drop.get(name, []):
                continue
            ln = int(torch.randint(1, n + 1, (1,), generator=g))
            toks[i, :ln] = torch.randn(ln, width, generator=g)
            mask[i, :ln] = False
        batch[name] = {"tokens": toks, "attention_mask": mask}
    vals = torch.randn(b, 11, generator=g)
    vals[0, 3] = -1.0            # value-path padding sentinel (encoders.py:63,88)
    vals[1, 5] = 250.0           # above max_value -> clamped
    vals[2, 7] = -10000.0        # a missing entry inside a present sample
    for i in drop.get("tab", []):
        vals[i] = -10000.0
    batch["tab"] = {"values": vals, "attention_mask": (vals == -10000).to(torch.long)}
    return batch


def perturb_(model, seed):
    """make gammas/betas/biases non-trivial and give the embedding rows norms on both sides of max_norm."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma") or (".token_encoder." in n and n.endswith("weight") and p.dim() == 1) \
                    or n.endswith("norm.weight"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("embedding.weight"):
                p[::2].mul_(0.1)


def run_case(refmodel, cfg, batch, seed, lr, clip, steps=2, cls="MCA"):
    torch.manual_seed(seed)
    torch.save_real = getattr(torch, "save_real", torch.save)
    real_save = torch.save
    torch.save = lambda *a, **k: None
    try:
        model = getattr(refmodel, cls)(**cfg)
        perturb_(model, seed + 1)
        init_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=lr)
        rec = {"config": cfg, "batch": batch, "init_state": init_sd, "lr": lr, "clip": clip}
        for s in range(steps):
            out = model(batch)
            opt.zero_grad()
            out["loss"].backward()
            if s == 0:
                rec["outputs"] = {
                    "embeddings": {("|".join(map(str, sorted(k))) if isinstance(k, frozenset) else k): v.detach().clone()
                                   for k, v in out.items()
                                   if isinstance(v, torch.Tensor) and v.dim() == 2},
                    "losses": {k: v.detach().clone() for k, v in out["losses"].items()},
                    "loss": out["loss"].detach().clone(),
                    "modality_sample_mask": {k: v.clone() for k, v in out["modality_sample_mask"].items()},
                }
                for extra in ("fcl_loss", "no-fcl_loss"):
                    if extra in out:
                        rec["outputs"][extra] = out[extra].detach().clone()
                rec["grads"] = {n: (p.grad.detach().clone() if p.grad is not None else None)
                                for n, p in model.named_parameters()}
                rec["state_after_forward"] = {k: v.detach().clone() for k, v in model.state_dict().items()
                                              if k.endswith("embedding.weight") or k.endswith("logit_scale")}
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            if s == 0:
                rec["grad_norm"] = gn.detach().clone()
            opt.step()
            rec[f"state_step{s + 1}"] = {k: v.detach().clone() for k, v in model.state_dict().items()
                                         if k in dict(model.named_parameters())}
        if cls == "MCA":
            rec["attn_mask"] = model.attn_mask.clone()
            rec["pool_mask"] = model.pool_mask.clone()
        rec["token_types"] = model.token_types.clone()
        rec["return_token_types"] = list(model.return_token_types)
        rec["loss_names"] = list(out["losses"].keys())
    finally:
        torch.save = real_save
    return rec


TINY_CASES = {
    # name: (variant, drop map, seed)
    "mca_fcl": ("mca", {}, 11),
    "mca_fcl_drop": ("mca", {"seqA": [1], "tab": [2]}, 12),
    "mca_fcl_nan": ("mca", {"seqB": [0, 1, 2, 3]}, 13),
    "zorro_drop": ("zorro", {"seqA": [0], "seqB": [0, 3]}, 14),
    "bimodal_drop": ("bimodal", {"seqB": [2], "tab": [0, 2]}, 15),
    "nofcl": ("nofcl", {"seqA": [3]}, 16),
}


def make_tiny(refmodel):
    for name, (variant, drop, seed) in TINY_CASES.items():
        cfg = tiny_model_config(variant)
        batch = tiny_batch(seed, drop)
        rec = run_case(refmodel, cfg, batch, seed, lr=1e-3, clip=2.0)
        torch.save(rec, os.path.join(GOLD, f"tiny_{name}.pt"))
        print(f"tiny_{name}: loss {float(rec['outputs']['loss']):.6f} terms {len(rec['outputs']['losses'])} "
              f"nan {sum(int(torch.isnan(v)) for v in rec['outputs']['losses'].values())} gn {float(rec['grad_norm']):.4f}")


EAO_CASES = {
    # name: (fcl, bimodal, non_fusion_fcl, fusion_combos, drop map, seed)
    "eao_fcl": (True, True, True, [2], {}, 31),
    "eao_fcl_drop": (True, True, True, [2], {"seqA": [1], "tab": [2, 3]}, 32),
    "eao_nofcl_drop": (False, False, False, [3, 2], {"seqB": [0, 2]}, 33),
}


def make_eao(refmodel):
    """The reference's EAO baseline (model.py:481-596) on the tiny configs: outputs, losses, gradients, two AdamW steps; and
    the CMU_config1_EAO shape at b = 2 with the build's own initialiser (pooled embeddings, losses, gradient norms)."""
    # (the CMU-shaped model first: the reference's default logit_scale is ONE module-level Parameter shared by every loss
    # instance, so a model trained earlier in this process would move the 'initial' temperature recorded below)
    import importlib
    sys.path.insert(0, REPO)
    pkg = importlib.import_module("mca-paper_amd")
    cfg = pkg.config.cmu_eao_model_config(batch_size=2)
    torch.manual_seed(43)
    real_save = torch.save
    torch.save = lambda *a, **k: None
    try:
        model = refmodel.EAO(**cfg)
        sd = pkg.params.init_state_dict(cfg, seed=43)
        assert set(sd.keys()) == set(model.state_dict().keys()), set(sd.keys()) ^ set(model.state_dict().keys())
        ref_sd = model.state_dict()
        sums = {k: (float(v.double().sum()), float(v.double().abs().sum()), tuple(v.shape)) for k, v in ref_sd.items() if v.dtype.is_floating_point}
        same = all(torch.equal(v, ref_sd[k]) for k, v in sd.items())
        model.load_state_dict(sd)
        batch = pkg.data.synthetic_batch(cfg, batch_size=2, seed=1234, p_drop=0.3, lengths="uniform")
        out = model(batch)
        out["loss"].backward()
    finally:
        torch.save = real_save
    names = list(cfg["encoder_configs"].keys())
    rec = {
        "case": "eao", "seed": 43, "data_seed": 1234, "p_drop": 0.3, "same_seed_init_equal": same, "init_checksums": sums,
        "pooled": torch.stack([out[n] for n in names] + [out[k] for k in model.fusion_combos], 1).detach(),
        "losses": {k: v.detach() for k, v in out["losses"].items()},
        "loss": out["loss"].detach(),
        "sample_mask": {k: v for k, v in out["modality_sample_mask"].items()},
        "grad_norms": {n: float(p.grad.norm()) for n, p in model.named_parameters()},
        "grad_slices": {n: p.grad.flatten()[:64].clone() for n, p in model.named_parameters()},
        "state_keys": list(model.state_dict().keys()),
    }
    torch.save(rec, os.path.join(GOLD, "cmu_eao_b2.pt"))
    print(f"cmu_eao: loss {float(rec['loss']):.5f}, {len(rec['losses'])} terms, pooled {tuple(rec['pooled'].shape)}, same-seed init equal: {same}")
    for name, (fcl, bimodal, nff, combos, drop, seed) in EAO_CASES.items():
        cfg = tiny_model_config("mca")
        cfg.update(fcl=fcl, bimodal_contrastive=bimodal, non_fusion_fcl=nff, fusion_combos=combos, eao=True, no_fusion=True,
                   mean_pool=True, fcl_root=[0, 1])
        batch = tiny_batch(seed, drop)
        rec = run_case(refmodel, cfg, batch, seed, lr=1e-3, clip=2.0, cls="EAO")
        torch.save(rec, os.path.join(GOLD, f"tiny_{name}.pt"))
        print(f"tiny_{name}: loss {float(rec['outputs']['loss']):.6f} terms {len(rec['outputs']['losses'])} "
              f"nan {sum(int(torch.isnan(v)) for v in rec['outputs']['losses'].values())} gn {float(rec['grad_norm']):.4f}")


def make_init_parity(refmodel):
    """Same torch seed -> the reference's own initial weights for the CMU config; only per-tensor
    checksums are stored (the weights are 70 MB)."""
    import importlib
    sys.path.insert(0, REPO)
    pkg = importlib.import_module("mca-paper_amd")
    cfg = pkg.config.cmu_model_config(batch_size=2)
    torch.manual_seed(43)
    real_save = torch.save
    m = refmodel.MCA(**cfg)
    sums = {k: (float(v.double().sum()), float(v.double().abs().sum()), tuple(v.shape))
            for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    torch.save({"seed": 43, "checksums": sums, "keys": list(m.state_dict().keys())},
               os.path.join(GOLD, "cmu_init_checksums.pt"))
    print("cmu init checksums:", len(sums), "tensors")


def make_cmu(refmodel):
    """CMU-shaped runs at b=2: weights from the build's own initialiser, inputs from its synthetic
    generator; store pooled embeddings, losses, grad norms and a few gradient slices."""
    import importlib
    sys.path.insert(0, REPO)
    pkg = importlib.import_module("mca-paper_amd")
    for case, (zorro, p_drop) in {"mca": (False, 0.0), "mma_d40": (True, 0.4)}.items():
        cfg = pkg.config.cmu_model_config(batch_size=2, zorro=zorro)
        torch.manual_seed(43)
        real_save = torch.save
        torch.save = lambda *a, **k: None
        try:
            model = refmodel.MCA(**cfg)
            sd = pkg.params.init_state_dict(cfg, seed=43)
            missing = model.load_state_dict(sd, strict=False)
            assert not missing.unexpected_keys, missing
            batch = pkg.data.synthetic_batch(cfg, batch_size=2, seed=1234, p_drop=p_drop, lengths="uniform")
            out = model(batch)
            out["loss"].backward()
        finally:
            torch.save = real_save
        names = list(cfg["encoder_configs"].keys())
        rec = {
            "case": case, "seed": 43, "data_seed": 1234, "p_drop": p_drop,
            "pooled": torch.stack([out[n] for n in names] +
                                  ([out[k] for k in model.fusion_combos] if (cfg["fcl"] and not zorro) else [out["fusion"]]), 1).detach(),
            "losses": {k: v.detach() for k, v in out["losses"].items()},
            "loss": out["loss"].detach(),
            "sample_mask": {k: v for k, v in out["modality_sample_mask"].items()},
            "grad_norms": {n: float(p.grad.norm()) for n, p in model.named_parameters()},
            "grad_slices": {n: p.grad.flatten()[:64].clone() for n, p in model.named_parameters()},
        }
        torch.save(rec, os.path.join(GOLD, f"cmu_{case}_b2.pt"))
        print(f"cmu_{case}: loss {float(rec['loss']):.5f}")


def make_cmu_autocast(refmodel):
    """The reference's OWN bf16 behaviour at CMU shape, b = 2: the cases of make_cmu with the forward under
    torch.autocast("cpu", torch.bfloat16) - what Accelerate's mixed_precision="bf16" does to it (reference model.py:448-478 runs
    unchanged; the loss and the backward see the autocast graph).  Stored beside the fp32 goldens: pooled embeddings, loss terms,
    per-tensor gradient norms.  The GPU tests bound the native step's distance to the fp32 golden by 2 x the reference's own
    bf16-vs-fp32 distance, per quantity."""
    import importlib
    sys.path.insert(0, REPO)
    pkg = importlib.import_module("mca-paper_amd")
    for case, (zorro, p_drop) in {"mca": (False, 0.0), "mma_d40": (True, 0.4)}.items():
        cfg = pkg.config.cmu_model_config(batch_size=2, zorro=zorro)
        torch.manual_seed(43)
        real_save = torch.save
        torch.save = lambda *a, **k: None
        try:
            model = refmodel.MCA(**cfg)
            sd = pkg.params.init_state_dict(cfg, seed=43)
            missing = model.load_state_dict(sd, strict=False)
            assert not missing.unexpected_keys, missing
            batch = pkg.data.synthetic_batch(cfg, batch_size=2, seed=1234, p_drop=p_drop, lengths="uniform")
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = model(batch)
            out["loss"].float().backward()
        finally:
            torch.save = real_save
        names = list(cfg["encoder_configs"].keys())
        rec = {
            "case": case, "seed": 43, "data_seed": 1234, "p_drop": p_drop, "autocast": "cpu bfloat16",
            "pooled": torch.stack([out[n] for n in names] +
                                  ([out[k] for k in model.fusion_combos] if (cfg["fcl"] and not zorro) else [out["fusion"]]), 1).detach().float(),
            "losses": {k: v.detach().float() for k, v in out["losses"].items()},
            "loss": out["loss"].detach().float(),
            "grad_norms": {n: float(p.grad.float().norm()) for n, p in model.named_parameters()},
        }
        torch.save(rec, os.path.join(GOLD, f"cmu_{case}_b2_autocast.pt"))
        print(f"cmu_{case} under autocast: loss {float(rec['loss']):.5f}")


def make_tcga(refmodel):
    """TCGA_config1-shaped run (4 TabularEncoder modalities, N = 2548, 60 loss terms) at b=2 through the reference."""
    import importlib
    sys.path.insert(0, REPO)
    pkg = importlib.import_module("mca-paper_amd")
    cfg = pkg.config.tcga_model_config(batch_size=2)
    torch.manual_seed(43)
    real_save = torch.save
    torch.save = lambda *a, **k: None
    try:
        model = refmodel.MCA(**cfg)
        sd = pkg.params.init_state_dict(cfg, seed=43)
        missing = model.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys, missing
        batch = pkg.data.synthetic_batch(cfg, batch_size=2, seed=77, p_drop=0.25)
        out = model(batch)
        out["loss"].backward()
    finally:
        torch.save = real_save
    names = list(cfg["encoder_configs"].keys())
    rec = {
        "case": "tcga", "seed": 43, "data_seed": 77, "p_drop": 0.25,
        "pooled": torch.stack([out[n] for n in names] + [out[k] for k in model.fusion_combos], 1).detach(),
        "losses": {k: v.detach() for k, v in out["losses"].items()},
        "loss": out["loss"].detach(),
        "sample_mask": {k: v for k, v in out["modality_sample_mask"].items()},
        "grad_norms": {n: float(p.grad.norm()) for n, p in model.named_parameters()},
        "grad_slices": {n: p.grad.flatten()[:64].clone() for n, p in model.named_parameters()},
    }
    torch.save(rec, os.path.join(GOLD, "tcga_b2.pt"))
    print(f"tcga: loss {float(rec['loss']):.5f}, {len(rec['losses'])} terms, pooled {tuple(rec['pooled'].shape)}")


def make_ckpt(refmodel):
    """A state directory as the reference's ``accelerator.save_state`` lays it out (train_accel_gpu.py:122-123):
    model.safetensors (``safetensors.torch.save_file(model.state_dict())``, which is what Accelerate's save_model does),
    optimizer.bin (``torch.save(AdamW.state_dict())``) and scheduler.bin (``get_scheduler('cosine')`` state), written after
    two training steps of the reference on a native-sized small model (dim 128 = 2 heads x 64, mixed sequence / tabular
    encoders), plus what the reference's inference loop (infer_accel_gpu.py:97-136) produces from that state:
    embeddings and modality masks of an eval batch.  The third optimizer step is recorded too, so a resumed native run
    can be checked against the reference continuing from the same state."""
    import importlib
    from safetensors.torch import save_file
    from transformers import get_scheduler
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    pkg = importlib.import_module("mca-paper_amd")
    from util_small import small_config
    cfg = small_config("tab", depth=1)
    out_dir = os.path.join(GOLD, "ref_state")
    os.makedirs(out_dir, exist_ok=True)
    real_save = torch.save
    torch.save = lambda *a, **k: None
    try:
        torch.manual_seed(5)
        model = refmodel.MCA(**cfg)
        perturb_(model, 6)
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        sched = get_scheduler(name="cosine", optimizer=opt, num_warmup_steps=2, num_training_steps=10)
        batches = [pkg.data.synthetic_batch(cfg, 4, seed=50 + i, p_drop=0.3) for i in range(3)]
        for i in range(2):
            out = model(batches[i])
            opt.zero_grad()
            out["loss"].backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0)
            opt.step(); sched.step()
        save_file({k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}, os.path.join(out_dir, "model.safetensors"))
        real_save(opt.state_dict(), os.path.join(out_dir, "optimizer.bin"))
        real_save(sched.state_dict(), os.path.join(out_dir, "scheduler.bin"))
        lr_next = opt.param_groups[0]["lr"]
        # inference from that state (eval mode, no grad), as infer_accel_gpu.py:97-136 collects it
        model.eval()
        eval_batch = pkg.data.synthetic_batch(cfg, 4, seed=99, p_drop=0.3)
        with torch.no_grad():
            o = model(eval_batch)
        emb = {("|".join(map(str, sorted(k))) if isinstance(k, frozenset) else k): v.detach().clone()
               for k, v in o.items() if isinstance(v, torch.Tensor) and v.dim() == 2}
        masks = {k: v.clone() for k, v in o["modality_sample_mask"].items()}
        # the reference continues: third step from the saved state
        model.train()
        out = model(batches[2])
        opt.zero_grad()
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0)
        opt.step(); sched.step()
        after3 = {n: p.detach().clone() for n, p in model.named_parameters()}
    finally:
        torch.save = real_save
    torch.save({"config": cfg, "train_batches": batches, "eval_batch": eval_batch, "embeddings": emb, "masks": masks,
                "lr_step3": lr_next, "loss_step3": out["loss"].detach().clone(), "state_step3": after3},
               os.path.join(GOLD, "ref_state_io.pt"))
    print("wrote ref_state/ and ref_state_io.pt:", {k: os.path.getsize(os.path.join(out_dir, k)) for k in os.listdir(out_dir)})


def make_unrunnable(refmodel):
    """Configurations of the reference that raise in the reference itself (recorded, not worked around)."""
    import json
    real_save = torch.save
    torch.save = lambda *a, **k: None
    rec = {}
    try:
        cfg = tiny_model_config("mca"); cfg["mean_pool"] = True
        try:
            refmodel.MCA(**cfg)(tiny_batch(3, {}))
            rec["MCA(mean_pool=True).forward"] = {"type": None, "message": "ran"}
        except Exception as e:          # model.py:264 `if self.token_types` on an N-element tensor
            rec["MCA(mean_pool=True).forward"] = {"type": type(e).__name__, "message": str(e), "where": "model.py:264 MeanTokenProjectionPool.forward"}
    finally:
        torch.save = real_save
    json.dump(rec, open(os.path.join(GOLD, "ref_unrunnable.json"), "w"), indent=1)
    print("ref_unrunnable:", rec)


def make_collators(refenc):
    """The reference's collators (encoders.py:286-403) on ragged samples with missing modalities, NaNs, over-long rows."""
    g = torch.Generator().manual_seed(7)
    cfg = {"seq": {"type": "embedded_sequence", "pad_len": 6, "embedding_size": 3, "data_col_name": "data"},
           "tab": {"type": "sequence", "pad_len": 5, "data_col_name": "values", "pad_token": -10000, "other_col": "extra"},
           "ids": {"type": "sequence", "pad_len": 7, "data_col_name": "indices", "pad_token": 0},
           "mat": {"type": "matrix", "pad_len": 4, "max_channels": 3}}
    def sample(i):
        s = {}
        n = int(torch.randint(0, 10, (1,), generator=g))
        x = torch.randn(n, 3, generator=g)
        if i == 2 and n: x[0, 0] = float("nan")
        s["seq"] = {"data": None if i == 1 else x}
        v = torch.randn(5, generator=g)
        if i == 3: v[1] = -10000.0
        s["tab"] = {"values": None if i == 4 else v, "extra": torch.arange(5.0)}
        k = int(torch.randint(0, 8, (1,), generator=g))
        s["ids"] = {"indices": None if i == 0 else torch.randint(1, 50, (k,), generator=g)}
        r = int(torch.randint(1, 5, (1,), generator=g))
        s["mat"] = {"values": torch.randn(r, 5, generator=g)}
        s["labels"] = {"y": torch.tensor(float(i))}
        return s
    samples = [sample(i) for i in range(6)]
    import copy
    out = refenc.MultimodalCollator(copy.deepcopy(cfg), labels="labels")(copy.deepcopy(samples))
    # a missing matrix sample only stacks when its placeholder shape (max_channels, pad_len) matches: square case
    cfg_sq = {"mat": {"type": "matrix", "pad_len": 3, "max_channels": 3}}
    samples_sq = [{"mat": {"values": torch.randn(2, 3, generator=g)}}, {"mat": {"values": None}}]
    out_sq = refenc.MultimodalCollator(copy.deepcopy(cfg_sq))(copy.deepcopy(samples_sq))
    torch.save({"config": cfg, "samples": samples, "out": {k: dict(v) for k, v in out.items()},
                "config_sq": cfg_sq, "samples_sq": samples_sq, "out_sq": {k: dict(v) for k, v in out_sq.items()}},
               os.path.join(GOLD, "collators.pt"))
    print("wrote collators.pt")


FLAGS = ("collators", "tcga", "cmu", "cmu_autocast", "init", "tiny", "ckpt", "eao")


if __name__ == "__main__":
    # ONE flag per process.  The reference's loss keeps its temperature in a module-level nn.Parameter shared by every model
    # of the process (utils/contrastive_loss_with_temperature.py:111,158: DEFAULT_LOGIT_SCALE), so every case starts from the
    # temperature the previous cases of the SAME process trained: the committed files were written one flag at a time and only
    # reproduce bit for bit that way (`--eao --tiny` in one process gave tiny_mca_* files with logit_scale 2.6532 instead of
    # 2.6593).  --all runs every flag in a process of its own.
    ap = argparse.ArgumentParser()
    for f in FLAGS:
        ap.add_argument("--" + f, action="store_true")
    ap.add_argument("--all", action="store_true", help="every flag, each in its own process")
    a = ap.parse_args()
    chosen = [f for f in FLAGS if getattr(a, f)]
    if a.all:
        if chosen:
            ap.error("--all takes no other flag")
        import subprocess
        for f in FLAGS:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--" + f], check=True)
        sys.exit(0)
    if len(chosen) > 1:
        ap.error(f"one flag per process (got {chosen}): the reference shares its loss temperature between the models of a process, "
                 "so the order of the cases changes the goldens; use --all")
    if not chosen:
        chosen = ["tiny"]
    os.makedirs(GOLD, exist_ok=True)
    refmodel, refenc = import_reference()
    flag = chosen[0]
    if flag == "collators":
        make_collators(refenc)
    elif flag == "tcga":
        make_tcga(refmodel)
    elif flag == "ckpt":
        make_ckpt(refmodel)
        make_unrunnable(refmodel)
    elif flag == "eao":
        make_eao(refmodel)
    elif flag == "tiny":
        make_tiny(refmodel)
    elif flag == "init":
        make_init_parity(refmodel)
    elif flag == "cmu":
        make_cmu(refmodel)
    elif flag == "cmu_autocast":
        make_cmu_autocast(refmodel)

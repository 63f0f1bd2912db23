"""Helpers shared by the GPU parity tests, smoke() and bench: a small configuration the native kernels
accept (dim = heads*64) and runners for the native step and the oracle step."""
import copy

import torch


def small_config(variant="mca", depth=2):
    enc = {
        "audio": {"type": "EmbeddedSequenceEncoder", "input_size": 10, "max_tokens": 70, "embedding_dim": 128},
        "video": {"type": "EmbeddedSequenceEncoder", "input_size": 7, "max_tokens": 45, "embedding_dim": 128},
        "text": {"type": "EmbeddedSequenceEncoder", "input_size": 20, "max_tokens": 30, "embedding_dim": 128},
    }
    cfg = dict(encoder_configs=enc, dim=128, depth=depth, heads=2, dim_head=64, ff_mult=4, num_fusion_tokens=8,
               batch_size=4, fcl=True, fcl_root=[0, 1, 2], bimodal_contrastive=False, non_fusion_fcl=False,
               fusion_combos=[3, 2], zorro=False, eao=False, no_fusion=False, mean_pool=False)
    if variant == "tab":        # mixed encoders: the middle modality is a dense table (TCGA-style)
        enc["video"] = {"type": "TabularEncoder", "num_embeddings": 45, "max_tokens": 45, "max_value": 100, "embedding_dim": 128}
        cfg.update(bimodal_contrastive=True, non_fusion_fcl=True)
    if variant == "zorro":
        cfg.update(zorro=True, fcl=False)
    elif variant == "bimodal":
        cfg.update(bimodal_contrastive=True, non_fusion_fcl=True)
    elif variant in ("eao", "eao_tab"):          # the EAO baseline (reference configs/CMU_config1_EAO.yaml flags)
        if variant == "eao_tab":
            enc["video"] = {"type": "TabularEncoder", "num_embeddings": 45, "max_tokens": 45, "max_value": 100, "embedding_dim": 128}
        cfg.update(eao=True, no_fusion=True, mean_pool=True, fcl=True, fcl_root=[0, 1], fusion_combos=[2], bimodal_contrastive=True,
                   non_fusion_fcl=True)
    return cfg


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def to_device(batch, dev):
    return {k: {kk: vv.to(dev) for kk, vv in v.items()} for k, v in batch.items()}


def run_native_step(pkg, cfg, sd, batch, lr=1e-3, clip=2.0, steps=1, device="cuda"):
    optim = __import__("importlib").import_module("mca-paper_amd.optim")
    model = pkg.build_model(copy.deepcopy(cfg))          # MCA, or EAO when cfg["eao"]
    model.load_state_dict(sd, strict=False)
    model = model.to(device)
    opt = optim.FusedAdamW(model, lr=lr)
    dbatch = to_device(batch, device)
    out = None
    for s in range(steps):
        out = model(dbatch)
        opt.zero_grad()
        out["loss"].backward()
        if s == 0:
            first = out
            grads = {n: p.grad.detach().clone().cpu() for n, p in model.named_parameters()}
        gn = optim.clip_grad_norm_(model, clip)
        if s == 0:
            gn0 = float(gn)
        opt.step()
    torch.cuda.synchronize()
    slots = model.output_slots()
    R = model.max_return_tokens
    by_slot = {}
    for k, sl in slots.items():
        by_slot.setdefault(sl, k)
    pooled = torch.stack([first[by_slot[sl]] for sl in sorted(by_slot)], 1)
    return dict(pooled=pooled.detach().cpu(), loss=float(first["loss"]),
                losses={k: float(v) for k, v in first["losses"].items()}, grads=grads, grad_norm=gn0,
                state={n: p.detach().clone().cpu() for n, p in model.named_parameters()}, model=model)


def run_oracle_step(O, cfg, sd, batch, mode="fp32", lr=1e-3, clip=2.0, steps=1):
    ocfg = copy.deepcopy(cfg)
    S = O.EAOStructure(ocfg) if ocfg.get("eao") else O.Structure(ocfg)
    sd = {k: v.clone() for k, v in sd.items()}
    opt = None
    for s in range(steps):
        out, grads, gn, opt = O.train_step(S, sd, batch, mode, lr=lr, clip=clip, opt_state=opt)
        if s == 0:
            first, g0, gn0 = out, grads, float(gn)
    names = S.modalities
    nslots = len(names) + (len(S.combos) if S.do_fcl else (0 if S.no_fusion else 1))
    pooled = first["pooled"][:, :nslots]
    return dict(pooled=pooled.detach(), pooled_full=first["pooled"].detach(), loss=float(first["loss"]),
                losses={k: float(v) for k, v in first["losses"].items()}, grads=g0, grad_norm=gn0,
                state={k: v.detach().clone() for k, v in sd.items() if O.is_param(k)})

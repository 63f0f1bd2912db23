"""How close the small-step parity test sits to its thresholds: median / max gradient error against the fp32 oracle and the
bf16-emulating oracle, for every parametrisation, repeated (atomic order changes the rounding pattern run to run)."""
import importlib, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from util_small import small_config, run_native_step, run_oracle_step, rel_err
P = importlib.import_module("mca-paper_amd")
O = importlib.import_module("oracle.mca_oracle")
H = importlib.import_module("mca-paper_amd.hip")
# (forward attention form: MCA_DEBUG=lazy_softmax=0 for the textbook recurrence; the lazy reference is the default)
for variant, p_drop in [("mca", 0.0), ("mca", 0.35), ("zorro", 0.35), ("bimodal", 0.35), ("tab", 0.35)]:
    cfg = small_config(variant)
    batch = P.data.synthetic_batch(cfg, 6, seed=5, p_drop=p_drop)
    if variant == "tab":
        v = batch["video"]["values"]
        v[0, 3], v[1, 5], v[2, 7] = -1.0, 250.0, -10000.0
        batch["video"]["attention_mask"] = (v == -10000).to(torch.long)
    sd = P.params.init_state_dict(cfg, seed=3)
    for k in sd:
        if k.endswith("embedding.weight"):
            sd[k][::2] *= 0.05
    g = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith("gamma") or k.endswith("bias") or ("token_encoder" in k and sd[k].dim() == 1):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    ref = run_oracle_step(O, cfg, sd, batch, "fp32", lr=1e-3, clip=2.0)
    emu = run_oracle_step(O, cfg, sd, batch, "bf16emu", lr=1e-3, clip=2.0)
    meds, maxs, pooled, worst_ratio = [], [], [], []
    for rep in range(3):
        nat = run_native_step(P, cfg, sd, batch, lr=1e-3, clip=2.0)
        errs, ratios = [], []
        for n, gref in ref["grads"].items():
            if gref.abs().max() == 0: continue
            e = rel_err(nat["grads"][n], gref); e_emu = rel_err(emu["grads"][n], gref)
            errs.append(e); ratios.append(e / (4 * e_emu + 2e-2))
        errs.sort()
        meds.append(errs[len(errs) // 2]); maxs.append(errs[-1]); worst_ratio.append(max(ratios))
        pooled.append(rel_err(nat["pooled"], ref["pooled"]))
    emu_errs = sorted(rel_err(emu["grads"][n], gref) for n, gref in ref["grads"].items() if gref.abs().max() > 0)
    print(f"{variant:8s} p_drop {p_drop}: emu median {emu_errs[len(emu_errs) // 2]:.4f} max {emu_errs[-1]:.3f} | median grad err {min(meds):.4f}..{max(meds):.4f} (limit 0.03)  max {max(maxs):.3f} (limit 0.20)  "
          f"worst e/(4 e_emu + 2e-2) {max(worst_ratio):.2f} (limit 1)  pooled {max(pooled):.2e} (limit 1e-3)", flush=True)

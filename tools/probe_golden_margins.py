"""Margins of the reference-golden tests (CMU MCA / MMA, TCGA at b = 2): per-tensor gradient-norm error against the
reference's own numbers, so that the test thresholds can sit a fixed factor above what bf16 arithmetic gives."""
import importlib, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from util_small import run_native_step, rel_err
P = importlib.import_module("mca-paper_amd")
# (forward attention form: MCA_DEBUG=lazy_softmax=0 for the textbook recurrence; the lazy reference is the default)
G = os.path.join(root, "tests", "golden")
for case in ("cmu_mca_b2", "cmu_mma_d40_b2", "tcga_b2"):
    rec = torch.load(os.path.join(G, case + ".pt"), weights_only=False)
    if case.startswith("cmu"):
        cfg = P.config.cmu_model_config(batch_size=2, zorro="mma" in case)
        batch = P.data.synthetic_batch(cfg, 2, seed=rec["data_seed"], p_drop=rec["p_drop"], lengths="uniform")
    else:
        cfg = P.config.tcga_model_config(batch_size=2)
        batch = P.data.synthetic_batch(cfg, 2, seed=rec["data_seed"], p_drop=rec["p_drop"])
    sd = P.params.init_state_dict(cfg, seed=rec["seed"])
    for rep in range(3):
        nat = run_native_step(P, cfg, sd, batch, lr=1e-4, clip=2.0)
        rels, sl = [], []
        for n, gn_ref in rec["grad_norms"].items():
            if n.endswith("logit_scale") or gn_ref < 1e-12: continue
            rels.append((abs(float(nat["grads"][n].norm()) - gn_ref) / gn_ref, n))
            ref_sl = rec["grad_slices"][n]
            if ref_sl.abs().max() > 0: sl.append((rel_err(nat["grads"][n].flatten()[:64], ref_sl), n))
        rels.sort(); sl.sort()
        print(f"{case} rep {rep}: pooled {rel_err(nat['pooled'], rec['pooled']):.2e}  grad-norm err median {rels[len(rels)//2][0]:.4f} max {rels[-1][0]:.4f} ({rels[-1][1]})  "
              f"slice err median {sl[len(sl)//2][0]:.4f} max {sl[-1][0]:.4f} ({sl[-1][1]})", flush=True)

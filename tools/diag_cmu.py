"""Diagnostic: native CMU b=2 step vs the reference-generated golden."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_small import run_native_step, rel_err
P = importlib.import_module("mca-paper_amd")
case = sys.argv[1] if len(sys.argv) > 1 else "mca"
rec = torch.load(os.path.join(ROOT, "tests", "golden", f"cmu_{case}_b2.pt"), weights_only=False)
cfg = P.config.cmu_model_config(batch_size=2, zorro=case != "mca")
sd = P.params.init_state_dict(cfg, seed=rec["seed"])
batch = P.data.synthetic_batch(cfg, 2, seed=rec["data_seed"], p_drop=rec["p_drop"], lengths="uniform")
nat = run_native_step(P, cfg, sd, batch, lr=1e-4, clip=2.0)
print("pooled shapes", nat["pooled"].shape, rec["pooled"].shape)
n = min(nat["pooled"].shape[1], rec["pooled"].shape[1])
print("pooled rel err %.3e" % rel_err(nat["pooled"][:, :n], rec["pooled"][:, :n]))
for i in range(n):
    print("  slot %d rel %.3e  |ref| %.3f" % (i, rel_err(nat["pooled"][:, i], rec["pooled"][:, i]), float(rec["pooled"][:, i].norm())))
T = float(torch.exp(torch.tensor(2.6593)))
pm = rec["pooled"]
mx = max(float((pm[:, i] @ pm[:, j].t()).abs().max()) for i in range(pm.shape[1]) for j in range(pm.shape[1]))
print("max |a.b| %.2f -> logit scale %.1f" % (mx, mx * T))
print("loss nat %.5f ref %.5f" % (nat["loss"], float(rec["loss"])))
for k, v in rec["losses"].items():
    print("  %-60s nat %.4f ref %.4f" % (k, nat["losses"][k], float(v)))
worst = []
for nme, gref in rec["grad_norms"].items():
    g = float(nat["grads"][nme].norm())
    sl = rel_err(nat["grads"][nme].flatten()[:64], rec["grad_slices"][nme]) if rec["grad_slices"][nme].abs().max() > 0 else 0.0
    worst.append((abs(g - gref) / (gref + 1e-30), nme, g, gref, sl))
for w in sorted(worst, reverse=True)[:12]:
    print("  gradnorm rel %.3e %-50s nat %.4e ref %.4e slice-rel %.2e" % w)
print("median gradnorm rel %.3e" % sorted(w[0] for w in worst)[len(worst) // 2])

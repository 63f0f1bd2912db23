"""Run-to-run gradient noise of the production step over many repeats (same weights, same batch): only the order of fp32 atomics
may differ, so every tensor must stay within ~1e-3 of the first run; a pipeline race shows as an outlier."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); data = importlib.import_module("mca-paper_amd.data")
b = 32; n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = sys.argv[2] if len(sys.argv) > 2 else "cmu"
cfg = {"cmu": lambda: P.config.cmu_model_config(batch_size=b), "mma": lambda: P.config.cmu_model_config(batch_size=b, zorro=True),
       "tcga": lambda: P.config.tcga_model_config(batch_size=b)}[kind]()
torch.manual_seed(43)
model = P.MCA(**cfg).cuda(); model.engine.check_finite = False
batch = data.synthetic_batch(cfg, b, seed=1234, lengths="uniform", p_drop=0.2, device="cuda")
ref, worst = None, (0.0, "", -1)
for it in range(n_rep):
    for p in model.parameters(): p.grad = None
    out = model(batch); out["loss"].backward(); torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in model.named_parameters()}
    if ref is None: ref = g; continue
    for n in g:
        d = float((g[n] - ref[n]).norm() / (ref[n].norm() + 1e-30))
        if d > worst[0]: worst = (d, n, it)
        if d > 5e-3: print(f"OUTLIER run {it}: {n} differs from run 0 by {d:.3e}")
print(f"{kind}: {n_rep - 1} repeats, worst relative difference {worst[0]:.3e} ({worst[1]}, run {worst[2]})")

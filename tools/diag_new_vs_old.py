"""Per-parameter gradient difference: production kernels vs conservative kernels, with the run-to-run noise of each."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); hipm = importlib.import_module("mca-paper_amd.hip"); data = importlib.import_module("mca-paper_amd.data")
b = 32
cfg = P.config.cmu_model_config(batch_size=b)
batch = data.synthetic_batch(cfg, b, seed=1234, lengths="uniform", p_drop=0.2, device="cuda")
def run(conservative):
    for key, val in ((7, 1), (5, 2), (9, 16)): hipm.lib().mca_debug_set(key, val if conservative else 0)
    torch.manual_seed(43)
    model = P.MCA(**cfg).cuda(); eng = model.engine; eng.check_finite = False
    eng.fuse_ln_residual = eng.fuse_geglu_bwd = not conservative
    out = model(batch); out["loss"].backward(); torch.cuda.synchronize()
    return float(out["loss"]), {n: p.grad.clone() for n, p in model.named_parameters()}
rel = lambda a, c: float((a - c).norm() / (c.norm() + 1e-30))
ln, gn = run(False); ln2, gn2 = run(False); lo, go = run(True); lo2, go2 = run(True)
print("loss new", ln, ln2, "old", lo, lo2)
for n in gn:
    print(f"{n:48s} new-vs-old {rel(gn[n], go[n]):.2e}  noise new {rel(gn2[n], gn[n]):.2e}  noise old {rel(go2[n], go[n]):.2e}  |g| {float(gn[n].norm()):.3e}")

"""Run-to-run noise of the production step's gradients (same weights, same batch): per-tensor rel-L2 between repeats, under
the engine's A/B switches.  usage: diag_step_noise.py [kind] [b] [repeats]   env: MCA_DEBUG=overlap_wgrad=1,... (engine.debug_options)"""
import sys, os, importlib, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("mca-paper_amd"); data = importlib.import_module("mca-paper_amd.data")
kind = sys.argv[1] if len(sys.argv) > 1 else "mma"; b = int(sys.argv[2]) if len(sys.argv) > 2 else 32; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cfg = P.config.cmu_model_config(batch_size=b, zorro=(kind == "mma"))
batch = data.synthetic_batch(cfg, b, seed=1234, lengths="uniform", p_drop=0.2, device="cuda")
def rel(a, b_): return float((a.double() - b_.double()).norm() / (b_.double().norm() + 1e-30))
runs = []
for r in range(reps):
    torch.manual_seed(43)
    model = P.MCA(**cfg).cuda(); eng = model.engine; eng.check_finite = False
    out = model(batch); out["loss"].backward(); torch.cuda.synchronize()
    runs.append((float(out["loss"]), {n: p.grad.clone() for n, p in model.named_parameters()}, eng._ws_last["pooled"].clone() if hasattr(eng, "_ws_last") else None))
l0, g0, _ = runs[0]
print("losses", [f"{r[0]:.6f}" for r in runs])
worst = {}
for l, g, _ in runs[1:]:
    for n in g0:
        worst[n] = max(worst.get(n, 0.0), rel(g[n], g0[n]))
top = sorted(worst.items(), key=lambda kv: -kv[1])[:8]
print("switches:", {k: v for k, v in os.environ.items() if k.startswith("MCA_")})
for n, e in top: print(f"  {e:.2e}  {n}  |g| {float(g0[n].norm()):.3e}")
import statistics
print("median over tensors", statistics.median(worst.values()))

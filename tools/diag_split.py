"""Per-parameter gradient difference between the split (two half batches, two streams) and unsplit schedules."""
import copy, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util_small import small_config
P = importlib.import_module("mca-paper_amd"); data = importlib.import_module("mca-paper_amd.data")
variant = sys.argv[1] if len(sys.argv) > 1 else "mca"
cfg = small_config(variant); b = 4
res = []
for split in (False, True, "serial"):
    torch.manual_seed(5)
    model = P.MCA(**copy.deepcopy(cfg)).cuda(); eng = model.engine
    eng.micro_batches, eng.micro_batch_min = (2, 2) if split else (1, 32)
    if split == "serial":                       # same split arithmetic, but everything on ONE stream: no concurrency at all
        eng.overlap_wgrad = False
        eng.split_workspace(b)["stream"] = torch.cuda.current_stream()
    batch = data.synthetic_batch(cfg, b, seed=3, lengths="uniform", p_drop=0.3, device="cuda")
    out = model(batch); out["loss"].backward(); torch.cuda.synchronize()
    res.append({n: p.grad.clone() for n, p in model.named_parameters()})
for n in res[0]:
    a, c, d = res[0][n], res[1][n], res[2][n]
    r = float((a - c).norm() / (a.norm() + 1e-30)); r2 = float((c - d).norm() / (c.norm() + 1e-30))
    if r > 1e-6 or r2 > 1e-6: print(f"{n:50s} unsplit-vs-split {r:.2e}   split-vs-serial-split {r2:.2e}  |g| {float(a.norm()):.3e}")

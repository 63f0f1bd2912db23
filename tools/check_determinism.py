"""Forward determinism at CMU size: the forward has no order-dependent accumulation except mca_attn_vmean (uniform rows), so
repeated forwards of the same weights and batch must agree bit for bit; reports the first tensor that differs."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
model = P.MCA(**cfg).cuda(); eng = model.engine; eng.check_finite = False
batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="full", device="cuda")
snaps = []
for it in range(4):
    with torch.no_grad():
        out = model(batch)
    torch.cuda.synchronize()
    ws = eng.workspace(b)
    snap = {"x0": ws["x"][0].clone()}
    for i, a in enumerate(ws["layers"]):
        for k in ("xn_b", "qkv", "o", "x1", "x1n_b", "h", "g"):
            snap[f"L{i}.{k}"] = a[k].clone()
        snap[f"L{i}.xout"] = ws["x"][i + 1].clone()
    snap["pooled"] = ws["pooled"].clone(); snap["loss"] = out["loss"].clone()
    snaps.append(snap)
ok = True
for it in range(1, 4):
    for k in snaps[0]:
        a, c = snaps[0][k], snaps[it][k]
        if not torch.equal(a, c):
            ne = (a != c)
            print(f"run {it}: {k} differs: {int(ne.sum())} of {ne.numel()} elements, max |d| {float((a.float() - c.float()).abs().max()):.3e}; first at {ne.nonzero()[0].tolist()}")
            ok = False
            break
print("forward deterministic" if ok else "FORWARD NOT DETERMINISTIC", "loss", [float(s["loss"]) for s in snaps])

"""Forward determinism at CMU size: the forward has no order-dependent accumulation except mca_attn_vmean (uniform rows), so
repeated forwards of the same weights and batch must agree bit for bit; reports the first tensor that differs.
usage: check_determinism.py [batch] [iterations]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NIT = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind = sys.argv[3] if len(sys.argv) > 3 else "cmu"
cfg = {"cmu": lambda: P.config.cmu_model_config(batch_size=b), "mma": lambda: P.config.cmu_model_config(batch_size=b, zorro=True),
       "long": lambda: P.config.cmu_model_config(batch_size=b, long_seq=True), "tcga": lambda: P.config.tcga_model_config(batch_size=b)}[kind]()
torch.manual_seed(43)
model = P.MCA(**cfg).cuda(); eng = model.engine; eng.check_finite = False
batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="full", device="cuda")          # no dropped modality: no atomically summed uniform rows
def snapshot(clone):
    ws = eng.workspace(b)
    snap = {"x0": ws["x"][0]}
    for i, a in enumerate(ws["layers"]):
        for k in ("xn_b", "qkv", "o", "x1", "x1n_b", "h", "g"):
            snap[f"L{i}.{k}"] = a[k]
        snap[f"L{i}.xout"] = ws["x"][i + 1]
    snap["pooled"] = ws["pooled"]
    return {k: v.clone() for k, v in snap.items()} if clone else snap
ref, losses, bad = None, set(), 0
for it in range(NIT):
    with torch.no_grad():
        out = model(batch)
    torch.cuda.synchronize()
    losses.add(float(out["loss"]))
    if ref is None:
        ref = snapshot(True); continue
    cur = snapshot(False)
    for k in ref:
        a, c = ref[k], cur[k]
        if not torch.equal(a, c):
            ne = (a != c); rows = ne.nonzero()[:, 0]
            print(f"run {it}: {k} differs: {int(ne.sum())} of {ne.numel()} elements, max |d| {float((a.float() - c.float()).abs().max()):.3e}; "
                  f"rows {torch.unique(rows)[:10].tolist()} cols of first row {ne[rows[0]].nonzero().flatten()[:10].tolist()}")
            bad += 1
            break
print("forward deterministic" if not bad else f"FORWARD NOT DETERMINISTIC in {bad} of {NIT - 1} repeats", "distinct losses", sorted(losses))

"""Buffer checksums after each step: eager (no side stream) vs graph replay (diagnostic)"""
import importlib, torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["MCA_OVERLAP_WGRAD"] = "0"
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim"); graph = importlib.import_module("mca-paper_amd.graph")
b = 2
cfg = P.config.cmu_model_config(batch_size=b)
def sums(eng):
    ws = eng.workspace(b); out = {}
    for k in ("dpool_b", "dop", "dqp32", "dkvp", "delta_p", "lse_p", "op", "kvp", "qp", "t_b", "mf", "rf"):
        out[k] = float(ws[k].float().abs().sum())
    for k in ("dxa", "dxb", "dx_b", "do", "xn", "x1n"):
        out[k] = float(ws[k].float().abs().sum())
    out["x0"] = float(ws["x"][0].abs().sum()); out["x5"] = float(ws["x"][5].abs().sum())
    for li in (4, 0):
        a = ws["layers"][li]
        for k in ("dqkv", "dxo_b", "dx1_b", "dh", "lse", "o", "qkv"):
            out[f"l{li}." + k] = float(a[k].float().abs().sum())
    e = ws["enc"]["COVAREP"]
    for k in ("dy", "y", "m2", "r2", "dxin"):
        out["enc." + k] = float(e[k].float().abs().sum())
    out["gflat"] = float(eng.gflat.abs().sum())
    return out
res = {}
for mode in ("eager", "graph"):
    torch.manual_seed(43)
    m = P.MCA(**cfg).cuda(); m.engine.check_finite = "deferred"
    opt = optim.FusedAdamW(m, lr=1e-6)
    batch = P.data.synthetic_batch(cfg, b, seed=1234, device="cuda")
    g = graph.GraphedStep(m, opt, batch, clip=2.0) if mode == "graph" else None
    rows = []
    for i in range(3):
        if g is None:
            out = m(batch); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(m, 2.0); opt.step()
        else:
            g.step(batch)
        torch.cuda.synchronize(); rows.append(sums(m.engine))
    res[mode] = rows
for i in range(3):
    print("step", i)
    for k in res["eager"][i]:
        a, c = res["eager"][i][k], res["graph"][i][k]
        flag = "" if abs(a - c) <= 1e-3 * abs(a) + 1e-6 else "   <<<<"
        print(f"   {k:10s} eager {a:14.6g} graph {c:14.6g}{flag}")

"""Replay vs eager over several optimizer steps on every model family / data shape: loss and global gradient norm per step
(a replayed step that goes wrong silently shows here: DESIGN.md section 5, "A memset node that was not ordered").
usage: soak_graph.py [steps]"""
import sys, importlib, torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim"); graph = importlib.import_module("mca-paper_amd.graph")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
CASES = {
    "cmu mca b=8 ragged+drop": (lambda: P.config.cmu_model_config(batch_size=8), dict(lengths="uniform", p_drop=0.3), {}),
    "cmu mma b=8 drop 0.4": (lambda: P.config.cmu_model_config(batch_size=8, zorro=True), dict(lengths="uniform", p_drop=0.4), {}),
    "cmu mca b=8 fp8": (lambda: P.config.cmu_model_config(batch_size=8), dict(lengths="uniform", p_drop=0.2), {"fp8": True}),
    "tcga b=4 drop": (lambda: P.config.tcga_model_config(batch_size=4), dict(p_drop=0.25), {}),
    "cmu eao b=2 drop": (lambda: P.config.cmu_eao_model_config(batch_size=2), dict(lengths="uniform", p_drop=0.3), {}),
    "long mca b=4": (lambda: P.config.cmu_model_config(batch_size=4, long_seq=True), dict(lengths="uniform", p_drop=0.2), {}),
}
bad = 0
for name, (mk, dkw, opts) in CASES.items():
    cfg = mk(); b = cfg["batch_size"]
    hist = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(43)
        m = P.build_model(cfg).cuda(); m.engine.check_finite = "deferred"
        if opts.get("fp8"): m.engine.set_attention_dtype("fp8")
        opt = optim.FusedAdamW(m, lr=1e-5)
        batches = [P.data.synthetic_batch(cfg, b, seed=100 + i, device="cuda", **dkw) for i in range(steps)]
        g = graph.GraphedStep(m, opt, batches[0], clip=2.0) if mode == "graph" else None
        rows = []
        for bt in batches:
            if g is None:
                out = m(bt); opt.zero_grad(); out["loss"].backward(); gn = optim.clip_grad_norm_(m, 2.0); opt.step()
                rows.append((float(out["loss"].detach()), float(gn)))
            else:
                rows.append((float(g.step(bt)), float(g.gnorm)))
        torch.cuda.synchronize(); m.engine.assert_finite()
        hist[mode] = rows
        del m, opt, g
        torch.cuda.empty_cache()
    worst = max(max(abs(a[0] - c[0]) / max(abs(a[0]), 1e-12), abs(a[1] - c[1]) / max(a[1], 1e-12)) for a, c in zip(hist["eager"], hist["graph"]))
    ok = worst < 3e-2 and all(x == x and abs(x) < 1e6 for r in hist["graph"] for x in r)
    bad += not ok
    print(f"{name:28s} {'OK ' if ok else 'BAD'} worst rel dev {worst:.2e}  eager last {hist['eager'][-1]}  graph last {hist['graph'][-1]}", flush=True)
print("soak:", "all ok" if not bad else f"{bad} BAD")
sys.exit(1 if bad else 0)

"""Which gradient tensors go wrong when the CMU-size step is replayed from a graph (diagnostic)."""
import importlib, torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim"); graph = importlib.import_module("mca-paper_amd.graph")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
m = P.MCA(**cfg).cuda(); m.engine.check_finite = "deferred"
if os.environ.get("NO_LNRES"): m.engine.fuse_ln_residual = False
if os.environ.get("NO_GEGLU"): m.engine.fuse_geglu_bwd = False
opt = optim.FusedAdamW(m, lr=1e-6)
batch = P.data.synthetic_batch(cfg, b, seed=1234, device="cuda")
g = graph.GraphedStep(m, opt, batch, clip=2.0)
for i in range(3):
    loss = g.step(batch); torch.cuda.synchronize()
    bad = [(n, float(m.engine.grad_of(p).abs().max())) for n, p in m.named_parameters() if not float(m.engine.grad_of(p).abs().max()) < 1e6]
    print("graph", i, float(loss), float(g.gnorm), "bad tensors:", bad[:6], len(bad), flush=True)

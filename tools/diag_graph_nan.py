"""Which parameters move when a non-finite batch goes through the graphed step (diagnostic)."""
import sys, os, copy, importlib, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from util_small import small_config, to_device
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim"); graph = importlib.import_module("mca-paper_amd.graph")
cfg = small_config("tab"); sd = P.params.init_state_dict(cfg, seed=3)
batches = [to_device(P.data.synthetic_batch(cfg, 4, seed=40 + i, p_drop=0.2), "cuda") for i in range(2)]
m = P.MCA(**copy.deepcopy(cfg)); m.load_state_dict(sd, strict=False); m = m.cuda(); m.engine.check_finite = "deferred"
opt = optim.FusedAdamW(m, lr=1e-3, weight_decay=0.0)
g = graph.GraphedStep(m, opt, batches[0], clip=2.0, warmup=2)
g.step(batches[1]); torch.cuda.synchronize(); m.engine.assert_finite()
before = m.engine.flat.clone(); mb, vb = opt.exp_avg.clone(), opt.exp_avg_sq.clone()
bad = copy.deepcopy(batches[0]); bad["audio"]["tokens"][0, 0, 0] = float("nan")
g.step(bad); torch.cuda.synchronize()
print("flag", m.engine.finite_flag.item(), "host", m.engine._flag_host.item())
d = (m.engine.flat != before)
print("changed elements", int(d.sum()), "of", d.numel(), "nan in flat", int(torch.isnan(m.engine.flat).sum()), "moments changed", int((opt.exp_avg != mb).sum()), int((opt.exp_avg_sq != vb).sum()))
for n, p in m.named_parameters():
    off = p.data_ptr() - m.engine.flat.data_ptr()
    if 0 <= off < m.engine.flat.numel() * 4:
        i0 = off // 4
        c = int(d[i0:i0 + p.numel()].sum())
        if c: print("  ", n, c, "/", p.numel(), (m.engine.flat[i0:i0 + p.numel()] - before[i0:i0 + p.numel()]).abs().max().item())

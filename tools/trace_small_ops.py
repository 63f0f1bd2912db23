"""Lists every GPU operation of one eager training step that is NOT a libmca_hip.so kernel (torch fills, copies, element-wise
ops): each is a launch / graph node of its own, ~3-5 us plus a boundary.  usage: trace_small_ops.py [batch]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
model = P.build_model(cfg).cuda(); model.engine.check_finite = "deferred"
opt = optim.FusedAdamW(model, lr=1e-4)
batch = P.data.synthetic_batch(cfg, b, seed=1234, device="cuda")
def step():
    out = model(batch); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step(); model.engine.poll_finite()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print(len(evs), "GPU operations in one step")
from collections import Counter
c = Counter(); t = Counter()
for e in evs:
    nm = e.name[:100]
    c[nm] += 1; t[nm] += e.device_time if hasattr(e, "device_time") else e.cuda_time
for nm, n in c.most_common():
    print(f"{n:4d} x {t[nm]/n:8.1f} us  {nm}")
# where do the torch-side ones come from?
cpu = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is None or (e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and not (e.cpu_parent.name.startswith("aten::") if e.cpu_parent else False))]
cc = Counter()
for e in cpu:
    st = [f for f in (e.stack or []) if "mca-paper_amd" in f or "bench" in f or "trace_small" in f]
    cc[(e.name, st[0] if st else "?")] += 1
for (nm, where), n in cc.most_common(60):
    print(f"{n:3d} {nm:28s} {where}")

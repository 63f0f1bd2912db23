"""Per-step wall time while alternating the split / exclusive schedules (what bench.py's sampled steps do)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim")
b = 32
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
model = P.MCA(**cfg).cuda(); eng = model.engine; eng.check_finite = False
opt = optim.FusedAdamW(model, lr=1e-4)
batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="full", device="cuda")
def step():
    out = model(batch); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
pattern = [1, 2, 2, 2, 1, 2, 2, 2, 2, 1, 1, 2, 2, 2, 2, 2]
for mb in pattern:
    eng.micro_batches = mb
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"micro_batches={mb}: enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms   mem {torch.cuda.memory_reserved()/2**30:.1f} GiB")

"""How long does the host take to ENQUEUE one step (no sync), vs. the GPU time of the step?"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
model = P.MCA(**cfg).cuda(); model.engine.check_finite = False
opt = optim.FusedAdamW(model, lr=1e-4)
batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="full", device="cuda")
def step():
    out = model(batch); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"b={b} enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

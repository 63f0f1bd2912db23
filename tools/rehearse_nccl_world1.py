"""Rehearsal of the N > 1 code path on ONE GPU with the real RCCL backend (world size 1): process-group init, the packed
all-gather of pooled embeddings, asynchronous bucketed all-reduce issued from the side stream, finish_backward.  Checks that the
step matches the plain single-GPU step and reports both step times."""
import importlib, os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim"); dpmod = importlib.import_module("mca-paper_amd.dp")
b = 32
cfg = P.config.cmu_model_config(batch_size=b)
res = {}
for mode in ("plain", "plain2", "dp"):
    torch.manual_seed(43)
    model = P.MCA(**cfg).cuda(); model.engine.check_finite = False
    opt = optim.FusedAdamW(model, lr=1e-4)
    dp = None
    if mode == "dp":
        dp = dpmod.DataParallelMCA(model)
        red = dp.reducer
        def bucket_ready(lo, hi, red=red):          # the world == 1 shortcut removed: every bucket goes through RCCL
            if hi > lo:
                red.pending.append(dist.all_reduce(red.flat[lo:hi], op=dist.ReduceOp.SUM, group=red.group, async_op=True))
        model.engine.grad_bucket_hook = bucket_ready
    batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="full", device="cuda")
    def step():
        out = model(batch); opt.zero_grad(); out["loss"].backward()
        if dp is not None: dp.finish_backward()
        optim.clip_grad_norm_(model, 2.0); opt.step()
        return out["loss"]
    first = float(step())
    for _ in range(2): loss = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): loss = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    res[mode] = (float(loss), ms, model.engine.flat.clone(), first)
    print(f"{mode}: first loss {first:.6f}, loss after 13 steps {float(loss):.6f}  {ms:.2f} ms/step", flush=True)
rel = lambda a, b: float((res[a][2] - res[b][2]).norm() / res[a][2].norm())
noise, d = rel("plain", "plain2"), rel("plain", "dp")
print(f"parameter difference after 13 steps (relative L2): plain vs plain (run-to-run noise) {noise:.2e}, plain vs dp {d:.2e}")
assert abs(res["plain"][3] - res["dp"][3]) <= 1e-5 * abs(res["plain"][3]), "first-step loss must agree"
assert d < 3 * noise + 1e-4
dist.destroy_process_group()
print("rccl world-size-1 rehearsal ok")

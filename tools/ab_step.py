"""A/B of a debug knob on the full step, interleaved rounds in ONE process (cdna guide rule 24).
usage: ab_step.py <knob> <valueA> <valueB> [batch]"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim"); H = importlib.import_module("mca-paper_amd.hip")
knob, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b = int(sys.argv[4]) if len(sys.argv) > 4 else 32
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
model = P.MCA(**cfg).cuda(); model.engine.check_finite = False
opt = optim.FusedAdamW(model, lr=1e-4)
batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="full", device="cuda")
def step():
    out = model(batch); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
for _ in range(3): step()
res = {va: [], vb: []}
for rnd in range(4):
    for v in (va, vb):
        if knob == 102: model.engine.micro_batches = v
        elif knob >= 100: setattr(model.engine, {100: "fuse_geglu_bwd", 101: "overlap_wgrad", 103: "fuse_ln_residual", 104: "group_wgrad", 105: "zero_dq_once"}[knob], bool(v))
        else: H.lib().mca_debug_set(knob, v)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 5 * 1e3)
for v in (va, vb):
    r = sorted(res[v]); print(f"knob {knob}={v}: median {r[len(r)//2]:.2f} ms  min {r[0]:.2f}  all {[round(x,2) for x in res[v]]}")

"""Step time of two builds of the library, alternating processes (each process: warm-up + 3 x 5 timed steps).
usage: ab_lib_step.py <libA.so> <libB.so> [more .so ...] [rounds]"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(%r))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim")
b = 32
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
model = P.MCA(**cfg).cuda(); model.engine.check_finite = False
opt = optim.FusedAdamW(model, lr=1e-4)
batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="full", device="cuda")
def step():
    out = model(batch); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
for _ in range(4): step()
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
print("%%.2f" %% sorted(ts)[1])
''' % here
libs = [a for a in sys.argv[1:] if a.endswith(".so")]; rounds = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [3])[0]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, MCA_HIP_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        res[l].append(float(out.stdout.strip().splitlines()[-1]))
for l in libs: print(l, "median", sorted(res[l])[len(res[l]) // 2], res[l])

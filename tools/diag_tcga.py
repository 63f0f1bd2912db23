import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_small import run_native_step, run_oracle_step, rel_err
from oracle import mca_oracle as O
P = importlib.import_module("mca-paper_amd")
torch.set_num_threads(16)
cfg = P.config.tcga_model_config(batch_size=2)
sd = P.params.init_state_dict(cfg, seed=43)
batch = P.data.synthetic_batch(cfg, 2, seed=77, p_drop=0.25)
nat = run_native_step(P, cfg, sd, batch, lr=1e-4)
ref = run_oracle_step(O, cfg, sd, batch, "fp32", lr=1e-4)
emu = run_oracle_step(O, cfg, sd, batch, "bf16emu", lr=1e-4)
print("pooled nat-fp32 %.2e emu-fp32 %.2e nat-emu %.2e" % (rel_err(nat["pooled"], ref["pooled"]), rel_err(emu["pooled"], ref["pooled"]), rel_err(nat["pooled"], emu["pooled"])))
print("loss", nat["loss"], ref["loss"], emu["loss"], "gn", nat["grad_norm"], ref["grad_norm"], emu["grad_norm"])
rows = []
for n, g in ref["grads"].items():
    if g.abs().max() == 0: continue
    rows.append((rel_err(nat["grads"][n], g), rel_err(emu["grads"][n], g), rel_err(nat["grads"][n], emu["grads"][n]), n, float(g.norm())))
for r in sorted(rows, reverse=True)[:14]: print("nat-fp32 %.3f emu-fp32 %.3f nat-emu %.3f %s |g|=%.3e" % r)
import statistics
print("median nat-fp32 %.3f emu-fp32 %.3f" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)))

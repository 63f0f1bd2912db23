"""Diagnostic: native step vs oracle (fp32 and bf16emu) on the small config; prints every error."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util_small import small_config, run_native_step, run_oracle_step, rel_err
from oracle import mca_oracle as O
P = importlib.import_module("mca-paper_amd")
variant = sys.argv[1] if len(sys.argv) > 1 else "mca"
p_drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
cfg = small_config(variant)
batch = P.data.synthetic_batch(cfg, 6, seed=5, p_drop=p_drop)
sd = P.params.init_state_dict(cfg, seed=3)
g = torch.Generator().manual_seed(9)
for k in sd:
    if k.endswith("gamma") or k.endswith("bias") or ("token_encoder" in k and sd[k].dim() == 1):
        sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
nat = run_native_step(P, cfg, sd, batch)
ref = run_oracle_step(O, cfg, sd, batch, "fp32")
emu = run_oracle_step(O, cfg, sd, batch, "bf16emu")
r64 = run_oracle_step(O, cfg, sd, batch, "fp64")
print("pooled: nat-vs-fp32 %.2e  nat-vs-emu %.2e  emu-vs-fp32 %.2e  fp32-vs-fp64 %.2e" % (
    rel_err(nat["pooled"], ref["pooled"]), rel_err(nat["pooled"], emu["pooled"]), rel_err(emu["pooled"], ref["pooled"]), rel_err(ref["pooled"], r64["pooled"])))
print("loss: nat %.5f fp32 %.5f emu %.5f fp64 %.5f" % (nat["loss"], ref["loss"], emu["loss"], r64["loss"]))
for k in ref["losses"]:
    print("  %-40s nat %.4f fp32 %.4f emu %.4f" % (k, nat["losses"][k], ref["losses"][k], emu["losses"][k]))
print("grad_norm nat %.4f fp32 %.4f emu %.4f" % (nat["grad_norm"], ref["grad_norm"], emu["grad_norm"]))
for n in ref["grads"]:
    gr = ref["grads"][n]
    print("  %-50s |g| %.3e nat-vs-fp32 %.2e emu-vs-fp32 %.2e nat-vs-emu %.2e" % (n, float(gr.norm()), rel_err(nat["grads"][n], gr),
          rel_err(emu["grads"][n], gr), rel_err(nat["grads"][n], emu["grads"][n])))

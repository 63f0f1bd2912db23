"""Which part of the round-2 attention changes moves the CMU golden gradients: q pre-scaling, the lazy-reference forward, the
two-pass backward?  Runs the b = 2 golden step under each combination (one process per combination: the toggles are read at
engine construction) and prints the pooled / gradient-norm errors against the reference's numbers."""
import os, subprocess, sys
code = r'''
import importlib, os, sys, torch
root = %r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from util_small import run_native_step, rel_err
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
rec = torch.load(os.path.join(root, "tests", "golden", "cmu_mca_b2.pt"), weights_only=False)
cfg = P.config.cmu_model_config(batch_size=2)
batch = P.data.synthetic_batch(cfg, 2, seed=rec["data_seed"], p_drop=rec["p_drop"], lengths="uniform")
sd = P.params.init_state_dict(cfg, seed=rec["seed"])
H.lib().mca_debug_set(13, int(os.environ.get("K13", "0")))
nat = run_native_step(P, cfg, sd, batch, lr=1e-4, clip=2.0)
rels = sorted((abs(float(nat["grads"][n].norm()) - g) / g, n) for n, g in rec["grad_norms"].items() if g > 1e-12 and not n.endswith("logit_scale"))
print(f"pooled {rel_err(nat['pooled'], rec['pooled']):.2e} loss {nat['loss']:.5f} (ref {float(rec['loss']):.5f}) grad-norm err median {rels[len(rels)//2][0]:.4f} max {rels[-1][0]:.4f} ({rels[-1][1]}) 2nd {rels[-2][0]:.4f} ({rels[-2][1]})")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, env in (("prescale + first-form fwd + two-pass bwd (default)", {}),
                  ("prescale + first-form fwd + one-pass bwd", {"MCA_ATTN_BWD_ONE_PASS": "1"}),
                  ("prescale + lazy second-form fwd + two-pass bwd", {"K13": "2"}),
                  ("no prescale + first-form fwd + one-pass bwd (round 1)", {"MCA_Q_PRESCALE": "0", "MCA_ATTN_BWD_ONE_PASS": "1"}),
                  ("no prescale + lazy second-form fwd + one-pass bwd", {"MCA_Q_PRESCALE": "0", "MCA_ATTN_BWD_ONE_PASS": "1", "K13": "2"})):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    print(f"{name:58s}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-800:]}", flush=True)

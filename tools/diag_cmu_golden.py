"""The CMU b = 2 golden step (tests/golden/cmu_mca_b2.pt: numbers of the REFERENCE) under kernel-selection knobs: relative error
of every gradient norm, worst first.  usage: diag_cmu_golden.py [k13=1] [k1=1] ...   (several runs: repeats)"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_small import run_native_step, rel_err
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
rec = torch.load(os.path.join(ROOT, "tests", "golden", "cmu_mca_b2.pt"), weights_only=False)
cfg = P.config.cmu_model_config(batch_size=2)
sd = P.params.init_state_dict(cfg, seed=rec["seed"])
batch = P.data.synthetic_batch(cfg, 2, seed=rec["data_seed"], p_drop=rec["p_drop"], lengths="uniform")
variants = [{}] + [dict([kv.split("=")]) for kv in sys.argv[1:]]
for rep in range(2):
    for kn in variants:
        kn = {k: int(v) for k, v in kn.items()}
        with H.knobs(**kn):
            nat = run_native_step(P, cfg, sd, batch, lr=1e-4, clip=2.0)
        errs = []
        for n, gn_ref in rec["grad_norms"].items():
            if n.endswith("logit_scale") or gn_ref < 1e-12: continue
            errs.append((abs(float(nat["grads"][n].norm()) - gn_ref) / gn_ref, n))
        errs.sort(reverse=True)
        print(kn or "default", "pooled", f"{rel_err(nat['pooled'], rec['pooled']):.2e}", "worst grad-norm errs:", [(f"{e:.3f}", n.split('.')[1] + '.' + n.split('.')[-2] + '.' + n.split('.')[-1]) for e, n in errs[:4]], "median", f"{errs[len(errs)//2][0]:.4f}", flush=True)

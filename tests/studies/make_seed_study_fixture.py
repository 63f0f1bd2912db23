"""Fixture of the seed study (tests/studies/lazy_softmax_seed_study.py): the fp64 oracle's pooled output, loss terms and per-tensor
gradient norms of ONE CMU-shaped step (N = 2538, D = 512, L = 5, b = 2, uniform lengths) for 8 data seeds x {MCA, MMA with 40 %
of the modalities dropped}.  CPU only (about 30 s per case on 8 cores): python tests/studies/make_seed_study_fixture.py
-> tests/golden/cmu_b2_seed_study.pt.  Test infrastructure: uses oracle/ (never imported by the product)."""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_small import run_oracle_step
from oracle import mca_oracle as O
P = importlib.import_module("mca-paper_amd")
SEEDS = [101, 102, 103, 104, 105, 106, 107, 108]
out = {"seeds": SEEDS, "init_seed": 0, "cases": {}}
for case, zorro, p_drop in (("mca", False, 0.0), ("mma_d40", True, 0.4)):
    cfg = P.config.cmu_model_config(batch_size=2, zorro=zorro)
    sd = P.params.init_state_dict(cfg, seed=0)
    for s in SEEDS:
        t = time.time()
        batch = P.data.synthetic_batch(cfg, 2, seed=s, p_drop=p_drop, lengths="uniform")
        r = run_oracle_step(O, cfg, sd, batch, "fp64", lr=1e-4, clip=2.0)
        out["cases"][(case, s)] = dict(pooled=r["pooled"].float(), loss=r["loss"], losses={k: float(v) for k, v in r["losses"].items()},
                                       grad_norm=r["grad_norm"], grad_norms={n: float(g.double().norm()) for n, g in r["grads"].items()},
                                       p_drop=p_drop)
        print(case, s, f"{time.time() - t:.0f} s  loss {r['loss']:.5f}  |g| {r['grad_norm']:.3f}", flush=True)
torch.save(out, os.path.join(ROOT, "tests", "golden", "cmu_b2_seed_study.pt"))

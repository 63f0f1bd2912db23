"""Upper bound of folding the MX-fp8 quantisation passes into GEMM epilogues (BASELINE configs[4], LONG b = 128): the replayed
step with the two quantisation launches (mca_attn_quant_mxfp8, mca_attn_quant_bwd_mxfp8) left out of the captured graph after
the warm-up steps (TIMING ONLY: the fp8 operands are then the warm-up's, the results are stale by construction), against the
normal fp8 step and the bf16 step of the same box.  usage: ablate_fp8_quant.py [batch]"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
E = importlib.import_module("mca-paper_amd.engine"); optim = importlib.import_module("mca-paper_amd.optim"); graph = importlib.import_module("mca-paper_amd.graph")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
real_call, state = H.call, {"skip": False}


def call(name, *a, **k):
    if state["skip"] and name in ("mca_attn_quant_mxfp8", "mca_attn_quant_bwd_mxfp8"):
        return
    return real_call(name, *a, **k)


H.call = call; E.call = call


def run(label, dtype, skip):
    cfg = P.config.cmu_model_config(batch_size=b, long_seq=True)
    torch.manual_seed(43)
    model = P.build_model(cfg).cuda(); model.engine.check_finite = "deferred"; model.engine.set_attention_dtype(dtype)
    opt = optim.FusedAdamW(model, lr=1e-4)
    batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="full", device="cuda")
    for _ in range(2):          # eager steps with the quantisation in place: every fp8 buffer holds real operands
        out = model(batch); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
    torch.cuda.synchronize()
    state["skip"] = skip
    g = graph.GraphedStep(model, opt, batch, clip=2.0, warmup=1)
    for _ in range(2): g.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): g.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    state["skip"] = False
    print(f"{label:46s}: {dt * 1e3:8.2f} ms / step = {b / dt:7.1f} samples/s", flush=True)
    del g, model, opt, batch; torch.cuda.empty_cache()
    return dt


t_bf = run("bf16 attention", "bf16", False)
t_f8 = run("fp8 attention (production)", "fp8", False)
t_nq = run("fp8 attention, quantisation launches left out", "fp8", True)
print(f"fp8 over bf16: {100 * (t_bf / t_f8 - 1):+.1f} %; with free quantisation: {100 * (t_bf / t_nq - 1):+.1f} %", flush=True)

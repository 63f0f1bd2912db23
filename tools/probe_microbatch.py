"""Feasibility probe: forward trunk of two half batches on two streams vs one full batch on one stream."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
b = 32
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
model = P.MCA(**cfg).cuda(); eng = model.engine; eng.check_finite = False
full = P.data.synthetic_batch(cfg, b, seed=1, lengths="full", device="cuda")
h0 = {k: {kk: vv[:b // 2].contiguous() for kk, vv in v.items()} for k, v in full.items()}
h1 = {k: {kk: vv[b // 2:].contiguous() for kk, vv in v.items()} for k, v in full.items()}
eng.refresh_weights()
ws_full = eng.workspace(b)
ws0 = eng.workspace(b // 2)
import copy
eng._ws.pop(b // 2); ws1 = eng.workspace(b // 2)          # a second, distinct half-size workspace
s1 = torch.cuda.Stream()
def fwd_full():
    with H.cached_stream():
        eng._encode(full, ws_full, False); eng.forward_trunk(ws_full)
def fwd_split():
    ev = torch.cuda.Event(); ev.record()
    with H.cached_stream():
        eng._encode(h0, ws0, False); eng.forward_trunk(ws0)
    with torch.cuda.stream(s1), H.use_stream(s1.cuda_stream):
        s1.wait_event(ev)
        eng._encode(h1, ws1, False); eng.forward_trunk(ws1)
    torch.cuda.current_stream().wait_stream(s1)
def fwd_seq_halves():
    with H.cached_stream():
        eng._encode(h0, ws0, False); eng.forward_trunk(ws0)
        eng._encode(h1, ws1, False); eng.forward_trunk(ws1)
for name, fn in (("full b=32", fwd_full), ("two halves, 2 streams", fwd_split), ("two halves, 1 stream", fwd_seq_halves)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
ref = ws_full["pooled"].clone(); fwd_split(); torch.cuda.synchronize()
got = torch.cat([ws0["pooled"], ws1["pooled"]])
print("max abs diff pooled", float((ref - got).abs().max()))

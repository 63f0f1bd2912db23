"""Interleaved A/B (one process, one box) of the forward attention kernels on the CMU structure: the 128-row-tile LDS-DMA kernel
with the lazy softmax reference (MCA_ATTN_LAZY_REFERENCE), the same kernel with the textbook recurrence (production), the
query-block kernel (round 4: 256-row blocks, one wavefront per SIMD, 64 rows each; knob 13 = 3 + the block schedule) and the
register-staged one (knob 13 = 1).  usage: ab_fwd_forms.py"""
import importlib, os, sys
os.environ.setdefault("MCA_DEBUG", "fwd64=1")          # the engine offers the query-block schedule only when asked
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
def setup(b, pad):
    cfg = P.config.cmu_model_config(batch_size=b); cfg["depth"] = 1
    torch.manual_seed(0)
    eng = P.MCA(**cfg).cuda().engine
    ws = eng.workspace(b); N, D = eng.N, eng.D
    ws["padding"].zero_()
    if pad:
        g = torch.Generator(device="cuda").manual_seed(7)
        for mi, n in enumerate(eng.st.token_dims):
            ln = torch.randint(1, n + 1, (b,), device="cuda", generator=g)
            ln[torch.rand(b, device="cuda", generator=g) < 0.2] = 0
            ws["padding"][:, eng.offsets[mi]:eng.offsets[mi] + n] = (torch.arange(n, device="cuda")[None] >= ln[:, None]).to(ws["padding"].dtype)
    H.call("mca_build_keyinfo", ws["padding"].data_ptr(), eng.kgroup.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr(), b, N, eng.nk_pad, H.stream_ptr())
    H.call("mca_build_keyhot", ws["keyinfo"].data_ptr(), ws["khot"].data_ptr(), b, eng.nk_pad, H.stream_ptr())
    a = ws["layers"][0]
    a["qkv"].copy_(torch.randn(b * N, 3 * D, device="cuda").bfloat16()); a["qkv"][:, :D] *= 0.18
    return eng, (lambda: eng._attn_fwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N))
def t(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for b, pad in ((32, 0), (32, 1), (16, 0), (8, 0), (8, 1), (4, 0)):
    eng, fn = setup(b, pad)
    bs = eng.bsched_attn
    r = {0: [], 1: [], 2: [], 3: []}
    for rnd in range(4):
        for k in (0, 2, 3, 1):
            eng.bsched_attn = bs if k == 3 else None          # the block schedule selects the query-block kernel
            eng.attn_flags = H.ATTN_Q_PRESCALED | (H.ATTN_LAZY_REFERENCE if k == 0 else 0)
            with H.knobs(k13=0 if k == 2 else k):
                r[k].append(t(fn))
    eng.bsched_attn = bs
    print(f"b={b} pad={pad}: lazy reference {min(r[0]):.1f} us   textbook recurrence {min(r[2]):.1f} us   query-block {min(r[3]):.1f} us   "
          f"register-staged {min(r[1]):.1f} us   lazy / textbook {min(r[0]) / min(r[2]):.3f}", flush=True)

"""Times the layer attention kernels (fwd, bwd) on the CMU structure at b=32, H=8 (or LONG with argv[2] == long)."""
import importlib, os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip"); E = importlib.import_module("mca-paper_amd.engine")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
long_seq = len(sys.argv) > 2 and sys.argv[2] == "long"
cfg = P.config.cmu_model_config(batch_size=b, long_seq=long_seq)
cfg["depth"] = 1
torch.manual_seed(0)
model = P.MCA(**cfg).cuda(); eng = model.engine
ws = eng.workspace(b)
N, D, Hh = eng.N, eng.D, eng.H
ws["padding"].zero_()
if os.environ.get("MCA_BENCH_ATTN_PAD"):          # the bench's padding (uniform lengths, 20 % of the modalities dropped) instead of none
    g = torch.Generator(device="cuda").manual_seed(7)
    for mi, n in enumerate(eng.st.token_dims):
        ln = torch.randint(1, n + 1, (b,), device="cuda", generator=g)
        ln[torch.rand(b, device="cuda", generator=g) < 0.2] = 0
        ws["padding"][:, eng.offsets[mi]:eng.offsets[mi] + n] = (torch.arange(n, device="cuda")[None] >= ln[:, None]).to(ws["padding"].dtype)
H.call("mca_build_keyinfo", ws["padding"].data_ptr(), eng.kgroup.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr(), b, N, eng.nk_pad, H.stream_ptr())
if ws.get("khot") is not None:
    H.call("mca_build_keyhot", ws["keyinfo"].data_ptr(), ws["khot"].data_ptr(), b, eng.nk_pad, H.stream_ptr())
a = ws["layers"][0]
a["qkv"].copy_(torch.randn_like(a["qkv"].float()).bfloat16())
a["qkv"][:, :D] *= 0.18          # q as the engine stores it (scale * log2 e folded in)
ws["do"].copy_(torch.randn_like(ws["do"].float()).bfloat16())
def fwd(): eng._attn_fwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N)
def bwd():
    eng._attn_bwd2(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], ws["do"], a["lse"], ws["delta"], a["dqkv"].data_ptr(), N*3*D, 3*D, False,
                   a["dqkv"], D, 2*D, 3*D, eng.qmask_attn, eng.sched_attn_f, eng.sched_attn_b2, ws, b, N)
def timeit(fn, n=10):
    for _ in range(12): fn()          # (the first launches of a process touch the workspaces for the first time: 3 warm-ups left the first case 150 us high)
    torch.cuda.synchronize()
    H.profile_start(("mca_attn_fwd", "mca_attn_fwd_fp8", "mca_attn_quant_mxfp8", "mca_attn_bwd_onepass", "mca_attn_bwd_prep_onepass", "mca_attn_bwd_prep", "mca_attn_bwd_dq", "mca_attn_bwd_dkv", "mca_attn_bwd_dq_fp8", "mca_attn_bwd_dkv_fp8", "mca_attn_quant_bwd_mxfp8"))
    for _ in range(n): fn()
    return H.profile_stop()
def fwd8():
    eng.set_attention_dtype("fp8"); fwd(); eng.set_attention_dtype("bf16")
def bwd8():
    eng.set_attention_dtype("fp8"); bwd(); eng.set_attention_dtype("bf16")
def bwd1():
    eng.dbg["onepass"] = True; bwd(); eng.dbg["onepass"] = False
eng.dbg["onepass"] = False          # "bwd" = the two-pass form; "bwd one-pass" = attention_bwd1.hip
def fwd_tb():
    fl = eng.attn_flags; eng.attn_flags = H.ATTN_Q_PRESCALED; fwd(); eng.attn_flags = fl
def fwd_rs():
    kh = ws.pop("khot"); fwd_tb(); ws["khot"] = kh
cases = [("fwd (lazy softmax reference: production)", fwd, {}), ("fwd textbook recurrence", fwd_tb, {}), ("fwd register-staged (3 wavefronts per SIMD)", fwd_rs, {}), ("fwd fp8", fwd8, {}), ("fwd fp8 register-staged", fwd8, {9: 32}), ("bwd", bwd, {}), ("bwd one-pass", bwd1, {}), ("bwd fp8", bwd8, {})]
if os.environ.get("MCA_BENCH_ATTN_ONLY"):
    cases = [c for c in cases if c[0] == os.environ["MCA_BENCH_ATTN_ONLY"]]
if os.environ.get("MCA_BENCH_ATTN_ABLATE"):          # the one-pass backward alone: pipelined and plain kernel (knob 9 bit 64)
    cases = [("bwd one-pass", bwd1, {}), ("bwd one-pass plain kernel", bwd1, {9: 64})]
    if os.environ["MCA_BENCH_ATTN_ABLATE"] == "2":
        cases = cases[:1]
if os.environ.get("MCA_BENCH_ATTN_EXTRA"):
    cases += [("bwd no-atomics", bwd, {6: 1})]
for nm, fn, knobs in cases:
    with H.knobs(**{f"k{k}": v for k, v in knobs.items()}):
        r = timeit(fn)
    print(nm, end=" -> ")
    for k, (n, ms, fl) in r.items():
        print(f"{k}: {ms/n*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s algorithmic ({fl/ms/1e9/2500*100:.1f}% of MFMA peak)", end="  ")
    print(flush=True)

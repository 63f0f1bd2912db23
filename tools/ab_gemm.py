"""A/B of a debug knob on the NT GEMM shapes of the CMU step + correctness of both against torch."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
knob = int(sys.argv[1]) if len(sys.argv) > 1 else 7
VA, VB = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 0)          # the two knob values compared
M = 32 * 2538
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def ab(make_fn, rounds=5):
    """interleaved rounds of the two knob values; returns {value: (median ms, min ms)}"""
    res = {VA: [], VB: []}
    for _ in range(rounds):
        for v in (VA, VB):
            H.lib().mca_debug_set(knob, v)
            res[v].append(timeit(make_fn()))
    return {v: (sorted(r)[len(r) // 2], min(r)) for v, r in res.items()}
shapes = [("qkv  bf16", 1536, 512, True, False), ("out  f32+res", 512, 512, False, True), ("ff1  bf16", 2816, 512, True, False),
          ("ff2  f32+res", 512, 1408, False, True), ("dgrad ff1 f32+res", 512, 2816, False, True),
          ("dgrad qkv f32+res", 512, 1536, False, True), ("plain f32", 512, 512, False, False), ("oT dgrad bf16", 512, 512, True, False), ("bf16 N=1024", 1024, 512, True, False), ("K=64 bf16", 1536, 64, True, False),
          ("K=128 bf16", 1536, 128, True, False), ("K=192 f32", 512, 192, False, False), ("big-K bf16", 2048, 4096, True, False)]
torch.manual_seed(0)
for nm, N, K, obf, res in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    R = torch.randn(M, N, device="cuda") if res else None
    ref = A[:4096].float() @ B.float().t() + (R[:4096] if res else 0)
    ref_tail = A[-300:].float() @ B.float().t() + (R[-300:] if res else 0)
    row = f"{nm:20s} N={N:5d} K={K:5d}"
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
    mk = lambda: (lambda: H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, int(obf), None, H.ptr(R), N, 0, M, N, K, H.stream_ptr()))
    t = ab(mk)
    for v in (VA, VB):
        H.lib().mca_debug_set(knob, v); C.zero_(); mk()(); torch.cuda.synchronize()
        err = max(float((C[:4096].float() - ref).abs().max()), float((C[-300:].float() - ref_tail).abs().max())) / float(ref.abs().max())
        row += f" | knob={v}: med {t[v][0]*1e3:7.1f} min {t[v][1]*1e3:7.1f} us {2.0*M*N*K/t[v][0]/1e9:6.1f} TF err {err:.1e}"
    print(row, flush=True)
# fused GEGLU-backward GEMM
N, K = 1408, 512
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
h = torch.randn(M, 2 * N, device="cuda").bfloat16(); outs = []
row = f"{'geglu-bwd fused':20s} N={N:5d} K={K:5d}"
dh = torch.zeros(M, 2 * N, device="cuda", dtype=torch.bfloat16)
mk = lambda: (lambda: H.call("mca_gemm_nt_geglu_bwd", A.data_ptr(), K, B.data_ptr(), K, h.data_ptr(), dh.data_ptr(), 2 * N, N, M, K, H.stream_ptr()))
t = ab(mk)
for v in (VA, VB):
    H.lib().mca_debug_set(knob, v); dh.zero_(); mk()(); torch.cuda.synchronize(); outs.append(dh.clone())
    row += f" | knob={v}: med {t[v][0]*1e3:7.1f} min {t[v][1]*1e3:7.1f} us {2.0*M*N*K/t[v][0]/1e9:6.1f} TF"
# reference from the unfused formula in fp32
dg = A[:2048].float() @ B.float().t(); a_, g_ = h[:2048, :N].float(), h[:2048, N:].float()
cdf = 0.5 * (1 + torch.erf(g_ * 0.7071067811865476)); pdf = torch.exp(-0.5 * g_ * g_) * 0.3989422804014327
refdh = torch.cat([dg * g_ * cdf, dg * a_ * (cdf + g_ * pdf)], 1)
print(row, f" max diff between the two: {float((outs[0].float()-outs[1].float()).abs().max()):.2e}; rel-L2 vs fp32 formula: "
      f"{float((outs[1][:2048].float()-refdh).norm()/refdh.norm()):.2e} / {float((outs[0][:2048].float()-refdh).norm()/refdh.norm()):.2e}")

"""Times the GEMM launches of one fusion layer at the step's shapes (b = argv[1], default 32) under the kernel-selection knobs
(include/mca_hip_debug.h): which kernel form serves which shape best.  One process, interleaved rounds."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, D, Ip = b * 2538, 512, 1408
dev = "cuda"
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
f32 = lambda *s: torch.randn(*s, device=dev)
x_b, w_qkv, qkv = bf(T, D), bf(3 * D, D), torch.empty(T, 3 * D, device=dev, dtype=torch.bfloat16)
dh, w1T, dx, res = bf(T, 2 * Ip), bf(D, 2 * Ip), torch.empty(T, D, device=dev), f32(T, D)
dqkv, wqkvT = bf(T, 3 * D), bf(D, 3 * D)
dxo, w2T, h, dh_out = bf(T, D), bf(Ip, D), bf(T, 2 * Ip), torch.empty(T, 2 * Ip, device=dev, dtype=torch.bfloat16)
o_b, w_o, x1, mean, rstd, gamma = bf(T, D), bf(D, D), torch.empty(T, D, device=dev), f32(T), f32(T).abs() + 0.5, f32(D)
g_b, w2 = bf(T, Ip), bf(D, Ip)
w1, hh, gg = bf(2 * Ip, D), torch.empty(T, 2 * Ip, device=dev, dtype=torch.bfloat16), torch.empty(T, Ip, device=dev, dtype=torch.bfloat16)
S = H.stream_ptr
cases = {
    "qkv 1536x512 bf16": (lambda: H.call("mca_gemm_nt", x_b.data_ptr(), D, w_qkv.data_ptr(), D, qkv.data_ptr(), 3 * D, 1, None, None, 0, 0, T, 3 * D, D, S()), 2.0 * T * 3 * D * D),
    "dgrad ff1 512x2816 f32+res": (lambda: H.call("mca_gemm_nt", dh.data_ptr(), 2 * Ip, w1T.data_ptr(), 2 * Ip, dx.data_ptr(), D, 0, None, res.data_ptr(), D, 0, T, D, 2 * Ip, S()), 2.0 * T * D * 2 * Ip),
    "dgrad qkv 512x1536 f32+res": (lambda: H.call("mca_gemm_nt", dqkv.data_ptr(), 3 * D, wqkvT.data_ptr(), 3 * D, dx.data_ptr(), D, 0, None, res.data_ptr(), D, 0, T, D, 3 * D, S()), 2.0 * T * D * 3 * D),
    "dgrad out 512x512 bf16": (lambda: H.call("mca_gemm_nt", dxo.data_ptr(), D, w_o.data_ptr(), D, o_b.data_ptr(), D, 1, None, None, 0, 0, T, D, D, S()), 2.0 * T * D * D),
    "geglu bwd 1408x512": (lambda: H.call("mca_gemm_nt_geglu_bwd", dxo.data_ptr(), D, w2T.data_ptr(), D, h.data_ptr(), dh_out.data_ptr(), 2 * Ip, Ip, T, D, S()), 2.0 * T * Ip * D),
    "geglu fwd 2816x512": (lambda: H.call("mca_gemm_nt_geglu_fwd", x_b.data_ptr(), D, w1.data_ptr(), D, hh.data_ptr(), 2 * Ip, gg.data_ptr(), Ip, Ip, T, D, S()), 2.0 * T * 2 * Ip * D),
    "lnres out-proj 512x512": (lambda: H.call("mca_gemm_nt_lnres", o_b.data_ptr(), D, w_o.data_ptr(), D, x1.data_ptr(), D, res.data_ptr(), D, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), T, D, D, S()), 2.0 * T * D * D),
    "lnres ff2 512x1408": (lambda: H.call("mca_gemm_nt_lnres", g_b.data_ptr(), Ip, w2.data_ptr(), Ip, x1.data_ptr(), D, res.data_ptr(), D, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), T, D, Ip, S()), 2.0 * T * D * Ip),
}
variants = [("default", {}), ("knob1=1 (128x128)", {"k1": 1}), ("knob7=1 (one tile per WG)", {"k7": 1}), ("knob7=3 (persistent +res)", {"k7": 3})]
if len(sys.argv) > 2:
    variants = [("default", {})] + [(a, dict(kv.split("=") for kv in a.split(","))) for a in sys.argv[2:]]
    variants = [(n, {k: int(v) for k, v in d.items()}) for n, d in variants]
res_t = {}
for rnd in range(3):
    for vn, kn in variants:
        with H.knobs(**kn):
            for cn, (fn, fl) in cases.items():
                for _ in range(2): fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                res_t.setdefault((cn, vn), []).append(s.elapsed_time(e) / 10 * 1e3)
for cn, (fn, fl) in cases.items():
    print(cn)
    for vn, _ in variants:
        us = min(res_t[(cn, vn)])
        print(f"    {vn:32s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")

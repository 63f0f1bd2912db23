"""The grouped weight-gradient GEMM at the step's shapes (b = argv[1], default 32): time, algorithmic TFLOP/s and the in-kernel
shader clock (knob 9 bit 8: two stamps per workgroup, read back through mca_dbg_trace_read_gemm), with the launch-shape knobs
(XCD remap off, row splits).  Interleaved rounds in one process.  (Round 3's ablation builds - no atomics 514 us, the DMA stream
alone 300 us, the LDS reads + MFMAs alone 380 us of 535 us - are recorded in DESIGN.md; the kernel no longer carries them.)"""
import importlib, os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, D, Ip = b * 2538, 512, 1408
dev = "cuda"
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
dqkv, xn, dh, x1n, dxo, g = bf(T, 3 * D), bf(T, D), bf(T, 2 * Ip), bf(T, D), bf(T, D), bf(T, Ip)
gq, gw1, gw2 = torch.zeros(3 * D, D, device=dev), torch.zeros(2 * Ip, D, device=dev), torch.zeros(D, Ip, device=dev)
members = [(dqkv, xn, gq, 3 * D, D), (dh, x1n, gw1, Ip, D), (dh[:, Ip:], x1n, gw1[Ip:], Ip, D), (dxo, g, gw2, D, Ip)]
arr = (H.TnDesc * len(members))()
fl = 0.0
for d, (A, B, Cg, N, K) in zip(arr, members):
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.N, d.K = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), Cg.data_ptr(), Cg.stride(0), N, K
    fl += 2.0 * T * N * K
run = lambda: H.call("mca_gemm_tn_acc_group", C.byref(arr), len(members), T, H.stream_ptr())
import numpy as np
L = H.lib(); rd = L.mca_dbg_trace_read_gemm; rd.restype = C.c_int; rd.argtypes = [C.c_void_p, C.c_int]
def clock_ghz():          # median over the workgroups of the last launch: shader cycles per 100 MHz tick
    buf = np.zeros(1024, dtype=np.uint64); rd(buf.ctypes.data, 1024)
    cy, tk = buf[0::2][:240].astype(np.float64), buf[1::2][:240].astype(np.float64)
    ok = tk > 0
    return float(np.median(cy[ok] / tk[ok]) * 0.1) if ok.any() else float("nan")
variants = [("whole kernel", {}), ("no XCD remap", {"k9": 16})] + [(f"splits={s}", {"k3": s}) for s in (4, 6, 8, 10)]
if os.environ.get("MCA_ABLATE_ONLY"):          # one variant (for a counter pass)
    variants = [v for v in variants if v[0] == os.environ["MCA_ABLATE_ONLY"]]
res, clk = {}, {}
for rnd in range(3):
    for vn, kn in variants:
        with H.knobs(**{**kn, "k9": kn.get("k9", 0) | 8}):          # (the probe: two stamps per workgroup)
            for _ in range(2): run()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(40): run()
            e.record(); torch.cuda.synchronize()
            res.setdefault(vn, []).append(s.elapsed_time(e) / 40 * 1e3)
            clk.setdefault(vn, []).append(clock_ghz())          # of the last of the 40 back-to-back launches
for vn, _ in variants:
    us = min(res[vn])
    print(f"{vn:28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s   in-kernel clock {np.median(clk[vn]):.2f} GHz", flush=True)

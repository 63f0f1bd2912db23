"""Timeline of workgroup 0 / wave 0 of the persistent GEGLU-backward GEMM (knob 0 = 8): k-loop vs epilogue per tile."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); L = H.lib()
M, N, K = 32 * 2538, 1408, 512
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
h = torch.randn(M, 2 * N, device="cuda").bfloat16(); dh = torch.zeros(M, 2 * N, device="cuda", dtype=torch.bfloat16)
L.mca_debug_set(0, 8)
for _ in range(3):
    H.call("mca_gemm_nt_geglu_bwd", A.data_ptr(), K, B.data_ptr(), K, h.data_ptr(), dh.data_ptr(), 2 * N, N, M, K, H.stream_ptr())
torch.cuda.synchronize()
buf = (C.c_ulonglong * 1024)()
fn = L.mca_dbg_trace_read_gemm; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, 1024) == 0
nkt = K // 64; per = 1 + 3 * nkt + 3; t = list(buf)
for n, i in enumerate(range(0, 1024 - per, per)):
    seg = t[i:i + per]
    if seg[-1] == 0 or (i and seg[0] < t[i - 1]): break
    kloop = seg[3 * nkt] - seg[0]
    print(f"tile {n}: total {seg[-1] - seg[0]:6d} cycles | k-loop {kloop:6d} | end barrier {seg[-3] - seg[-4]:5d} | next-tile DMA issue + preload wait {seg[-2] - seg[-3]:5d} | epilogue {seg[-1] - seg[-2]:6d}")
    if n >= 13: break

"""Timeline of workgroup 0 / wave 0 of the persistent NT kernel (s_memtime stamps, knob 0 bit 3)."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); L = H.lib()
M = 32 * 2538
N, K, obf = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1536, 512, 1)
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
Cm = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
L.mca_debug_set(0, 8)
for _ in range(3):
    H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, Cm.data_ptr(), N, obf, None, None, 0, 0, M, N, K, H.stream_ptr())
torch.cuda.synchronize()
buf = (C.c_ulonglong * 1024)()
fn = L.mca_dbg_trace_read_gemm; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, 1024) == 0
nkt = K // 64
per = 1 + 3 * nkt + 3
t = list(buf)
ntiles = 0
print(f"N={N} K={K}: stamps per tile {per}; s_memtime ticks (100 MHz => 10 ns each)")
for i in range(0, 1024 - per, per):
    seg = t[i:i + per]
    if seg[-1] == 0 or (i and seg[0] < t[i - 1]): break
    base = seg[0]
    steps = [(seg[1 + 3 * k] - (seg[3 * k] if k else base), seg[2 + 3 * k] - seg[1 + 3 * k], seg[3 + 3 * k] - seg[2 + 3 * k]) for k in range(nkt)]
    e0, e1, e2 = seg[-3] - seg[-4], seg[-2] - seg[-3], seg[-1] - seg[-2]
    print(f"tile {ntiles}: total {seg[-1] - base:5d} | k-steps (vmwait, barrier, compute): " + " ".join(f"({a},{b},{c})" for a, b, c in steps) +
          f" | end barrier {e0}, preload+dma issue {e1}, epilogue {e2}")
    ntiles += 1
    if ntiles >= 16: break

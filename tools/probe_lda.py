"""Does the row stride of A (power of two vs padded) change the NT GEMM time?  (L2 channel spread of the LDS-DMA requests)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
M = 32 * 2538
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for N, K, obf in [(1536, 512, 1), (2816, 512, 1), (512, 512, 0), (512, 1408, 0), (512, 2816, 0)]:
    for pa, pb, pc in [(0, 0, 0), (64, 0, 0), (64, 64, 0), (64, 64, 64), (8, 8, 8), (32, 32, 32)]:
        A = torch.randn(M, K + pa, device="cuda").bfloat16(); B = torch.randn(N, K + pb, device="cuda").bfloat16()
        C = torch.empty(M, N + pc, device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
        ms = timeit(lambda: H.call("mca_gemm_nt", A.data_ptr(), K + pa, B.data_ptr(), K + pb, C.data_ptr(), N + pc, obf, None, None, 0, 0, M, N, K, H.stream_ptr()))
        print(f"N={N:5d} K={K:5d} pad A/B/C = {pa:2d}/{pb:2d}/{pc:2d}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)

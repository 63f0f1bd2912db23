"""NT GEMM time vs K at fixed M, N: the intercept is the per-tile fixed cost (prologue + epilogue), the slope the k-step cost."""
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
for N, obf, res in [(1536, True, False), (512, False, True), (2816, True, False)]:
    for K in (64, 128, 256, 512, 1024, 2048):
        A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
        R = torch.randn(M, N, device="cuda") if res else None
        ms = timeit(lambda: H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, int(obf), None, H.ptr(R), N, 0, M, N, K, H.stream_ptr()))
        tiles = ((M + 255) // 256) * (N // 128)
        print(f"N={N:5d} {'bf16' if obf else 'f32+res'} K={K:5d}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF   per tile-round {ms*1e3/(tiles/256):6.2f} us", flush=True)

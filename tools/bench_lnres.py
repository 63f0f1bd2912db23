"""mca_gemm_nt_lnres (out-proj K = 512, FF2 K = 1408) and the data-gradient GEMMs with residual on the step's shapes."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
M, N = 32 * 2538, 512
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = torch.randn(M, N, device="cuda"); gamma = torch.randn(N, device="cuda")
mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
C = torch.empty(M, N, device="cuda")
for K in (512, 1408):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
    us = timeit(lambda: H.call("mca_gemm_nt_lnres", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, x.data_ptr(), N, mean.data_ptr(),
                               rstd.data_ptr(), gamma.data_ptr(), M, N, K, H.stream_ptr()))
    print(f"lnres K={K}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s  HBM {(M * K * 2 + 2 * M * N * 4) / us / 1e6:.2f} TB/s")
for K in (1536, 2816):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
    us = timeit(lambda: H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, 0, None, x.data_ptr(), N, 0, M, N, K, H.stream_ptr()))
    print(f"fp32 + residual K={K}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s  HBM {(M * K * 2 + 2 * M * N * 4) / us / 1e6:.2f} TB/s")

"""Micro-benchmark of the GEMM entry points on the CMU-step shapes (M = 32*2538 token rows)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
M = 32 * 2538
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [("qkv  bf16", 1536, 512, True, False), ("out  f32+res", 512, 512, False, True), ("ff1  bf16", 2816, 512, True, False),
          ("ff2  f32+res", 512, 1408, False, True), ("dgrad ff2 bf16", 1408, 512, True, False), ("dgrad ff1 f32+res", 512, 2816, False, True),
          ("dgrad qkv f32+res", 512, 1536, False, True), ("plain f32", 512, 512, False, False), ("big-K bf16", 2048, 4096, True, False)]
names = ["mca_gemm_nt"]
pfs = [1, 0]   # knob 4: 1 = residual prefetch off
for nm, N, K, obf, res in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if obf else torch.float32)
    R = torch.randn(M, N, device=dev) if res else None
    fl = 2.0 * M * N * K
    row = f"{nm:20s} N={N:5d} K={K:5d}"
    for pf in pfs:
        H.lib().mca_debug_set(4, pf)
        fn_name = "mca_gemm_nt"
        ms = timeit(lambda: H.call(fn_name, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, int(obf), None, H.ptr(R), N, 0, M, N, K, H.stream_ptr()))
        row += f" | {['res-prefetch','no-prefetch'][pf]} {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF"
    ms = timeit(lambda: torch.matmul(A, B.t()))
    row += f" | torch.matmul(bf16 out) {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF"
    print(row, flush=True)
# weight-gradient shapes
for nm, N, K in [("wgrad qkv", 1536, 512), ("wgrad ff1", 2816, 512), ("wgrad ff2", 512, 1408), ("wgrad out", 512, 512)]:
    A = torch.randn(M, N, device=dev).bfloat16(); B = torch.randn(M, K, device=dev).bfloat16()
    C = torch.zeros(N, K, device=dev)
    fl = 2.0 * M * N * K
    ms = timeit(lambda: H.call("mca_gemm_tn_acc", A.data_ptr(), N, B.data_ptr(), K, C.data_ptr(), K, M, N, K, H.stream_ptr()))
    ms2 = timeit(lambda: torch.matmul(A.t(), B))
    print(f"{nm:20s} N={N:5d} K={K:5d} | tn_acc {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF | torch {ms2*1e3:7.1f} us {fl/ms2/1e9:7.1f} TF", flush=True)

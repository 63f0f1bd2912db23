import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
M = 32 * 2538
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for nm, N, K in [("wgrad qkv", 1536, 512), ("wgrad ff1", 2816, 512), ("wgrad ff2", 512, 1408), ("wgrad out", 512, 512)]:
    A = torch.randn(M, N, device="cuda").bfloat16(); B = torch.randn(M, K, device="cuda").bfloat16()
    C = torch.zeros(N, K, device="cuda")
    fl = 2.0 * M * N * K
    row = f"{nm:12s}"
    for label, k5, k3 in [("256x256", 0, 0), ("256x128", 2, 0), ("128", 1, 0), ("256x256 s=10", 0, 10), ("256x256 s=16", 0, 16), ("256x256 s=32", 0, 32), ("256x256 s=42", 0, 42)]:
        H.lib().mca_debug_set(5, k5); H.lib().mca_debug_set(3, k3)
        ms = timeit(lambda: H.call("mca_gemm_tn_acc", A.data_ptr(), N, B.data_ptr(), K, C.data_ptr(), K, M, N, K, H.stream_ptr()))
        row += f" | {label} {ms*1e3:6.1f}us {fl/ms/1e9:5.0f}TF"
    print(row, flush=True)

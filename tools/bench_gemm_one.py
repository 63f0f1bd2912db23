import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
M = 32 * 2538
for N, K, obf in [(2048, 4096, True), (1536, 512, True), (2816, 512, True)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, 1, None, None, 0, 0, M, N, K, H.stream_ptr())
    torch.cuda.synchronize()
for N, K in [(1536, 512), (2816, 512)]:
    A = torch.randn(M, N, device="cuda").bfloat16(); B = torch.randn(M, K, device="cuda").bfloat16(); C = torch.zeros(N, K, device="cuda")
    for _ in range(5):
        H.call("mca_gemm_tn_acc", A.data_ptr(), N, B.data_ptr(), K, C.data_ptr(), K, M, N, K, H.stream_ptr())
    torch.cuda.synchronize()

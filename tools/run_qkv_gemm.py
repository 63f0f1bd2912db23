"""One GEMM shape, a few launches (for rocprofv3 --pmc): usage run_qkv_gemm.py N K out_bf16 [knob7]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()
M = 32 * 2538
N, K, obf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4: H.lib().mca_debug_set(7, int(sys.argv[4]))
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
for _ in range(5):
    H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, obf, None, None, 0, 0, M, N, K, H.stream_ptr())
torch.cuda.synchronize()

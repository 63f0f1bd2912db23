"""The weight-gradient GEMM on the exact shapes / strides of the CMU step, per TN kernel variant (knob 5)."""
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
shapes = [("qkv", 1536, 512, 1536, 512), ("ff1 half", 1365, 512, 2816, 512), ("ff2", 512, 1365, 512, 1408), ("out", 512, 512, 512, 512)]
bufs = [(torch.randn(M, lda, device="cuda").bfloat16(), torch.randn(M, ldb, device="cuda").bfloat16(), torch.zeros(N, K, device="cuda")) for _, N, K, lda, ldb in shapes]
for rnd in range(2):
    for knob, label in ((2, "256x128"), (0, "256x256 where >= 8 tiles")):
        H.lib().mca_debug_set(5, knob)
        tot = 0.0; row = ""
        for (nm, N, K, lda, ldb), (A, B, C) in zip(shapes, bufs):
            ms = timeit(lambda: H.call("mca_gemm_tn_acc", A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), K, M, N, K, H.stream_ptr()))
            tot += ms * (2 if nm == "ff1 half" else 1)
            row += f" {nm} {ms*1e3:6.1f}us {2.0*M*N*K/ms/1e9:4.0f}TF |"
        print(f"{label:26s}|{row} per layer {tot*1e3:6.1f} us", flush=True)

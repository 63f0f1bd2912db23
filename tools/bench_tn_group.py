"""The grouped weight-gradient launch on the exact members of a CMU layer (with the out-projection; the top layer's also with the
pooling key / value projection), per row partition: uniform splits (knob 3) against the balanced partition with a relief of r rows
for the workgroups that take more than one tile (knob 6 = r / 32 + 1; negative: without owner segments, a purely tile-major line).  usage: bench_tn_group.py [batch ...]"""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); H.lib()


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


layer = [(1536, 512, 1536, 512), (1365, 512, 2816, 512), (1365, 512, 2816, 512), (512, 1365, 512, 1408), (512, 512, 512, 512)]
for b in [int(a) for a in sys.argv[1:]] or [8, 32]:
    R = b * 2538
    for label, members in (("layer (5 members, 52 tiles)", layer), ("top layer (+ pooling kv, 60 tiles)", layer + [(1024, 512, 1024, 512)]),
                           ("round 3 (4 members, 48 tiles)", layer[:4])):
        arr = (H.TnDesc * len(members))(); keep = []; fl = 0.0
        for d, (N, K, lda, ldb) in zip(arr, members):
            A = torch.randn(R, lda, device="cuda").bfloat16(); B = torch.randn(R, ldb, device="cuda").bfloat16(); Cg = torch.zeros(N, K, device="cuda")
            d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.N, d.K = A.data_ptr(), lda, B.data_ptr(), ldb, Cg.data_ptr(), K, N, K
            keep.append((A, B, Cg)); fl += 2.0 * R * N * K
        run = lambda: H.call("mca_gemm_tn_acc_group", C.byref(arr), len(members), R, H.stream_ptr())
        row = f"b={b:3d} {label:36s}"
        for rnd in range(2):
            for k3, k6, nm in ((4, 0, "4 splits"), (5, 0, "5 splits"), (0, -49, "line r=1536"), (0, 1, "own r=0"), (0, 17, "r=512"), (0, 33, "r=1024"), (0, 49, "r=1536"), (0, 65, "r=2048")):
                H.lib().mca_debug_set(3, k3); H.lib().mca_debug_set(6, k6)
                us = timeit(run)
                row += f" | {nm} {us:6.1f}"
            row += "\n" + " " * 44
        H.lib().mca_debug_set(3, 0); H.lib().mca_debug_set(6, 0)
        print(row.rstrip() + f"   ({fl / 1e9:.0f} GF)", flush=True)
        del keep

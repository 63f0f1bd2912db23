"""LayerNorm forward / backward on the trunk's shape (b*N x 512; b = argv[1], default 32), HBM rate; the backward under workgroup
caps (knob 14: its dgamma tail is one atomic per column and workgroup)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); E = importlib.import_module("mca-paper_amd.engine")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, D = b * 2538, 512
x = torch.randn(T, D, device="cuda"); g = torch.randn(D, device="cuda"); yb = torch.empty(T, D, device="cuda", dtype=torch.bfloat16)
dy = torch.randn(T, D, device="cuda"); dx = torch.empty(T, D, device="cuda"); dxb = torch.empty(T, D, device="cuda", dtype=torch.bfloat16)
dg = torch.zeros(D, device="cuda")
m = torch.empty(T, device="cuda"); r = torch.empty(T, device="cuda")
def fwd(): E.FusionEngine.ln_fwd(x, g, T, D, m, r, y_bf16=yb)
def bwd(): E.FusionEngine.ln_bwd(dy, D, x, g, m, r, T, D, dg, dx=dx, dx_bf16=dxb)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
us = timeit(fwd)
print(f"ln_fwd {us:.1f} us  {(T * D * 6 + T * 8) / us / 1e6:.2f} TB/s")
for rnd in range(2):
    for cap in (1024, 512, 384, 0, 128):
        with H.knobs(k14=cap):
            us = timeit(bwd)
        print(f"ln_bwd workgroups <= {cap or 256:5d}: {us:.1f} us  {(T * D * 14) / us / 1e6:.2f} TB/s", flush=True)

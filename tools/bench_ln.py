"""LayerNorm forward / backward on the trunk's shape (b*N x 512), HBM rate."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); E = importlib.import_module("mca-paper_amd.engine")
T, D = 32 * 2538, 512
x = torch.randn(T, D, device="cuda"); g = torch.randn(D, device="cuda"); yb = torch.empty(T, D, device="cuda", dtype=torch.bfloat16)
m = torch.empty(T, device="cuda"); r = torch.empty(T, device="cuda")
H.lib().mca_debug_set(12, int(os.environ.get("LN_GENERAL", "0")))
def fwd(): E.FusionEngine.ln_fwd(x, g, T, D, m, r, y_bf16=yb)
for _ in range(3): fwd()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): fwd()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
print(f"ln_fwd {us:.1f} us  {(T * D * 6 + T * 8) / us / 1e6:.2f} TB/s")

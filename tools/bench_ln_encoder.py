"""The general LayerNorm backward on an encoder's output norm (rows = b x tokens, 512 columns, pad mask, positional period, beta,
dx + bf16 copy + the Linear's bias gradient) under workgroup caps (knob 14), and the parameter-only form of the input norm.
usage: bench_ln_encoder.py [batch]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module("mca-paper_amd.hip"); E = importlib.import_module("mca-paper_amd.engine"); H.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for n, cin in ((1500, 74), (450, 713)):
    rows, D, N = b * n, 512, 2538
    dyfull = torch.randn(b, N, D, device="cuda"); y = torch.randn(rows, D, device="cuda"); gam = torch.randn(D, device="cuda")
    m, r = torch.randn(rows, device="cuda"), torch.rand(rows, device="cuda") + 0.5
    mask = (torch.rand(rows, device="cuda") < 0.2).to(torch.uint8)
    dx, dxb = torch.empty(rows, D, device="cuda"), torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    dg, db, ds = (torch.zeros(D, device="cuda") for _ in range(3))
    fn = lambda: E.FusionEngine.ln_bwd(dyfull, D, y, gam, m, r, rows, D, dg, dbeta=db, rowmask=mask, dx=dx, dx_bf16=dxb, y_bstride=N * D, period=n, dxsum=ds)
    row = f"output norm {rows:6d} x 512:"
    for cap in (0, 128, 512, 1024):
        H.lib().mca_debug_set(14, cap)
        row += f"  cap {cap or 256:4d}: {timeit(fn):5.1f} us"
    H.lib().mca_debug_set(14, 0)
    print(row, flush=True)
    x = torch.randn(rows, cin, device="cuda"); dyi = torch.randn(rows, ((cin + 7) // 8) * 8, device="cuda"); g2 = torch.randn(cin, device="cuda")
    dg2, db2 = torch.zeros(cin, device="cuda"), torch.zeros(cin, device="cuda")
    fn2 = lambda: E.FusionEngine.ln_bwd(dyi, dyi.stride(0), x, g2, m, r, rows, cin, dg2, dbeta=db2, rowmask=mask)
    t_new = timeit(fn2)
    H.lib().mca_debug_set(12, 1); t_old = timeit(fn2); H.lib().mca_debug_set(12, 0)
    print(f"input norm {rows:6d} x {cin:3d}, parameter gradients only: column-parallel {t_new:5.1f} us, row form {t_old:5.1f} us", flush=True)

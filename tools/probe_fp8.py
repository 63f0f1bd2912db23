"""Layout / rate probe of v_mfma_scale_f32_32x32x64_f8f6f4 (see probe_fp8.hip); builds its .so on the spot.  usage: probe_fp8.py"""
import ctypes as C, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/probe_fp8.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(here, "probe_fp8.hip")])
lib = C.CDLL(so)
out = torch.zeros(142 * 64 * 16, device="cuda")
assert lib.probe_fp8_layout(C.c_void_p(out.data_ptr())) == 0
o = out.view(142, 64, 16).cpu()
def dense(t):          # D[row][col] under the standard 32x32 C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    D = torch.zeros(32, 32)
    for lane in range(64):
        for r in range(16):
            D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31] = o[t, lane, r]
    return D
print("all-ones x all-ones (expect 64 everywhere):", dense(128).unique().tolist())
D = dense(129); print("A byte0=1 in every lane, B ones: D values", D.unique().tolist(), "(2 => byte 0 of lanes l and l+32 feed the same row at two k)")
rows = torch.zeros(64, dtype=torch.long); cols = torch.zeros(64, dtype=torch.long)
# row of lane l: bit q of the row index ... derive per-lane row by one-hot over lane bits: D[row] > 0 for rows fed by lanes with bit q set
row_bits = [dense(130 + q)[:, 0] for q in range(6)]          # per row: how many lanes with bit q feed it (0, 1 or 2)
col_bits = [dense(136 + q)[0, :] for q in range(6)]
print("rows fed by lanes with bit q set (q=0..5):"); [print("  q", q, row_bits[q].int().tolist()) for q in range(6)]
print("cols fed by lanes with bit q set (q=0..5):"); [print("  q", q, col_bits[q].int().tolist()) for q in range(6)]
# k pairing: A one-hot (half, byte) meets B (half', byte') with value codes
pair = {}
for t in range(64):
    D1, D2 = dense(t), dense(64 + t)
    v1, v2 = float(D1[0, 0]), float(D2[0, 0])          # row 0 (A lane 0 / 32 -> row 0), col 0: B lanes 0 and 32
    # the same for every column?
    same = bool((D1[0] == v1).all()) and bool((D2[0] == v2).all())
    lo = int(v1) - 1; code = int(v2) - 1; hi, halfB = code & 1, code >> 1
    pair[(t >> 5, t & 31)] = (halfB, lo + 16 * hi, same, float(D1[1:].abs().max()))
ident = all(v[0] == k[0] and v[1] == k[1] for k, v in pair.items())
print("A(half, byte) -> B(half, byte) met in the K sum: identity" if ident else "NOT identity:", "" if ident else pair)
print("  (row-0 result identical over all 32 columns:", all(v[2] for v in pair.values()), "; other rows zero:", all(v[3] == 0 for v in pair.values()), ")")
# conversion
x = torch.tensor([0.0, 1.0, 448.0, 500.0, 1e-3, 2.0 ** -9, 2.0 ** -10, 300.0, -1.5, 0.0624, 17.0, 255.0], device="cuda")
oi = torch.zeros(6, dtype=torch.int32, device="cuda")
assert lib.probe_fp8_cvt(C.c_void_p(x.data_ptr()), C.c_void_p(oi.data_ptr()), 6) == 0
bits = oi.cpu().tolist()
fp8 = torch.tensor([[b & 0xff, (b >> 8) & 0xff] for b in bits], dtype=torch.uint8).flatten().view(torch.float8_e4m3fn).float()
print("cvt_pk_fp8_f32:", list(zip(x.cpu().tolist(), fp8.tolist())))
# rate
lib.probe_fp8_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
o2 = torch.zeros(4, device="cuda"); st = torch.zeros(4, dtype=torch.int64, device="cuda")
names = {0: "fp8 scaled 32x32x64", 1: "bf16 32x32x16", 2: "fp8 32x32x16"}
flop = {0: 2 * 32 * 32 * 64, 1: 2 * 32 * 32 * 16, 2: 2 * 32 * 32 * 16}
for mode in range(3):
    for blocks, threads in ((1, 64), (256, 256), (256, 512)):
        iters = 20000; ms = C.c_float()
        assert lib.probe_fp8_rate(mode, blocks, threads, iters, o2.data_ptr(), st.data_ptr(), C.byref(ms)) == 0
        n = iters * 4; cyc = st.tolist()[0]
        print(f"{names[mode]:22s} {blocks:4d}x{threads:4d}: {ms.value:8.3f} ms  cycles/MFMA/wave {cyc / n:6.2f}  "
              f"{blocks * (threads // 64) * n * flop[mode] / (ms.value * 1e-3) / 1e12:8.1f} TFLOP/s  clock {cyc / ms.value / 1e3:7.1f} MHz", flush=True)

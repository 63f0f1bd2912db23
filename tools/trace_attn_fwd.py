"""Per-phase cycle anatomy of one wavefront of the forward attention kernel (s_memtime stamps, knob 8 = 8)."""
import ctypes as C, importlib, os, sys, torch
# needs the trace build: python mca-paper_amd/build.py --trace; it is picked up here through MCA_HIP_LIB
os.environ.setdefault("MCA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mca-paper_amd", "libmca_hip_trace.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
b = 32
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(0)
model = P.MCA(**cfg).cuda(); eng = model.engine
ws = eng.workspace(b); N, D = eng.N, eng.D
ws["padding"].zero_()
H.call("mca_build_keyinfo", ws["padding"].data_ptr(), eng.kgroup.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr(), b, N, eng.nk_pad, H.stream_ptr())
a = ws["layers"][0]
a["qkv"].copy_(torch.randn_like(a["qkv"].float()).bfloat16())
L = H.lib(); L.mca_debug_set(8, 8)
for _ in range(3):
    eng._attn_fwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 1024)()
fn = L.mca_dbg_trace_read_attn_fwd; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, 1024) == 0
t = list(buf); per = 7
names = ["S mfma (+gload issue)", "mask", "max+exp+pack", "rescale+PV mfma", "swrite", "barrier"]
rows = []
for i in range(0, 1024 - per, per):
    seg = t[i:i + per]
    if seg[-1] == 0 or seg[-1] < seg[0]: break
    nxt = t[i + per] if t[i + per] else seg[-1]
    rows.append([seg[k + 1] - seg[k] for k in range(6)] + [nxt - seg[0]])
print(f"{len(rows)} key tiles traced (workgroup 3 of head 0 / sample 0, wave 0)")
for r in rows[:24]: print("  " + "  ".join(f"{n}={v}" for n, v in zip(names, r[:6])) + f"   | iteration {r[6]}")
import statistics as st
print("median per phase:", {n: st.median(r[k] for r in rows) for k, n in enumerate(names)}, "iteration", st.median(r[6] for r in rows))
print(f"workgroup loop: {t[1021]} cycles in {t[1020]} ticks of 100 MHz -> shader clock {t[1021] / max(t[1020], 1) * 100:.0f} MHz")

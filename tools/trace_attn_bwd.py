"""Per-phase cycle anatomy of one wavefront of the backward attention kernel (s_memtime stamps, knob 6 = 8)."""
import ctypes as C, importlib, os, sys, statistics as st, torch
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
a["qkv"].copy_(torch.randn_like(a["qkv"].float()).bfloat16()); ws["do"].copy_(torch.randn_like(ws["do"].float()).bfloat16())
eng._attn_fwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N)
L = H.lib(); L.mca_debug_set(6, 8)
for _ in range(2):
    ws["dq32"].zero_()
    eng._attn_bwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], ws["do"], a["lse"], ws["delta"], ws["dq32"], N*D, a["dqkv"], D, 2*D, 3*D, eng.qmask_attn, eng.sched_attn_b, ws, b, N)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 1024)()
fn = L.mca_dbg_trace_read_attn_bwd; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, 1024) == 0
t = list(buf)
names = ["S,dP sub0", "P,dS valu sub0", "dS write+dV,dK sub0", "S,dP sub1", "valu sub1", "dV,dK sub1", "stage write", "barrier", "dQ mfma", "dQ atomics"]
rows = []
for i in range(0, 1024 - 11, 11):          # 11 stamps per 64-query step
    seg = t[i:i + 11]
    if seg[-1] == 0 or seg[-1] < seg[0]: break
    rows.append([seg[k + 1] - seg[k] for k in range(10)] + [seg[10] - seg[0]])
print(f"{len(rows)} query steps traced")
for r in rows[:8]: print("  " + "  ".join(f"{n}={v}" for n, v in zip(names, r[:10])) + f"  | step {r[10]}")
print("median:", {n: st.median(r[k] for r in rows) for k, n in enumerate(names)}, "step", st.median(r[10] for r in rows))
if t[1003] > t[1000] > 0:
    print(f"workgroup: prologue {t[1001] - t[1000]} cycles, loop {t[1002] - t[1001]}, epilogue (stores retired) {t[1003] - t[1002]}")
    print(f"  prologue: scalar chain + first-tile load issue {t[1004] - t[1000]}, K image written {t[1005] - t[1004]}, first tile written {t[1006] - t[1005]}, barrier + entries {t[1001] - t[1006]}")

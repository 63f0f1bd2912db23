"""Where a key tile's cycles go in the query-block forward attention kernel (attention_fwd64.hip): s_memtime stamps of wavefront 0
of one workgroup (trace build: python mca-paper_amd/build.py --trace; knob 8 = 8).  Per loop iteration six stamps: loop top |
even step | vmcnt wait | s_barrier | DMA job set-up | odd step.  Read the SHARES, not the length (the stamps fence the
schedule)."""
import ctypes as C, importlib, os, statistics as st, sys
os.environ.setdefault("MCA_DEBUG", "fwd64=1")
import torch
os.environ.setdefault("MCA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mca-paper_amd", "libmca_hip_trace.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = P.config.cmu_model_config(batch_size=b); cfg["depth"] = 1
torch.manual_seed(0)
eng = P.MCA(**cfg).cuda().engine
ws = eng.workspace(b); N, D = eng.N, eng.D
ws["padding"].zero_()
H.call("mca_build_keyinfo", ws["padding"].data_ptr(), eng.kgroup.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr(), b, N, eng.nk_pad, H.stream_ptr())
H.call("mca_build_keyhot", ws["keyinfo"].data_ptr(), ws["khot"].data_ptr(), b, eng.nk_pad, H.stream_ptr())
a = ws["layers"][0]
a["qkv"].copy_(torch.randn(b * N, 3 * D, device="cuda").bfloat16()); a["qkv"][:, :D] *= 0.18
L = H.lib(); L.mca_debug_set(8, 8); L.mca_debug_set(9, 8)
for _ in range(3):
    eng._attn_fwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 1024)()
fn = L.mca_dbg_trace_read_attn_fwd64; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, 1024) == 0
t = list(buf); n = int(t[1023]); per = 9
names = ["even: maxima", "even: 4 PV mfma + decision", "even: slots 0-7", "even: slots 8-15", "vmcnt wait", "barrier", "job set-up", "odd step (+ DMA pieces)", "loop back"]
rows = []
for i in range(0, n - per, per):
    seg = t[i:i + per + 1]
    rows.append([seg[k + 1] - seg[k] for k in range(per)])
print(f"{len(rows)} tiles traced")
for r in rows[:12]: print("  " + "  ".join(f"{nm}={v}" for nm, v in zip(names, r)) + f"   | tile {sum(r)}")
print("median:", {nm: st.median(r[k] for r in rows) for k, nm in enumerate(names)}, "tile", st.median(sum(r) for r in rows))

"""Where a key block's cycles go OUTSIDE the step loop of the one-pass attention backward (attention_bwd1.hip): s_memtime stamps
of wavefront 0 of workgroup 0 (trace build: python mca-paper_amd/build.py --trace; knob 9 bit 8).  Per key block eight stamps:
top | fragments arrived | first barrier | loop start | loop end | end barrier | next block requested | epilogue stores issued.
usage: trace_bwd1.py [batch]"""
import ctypes as C, importlib, os, sys
os.environ.setdefault("MCA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mca-paper_amd", "libmca_hip_trace.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
a["qkv"].copy_(torch.randn_like(a["qkv"].float()).bfloat16()); a["qkv"][:, :D] *= 0.18
ws["do"].copy_(torch.randn_like(ws["do"].float()).bfloat16())
eng._attn_fwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N)
eng.dbg["onepass"] = True
L = H.lib(); L.mca_debug_set(9, 8)
for _ in range(6):
    eng._attn_bwd2(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], ws["do"], a["lse"], ws["delta"], a["dqkv"].data_ptr(), N*3*D, 3*D, False,
                   a["dqkv"], D, 2*D, 3*D, eng.qmask_attn, eng.sched_attn_f, eng.sched_attn_b2, ws, b, N)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 1024)()
fn = L.mca_dbg_trace_read_attn_bwd1; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, 1024) == 0
t = list(buf); n = int(t[1023]); per = 9
names = ["wait for fragments", "zero + first barrier", "first-step init", "LOOP", "end barrier", "request next block", "epilogue", "(to next block's top)"]
tot = [0] * 8; steps = 0
print(f"{n // per} key blocks traced (cycles of the 100 MHz-independent shader counter s_memtime)")
for i in range(0, n - per + 1, per):
    seg = t[i:i + 8]; n_it = t[i + 8]
    nxt = t[i + per] if i + per + 8 <= n else seg[7]
    d = [seg[k + 1] - seg[k] for k in range(7)] + [nxt - seg[7]]
    for k in range(8): tot[k] += d[k]
    steps += n_it
    print(f"  block {i // per:2d}: {n_it:3d} iterations, " + "  ".join(f"{nm}={v}" for nm, v in zip(names, d)) + f"  | per iteration {d[3] / max(n_it, 1):.0f}")
print("totals:", {nm: v for nm, v in zip(names, tot)}, "iterations", steps, "loop cycles per iteration", round(tot[3] / max(steps, 1)))

# one iteration (the ninth) of every block with more than eight: stamps at the head of slot 0 | the barrier's slot | the slot behind
# it | slot 24 | 44 | 64 | end of the iteration
print("inside iteration 8 of a block (cycles): slots 0..barrier | the barrier's slot | ..24 | ..44 | ..64 | ..end")
for blk in range(n // per):
    seg = t[512 + blk * 8: 512 + blk * 8 + 7]
    if seg[0] and all(seg[k + 1] >= seg[k] for k in range(6)):
        print(f"  block {blk:2d}: " + "  ".join(str(seg[k + 1] - seg[k]) for k in range(6)) + f"  | total {seg[6] - seg[0]}")

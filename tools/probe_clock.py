"""Issue-rate / clock probe on the GPU box (see probe_clock.hip).  usage: probe_clock.py"""
import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "probe_clock.so"))
lib.probe_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
out = torch.tensor([1.0001, 1e-6, 0, 0], device="cuda"); stamps = torch.zeros(4, dtype=torch.int64, device="cuda")
names = {0: "mfma 32x32x16 x4", 1: "v_exp_f32 x16", 2: "v_fma_f32 x16", 3: "v_pk_mul_f32 x16", 4: "mfma + 4 exp",
         5: "mfma + 8 fma", 6: "mfma + 4 exp + 8 fma", 7: "mfma 16x16x32 x4"}
per_iter = {0: 4, 1: 16, 2: 16, 3: 16, 4: 4, 5: 4, 6: 4, 7: 4}
flop = {0: 32768, 4: 32768, 5: 32768, 6: 32768, 7: 16384}
for mode in range(8):
    for blocks, threads in ((1, 64), (256, 256), (256, 512), (256, 1024)):
        iters = 20000
        ms = C.c_float()
        rc = lib.probe_run(mode, blocks, threads, iters, out.data_ptr(), stamps.data_ptr(), C.byref(ms))
        torch.cuda.synchronize()
        assert rc == 0, rc
        st = stamps.tolist()
        n = iters * per_iter[mode]
        wps = max(1, threads // 256)
        extra = ""
        if mode in flop:
            extra = f"  {blocks * (threads // 64) * n * flop[mode] / (ms.value * 1e-3) / 1e12:8.1f} TFLOP/s"
        print(f"{names[mode]:22s} {blocks:4d}x{threads:4d} ({wps} wave/SIMD): {ms.value:8.3f} ms  cycles/unit/wave {st[0]/n:6.2f}  "
              f"cycles/unit/SIMD {st[0]/n/wps:6.2f}  clock {st[0]/ms.value/1e3:7.1f} MHz{extra}", flush=True)

"""Slot-cost probe of the query-block attention forward (see probe_slot.hip); builds its .so on the spot.  usage: probe_slot.py"""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "probe_slot.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC", "-o", so, os.path.join(here, "probe_slot.hip")])
lib = C.CDLL(so)
lib.probe_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
out = torch.zeros(4, device="cuda"); stamps = torch.zeros(4, dtype=torch.int64, device="cuda")
names = {256: "+ 5 fragment reads per step, one per other slot, into AGPRs", 512: "+ 5 reads per step into VGPRs", 1280: "+ 5 reads in a burst, AGPRs",
         1536: "+ 5 reads in a burst, VGPRs", 258: "matrix + 5 spread reads (AGPR), no vector work", 514: "matrix + 5 spread reads (VGPR), no vector work",
         0: "step as in the kernel: S and O in VGPRs", 1: "O in AGPRs, S in VGPRs", 9: "O and S in AGPRs (exp on stand-ins)",
         16: "S operands from VGPRs, O VGPR", 17: "S operands from VGPRs, O AGPR",
         2: "matrix only, O VGPR", 3: "matrix only, O AGPR", 4: "vector only (maxima + 16 groups)", 36: "vector only, no maxima",
         68: "exponentials only (32 v_exp)", 196: "16 exponentials only", 32: "no maxima, O VGPR", 33: "no maxima, O AGPR",
         64: "matrix + exponentials only, O VGPR", 65: "matrix + exponentials only, O AGPR", 128: "one exp per slot, O VGPR", 129: "one exp per slot, O AGPR"}
iters = 2000
for mode, nm in names.items():
    for blocks in (1, 256):
        ms = C.c_float()
        rc = lib.probe_run(mode, blocks, iters, out.data_ptr(), stamps.data_ptr(), C.byref(ms))
        torch.cuda.synchronize()
        assert rc == 0, (mode, rc)
        cyc = stamps.tolist()[0] / (2 * iters)
        print(f"mode {mode:3d} {nm:45s} blocks {blocks:3d}: {cyc:7.1f} cycles / step   ({ms.value * 1e3 / (2 * iters):6.3f} us/step -> {cyc / (ms.value * 1e3 / (2 * iters)) :7.1f} MHz)", flush=True)

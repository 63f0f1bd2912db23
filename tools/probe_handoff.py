"""Ordered hand-off probe on the GPU box (see probe_handoff.hip): time per tile of a chain of workgroups that sum 64 x 64 fp32
tiles in a fixed order through L2, against the same loop without the hand-off and against fp32 atomics.  usage: probe_handoff.py"""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/probe_handoff.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-shared", "-fPIC", "-w", "-o", so,
                       os.path.join(here, "probe_handoff.hip")])
lib = C.CDLL(so)
lib.handoff_run.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 7 + [C.POINTER(C.c_float)]
names = {0: "ordered hand-off", 1: "compute alone", 2: "fp32 atomics", 3: "hand-off, L1 fences"}


def run(mode, chains, length, tiles, rot, work, same_xcd, reps=3):
    grid = 8 * ((chains + 7) // 8) * length if same_xcd else chains * length
    mem = torch.zeros(chains * tiles * 4096, device="cuda")
    prog = torch.zeros(chains * length, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    cyc = torch.zeros(2 * grid, dtype=torch.int64, device="cuda")
    best = None
    for _ in range(reps):
        prog.zero_(); mem.zero_(); err.zero_(); cyc.zero_(); torch.cuda.synchronize()
        ms = C.c_float()
        rc = lib.handoff_run(mode, mem.data_ptr(), prog.data_ptr(), err.data_ptr(), cyc.data_ptr(), chains, length, tiles, rot, work, same_xcd, grid, C.byref(ms))
        assert rc == 0, rc
        torch.cuda.synchronize()
        if best is None or ms.value < best[0]:
            c = cyc.view(-1, 2).float()
            live = c[:, 0] > 0
            best = (ms.value, float(c[live, 0].mean()), float(c[live, 1].mean()), int(err), mem.clone())
    ms, total_c, wait_c, errors, m = best
    ok = bool((m == float(length)).all()) if mode in (0, 2, 3) else True
    print(f"{names[mode]:19s} {chains:2d} chains x {length:2d} workgroups, {tiles} tiles, rot {rot}, {'one XCD per chain ' if same_xcd else 'chains across XCDs'}: "
          f"{ms * 1e3:8.1f} us = {ms * 1e3 / tiles:6.2f} us per tile; waiting {100 * wait_c / max(total_c, 1):5.1f} % of a workgroup's cycles; "
          f"errors {errors}; sums {'ok' if ok else 'WRONG'}", flush=True)


for work in (3000, 1000):
    print(f"--- {work} dependent FMAs of stand-in work per tile", flush=True)
    for chains in (8, 12):
        run(1, chains, 20, 40, 0, work, 1)
        run(0, chains, 20, 40, 0, work, 1)
        run(0, chains, 20, 40, 2, work, 1)
        run(0, chains, 20, 40, 2, work, 0)
        run(3, chains, 20, 40, 0, work, 1)
        run(3, chains, 20, 40, 2, work, 1)
        run(3, chains, 20, 40, 2, work, 0)          # (fences too weak for chains across XCDs: expect WRONG sums)
        run(2, chains, 20, 40, 2, work, 1)

"""Atomic-rate probe on the GPU box (see probe_atomics.hip)."""
import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "probe_atomics.so"))
lib.atomics_run.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
rows, ld, passes = 32 * 2538, 512, 5
dq = torch.zeros(rows, ld, device="cuda")
names = {0: "fp32 atomic, lane = column (kernel's pattern)", 1: "plain nontemporal store", 2: "u64 int atomic (2 columns each)",
         3: "pk_bf16 atomic (2 columns each)", 4: "fp32 atomic, lane = 4 columns of a row", 5: "f64 atomic (2 per lane)",
         6: "fp32 atomic, 2 rows x 128 B per instruction", 7: "fp32 atomic, 1 row x 256 B per instruction", 8: "plain store, 2 rows x 128 B",
         9: "u64 int atomic, dense (2 rows x 256 B)", 10: "pk_bf16 atomic, dense (2 rows x 128 B of bf16)"}
elems = {0: 1, 1: 1, 2: 1, 3: 0.5, 4: 1, 5: 1, 6: 1, 7: 1, 8: 1, 9: 1, 10: 1}      # fraction of the (rows x 512) elements touched per pass
for mode in (0, 6, 9, 10):
    for _ in range(2):
        ms = C.c_float()
        rc = lib.atomics_run(mode, dq.data_ptr(), rows, ld, passes, C.byref(ms))
        torch.cuda.synchronize(); assert rc == 0, rc
    n = rows * ld * passes * elems[mode]
    print(f"{names[mode]:48s}: {ms.value * 1e3:8.1f} us  {n / ms.value / 1e6:7.1f} G elements/s  ({n * 4 / ms.value / 1e9:6.2f} TB/s as fp32)", flush=True)
# L2-resident variant: a buffer of one sample (2538 x 512 fp32 = 5.2 MB over 8 XCD L2s), many passes
rows2, passes2 = 2538, 160
dq2 = torch.zeros(rows2, ld, device="cuda")
for mode in (0, 6, 1):
    for _ in range(2):
        ms = C.c_float()
        rc = lib.atomics_run(mode, dq2.data_ptr(), rows2, ld, passes2, C.byref(ms))
        torch.cuda.synchronize(); assert rc == 0, rc
    n = rows2 * ld * passes2
    print(f"[5 MB buffer] {names[mode]:48s}: {ms.value * 1e3:8.1f} us  {n / ms.value / 1e6:7.1f} G elements/s  ({n * 4 / ms.value / 1e9:6.2f} TB/s as fp32)", flush=True)

"""Scale-operand semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (see probe_fp8.hip, probe_fp8_scale): which lane's scale
byte multiplies which (row | column, 32-element K block).  usage: probe_fp8_scale.py"""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/probe_fp8.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(here, "probe_fp8.hip")])
lib = C.CDLL(so)
dev = "cuda"
def run(A, B, sa, sb, sel=0):
    out = torch.zeros(64, 16, device=dev)
    assert lib.probe_fp8_scale(*(C.c_void_p(t.data_ptr()) for t in (A, B, sa, sb, out)), sel) == 0
    o = out.cpu(); D = torch.zeros(32, 32)
    for lane in range(64):
        for r in range(16):
            D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31] = o[lane, r]
    return D
ones = torch.full((64, 8), 0x38383838, dtype=torch.int32, device=dev)          # e4m3 1.0 in every byte
unit = torch.full((64,), 127, dtype=torch.int32, device=dev)
print("unit scales, ones x ones:", run(ones, ones, unit, unit).unique().tolist())
for L in (5, 37):
    sa = unit.clone(); sa[L] = 129          # x4 in byte 0 of lane L only
    D = run(ones, ones, sa, unit)
    ch = (D != 64).nonzero()
    print(f"scale_a x4 at lane {L}: rows changed {sorted(set(ch[:, 0].tolist()))} cols {len(set(ch[:, 1].tolist()))} values {D[D != 64].unique().tolist()} (32*4+32 = 160 expected in one row)")
    sb = unit.clone(); sb[L] = 129
    D = run(ones, ones, unit, sb)
    ch = (D != 64).nonzero()
    print(f"scale_b x4 at lane {L}: cols changed {sorted(set(ch[:, 1].tolist()))} rows {len(set(ch[:, 0].tolist()))} values {D[D != 64].unique().tolist()}")
# which K half does lane L's scale multiply: A ones only in half 0 (lanes < 32 hold K 0..31)
A0 = ones.clone(); A0[32:] = 0
sa = unit.clone(); sa[5] = 129
print("A nonzero only in lanes < 32; scale_a x4 at lane 5 -> row 5 value", run(A0, ones, sa, unit)[5, 0].item(), "(128 if lane 5's scale covers its own K block)")
sa = unit.clone(); sa[37] = 129
print("A nonzero only in lanes < 32; scale_a x4 at lane 37 -> row 5 value", run(A0, ones, sa, unit)[5, 0].item(), "(32 if lane 37's scale covers only its own K block)")
# byte selection
sa = unit.clone(); sa[5] = 127 | (129 << 8)
print("byte 1 = x4, byte 0 = unit, selector 0: row 5", run(ones, ones, sa, unit)[5, 0].item(), " selector 1:", run(ones, ones, sa, unit, 1)[5, 0].item())
sa = torch.full((64,), 127, dtype=torch.int32, device=dev); sa[5] = 129          # upper bytes zero
print("upper bytes zero (scale word = 129), selector 0: row 5", run(ones, ones, sa, unit)[5, 0].item())
# random check of the whole hypothesis
g = torch.Generator().manual_seed(0)
Af = torch.randint(0, 0x7f, (64, 32), generator=g).to(torch.uint8); Bf = torch.randint(0, 0x7f, (64, 32), generator=g).to(torch.uint8)
Af[Af == 0x7f] = 0; Bf[Bf == 0x7f] = 0
sa = torch.randint(120, 135, (64,), generator=g, dtype=torch.int32); sb = torch.randint(120, 135, (64,), generator=g, dtype=torch.int32)
def mat(F, sc):          # lane l = row l & 31; bytes 0..15 -> K 16 (l >> 5) + j (block 0), bytes 16..31 -> K 32 + 16 (l >> 5) + j - 16 (block 1);
    V = F.view(torch.float8_e4m3fn).double()          # block 0's scale from lane (l & 31), block 1's from lane 32 + (l & 31)
    M = torch.cat([V[:32, :16], V[32:, :16], V[:32, 16:], V[32:, 16:]], 1)
    s0, s1 = torch.exp2(sc[:32].double() - 127)[:, None], torch.exp2(sc[32:].double() - 127)[:, None]
    return torch.cat([M[:, :32] * s0, M[:, 32:] * s1], 1)
want = mat(Af, sa) @ mat(Bf, sb).T
got = run(Af.view(torch.int32).to(dev), Bf.view(torch.int32).to(dev), sa.to(dev), sb.to(dev)).double()
print("random operands and scales, hypothesis 'lanes < 32 scale bytes 0..15 of both halves, lanes >= 32 bytes 16..31': max rel err", ((got - want).abs().max() / want.abs().max()).item())

"""Stress of the one-pass attention backward against the two-pass kernels on the CMU structure at b = 32 (and b = 8: key blocks
split four ways): many rounds with fresh random q / k / v / dO and a fresh random padding pattern each (ragged lengths, dropped
modalities - i.e. dead wavefronts, dead key blocks, uniform rows), every round: the one-pass result twice (must be the same bits)
and the two-pass result (must agree to the parity tolerance, 6e-3 relative per tensor).  The kernel's asynchronous pieces - counted
waits, owned accumulator registers, cross-block requests - are exercised under varying block / tile liveness.
usage: stress_onepass.py [seconds per batch size, default 60] [long]      (long: the LONG structure, N = 6088, at b = 32 and b = 16)"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
def rel(a, b): return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
long_seq = len(sys.argv) > 2 and sys.argv[2] == "long"
for b in ((32, 16) if long_seq else (32, 8)):
    cfg = P.config.cmu_model_config(batch_size=b, long_seq=long_seq); cfg["depth"] = 1
    torch.manual_seed(0)
    eng = P.MCA(**cfg).cuda().engine
    ws = eng.workspace(b); N, D = eng.N, eng.D
    a = ws["layers"][0]
    g = torch.Generator(device="cuda").manual_seed(11 + b)
    eng.dbg["onepass"] = True
    form = eng.backward_form(b)
    worst, rounds, t0 = [0.0, 0.0, 0.0], 0, time.time()
    while time.time() - t0 < secs:
        ws["padding"].zero_()
        for mi, n in enumerate(eng.st.token_dims):          # ragged lengths, 25 % of the modalities dropped (never all of a sample's)
            ln = torch.randint(1, n + 1, (b,), device="cuda", generator=g)
            if mi > 0: ln[torch.rand(b, device="cuda", generator=g) < 0.25] = 0
            ws["padding"][:, eng.offsets[mi]:eng.offsets[mi] + n] = (torch.arange(n, device="cuda")[None] >= ln[:, None]).to(ws["padding"].dtype)
        H.call("mca_build_keyinfo", ws["padding"].data_ptr(), eng.kgroup.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr(), b, N, eng.nk_pad, H.stream_ptr())
        H.call("mca_build_keyhot", ws["keyinfo"].data_ptr(), ws["khot"].data_ptr(), b, eng.nk_pad, H.stream_ptr())
        a["qkv"].copy_(torch.randn(a["qkv"].shape, device="cuda", generator=g).bfloat16()); a["qkv"][:, :D] *= 0.18
        ws["do"].copy_((torch.randn(ws["do"].shape, device="cuda", generator=g) * 0.1).bfloat16())
        eng._attn_fwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N)
        outs = []
        for mode in (True, True, False):
            eng.dbg["onepass"] = mode
            a["dqkv"].fill_(7.0)
            eng._attn_bwd2(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], ws["do"], a["lse"], ws["delta"], a["dqkv"].data_ptr(), N*3*D, 3*D, False,
                           a["dqkv"], D, 2*D, 3*D, eng.qmask_attn, eng.sched_attn_f, eng.sched_attn_b2, ws, b, N)
            outs.append(a["dqkv"].clone())
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), f"b={b} round {rounds}: the one-pass backward is not repeatable"
        assert not torch.isnan(outs[0].float()).any(), f"b={b} round {rounds}: NaN"
        for i, nm in enumerate(("dq", "dk", "dv")):
            x, y = outs[0].view(b * N, 3, D)[:, i], outs[2].view(b * N, 3, D)[:, i]
            e = rel(x, y); worst[i] = max(worst[i], e)
            assert e < 6e-3, f"b={b} round {rounds}: {nm} differs from the two-pass kernels by {e}"
        rounds += 1
    print(f"b = {b:2d} ({form}): {rounds} rounds in {time.time() - t0:.0f} s, every one repeatable bit for bit; worst relative distance to the two-pass kernels "
          f"dq {worst[0]:.2e} dk {worst[1]:.2e} dv {worst[2]:.2e}", flush=True)
    eng.dbg["onepass"] = None

"""Region-by-region comparison of the pipelined one-pass backward against the plain form of the same algorithm (knob 9 bit 64)."""
import importlib, os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
H = importlib.import_module("mca-paper_amd.hip"); S = importlib.import_module("mca-paper_amd.structure"); E = importlib.import_module("mca-paper_amd.engine")
import test_attention_gpu as T

shape = sys.argv[1] if len(sys.argv) > 1 else "small"
if shape == "small":
    st, b, heads = S.FusionStructure([70, 45, 30], 8, (3, 2), fcl=True), 3, 2
else:
    st, b, heads = S.FusionStructure([1500, 450, 450, 50], 88, (4, 3, 2), fcl=True), 2, 2
N, D, dev = st.n_tokens, heads * 64, "cuda"
g = torch.Generator(device=dev).manual_seed(int(os.environ.get("DBG_SEED", "11")))
pad = torch.zeros(b, N, dtype=torch.bool, device=dev)
if os.environ.get("DBG_PAD"):          # the test's padding: a random valid prefix per modality
    off_ = 0
    for mi, n_ in enumerate(st.token_dims):
        ln_ = torch.randint(1, n_ + 1, (b,), generator=g, device=dev)
        if os.environ.get("DBG_DROP") and mi == 0:
            ln_[0] = 0
        pad[:, off_:off_ + n_] = torch.arange(n_, device=dev)[None] >= ln_[:, None]
        off_ += n_
    offs_ = np.concatenate([[0], np.cumsum(st.token_dims)]).tolist()
    print("valid lengths per modality:", [(~pad[:, o_:o_ + n_]).sum(1).tolist() for o_, n_ in zip(offs_, st.token_dims)])
qkv = T.bf(torch.randn(b, N, 3 * D, device=dev, generator=g))
qkv[:, :, :D] = T.bf(qkv[:, :, :D].float() * T.C2)
nk_pad = (N + 255) // 256 * 256
kgroup = torch.from_numpy(st.kgroup).to(dev)
keyinfo = torch.empty(b, nk_pad, dtype=torch.uint8, device=dev); kflags = torch.empty(b, (N + 63) // 64, dtype=torch.uint8, device=dev)
H.call("mca_build_keyinfo", pad.to(torch.uint8).data_ptr(), kgroup.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr(), b, N, nk_pad, H.stream_ptr())
khot = torch.empty(b, nk_pad, 16, dtype=torch.bfloat16, device=dev)
H.call("mca_build_keyhot", keyinfo.data_ptr(), khot.data_ptr(), b, nk_pad, H.stream_ptr())
bits = (torch.from_numpy(st.qmask_attn.astype(np.int64)).to(dev)[:, None] >> torch.arange(16, device=dev)[None, :]) & 1
bits[:, 15] = 0
qblk = torch.where(bits == 1, 0.0, -32768.0).to(torch.bfloat16).contiguous()
sf = E._Sched(st.attn_schedule(128, 64), dev)
qmask = torch.from_numpy(st.qmask_attn.astype(np.uint32).view(np.int32)).to(dev)
vmean = torch.empty(b, D, device=dev)
H.call("mca_attn_vmean", qkv.data_ptr() + 2 * D * 2, N * 3 * D, 3 * D, vmean.data_ptr(), b, N, heads, H.stream_ptr())
o = torch.zeros(b * N, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(b, heads, N, device=dev)
a = H.AttnFwdArgs()
a.q, a.q_bstride, a.q_ld = qkv.data_ptr(), N * 3 * D, 3 * D
a.k, a.v, a.kv_bstride, a.kv_ld = qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2, N * 3 * D, 3 * D
a.o, a.o_bstride, a.o_ld, a.lse = o.data_ptr(), N * D, D, lse.data_ptr()
a.qmask, a.keyinfo, a.ktile_flags = qmask.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr()
a.q_ptr, a.q_kt, a.q_order = sf.q_ptr.data_ptr(), sf.q_kt.data_ptr(), sf.q_order.data_ptr()
a.vmean = vmean.data_ptr()
a.batch, a.heads, a.nq, a.nk, a.nk_pad, a.n_qtiles, a.n_ktiles, a.scale = b, heads, N, N, nk_pad, sf.s.n_q, sf.s.n_k, 0.125
a.flags, a.khot = H.ATTN_Q_PRESCALED, khot.data_ptr()
H.call("mca_attn_fwd", C.byref(a), H.stream_ptr())
d_o = T.bf(torch.randn(b, N, D, device=dev, generator=g))
dvmean = torch.zeros(b, D, device=dev)
sc = S.build_onepass_schedule(st.qmask_attn, st.kgroup, 64, 256, True)
print("q tiles", sc.qt_desc.tolist()); print("key blocks", sc.kb_desc.tolist())
with H.knobs(k9=64):
    ref = T._run_onepass(H, sc, b, heads, N, nk_pad, qkv, o, d_o, lse, dvmean, keyinfo, kflags, khot, qblk)[0]
got = T._run_onepass(H, sc, b, heads, N, nk_pad, qkv, o, d_o, lse, dvmean, keyinfo, kflags, khot, qblk)[0]
rel = lambda x, y: float((x.float() - y.float()).norm() / (y.float().norm() + 1e-30))
print("dq", rel(got[0], ref[0]), "dk", rel(got[1][:, :, D:2 * D], ref[1][:, :, D:2 * D]), "dv", rel(got[1][:, :, 2 * D:], ref[1][:, :, 2 * D:]))
for hh in range(heads):
    sl = slice(hh * 64, hh * 64 + 64)
    print(f"head {hh}:")
    for t, (r0, rn) in enumerate(sc.qt_desc.tolist()):
        print(f"  dq tile {t} rows {r0}+{rn}: " + " ".join(f"{rel(got[0][s_, r0:r0 + rn, sl], ref[0][s_, r0:r0 + rn, sl]):.3f}" for s_ in range(b))
              + "   halves d0-31/d32-63 rows0-31/32-63: " + " ".join(f"{rel(got[0][0, r0 + ra:r0 + min(rn, rb), hh * 64 + da:hh * 64 + db], ref[0][0, r0 + ra:r0 + min(rn, rb), hh * 64 + da:hh * 64 + db]):.3f}" for (ra, rb) in ((0, 32), (32, 64)) for (da, db) in ((0, 32), (32, 64)) if ra < rn))
    for kb, (k0, kn, _, _) in enumerate(sc.kb_desc.tolist()):
        for w in range(0, kn, 32):
            e = min(kn, w + 32)
            print(f"  key block {kb} keys {k0 + w}..{k0 + e - 1}: dk " + " ".join(f"{rel(got[1][s_, k0 + w:k0 + e, D + hh * 64:D + hh * 64 + 64], ref[1][s_, k0 + w:k0 + e, D + hh * 64:D + hh * 64 + 64]):.3f}" for s_ in range(b))
                  + "  dv " + " ".join(f"{rel(got[1][s_, k0 + w:k0 + e, 2 * D + hh * 64:2 * D + hh * 64 + 64], ref[1][s_, k0 + w:k0 + e, 2 * D + hh * 64:2 * D + hh * 64 + 64]):.3f}" for s_ in range(b)))
nan = torch.isnan(got[1].float())
print("NaN count total", int(nan.sum()), "per sample", [int(nan[i].sum()) for i in range(b)])
idx = nan.nonzero()
if len(idx):
    print("keys with NaN:", sorted(set(idx[:, 1].tolist()))[:40], "columns:", sorted(set((idx[:, 2] // 64).tolist())), "d within head:", sorted(set((idx[:, 2] % 64).tolist()))[:70])
    # finite part of the affected rows vs reference
    print("inf count", int(torch.isinf(got[1].float()).sum()))
nq_ = torch.isnan(got[0].float())
print("dq NaN count", int(nq_.sum()))
for t, (r0, rn) in enumerate(sc.qt_desc.tolist()):
    for s_ in range(b):
        for hh in range(heads):
            blk_ = nq_[s_, r0:r0 + rn, hh * 64:hh * 64 + 64]
            if blk_.any():
                print(f"  sample {s_} head {hh} tile {t}: NaN rows {sorted(set(blk_.nonzero()[:, 0].tolist()))[:40]} cols {sorted(set(blk_.nonzero()[:, 1].tolist()))[:70]}")
print("---- repeat 6x each kernel, NaN counts (dq, dkv):")
for rep in range(6):
    with H.knobs(k9=64):
        r_ = T._run_onepass(H, sc, b, heads, N, nk_pad, qkv, o, d_o, lse, dvmean, keyinfo, kflags, khot, qblk)
    g_ = T._run_onepass(H, sc, b, heads, N, nk_pad, qkv, o, d_o, lse, dvmean, keyinfo, kflags, khot, qblk)
    cnt = lambda t: int(torch.isnan(t.float()).sum())
    print("plain", [(cnt(x[0]), cnt(x[1])) for x in r_], "pipelined", [(cnt(x[0]), cnt(x[1])) for x in g_])
    for name, outs in (("plain", r_), ("pipelined", g_)):
        for x in outs:
            n_ = torch.isnan(x[1].float())
            if n_.any():
                i_ = n_.nonzero()
                print("   ", name, "NaN samples", sorted(set(i_[:, 0].tolist())), "keys", sorted(set(i_[:, 1].tolist()))[:12], "..", "col blocks", sorted(set((i_[:, 2] // 64).tolist())))
print("---- NaN positions (flat element offsets) over 12 launches of the pipelined kernel")
for rep in range(12):
    g_ = T._run_onepass(H, sc, b, heads, N, nk_pad, qkv, o, d_o, lse, dvmean, keyinfo, kflags, khot, qblk)
    for li, x in enumerate(g_):
        for nm, t in (("dq", x[0]), ("dkv", x[1])):
            f = t.view(-1).view(torch.int16)
            n_ = torch.isnan(t.float()).view(-1)
            if n_.any():
                pos = n_.nonzero().view(-1).tolist()
                print(rep, li, nm, "ptr", hex(t.data_ptr()), "n", len(pos), "offsets", pos[:16], "bits", [hex(int(f[p]) & 0xffff) for p in pos[:8]], "row len", t.shape[-1])

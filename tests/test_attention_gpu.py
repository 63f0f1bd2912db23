"""GPU parity tests of the attention kernels (forward forms, two-pass and one-pass backward, fp8 forms, prep / vmean / key tables)
through the C ABI: against dense fp64 torch formulas, bitwise repeatability, edge cases (dropped modalities, spikes, group count)."""
import ctypes as C
import importlib
import math
import numpy as np
import pytest
import torch



pytestmark = pytest.mark.gpu
C2 = 0.125 * 1.4426950408889634          # scale * log2(e): what the engine folds into the forward copy of W_q


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    hip = importlib.import_module("mca-paper_amd.hip")
    hip.lib()
    return hip


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------- attention
def dense_attention(q, k, v, allowed, pad, scale):
    """model.py:87-99 on (b,h,n,d) tensors; allowed (nq,nk) bool, pad (b,nk) bool."""
    sim = torch.einsum("bhid,bhjd->bhij", q * scale, k)
    neg = -torch.finfo(sim.dtype).max
    sim = sim.masked_fill(~allowed[None, None], neg)
    sim = sim.masked_fill(pad[:, None, None, :], neg)
    return torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v)


def _attention_case(H, st, b, heads, pool, seed, drop_first, prescaled=True, spike=False):
    """prescaled: the q operand in memory is q' = bf16(q * scale * log2 e) (MCA_ATTN_Q_PRESCALED, the production form); the
    dense reference sees q = q' / (scale * log2 e), and dq is the gradient w.r.t. that q.  spike: a few keys 40x larger, so
    that the lazy softmax reference of the forward kernel has to move mid-row (at a row's later tiles, up and from a very
    negative start) - cdna guide rule 26: a rare data-dependent branch needs an input that forces it."""
    eng = importlib.import_module("mca-paper_amd.engine")
    N, D = st.n_tokens, heads * 64
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(seed)
    qmask_np = st.qmask_pool if pool else st.qmask_attn
    nq = len(qmask_np)
    sf = eng._Sched(st.pool_schedule(128, 64) if pool else st.attn_schedule(128, 64), dev)
    sb = eng._Sched(st.pool_schedule(64, 256) if pool else st.attn_schedule(64, 256), dev)
    sb128 = eng._Sched(st.pool_schedule(64, 128) if pool else st.attn_schedule(64, 128), dev)          # the dkv pass's 4-wavefront form
    qmask = torch.from_numpy(qmask_np.astype(np.uint32).view(np.int32)).to(dev)
    kgroup = torch.from_numpy(st.kgroup).to(dev)
    allowed = torch.from_numpy(~(st.dense_pool_mask() if pool else st.dense_attn_mask())).to(dev)
    # padding: random valid prefix per modality; sample 0 optionally loses its first modality entirely
    pad = torch.zeros(b, N, dtype=torch.bool, device=dev)
    off = 0
    for mi, n in enumerate(st.token_dims):
        ln = torch.randint(1, n + 1, (b,), generator=g, device=dev)
        if drop_first and mi == 0:
            ln[0] = 0
        pad[:, off:off + n] = torch.arange(n, device=dev)[None] >= ln[:, None]
        off += n
    qkv = torch.randn(b, N, 3 * D, device=dev, generator=g)
    if spike:
        for j in (3, 70, 131, N - 2):
            qkv[:, j, D:2 * D] *= 40.0
    qkv = bf(qkv)
    qdiv = C2 if prescaled else 1.0          # what the stored q operand has to be divided by to get the reference's q
    if pool:
        qsrc = bf(torch.randn(nq, D, device=dev, generator=g) * qdiv)
        q4 = (qsrc.float() / qdiv).view(1, nq, heads, 64).permute(0, 2, 1, 3).expand(b, -1, -1, -1)
    else:
        qkv[:, :, :D] = bf(qkv[:, :, :D].float() * qdiv)
        q4 = (qkv[:, :, :D].float() / qdiv).view(b, N, heads, 64).permute(0, 2, 1, 3)
    k4 = qkv[:, :, D:2 * D].float().view(b, N, heads, 64).permute(0, 2, 1, 3)
    v4 = qkv[:, :, 2 * D:].float().view(b, N, heads, 64).permute(0, 2, 1, 3)
    q4r, k4r, v4r = (t.double().clone().requires_grad_(True) for t in (q4, k4, v4))
    ref = dense_attention(q4r, k4r, v4r, allowed, pad, 0.125)          # (b,h,nq,64)
    ref_o = ref.permute(0, 2, 1, 3).reshape(b, nq, D)

    nk_pad = (N + 255) // 256 * 256
    keyinfo = torch.empty(b, nk_pad, dtype=torch.uint8, device=dev)
    kflags = torch.empty(b, (N + 63) // 64, dtype=torch.uint8, device=dev)
    H.call("mca_build_keyinfo", pad.to(torch.uint8).data_ptr(), kgroup.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr(), b, N, nk_pad,
           H.stream_ptr())
    exp_info = torch.where(pad, torch.full_like(pad, 31, dtype=torch.uint8), kgroup[None].expand(b, -1))
    assert torch.equal(keyinfo[:, :N], exp_info) and (keyinfo[:, N:] == 31).all()
    vmean = torch.empty(b, D, device=dev)
    vptr = qkv.data_ptr() + 2 * D * 2
    H.call("mca_attn_vmean", vptr, N * 3 * D, 3 * D, vmean.data_ptr(), b, N, heads, H.stream_ptr())
    assert rel(vmean, qkv[:, :, 2 * D:].float().mean(1)) < 1e-5
    o = torch.zeros(b * nq, D, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(b, heads, nq, device=dev)
    a = H.AttnFwdArgs()
    if pool:
        a.q, a.q_bstride, a.q_ld = qsrc.data_ptr(), 0, D
    else:
        a.q, a.q_bstride, a.q_ld = qkv.data_ptr(), N * 3 * D, 3 * D
    a.k, a.v, a.kv_bstride, a.kv_ld = qkv.data_ptr() + D * 2, vptr, N * 3 * D, 3 * D
    a.o, a.o_bstride, a.o_ld, a.lse = o.data_ptr(), nq * D, D, lse.data_ptr()
    a.qmask, a.keyinfo, a.ktile_flags = qmask.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr()
    a.q_ptr, a.q_kt, a.q_order = sf.q_ptr.data_ptr(), sf.q_kt.data_ptr(), sf.q_order.data_ptr()
    a.vmean = vmean.data_ptr()
    a.batch, a.heads, a.nq, a.nk, a.nk_pad, a.n_qtiles, a.n_ktiles, a.scale = b, heads, nq, N, nk_pad, sf.s.n_q, sf.s.n_k, 0.125
    a.flags = H.ATTN_Q_PRESCALED if prescaled else 0
    H.call("mca_attn_fwd", C.byref(a), H.stream_ptr())
    torch.cuda.synchronize()
    got_o = o.float().view(b, nq, D)
    e = rel(got_o, ref_o)
    assert e < 6e-3, f"attention forward rel err {e}"
    # ... row by row as well: a form that is wrong on a few rows only (a rare softmax branch) hides in the global norm
    row_err = (got_o.double() - ref_o.detach()).norm(dim=-1) / (ref_o.detach().norm(dim=-1) + 1e-9)
    assert float(row_err.max()) < 3e-2, f"worst row of the attention forward: rel err {float(row_err.max())} (median {float(row_err.median())})"
    # uniform rows are flagged
    uni_ref = ((~allowed)[None] | pad[:, None, :]).all(-1)             # (b, nq)
    assert torch.equal(torch.isinf(lse[:, 0]), uni_ref)
    # ... and the log-sum-exp of every other row (log2 domain) is the dense one: the backward scales P by 2^-lse, an lse off
    # by 0.05 is a 3.5 % error on that row's gradients with a perfectly good forward output
    with torch.no_grad():
        sim = torch.einsum("bhid,bhjd->bhij", q4.double() * 0.125, k4.double()) * 1.4426950408889634
        sim = sim.masked_fill((~allowed)[None, None] | pad[:, None, None, :], float("-inf"))
        lse_ref = torch.logsumexp(sim * 0.6931471805599453, -1) * 1.4426950408889634          # (b, h, nq)
    fin_rows = ~uni_ref[:, None, :].expand(-1, heads, -1)
    lse_err = (lse[fin_rows].double() - lse_ref[fin_rows]).abs().max()
    assert lse_err < 1e-3, f"log-sum-exp off by {float(lse_err)} (log2 units)"
    if drop_first and not pool:
        assert uni_ref.any()
    # ---- the mask as a matrix product (mca_build_keyhot; structures with at most 15 key groups): same output, same flags
    if int(st.kgroup.max()) <= 14:
        khot = torch.empty(b, nk_pad, 16, dtype=torch.bfloat16, device=dev)
        H.call("mca_build_keyhot", keyinfo.data_ptr(), khot.data_ptr(), b, nk_pad, H.stream_ptr())
        want_hot = torch.nn.functional.one_hot(keyinfo.long().clamp_max(15), 16).to(torch.bfloat16)
        assert torch.equal(khot, want_hot)
        o2 = torch.zeros_like(o); lse2 = torch.empty_like(lse)
        a.o, a.lse, a.khot = o2.data_ptr(), lse2.data_ptr(), khot.data_ptr()
        H.call("mca_attn_fwd", C.byref(a), H.stream_ptr())          # the LDS-DMA kernel
        torch.cuda.synchronize()
        assert rel(o2.float().view(b, nq, D), ref_o) < 6e-3
        assert torch.equal(torch.isinf(lse2[:, 0]), uni_ref)
        fin = ~torch.isinf(lse)
        assert (lse2[fin] - lse[fin]).abs().max() < 1e-4          # adding an exact 0 / an exp2 that underflows to exactly 0
        assert rel(o2.float(), o.float()) < 2e-3
        # ---- MCA_ATTN_LAZY_REFERENCE (the engine's default): that kernel with a LAZY softmax reference (-m as the MFMA C operand,
        # moved by a rare slow path - test_attention_spiked_keys drives it through that path with data).  Same contract against the
        # dense fp64 reference; bitwise repeatable
        if prescaled:
            a.flags = H.ATTN_Q_PRESCALED | H.ATTN_LAZY_REFERENCE
            outs = []
            for _ in range(2):
                o3 = torch.zeros_like(o); lse3 = torch.empty_like(lse)
                a.o, a.lse = o3.data_ptr(), lse3.data_ptr()
                H.call("mca_attn_fwd", C.byref(a), H.stream_ptr())
                torch.cuda.synchronize()
                got3 = o3.float().view(b, nq, D)
                assert rel(got3, ref_o) < 6e-3, f"lazy-reference forward rel err {rel(got3, ref_o)}"
                row3 = (got3.double() - ref_o.detach()).norm(dim=-1) / (ref_o.detach().norm(dim=-1) + 1e-9)
                assert float(row3.max()) < 3e-2, f"worst row of the lazy-reference forward: {float(row3.max())}"
                assert torch.equal(torch.isinf(lse3[:, 0]), uni_ref)
                assert (lse3[fin_rows].double() - lse_ref[fin_rows]).abs().max() < 1e-3
                outs.append((o3, lse3))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])          # bitwise repeatable
            # (P is rounded to bf16 against different references: two independent roundings, 2.2-2.6e-3 at the CMU / LONG shapes)
            assert rel(outs[0][0].float(), o2.float()) < 4e-3
            assert not torch.equal(outs[0][0], o2)          # (the flag really selected another kernel)
            a.flags = H.ATTN_Q_PRESCALED
        a.o, a.lse = o.data_ptr(), lse.data_ptr()

    # ---- backward
    d_o = bf(torch.randn(b, nq, D, device=dev, generator=g))
    ref_o.backward(d_o.double())
    delta = torch.empty(b, heads, nq, device=dev)
    dvmean = torch.empty(b, D, device=dev)
    H.call("mca_attn_bwd_prep", o.data_ptr(), d_o.data_ptr(), nq * D, D, lse.data_ptr(), delta.data_ptr(), dvmean.data_ptr(), b, heads,
           nq, N, H.stream_ptr())
    ref_delta = (d_o.float() * got_o).view(b, nq, heads, 64).sum(-1).permute(0, 2, 1)
    assert rel(delta, ref_delta) < 1e-4
    rdq = q4r.grad.permute(0, 2, 1, 3).reshape(b, nq, D)
    rdk = k4r.grad.permute(0, 2, 1, 3).reshape(b, N, D)
    rdv = v4r.grad.permute(0, 2, 1, 3).reshape(b, N, D)

    # ---- backward in two passes without atomics: dq written once (bf16 and fp32 forms)
    for dq_f32 in (False, True):
        dq2 = torch.full((b, nq, D), 7.0, device=dev, dtype=torch.float32 if dq_f32 else torch.bfloat16)          # no pre-zeroing needed
        dkv2 = torch.zeros(b, N, 3 * D, dtype=torch.bfloat16, device=dev)
        a2 = H.AttnBwd2Args()
        a2.q, a2.q_bstride, a2.q_ld = a.q, a.q_bstride, a.q_ld
        a2.k, a2.v, a2.kv_bstride, a2.kv_ld = a.k, a.v, a.kv_bstride, a.kv_ld
        a2.d_o, a2.o_bstride, a2.o_ld = d_o.data_ptr(), nq * D, D
        a2.lse, a2.delta, a2.dvmean = lse.data_ptr(), delta.data_ptr(), dvmean.data_ptr()
        a2.dq, a2.dq_bstride, a2.dq_ld, a2.dq_f32 = dq2.data_ptr(), nq * D, D, int(dq_f32)
        a2.dk, a2.dv, a2.dkv_bstride, a2.dkv_ld = dkv2.data_ptr() + D * 2, dkv2.data_ptr() + 2 * D * 2, N * 3 * D, 3 * D
        a2.qmask, a2.keyinfo, a2.ktile_flags = qmask.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr()
        a2.q_ptr, a2.q_kt, a2.q_order, a2.n_qtiles128, a2.n_ktiles64 = sf.q_ptr.data_ptr(), sf.q_kt.data_ptr(), sf.q_order.data_ptr(), sf.s.n_q, sf.s.n_k
        a2.k_wg, a2.k_qt, a2.n_qtiles64, a2.n_kblocks256 = sb.k_wg.data_ptr(), sb.k_qt.data_ptr(), sb.s.n_q, sb.s.n_k
        a2.batch, a2.heads, a2.nq, a2.nk, a2.nk_pad, a2.scale, a2.flags = b, heads, nq, N, nk_pad, 0.125, a.flags
        H.call("mca_attn_bwd_dq", C.byref(a2), H.stream_ptr())
        H.call("mca_attn_bwd_dkv", C.byref(a2), H.stream_ptr())
        torch.cuda.synchronize()
        e_q, e_k, e_v = rel(dq2.float(), rdq), rel(dkv2[:, :, D:2 * D].float(), rdk), rel(dkv2[:, :, 2 * D:].float(), rdv)
        assert e_q < 1.5e-2 and e_k < 1.5e-2 and e_v < 1.5e-2, f"two-pass backward rel err dq {e_q} dk {e_k} dv {e_v} (dq_f32={dq_f32})"
        assert (dkv2[:, :, :D] == 0).all()
        # bitwise reproducible: a second launch gives the same bits
        dq3 = torch.empty_like(dq2); dkv3 = torch.zeros_like(dkv2)
        a2.dq, a2.dk, a2.dv = dq3.data_ptr(), dkv3.data_ptr() + D * 2, dkv3.data_ptr() + 2 * D * 2
        H.call("mca_attn_bwd_dq", C.byref(a2), H.stream_ptr())
        H.call("mca_attn_bwd_dkv", C.byref(a2), H.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(dq2, dq3) and torch.equal(dkv2, dkv3)
        # 128-key blocks (4 wavefronts per workgroup): every key sees the same query steps in the same order: the same bits
        dkv5 = torch.zeros_like(dkv2)
        a2.dk, a2.dv = dkv5.data_ptr() + D * 2, dkv5.data_ptr() + 2 * D * 2
        a2.k_wg, a2.k_qt, a2.n_kblocks256, a2.kblock_keys = sb128.k_wg.data_ptr(), sb128.k_qt.data_ptr(), sb128.s.n_k, 128
        H.call("mca_attn_bwd_dkv", C.byref(a2), H.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(dkv2, dkv5)
        a2.k_wg, a2.k_qt, a2.n_kblocks256, a2.kblock_keys = sb.k_wg.data_ptr(), sb.k_qt.data_ptr(), sb.s.n_k, 0
        # the mask as a matrix product (khot + qblk): allowed scores get an exact +0, blocked ones an exp2 that is exactly 0:
        # the same bits as the element-wise mask
        if int(st.kgroup.max()) <= 14:
            bits = (torch.from_numpy(qmask_np.astype(np.int64)).to(dev)[:, None] >> torch.arange(16, device=dev)[None, :]) & 1
            bits[:, 15] = 0
            qblk = torch.where(bits == 1, 0.0, -32768.0).to(torch.bfloat16).contiguous()
            dq4 = torch.empty_like(dq2); dkv4 = torch.zeros_like(dkv2)
            a2.dq, a2.dk, a2.dv = dq4.data_ptr(), dkv4.data_ptr() + D * 2, dkv4.data_ptr() + 2 * D * 2
            a2.khot, a2.qblk = khot.data_ptr(), qblk.data_ptr()
            H.call("mca_attn_bwd_dq", C.byref(a2), H.stream_ptr())
            H.call("mca_attn_bwd_dkv", C.byref(a2), H.stream_ptr())
            torch.cuda.synchronize()
            assert torch.equal(dq2, dq4) and torch.equal(dkv2, dkv4)

    # ---- backward in ONE pass (attention_bwd1.hip): five products, dQ summed over the key blocks by a private read-modify-write.
    # Against the dense fp64 gradients (the two-pass bound), against the two-pass results, bitwise repeatable, and the same on the
    # structure-aligned tables and on the plain 64 x 256 grid
    if not pool and prescaled and int(st.kgroup.max()) <= 14:
        S_ = importlib.import_module("mca-paper_amd.structure")
        ref2 = (dq2.float(), dkv2[:, :, D:2 * D].float(), dkv2[:, :, 2 * D:].float())          # (the fp32-dq form of the loop's last turn)
        outs = []
        for aligned in (True, False):
            sc_ = S_.build_onepass_schedule(qmask_np, st.kgroup, 64, 256, aligned)
            too_big = len(sc_.qt_desc) >= 256 or len(sc_.kb_desc) > 64 or int(sc_.kb_desc[:, 3].max()) + 6 > 256 or len(sc_.kb_qt) + 4 * len(sc_.kb_desc) > 768
            try:
                got = _run_onepass(H, sc_, b, heads, N, nk_pad, qkv, o, d_o, lse, dvmean, keyinfo, kflags, khot, qblk)
            except H.MCAHipError as exc:          # tables past the kernel's LDS budget: refused, the caller keeps the two-pass form
                assert too_big and "unsupported" in str(exc), str(exc)
                continue
            assert not too_big
            for rep in got:
                if torch.isnan(rep[0].float()).any() or torch.isnan(rep[1].float()).any():          # where: tile, wavefront block, sample
                    nq_, msg = torch.isnan(rep[0].float()), []
                    for t, (r0, rn) in enumerate(sc_.qt_desc.tolist()):
                        for s_ in range(b):
                            for hh in range(heads):
                                blk_ = nq_[s_, r0:r0 + rn, hh * 64:hh * 64 + 64]
                                if blk_.any():
                                    ii = blk_.nonzero()
                                    msg.append(f"sample {s_} head {hh} tile {t} rows {sorted(set(ii[:, 0].tolist()))[:6]}.. cols {sorted(set(ii[:, 1].tolist()))[:6]}..")
                    raise AssertionError(f"NaN in the one-pass backward (aligned={aligned}): dq {int(nq_.sum())}, dkv {int(torch.isnan(rep[1].float()).sum())}; " + "; ".join(msg[:12]))
                e_q, e_k, e_v = rel(rep[0].float(), rdq), rel(rep[1][:, :, D:2 * D].float(), rdk), rel(rep[1][:, :, 2 * D:].float(), rdv)
                assert e_q < 1.5e-2 and e_k < 1.5e-2 and e_v < 1.5e-2, f"one-pass backward rel err dq {e_q} dk {e_k} dv {e_v} (aligned={aligned})"
                assert (rep[1][:, :, :D] == 0).all()
            assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])          # a second launch: the same bits
            d_q, d_k, d_v = rel(got[0][0].float(), ref2[0]), rel(got[0][1][:, :, D:2 * D].float(), ref2[1]), rel(got[0][1][:, :, 2 * D:].float(), ref2[2])
            assert d_q < 6e-3 and d_k < 6e-3 and d_v < 6e-3, f"one-pass vs two-pass: dq {d_q} dk {d_k} dv {d_v} (aligned={aligned})"
            outs.append(got[0])
        if len(outs) == 2:
            assert rel(outs[0][0].float(), outs[1][0].float()) < 6e-3


def _run_onepass(H, sc, b, heads, N, nk_pad, qkv, o, d_o, lse, dvmean_ref, keyinfo, kflags, khot, qblk):
    """mca_attn_bwd_prep_onepass + mca_attn_bwd_onepass on the tables `sc`, twice; returns [(dq, dkv), (dq, dkv)]"""
    dev, D = "cuda", heads * 64
    i32 = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32) if x.dtype == np.uint32 else np.ascontiguousarray(x)).to(dev)
    qt_desc, kb_desc, kb_qt, visit, row_slot = i32(sc.qt_desc), i32(sc.kb_desc), i32(sc.kb_qt), i32(sc.visit), i32(sc.row_slot)
    nqt, nkb = len(sc.qt_desc), len(sc.kb_desc)
    rowc = torch.empty(b, heads, nqt + 1, 2, 64, device=dev)          # (+ the null tile)
    rowc[:, :, :, 0] = float("-inf"); rowc[:, :, :, 1] = 0.0
    dvmean = torch.full_like(dvmean_ref, 3.0)
    # (+ 64 rows of slack: the kernel reads whole 64-row tiles of the packed copies, the last one past its rows)
    q_buf = torch.zeros((b * heads * N + 64) * 64, dtype=torch.bfloat16, device=dev); do_buf = torch.zeros_like(q_buf)
    q_hm, do_hm = q_buf[:b * heads * N * 64].view(b, heads, N, 64), do_buf[:b * heads * N * 64].view(b, heads, N, 64)
    H.call("mca_attn_bwd_prep_onepass", o.data_ptr(), d_o.data_ptr(), N * D, D, lse.data_ptr(), row_slot.data_ptr(), rowc.data_ptr(),
           dvmean.data_ptr(), b, heads, N, nqt, qkv.data_ptr(), N * 3 * D, 3 * D, q_hm.data_ptr(), do_hm.data_ptr(), H.stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(dvmean, dvmean_ref)
    # the head-major packed copies, bit for bit
    assert torch.equal(q_hm, qkv[:, :, :D].view(b, N, heads, 64).permute(0, 2, 1, 3)) and torch.equal(do_hm, d_o.view(b, N, heads, 64).permute(0, 2, 1, 3))
    # the row constants, tile by tile
    delta_ref = (d_o.float() * o.float().view(b, N, D)).view(b, N, heads, 64).sum(-1).permute(0, 2, 1)          # (b, h, N)
    for t, (r0, rn) in enumerate(sc.qt_desc.tolist()):
        assert torch.equal(rowc[:, :, t, 0, :rn], -lse[:, :, r0:r0 + rn])
        assert (rowc[:, :, t, 1, :rn] + delta_ref[:, :, r0:r0 + rn]).abs().max() < 1e-3 * (1 + float(delta_ref.abs().max()))
        assert torch.isinf(rowc[:, :, t, 0, rn:]).all() and (rowc[:, :, t, 1, rn:] == 0).all()
    assert torch.isinf(rowc[:, :, nqt, 0]).all() and (rowc[:, :, nqt, 1] == 0).all()
    acc = torch.full((b * heads * (nqt + 1) * 4096,), float("nan"), device=dev)          # contents irrelevant on entry
    out = []
    for rep_i in range(2):          # first launch: q / dO from the (b, n, heads*64) matrices; second: from the packed copies (same bits)
        dq = torch.full((b, N, D), 7.0, device=dev, dtype=torch.bfloat16)
        dkv = torch.zeros(b, N, 3 * D, dtype=torch.bfloat16, device=dev)
        a1 = H.AttnBwd1Args()
        a1.k, a1.v, a1.kv_bstride, a1.kv_ld = qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2, N * 3 * D, 3 * D
        if rep_i == 0:
            a1.q, a1.q_bstride, a1.q_ld = qkv.data_ptr(), N * 3 * D, 3 * D
            a1.d_o, a1.o_bstride, a1.o_ld = d_o.data_ptr(), N * D, D
        else:
            a1.q, a1.q_bstride, a1.q_hstride, a1.q_ld = q_hm.data_ptr(), heads * N * 64, N * 64, 64
            a1.d_o, a1.o_bstride, a1.o_hstride, a1.o_ld = do_hm.data_ptr(), heads * N * 64, N * 64, 64
        a1.rowc, a1.dvmean = rowc.data_ptr(), dvmean.data_ptr()
        a1.dq, a1.dq_bstride, a1.dq_ld = dq.data_ptr(), N * D, D
        a1.dk, a1.dv, a1.dkv_bstride, a1.dkv_ld = dkv.data_ptr() + D * 2, dkv.data_ptr() + 2 * D * 2, N * 3 * D, 3 * D
        a1.dq_acc = acc.data_ptr()
        a1.keyinfo, a1.ktile_flags, a1.khot, a1.qblk = keyinfo.data_ptr(), kflags.data_ptr(), khot.data_ptr(), qblk.data_ptr()
        a1.qt_desc, a1.kb_desc, a1.kb_qt, a1.visit = qt_desc.data_ptr(), kb_desc.data_ptr(), kb_qt.data_ptr(), visit.data_ptr()
        a1.n_qtiles, a1.n_kblocks, a1.max_list, a1.n_entries = nqt, nkb, int(sc.kb_desc[:, 3].max()), int(len(sc.kb_qt))
        a1.batch, a1.heads, a1.n, a1.nk_pad, a1.n_ktiles64 = b, heads, N, nk_pad, (N + 63) // 64
        a1.scale, a1.flags = 0.125, H.ATTN_Q_PRESCALED
        H.call("mca_attn_bwd_onepass", C.byref(a1), H.stream_ptr())
        torch.cuda.synchronize()
        out.append((dq, dkv))
        if rep_i == 1:
            # ---- SPLIT mode (small batches): S workgroups per (sample, head), key block kb swept by workgroup kb mod S, the S slices of
            # dQ partials added in slice order by the call's second launch.  dK / dV: the SAME bits as one workgroup per (sample, head)
            # (a key block's sweep does not depend on who runs it); dQ: the same sum in another order (equal up to fp32 rounding before
            # the one bf16 rounding); bitwise repeatable; NaN-poisoned slices on entry
            for S in (2, 3):
                res = []
                for _ in range(2):
                    accS = torch.full((b * heads * S * (nqt + 1) * 4096,), float("nan"), device=dev)
                    dqS = torch.full((b, N, D), 7.0, device=dev, dtype=torch.bfloat16)
                    dkvS = torch.zeros(b, N, 3 * D, dtype=torch.bfloat16, device=dev)
                    a1.dq, a1.dk, a1.dv, a1.dq_acc, a1.split = dqS.data_ptr(), dkvS.data_ptr() + D * 2, dkvS.data_ptr() + 2 * D * 2, accS.data_ptr(), S
                    H.call("mca_attn_bwd_onepass", C.byref(a1), H.stream_ptr())
                    torch.cuda.synchronize()
                    res.append((dqS, dkvS))
                assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), f"split {S}: not repeatable"
                assert not torch.isnan(res[0][0].float()).any() and not torch.isnan(res[0][1].float()).any(), f"split {S}: NaN"
                assert torch.equal(res[0][1], dkv), f"split {S}: dK / dV differ from the unsplit kernel"
                e = rel(res[0][0].float(), dq.float())
                assert e < 3e-3, f"split {S}: dq differs from the unsplit kernel by {e}"
            a1.split = 0
    return out


@pytest.mark.parametrize("variant,pool,drop", [("mca", False, False), ("mca", False, True), ("zorro", False, True),
                                               ("mca", True, True), ("zorro", True, False)])
def test_attention_small(H, variant, pool, drop):
    S = importlib.import_module("mca-paper_amd.structure")
    st = S.FusionStructure([70, 45, 30], 8, (3, 2), fcl=variant == "mca", zorro=variant == "zorro")
    _attention_case(H, st, b=3, heads=2, pool=pool, seed=11, drop_first=drop)


def test_attention_forward_kernels_agree_bit_for_bit(H):
    """the 128-row-tile LDS-DMA forward with the textbook recurrence (4 wavefronts per SIMD) and the register-staged one (what a
    structure with more than 15 key groups gets: no one-hot operand) do the same arithmetic in the same order: identical o and lse
    on the CMU structure with ragged lengths and a dropped modality.  (The lazy-reference form - the engine's default - is equal up
    to rounding: _attention_case.)"""
    P = importlib.import_module("mca-paper_amd")
    b = 3
    cfg = P.config.cmu_model_config(batch_size=b); cfg["depth"] = 1
    torch.manual_seed(0)
    eng = P.MCA(**cfg).cuda().engine
    ws = eng.workspace(b); N, D = eng.N, eng.D
    g = torch.Generator(device="cuda").manual_seed(7)
    ws["padding"].zero_()
    for mi, n in enumerate(eng.st.token_dims):
        ln = torch.randint(1, n + 1, (b,), device="cuda", generator=g)
        if mi == 2:
            ln[1] = 0
        ws["padding"][:, eng.offsets[mi]:eng.offsets[mi] + n] = (torch.arange(n, device="cuda")[None] >= ln[:, None]).to(ws["padding"].dtype)
    H.call("mca_build_keyinfo", ws["padding"].data_ptr(), eng.kgroup.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr(), b, N, eng.nk_pad, H.stream_ptr())
    H.call("mca_build_keyhot", ws["keyinfo"].data_ptr(), ws["khot"].data_ptr(), b, eng.nk_pad, H.stream_ptr())
    a = ws["layers"][0]
    a["qkv"].copy_((torch.randn(b * N, 3 * D, device="cuda", generator=g) * 2.0).bfloat16())
    a["qkv"][:, :D] *= 0.18

    def fwd():
        a["o"].zero_()
        eng._attn_fwd(a["qkv"].data_ptr(), N * 3 * D, 3 * D, a["qkv"], D, 2 * D, 3 * D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N)
        torch.cuda.synchronize()
        return a["o"].clone(), a["lse"].clone()

    eng.attn_flags = H.ATTN_Q_PRESCALED          # (the textbook recurrence in both)
    o4, l4 = fwd()
    khot = ws.pop("khot")
    o1, l1 = fwd()
    ws["khot"] = khot
    assert torch.isinf(l4).any() and torch.isfinite(l4).any()
    assert torch.equal(o4, o1) and torch.equal(l4, l1)


def test_attention_backward_prep_is_bitwise_repeatable_with_uniform_rows(H):
    """mca_attn_bwd_prep: delta, and dvmean = (1/nk) sum of dO over the uniform rows added in row order (no atomics): the same
    bits on every launch, the values of a direct fp64 sum, zero for samples / heads without uniform rows"""
    g = torch.Generator(device="cuda").manual_seed(3)
    b, heads, nq, nk = 3, 2, 1000, 1000
    D = heads * 64
    o = bf(torch.randn(b * nq, D, device="cuda", generator=g)); d_o = bf(torch.randn(b * nq, D, device="cuda", generator=g))
    lse = torch.randn(b, heads, nq, device="cuda", generator=g)
    uni = torch.rand(b, heads, nq, device="cuda", generator=g) < 0.3
    uni[1] = False                                      # a sample without uniform rows
    uni[2, 1] = False                                   # a head without
    lse[uni] = float("inf")
    outs = []
    for _ in range(3):
        delta = torch.empty(b, heads, nq, device="cuda"); dvm = torch.full((b, D), 7.0, device="cuda")
        H.call("mca_attn_bwd_prep", o.data_ptr(), d_o.data_ptr(), nq * D, D, lse.data_ptr(), delta.data_ptr(), dvm.data_ptr(), b, heads, nq, nk, H.stream_ptr())
        torch.cuda.synchronize()
        outs.append((delta.clone(), dvm.clone()))
    assert all(torch.equal(outs[0][0], x[0]) and torch.equal(outs[0][1], x[1]) for x in outs[1:])
    d4 = d_o.double().view(b, nq, heads, 64)
    want = (d4 * uni.permute(0, 2, 1)[..., None]).sum(1).reshape(b, D) / nk
    assert rel(outs[0][1], want.float()) < 1e-5 and float(outs[0][1][1].abs().max()) == 0 and float(outs[0][1][2, 64:].abs().max()) == 0
    want_delta = (d_o.double() * o.double()).view(b, nq, heads, 64).sum(-1).permute(0, 2, 1)
    assert rel(outs[0][0], want_delta.float()) < 1e-5


def test_attention_unprescaled_q_is_refused(H):
    """q must carry scale * log2 e (MCA_ATTN_Q_PRESCALED): the un-prescaled kernel forms (one of which spilled 579 registers)
    left the library in round 3; every attention entry point refuses a call without the flag instead of computing something
    else (MCA_E_UNSUPPORTED = -3)."""
    a = H.AttnFwdArgs()
    one = torch.zeros(64, dtype=torch.int32, device="cuda")
    for f_ in ("q", "k", "v", "o", "lse", "qmask", "keyinfo", "ktile_flags", "q_ptr", "q_kt", "q_order", "vmean"):
        setattr(a, f_, one.data_ptr())
    a.batch, a.heads, a.nq, a.nk, a.nk_pad, a.n_qtiles, a.n_ktiles, a.scale, a.flags = 1, 1, 16, 16, 64, 1, 1, 0.125, 0
    a.q_ld = a.kv_ld = a.o_ld = 64
    assert H.lib().mca_attn_fwd(C.byref(a), None) == -3
    a2 = H.AttnBwd2Args()
    for f_ in ("q", "k", "v", "d_o", "lse", "delta", "dvmean", "dq", "dk", "dv", "qmask", "keyinfo", "ktile_flags", "q_ptr", "q_kt", "q_order", "k_wg", "k_qt"):
        setattr(a2, f_, one.data_ptr())
    a2.batch, a2.heads, a2.nq, a2.nk, a2.nk_pad, a2.scale, a2.flags = 1, 1, 16, 16, 256, 0.125, 0
    a2.q_ld = a2.kv_ld = a2.o_ld = a2.dq_ld = a2.dkv_ld = 64
    a2.n_qtiles128, a2.n_ktiles64, a2.n_qtiles64, a2.n_kblocks256, a2.kblock_keys = 1, 1, 1, 1, 256
    assert H.lib().mca_attn_bwd_dq(C.byref(a2), None) == -3 and H.lib().mca_attn_bwd_dkv(C.byref(a2), None) == -3


def test_attention_spiked_keys(H):
    """keys 40x larger than their neighbours in the first, second and third key tile of a row and at the very end: the running
    maximum of the forward softmax jumps mid-row and starts from scores far below zero."""
    S = importlib.import_module("mca-paper_amd.structure")
    st = S.FusionStructure([300, 100, 60], 8, (3, 2), fcl=True)
    _attention_case(H, st, b=2, heads=2, pool=False, seed=17, drop_first=True, spike=True)


@pytest.mark.parametrize("variant,pool", [("mca", False), ("zorro", False), ("mca", True)])
def test_attention_cmu_shape(H, variant, pool):
    S = importlib.import_module("mca-paper_amd.structure")
    st = S.FusionStructure([1500, 450, 450, 50], 88, (4, 3, 2), fcl=variant == "mca", zorro=variant == "zorro")
    _attention_case(H, st, b=2, heads=2, pool=pool, seed=12, drop_first=True)


@pytest.mark.parametrize("mods,powers,n_groups", [([40, 30, 20, 10], (4, 3, 2), 15), ([40, 30, 20, 10, 24], (5, 4, 3), 21)])
def test_attention_group_count_boundary(H, mods, powers, n_groups):
    """15 key groups is the most the mask product holds (slot 15 = padded keys): the 4-modality structure uses every slot and
    runs both mask paths (bitwise-equal gradients asserted in _attention_case); with 5 modalities and 16 combinations (21
    groups) only the element-wise mask applies and the engine must not offer the one-hot operand."""
    S = importlib.import_module("mca-paper_amd.structure")
    st = S.FusionStructure(mods, 32 if n_groups > 15 else 33, powers, fcl=True)
    assert int(st.kgroup.max()) + 1 == n_groups
    _attention_case(H, st, b=2, heads=2, pool=False, seed=19, drop_first=True)
    _attention_case(H, st, b=2, heads=2, pool=True, seed=20, drop_first=True)


def test_attention_eao_block_diagonal(H):
    """The EAO super-sequence (structure.EAOStructure): 3 + 3 segments, attention block-diagonal over them; a dropped
    modality empties its own segment (uniform rows) and the part of the combination segments it occupies."""
    S = importlib.import_module("mca-paper_amd.structure")
    import types
    e = S.EAOStructure([70, 45, 30], (2,), fcl=True, zorro=False)
    # the harness pads per entry of token_dims: give it every block of the super-sequence (first block dropped in sample 0)
    st = types.SimpleNamespace(n_tokens=e.n_tokens, qmask_attn=e.qmask_attn, qmask_pool=e.qmask_pool, kgroup=e.kgroup,
                               token_dims=e.block_dims, attn_schedule=e.attn_schedule, dense_attn_mask=e.dense_attn_mask)
    assert sum(st.token_dims) == st.n_tokens == 435
    _attention_case(H, st, b=3, heads=2, pool=False, seed=23, drop_first=True)


def test_attention_long_sequence_shape(H):
    """BASELINE config 5 shape: every modality padded to 1500 tokens (N = 6088, 96 key tiles)."""
    S = importlib.import_module("mca-paper_amd.structure")
    st = S.FusionStructure([1500, 1500, 1500, 1500], 88, (4, 3, 2), fcl=True)
    _attention_case(H, st, b=1, heads=2, pool=False, seed=13, drop_first=False)


# ------------------------------------------------------------------------------------------- fp8 attention
@pytest.mark.parametrize("shape", ["small", "cmu", "long"])
def test_attention_fp8_forward(H, shape):
    """BASELINE configs[4]: Q K^T and P V on the block-scaled fp8 matrix instruction.  Checked against (a) the oracle's
    emulation of the same MX-fp8 arithmetic (oracle.fp8_attention_core: e4m3 elements, one power-of-two scale per 32 elements
    along d for Q and K and per 32 consecutive keys for V, P as e4m3(128 * 2^(S - m))): STATED TOLERANCE 1e-2 rel-L2 (accumulation
    order and the fp32 exp2 are the only differences) and (b) the exact fp64 attention: what e4m3 operands cost on these
    inputs (unit-variance q, k rows spread over 6 octaves: logits of +-20, a stress case) is 7-10 %; the kernel must be within
    1.05 x the emulation's own distance + 1e-3 and below 0.12.
    The quantised operands themselves are compared value for value."""
    from oracle import mca_oracle as O
    S = importlib.import_module("mca-paper_amd.structure")
    eng = importlib.import_module("mca-paper_amd.engine")
    if shape == "small":
        st, b, heads = S.FusionStructure([70, 45, 30], 8, (3, 2), fcl=True), 3, 2
    elif shape == "cmu":
        st, b, heads = S.FusionStructure([1500, 450, 450, 50], 88, (4, 3, 2), fcl=True), 2, 2
    else:
        st, b, heads = S.FusionStructure([1500, 1500, 1500, 1500], 88, (4, 3, 2), fcl=True), 1, 2          # N = 6088
    dev = "cuda"
    N, D = st.n_tokens, heads * 64
    g = torch.Generator(device=dev).manual_seed(21)
    sf = eng._Sched(st.attn_schedule(128, 64), dev)
    qmask = torch.from_numpy(st.qmask_attn.astype(np.uint32).view(np.int32)).to(dev)
    kgroup = torch.from_numpy(st.kgroup).to(dev)
    allowed = torch.from_numpy(~st.dense_attn_mask()).to(dev)
    pad = torch.zeros(b, N, dtype=torch.bool, device=dev)
    off = 0
    for mi, n in enumerate(st.token_dims):
        ln = torch.randint(1, n + 1, (b,), generator=g, device=dev)
        if mi == 0:
            ln[0] = 0                                   # a dropped modality: uniform rows
        pad[:, off:off + n] = torch.arange(n, device=dev)[None] >= ln[:, None]
        off += n
    qkv = torch.randn(b, N, 3 * D, device=dev, generator=g)
    qkv[:, :, D:2 * D] *= torch.exp2(torch.randint(-3, 4, (b, N, 1), device=dev, generator=g).float())          # rows of very different magnitude
    qkv = bf(qkv)
    qkv[:, :, :D] = bf(qkv[:, :, :D].float() * C2)          # q as the engine stores it
    nk_pad = (N + 255) // 256 * 256
    nt = (N + 63) // 64
    keyinfo = torch.empty(b, nk_pad, dtype=torch.uint8, device=dev)
    kflags = torch.empty(b, nt, dtype=torch.uint8, device=dev)
    H.call("mca_build_keyinfo", pad.to(torch.uint8).data_ptr(), kgroup.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr(), b, N, nk_pad, H.stream_ptr())
    vmean = torch.empty(b, D, device=dev)
    vptr = qkv.data_ptr() + 2 * D * 2
    H.call("mca_attn_vmean", vptr, N * 3 * D, 3 * D, vmean.data_ptr(), b, N, heads, H.stream_ptr())
    u8 = lambda *s_: torch.zeros(*s_, dtype=torch.uint8, device=dev)
    q8, qs, k8, ks, v8t, vs = u8(b, heads, nt * 64, 64), u8(b, heads, nt * 64, 2), u8(b, heads, nt * 64, 64), u8(b, heads, nt * 64, 2), u8(b, heads, nt, 64, 64), u8(b, heads, nt, 64, 2)
    f = H.AttnFp8Operands()
    f.q8, f.qs, f.k8, f.ks, f.v8t, f.vs, f.n_ktiles = q8.data_ptr(), qs.data_ptr(), k8.data_ptr(), ks.data_ptr(), v8t.data_ptr(), vs.data_ptr(), nt
    H.call("mca_attn_quant_mxfp8", qkv.data_ptr(), N * 3 * D, 3 * D, qkv.data_ptr() + D * 2, vptr, N * 3 * D, 3 * D, C.byref(f), b, heads, N, H.stream_ptr())
    torch.cuda.synchronize()
    # ---- operands, value for value
    q4 = qkv[:, :, :D].float().view(b, N, heads, 64).permute(0, 2, 1, 3)
    k4 = qkv[:, :, D:2 * D].float().view(b, N, heads, 64).permute(0, 2, 1, 3)
    v4 = qkv[:, :, 2 * D:].float().view(b, N, heads, 64).permute(0, 2, 1, 3)
    deq = lambda x8, xs: x8.view(torch.float8_e4m3fn).float().view(*x8.shape[:-1], 2, 32) * torch.exp2(xs.float() - 127.0)[..., None]
    assert torch.equal(deq(q8, qs).flatten(-2)[:, :, :N], O.mx_e4m3(q4, -1))
    assert torch.equal(deq(k8, ks).flatten(-2)[:, :, :N], O.mx_e4m3(k4, -1))
    pp = torch.arange(64, device=dev)          # position -> key inside a 64-key tile (include/mca_hip.h, mca_attn_fp8_operands)
    pos = (torch.arange(0, nt * 64, 64, device=dev)[:, None] + (32 * (pp >> 5) + 8 * ((pp >> 2) & 3) + 4 * ((pp >> 4) & 1) + (pp & 3))[None]).reshape(-1)
    vp = torch.nn.functional.pad(v4, (0, 0, 0, nt * 64 - N))[:, :, pos, :]                    # (b, h, positions, d)
    want_v = O.mx_e4m3(vp, 2).view(b, heads, nt, 64, 64).transpose(-1, -2)                    # (b, h, tile, d, position)
    assert torch.equal(deq(v8t, vs).flatten(-2), want_v)
    # ---- forward
    o = torch.zeros(b * N, D, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(b, heads, N, device=dev)
    a = H.AttnFwdArgs()
    a.q, a.q_bstride, a.q_ld = qkv.data_ptr(), N * 3 * D, 3 * D
    a.k, a.v, a.kv_bstride, a.kv_ld = qkv.data_ptr() + D * 2, vptr, N * 3 * D, 3 * D
    a.o, a.o_bstride, a.o_ld, a.lse = o.data_ptr(), N * D, D, lse.data_ptr()
    a.qmask, a.keyinfo, a.ktile_flags = qmask.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr()
    a.q_ptr, a.q_kt, a.q_order = sf.q_ptr.data_ptr(), sf.q_kt.data_ptr(), sf.q_order.data_ptr()
    a.vmean = vmean.data_ptr()
    a.batch, a.heads, a.nq, a.nk, a.nk_pad, a.n_qtiles, a.n_ktiles, a.scale = b, heads, N, N, nk_pad, sf.s.n_q, sf.s.n_k, 0.125
    a.flags = H.ATTN_Q_PRESCALED
    H.call("mca_attn_fwd_fp8", C.byref(a), C.byref(f), H.stream_ptr())
    torch.cuda.synchronize()
    # the mask as a matrix product (mca_build_keyhot): the same bits
    khot = torch.empty(b, nk_pad, 16, dtype=torch.bfloat16, device=dev)
    H.call("mca_build_keyhot", keyinfo.data_ptr(), khot.data_ptr(), b, nk_pad, H.stream_ptr())
    o_h = torch.zeros_like(o); lse_h = torch.empty_like(lse)
    a.o, a.lse, a.khot = o_h.data_ptr(), lse_h.data_ptr(), khot.data_ptr()
    H.call("mca_attn_fwd_fp8", C.byref(a), C.byref(f), H.stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(o_h, o) and torch.equal(lse_h, lse)
    got = o.float().view(b, N, heads, 64).permute(0, 2, 1, 3)
    blocked = (~allowed)[None, None] | pad[:, None, None, :]
    emu = O.fp8_attention_core(q4, k4, v4, blocked)
    exact = dense_attention((q4 / C2).double(), k4.double(), v4.double(), allowed, pad, 0.125)
    e_emu, e_exact, e_floor = rel(got, emu), rel(got, exact.float()), rel(emu, exact.float())
    assert e_emu < 1e-2, f"fp8 forward vs its emulation: {e_emu}"
    assert e_exact < 0.12 and e_exact < 1.05 * e_floor + 1e-3, f"fp8 forward vs exact attention: {e_exact} (emulation itself: {e_floor})"
    uni = blocked.all(-1)[:, 0]                                                               # (b, N)
    assert torch.equal(torch.isinf(lse[:, 0]), uni) and uni.any()
    fin = ~uni
    lse_ref = torch.logsumexp((torch.einsum("bhid,bhjd->bhij", O.mx_e4m3(q4, -1), O.mx_e4m3(k4, -1)).masked_fill(blocked, float("-inf")) * 0.6931471805599453)[:, 0], -1) * 1.4426950408889634
    assert (lse[:, 0][fin] - lse_ref[fin]).abs().max() < 2e-2


@pytest.mark.parametrize("shape", ["small", "cmu", "long"])
def test_attention_fp8_backward(H, shape):
    """BASELINE configs[4], backward: S = Q K^T and dP = dO V^T recomputed on the block-scaled fp8 matrix instruction in both
    passes (mca_attn_quant_bwd_mxfp8 + mca_attn_bwd_dq_fp8 / mca_attn_bwd_dkv_fp8), gradient products in bf16.  Checked against
    the oracle's restatement of the same arithmetic (oracle._Fp8AttentionCore.backward): the four quantised operands value for
    value, dq / dk / dv within a STATED 2e-2 rel-L2 (accumulation order, fp32 exp2, and delta from the kernel's own bf16 O), and
    against the exact fp64 gradients: within 1.1 x the emulation's own distance + 5e-3 (what e4m3 operands cost on this stress
    input).  Bitwise repeatable."""
    from oracle import mca_oracle as O
    S = importlib.import_module("mca-paper_amd.structure")
    eng = importlib.import_module("mca-paper_amd.engine")
    if shape == "small":
        st, b, heads = S.FusionStructure([70, 45, 30], 8, (3, 2), fcl=True), 3, 2
    elif shape == "cmu":
        st, b, heads = S.FusionStructure([1500, 450, 450, 50], 88, (4, 3, 2), fcl=True), 2, 2
    else:
        st, b, heads = S.FusionStructure([1500, 1500, 1500, 1500], 88, (4, 3, 2), fcl=True), 1, 1          # N = 6088
    dev = "cuda"
    N, D = st.n_tokens, heads * 64
    g = torch.Generator(device=dev).manual_seed(23)
    sf = eng._Sched(st.attn_schedule(128, 64), dev)
    sb = eng._Sched(st.attn_schedule(64, 128), dev)
    qmask = torch.from_numpy(st.qmask_attn.astype(np.uint32).view(np.int32)).to(dev)
    bits = (st.qmask_attn.astype(np.uint32)[:, None] >> np.arange(16, dtype=np.uint32)[None, :]) & 1
    bits[:, 15] = 0
    qblk = torch.from_numpy(np.where(bits == 1, 0.0, -32768.0).astype(np.float32)).to(torch.bfloat16).to(dev).contiguous()
    kgroup = torch.from_numpy(st.kgroup).to(dev)
    allowed = torch.from_numpy(~st.dense_attn_mask()).to(dev)
    pad = torch.zeros(b, N, dtype=torch.bool, device=dev)
    off = 0
    for mi, n in enumerate(st.token_dims):
        ln = torch.randint(1, n + 1, (b,), generator=g, device=dev)
        if mi == 0:
            ln[0] = 0                                   # a dropped modality: uniform rows
        pad[:, off:off + n] = torch.arange(n, device=dev)[None] >= ln[:, None]
        off += n
    qkv = torch.randn(b, N, 3 * D, device=dev, generator=g)
    qkv[:, :, D:2 * D] *= torch.exp2(torch.randint(-2, 3, (b, N, 1), device=dev, generator=g).float())
    qkv = bf(qkv)
    qkv[:, :, :D] = bf(qkv[:, :, :D].float() * C2)
    d_o = bf(torch.randn(b * N, D, device=dev, generator=g) * torch.exp2(torch.randint(-2, 3, (b * N, 1), device=dev, generator=g).float()))
    nk_pad = (N + 255) // 256 * 256
    nt = (N + 63) // 64
    keyinfo = torch.empty(b, nk_pad, dtype=torch.uint8, device=dev)
    kflags = torch.empty(b, nt, dtype=torch.uint8, device=dev)
    H.call("mca_build_keyinfo", pad.to(torch.uint8).data_ptr(), kgroup.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr(), b, N, nk_pad, H.stream_ptr())
    khot = torch.empty(b, nk_pad, 16, dtype=torch.bfloat16, device=dev)
    H.call("mca_build_keyhot", keyinfo.data_ptr(), khot.data_ptr(), b, nk_pad, H.stream_ptr())
    vmean = torch.empty(b, D, device=dev)
    kptr, vptr = qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2
    H.call("mca_attn_vmean", vptr, N * 3 * D, 3 * D, vmean.data_ptr(), b, N, heads, H.stream_ptr())
    u8 = lambda *s_: torch.zeros(*s_, dtype=torch.uint8, device=dev)
    f = H.AttnFp8Operands()
    fb_ = {k: (u8(b, heads, nt, 64, 64) if k == "v8t" else u8(b, heads, nt, 64, 2) if k == "vs" else u8(b, heads, nt * 64, 64 if k.endswith("8") else 2))
           for k in ("q8", "qs", "k8", "ks", "v8t", "vs")}
    for k_, t in fb_.items():
        setattr(f, k_, t.data_ptr())
    f.n_ktiles = nt
    H.call("mca_attn_quant_mxfp8", qkv.data_ptr(), N * 3 * D, 3 * D, kptr, vptr, N * 3 * D, 3 * D, C.byref(f), b, heads, N, H.stream_ptr())
    o = torch.zeros(b * N, D, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(b, heads, N, device=dev)
    a = H.AttnFwdArgs()
    a.q, a.q_bstride, a.q_ld = qkv.data_ptr(), N * 3 * D, 3 * D
    a.k, a.v, a.kv_bstride, a.kv_ld = kptr, vptr, N * 3 * D, 3 * D
    a.o, a.o_bstride, a.o_ld, a.lse = o.data_ptr(), N * D, D, lse.data_ptr()
    a.qmask, a.keyinfo, a.ktile_flags = qmask.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr()
    a.q_ptr, a.q_kt, a.q_order = sf.q_ptr.data_ptr(), sf.q_kt.data_ptr(), sf.q_order.data_ptr()
    a.vmean, a.khot = vmean.data_ptr(), khot.data_ptr()
    a.batch, a.heads, a.nq, a.nk, a.nk_pad, a.n_qtiles, a.n_ktiles, a.scale = b, heads, N, N, nk_pad, sf.s.n_q, sf.s.n_k, 0.125
    a.flags = H.ATTN_Q_PRESCALED
    H.call("mca_attn_fwd_fp8", C.byref(a), C.byref(f), H.stream_ptr())
    # ---- backward
    delta = torch.empty(b, heads, N, device=dev)
    dvmean = torch.empty(b, D, device=dev)
    H.call("mca_attn_bwd_prep", o.data_ptr(), d_o.data_ptr(), N * D, D, lse.data_ptr(), delta.data_ptr(), dvmean.data_ptr(), b, heads, N, N, H.stream_ptr())
    fo = H.AttnFp8BwdOperands()
    ob = {k: u8(b, heads, nt * 64, 64 if k.endswith("8") else 2) for k in ("q8", "qs", "k8", "ks", "v8", "vs", "do8", "dos")}
    for k_, t in ob.items():
        setattr(fo, k_, t.data_ptr())
    fo.n_ktiles = nt
    H.call("mca_attn_quant_bwd_mxfp8", qkv.data_ptr(), N * 3 * D, 3 * D, kptr, vptr, N * 3 * D, 3 * D, d_o.data_ptr(), N * D, D, C.byref(fo), 15, b, heads, N, H.stream_ptr())
    torch.cuda.synchronize()
    sp4 = lambda t: t.float().view(b, N, heads, 64).permute(0, 2, 1, 3)
    q4, k4, v4, do4 = sp4(qkv[:, :, :D]), sp4(qkv[:, :, D:2 * D]), sp4(qkv[:, :, 2 * D:]), sp4(d_o.view(b, N, D))
    deq = lambda x8, xs: x8.view(torch.float8_e4m3fn).float().view(*x8.shape[:-1], 2, 32) * torch.exp2(xs.float() - 127.0)[..., None]
    for name, ref in (("q", q4), ("k", k4), ("v", v4), ("do", do4)):
        got = deq(ob[name + "8"], ob[name + "s"]).flatten(-2)
        assert torch.equal(got[:, :, :N], O.mx_e4m3(ref, -1)), name
        assert float(got[:, :, N:].abs().max()) == 0 if nt * 64 > N else True
    assert torch.equal(ob["q8"], fb_["q8"]) and torch.equal(ob["k8"], fb_["k8"])          # the forward's operands, bit for bit
    dqkv = torch.zeros(b * N, 3 * D, dtype=torch.bfloat16, device=dev)

    def run():
        dqkv.zero_()
        a2 = H.AttnBwd2Args()
        a2.q, a2.q_bstride, a2.q_ld = qkv.data_ptr(), N * 3 * D, 3 * D
        a2.k, a2.v, a2.kv_bstride, a2.kv_ld = kptr, vptr, N * 3 * D, 3 * D
        a2.d_o, a2.o_bstride, a2.o_ld = d_o.data_ptr(), N * D, D
        a2.lse, a2.delta, a2.dvmean = lse.data_ptr(), delta.data_ptr(), dvmean.data_ptr()
        a2.dq, a2.dq_bstride, a2.dq_ld, a2.dq_f32 = dqkv.data_ptr(), N * 3 * D, 3 * D, 0
        a2.dk, a2.dv, a2.dkv_bstride, a2.dkv_ld = dqkv.data_ptr() + D * 2, dqkv.data_ptr() + 2 * D * 2, N * 3 * D, 3 * D
        a2.qmask, a2.keyinfo, a2.ktile_flags = qmask.data_ptr(), keyinfo.data_ptr(), kflags.data_ptr()
        a2.q_ptr, a2.q_kt, a2.q_order, a2.n_qtiles128, a2.n_ktiles64 = sf.q_ptr.data_ptr(), sf.q_kt.data_ptr(), sf.q_order.data_ptr(), sf.s.n_q, sf.s.n_k
        a2.k_wg, a2.k_qt, a2.n_qtiles64, a2.n_kblocks256 = sb.k_wg.data_ptr(), sb.k_qt.data_ptr(), sb.s.n_q, sb.s.n_k
        a2.batch, a2.heads, a2.nq, a2.nk, a2.nk_pad, a2.scale, a2.flags = b, heads, N, N, nk_pad, 0.125, H.ATTN_Q_PRESCALED
        a2.khot, a2.qblk, a2.kblock_keys = khot.data_ptr(), qblk.data_ptr(), 128
        H.call("mca_attn_bwd_dkv_fp8", C.byref(a2), C.byref(fo), H.stream_ptr())
        H.call("mca_attn_bwd_dq_fp8", C.byref(a2), C.byref(fo), H.stream_ptr())
        torch.cuda.synchronize()
        return dqkv.clone()

    g1, g2 = run(), run()
    assert torch.equal(g1, g2)                                   # no atomics: bitwise repeatable
    # ---- the oracle's restatement of both directions
    blocked = (~allowed)[None, None] | pad[:, None, None, :]
    # (a) both directions emulated; (b) the backward alone, from the KERNEL's forward results (o, lse): isolates the two
    # backward kernels from the forward's own rounding (delta = rowsum(dO o O) amplifies an O difference on peaked rows)
    qe, ke, ve = (t.clone().requires_grad_(True) for t in (q4, k4, v4))
    oe = O.fp8_attention_core(qe, ke, ve, blocked)
    oe.backward(do4)
    iso = O.fp8_attention_backward(q4, k4, v4, blocked, sp4(o.view(b, N, D)), lse[..., None], do4)
    # exact fp64 attention gradients on the same (bf16) inputs; q4 is the log2-domain query: S = q4 . k * ln 2
    qx, kx, vx = (t.double().clone().requires_grad_(True) for t in (q4, k4, v4))
    sx = (torch.einsum("bhid,bhjd->bhij", qx, kx) * 0.6931471805599453).masked_fill(blocked, -torch.finfo(torch.float64).max)
    torch.einsum("bhij,bhjd->bhid", sx.softmax(-1), vx).backward(do4.double())
    got = g1.float().view(b, N, 3, heads, 64).permute(2, 0, 3, 1, 4)          # (3, b, h, N, 64)
    # the kernel's dq is the gradient w.r.t. the UNSCALED q (= dq2 * scale * log2 e, include/mca_hip.h)
    for i, (name, emu, exact, fac) in enumerate((("dq", qe.grad, qx.grad, C2), ("dk", ke.grad, kx.grad, 1.0), ("dv", ve.grad, vx.grad, 1.0))):
        e_emu, e_exact, e_floor = rel(got[i], emu * fac), rel(got[i], (exact * fac).float()), rel(emu * fac, (exact * fac).float())
        e_iso = rel(got[i], iso[i] * fac)
        assert e_emu < 2e-2, f"{name}: fp8 backward vs its emulation {e_emu}"
        assert e_iso < 1e-2, f"{name}: fp8 backward kernels vs their restatement on the kernel's own forward results {e_iso}"
        assert e_exact < 1.1 * e_floor + 5e-3, f"{name}: vs exact {e_exact} (emulation itself {e_floor})"
        # an ABSOLUTE cap on the distance from the exact fp64 gradients (the forward has one too): e4m3 operands in S and dP
        # cost 3-5 % at the CMU / LONG shapes and 10.1 % (dq) on the small shape's stress inputs, the emulation's own distance
        # being the same to three digits; 0.15 stated
        assert e_exact < 0.15 and e_floor < 0.15, f"{name}: fp8 backward vs exact gradients {e_exact}, its emulation {e_floor}"
        # per-row check (a wrong sub-tile hides in the global norm); rows with a near-zero gradient are measured against a
        # twentieth of the mean row norm
        rn = (iso[i] * fac).norm(dim=-1)
        rows = (got[i] - iso[i] * fac).norm(dim=-1) / (rn + 0.05 * rn.mean())
        assert float(rows.max()) < 0.1, f"{name}: worst row {float(rows.max())} e_iso {e_iso}"

"""GPU parity tests of the non-attention kernels (GEMM forms, LayerNorm, GEGLU, loss, optimizer, copies / masks) through the C ABI against
fp32 / fp64 torch formulas."""
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


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(300, 200, 128), (1000, 1536, 512), (129, 2816, 512), (16, 512, 512), (4060, 512, 1408),
                                   (4100, 1536, 320), (2600, 2816, 512),          # persistent kernels, grouped column tiles
                                   (17920, 1024, 192), (17700, 1024, 256)])       # 280 tiles of 256 x 256 on 256 CUs: second tile per workgroup, shortest k-loops
def test_gemm_nt(H, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = bf(torch.randn(M, K, device="cuda", generator=g))
    B = bf(torch.randn(N, K, device="cuda", generator=g))
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    ref = A.float() @ B.float().t()
    C32 = torch.empty(M, N, device="cuda")
    H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, C32.data_ptr(), N, 0, None, None, 0, 0, M, N, K, H.stream_ptr())
    assert rel(C32, ref) < 1e-5
    H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, C32.data_ptr(), N, 0, bias.data_ptr(), res.data_ptr(), N, 0, M, N, K, H.stream_ptr())
    assert rel(C32, ref + bias + res) < 1e-5
    Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, Cb.data_ptr(), N, 1, None, None, 0, 0, M, N, K, H.stream_ptr())
    assert rel(Cb.float(), ref) < 4e-3
    # broadcast residual (row % period)
    per = 16 if M % 16 == 0 else 0
    if per:
        r2 = torch.randn(per, N, device="cuda", generator=g)
        H.call("mca_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, C32.data_ptr(), N, 0, None, r2.data_ptr(), N, per, M, N, K, H.stream_ptr())
        assert rel(C32, ref + r2.repeat(M // per, 1)) < 1e-5


@pytest.mark.parametrize("M,N,K", [(2304, 512, 512), (4100, 512, 1408), (2100, 256, 576)])
def test_gemm_nt_lnres(H, M, N, K):
    """C = A B^T + LayerNorm(x), the LayerNorm recomputed in the epilogue from x and the statistics the LN kernel saved."""
    g = torch.Generator(device="cuda").manual_seed(11)
    A = bf(torch.randn(M, K, device="cuda", generator=g))
    B = bf(torch.randn(N, K, device="cuda", generator=g) * 0.1)
    x = torch.randn(M, N, device="cuda", generator=g) * 3 + 0.5
    gamma = torch.randn(N, device="cuda", generator=g)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    xn_b = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    H.call("mca_layernorm_fwd", x.data_ptr(), N, gamma.data_ptr(), None, None, None, 0, None, 0, 0, xn_b.data_ptr(), N, N,
           mean.data_ptr(), rstd.data_ptr(), M, N, 1e-5, H.stream_ptr())
    C = torch.empty(M, N, device="cuda")
    H.call("mca_gemm_nt_lnres", A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, x.data_ptr(), N, mean.data_ptr(), rstd.data_ptr(),
           gamma.data_ptr(), M, N, K, H.stream_ptr())
    ref = A.float() @ B.float().t() + torch.nn.functional.layer_norm(x, (N,), gamma, None, 1e-5)
    assert rel(C, ref) < 2e-6
    # unsupported shapes are refused (the caller keeps the two-kernel form)
    assert H.lib().mca_gemm_nt_lnres(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, x.data_ptr(), N, mean.data_ptr(), rstd.data_ptr(),
                                     gamma.data_ptr(), 100, N, K, H.stream_ptr()) == -3


@pytest.mark.parametrize("R,N,K,lda,ldb", [(1000, 512, 512, 512, 512), (777, 1365, 512, 2816, 512), (2048, 512, 1365, 512, 1408),
                                           (16, 512, 512, 512, 512), (500, 128, 74, 128, 128), (5000, 1024, 512, 1536, 512)])
def test_gemm_tn_acc(H, R, N, K, lda, ldb):
    g = torch.Generator(device="cuda").manual_seed(2)
    A = bf(torch.randn(R, lda, device="cuda", generator=g))
    B = bf(torch.randn(R, ldb, device="cuda", generator=g))
    Cg = torch.randn(N, K, device="cuda", generator=g)
    ref = Cg + A[:, :N].float().t() @ B[:, :K].float()
    H.call("mca_gemm_tn_acc", A.data_ptr(), lda, B.data_ptr(), ldb, Cg.data_ptr(), K, R, N, K, H.stream_ptr())
    assert rel(Cg, ref) < 2e-5


@pytest.mark.parametrize("R,members", [
    (8192, [(1536, 512, 1536, 512), (1365, 512, 2816, 512), (1365, 512, 2816, 512), (512, 1365, 512, 1408)]),   # a layer's four
    # a layer's five (with the out-projection): 52 tiles = 4 whole splits + 48 spans that end one tile's rows and begin the next's
    (8200, [(1536, 512, 1536, 512), (1365, 512, 2816, 512), (1365, 512, 2816, 512), (512, 1365, 512, 1408), (512, 512, 512, 512)]),
    # ... and the pooling key/value projection on top (the top layer's launch), at the b = 8 row count
    (20304, [(1536, 512, 1536, 512), (1365, 512, 2816, 512), (1365, 512, 2816, 512), (512, 1365, 512, 1408), (512, 512, 512, 512),
             (1024, 512, 1024, 512)]),
    (4100, [(512, 512, 512, 512), (300, 700, 304, 704), (1024, 256, 1024, 256), (256, 256, 256, 256), (515, 260, 520, 264)]),
    (5000, [(512, 512, 512, 512), (100, 512, 104, 512)]),          # a member the grouped kernel does not take -> single launches
    (300, [(512, 512, 512, 512), (512, 256, 512, 256)]),           # too few rows -> single launches
])
def test_gemm_tn_acc_group(H, R, members):
    """mca_gemm_tn_acc_group == the single-problem results, member by member (incl. column offsets into a shared operand)"""
    g = torch.Generator(device="cuda").manual_seed(12)
    arr = (H.TnDesc * len(members))()
    keep, refs = [], []
    for d, (N, K, lda, ldb) in zip(arr, members):
        A = bf(torch.randn(R, lda, device="cuda", generator=g))
        B = bf(torch.randn(R, ldb, device="cuda", generator=g))
        Cg = torch.randn(N, K, device="cuda", generator=g)
        refs.append(Cg + A[:, :N].float().t() @ B[:, :K].float())
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.N, d.K = A.data_ptr(), lda, B.data_ptr(), ldb, Cg.data_ptr(), K, N, K
        keep.append((A, B, Cg))
    H.call("mca_gemm_tn_acc_group", C.byref(arr), len(members), R, H.stream_ptr())
    torch.cuda.synchronize()
    for (A, B, Cg), ref in zip(keep, refs):
        assert rel(Cg, ref) < 2e-5
    # uniform row splits (knob 3: the partition without the tile-major line) add the same products once more
    H.lib().mca_debug_set(3, 3)
    try:
        H.call("mca_gemm_tn_acc_group", C.byref(arr), len(members), R, H.stream_ptr())
        torch.cuda.synchronize()
    finally:
        H.lib().mca_debug_set(3, 0)
    for (A, B, Cg), ref in zip(keep, refs):
        assert rel(Cg - A[:, :Cg.shape[0]].float().t() @ B[:, :Cg.shape[1]].float(), ref) < 4e-5


# ------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,cols", [(4099, 512), (8192, 256), (4097, 1024), (5, 512)])
def test_layernorm_fwd_trunk_form(H, rows, cols):
    """gamma-only, bf16 output + statistics: the two-rows-per-wavefront kernel (odd row counts included) against torch and
    against the general kernel (knob 12)."""
    g = torch.Generator(device="cuda").manual_seed(31)
    x = torch.randn(rows, cols, device="cuda", generator=g) * 3 + 1.5
    gamma = torch.randn(cols, device="cuda", generator=g)
    outs = []
    for general in (0, 1):
        H.lib().mca_debug_set(12, general)
        yb = torch.full((rows, cols), 7.0, device="cuda", dtype=torch.bfloat16)
        mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        H.call("mca_layernorm_fwd", x.data_ptr(), cols, gamma.data_ptr(), None, None, None, 0, None, 0, 0, yb.data_ptr(), cols, cols,
               mean.data_ptr(), rstd.data_ptr(), rows, cols, 1e-5, H.stream_ptr())
        torch.cuda.synchronize()
        outs.append((yb, mean, rstd))
    H.lib().mca_debug_set(12, 0)
    ref = torch.nn.functional.layer_norm(x, (cols,), gamma, None, 1e-5)
    assert rel(outs[0][0].float(), ref) < 4e-3
    assert rel(outs[0][1], x.mean(1)) < 1e-5 and rel(outs[0][2], (x.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    assert rel(outs[0][0].float(), outs[1][0].float()) < 1e-3 and rel(outs[0][1], outs[1][1]) < 1e-6 and rel(outs[0][2], outs[1][2]) < 1e-6


@pytest.mark.parametrize("rows,cols", [(4099, 74), (1000, 35), (333, 256), (50, 20), (7, 1)])
def test_layernorm_fwd_narrow_rows(H, rows, cols):
    """An encoder's input norm (encoders.py:189): affine, pad mask, bf16 output padded to the GEMM's K, no fp32 output: the
    sixteen-lanes-per-row kernel against torch and against the general kernel (knob 12)."""
    g = torch.Generator(device="cuda").manual_seed(37)
    x = torch.randn(rows, cols, device="cuda", generator=g) * 2 + 0.5
    gamma, beta = torch.randn(cols, device="cuda", generator=g), torch.randn(cols, device="cuda", generator=g)
    mask = torch.rand(rows, device="cuda", generator=g) < 0.3
    mm = mask.to(torch.uint8)
    cols_pad = (cols + 7) // 8 * 8 + 8
    outs = []
    for general in (0, 1):
        H.lib().mca_debug_set(12, general)
        yb = torch.full((rows, cols_pad), 7.0, device="cuda", dtype=torch.bfloat16)
        mean, rstd = torch.full((rows,), 3.0, device="cuda"), torch.full((rows,), 3.0, device="cuda")
        H.call("mca_layernorm_fwd", x.data_ptr(), cols, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), None, 0, None, 0, 0, yb.data_ptr(), cols_pad,
               cols_pad, mean.data_ptr(), rstd.data_ptr(), rows, cols, 1e-5, H.stream_ptr())
        torch.cuda.synchronize()
        outs.append((yb, mean, rstd))
    H.lib().mca_debug_set(12, 0)
    ref = torch.nn.functional.layer_norm(x, (cols,), gamma, beta, 1e-5).masked_fill(mask[:, None], 0.0)
    yb, mean, rstd = outs[0]
    assert rel(yb[:, :cols].float(), ref) < 4e-3 and (yb[:, cols:] == 0).all()
    keep = ~mask
    assert (mean[mask] == 0).all() and (rstd[mask] == 0).all()
    assert rel(mean[keep], x[keep].mean(1)) < 1e-5 and rel(rstd[keep], (x[keep].var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    assert rel(yb.float(), outs[1][0].float()) < 1e-3 and rel(mean, outs[1][1]) < 1e-6 and rel(rstd, outs[1][2]) < 1e-6


@pytest.mark.parametrize("rows,cols,affine,masked", [(1000, 512, False, False), (333, 74, True, True), (64, 713, True, True),
                                                      (90, 128, True, True), (4099, 512, False, False), (777, 256, False, False),
                                                      (33, 1024, False, False)])
def test_layernorm_fwd_bwd(H, rows, cols, affine, masked):
    g = torch.Generator(device="cuda").manual_seed(3)
    period = 30 if masked else 0
    nb = rows // period if period else 0
    if period:
        rows = nb * period
    x = torch.randn(rows, cols, device="cuda", generator=g) * 2 + 0.5
    gamma = torch.randn(cols, device="cuda", generator=g)
    beta = torch.randn(cols, device="cuda", generator=g) if affine else None
    mask = (torch.rand(rows, device="cuda", generator=g) < 0.3) if masked else None
    add = torch.randn(period, cols, device="cuda", generator=g) if period else None
    cols_pad = (cols + 63) // 64 * 64
    # packed output: (nb, NTOT, cols) with this block at row offset 5
    NTOT = period + 11 if period else 0
    y = torch.zeros(nb, NTOT, cols, device="cuda") if period else torch.empty(rows, cols, device="cuda")
    yb = torch.full((rows, cols_pad), 7.0, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    yptr = y.data_ptr() + (5 * cols * 4 if period else 0)
    mm = mask.to(torch.uint8) if masked else None
    H.call("mca_layernorm_fwd", x.data_ptr(), cols, gamma.data_ptr(), H.ptr(beta), H.ptr(mm),
           H.ptr(add), period, yptr, cols, NTOT * cols, yb.data_ptr(), cols_pad, cols_pad, mean.data_ptr(), rstd.data_ptr(), rows, cols,
           1e-5, H.stream_ptr())
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True) if affine else None
    xin = xr.masked_fill(mask[:, None], 0.0) if masked else xr
    ref = torch.nn.functional.layer_norm(xin, (cols,), gr, br if affine else torch.zeros_like(gr), 1e-5)
    if masked:
        ref = ref.masked_fill(mask[:, None], 0.0)
    ref_b = ref
    ref_y = ref + add.repeat(nb, 1) if period else ref
    got_y = y[:, 5:5 + period].reshape(rows, cols) if period else y
    assert rel(got_y, ref_y) < 1e-5
    assert rel(yb[:, :cols].float(), ref_b) < 4e-3
    assert (yb[:, cols:] == 0).all()
    # backward
    dy_full = torch.randn_like(y)
    dy = dy_full[:, 5:5 + period].reshape(rows, cols) if period else dy_full
    ref_y.backward(dy)
    dx = torch.empty(rows, cols, device="cuda")
    dxb = torch.empty(rows, cols_pad, device="cuda", dtype=torch.bfloat16)
    dgamma = torch.zeros(cols, device="cuda")
    dbeta = torch.zeros(cols, device="cuda") if affine else None
    dyptr = dy_full.data_ptr() + (5 * cols * 4 if period else 0)
    dxsum = torch.randn(cols, device="cuda", generator=g) if affine else None          # (accumulated into: not zero)
    dxsum0 = dxsum.clone() if affine else None
    H.call("mca_layernorm_bwd", dyptr, cols, NTOT * cols, period, x.data_ptr(), cols, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
           H.ptr(mm), dx.data_ptr(), cols, dxb.data_ptr(), cols_pad, dgamma.data_ptr(), H.ptr(dbeta), H.ptr(dxsum), rows, cols, H.stream_ptr())
    gx = xr.grad
    assert rel(dx, gx) < 2e-5
    assert rel(dxb[:, :cols].float(), gx) < 4e-3
    assert rel(dgamma, gr.grad) < 2e-5
    if affine:
        assert rel(dbeta, br.grad) < 2e-5
        # the column sums of dx (bias gradient of the Linear in front of the norm), same launch.  dx sums to ~0 along a row, not along a column
        assert (dxsum - dxsum0 - gx.sum(0)).abs().max() < 2e-4 * gx.abs().sum(0).max()
    # parameter gradients only (no dx asked for: an encoder's input norm): the column-parallel kernel
    dgamma2 = torch.zeros(cols, device="cuda")
    dbeta2 = torch.zeros(cols, device="cuda") if affine else None
    H.call("mca_layernorm_bwd", dyptr, cols, NTOT * cols, period, x.data_ptr(), cols, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
           H.ptr(mm), None, 0, None, 0, dgamma2.data_ptr(), H.ptr(dbeta2), None, rows, cols, H.stream_ptr())
    assert rel(dgamma2, gr.grad) < 2e-5
    if affine:
        assert rel(dbeta2, br.grad) < 2e-5


# ----------------------------------------------------------------------------------------------- GEGLU
def test_geglu(H):
    g = torch.Generator(device="cuda").manual_seed(4)
    rows, ip = 777, 1408
    h = bf(torch.randn(rows, 2 * ip, device="cuda", generator=g))
    out = torch.empty(rows, ip, device="cuda", dtype=torch.bfloat16)
    H.call("mca_geglu_fwd", h.data_ptr(), out.data_ptr(), rows, ip, H.stream_ptr())
    hr = h.float().requires_grad_(True)
    a, gate = hr[:, :ip], hr[:, ip:]
    ref = torch.nn.functional.gelu(gate) * a
    assert rel(out.float(), ref) < 4e-3
    dg = bf(torch.randn(rows, ip, device="cuda", generator=g))
    ref.backward(dg.float())
    dh = torch.empty(rows, 2 * ip, device="cuda", dtype=torch.bfloat16)
    H.call("mca_geglu_bwd", dg.data_ptr(), h.data_ptr(), dh.data_ptr(), rows, ip, H.stream_ptr())
    assert rel(dh.float(), hr.grad) < 4e-3


@pytest.mark.parametrize("rows,ip,D", [(500, 384, 128), (4100, 384, 128), (4100, 384, 512), (2304, 448, 320), (2100, 1408, 512),
                                       (8200, 1408, 192), (8200, 1408, 512)])          # 363 tiles: a second tile per workgroup
def test_gemm_geglu_fwd_fused(H, rows, ip, D):
    """h = x @ W1^T (both halves, bf16) and g = a * gelu(gate) in one pass; the large-K cases run the persistent kernel
    (tile columns = 64 "a" + 64 "gate" rows of W1), 4100 rows end in a partial 256-row tile."""
    g = torch.Generator(device="cuda").manual_seed(43)
    x = bf(torch.randn(rows, D, device="cuda", generator=g))
    w1 = bf(torch.randn(2 * ip, D, device="cuda", generator=g) * (2.0 / D ** 0.5))
    h = torch.zeros(rows, 2 * ip, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(rows, ip, device="cuda", dtype=torch.bfloat16)
    H.call("mca_gemm_nt_geglu_fwd", x.data_ptr(), D, w1.data_ptr(), D, h.data_ptr(), 2 * ip, out.data_ptr(), ip, ip, rows, D, H.stream_ptr())
    href = x.float() @ w1.float().t()
    assert rel(h.float(), href) < 4e-3
    hb = h.float()          # g is computed from the ROUNDED h, as the unfused pair of kernels does
    assert rel(out.float(), torch.nn.functional.gelu(hb[:, ip:]) * hb[:, :ip]) < 4e-3


@pytest.mark.parametrize("rows,ip,D", [(500, 384, 128), (4100, 384, 128), (4100, 384, 512), (2304, 640, 320), (2100, 1408, 512),
                                       (41100, 384, 128), (41100, 384, 512)])          # >= 40,960 rows: the 256-row kernels
def test_gemm_geglu_bwd_fused(H, rows, ip, D):
    g = torch.Generator(device="cuda").manual_seed(41)
    h = bf(torch.randn(rows, 2 * ip, device="cuda", generator=g))
    dx = bf(torch.randn(rows, D, device="cuda", generator=g))
    w2T = bf(torch.randn(ip, D, device="cuda", generator=g) * 0.1)
    dh = torch.zeros(rows, 2 * ip, device="cuda", dtype=torch.bfloat16)
    H.call("mca_gemm_nt_geglu_bwd", dx.data_ptr(), D, w2T.data_ptr(), D, h.data_ptr(), dh.data_ptr(), 2 * ip, ip, rows, D, H.stream_ptr())
    hr = h.float().requires_grad_(True)
    out = torch.nn.functional.gelu(hr[:, ip:]) * hr[:, :ip]
    out.backward(dx.float() @ w2T.float().t())
    assert rel(dh.float(), hr.grad) < 4e-3


def test_cast_pad_multi(H):
    """One launch refreshing several bf16 weight copies (plain and transposed, zero-padded) = the per-tensor kernel."""
    g = torch.Generator(device="cuda").manual_seed(13)
    cases = [(100, 74, 128, 128, 0), (100, 74, 128, 128, 1), (1365, 512, 1408, 512, 0), (1365, 512, 512, 1408, 1), (512, 512, 512, 512, 1),
             (3, 5, 64, 64, 1)]
    srcs = [torch.randn(r, c, device="cuda", generator=g) for r, c, _, _, _ in cases]
    want, got, descs = [], [], (H.CastDesc * len(cases))()
    for i, ((r, c, rp, cp, tr), src) in enumerate(zip(cases, srcs)):
        w = torch.full((rp, cp), 3.0, device="cuda", dtype=torch.bfloat16)
        H.call("mca_cast_pad_bf16", src.data_ptr(), c, r, c, w.data_ptr(), cp, rp, cp, tr, H.stream_ptr())
        want.append(w)
        o = torch.full((rp, cp), 5.0, device="cuda", dtype=torch.bfloat16)
        got.append(o)
        d = descs[i]
        d.src, d.dst, d.lds, d.rows, d.cols, d.ldd, d.rows_pad, d.cols_pad, d.transpose = src.data_ptr(), o.data_ptr(), c, r, c, cp, rp, cp, tr
    dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).cuda()
    H.call("mca_cast_pad_bf16_multi", dev.data_ptr(), len(cases), H.stream_ptr())
    torch.cuda.synchronize()
    for i, (w, o) in enumerate(zip(want, got)):
        assert torch.equal(w, o), cases[i]


def test_pack_masks(H):
    """padding / row masks / presence bits of every modality in one launch (bool and int64 masks, a fully padded sample)."""
    g = torch.Generator(device="cuda").manual_seed(12)
    b, dims, F = 5, [70, 45, 30], 8
    N = sum(dims) + F
    masks = [torch.rand(b, n, device="cuda", generator=g) < 0.4 for n in dims]
    masks[1][2] = True                                      # modality 1 entirely padded in sample 2
    masks[2] = masks[2].to(torch.int64) * 7                 # int64 mask, any non-zero value = padded
    rowm = [torch.full((b * n,), 9, device="cuda", dtype=torch.uint8) for n in dims]
    pk = H.PackMasksArgs()
    off = 0
    for i, (mk, n) in enumerate(zip(masks, dims)):
        d = pk.m[i]
        d.mask, d.elem_bytes, d.n, d.offset = mk.data_ptr(), mk.element_size(), n, off
        d.rowmask = rowm[i].data_ptr() if i != 2 else None
        off += n
    pk.n_mod, pk.batch, pk.n_tokens, pk.n_fusion = 3, b, N, F
    padding = torch.full((b, N), 5, device="cuda", dtype=torch.uint8)
    present = torch.full((b,), -1, device="cuda", dtype=torch.int32)
    H.call("mca_pack_masks", C.byref(pk), padding.data_ptr(), present.data_ptr(), H.stream_ptr())
    want = torch.cat([(mk != 0) for mk in masks] + [torch.zeros(b, F, dtype=torch.bool, device="cuda")], 1).to(torch.uint8)
    assert torch.equal(padding, want)
    assert torch.equal(rowm[0], (masks[0] != 0).reshape(-1).to(torch.uint8)) and torch.equal(rowm[1], masks[1].reshape(-1).to(torch.uint8))
    assert (rowm[2] == 9).all()                             # NULL row mask: untouched
    bits = sum(((mk == 0).any(1).to(torch.int32) << i) for i, mk in enumerate(masks))
    assert torch.equal(present, bits) and int(present[2]) & 2 == 0


# ------------------------------------------------------------------------------------- data movement
def test_cast_bcast_reduce(H):
    g = torch.Generator(device="cuda").manual_seed(5)
    src = torch.randn(100, 74, device="cuda", generator=g)
    dst = torch.full((128, 128), 3.0, device="cuda", dtype=torch.bfloat16)
    H.call("mca_cast_pad_bf16", src.data_ptr(), 74, 100, 74, dst.data_ptr(), 128, 128, 128, 0, H.stream_ptr())
    assert torch.equal(dst[:100, :74], bf(src)) and (dst[100:] == 0).all() and (dst[:, 74:] == 0).all()
    dstT = torch.full((128, 128), 3.0, device="cuda", dtype=torch.bfloat16)
    H.call("mca_cast_pad_bf16", src.data_ptr(), 74, 100, 74, dstT.data_ptr(), 128, 128, 128, 1, H.stream_ptr())
    assert torch.equal(dstT[:74, :100], bf(src).t()) and (dstT[74:] == 0).all() and (dstT[:, 100:] == 0).all()
    s2 = torch.randn(50, 512, device="cuda", generator=g)
    d2 = torch.zeros(50, 1536, device="cuda", dtype=torch.bfloat16)
    H.call("mca_f32_to_bf16", s2.data_ptr(), 512, d2.data_ptr() + 512 * 2, 1536, 50, 512, 0.5, H.stream_ptr())
    assert torch.equal(d2[:, 512:1024], bf(s2 * 0.5)) and (d2[:, :512] == 0).all()
    # broadcast 8 learned rows into a (b=3, N=20, D=64) buffer at row offset 12
    toks = torch.randn(8, 64, device="cuda", generator=g)
    buf = torch.zeros(3, 20, 64, device="cuda")
    H.call("mca_bcast_rows", toks.data_ptr(), 64, buf.data_ptr() + 12 * 64 * 4, 64, 20 * 64, 8, 24, 64, H.stream_ptr())
    assert torch.equal(buf[:, 12:], toks[None].expand(3, -1, -1)) and (buf[:, :12] == 0).all()
    # reduce back
    gsrc = torch.randn(3, 20, 64, device="cuda", generator=g)
    acc = torch.ones(8, 64, device="cuda")
    H.call("mca_reduce_rows", gsrc.data_ptr() + 12 * 64 * 4, 64, 20 * 64, 8, acc.data_ptr(), 64, 24, 64, H.stream_ptr())
    assert rel(acc, 1 + gsrc[:, 12:].sum(0)) < 1e-6
    cs = torch.zeros(64, device="cuda")
    flat = gsrc.reshape(60, 64)
    H.call("mca_reduce_rows", flat.data_ptr(), 64, 64, 1, cs.data_ptr(), 64, 60, 64, H.stream_ptr())
    assert rel(cs, flat.sum(0)) < 1e-6


# ------------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("variant,world", [("mca", 1), ("bimodal", 1), ("zorro", 1), ("mca", 2), ("bimodal", 4)])
def test_contrastive_loss(H, variant, world):
    from oracle import mca_oracle as O
    from util_small import small_config
    S = importlib.import_module("mca-paper_amd.structure")
    cfg = small_config(variant)
    OS = O.Structure(cfg)
    st = S.FusionStructure([70, 45, 30], 8, (3, 2), fcl=cfg["fcl"], zorro=cfg["zorro"])
    names = list(cfg["encoder_configs"].keys())
    terms = S.loss_terms(names, st, cfg["bimodal_contrastive"], cfg["non_fusion_fcl"])
    assert [t.name for t in terms] == [t[0] for t in O.loss_schedule(OS)]
    b, R, D = 6, st.n_return, 128
    B = b * world
    g = torch.Generator().manual_seed(21)
    pooled_all = torch.randn(B, R, D, generator=g) * 0.3
    pooled_dev, = (pooled_all.cuda(),)
    present = torch.randint(0, 8, (B,), generator=g)
    present[present == 0] = 5
    present[:b] &= 0b101                                   # rank 0 never sees modality 1 -> NaN terms there
    logit = torch.tensor(math.log(1 / 0.07))
    arr = (H.LossTerm * len(terms))()
    for i, t in enumerate(terms):
        arr[i] = H.LossTerm(t.slot_a, t.slot_b, t.and_bits, t.or_bits)
    terms_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    ws = torch.empty(H.lib().mca_contrastive_workspace_bytes(B, len(terms)), dtype=torch.uint8, device="cuda")
    total_ref_grad = torch.zeros(B, R, D, dtype=torch.float64)
    pa = pooled_all.double().requires_grad_(True)
    results = []
    for r in range(world):
        lg = logit.double().clone().requires_grad_(True)
        sm = {n: ((present[r * b:(r + 1) * b] >> i) & 1).bool() for i, n in enumerate(names)}
        out = O.pretraining_loss(OS, pa[r * b:(r + 1) * b], sm, lg, pooled_all=pa, rank=r)
        gp, gl = torch.autograd.grad(out["loss"], [pa, lg])
        total_ref_grad += gp
        results.append((out, gl))
    for r in range(world):
        tl = torch.empty(len(terms), device="cuda"); ls = torch.empty(1, device="cuda")
        dp = torch.empty(b, R, D, device="cuda"); dl = torch.empty(1, device="cuda")
        present_dev, logit_dev = present.to(torch.int32).cuda(), logit.cuda()
        H.call("mca_contrastive_fwd_bwd", pooled_dev.data_ptr(), present_dev.data_ptr(), terms_dev.data_ptr(),
               len(terms), logit_dev.data_ptr(), B, b, r * b, R, D, tl.data_ptr(), ls.data_ptr(), dp.data_ptr(), dl.data_ptr(),
               ws.data_ptr(), H.stream_ptr())
        torch.cuda.synchronize()
        out, gl = results[r]
        ref_terms = torch.stack([out["losses"][t.name] for t in terms]).float()
        assert torch.equal(torch.isnan(tl.cpu()), torch.isnan(ref_terms))
        ok = ~torch.isnan(ref_terms)
        assert rel(tl.cpu()[ok], ref_terms[ok]) < 1e-5
        assert abs(float(ls) - float(out["loss"])) < 1e-5 * abs(float(out["loss"]))
        assert rel(dp.cpu(), total_ref_grad[r * b:(r + 1) * b]) < 1e-4
        assert abs(float(dl) - float(gl)) < 1e-4 * max(1.0, abs(float(gl)))
    if variant == "mca" and world == 1:
        assert torch.isnan(ref_terms).any()


# --------------------------------------------------------------------------------------------- optimizer
def test_clip_adamw(H):
    g = torch.Generator().manual_seed(31)
    n = 100003
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p_ref], lr=1e-3)
    p = p0.cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (3.0 if step == 1 else 0.001)
        p_ref.grad = grad.clone()
        gn = torch.nn.utils.clip_grad_norm_([p_ref], 2.0)
        opt.step()
        gd = grad.cuda()
        sq = torch.full((1026,), 7.0, device="cuda")          # MCA_SQNORM_WORDS: sum g^2, the caller-owned scratch, the norm
        H.call("mca_grad_sqnorm", gd.data_ptr(), n, sq.data_ptr(), H.stream_ptr())
        assert abs(float(sq[0].sqrt()) - float(gn)) < 1e-4 * float(gn) and float(sq[1025]) == float(sq[0].sqrt())
        H.call("mca_adamw_step", p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 0.01,
               1 - 0.9 ** step, 1 - 0.999 ** step, 2.0, sq.data_ptr(), None, None, H.stream_ptr())
        err = (p.cpu() - p_ref.detach()).abs().max()
        assert err < 5e-6, f"step {step}: {err}"

"""TEST INFRASTRUCTURE (see oracle/__init__.py) — torch-CPU restatement of the MCA/MMA hot path.

Functional restatement of the reference's forward for the path SURVEY.md §8a names; gradients come from
torch autograd over this restatement, the optimizer step from the same torch primitives the reference
calls (``clip_grad_norm_`` + ``AdamW``, train_accel_gpu.py:80,116-118).  Every function cites the
reference lines it follows.  The state dict uses the reference's own key names.

Precision modes
  * ``fp32`` / ``fp64``: the reference's arithmetic in that dtype.
  * ``bf16emu``: fp32 arithmetic with round-to-nearest-even bf16 rounding inserted at exactly the points
    where the HIP path stores or feeds bf16 (GEMM operands, q/k/v, softmax probabilities fed to PV,
    attention output, FF hidden).  Used to check the kernels tightly (accumulation order is then the
    only difference).

Pinned: tests/test_oracle_golden.py compares this file with tests/golden/*.pt, which
oracle/make_goldens.py produced by running the reference itself.
"""
from __future__ import annotations

import math
from itertools import chain, combinations
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

FUSION_TOKEN = -1   # model.py:310
GLOBAL_TOKEN = -2   # model.py:311


# --------------------------------------------------------------------------------------------------
# precision helpers
# --------------------------------------------------------------------------------------------------
class Prec:
    def __init__(self, mode: str = "fp32"):
        assert mode in ("fp32", "fp64", "bf16emu", "fp8emu")
        # fp8emu = bf16emu + MX-fp8 (e4m3, power-of-two scale per 32 elements) operands of Q K^T and P V in the SELF-attention
        # layers, as mca_attn_quant_mxfp8 / mca_attn_fwd_fp8 compute them, and the fp8 score recomputes of their backward
        # (BASELINE configs[4]; _Fp8AttentionCore restates both directions, nothing is straight-through there)
        self.fp8 = mode == "fp8emu"
        if self.fp8:
            mode = "bf16emu"
        self.mode = mode
        self.dtype = torch.float64 if mode == "fp64" else torch.float32

    def r(self, t: torch.Tensor) -> torch.Tensor:
        """bf16 rounding point (identity unless bf16emu). Straight-through for autograd."""
        if self.mode != "bf16emu":
            return t
        return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())

    def rg(self, t: torch.Tensor) -> torch.Tensor:
        """rounding point that also rounds the incoming gradient (a tensor the HIP path keeps in bf16
        in both directions)."""
        if self.mode != "bf16emu":
            return t
        return _RoundBoth.apply(t)


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def mx_e4m3(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """OCP MX-fp8 round trip along ``dim`` (length a multiple of 32): per 32-element block one power-of-two scale
    2^(floor(log2(amax)) - 7) and e4m3 elements (|x| / scale < 256: no saturation), as mca_attn_quant_mxfp8 does.
    Returns the de-quantised values; straight-through gradient."""
    xd = x.detach().movedim(dim, -1)
    shp = xd.shape
    xb = xd.reshape(*shp[:-1], shp[-1] // 32, 32).to(torch.float32)
    am = xb.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(am.clamp_min(2.0 ** -126))) - 7.0
    e = torch.where(am > 0, e.clamp_min(-127.0), torch.full_like(e, -127.0))
    sc = torch.exp2(e)
    q = (xb / sc).to(torch.float8_e4m3fn).to(torch.float32) * sc
    q = q.reshape(shp).movedim(-1, dim).to(x.dtype)
    return x + (q - x.detach())


def _mx(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """mx_e4m3 without the straight-through identity (plain values)"""
    return mx_e4m3(x.detach(), dim)


def _bf(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(t.dtype)


class _Fp8AttentionCore(torch.autograd.Function):
    """The MX-fp8 self-attention of BASELINE configs[4] with the arithmetic of BOTH directions of the HIP path
    (mca-paper_amd/csrc/attention_fp8.hip), not a straight-through estimate:

    forward  (mca_attn_quant_mxfp8 + mca_attn_fwd_fp8): S = MX(q2) . MX(k)^T in the log2 domain (blocks of 32 along d),
             m = row maximum over the allowed keys, P = 2^(S - m), l = sum of the UNROUNDED P, O = e4m3(128 P) . MX_keys(V) / 128 l
             (V quantised in blocks of 32 consecutive keys); a row with every key blocked is uniform over all n keys (mean of V);
             lse = m + log2 l.
    backward (mca_attn_bwd_prep + mca_attn_quant_bwd_mxfp8 + mca_attn_bwd_dq_fp8 / mca_attn_bwd_dkv_fp8): delta = rowsum(dO o
             bf16(O)); S recomputed from the SAME MX(q2), MX(k); P = 2^(S - lse) (0 on blocked pairs and on uniform rows);
             dP = MX(dO) . MX_d(V)^T with BOTH operands quantised along d; dS = P o (dP - delta); P and dS rounded to bf16 feed the
             three gradient products with bf16 operands: dV = P^T dO (+ mean of dO over the uniform rows / n), dK = ln2 dS^T q2,
             dq2 = ln2 dS k   (q2 = q . scale . log2 e: the log2-domain query the kernels see)."""

    @staticmethod
    def forward(ctx, q2, k, v, blocked):
        b, h, n, d = q2.shape
        n_pad = (n + 31) // 32 * 32
        q8, k8 = _mx(q2), _mx(k)
        v8 = _mx(torch.nn.functional.pad(v.detach(), (0, 0, 0, n_pad - n)), 2)[:, :, :n, :]
        s2 = torch.einsum("bhid,bhjd->bhij", q8, k8).masked_fill(blocked, float("-inf"))
        m = s2.max(dim=-1, keepdim=True).values
        uniform = torch.isinf(m) & (m < 0)
        e = torch.exp2(s2 - torch.where(uniform, torch.zeros_like(m), m))
        l = e.sum(-1, keepdim=True)
        p8 = (128.0 * e).to(torch.float8_e4m3fn).to(e.dtype)
        out = torch.einsum("bhij,bhjd->bhid", p8, v8) / (128.0 * l.clamp_min(1e-30))
        out = torch.where(uniform, v.detach().mean(dim=2, keepdim=True).expand_as(out), out)
        lse = torch.where(uniform, torch.full_like(m, float("inf")), m + torch.log2(l.clamp_min(1e-30)))
        ctx.save_for_backward(q2.detach(), k.detach(), v.detach(), blocked, out, lse)
        return out

    @staticmethod
    def backward(ctx, d_out):
        q2, k, v, blocked, out, lse = ctx.saved_tensors
        return (*fp8_attention_backward(q2, k, v, blocked, out, lse, d_out), None)


def fp8_attention_backward(q2, k, v, blocked, out, lse, d_out):
    """(dq2, dk, dv) of _Fp8AttentionCore from the saved forward results ``out`` (b,h,n,64) and ``lse`` (b,h,n,1; +inf marks a
    uniform row): the arithmetic of mca_attn_bwd_prep + mca_attn_quant_bwd_mxfp8 + mca_attn_bwd_dq_fp8 / mca_attn_bwd_dkv_fp8
    (see the class docstring).  A kernel test may pass the kernel's own forward results to isolate the backward."""
    n = q2.shape[2]
    d_o = _bf(d_out)
    delta = (d_o * _bf(out)).sum(-1, keepdim=True)
    s2 = torch.einsum("bhid,bhjd->bhij", _mx(q2), _mx(k))
    p = torch.exp2(s2 - lse).masked_fill(blocked, 0.0)          # lse = +inf on uniform rows: P = 0
    dp = torch.einsum("bhid,bhjd->bhij", _mx(d_o), _mx(v))
    ds = p * (dp - delta)
    pb, dsb = _bf(p), _bf(ds)
    uniform = torch.isinf(lse)
    dvmean = (d_o * uniform).sum(dim=2, keepdim=True) / n
    dv = torch.einsum("bhij,bhid->bhjd", pb, d_o) + dvmean
    ln2 = 0.6931471805599453
    dk = ln2 * torch.einsum("bhij,bhid->bhjd", dsb, _bf(q2))
    dq2 = ln2 * torch.einsum("bhij,bhjd->bhid", dsb, _bf(k))
    return dq2, dk, dv


def fp8_attention_core(q2, k, v, blocked):
    """q2 (b,h,n,64) already in the log2 domain (scale * log2 e folded in), k, v (b,h,n,64), blocked (b,1|h,n,n) bool."""
    return _Fp8AttentionCore.apply(q2, k, v, blocked)


# --------------------------------------------------------------------------------------------------
# static structure: token types, masks, combos           (model.py:11-12, 312-327, 383-446)
# --------------------------------------------------------------------------------------------------
def adjusted_powerset(items, powers):
    """model.py:11-12."""
    return list(chain.from_iterable(combinations(items, r) for r in powers))


def fusion_combos_of(n_modalities: int, powers) -> List[frozenset]:
    """model.py:312."""
    return [frozenset(x) for x in adjusted_powerset(list(range(n_modalities)), powers)]


def return_token_types_of(n_modalities, fusion_combos, fcl, zorro, no_fusion) -> List[int]:
    """model.py:313-325."""
    if no_fusion:
        return list(range(n_modalities)) + [GLOBAL_TOKEN]
    if (not fcl) or zorro:
        return list(range(n_modalities)) + [FUSION_TOKEN, GLOBAL_TOKEN]
    return list(range(n_modalities)) + [FUSION_TOKEN] * len(fusion_combos) + [GLOBAL_TOKEN]


def token_types_of(token_dims, num_fusion_tokens) -> torch.Tensor:
    """model.py:383-390."""
    out = []
    for i, n in enumerate(token_dims):
        out += [i] * n
    out += [FUSION_TOKEN] * num_fusion_tokens
    return torch.tensor(out, dtype=torch.long)


def zorro_mask(token_types, no_fusion) -> torch.Tensor:
    """model.py:392-398. True = blocked."""
    frm = token_types[:, None]
    to = token_types[None, :]
    allowed = frm == to
    if not no_fusion:
        allowed = allowed | (frm == FUSION_TOKEN)
    return ~allowed


def zorro_pool_mask(token_types, ret_types) -> torch.Tensor:
    """model.py:400-406."""
    allowed = ret_types[:, None] == token_types[None, :]
    allowed = allowed | (ret_types[:, None] == GLOBAL_TOKEN)
    return ~allowed


def mca_mask(token_types, combos, zmask) -> torch.Tensor:
    """model.py:408-430: fusion sub-block c attends modality tokens of combo c plus itself."""
    zmask = zmask.clone()
    fus = token_types == FUSION_TOKEN
    nf = int(fus.sum())
    assert nf % len(combos) == 0
    nsub = nf // len(combos)
    fus_idx = fus.nonzero().flatten()
    rows = []
    for c, combo in enumerate(combos):
        blocked = ~torch.isin(token_types, torch.tensor(sorted(combo), dtype=token_types.dtype))
        blocked[fus] = True
        blocked[fus_idx[c * nsub:(c + 1) * nsub]] = False
        rows += [blocked] * nsub
    zmask[fus] = torch.stack(rows)
    return zmask


def mca_pool_mask(token_types, combos, ret_types, num_fusion_tokens, pmask) -> torch.Tensor:
    """model.py:432-446: fusion return-token c attends only fusion sub-block c."""
    pmask = pmask.clone()
    nsub = num_fusion_tokens // len(combos)
    blocks = torch.block_diag(*[torch.ones((1, nsub)) for _ in combos]).to(torch.bool)
    sel = (ret_types == FUSION_TOKEN)[:, None] & (token_types == FUSION_TOKEN)[None, :]
    pmask[sel] = ~blocks.flatten()
    return pmask


class Structure:
    """Everything MCA.__init__ derives from the config (model.py:283-380)."""

    def __init__(self, cfg: dict):
        enc = cfg["encoder_configs"]
        self.modalities = list(enc.keys())
        M = len(self.modalities)
        self.fcl = bool(cfg.get("fcl", False))
        self.zorro = bool(cfg.get("zorro", False))
        self.no_fusion = bool(cfg.get("no_fusion", False))
        self.bimodal = bool(cfg.get("bimodal_contrastive", False))
        self.non_fusion_fcl = bool(cfg.get("non_fusion_fcl", False))
        self.combos = fusion_combos_of(M, cfg.get("fusion_combos", [4, 5]))
        self.num_fusion_tokens = 0 if self.no_fusion else int(cfg.get("num_fusion_tokens", 16))
        self.ret_types = return_token_types_of(M, self.combos, self.fcl, self.zorro, self.no_fusion)
        self.token_dims = [enc[m]["max_tokens"] for m in self.modalities]
        self.token_types = token_types_of(self.token_dims, self.num_fusion_tokens)
        rt = torch.tensor(self.ret_types)
        am = zorro_mask(self.token_types, self.no_fusion)
        pm = zorro_pool_mask(self.token_types, rt)
        if not self.zorro:
            am = mca_mask(self.token_types, self.combos, am)
            if self.fcl:
                pm = mca_pool_mask(self.token_types, self.combos, rt, self.num_fusion_tokens, pm)
        self.attn_mask, self.pool_mask = am, pm
        self.heads = int(cfg.get("heads", 8))
        self.dim_head = int(cfg.get("dim_head", 64))
        self.depth = int(cfg["depth"])
        self.dim = int(cfg["dim"])
        self.do_fcl = self.fcl and not self.zorro          # model.py:375
        self.enc_cfg = enc


# --------------------------------------------------------------------------------------------------
# encoders                                                                (encoders.py:17-214)
# --------------------------------------------------------------------------------------------------
def sinusoid_pe(max_len: int, d_model: int, dtype=torch.float32) -> torch.Tensor:
    """encoders.py:128-135 (computed in fp32 like the reference buffer, then cast)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


def embedded_sequence_encoder(sd, pfx, batch_m, P: Prec):
    """encoders.py:196-214.  LN(in) -> Linear(in,D)+b -> LN(D), pad rows zeroed before and after, + PE
    (PE is added at padded positions too)."""
    tokens = batch_m["tokens"].to(P.dtype)
    pad = batch_m["attention_mask"].to(torch.bool)
    if not torch.isfinite(tokens).all():
        raise Exception("Tokens are not finite")                       # encoders.py:197-198
    x = tokens.masked_fill(pad[..., None], 0.0)
    w0, b0 = sd[pfx + "token_encoder.0.weight"].to(P.dtype), sd[pfx + "token_encoder.0.bias"].to(P.dtype)
    w1, b1 = sd[pfx + "token_encoder.1.weight"].to(P.dtype), sd[pfx + "token_encoder.1.bias"].to(P.dtype)
    w2, b2 = sd[pfx + "token_encoder.2.weight"].to(P.dtype), sd[pfx + "token_encoder.2.bias"].to(P.dtype)
    x = F.layer_norm(x, x.shape[-1:], w0, b0)
    x = F.linear(P.r(x), P.r(w1)) + b1
    x = F.layer_norm(x, x.shape[-1:], w2, b2)
    x = x.masked_fill(pad[..., None], 0.0)
    if not torch.isfinite(x).all():
        raise Exception("Encoder transform resulted in non-finite values")   # encoders.py:206-207
    pe = sd[pfx + "positional_encoder.pe"].to(P.dtype)
    x = x + pe[: x.shape[1]][None]
    return x, batch_m["attention_mask"]


def embedding_renorm_(weight: torch.Tensor, max_norm: float = 1.0):
    """nn.Embedding(max_norm=1.0) side effect (encoders.py:31-33): rows with L2 norm > max_norm are
    rescaled IN PLACE by max_norm/(norm+1e-7) on every forward, outside autograd."""
    with torch.no_grad():
        n = weight.norm(2, dim=1, keepdim=True)
        scale = torch.where(n > max_norm, max_norm / (n + 1e-7), torch.ones_like(n))
        weight.mul_(scale)


def tabular_encoder(sd, pfx, batch_m, P: Prec, cfg_m: dict):
    """encoders.py:90-96 with TokenEncoder :35-37 and ContinuousValueEncoder :55-72.
    Padding for the VALUE path is ``x == padding_idx`` (= -1 by default, encoders.py:80,88), not the
    collator's -10000; values are clamped to max_value first."""
    values = batch_m["values"].to(P.dtype)
    E = sd[pfx + "token_encoder.embedding.weight"]
    embedding_renorm_(E.data if E.requires_grad else E, 1.0)
    E = E.to(P.dtype)
    # padding_idx = -1 -> row n-1 never receives a gradient (nn.Embedding semantics, encoders.py:80,87)
    E = torch.cat([E[:-1], E[-1:].detach()], dim=0)
    padding_value = cfg_m.get("padding_idx", -1)
    max_value = cfg_m.get("max_value", 10000)
    x = values.unsqueeze(-1)
    pad = x == padding_value
    x = torch.clamp(x, max=max_value)
    w1, b1 = sd[pfx + "value_encoder.linear1.weight"].to(P.dtype), sd[pfx + "value_encoder.linear1.bias"].to(P.dtype)
    w2, b2 = sd[pfx + "value_encoder.linear2.weight"].to(P.dtype), sd[pfx + "value_encoder.linear2.bias"].to(P.dtype)
    g, be = sd[pfx + "value_encoder.norm.weight"].to(P.dtype), sd[pfx + "value_encoder.norm.bias"].to(P.dtype)
    h = torch.relu(x * w1[:, 0] + b1)                      # Linear(1, D)
    h = F.linear(P.r(h), P.r(w2)) + b2
    h = F.layer_norm(h, h.shape[-1:], g, be)
    h = h.masked_fill(pad, 0.0)
    return E[None] + h, batch_m["attention_mask"]


# --------------------------------------------------------------------------------------------------
# fusion transformer                                                        (model.py:24-122)
# --------------------------------------------------------------------------------------------------
def layer_norm(x, gamma):
    """model.py:24-31: F.layer_norm with learnable gamma and a zero beta buffer, eps 1e-5."""
    return F.layer_norm(x, x.shape[-1:], gamma, torch.zeros_like(gamma))


def attention(x, ctx, wq, wkv, wo, attn_mask, kpm, heads, P: Prec):
    """model.py:73-105.  masked_fill(-finfo.max) twice then softmax: a row whose keys are all masked
    becomes UNIFORM over all N keys (SURVEY.md §7 hard part 1)."""
    b, n, _ = x.shape
    kv_x = x if ctx is None else ctx
    q = P.rg(F.linear(P.r(x), P.r(wq)))
    kv = P.rg(F.linear(P.r(kv_x), P.r(wkv)))
    k, v = kv.chunk(2, dim=-1)
    dh = q.shape[-1] // heads
    sp = lambda t: t.reshape(t.shape[0], t.shape[1], heads, dh).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    q = q * dh ** -0.5
    sim = torch.einsum("bhid,bhjd->bhij", q, k)
    neg = -torch.finfo(sim.dtype).max
    if attn_mask is not None:
        sim = sim.masked_fill(attn_mask, neg)
    if kpm is not None:
        sim = sim.masked_fill(kpm[:, None, None, :], neg)
    attn = sim.softmax(dim=-1)
    if P.fp8 and ctx is None:
        blocked = torch.zeros(b, 1, n, n, dtype=torch.bool, device=x.device)
        if attn_mask is not None:
            blocked = blocked | attn_mask[None, None]
        if kpm is not None:
            blocked = blocked | kpm[:, None, None, :]
        out = fp8_attention_core(P.r(q * 1.4426950408889634), k, v, blocked)
    elif P.mode == "bf16emu":
        # the HIP path feeds P (un-normalised) to the PV MFMA in bf16 and divides by the fp32 row sum
        m = sim.max(dim=-1, keepdim=True).values
        e = torch.exp(sim - m)
        l = e.sum(-1, keepdim=True)
        out = torch.einsum("bhij,bhjd->bhid", P.r(e), v) / l
    else:
        out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, heads * dh)
    return F.linear(P.rg(out), P.r(wo))


def feed_forward(x, w1, w2, P: Prec):
    """model.py:35-54: Linear(D,2I) -> chunk (x, gate) -> gelu(gate)*x (exact erf GELU) -> Linear(I,D)."""
    h = P.rg(F.linear(P.r(x), P.r(w1)))
    a, gate = h.chunk(2, dim=-1)
    g = P.rg(F.gelu(gate) * a)
    return F.linear(g, P.r(w2))


def mca_layer(x, sd, pfx, attn_mask, kpm, heads, P: Prec):
    """model.py:117-122: ONE LayerNorm used twice; the residual is the NORMED tensor."""
    gamma = sd[pfx + "norm.gamma"].to(P.dtype)
    x = layer_norm(x, gamma)
    x = attention(x, None, sd[pfx + "attn.to_q.weight"].to(P.dtype), sd[pfx + "attn.to_kv.weight"].to(P.dtype),
                  sd[pfx + "attn.to_out.weight"].to(P.dtype), attn_mask, kpm, heads, P) + x
    x = layer_norm(x, gamma)
    x = feed_forward(x, sd[pfx + "ff.feedforward.0.weight"].to(P.dtype),
                     sd[pfx + "ff.feedforward.2.weight"].to(P.dtype), P) + x
    return x


# --------------------------------------------------------------------------------------------------
# contrastive loss (third-party torchmultimodal; formula per utils/contrastive_loss_with_temperature.py
# :40-108,178-195) and the pair schedule of MCAPretrainingLoss (model.py:132-233)
# --------------------------------------------------------------------------------------------------
LOGIT_SCALE_MIN, LOGIT_SCALE_MAX = math.log(1.0), math.log(100.0)


def contrastive_loss(a, b, logit_scale, mask=None, a_all=None, b_all=None, rank=0):
    """CE(a·b_allᵀ·e^s) symmetrised; rows selected by ``mask``; empty selection -> NaN (mean over 0 rows).
    ``a_all``/``b_all`` are the embeddings of all ranks concatenated along dim 0 (rank-major); labels are
    ``rank*b + arange(b)`` (utils/contrastive_loss_with_temperature.py:28-31)."""
    a_all = a if a_all is None else a_all
    b_all = b if b_all is None else b_all
    T = torch.exp(logit_scale)
    la = a @ b_all.t() * T
    lb = b @ a_all.t() * T
    labels = rank * a.shape[0] + torch.arange(a.shape[0])
    if mask is not None:
        la, lb, labels = la[mask], lb[mask], labels[mask]
    return (F.cross_entropy(la, labels) + F.cross_entropy(lb, labels)) / 2


def loss_schedule(S: Structure):
    """The ordered list of loss terms MCAPretrainingLoss evaluates (model.py:160-168,198-220).
    Each entry: (name, key_a, key_b, and_mods, or_mods) where the row mask is
    AND_{m in and_mods} sample_mask[m]  &  (OR_{m in or_mods} sample_mask[m] if or_mods else True)."""
    names = S.modalities
    if S.no_fusion:
        pairs = list(combinations(names, 2))
    elif S.bimodal:
        pairs = list(combinations(names + ["fusion"], 2))
    else:
        pairs = [(m, "fusion") for m in names]
    # model.py:167 keys the dict by frozenset(pair); iteration order of the two members follows the
    # frozenset, which the reference unpacks as (moda, modb).  The loss is symmetric in (a, b), and the
    # name sorts the pair, so the order does not matter.
    terms = []
    for pa, pb in pairs:
        if pa == "fusion":
            am = [pb]
        elif pb == "fusion":
            am = [pa]
        else:
            am = [pa, pb]
        terms.append(("_".join(sorted((pa, pb))), pa, pb, am, []))
    if S.do_fcl:
        root = S.combos[0]                                          # model.py:151
        for k in S.combos:
            if k == root:
                continue
            kn = "_".join(sorted(names[i] for i in k))
            orm = [names[i] for i in k]
            if not S.no_fusion:
                terms.append((f"fcl_fusion|{kn}", "fusion", k, [], orm))
            if S.non_fusion_fcl:
                for mod in names:
                    terms.append((f"fcl_{mod}|{kn}", mod, k, [mod], orm))
    return terms


def pretraining_loss(S: Structure, pooled, sample_mask, logit_scale, no_loss=False,
                     pooled_all=None, rank=0):
    """model.py:175-233."""
    names = S.modalities
    out = {m: pooled[:, i] for i, m in enumerate(names)}
    slot = {m: i for i, m in enumerate(names)}
    if S.do_fcl:
        for i, k in enumerate(S.combos):
            out[k] = pooled[:, i + len(names)]
            slot[k] = i + len(names)
        if not S.no_fusion:
            out["fusion"] = out[S.combos[0]]
            slot["fusion"] = slot[S.combos[0]]
    elif not S.no_fusion:
        out["fusion"] = pooled[:, len(names)]
        slot["fusion"] = len(names)
    if no_loss:
        return out
    with torch.no_grad():                                            # ...:187 in-place clamp of the parameter
        logit_scale.clamp_(LOGIT_SCALE_MIN, LOGIT_SCALE_MAX)
    out["losses"] = {}
    for name, ka, kb, and_mods, or_mods in loss_schedule(S):
        mask = torch.ones(pooled.shape[0], dtype=torch.bool)
        for m in and_mods:
            mask = mask & sample_mask[m]
        if or_mods:
            o = torch.zeros_like(mask)
            for m in or_mods:
                o = o | sample_mask[m]
            mask = mask & o
        a_all = None if pooled_all is None else pooled_all[:, slot[ka]]
        b_all = None if pooled_all is None else pooled_all[:, slot[kb]]
        out["losses"][name] = contrastive_loss(out[ka], out[kb], logit_scale, mask, a_all, b_all, rank)
    if S.do_fcl:
        out["fcl_loss"] = torch.stack([torch.nan_to_num(v) for k, v in out["losses"].items() if "fcl" in k]).mean()
        out["no-fcl_loss"] = torch.stack([torch.nan_to_num(v) for k, v in out["losses"].items() if "fcl" not in k]).mean()
    vals = list(out["losses"].values())
    nl = sum(int(not torch.isnan(v)) for v in vals)
    tot = sum(torch.nan_to_num(v) for v in vals)
    out["loss"] = tot if nl == 0 else tot / float(nl)
    return out


# --------------------------------------------------------------------------------------------------
# whole forward                                                              (model.py:448-478)
# --------------------------------------------------------------------------------------------------
def encode_and_pack(S: Structure, sd, batch, P: Prec):
    """model.py:455-466."""
    toks, masks = [], []
    for m in S.modalities:
        c = S.enc_cfg[m]
        pfx = f"encoders.{m}."
        if c["type"] == "EmbeddedSequenceEncoder":
            t, a = embedded_sequence_encoder(sd, pfx, batch[m], P)
        elif c["type"] == "TabularEncoder":
            t, a = tabular_encoder(sd, pfx, batch[m], P, c)
        else:
            raise NotImplementedError(c["type"])
        toks.append(t)
        masks.append(a)
    sample_mask = {m: ((a == 0).sum(dim=1) != 0) for m, a in zip(S.modalities, masks)}
    b = toks[0].shape[0]
    if not S.no_fusion:
        toks.append(sd["fusion_tokens"].to(P.dtype)[None].expand(b, -1, -1))
        masks.append(torch.zeros(b, S.num_fusion_tokens, dtype=torch.bool))
    tokens = torch.cat(toks, dim=1)
    padding = torch.cat([a.to(torch.bool) for a in masks], dim=1)
    return tokens, padding, sample_mask


def mca_trunk(S: Structure, sd, tokens, padding, P: Prec):
    """model.py:468-473: layers, final norm, attentive pooling (+ return-token residual)."""
    for i in range(S.depth):
        tokens = mca_layer(tokens, sd, f"layers.{i}.", S.attn_mask, padding, S.heads, P)
    tokens = layer_norm(tokens, sd["norm.gamma"].to(P.dtype))
    b = tokens.shape[0]
    rt = sd["return_tokens"].to(P.dtype)[None].expand(b, -1, -1)
    pooled = attention(rt, tokens, sd["attn_pool.to_q.weight"].to(P.dtype), sd["attn_pool.to_kv.weight"].to(P.dtype),
                       sd["attn_pool.to_out.weight"].to(P.dtype), S.pool_mask, padding, S.heads, P) + rt
    return pooled


def mca_forward(S: Structure, sd: Dict[str, torch.Tensor], batch, mode="fp32", no_loss=False,
                gather=None, rank=0):
    """Full MCA.forward.  ``gather`` (optional) maps the local pooled block to the all-rank block
    (autograd-aware), standing in for gather_tensor (utils/distributed.py:23-56)."""
    P = Prec(mode)
    tokens, padding, sample_mask = encode_and_pack(S, sd, batch, P)
    pooled = mca_trunk(S, sd, tokens, padding, P)
    pooled_all = None if gather is None else gather(pooled)
    out = pretraining_loss(S, pooled, sample_mask, sd["loss.loss_fn.logit_scale"], no_loss, pooled_all, rank)
    out["modality_sample_mask"] = sample_mask
    out["pooled"] = pooled
    return out


# --------------------------------------------------------------------------------------------------
# EAO baseline                                                              (model.py:481-596, 235-280)
# --------------------------------------------------------------------------------------------------
class EAOStructure(Structure):
    """EAO.__init__ (model.py:482-538): no fusion tokens, the loss built with the default no_fusion=True and the first
    combination as the (unused) root; everything else as MCA."""

    def __init__(self, cfg: dict):
        enc = cfg["encoder_configs"]
        self.modalities = list(enc.keys())
        M = len(self.modalities)
        self.fcl = bool(cfg.get("fcl", False))
        self.zorro = bool(cfg.get("zorro", False))
        self.no_fusion = True                                            # MCAPretrainingLoss(no_fusion=True), model.py:538
        self.bimodal = bool(cfg.get("bimodal_contrastive", False))
        self.non_fusion_fcl = bool(cfg.get("non_fusion_fcl", False))
        self.combos = fusion_combos_of(M, cfg.get("fusion_combos", [4, 5]))
        self.num_fusion_tokens = 0                                       # model.py:501
        self.token_dims = [enc[m]["max_tokens"] for m in self.modalities]
        self.token_types = token_types_of(self.token_dims, 0)
        self.attn_mask = self.pool_mask = None
        self.heads = int(cfg.get("heads", 8))
        self.dim_head = int(cfg.get("dim_head", 64))
        self.depth = int(cfg["depth"])
        self.dim = int(cfg["dim"])
        self.do_fcl = self.fcl and not self.zorro
        self.enc_cfg = enc
        # one pass per modality, then one per combination (model.py:586); a frozenset of small ints iterates ascending
        self.passes = [[i] for i in range(M)] + [sorted(c) for c in self.combos]


def mean_token_pool(tokens, padding):
    """MeanTokenProjectionPool(None, projection=False).forward (model.py:255-276): per sample the mean of the un-padded rows,
    zeros when there is none."""
    keep = (~padding).to(tokens.dtype)[..., None]
    cnt = keep.sum(dim=1)
    return (tokens * keep).sum(dim=1) / cnt.clamp_min(1.0)


def eao_forward(S: EAOStructure, sd, batch, mode="fp32", no_loss=False):
    """EAO.forward (model.py:573-596): the layer stack + final norm + mean pool, once per pass, padding mask only."""
    P = Prec(mode)
    toks, masks = [], []
    for m in S.modalities:
        c = S.enc_cfg[m]
        pfx = f"encoders.{m}."
        if c["type"] == "EmbeddedSequenceEncoder":
            t, a = embedded_sequence_encoder(sd, pfx, batch[m], P)
        elif c["type"] == "TabularEncoder":
            t, a = tabular_encoder(sd, pfx, batch[m], P, c)
        else:
            raise NotImplementedError(c["type"])
        toks.append(t)
        masks.append(a.to(torch.bool))
    sample_mask = {m: ((a == 0).sum(dim=1) != 0) for m, a in zip(S.modalities, masks)}
    pooled = []
    for mods in S.passes:
        x = torch.cat([toks[i] for i in mods], dim=1)
        pad = torch.cat([masks[i] for i in mods], dim=1)
        for i in range(S.depth):
            x = mca_layer(x, sd, f"layers.{i}.", None, pad, S.heads, P)
        x = layer_norm(x, sd["norm.gamma"].to(P.dtype))
        pooled.append(mean_token_pool(x, pad))
    pooled = torch.stack(pooled, dim=1)
    out = pretraining_loss(S, pooled, sample_mask, sd["loss.loss_fn.logit_scale"], no_loss)
    out["modality_sample_mask"] = sample_mask
    out["pooled"] = pooled
    return out


# --------------------------------------------------------------------------------------------------
# parameters: names/shapes/initialisation follow the reference modules
# --------------------------------------------------------------------------------------------------
PARAM_BUFFERS = ("positional_encoder.pe", ".index", "fusion_mask", "token_types", "attn_mask", "pool_mask", ".beta")


def is_param(key: str) -> bool:
    return not any(key.endswith(s) or s in key for s in PARAM_BUFFERS)


def train_step(S: Structure, sd, batch, mode="fp32", lr=1e-4, clip=2.0, opt_state=None, step_fn=None):
    """One reference training step (train_accel_gpu.py:112-119): forward, backward, clip_grad_norm_(2.0),
    AdamW (torch defaults: betas .9/.999, eps 1e-8, weight_decay 0.01 on every tensor).
    Mutates ``sd`` in place; returns (outputs, grads, grad_norm)."""
    params = {k: v for k, v in sd.items() if is_param(k)}
    for p in params.values():
        p.requires_grad_(True)
        p.grad = None
    out = eao_forward(S, sd, batch, mode) if isinstance(S, EAOStructure) else mca_forward(S, sd, batch, mode)
    out["loss"].backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
    plist = [p for p in params.values() if p.grad is not None]
    gn = torch.nn.utils.clip_grad_norm_(plist, clip) if clip else None
    opt = torch.optim.AdamW(plist, lr=lr) if opt_state is None else opt_state
    opt.step()
    for p in params.values():
        p.requires_grad_(False)
    return out, grads, gn, opt

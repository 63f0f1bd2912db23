"""FusionEngine: drives the HIP kernels for one MCA/MMA training step (forward, loss, backward).

Data layout in HBM (b = per-GPU batch, N = tokens per sample, D = 512, H = 8, Ip = FF inner dim padded to
a multiple of 64):
  * parameters and gradients: two flat fp32 buffers ordered by BACKWARD COMPLETION (loss/pool first,
    encoders last) so that gradient buckets for the RCCL all-reduce are contiguous prefixes;
  * bf16 copies of every weight (plain and transposed, zero-padded to MFMA-friendly shapes), refreshed
    once per optimizer step;
  * residual stream fp32 (b*N, D) per layer (LayerNorm inputs are kept for the backward); every GEMM
    operand bf16; q|k|v packed as one (b*N, 3D) bf16 matrix written by a single fused QKV GEMM;
  * nothing of size N x N exists: the attention kernels recompute scores from q, k and a (b,H,N) fp32
    log-sum-exp.

Reference lines: forward model.py:448-478, layer algebra :117-122, pooling :470-473, loss :175-233,
encoders encoders.py:196-214 / :90-96, backward = autograd of those (train_accel_gpu.py:115).
"""
from __future__ import annotations

import os

import ctypes as C
import math
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import hip
from .encoders import EmbeddedSequenceEncoder, TabularEncoder
from .hip import AttnBwd1Args, AttnBwd2Args, AttnFp8BwdOperands, AttnFp8Operands, AttnFwdArgs, LossTerm, call, ptr, stream_ptr

LN_EPS = 1e-5
FWD_BQ, FWD_BK = 128, 64
BWD_BQ = 64          # query rows per step of the dK/dV pass (its key-block size: debug_options()['dkv_keys'])


def _pad_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _dev(a: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def debug_options() -> dict:
    """The engine's A/B switches, all behind ONE environment variable: ``MCA_DEBUG=key=value,key=value``.  Production runs set
    none of them.  overlap_wgrad=0|1 (weight-gradient GEMMs on a side stream: default by size), group_wgrad=0 (one launch per
    weight gradient instead of one per layer), mask_mfma=0 (element-wise attention mask instead of the mask product),
    dkv_keys=256 (8-wavefront key blocks in the dK/dV pass), lazy_softmax=0 (the textbook running-maximum recurrence in the forward
    attention instead of MCA_ATTN_LAZY_REFERENCE, the default since round 5: -9 % on that kernel, statistically indistinguishable
    from the textbook form over 8 data seeds, profiles/r05_lazy_softmax_seed_study.txt).  Kernel-level knobs: include/mca_hip_debug.h (hip.knobs)."""
    opts = {"overlap_wgrad": None, "group_wgrad": True, "mask_mfma": True, "dkv_keys": 128, "lazy_softmax": True,
            "onepass": None}          # onepass=0|1: the one-pass attention backward (attention_bwd1.hip); default: by size
    for item in filter(None, os.environ.get("MCA_DEBUG", "").split(",")):
        k, _, v = item.partition("=")
        k = k.strip()
        if k not in opts:
            raise ValueError(f"MCA_DEBUG: unknown switch '{k}' (known: {sorted(opts)})")
        opts[k] = int(v) if k == "dkv_keys" else (v.strip() not in ("0", "", "false"))
    return opts


class _Sched:
    """device copies of a TileSchedule"""

    def __init__(self, s, device):
        self.s = s
        # tile index with the "structurally full" flag in bit 31: one scalar load per tile in the kernels
        pack = lambda idx, full: (idx.astype(np.uint32) | (full.astype(np.uint32) << 31)).view(np.int32)
        self.q_ptr, self.q_kt, self.q_order = _dev(s.q_ptr, device), _dev(pack(s.q_kt, s.q_full), device), _dev(s.q_order, device)
        self.k_ptr, self.k_qt, self.k_order = _dev(s.k_ptr, device), _dev(pack(s.k_qt, s.k_full), device), _dev(s.k_order, device)
        # per launch slot of the backward: {key block, first entry, number of entries, query tile of the first entry}
        wg = np.zeros((len(s.k_order), 4), np.int32)
        for i, kb in enumerate(s.k_order):
            lo, hi = int(s.k_ptr[kb]), int(s.k_ptr[kb + 1])
            wg[i] = (kb, lo, hi - lo, int(s.k_qt[lo]) if hi > lo else 0)
        self.k_wg = _dev(wg, device)


class _OnePassSched:
    """device copies of a structure.OnePassSchedule (mca_attn_bwd_onepass)"""

    def __init__(self, s, device):
        self.s = s
        self.qt_desc, self.kb_desc = _dev(s.qt_desc.astype(np.int32), device), _dev(s.kb_desc.astype(np.int32), device)
        self.kb_qt = _dev(s.kb_qt.astype(np.uint32).view(np.int32), device)
        self.visit, self.row_slot = _dev(s.visit.astype(np.uint8), device), _dev(s.row_slot.astype(np.int32), device)
        self.n_qt, self.n_kb, self.max_list = len(s.qt_desc), len(s.kb_desc), int(s.kb_desc[:, 3].max())
        self.n_entries = int(len(s.kb_qt))
        self.fits = self.n_qt < 256 and self.n_kb <= 64 and self.max_list + 6 <= 256 and self.n_entries + 4 * self.n_kb <= 768          # the kernel's LDS tables


class FusionEngine:
    def __init__(self, model):
        self.model = model
        p0 = next(model.parameters())
        # EAO baseline (model.EAO): segments instead of fusion tokens, mean pooling instead of attentive pooling
        self.eao = model.attn_pool is None
        if p0.device.type != "cuda":
            raise hip.MCAHipError("the MCA step runs only on a HIP device: move the model to cuda first "
                                  "(there is no CPU fallback)")
        hip.lib()                                   # fail loudly if the extension is missing
        self.device = p0.device
        self.D, self.H, self.L = model.dim, model.heads, model.depth
        st = model.structure
        self.st = st
        self.N, self.R, self.F, self.M = st.n_tokens, st.n_return, st.num_fusion_tokens, st.n_modalities
        if self.D != self.H * 64:
            raise NotImplementedError("native path needs dim == heads * 64")
        self.I = model.layers[0].ff.inner_dim if self.L else int(self.D * 4 * 2 / 3)
        self.Ip = _pad_to(self.I, 64)
        self.scale = model.dim_head ** -0.5
        self.q_scale = self.scale * 1.4426950408889634          # folded into the forward bf16 copy of every to_q.weight
        self.attn_flags = hip.ATTN_Q_PRESCALED                  # (the kernels take pre-scaled q only)
        self.attn_dtype = "bf16"
        self.dbg = debug_options()                               # A/B switches: ONE environment variable, MCA_DEBUG
        if self.dbg["lazy_softmax"]:
            self.attn_flags |= hip.ATTN_LAZY_REFERENCE
        self.nk_pad = _pad_to(self.N, 256)
        # the attention mask as a matrix product (mca_build_keyhot): needs every key group id <= 14
        self.mask_mfma = int(self.st.kgroup.max()) <= 14 and self.dbg["mask_mfma"]
        self._flatten_parameters()
        self._build_static()
        self._alloc_weights()
        self._ws: Dict[int, dict] = {}
        self._weights_version = -1
        self.grad_bucket_hook: Optional[Callable[[int, int], None]] = None   # (lo, hi) offsets ready
        self.gather_hook: Optional[Callable] = None                          # DP: pooled/present all-gather
        # Finite checks of the reference (encoders.py:197-213) without its host syncs: the kernels OR bits into one device
        # word; the host reads it once per step through a pinned mirror.  check_finite: True = raise inside the forward that
        # saw the bad values (one sync per forward, the reference's behaviour); "deferred" = no sync: poll_finite() /
        # assert_finite() raise afterwards, and FusedAdamW skips the update of a flagged step on the device; False = off.
        self.check_finite = True
        self.finite_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._flag_event: Optional[torch.cuda.Event] = None
        self.fuse_geglu_bwd = True
        self._mod_shifts = torch.arange(len(model.modality_types), dtype=torch.int32, device=self.device)
        self.fuse_ln_residual = True                # residual LayerNorm recomputed in the GEMM epilogue (large batches)
        # Weight-gradient GEMMs on a side stream, concurrent with the backward chain: None = by size (on below OVERLAP_ROWS_MAX
        # token rows).  Since the attention backward lost its atomics the main stream keeps every CU busy and the persistent
        # weight-gradient GEMMs of a second stream only take CUs from it (one box, alternating processes: 23.7 / 25.6 ms per
        # b = 32 step with the side stream, 22.17 / 22.16 without); at small batches (b = 8: the N = 512 GEMMs fill 160 of 256
        # CUs) the side stream still pays in the eager loop (8.7-8.8 against 9.5-10.2 ms).  MCA_DEBUG=overlap_wgrad=0|1 forces it.
        self.overlap_wgrad = self.dbg["overlap_wgrad"]
        self.group_wgrad = self.dbg["group_wgrad"]          # one weight-gradient launch per layer

    # ------------------------------------------------------------------------------------------------
    # parameters -> one flat buffer (and one for gradients)
    # ------------------------------------------------------------------------------------------------
    def _flatten_parameters(self):
        m = self.model
        order: List[torch.nn.Parameter] = []
        marks: List[int] = []                        # bucket boundaries (indices into `order`)
        order += [m.loss.loss_fn.logit_scale]
        # the pooling key/value projection's gradient is reduced over all T token rows: it rides in the top layer's grouped
        # weight-gradient launch (_backward_layers_and_encoders) and therefore belongs to that layer's bucket
        pool_kv_late = not self.eao and self.L > 0
        if not self.eao:
            order += [m.return_tokens, m.attn_pool.to_out.weight, m.attn_pool.to_q.weight]
            if not pool_kv_late:
                order += [m.attn_pool.to_kv.weight]
        order += [m.norm.gamma]
        marks.append(len(order))
        if pool_kv_late:
            order += [m.attn_pool.to_kv.weight]
        for i in reversed(range(self.L)):
            ly = m.layers[i]
            order += [ly.ff.feedforward[2].weight, ly.ff.feedforward[0].weight, ly.attn.to_out.weight,
                      ly.attn.to_q.weight, ly.attn.to_kv.weight, ly.norm.gamma]
            marks.append(len(order))
        if not self.eao:
            order.append(m.fusion_tokens)
        for name in m.modality_types:
            order += list(m.encoders[name].parameters())
        marks.append(len(order))
        seen = {id(p) for p in order}
        for p in m.parameters():
            if id(p) not in seen:
                order.append(p)
        marks[-1] = len(order)
        offs, n = [], 0
        for p in order:
            n = _pad_to(n, 4)                        # keep every tensor 16-byte aligned
            offs.append(n)
            n += p.numel()
        total = _pad_to(n, 4)
        flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        gflat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.param_views, self.grad_views = {}, {}
        for p, o in zip(order, offs):
            if p.dtype != torch.float32:
                raise TypeError("parameters must be fp32 (bf16 is used for GEMM operands only)")
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
            self.grad_views[id(p)] = gflat[o:o + p.numel()].view(p.shape)
        self.flat, self.gflat = flat, gflat
        self.param_order, self.param_offsets = order, offs
        self.bucket_bounds = [offs[i] if i < len(offs) else total for i in marks]
        self.bucket_bounds[-1] = total
        self.n_params = total

    def grad_of(self, p) -> torch.Tensor:
        return self.grad_views[id(p)]

    # ------------------------------------------------------------------------------------------------
    def _build_static(self):
        st, dev = self.st, self.device
        self.kgroup = _dev(st.kgroup.astype(np.uint8), dev)
        self.qmask_attn = _dev(st.qmask_attn.astype(np.uint32).view(np.int32), dev)
        self.qmask_pool = _dev(st.qmask_pool.astype(np.uint32).view(np.int32), dev)
        # query side of the mask product (mca_hip.h, mca_build_keyhot): qblk[i][g] = group g visible ? 0 : -32768, slot 15 blocked
        def qblk_of(qm):
            bits = (qm.astype(np.uint32)[:, None] >> np.arange(16, dtype=np.uint32)[None, :]) & 1
            bits[:, 15] = 0
            return torch.from_numpy(np.where(bits == 1, 0.0, -32768.0).astype(np.float32)).to(torch.bfloat16).to(dev).contiguous()
        self.qblk_attn, self.qblk_pool = qblk_of(st.qmask_attn), qblk_of(st.qmask_pool)
        self.sched_attn_f = _Sched(st.attn_schedule(FWD_BQ, FWD_BK), dev)
        # key-block size of the dkv pass: 128 (4 wavefronts, two independent workgroups per CU) is 3 % faster than 256 (8 wavefronts,
        # one workgroup per CU) at N = 2538 and equal at N = 6088
        dkv_keys = self.dbg["dkv_keys"]
        self.dkv_keys = dkv_keys
        self.sched_attn_b2 = _Sched(st.attn_schedule(BWD_BQ, dkv_keys), dev)
        # one-pass backward of the layer attention: needs the mask product and tables that fit the kernel's LDS
        self.sched_onepass = None
        if self.mask_mfma and self.dbg["onepass"] is not False:
            sc = _OnePassSched(st.attn_onepass_schedule(aligned=True), dev)
            self.sched_onepass = sc if sc.fits else None
        if self.eao:
            self.seg_start = _dev(st.seg_start, dev)
        else:
            self.sched_pool_f = _Sched(st.pool_schedule(FWD_BQ, FWD_BK), dev)
            self.sched_pool_b2 = _Sched(st.pool_schedule(BWD_BQ, dkv_keys), dev)
        terms = self.model.loss_terms
        arr = (LossTerm * len(terms))()
        for i, t in enumerate(terms):
            arr[i] = LossTerm(t.slot_a, t.slot_b, t.and_bits, t.or_bits)
        self.n_terms = len(terms)
        self.loss_terms_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.offsets = np.concatenate([[0], np.cumsum(st.token_dims)]).astype(int).tolist()

    # ------------------------------------------------------------------------------------------------
    # bf16 weight copies
    # ------------------------------------------------------------------------------------------------
    def _alloc_weights(self):
        D, Ip, dev = self.D, self.Ip, self.device
        bf = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
        self.wl = []
        for _ in range(self.L):
            self.wl.append(dict(qkv=bf(3 * D, D), qkvT=bf(D, 3 * D), o=bf(D, D), oT=bf(D, D),
                                w1=bf(2 * Ip, D), w1T=bf(D, 2 * Ip), w2=bf(D, Ip), w2T=bf(Ip, D)))
        self.wp = dict(q=bf(D, D), qT=bf(D, D), kv=bf(2 * D, D), kvT=bf(D, 2 * D), o=bf(D, D), oT=bf(D, D))
        self.we = {}
        for name in self.model.modality_types:
            enc = self.model.encoders[name]
            if isinstance(enc, EmbeddedSequenceEncoder):
                kp = _pad_to(enc.input_size, 64)
                self.we[name] = dict(kp=kp, w=bf(D, kp), wT=bf(kp, D))
            elif isinstance(enc, TabularEncoder):
                self.we[name] = dict(w2=bf(D, D), w2T=bf(D, D))

    def _cast(self, src: torch.Tensor, dst: torch.Tensor, transpose=False, dst_row0=0, dst_col0=0, scale=0.0):
        """dst[dst_row0:, dst_col0:] (bf16) <- src (fp32 2-D), zero padding untouched (buffers start zeroed).  Only
        RECORDS the copy: all of them run as one multi-tensor launch (the parameter / copy addresses never change)."""
        r, c = src.shape
        d = dst[dst_row0:, dst_col0:]
        rp, cp = (c, r) if transpose else (r, c)
        self._cast_list.append(hip.CastDesc(ptr(src), ptr(d), src.stride(0), r, c, dst.stride(0), rp, cp, int(transpose), float(scale)))

    def set_attention_dtype(self, dtype: str):
        """'bf16' (default) or 'fp8' (BASELINE configs[4]): the fusion layers' forward attention computes Q K^T and P V on the
        block-scaled fp8 matrix instruction from MX-fp8 copies of q, k, v (one quantisation pass per layer,
        mca_attn_quant_mxfp8), and their backward recomputes S = Q K^T and dP = dO V^T on it (mca_attn_quant_bwd_mxfp8 +
        mca_attn_bwd_dq_fp8 / mca_attn_bwd_dkv_fp8; the three gradient products stay bf16; needs the mask product and
        128-key dkv blocks, else the backward keeps bf16 recomputes).  Pooling attention stays bf16."""
        if dtype not in ("bf16", "fp8"):
            raise ValueError(dtype)
        self.attn_dtype = dtype

    def _fp8_operands(self, ws, b, layer):
        """MX-fp8 forward operands: q8 / k8 (+ scales) PER LAYER (the backward's score recomputes read them again: 2 x 1.03 bytes
        per q / k element and layer, 4 GB at configs[4]'s b = 128), one shared V^T set (forward only)"""
        key = ("fp8", layer)
        if key not in ws:
            nt = (self.N + 63) // 64
            u8 = lambda *s: torch.zeros(*s, dtype=torch.uint8, device=self.device)
            if "fp8v" not in ws:
                ws["fp8v"] = dict(v8t=u8(b, self.H, nt, 64, 64), vs=u8(b, self.H, nt, 64, 2))
            bufs = dict(q8=u8(b, self.H, nt * 64, 64), qs=u8(b, self.H, nt * 64, 2), k8=u8(b, self.H, nt * 64, 64), ks=u8(b, self.H, nt * 64, 2),
                        **ws["fp8v"])
            f = AttnFp8Operands()
            f.q8, f.qs, f.k8, f.ks, f.v8t, f.vs = (bufs[k].data_ptr() for k in ("q8", "qs", "k8", "ks", "v8t", "vs"))
            f.n_ktiles = nt
            ws[key] = (bufs, f)
        return ws[key]

    def _fp8_bwd_operands(self, ws, b, layer):
        """MX-fp8 backward operands of a layer: q8 / k8 are the arrays its forward wrote (same bits as a re-quantisation), v and
        dO (quantised along d) share one set of buffers across layers.  -> (operands, which-mask for mca_attn_quant_bwd_mxfp8)"""
        key = ("fp8b", layer)
        if key not in ws:
            nt = (self.N + 63) // 64
            u8 = lambda *s: torch.zeros(*s, dtype=torch.uint8, device=self.device)
            if "fp8bv" not in ws:
                ws["fp8bv"] = {**{k: u8(b, self.H, nt * 64, 64) for k in ("v8", "do8")}, **{k: u8(b, self.H, nt * 64, 2) for k in ("vs", "dos")}}
            fwd = self._fp8_operands(ws, b, layer)[0]
            bufs = {**{k: fwd[k] for k in ("q8", "qs", "k8", "ks")}, **ws["fp8bv"]}
            f = AttnFp8BwdOperands()
            for k in ("q8", "qs", "k8", "ks", "v8", "vs", "do8", "dos"):
                setattr(f, k, bufs[k].data_ptr())
            f.n_ktiles = nt
            ws[key] = (bufs, f)
        # q8 / k8 hold this step's q, k only if THIS forward quantised them (a forward run in bf16 leaves them stale)
        fresh = ws.get(("fp8gen", layer)) == ws["gen"]
        return ws[key][1], (0b1100 if fresh else 0b1111)

    def fp8_backward_on(self, ws, nq) -> bool:
        return self.attn_dtype == "fp8" and nq == self.N and ws.get("khot") is not None and self.dkv_keys == 128

    def invalidate_weights(self):
        """Call after writing parameters through an alias autograd's version counters cannot see (``p.data.op_()``, a raw
        pointer): the bf16 GEMM-weight copies are rebuilt by the next forward."""
        self._weights_version = -1

    def _param_version(self) -> int:
        # every Parameter is a view of `flat` with its OWN version counter (p.data = flat[...]): load_state_dict, p.copy_,
        # torch.optim steps and dist.broadcast(p) bump the parameter's counter, FusedAdamW / broadcast(flat) bump flat's
        return self.flat._version + sum(p._version for p in self.param_order)

    def refresh_weights(self, force=False):
        v = self._param_version()
        if not force and v == self._weights_version:
            return
        if getattr(self, "_cast_table", None) is not None:
            call("mca_cast_pad_bf16_multi", ptr(self._cast_table), self._cast_n, stream_ptr())
            self._weights_version = v
            return
        self._cast_list = []
        m, D, I, Ip = self.model, self.D, self.I, self.Ip
        for i, ly in enumerate(m.layers):
            w = self.wl[i]
            # forward copy of W_q carries scale * log2(e): q.k comes out of the QKV GEMM as the log2-domain logit (the
            # transposed copies used by the backward data-gradient GEMMs stay unscaled: mca_hip.h, MCA_ATTN_Q_PRESCALED)
            self._cast(ly.attn.to_q.weight.data, w["qkv"], scale=self.q_scale)
            self._cast(ly.attn.to_kv.weight.data, w["qkv"], dst_row0=D)
            self._cast(ly.attn.to_q.weight.data, w["qkvT"], transpose=True)
            self._cast(ly.attn.to_kv.weight.data, w["qkvT"], transpose=True, dst_col0=D)
            self._cast(ly.attn.to_out.weight.data, w["o"])
            self._cast(ly.attn.to_out.weight.data, w["oT"], transpose=True)
            w1 = ly.ff.feedforward[0].weight.data
            self._cast(w1[:I], w["w1"])
            self._cast(w1[I:], w["w1"], dst_row0=Ip)
            self._cast(w1[:I], w["w1T"], transpose=True)
            self._cast(w1[I:], w["w1T"], transpose=True, dst_col0=Ip)
            w2 = ly.ff.feedforward[2].weight.data
            self._cast(w2, w["w2"])
            self._cast(w2, w["w2T"], transpose=True)
        ap = m.attn_pool
        if ap is not None:
            self._cast(ap.to_q.weight.data, self.wp["q"], scale=self.q_scale); self._cast(ap.to_q.weight.data, self.wp["qT"], transpose=True)
            self._cast(ap.to_kv.weight.data, self.wp["kv"]); self._cast(ap.to_kv.weight.data, self.wp["kvT"], transpose=True)
            self._cast(ap.to_out.weight.data, self.wp["o"]); self._cast(ap.to_out.weight.data, self.wp["oT"], transpose=True)
        for name in m.modality_types:
            enc = m.encoders[name]
            if isinstance(enc, EmbeddedSequenceEncoder):
                self._cast(enc.token_encoder[1].weight.data, self.we[name]["w"])
                self._cast(enc.token_encoder[1].weight.data, self.we[name]["wT"], transpose=True)
            elif isinstance(enc, TabularEncoder):
                self._cast(enc.value_encoder.linear2.weight.data, self.we[name]["w2"])
                self._cast(enc.value_encoder.linear2.weight.data, self.we[name]["w2T"], transpose=True)
        arr = (hip.CastDesc * len(self._cast_list))(*self._cast_list)
        self._cast_n = len(self._cast_list)
        self._cast_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        call("mca_cast_pad_bf16_multi", ptr(self._cast_table), self._cast_n, stream_ptr())
        self._weights_version = v

    # ------------------------------------------------------------------------------------------------
    # workspaces for a given local batch size
    # ------------------------------------------------------------------------------------------------
    def workspace(self, b: int) -> dict:
        if b in self._ws:
            return self._ws[b]
        D, N, H, Ip, R, dev = self.D, self.N, self.H, self.Ip, self.R, self.device
        T = b * N
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        u8 = lambda *s: torch.empty(*s, dtype=torch.uint8, device=dev)
        ws = dict(b=b, T=T, gen=0)
        ws["x"] = [f32(T, D) for _ in range(self.L + 1)]
        ws["layers"] = [dict(x1=f32(T, D), m1=f32(T), r1=f32(T), m2=f32(T), r2=f32(T), xn_b=bf(T, D), qkv=bf(T, 3 * D),
                             o=bf(T, D), lse=f32(b, H, N), x1n_b=bf(T, D), h=bf(T, 2 * Ip), g=bf(T, Ip),
                             # backward operands of the weight-gradient GEMMs: one set PER LAYER, so those GEMMs can run
                             # on a side stream without write-after-read hazards against the next layer's backward
                             dxo_b=bf(T, D), dx1_b=bf(T, D), dh=bf(T, 2 * Ip), dqkv=bf(T, 3 * D))
                        for _ in range(self.L)]
        ws["xn"], ws["x1n"] = f32(T, D), f32(T, D)
        ws["mf"], ws["rf"] = f32(T), f32(T)
        ws["t_b"], ws["kvp"] = (None, None) if self.eao else (bf(T, D), bf(T, 2 * D))          # operands of the attentive pooling
        ws["rt_b"], ws["qp"], ws["op"], ws["lse_p"] = bf(R, D), bf(R, D), bf(b * R, D), f32(b, H, R)
        ws["pooled"] = f32(b * R, D)
        if self.eao:
            ws["seg_counts"] = torch.zeros(b, R, dtype=torch.int32, device=dev)
        ws["vmean"], ws["dvmean"], ws["delta"], ws["delta_p"] = torch.zeros(b, D, dtype=torch.float32, device=dev), f32(b, D), f32(b, H, N), f32(b, H, R)
        ws["keyinfo"], ws["kflags"] = u8(b, self.nk_pad), u8(b, (N + 63) // 64)
        ws["khot"] = bf(b, self.nk_pad, 16) if self.mask_mfma else None
        ws["padding"] = u8(b, N)
        ws["present"] = torch.zeros(b, dtype=torch.int32, device=dev)
        ws["present_native"] = torch.zeros(b, dtype=torch.int32, device=dev)
        # backward
        ws["dxa"], ws["dxb"], ws["dx_b"] = f32(T, D), f32(T, D), bf(T, D)
        ws["dg"], ws["do"] = bf(T, Ip), bf(T, D)
        ws["dpool_b"], ws["dop"], ws["dqp32"], ws["dqp_sum"], ws["dqp_b"] = bf(b * R, D), bf(b * R, D), f32(b * R, D), f32(R, D), bf(R, D)
        ws["dkvp"], ws["drt"] = (None if self.eao else bf(T, 2 * D)), f32(R, D)
        ws["enc"] = {}
        for mi, name in enumerate(self.model.modality_types):
            enc = self.model.encoders[name]
            n = self.st.token_dims[mi]
            rows = b * n
            if isinstance(enc, EmbeddedSequenceEncoder):
                kp = self.we[name]["kp"]
                ws["enc"][name] = dict(xin_b=bf(rows, kp), m0=f32(rows), r0=f32(rows), y=f32(rows, D), m2=f32(rows), r2=f32(rows),
                                       dy=f32(rows, D), dy_b=bf(rows, D), dxin=f32(rows, kp), mask=u8(rows))
            elif isinstance(enc, TabularEncoder):
                ws["enc"][name] = dict(h1_b=bf(rows, D), y=f32(rows, D), m2=f32(rows), r2=f32(rows), dy=f32(rows, D),
                                       dy_b=bf(rows, D), dh1=f32(rows, D), mask=u8(rows))
        ws["side"], ws["side_events"] = torch.cuda.Stream(device=dev), []          # side stream of the weight-gradient GEMMs
        self._ws[b] = ws
        return ws

    # ------------------------------------------------------------------------------------------------
    # thin kernel wrappers
    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def gemm_nt(A, B, C, M, N, K, bias=None, residual=None, res_period=0):
        """C[M,N] = A[M,K] B[N,K]^T (+bias) (+residual)."""
        if hip.PROFILE is not None:          # live timing (bench.py): one key per problem shape, so an average is over equal launches
            hip.set_tag(f"{M}x{N}x{K}{'' if C.dtype == torch.bfloat16 else ' f32'}{' +res' if residual is not None else ''}")
        call("mca_gemm_nt", ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0), int(C.dtype == torch.bfloat16),
             ptr(bias), ptr(residual), residual.stride(0) if residual is not None else 0, res_period, M, N, K, stream_ptr(),
             flops=2.0 * M * N * K)
        if hip.PROFILE is not None:
            hip.set_tag("")

    @staticmethod
    def gemm_tn_acc(A, B, Cgrad, R, N, K):
        """Cgrad[N,K] += A[R,N]^T B[R,K]"""
        call("mca_gemm_tn_acc", ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(Cgrad), Cgrad.stride(0), R, N, K, stream_ptr(),
             flops=2.0 * R * N * K)

    @staticmethod
    def gemm_tn_acc_group(members, R):
        """members: [(A, B, Cgrad, N, K)]: Cgrad[N,K] += A[R,N]^T B[R,K] for every member, one launch (mca_gemm_tn_acc_group)"""
        arr = (hip.TnDesc * len(members))()
        fl = 0.0
        for d, (A, B, Cg, N, K) in zip(arr, members):
            d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.N, d.K = ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(Cg), Cg.stride(0), N, K
            fl += 2.0 * R * N * K
        call("mca_gemm_tn_acc_group", C.byref(arr), len(members), R, stream_ptr(), flops=fl)

    @staticmethod
    def ln_fwd(x, gamma, rows, cols, mean, rstd, beta=None, rowmask=None, add=None, period=0, y=None, ldy=0, y_bstride=0,
               y_bf16=None, cols_pad=0):
        call("mca_layernorm_fwd", ptr(x), x.stride(0), ptr(gamma), ptr(beta), ptr(rowmask), ptr(add), period,
             ptr(y), ldy, y_bstride, ptr(y_bf16), y_bf16.stride(0) if y_bf16 is not None else 0, cols_pad,
             ptr(mean), ptr(rstd), rows, cols, LN_EPS, stream_ptr())

    @staticmethod
    def ln_bwd(dy, ldy, x, gamma, mean, rstd, rows, cols, dgamma, dbeta=None, rowmask=None, dx=None, dx_bf16=None,
               y_bstride=0, period=0, dxsum=None):
        call("mca_layernorm_bwd", ptr(dy), ldy, y_bstride, period, ptr(x), x.stride(0), ptr(gamma), ptr(mean), ptr(rstd),
             ptr(rowmask), ptr(dx), dx.stride(0) if dx is not None else 0, ptr(dx_bf16),
             dx_bf16.stride(0) if dx_bf16 is not None else 0, ptr(dgamma), ptr(dbeta), ptr(dxsum), rows, cols, stream_ptr())

    def _attn_fwd(self, q, q_bstride, q_ld, kv, k_off, v_off, kv_ld, o, lse, qmask, sched, ws, b, nq, layer=0):
        N = self.N
        a = AttnFwdArgs()
        esz = 2
        a.q, a.q_bstride, a.q_ld = q, q_bstride, q_ld
        a.k, a.v = kv.data_ptr() + k_off * esz, kv.data_ptr() + v_off * esz
        a.kv_bstride, a.kv_ld = N * kv_ld, kv_ld
        a.o, a.o_bstride, a.o_ld = o.data_ptr(), nq * o.stride(0), o.stride(0)
        a.lse = lse.data_ptr()
        a.qmask, a.keyinfo, a.ktile_flags = qmask.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr()
        a.q_ptr, a.q_kt, a.q_order = sched.q_ptr.data_ptr(), sched.q_kt.data_ptr(), sched.q_order.data_ptr()
        a.vmean = ws["vmean"].data_ptr()
        a.batch, a.heads, a.nq, a.nk, a.nk_pad = b, self.H, nq, N, self.nk_pad
        a.n_qtiles, a.n_ktiles, a.scale, a.flags = sched.s.n_q, sched.s.n_k, self.scale, self.attn_flags
        a.khot = ws["khot"].data_ptr() if ws.get("khot") is not None else None
        # mean(V) is the output of fully masked rows only: samples with every modality present have none and are skipped
        if ws.get("present_cur") is not None:
            call("mca_attn_vmean_if_needed", a.v, a.kv_bstride, a.kv_ld, ws["vmean"].data_ptr(), b, N, self.H, ptr(ws["present_cur"]),
                 (1 << self.M) - 1, stream_ptr())
        else:
            call("mca_attn_vmean", a.v, a.kv_bstride, a.kv_ld, ws["vmean"].data_ptr(), b, N, self.H, stream_ptr())
        hip.set_tag("pool" if nq != N else "layer")
        if self.attn_dtype == "fp8" and nq == N:
            f = self._fp8_operands(ws, b, layer)[1]
            ws[("fp8gen", layer)] = ws["gen"]
            call("mca_attn_quant_mxfp8", a.q, a.q_bstride, a.q_ld, a.k, a.v, a.kv_bstride, a.kv_ld, C.byref(f), b, self.H, N, stream_ptr())
            call("mca_attn_fwd_fp8", C.byref(a), C.byref(f), stream_ptr(), flops=4.0 * 64 * sched.s.allowed_pairs * self.H * b)
        else:
            call("mca_attn_fwd", C.byref(a), stream_ptr(), flops=4.0 * 64 * sched.s.allowed_pairs * self.H * b)
        hip.set_tag("")

    def _attn_bwd2(self, q, q_bstride, q_ld, kv, k_off, v_off, kv_ld, o, d_o, lse, delta, dq_ptr, dq_bstride, dq_ld, dq_f32, dkv,
                   dk_off, dv_off, dkv_ld, qmask, sched_f, sched_b, ws, b, nq, layer=0):
        """two-pass backward (attention_bwd2.hip): dq (bf16 or fp32) is WRITTEN, not accumulated.  The layer attention at a
        batch that gives every CU a (sample, head) takes the one-pass form (attention_bwd1.hip) instead."""
        N, esz = self.N, 2
        if self.use_onepass(ws, b, nq, dq_f32):
            return self._attn_bwd1(q, q_bstride, q_ld, kv, k_off, v_off, kv_ld, o, d_o, lse, dq_ptr, dq_bstride, dq_ld, dkv, dk_off, dv_off,
                                   dkv_ld, ws, b)
        call("mca_attn_bwd_prep", o.data_ptr(), d_o.data_ptr(), nq * o.stride(0), o.stride(0), lse.data_ptr(),
             delta.data_ptr(), ws["dvmean"].data_ptr(), b, self.H, nq, N, stream_ptr())
        a = AttnBwd2Args()
        a.q, a.q_bstride, a.q_ld = q, q_bstride, q_ld
        a.k, a.v = kv.data_ptr() + k_off * esz, kv.data_ptr() + v_off * esz
        a.kv_bstride, a.kv_ld = N * kv_ld, kv_ld
        a.d_o, a.o_bstride, a.o_ld = d_o.data_ptr(), nq * d_o.stride(0), d_o.stride(0)
        a.lse, a.delta, a.dvmean = lse.data_ptr(), delta.data_ptr(), ws["dvmean"].data_ptr()
        a.dq, a.dq_bstride, a.dq_ld, a.dq_f32 = dq_ptr, dq_bstride, dq_ld, int(dq_f32)
        a.dk, a.dv = dkv.data_ptr() + dk_off * esz, dkv.data_ptr() + dv_off * esz
        a.dkv_bstride, a.dkv_ld = N * dkv_ld, dkv_ld
        a.qmask, a.keyinfo, a.ktile_flags = qmask.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr()
        a.q_ptr, a.q_kt, a.q_order = sched_f.q_ptr.data_ptr(), sched_f.q_kt.data_ptr(), sched_f.q_order.data_ptr()
        a.n_qtiles128, a.n_ktiles64 = sched_f.s.n_q, sched_f.s.n_k
        a.k_wg, a.k_qt, a.n_qtiles64, a.n_kblocks256 = sched_b.k_wg.data_ptr(), sched_b.k_qt.data_ptr(), sched_b.s.n_q, sched_b.s.n_k
        a.batch, a.heads, a.nq, a.nk, a.nk_pad, a.scale, a.flags = b, self.H, nq, N, self.nk_pad, self.scale, self.attn_flags
        if ws.get("khot") is not None:
            a.khot, a.qblk = ws["khot"].data_ptr(), (self.qblk_attn if qmask is self.qmask_attn else self.qblk_pool).data_ptr()
        a.kblock_keys = sched_b.s.bk
        pairs = sched_b.s.allowed_pairs
        hip.set_tag("pool" if nq != N else "layer")
        # algorithmic flops of the whole backward (2 x forward) split 3 : 5 over the passes by their share of the five
        # products a one-pass backward needs (dq pass: S, dP, dQ minus the recomputed S, dP counted once)
        if self.fp8_backward_on(ws, nq):
            f, which = self._fp8_bwd_operands(ws, b, layer)
            call("mca_attn_quant_bwd_mxfp8", a.q, a.q_bstride, a.q_ld, a.k, a.v, a.kv_bstride, a.kv_ld, a.d_o, a.o_bstride, a.o_ld,
                 C.byref(f), which, b, self.H, N, stream_ptr())
            call("mca_attn_bwd_dkv_fp8", C.byref(a), C.byref(f), stream_ptr(), flops=8.0 * 64 * pairs * self.H * b * 0.6)
            call("mca_attn_bwd_dq_fp8", C.byref(a), C.byref(f), stream_ptr(), flops=8.0 * 64 * pairs * self.H * b * 0.4)
        else:
            call("mca_attn_bwd_dkv", C.byref(a), stream_ptr(), flops=8.0 * 64 * pairs * self.H * b * 0.6)
            call("mca_attn_bwd_dq", C.byref(a), stream_ptr(), flops=8.0 * 64 * pairs * self.H * b * 0.4)
        hip.set_tag("")

    ONEPASS_MIN_WG = 192          # (sample, head) pairs from which the one-pass backward is the default: one workgroup per CU

    def use_onepass(self, ws, b, nq, dq_f32=False) -> bool:
        """The one-pass bf16 backward wherever it applies - also with fp8 attention operands: it is faster than the two-pass backward
        with fp8 score recomputes (LONG b = 128: 9.1 against 10.5 ms per layer), so `set_attention_dtype("fp8")` then means the fp8
        forward + this backward; the fp8 two-pass backward remains for small batches (and MCA_DEBUG=onepass=0)."""
        if self.sched_onepass is None or nq != self.N or dq_f32 or ws.get("khot") is None:
            return False
        want = self.dbg["onepass"]
        return bool(want) if want is not None else b * self.H * self.onepass_split(b) >= self.ONEPASS_MIN_WG

    def onepass_split(self, b) -> int:
        """workgroups per (sample, head) of the one-pass backward: 1 where the batch gives every CU a (sample, head), else up to 4
        (key blocks dealt round robin, partial dQ sums added by the call's second launch) - b = 8, 8 heads: 4 x 64 = 256 workgroups"""
        wg = b * self.H
        return 1 if wg >= self.ONEPASS_MIN_WG else max(1, min(4, self.sched_onepass.n_kb if self.sched_onepass else 1, -(-256 // wg)))

    def backward_form(self, b) -> str:
        """what the layer attention's backward runs at batch b (reported by bench.py)"""
        ws = {"khot": True if self.mask_mfma else None}
        if self.use_onepass(ws, b, self.N):
            return "bf16 one-pass" + (f" (key blocks split {self.onepass_split(b)} ways)" if self.onepass_split(b) > 1 else "")
        return "fp8 two-pass" if (self.attn_dtype == "fp8" and self.mask_mfma and self.dkv_keys == 128) else "bf16 two-pass"

    def _attn_bwd1(self, q, q_bstride, q_ld, kv, k_off, v_off, kv_ld, o, d_o, lse, dq_ptr, dq_bstride, dq_ld, dkv, dk_off, dv_off, dkv_ld, ws, b):
        """one-pass backward of the layer attention (attention_bwd1.hip): dq, dk, dv bf16, every element written"""
        N, esz, sc = self.N, 2, self.sched_onepass
        if "rowc" not in ws:          # positions past a tile's rows: -inf | 0 (written once; the prep kernel only touches real rows)
            rc = torch.empty(b, self.H, sc.n_qt + 1, 2, 64, dtype=torch.float32, device=self.device)          # (+ the null tile)
            rc[:, :, :, 0] = float("-inf"); rc[:, :, :, 1] = 0.0
            ws["rowc"] = rc
            ws["dq_acc"] = torch.empty(b * self.H * self.onepass_split(b) * (sc.n_qt + 1) * 4096, dtype=torch.float32, device=self.device)          # (+ the null tile's slot)
            # head-major packed copies of q and dO (written by the prep launch): a query tile of a head is contiguous memory
            # (+ 64 rows: the kernel reads whole 64-row tiles, the last one past its rows)
            ws["q_hm"] = torch.zeros((b * self.H * N + 64) * 64, dtype=torch.bfloat16, device=self.device)
            ws["do_hm"] = torch.zeros((b * self.H * N + 64) * 64, dtype=torch.bfloat16, device=self.device)
        call("mca_attn_bwd_prep_onepass", o.data_ptr(), d_o.data_ptr(), N * o.stride(0), o.stride(0), lse.data_ptr(), sc.row_slot.data_ptr(),
             ws["rowc"].data_ptr(), ws["dvmean"].data_ptr(), b, self.H, N, sc.n_qt, q, q_bstride, q_ld, ws["q_hm"].data_ptr(),
             ws["do_hm"].data_ptr(), stream_ptr())
        a = AttnBwd1Args()
        a.q, a.q_bstride, a.q_hstride, a.q_ld = ws["q_hm"].data_ptr(), self.H * N * 64, N * 64, 64
        a.k, a.v = kv.data_ptr() + k_off * esz, kv.data_ptr() + v_off * esz
        a.kv_bstride, a.kv_ld = N * kv_ld, kv_ld
        a.d_o, a.o_bstride, a.o_hstride, a.o_ld = ws["do_hm"].data_ptr(), self.H * N * 64, N * 64, 64
        a.rowc, a.dvmean = ws["rowc"].data_ptr(), ws["dvmean"].data_ptr()
        a.dq, a.dq_bstride, a.dq_ld = dq_ptr, dq_bstride, dq_ld
        a.dk, a.dv = dkv.data_ptr() + dk_off * esz, dkv.data_ptr() + dv_off * esz
        a.dkv_bstride, a.dkv_ld = N * dkv_ld, dkv_ld
        a.dq_acc = ws["dq_acc"].data_ptr()
        a.keyinfo, a.ktile_flags, a.khot, a.qblk = ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr(), ws["khot"].data_ptr(), self.qblk_attn.data_ptr()
        a.qt_desc, a.kb_desc, a.kb_qt, a.visit = sc.qt_desc.data_ptr(), sc.kb_desc.data_ptr(), sc.kb_qt.data_ptr(), sc.visit.data_ptr()
        a.n_qtiles, a.n_kblocks, a.max_list, a.n_entries = sc.n_qt, sc.n_kb, sc.max_list, sc.n_entries
        a.batch, a.heads, a.n, a.nk_pad, a.n_ktiles64 = b, self.H, N, self.nk_pad, (N + 63) // 64
        a.scale, a.flags, a.split = self.scale, self.attn_flags, self.onepass_split(b)
        hip.set_tag("layer")
        call("mca_attn_bwd_onepass", C.byref(a), stream_ptr(), flops=8.0 * 64 * sc.s.allowed_pairs * self.H * b)
        hip.set_tag("")

    # ------------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------------
    def _encode(self, batch, ws, need_grad: bool, renorm: bool = True):
        """encoders + packing (model.py:455-466): writes ws['x'][0] (b, N, D), ws['padding'] (b, N) and ws['present'] (b,)
        int32 (bit i = modality i has a valid token in that sample); returns modality_sample_mask {name: (b,) bool}."""
        m, D, N, b = self.model, self.D, self.N, ws["b"]
        x0 = ws["x"][0]
        ws["foreign"] = {}
        # ---- masks of every native modality in ONE launch: padding, the encoders' row masks, presence bits
        pk = hip.PackMasksArgs()
        keep = []                         # keeps converted mask tensors alive until the launch is enqueued
        native_idx = []
        for mi, name in enumerate(m.modality_types):
            enc = m.encoders[name]
            if not isinstance(enc, (EmbeddedSequenceEncoder, TabularEncoder)):
                continue
            n, off = self.st.token_dims[mi], self.offsets[mi]
            am = batch[name]["attention_mask"]
            if am.dtype == torch.bool or am.dtype == torch.uint8:
                eb = 1
            elif am.dtype == torch.int64:
                eb = 8
            else:
                am, eb = am.to(torch.bool), 1
            if not am.is_contiguous():
                am = am.contiguous()
            if am.shape != (b, n):
                raise AssertionError(f"{name}: attention_mask {tuple(am.shape)} != {(b, n)}")
            keep.append(am)
            d = pk.m[len(native_idx)]
            d.mask, d.elem_bytes, d.n, d.offset = am.data_ptr(), eb, n, off
            d.rowmask = ws["enc"][name]["mask"].data_ptr() if isinstance(enc, EmbeddedSequenceEncoder) else None
            native_idx.append(mi)
        foreign = len(native_idx) != len(m.modality_types)
        if native_idx:
            pk.n_mod, pk.batch, pk.n_tokens, pk.n_fusion = len(native_idx), b, N, (0 if foreign else self.F)
            call("mca_pack_masks", C.byref(pk), ptr(ws["padding"]), ptr(ws["present_native"]), stream_ptr())
        present = ws["present"]
        if not foreign:
            present = ws["present_native"]          # bit k = k-th modality: the native list IS the modality list
        else:
            present.zero_()
            for k, mi in enumerate(native_idx):
                present |= ((ws["present_native"] >> k) & 1) << mi
        for mi, name in enumerate(m.modality_types):
            enc = m.encoders[name]
            n, off = self.st.token_dims[mi], self.offsets[mi]
            bm = batch[name]
            if isinstance(enc, EmbeddedSequenceEncoder):
                e = ws["enc"][name]
                toks = bm["tokens"]
                if toks.dtype != torch.float32 or not toks.is_contiguous():
                    toks = toks.float().contiguous()
                if toks.shape != (b, n, enc.input_size):
                    raise AssertionError(f"{name}: tokens {tuple(toks.shape)} != {(b, n, enc.input_size)}")
                e["tokens"] = toks
                te = enc.token_encoder
                rows = b * n
                t2 = toks.view(rows, enc.input_size)
                self.ln_fwd(t2, te[0].weight, rows, enc.input_size, e["m0"], e["r0"], beta=te[0].bias, rowmask=e["mask"],
                            y_bf16=e["xin_b"], cols_pad=self.we[name]["kp"])
                self.gemm_nt(e["xin_b"], self.we[name]["w"], e["y"], rows, D, self.we[name]["kp"], bias=te[1].bias)
                pe = enc.positional_encoder.pe
                self.ln_fwd(e["y"], te[2].weight, rows, D, e["m2"], e["r2"], beta=te[2].bias, rowmask=e["mask"], add=pe,
                            period=n, y=x0[off:], ldy=D, y_bstride=N * D)
            elif isinstance(enc, TabularEncoder):
                self._encode_tabular(name, enc, bm, ws, mi, renorm)
            else:
                # user-registered torch encoder: run it with autograd and feed its tokens to the native trunk
                with torch.enable_grad() if need_grad else torch.no_grad():
                    toks, amask = enc(bm)
                ws["foreign"][name] = toks
                x0.view(b, N, D)[:, off:off + n].copy_(toks.detach().float())
                ws["padding"].view(b, N)[:, off:off + n].copy_(amask.to(torch.bool))
                present |= ((amask == 0).sum(dim=1) != 0).to(torch.int32) << mi
        if self.F:
            call("mca_bcast_rows", ptr(m.fusion_tokens.data), D, x0.data_ptr() + (N - self.F) * D * 4, D, N * D, self.F,
                 b * self.F, D, stream_ptr())
            if foreign:
                ws["padding"].view(b, N)[:, N - self.F:] = 0
        if self.eao:
            # the token block and the padding bytes of a modality, replicated into every combination segment that holds it
            pad2 = ws["padding"].view(b, N)
            for src, dst, n in self.st.copies:
                call("mca_rows_copy_add", x0.data_ptr() + src * D * 4, N * D, x0.data_ptr() + dst * D * 4, N * D, n, D, b, 0, stream_ptr())
                pad2[:, dst:dst + n].copy_(pad2[:, src:src + n])
        ws["present_cur"] = present
        bits = ((present[:, None] >> self._mod_shifts) & 1).to(torch.bool)          # (b, M): one small op for every modality
        return {name: bits[:, mi] for mi, name in enumerate(m.modality_types)}

    def _renorm_table(self, enc, n):
        emb = enc.token_encoder.embedding.weight
        call("mca_embedding_renorm", ptr(emb.data), n, self.D, float(enc.token_encoder.max_norm), stream_ptr())

    def _encode_tabular(self, name, enc, bm, ws, mi, renorm=True):
        """encoders.py:90-96: E[t] (max_norm-renormalised in place) + LN(Linear2(ReLU(Linear1(min(x, max)))))), the value
        part zeroed where x == padding_idx (-1).  The trunk's key-padding mask is the collator's attention_mask."""
        D, N, b = self.D, self.N, ws["b"]
        n, off = self.st.token_dims[mi], self.offsets[mi]
        rows = b * n
        e, ve = ws["enc"][name], enc.value_encoder
        vals = bm["values"]
        if vals.dtype != torch.float32 or not vals.is_contiguous():
            vals = vals.float().contiguous()
        if vals.shape != (b, n):
            raise AssertionError(f"{vals.shape[1]} - {n}")                  # encoders.py:93
        e["values"] = vals
        emb = enc.token_encoder.embedding.weight
        if renorm:
            self._renorm_table(enc, n)
        call("mca_tab_value_fwd", ptr(vals), ptr(ve.linear1.weight.data), ptr(ve.linear1.bias.data), ptr(e["h1_b"]), ptr(e["mask"]),
             rows, D, float(ve.max_value), float(ve.padding_value), stream_ptr())
        self.gemm_nt(e["h1_b"], self.we[name]["w2"], e["y"], rows, D, D, bias=ve.linear2.bias)
        self.ln_fwd(e["y"], ve.norm.weight, rows, D, e["m2"], e["r2"], beta=ve.norm.bias, rowmask=e["mask"], add=emb.data,
                    period=n, y=ws["x"][0][off:], ldy=D, y_bstride=N * D)

    def forward_trunk(self, ws):
        """fusion layers + final norm + attentive pooling -> ws['pooled'] (b*R, D)."""
        m, D, N, H, Ip, R, b, T = self.model, self.D, self.N, self.H, self.Ip, self.R, ws["b"], ws["T"]
        call("mca_build_keyinfo", ptr(ws["padding"]), ptr(self.kgroup), ptr(ws["keyinfo"]), ptr(ws["kflags"]), b, N,
             self.nk_pad, stream_ptr())
        if ws["khot"] is not None:
            call("mca_build_keyhot", ptr(ws["keyinfo"]), ptr(ws["khot"]), b, self.nk_pad, stream_ptr())
        # T >= 2048: the residual LayerNorm(x) is recomputed inside the out-proj / FF2 GEMM epilogues from x and the saved row
        # statistics (mca_gemm_nt_lnres): the LayerNorm kernels then write the bf16 GEMM operand only
        ln_in_gemm = self.fuse_ln_residual and T >= 2048 and D % 128 == 0 and D >= 512 and Ip >= 512
        for i, ly in enumerate(m.layers):
            w, a = self.wl[i], ws["layers"][i]
            xin, xout = ws["x"][i], ws["x"][i + 1]
            g = ly.norm.gamma
            self.ln_fwd(xin, g, T, D, a["m1"], a["r1"], y=None if ln_in_gemm else ws["xn"], ldy=D, y_bf16=a["xn_b"], cols_pad=D)
            self.gemm_nt(a["xn_b"], w["qkv"], a["qkv"], T, 3 * D, D)
            self._attn_fwd(a["qkv"].data_ptr(), N * 3 * D, 3 * D, a["qkv"], D, 2 * D, 3 * D, a["o"], a["lse"],
                           self.qmask_attn, self.sched_attn_f, ws, b, N, layer=i)
            if ln_in_gemm:
                call("mca_gemm_nt_lnres", ptr(a["o"]), D, ptr(w["o"]), D, ptr(a["x1"]), D, ptr(xin), D, ptr(a["m1"]), ptr(a["r1"]),
                     ptr(g.data), T, D, D, stream_ptr(), flops=2.0 * T * D * D)
            else:
                self.gemm_nt(a["o"], w["o"], a["x1"], T, D, D, residual=ws["xn"])
            self.ln_fwd(a["x1"], g, T, D, a["m2"], a["r2"], y=None if ln_in_gemm else ws["x1n"], ldy=D, y_bf16=a["x1n_b"], cols_pad=D)
            # h = x1n @ W1^T and g = GEGLU(h) in one pass (h is kept for the backward, not read back here)
            call("mca_gemm_nt_geglu_fwd", ptr(a["x1n_b"]), D, ptr(w["w1"]), D, ptr(a["h"]), 2 * Ip, ptr(a["g"]), Ip, Ip, T, D,
                 stream_ptr(), flops=2.0 * T * 2 * Ip * D)
            if ln_in_gemm:
                call("mca_gemm_nt_lnres", ptr(a["g"]), Ip, ptr(w["w2"]), Ip, ptr(xout), D, ptr(a["x1"]), D, ptr(a["m2"]), ptr(a["r2"]),
                     ptr(g.data), T, D, Ip, stream_ptr(), flops=2.0 * T * D * Ip)
            else:
                self.gemm_nt(a["g"], w["w2"], xout, T, D, Ip, residual=ws["x1n"])
        xl = ws["x"][self.L]
        if self.eao:
            # final norm (fp32 out), then the mean of every segment's un-padded rows (model.py:563-570, 255-276)
            self.ln_fwd(xl, m.norm.gamma, T, D, ws["mf"], ws["rf"], y=ws["xn"], ldy=D)
            call("mca_segment_mean_fwd", ptr(ws["xn"]), ptr(ws["padding"]), ptr(self.seg_start), R, ptr(ws["pooled"]),
                 ptr(ws["seg_counts"]), b, N, D, stream_ptr())
            return ws["pooled"]
        self.ln_fwd(xl, m.norm.gamma, T, D, ws["mf"], ws["rf"], y_bf16=ws["t_b"], cols_pad=D)
        self.gemm_nt(ws["t_b"], self.wp["kv"], ws["kvp"], T, 2 * D, D)
        call("mca_f32_to_bf16", ptr(m.return_tokens.data), D, ptr(ws["rt_b"]), D, R, D, 1.0, stream_ptr())
        self.gemm_nt(ws["rt_b"], self.wp["q"], ws["qp"], R, D, D)
        self._attn_fwd(ws["qp"].data_ptr(), 0, D, ws["kvp"], 0, D, 2 * D, ws["op"], ws["lse_p"], self.qmask_pool,
                       self.sched_pool_f, ws, b, R)
        self.gemm_nt(ws["op"], self.wp["o"], ws["pooled"], b * R, D, D, residual=m.return_tokens.data, res_period=R)
        return ws["pooled"]

    # ------------------------------------------------------------------------------------------------
    # loss (+ its gradient w.r.t. the local pooled block and the temperature)
    # ------------------------------------------------------------------------------------------------
    def loss_fwd_bwd(self, pooled_all, present_all, b_local, row0):
        B, R, D, T = pooled_all.shape[0], self.R, self.D, self.n_terms
        dev = self.device
        nbytes = hip.lib().mca_contrastive_workspace_bytes(B, T)
        key = ("lossws", B)
        if key not in self._ws:
            self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = dict(term_loss=torch.empty(T, dtype=torch.float32, device=dev), loss=torch.empty(1, dtype=torch.float32, device=dev),
                   d_pooled=torch.empty(b_local, R, D, dtype=torch.float32, device=dev),
                   d_logit=torch.empty(1, dtype=torch.float32, device=dev))
        ls = self.model.loss.loss_fn.logit_scale
        call("mca_contrastive_fwd_bwd", ptr(pooled_all), ptr(present_all), ptr(self.loss_terms_dev), T, ptr(ls.data), B, b_local,
             row0, R, D, ptr(out["term_loss"]), ptr(out["loss"]), ptr(out["d_pooled"]), ptr(out["d_logit"]),
             ptr(self._ws[key]), stream_ptr())
        return out

    # ------------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------------
    def _on_side(self, fn, slot: int, ws: dict):
        """Run fn() on the workspace's side stream, ordered after everything enqueued so far on the current stream."""
        if not self.overlap_on(ws):
            fn()
            return
        side, events = ws["side"], ws["side_events"]
        while len(events) <= slot:
            events.append(torch.cuda.Event())
        ev = events[slot]
        ev.record()
        with torch.cuda.stream(side), hip.use_stream(side.cuda_stream):
            side.wait_event(ev)
            fn()

    OVERLAP_ROWS_MAX = 40960

    def overlap_on(self, ws) -> bool:
        """weight-gradient GEMMs of this workspace on its side stream?"""
        if self.overlap_wgrad is None:
            return ws["T"] < self.OVERLAP_ROWS_MAX
        return bool(self.overlap_wgrad)

    def _bucket_ready(self, idx):
        if self.grad_bucket_hook is not None:
            lo = 0 if idx == 0 else self.bucket_bounds[idx - 1]
            self.grad_bucket_hook(lo, self.bucket_bounds[idx])

    def backward(self, ws, d_pooled, d_logit_scale=None, accumulate=False):
        """d_pooled: (b, R, D) fp32 gradient of the objective w.r.t. the pooled tokens.  Writes every parameter
        gradient into the flat gradient buffer."""
        if not accumulate:
            self.gflat.zero_()
        if d_logit_scale is not None:
            self.grad_of(self.model.loss.loss_fn.logit_scale).add_(d_logit_scale.reshape(()))
        self._backward_part(ws, d_pooled, self._bucket_ready)
        if self.overlap_on(ws):
            torch.cuda.current_stream().wait_stream(ws["side"])          # every weight gradient is in before clip / AdamW

    def _backward_part(self, ws, d_pooled, bucket_ready):
        m, D, N, H, Ip, I, R, b, T = self.model, self.D, self.N, self.H, self.Ip, self.I, self.R, ws["b"], ws["T"]
        G = self.grad_of
        dpool = d_pooled.reshape(b * R, D).contiguous()
        ap = m.attn_pool
        # Weight-gradient GEMMs (mca_gemm_tn_acc) only feed the optimizer: they are issued on a side stream, ordered after
        # the kernel that produced their operand, and run concurrently with the rest of the backward chain (their
        # operands live in per-layer buffers, so nothing overwrites them).  slot = unique id of the call site.
        side, tn = self._on_side, self.gemm_tn_acc
        slot = [0]

        def on_side(fn):
            side(fn, slot[0], ws); slot[0] += 1

        if self.eao:
            return self._backward_part_eao(ws, dpool, bucket_ready, on_side)
        # pooled = op @ Wo^T + return_tokens
        call("mca_reduce_rows", ptr(dpool), D, R * D, R, ptr(G(m.return_tokens)), D, b * R, D, stream_ptr())
        call("mca_f32_to_bf16", ptr(dpool), D, ptr(ws["dpool_b"]), D, b * R, D, 1.0, stream_ptr())
        self.gemm_nt(ws["dpool_b"], self.wp["oT"], ws["dop"], b * R, D, D)
        on_side(lambda: tn(ws["dpool_b"], ws["op"], G(ap.to_out.weight), b * R, D, D))
        # pooling attention
        self._attn_bwd2(ws["qp"].data_ptr(), 0, D, ws["kvp"], 0, D, 2 * D, ws["op"], ws["dop"], ws["lse_p"], ws["delta_p"],
                        ws["dqp32"].data_ptr(), R * D, D, True, ws["dkvp"], 0, D, 2 * D, self.qmask_pool, self.sched_pool_f,
                        self.sched_pool_b2, ws, b, R)
        ws["dqp_sum"].zero_()
        call("mca_reduce_rows", ptr(ws["dqp32"]), D, R * D, R, ptr(ws["dqp_sum"]), D, b * R, D, stream_ptr())
        call("mca_f32_to_bf16", ptr(ws["dqp_sum"]), D, ptr(ws["dqp_b"]), D, R, D, 1.0, stream_ptr())
        self.gemm_nt(ws["dqp_b"], self.wp["qT"], ws["drt"], R, D, D)                 # d return_tokens via to_q
        call("mca_reduce_rows", ptr(ws["drt"]), D, R * D, R, ptr(G(m.return_tokens)), D, R, D, stream_ptr())
        on_side(lambda: tn(ws["dqp_b"], ws["rt_b"], G(ap.to_q.weight), R, D, D))
        pool_kv = (ws["dkvp"], ws["t_b"], G(ap.to_kv.weight), 2 * D, D)
        if not (self.L > 0 and self.group_wgrad and T >= 4096):          # else: a member of the top layer's grouped launch
            on_side(lambda pk=pool_kv: tn(pk[0], pk[1], pk[2], T, 2 * D, D))
            pool_kv = None
        dx, dx_other = ws["dxa"], ws["dxb"]
        self.gemm_nt(ws["dkvp"], self.wp["kvT"], dx, T, D, 2 * D)                    # d (final-normed tokens), fp32
        top = ws["layers"][self.L - 1]["dxo_b"] if self.L else ws["dx_b"]
        self.ln_bwd(dx, D, ws["x"][self.L], m.norm.gamma, ws["mf"], ws["rf"], T, D, G(m.norm.gamma), dx=dx_other, dx_bf16=top)
        dx, dx_other = dx_other, dx
        on_side(lambda: bucket_ready(0))
        self._backward_layers_and_encoders(ws, dx, dx_other, bucket_ready, on_side, extra_top=pool_kv)

    def _backward_part_eao(self, ws, dpool, bucket_ready, on_side):
        """EAO: d pooled (b, segments, D) -> mean-pool backward -> final norm backward -> the shared layer / encoder chain."""
        m, D, N, R, b, T = self.model, self.D, self.N, self.R, ws["b"], ws["T"]
        dx, dx_other = ws["dxa"], ws["dxb"]
        call("mca_segment_mean_bwd", ptr(dpool), ptr(ws["padding"]), ptr(self.kgroup), ptr(ws["seg_counts"]), R, ptr(dx), b, N, D,
             stream_ptr())
        top = ws["layers"][self.L - 1]["dxo_b"] if self.L else ws["dx_b"]
        self.ln_bwd(dx, D, ws["x"][self.L], m.norm.gamma, ws["mf"], ws["rf"], T, D, self.grad_of(m.norm.gamma), dx=dx_other, dx_bf16=top)
        dx, dx_other = dx_other, dx
        on_side(lambda: bucket_ready(0))
        self._backward_layers_and_encoders(ws, dx, dx_other, bucket_ready, on_side)

    def _backward_layers_and_encoders(self, ws, dx, dx_other, bucket_ready, on_side, extra_top=None):
        m, D, N, H, Ip, I, R, b, T = self.model, self.D, self.N, self.H, self.Ip, self.I, self.R, ws["b"], ws["T"]
        G = self.grad_of
        tn = self.gemm_tn_acc
        for bi, i in enumerate(reversed(range(self.L))):
            ly, w, a = m.layers[i], self.wl[i], ws["layers"][i]
            g = ly.norm.gamma
            dxo, dx1, dh, dqkv = a["dxo_b"], a["dx1_b"], a["dh"], a["dqkv"]
            below = ws["layers"][i - 1]["dxo_b"] if i > 0 else ws["dx_b"]
            # x_out = g @ W2^T + x1n            (dx = d x_out fp32, dxo = its bf16 copy)
            # The five weight gradients of the layer reduce over the same T rows: grouped into one launch after the layer's
            # attention backward (52 tiles of 256 x 256 -> 5 row splits instead of 21-32 per gradient: a quarter of the fp32 atomic
            # traffic; the launch's balanced row partition fills all CUs for any tile count, gemm.hip struct tn_group).
            grouped = self.group_wgrad and T >= 4096
            if not grouped:
                on_side(lambda dxo=dxo, a=a, ly=ly: tn(dxo, a["g"], G(ly.ff.feedforward[2].weight), T, D, I))
            if self.fuse_geglu_bwd:
                # dh = GEGLU'(h) * (dx @ W2): the (T, Ip) intermediate dg is never written (fused GEMM epilogue)
                call("mca_gemm_nt_geglu_bwd", ptr(dxo), D, ptr(w["w2T"]), D, ptr(a["h"]), ptr(dh), 2 * Ip, Ip, T, D,
                     stream_ptr(), flops=2.0 * T * Ip * D)
            else:
                self.gemm_nt(dxo, w["w2T"], ws["dg"], T, Ip, D)
                call("mca_geglu_bwd", ptr(ws["dg"]), ptr(a["h"]), ptr(dh), T, Ip, stream_ptr())
            gw1 = G(ly.ff.feedforward[0].weight)
            if not grouped:
                on_side(lambda dh=dh, a=a, gw1=gw1: (tn(dh, a["x1n_b"], gw1, T, I, D), tn(dh[:, Ip:], a["x1n_b"], gw1[I:], T, I, D)))
            self.gemm_nt(dh, w["w1T"], dx_other, T, D, 2 * Ip, residual=dx)            # d x1n = dh @ W1 + dx
            self.ln_bwd(dx_other, D, a["x1"], g, a["m2"], a["r2"], T, D, G(g), dx=dx, dx_bf16=dx1)   # dx = d x1
            # x1 = o @ Wo^T + xn
            if not grouped:
                on_side(lambda dx1=dx1, a=a, ly=ly: tn(dx1, a["o"], G(ly.attn.to_out.weight), T, D, D))
            self.gemm_nt(dx1, w["oT"], ws["do"], T, D, D)
            # dq | dk | dv land in dqkv as bf16, each element written once
            self._attn_bwd2(a["qkv"].data_ptr(), N * 3 * D, 3 * D, a["qkv"], D, 2 * D, 3 * D, a["o"], ws["do"], a["lse"],
                            ws["delta"], dqkv.data_ptr(), N * 3 * D, 3 * D, False, dqkv, D, 2 * D, 3 * D, self.qmask_attn,
                            self.sched_attn_f, self.sched_attn_b2, ws, b, N, layer=i)
            # to_q.weight and to_kv.weight are adjacent in the flat gradient buffer: one (3D, D) weight-gradient GEMM
            gq = G(ly.attn.to_q.weight)
            assert G(ly.attn.to_kv.weight).data_ptr() == gq.data_ptr() + D * D * 4
            if grouped:
                gw2, gwo = G(ly.ff.feedforward[2].weight), G(ly.attn.to_out.weight)
                extra = [extra_top] if (bi == 0 and extra_top is not None) else []
                on_side(lambda dqkv=dqkv, dh=dh, dxo=dxo, dx1=dx1, a=a, gq=gq, gw1=gw1, gw2=gw2, gwo=gwo, extra=extra: self.gemm_tn_acc_group(
                    [(dqkv, a["xn_b"], gq, 3 * D, D), (dh, a["x1n_b"], gw1, I, D), (dh[:, Ip:], a["x1n_b"], gw1[I:], I, D),
                     (dxo, a["g"], gw2, D, I), (dx1, a["o"], gwo, D, D)] + extra, T))
            else:
                on_side(lambda dqkv=dqkv, a=a, gq=gq: tn(dqkv, a["xn_b"], gq, T, 3 * D, D))
            self.gemm_nt(dqkv, w["qkvT"], dx_other, T, D, 3 * D, residual=dx)          # d xn = dqkv @ Wqkv + d x1
            self.ln_bwd(dx_other, D, ws["x"][i], g, a["m1"], a["r1"], T, D, G(g), dx=dx, dx_bf16=below)  # dx = d x_in
            on_side(lambda bi=bi: bucket_ready(bi + 1))
        # dx = gradient w.r.t. the packed encoder output (b, N, D)
        if self.eao:          # the gradients of a modality's replicas, summed into the segment its encoder wrote
            for src, dst, n in self.st.copies:
                call("mca_rows_copy_add", dx.data_ptr() + dst * D * 4, N * D, dx.data_ptr() + src * D * 4, N * D, n, D, b, 1, stream_ptr())
        if self.F:
            call("mca_reduce_rows", dx.data_ptr() + (N - self.F) * D * 4, D, N * D, self.F, ptr(G(m.fusion_tokens)), D,
                 b * self.F, D, stream_ptr())
        for mi, name in enumerate(m.modality_types):
            enc = m.encoders[name]
            n, off = self.st.token_dims[mi], self.offsets[mi]
            rows = b * n
            if isinstance(enc, EmbeddedSequenceEncoder):
                e, te = ws["enc"][name], enc.token_encoder
                kp = self.we[name]["kp"]
                # (the Linear's bias gradient = column sums of this norm's dx: same launch)
                self.ln_bwd(dx[off:], D, e["y"], te[2].weight, e["m2"], e["r2"], rows, D, G(te[2].weight), dbeta=G(te[2].bias),
                            rowmask=e["mask"], dx=e["dy"], dx_bf16=e["dy_b"], y_bstride=N * D, period=n, dxsum=G(te[1].bias))
                on_side(lambda e=e, te=te, rows=rows, enc=enc: tn(e["dy_b"], e["xin_b"], G(te[1].weight), rows, D, enc.input_size))
                self.gemm_nt(e["dy_b"], self.we[name]["wT"], e["dxin"], rows, kp, D)
                t2 = e["tokens"].view(rows, enc.input_size)
                self.ln_bwd(e["dxin"], kp, t2, te[0].weight, e["m0"], e["r0"], rows, enc.input_size, G(te[0].weight),
                            dbeta=G(te[0].bias), rowmask=e["mask"])
            elif isinstance(enc, TabularEncoder):
                self._backward_tabular(name, enc, ws, mi, dx)
            else:
                toks = ws["foreign"][name]
                if toks.requires_grad:
                    # the foreign encoder's parameters live in the flat buffers too: point their .grad at the flat views so
                    # that autograd ACCUMULATES in place (FusedAdamW.zero_grad leaves None, and a fresh .grad tensor would be
                    # replaced by the zeroed flat view after this backward)
                    for p in enc.parameters():
                        p.grad = G(p)
                    torch.autograd.backward(toks, dx.view(b, N, D)[:, off:off + n].to(toks.dtype))
        on_side(lambda: bucket_ready(len(self.bucket_bounds) - 1))

    def _backward_tabular(self, name, enc, ws, mi, dx):
        D, N, b = self.D, self.N, ws["b"]
        n, off = self.st.token_dims[mi], self.offsets[mi]
        rows = b * n
        e, ve, G = ws["enc"][name], enc.value_encoder, self.grad_of
        gemb = G(enc.token_encoder.embedding.weight)
        # the table is added after the value path is masked: every row of dx reaches it; padding_idx row stays frozen
        call("mca_reduce_rows", dx.data_ptr() + off * D * 4, D, N * D, n, ptr(gemb), D, rows, D, stream_ptr())
        gemb[n - 1].zero_()
        self.ln_bwd(dx[off:], D, e["y"], ve.norm.weight, e["m2"], e["r2"], rows, D, G(ve.norm.weight), dbeta=G(ve.norm.bias),
                    rowmask=e["mask"], dx=e["dy"], dx_bf16=e["dy_b"], y_bstride=N * D, period=n, dxsum=G(ve.linear2.bias))
        self._on_side(lambda: self.gemm_tn_acc(e["dy_b"], e["h1_b"], G(ve.linear2.weight), rows, D, D), 200 + mi, ws)
        self.gemm_nt(e["dy_b"], self.we[name]["w2T"], e["dh1"], rows, D, D)
        call("mca_tab_value_bwd", ptr(e["dh1"]), D, ptr(e["h1_b"]), ptr(e["values"]), ptr(G(ve.linear1.weight)),
             ptr(G(ve.linear1.bias)), rows, D, float(ve.max_value), stream_ptr())

    # ------------------------------------------------------------------------------------------------
    # model-level forward (autograd node)
    # ------------------------------------------------------------------------------------------------
    def model_forward(self, batch, no_loss=False):
        with hip.cached_stream():
            return self._model_forward(batch, no_loss)

    def can_forward_backward(self) -> bool:
        m = self.model
        return all(isinstance(m.encoders[n], (EmbeddedSequenceEncoder, TabularEncoder)) for n in m.modality_types)

    def forward_backward(self, batch):
        """forward + loss + backward WITHOUT autograd: the same kernels as ``model(batch)`` followed by ``loss.backward()``, issued
        from the calling thread (autograd runs a CUDA backward on its device thread; a stream capture that is cut into segments
        at the data-parallel collectives must begin and end its captures on one thread: graph.GraphedStep).  Every ``p.grad`` is
        the view of the flat gradient buffer.  Returns the output dict of the forward."""
        m = self.model
        if not self.can_forward_backward():
            raise NotImplementedError("forward_backward needs native encoders (a torch encoder's backward runs under autograd)")
        with hip.cached_stream(), torch.no_grad():
            self._last_direct = None
            out = self._model_forward(batch, no_loss=False)
            ws, res = self._last_direct
            self.backward(ws, res["d_pooled"], res["d_logit"], accumulate=False)
        for p in self.param_order:
            p.grad = self.grad_of(p)
        return out

    def _model_forward(self, batch, no_loss=False):
        m = self.model
        first = batch[m.modality_types[0]]
        b = next(iter(first.values())).shape[0]
        need_grad = torch.is_grad_enabled() and not no_loss
        if self.check_finite:
            if not torch.cuda.is_current_stream_capturing():           # (hipEventQuery would invalidate a stream capture)
                self.poll_finite()                                     # a flagged EARLIER step raises here (no sync)
            self._flag_inputs(batch)
        self.refresh_weights()
        ws = self.workspace(b)
        ws["gen"] += 1          # the saved activations of any earlier forward of this batch size are gone from here on
        sample_mask = self._encode(batch, ws, need_grad)
        pooled = self.forward_trunk(ws).view(b, self.R, self.D)
        slots = m.output_slots()
        if self.check_finite:
            self._flag_tensors([pooled], 2)
            call("mca_flag_to_host", ptr(self.finite_flag), self._flag_host.data_ptr(), stream_ptr())          # (a kernel, not a copy node)
            if not torch.cuda.is_current_stream_capturing():          # (a captured step: graph.GraphedStep records it after the replay)
                self._flag_event = torch.cuda.Event()
                self._flag_event.record()
        if no_loss:
            out = {k: pooled[:, s] for k, s in slots.items()}
            out["modality_sample_mask"] = sample_mask
            if self.check_finite and self.check_finite != "deferred":
                self.assert_finite()
            return out
        # in-place clamp of the temperature parameter (utils/contrastive_loss_with_temperature.py:187)
        ls = m.loss.loss_fn
        ls.logit_scale.data.clamp_(ls.logit_scale_min, ls.logit_scale_max)
        present = ws["present_cur"]
        if need_grad:
            pooled_out, terms, loss = _MCAStep.apply(self, ws, pooled, present, *self.param_order)
        else:
            pooled_out, terms, loss, res = _loss_forward(self, pooled, present)
            self._last_direct = (ws, res)          # forward_backward(): the backward chain without an autograd node
        out = {k: pooled_out[:, s] for k, s in slots.items()}
        names = [t.name for t in m.loss_terms]
        out["losses"] = {n: terms[i] for i, n in enumerate(names)}
        if m.fcl and not m.zorro:
            if getattr(self, "_fc_w", None) is None:          # (2, terms) averaging weights, built once: three small kernels per step
                fc = torch.tensor([1.0 if "fcl" in n else 0.0 for n in names], dtype=torch.float32, device=self.device)
                self._fc_w = torch.stack([fc / fc.sum().clamp(min=1), (1 - fc) / (1 - fc).sum().clamp(min=1)])
            both = (torch.nan_to_num(terms)[None, :] * self._fc_w).sum(1)
            out["fcl_loss"], out["no-fcl_loss"] = both[0], both[1]
        out["loss"] = loss.reshape(())
        out["modality_sample_mask"] = sample_mask
        if self.check_finite and self.check_finite != "deferred":
            self.assert_finite()
        return out

    # ---- finite flag ---------------------------------------------------------------------------------
    def _flag_tensors(self, tensors, bit: int):
        fa = hip.FiniteArgs()
        keep = []
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            keep.append(t)
            fa.p[len(keep) - 1], fa.n[len(keep) - 1] = t.data_ptr(), t.numel()
        fa.count = len(keep)
        if keep:
            call("mca_nonfinite_flag", C.byref(fa), ptr(self.finite_flag), bit, stream_ptr())

    def _flag_inputs(self, batch):
        """bit 0: a non-finite value anywhere in an encoder input (encoders.py:197-198 checks the whole `tokens` tensor,
        padded positions included); one launch for every modality."""
        ts = [v[k] for v in batch.values() if isinstance(v, dict) for k in ("tokens", "values") if k in v and torch.is_tensor(v[k])]
        self._flag_tensors(ts[:hip.MAX_MODALITIES], 1)

    def _raise_flag(self, bits: int):
        self.finite_flag.zero_()
        self._flag_host.zero_()
        self._flag_event = None
        if bits & 1:
            raise Exception("Tokens are not finite")                                      # encoders.py:197-198
        raise Exception("Encoder transform / fusion resulted in non-finite values")       # encoders.py:206-213

    def poll_finite(self):
        """Non-blocking: raises if a step whose flag copy has already landed on the host saw non-finite values."""
        ev = self._flag_event
        if ev is not None and ev.query():
            bits = int(self._flag_host[0])
            if bits:
                self._raise_flag(bits)

    def assert_finite(self):
        """Blocking form: waits for the last step's flag copy and raises if it is set."""
        ev = self._flag_event
        if ev is not None:
            ev.synchronize()
            bits = int(self._flag_host[0])
            if bits:
                self._raise_flag(bits)


def _loss_forward(engine, pooled, present):
    """(optional all-gather of pooled embeddings + presence bits) then the fused loss kernels."""
    b = pooled.shape[0]
    if engine.gather_hook is not None:
        pooled_all, present_all, row0 = engine.gather_hook(pooled, present)
    else:
        pooled_all, present_all, row0 = pooled, present, 0
    res = engine.loss_fwd_bwd(pooled_all.contiguous(), present_all.contiguous(), b, row0)
    pooled_out = torch.empty_like(pooled)          # the embeddings handed to the caller: copied by a kernel (no copy node in a captured step)
    flat_in, flat_out = pooled.reshape(-1, pooled.shape[-1]), pooled_out.view(-1, pooled.shape[-1])
    call("mca_rows_copy_add", ptr(flat_in), 0, ptr(flat_out), 0, flat_in.shape[0], flat_in.shape[1], 1, 0, stream_ptr())
    return pooled_out, res["term_loss"], res["loss"], res


class _MCAStep(torch.autograd.Function):
    """One autograd node for the whole step.  forward: (optional all-gather) + loss kernels, which also
    produce d loss/d pooled; backward: the engine's backward chain, gradients written straight into the flat
    gradient buffer (each parameter's ``.grad`` is a view of it)."""

    @staticmethod
    def forward(ctx, engine, ws, pooled, present, *params):
        pooled_out, terms, loss, res = _loss_forward(engine, pooled, present)
        ctx.engine, ctx.ws, ctx.res, ctx.gen = engine, ws, res, ws["gen"]
        ctx.mark_non_differentiable(terms)
        return pooled_out, terms, loss

    @staticmethod
    def backward(ctx, g_pooled, g_terms, g_loss):
        engine, ws, res = ctx.engine, ctx.ws, ctx.res
        if ws["gen"] != ctx.gen:
            raise RuntimeError(
                "MCA.backward: another forward of the same batch size ran after the forward this loss came from; the step keeps "
                "ONE set of saved activations per batch size, so run backward() before the next forward (train or eval)")
        d_pooled = res["d_pooled"] * g_loss.reshape(())
        if g_pooled is not None:
            d_pooled = d_pooled + g_pooled
        params = engine.param_order
        live = params[0].grad is not None and params[0].grad.data_ptr() == engine.grad_of(params[0]).data_ptr()
        with hip.cached_stream():
            engine.backward(ws, d_pooled, res["d_logit"] * g_loss.reshape(()), accumulate=live)
        for p in params:
            p.grad = engine.grad_of(p)
        return (None, None, None, None) + tuple(None for _ in params)

"""``MCA(**model_config)``: the reference's model surface (model.py:282-478) over the native engine.

What is kept from the reference: constructor keywords, module tree and therefore
state_dict keys (incl. the persistent ``attn_mask`` / ``pool_mask`` / ``token_types`` / ``fusion_mask``
buffers), parameter initialisation order (same ``torch.manual_seed`` -> same weights), the call
``model(batch, no_loss=False) -> dict`` with the same keys, and its error behaviour (Python exceptions on
non-finite encoder inputs/outputs; NaN loss terms are not errors).

What is different underneath: ``forward`` is ONE autograd node.  All arithmetic — encoders, 5 fusion
layers, attentive pooling, all-pairs contrastive loss, and the whole backward — runs in the HIP kernels of
``libmca_hip.so`` driven by ``engine.FusionEngine``; parameters and gradients live in two flat fp32
buffers (fused clip+AdamW, one bucketed all-reduce).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from .encoders import encoders_dict, NativeEncoder
from .structure import FusionStructure, EAOStructure, loss_terms, FUSION_TOKEN, GLOBAL_TOKEN


class LayerNorm(nn.Module):
    """gamma is learned, beta is a zero buffer (model.py:24-31)."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))


class FeedForward(nn.Module):
    """Linear(dim, 2I) -> GEGLU -> Linear(I, dim), I = int(dim*mult*2/3), no biases (model.py:41-54)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        inner = int(dim * mult * 2 / 3)
        self.inner_dim = inner
        self.feedforward = nn.Sequential(nn.Linear(dim, inner * 2, bias=False), nn.Identity(),
                                         nn.Linear(inner, dim, bias=False))


class Attention(nn.Module):
    """to_q / to_kv / to_out, no biases (model.py:57-71)."""

    def __init__(self, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner = dim_head * heads
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


class MCALayer(nn.Module):
    def __init__(self, dim, dim_head, heads, ff_mult):
        super().__init__()
        self.attn = Attention(dim, dim_head, heads)
        self.ff = FeedForward(dim, ff_mult)
        self.norm = LayerNorm(dim)


class _Temperature(nn.Module):
    """``loss.loss_fn.logit_scale``: log-temperature of the contrastive loss, init ln(1/0.07), clamped to
    [ln 1, ln 100] in place on every forward (utils/contrastive_loss_with_temperature.py:111,187)."""

    def __init__(self):
        super().__init__()
        self.logit_scale = nn.Parameter(math.log(1 / 0.07) * torch.ones([]))
        self.logit_scale_min, self.logit_scale_max = math.log(1), math.log(100)


class MCAPretrainingLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.loss_fn = _Temperature()


class MCA(nn.Module):
    def __init__(self, encoder_configs, dim, depth, dim_head=64, heads=8, ff_mult=4, num_fusion_tokens=16,
                 batch_size=8, return_padding=False, return_logits=False, bimodal_contrastive=False,
                 non_fusion_fcl=False, fcl=False, fcl_root=(1, 2, 3, 4, 5), fusion_combos=(4, 5), zorro=False,
                 no_fusion=False, mean_pool=False, **kwargs):
        super().__init__()
        if mean_pool:
            raise NotImplementedError("mean_pool=True (MeanTokenProjectionPool) is outside the native hot path (SURVEY.md §2 #14)")
        if dim_head != 64:
            raise NotImplementedError("the gfx950 attention kernels are specialised for dim_head = 64")
        self.extra_kwargs = dict(kwargs)
        self.batch_size = batch_size
        self.dim, self.depth, self.heads, self.dim_head = dim, depth, heads, dim_head
        self.bimodal_contrastive, self.non_fusion_fcl = bimodal_contrastive, non_fusion_fcl
        self.modality_types = list(encoder_configs.keys())
        self.token_dims = [encoder_configs[m]["max_tokens"] for m in self.modality_types]
        self.structure = FusionStructure(self.token_dims, num_fusion_tokens, tuple(fusion_combos), fcl=fcl,
                                         zorro=zorro, no_fusion=no_fusion)
        st = self.structure
        self.fusion_combos = st.combos
        self.no_fusion, self.zorro, self.fcl = no_fusion, zorro, fcl
        self.fcl_root = frozenset(fcl_root) if (fcl and not zorro and not no_fusion) else None
        self.num_fusion_tokens = st.num_fusion_tokens
        self.return_token_types = st.return_token_types
        self.max_return_tokens = st.n_return
        self.register_buffer("return_token_types_tensor", torch.tensor(st.return_token_types), persistent=False)

        # ---- parameters, created in the reference's order so that the same seed gives the same weights
        self.encoders = nn.ModuleDict({name: encoders_dict[cfg["type"]](**cfg) for name, cfg in encoder_configs.items()})
        for name, enc in self.encoders.items():
            ed = getattr(enc, "embedding_dim", dim)
            if ed != dim:
                raise ValueError(f"encoder {name}: embedding_dim {ed} != model dim {dim}")
            # TabularEncoder: the table is indexed by arange(max_tokens) (encoders.py:80-94); the kernels renormalise and read
            # max_tokens rows and freeze row max_tokens - 1 (padding_idx = -1), so the table must have exactly that many rows
            ne = getattr(enc, "num_embeddings", None)
            if getattr(enc, "kind", "") == "tabular" and ne is not None and ne != encoder_configs[name]["max_tokens"]:
                raise ValueError(f"encoder {name}: num_embeddings {ne} != max_tokens {encoder_configs[name]['max_tokens']}")
        self.fusion_tokens = nn.Parameter(torch.randn(st.num_fusion_tokens, dim))
        self.register_buffer("fusion_mask", torch.zeros(st.num_fusion_tokens, dtype=torch.bool))
        self.layers = nn.ModuleList([MCALayer(dim, dim_head, heads, ff_mult) for _ in range(depth)])
        self.norm = LayerNorm(dim)
        self.register_buffer("token_types", torch.from_numpy(st.token_types.copy()))
        self.return_tokens = nn.Parameter(torch.randn(st.n_return, dim))
        self.attn_pool = Attention(dim, dim_head, heads)
        self.register_buffer("attn_mask", torch.from_numpy(st.dense_attn_mask()))
        self.register_buffer("pool_mask", torch.from_numpy(st.dense_pool_mask()))
        self.loss = MCAPretrainingLoss()

        self.loss_terms = loss_terms(self.modality_types, st, bimodal_contrastive, non_fusion_fcl)
        self._engine = None
        self._dp_wrapper = None
        # load_state_dict writes the parameters in place: the bf16 GEMM-weight copies of the engine must follow
        self.register_load_state_dict_post_hook(MCA._after_load_state_dict)

    @staticmethod
    def _after_load_state_dict(module, incompatible_keys):
        if module._engine is not None:
            module._engine.invalidate_weights()

    # ---- engine ------------------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            from .engine import FusionEngine
            self._engine = FusionEngine(self)
            dp = getattr(self, "_dp_wrapper", None)
            if dp is not None:          # the data-parallel hooks live on the engine: a rebuilt engine gets them again
                dp.install(self._engine)
        return self._engine

    def _apply(self, fn, *args, **kwargs):
        # moving / casting the module invalidates the flat buffers
        self._engine = None
        return super()._apply(fn, *args, **kwargs)

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, batch, no_loss: bool = False):
        """Same contract as the reference's ``MCA.forward`` (model.py:448-478)."""
        return self.engine.model_forward(batch, no_loss=no_loss)

    # ---- names of the pooled slots (model.py:181-191) ------------------------------------------------
    def output_slots(self) -> Dict[object, int]:
        M = len(self.modality_types)
        slots: Dict[object, int] = {m: i for i, m in enumerate(self.modality_types)}
        if self.fcl and not self.zorro:
            for c, combo in enumerate(self.fusion_combos):
                slots[combo] = M + c
            if not self.no_fusion:
                slots["fusion"] = M
        elif not self.no_fusion:
            slots["fusion"] = M
        return slots


class EAO(MCA):
    """``EAO(**model_config)``: the paper's "everything at once" baseline (model.py:481-596) on the native engine.

    The reference runs its layer stack once per modality and once per modality combination and mean-pools each pass
    (``mean_pool=True``: ``MeanTokenProjectionPool(None, projection=False)``, model.py:530-532, 255-276); here the passes are the
    segments of one block-diagonal super-sequence (``structure.EAOStructure``), so the whole model is ONE pass of the same HIP
    kernels.  Same constructor keywords, module tree and state_dict keys (``encoders.*``, ``layers.*``, ``norm.*``,
    ``token_types``, ``loss.loss_fn.logit_scale``); module creation order = the reference's, so the same seed gives the same
    weights.  As in the reference (model.py:500-503) ``num_fusion_tokens`` is forced to 0, ``no_fusion`` to its default True
    and ``fcl_root`` to the first combination.  ``mean_pool=False`` is refused: the reference's own EAO cannot run it
    (``self.pool_mask`` is never defined, model.py:566)."""

    def __init__(self, encoder_configs, dim, depth, dim_head=64, heads=8, ff_mult=4, num_fusion_tokens=16,
                 batch_size=8, return_padding=False, return_logits=False, bimodal_contrastive=False,
                 non_fusion_fcl=False, fcl=False, fcl_root=(1, 2, 3, 4, 5), fusion_combos=(4, 5), zorro=False,
                 no_fusion=True, mean_pool=True, **kwargs):
        nn.Module.__init__(self)
        if not mean_pool:
            raise NotImplementedError("EAO(mean_pool=False): the reference's EAO reads self.pool_mask, which it never defines "
                                      "(model.py:566): only the mean-pooled form exists")
        if dim_head != 64:
            raise NotImplementedError("the gfx950 attention kernels are specialised for dim_head = 64")
        self.extra_kwargs = dict(kwargs)
        self.batch_size = batch_size
        self.dim, self.depth, self.heads, self.dim_head = dim, depth, heads, dim_head
        self.bimodal_contrastive, self.non_fusion_fcl = bimodal_contrastive, non_fusion_fcl
        self.modality_types = list(encoder_configs.keys())
        self.token_dims = [encoder_configs[m]["max_tokens"] for m in self.modality_types]
        self.structure = EAOStructure(self.token_dims, tuple(fusion_combos), fcl=fcl, zorro=zorro)
        st = self.structure
        self.fusion_combos = st.combos
        self.no_fusion, self.zorro, self.fcl = True, zorro, fcl          # the loss is built with no_fusion (model.py:538)
        self.fcl_root = None
        self.num_fusion_tokens = 0
        self.fusion_token = -1
        self.return_token_types = st.return_token_types
        self.max_return_tokens = len(st.return_token_types)
        self.register_buffer("return_token_types_tensor", torch.tensor(st.return_token_types), persistent=False)
        self.encoders = nn.ModuleDict({name: encoders_dict[cfg["type"]](**cfg) for name, cfg in encoder_configs.items()})
        for name, enc in self.encoders.items():
            ed = getattr(enc, "embedding_dim", dim)
            if ed != dim:
                raise ValueError(f"encoder {name}: embedding_dim {ed} != model dim {dim}")
            ne = getattr(enc, "num_embeddings", None)
            if getattr(enc, "kind", "") == "tabular" and ne is not None and ne != encoder_configs[name]["max_tokens"]:
                raise ValueError(f"encoder {name}: num_embeddings {ne} != max_tokens {encoder_configs[name]['max_tokens']}")
        self.layers = nn.ModuleList([MCALayer(dim, dim_head, heads, ff_mult) for _ in range(depth)])
        self.norm = LayerNorm(dim)
        self.register_buffer("token_types", torch.from_numpy(st.token_types.copy()))
        self.return_tokens = None
        self.fusion_tokens = None
        self.attn_pool = None                      # MeanTokenProjectionPool(None, projection=False): no parameters
        self.loss = MCAPretrainingLoss()
        self.loss_terms = loss_terms(self.modality_types, st, bimodal_contrastive, non_fusion_fcl)
        self._engine = None
        self._dp_wrapper = None
        self.register_load_state_dict_post_hook(MCA._after_load_state_dict)

    def output_slots(self) -> Dict[object, int]:
        """model.py:181-189 with no_fusion: the modalities, and the combinations when the fusion-channel loss is on."""
        M = len(self.modality_types)
        slots: Dict[object, int] = {m: i for i, m in enumerate(self.modality_types)}
        if self.fcl and not self.zorro:
            for c, combo in enumerate(self.fusion_combos):
                slots[combo] = M + c
        return slots

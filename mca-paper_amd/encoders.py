"""Modality encoders and collators: the reference's plugin surface (encoders.py:277-283 ``encoders_dict``,
:367-371 ``collators``, :374-403 ``MultimodalCollator``).

The encoder classes here are PARAMETER CONTAINERS with the reference's module tree (so state_dict keys and
``torch.manual_seed`` initialisation are identical); their arithmetic runs in the HIP kernels driven by
``engine.FusionEngine``.  A modality whose ``type`` is registered by the user with an ordinary
``nn.Module`` (any class not derived from ``NativeEncoder``) still works: the engine runs it with torch
and feeds its tokens to the native trunk.

Encoder contract (encoders.py:196-214, :90-96): ``forward(batch_for_modality) -> (tokens (b,n,D),
attention_mask (b,n))`` with non-zero / True = padded position.
"""
from __future__ import annotations

import math
from collections import defaultdict
from typing import Dict, List, Optional

import torch
from torch import nn
from torch.nn.functional import pad


class NativeEncoder(nn.Module):
    """Marker base class: the engine has a fused HIP path for this encoder type."""
    kind = ""

    def forward(self, batch):
        raise NotImplementedError("a native encoder is a parameter container: its arithmetic is fused into the model's step "
                                  "(construct an MCA / EAO model and call it)")


class PositionalEncoder(nn.Module):
    """Sinusoidal table, buffer ``pe`` (max_len, d_model) (encoders.py:123-142).  Dropout p is 0.0 as the
    sequence encoders construct it, so the table is added as is."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 2048, **kwargs):
        super().__init__()
        self.p = dropout
        pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        table = torch.zeros(max_len, d_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table)


class EmbeddedSequenceEncoder(NativeEncoder):
    """Pre-embedded sequence: LN(in) -> Linear(in, D) -> LN(D), pad rows zeroed, + positional table
    (encoders.py:169-214)."""
    kind = "embedded_sequence"

    def __init__(self, input_size=128, embedding_dim=512, padding_idx=0, dropout=0.0, max_tokens=1024, **kwargs):
        super().__init__()
        self.input_size = input_size
        self.embedding_dim = embedding_dim
        self.max_tokens = max_tokens
        self.token_encoder = nn.Sequential(nn.LayerNorm(input_size), nn.Linear(input_size, embedding_dim),
                                           nn.LayerNorm(embedding_dim))
        self.positional_encoder = PositionalEncoder(embedding_dim, dropout, max_tokens)


class _TokenTable(nn.Module):
    """``nn.Embedding(n, D, padding_idx, max_norm=1.0)`` as a parameter holder (encoders.py:17-37).  The
    max_norm renormalisation (rows with L2 norm > 1 rescaled in place on every forward) is done by the
    engine before the table is used."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=1.0):
        super().__init__()
        self.max_norm = max_norm
        self.embedding = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)


class _ValueMLP(nn.Module):
    """Linear(1,D) -> ReLU -> Linear(D,D) -> LN(D), zero where x == padding_value (encoders.py:40-72)."""

    def __init__(self, d_model, dropout=0.1, max_value=512, padding_value=0.0):
        super().__init__()
        self.linear1 = nn.Linear(1, d_model)
        self.linear2 = nn.Linear(d_model, d_model)
        self.norm = nn.LayerNorm(d_model)
        self.max_value = max_value
        self.padding_value = padding_value


class TabularEncoder(NativeEncoder):
    """Dense table: learned per-column embedding + value MLP (encoders.py:75-96)."""
    kind = "tabular"

    def __init__(self, num_embeddings=128, embedding_dim=512, padding_idx=-1, dropout=0.0, max_value=10000, **kwargs):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.register_buffer("index", torch.arange(num_embeddings))
        self.token_encoder = _TokenTable(num_embeddings, embedding_dim, padding_idx)
        self.value_encoder = _ValueMLP(embedding_dim, dropout, max_value, padding_idx)


encoders_dict: Dict[str, type] = {
    "EmbeddedSequenceEncoder": EmbeddedSequenceEncoder,
    "TabularEncoder": TabularEncoder,
}


# ------------------------------------------------------------------------------------------------------
# collators (host side; run in DataLoader workers)
# ------------------------------------------------------------------------------------------------------
def _batch_buffer(shape, fill, dtype):
    """The output buffer of a collator, filled with ``fill``.  Inside a DataLoader worker it is allocated in shared memory, as
    torch's default_collate does: a batch built in ordinary memory is copied into shared memory again when the worker hands it
    to the loop (58 MB per CMU batch of 32: 46 ms per batch with 8 workers against 12 ms, the difference between a loader-bound
    and a GPU-bound training loop, profiles/r04_slow_regime.md).  Same values either way."""
    if torch.utils.data.get_worker_info() is None:
        return torch.full(shape, fill, dtype=dtype)
    proto = torch.empty(0, dtype=dtype)
    numel = 1
    for d in shape:
        numel *= int(d)
    storage = proto._typed_storage()._new_shared(numel, device="cpu")
    return proto.new(storage).resize_(*shape).fill_(fill)


class SequenceCollator:
    """1-D sequences / dense tables padded (or cropped) to ``pad_len`` with ``pad_token``; ``attention_mask`` is int64 with
    1 where the padded value equals pad_token (encoders.py:286-311).  A missing sample (None) becomes an all-pad row.
    ``other_col`` is accepted and ignored, as in the reference (its ``__call__`` rebuilds the input dict with the data
    column only, so the second column never reaches the output).  One output buffer is filled in place."""

    def __init__(self, pad_token=0, pad_len=2048, data_col_name="indices", other_col="data", attn_mask=True, **kwargs):
        self.pad_token, self.pad_len, self.attn_mask = pad_token, pad_len, attn_mask
        self.data_col_name, self.other_col = data_col_name, other_col

    def _fill(self, rows, fill):
        # dtype of the stacked batch: promotion over the rows, a missing row counting as the float32 empty tensor the
        # reference substitutes (an int index column with a missing sample therefore comes out float32)
        dtype = None
        for x in rows:
            d = x.dtype if x is not None else torch.float32
            dtype = d if dtype is None else torch.promote_types(dtype, d)
        out = _batch_buffer((len(rows), self.pad_len), fill, dtype or torch.float32)
        for i, x in enumerate(rows):
            if x is not None and x.numel():
                n = min(x.shape[-1], self.pad_len)          # longer rows are cropped (the reference's negative F.pad does that)
                out[i, :n] = x[..., :n]
        return out

    def __call__(self, data):
        out = {self.data_col_name: self._fill(data[self.data_col_name], self.pad_token)}
        if self.attn_mask:
            out["attention_mask"] = (out[self.data_col_name] == self.pad_token).to(torch.long)
        return out


class EmbeddedSequenceCollator:
    """(len, emb) float sequences truncated / padded with ``fill_value`` to ``pad_len``; bool ``attention_mask`` True = pad
    (encoders.py:314-343).  None -> fully padded row.  NaN / inf are cleaned (``nan_to_num``) when ``clean``."""

    def __init__(self, pad_token=-1, fill_value=0.0, pad_len=2048, embedding_size=512, data_col_name="values",
                 attn_mask=True, truncate=True, clean=True, **kwargs):
        self.fill_value, self.pad_len, self.embedding_size = fill_value, pad_len, embedding_size
        self.data_col_name, self.attn_mask, self.truncate, self.clean = data_col_name, attn_mask, truncate, clean

    def __call__(self, data):
        seqs = data[self.data_col_name]
        first = next((x for x in seqs if x is not None), None)
        width = first.shape[-1] if first is not None else self.embedding_size
        dtype = first.dtype if first is not None else torch.float32
        tokens = _batch_buffer((len(seqs), self.pad_len, width), self.fill_value, dtype)
        mask = _batch_buffer((len(seqs), self.pad_len), True, torch.bool)
        for i, x in enumerate(seqs):
            if x is None:
                continue
            n = min(x.shape[0], self.pad_len)          # truncate=False crops as well: the reference's negative F.pad
            tokens[i, :n] = x[:n]
            mask[i, :n] = False
        if self.clean:
            tokens = tokens.nan_to_num_()
        out = {}
        if self.attn_mask:
            out["attention_mask"] = mask
        out["tokens"] = tokens
        return out


class MatrixCollator:
    """(rows, channels) float matrices padded along rows to ``pad_len`` with ``pad_token`` and cut to the first
    ``max_channels`` columns when that is non-zero; a missing sample is a (max_channels, pad_len) block of pad_token,
    as the reference builds it (encoders.py:346-364: note the transposed shape of the placeholder, kept)."""

    def __init__(self, pad_token=-10000, pad_len=2048, attn_mask=True, max_channels=0, **kwargs):
        self.pad_token, self.pad_len, self.max_channels = pad_token, pad_len, max_channels

    def __call__(self, data):
        mats = [torch.full((self.max_channels, self.pad_len), self.pad_token, dtype=torch.float) if x is None else x
                for x in data["values"]]
        outs = []
        for x in mats:
            n = min(x.shape[0], self.pad_len)
            o = torch.full((self.pad_len, x.shape[1]), self.pad_token, dtype=x.dtype)
            o[:n] = x[:n]
            outs.append(o[:, : self.max_channels] if self.max_channels else o)
        return {"values": torch.stack(outs)}


collators: Dict[str, type] = {
    "matrix": MatrixCollator,
    "sequence": SequenceCollator,
    "embedded_sequence": EmbeddedSequenceCollator,
}


class MultimodalCollator:
    """list of samples {modality: {column: tensor|None}} -> {modality: {name: batched tensor}}
    (encoders.py:374-403)."""

    def __init__(self, modality_config, labels=None, **kwargs):
        self.modality_collators = {name: collators[c["type"]](**c) for name, c in modality_config.items()}
        self.labels = labels

    def __call__(self, batch: List[dict]):
        assert self.modality_collators.keys() <= batch[0].keys(), f"{self.modality_collators.keys()} - {batch[0].keys()}"
        cols = defaultdict(lambda: defaultdict(list))
        for sample in batch:
            for name in self.modality_collators:
                for col, v in sample[name].items():
                    cols[name][col].append(v)
        out = {name: self.modality_collators[name](cols[name]) for name in self.modality_collators}
        if self.labels:
            lab = defaultdict(list)
            for sample in batch:
                for col, v in sample[self.labels].items():
                    lab[col].append(v)
            out[self.labels] = {k: torch.stack(v) for k, v in lab.items()}
        return out

"""Synthetic batches in the exact layouts the reference's collators emit.

  embedded_sequence: ``tokens (b, pad_len, in) f32`` zero-filled past the valid length,
                     ``attention_mask (b, pad_len) bool`` True = pad; dropped modality = all-pad row.
  sequence / tabular: ``values (b, n) f32`` with -10000 for a dropped modality,
                     ``attention_mask (b, n) int64`` = (values == -10000).
"""
from __future__ import annotations

from typing import Dict

import torch


def synthetic_batch(model_config: dict, batch_size: int, seed: int = 1234, p_drop: float = 0.0,
                    lengths: str = "uniform", device: str = "cpu") -> Dict[str, Dict[str, torch.Tensor]]:
    """lengths: "uniform" -> valid length ~ U{1..pad_len}; "full" -> every sequence fills pad_len.
    At least one modality is kept per sample."""
    g = torch.Generator().manual_seed(seed)
    enc = model_config["encoder_configs"]
    names = list(enc.keys())
    drop = torch.rand(batch_size, len(names), generator=g) < p_drop
    for i in range(batch_size):
        if drop[i].all():
            drop[i, int(torch.randint(0, len(names), (1,), generator=g))] = False
    batch = {}
    for mi, name in enumerate(names):
        c = enc[name]
        n = c["max_tokens"]
        if c["type"] == "EmbeddedSequenceEncoder":
            width = c["input_size"]
            if lengths == "full":
                ln = torch.full((batch_size,), n, dtype=torch.long)
            else:
                ln = torch.randint(1, n + 1, (batch_size,), generator=g)
            ln = torch.where(drop[:, mi], torch.zeros_like(ln), ln)
            mask = torch.arange(n)[None, :] >= ln[:, None]
            toks = torch.randn(batch_size, n, width, generator=g)
            toks = toks.masked_fill(mask[..., None], 0.0)
            batch[name] = {"tokens": toks.to(device), "attention_mask": mask.to(device)}
        elif c["type"] == "TabularEncoder":
            vals = torch.randn(batch_size, n, generator=g)
            vals = torch.where(drop[:, mi, None], torch.full_like(vals, -10000.0), vals)
            batch[name] = {"values": vals.to(device), "attention_mask": (vals == -10000).to(torch.long).to(device)}
        else:
            raise NotImplementedError(c["type"])
    return batch


class BatchPreDropout:
    """Per-sample modality pre-dropout of the reference's dataset pipeline (utils/dataset.py:29-57): with probability
    ``dropout`` a sample's modality is deleted (every column -> None, which the collators turn into a fully padded row) or
    filled with constants.  Uses torch's global RNG exactly like the reference (``torch.rand(1) < dropout``)."""

    def __init__(self, mode: str = "delete", kvs=None, dropout: float = 0.1, random_seed: int = 42):
        self.mode = mode
        self.kvs = kvs or {"attention_mask": 1, "tokens": 0}
        self.dropout = dropout

    def drop(self) -> bool:
        return bool(torch.rand(1) < self.dropout)

    def __call__(self, sample_modality: dict) -> dict:
        if not self.drop():
            return sample_modality
        if self.mode == "delete":
            return {k: None for k in sample_modality}
        if self.mode == "fill":
            return {k: (torch.full_like(v, self.kvs[k]) if v is not None else None) for k, v in sample_modality.items()}
        raise Exception(f"Did not recognize batch dropout mode {self.mode}")


def batch_predrop(modality_config: dict, random_seed: int = 42):
    """-> callable(sample) for ``datasets.Dataset.map(..., batched=False)`` (utils/dataset.py:59-69): modalities whose
    config has a non-zero ``dropout`` are pre-dropped per sample."""
    droppers = {name: BatchPreDropout(kvs={"attention_mask": c.get("pad_token", 1), "data": 0.0}, dropout=c["dropout"],
                                      random_seed=random_seed)
                for name, c in modality_config.items() if c.get("dropout")}

    def apply(sample: dict) -> dict:
        return {k: (droppers[k](v) if k in droppers else v) for k, v in sample.items()}

    return apply


def setup_data(dataset_path, split=0.1, ds_frac=1.0, ds_seed=42, model=3, predrop=False, predrop_config=None):
    """The reference's dataset pipeline (utils/dataset.py:72-84), same signature and order: load_from_disk -> optional
    prefix selection (ds_frac) -> optional per-sample modality pre-dropout (``dataset.map``, BEFORE the split, so train and
    test are both pre-dropped) -> train_test_split.  ``predrop=True`` without a usable ``predrop_config`` raises instead of
    silently training on fully-present data."""
    from datasets import load_from_disk
    dataset = load_from_disk(dataset_path).with_format("torch")
    if ds_frac < 1.0:
        dataset = dataset.select(list(range(0, int(len(dataset) * ds_frac))))
    if predrop:
        if not predrop_config:
            raise ValueError("predrop is set but the config has no modality_config to take the dropout rates from")
        missing = [k for k, c in predrop_config.items() if "dropout" not in c]
        if missing:
            raise KeyError(f"predrop: modality_config entries without a 'dropout' key: {missing}")           # config['dropout'] in the reference
        print(f"Running preprocessing dropout of modalities using random seed {torch.random.initial_seed()}")
        dataset = dataset.map(batch_predrop(predrop_config, ds_seed), batched=False)
    if split and split != 1.0:
        dataset = dataset.train_test_split(split, seed=ds_seed)
    return dataset


def shard_eval_batches(n: int, batch_size: int, world: int, rank: int):
    """Index batches of rank ``rank`` for an UNSHUFFLED loader over ``n`` samples prepared by Accelerate for ``world`` processes
    (the reference's ``eval_dl`` through ``accelerator.prepare``, train_accel_gpu.py:71,93): Accelerate's ``BatchSamplerShard``
    with its defaults (``split_batches=False, even_batches=True``, the loader's ``drop_last=False``) deals the sequential
    batches round-robin - rank r takes batches r, r + W, ... - and completes the last round with samples from the START of
    the dataset, so that every rank sees the same number of FULL batches (the contrastive labels need equal local batches,
    utils/contrastive_loss_with_temperature.py:28-31).  Restated in closed form: the sequence 0 .. n-1 followed by the first
    min(n, W * batch_size) indices cycled, cut into ceil(ceil(n / b) / W) * W batches of b.  Checked against
    ``accelerate.data_loader.BatchSamplerShard`` itself in tests/test_aux_cpu.py."""
    nb = -(-n // batch_size)
    rounds = -(-nb // world)
    total = rounds * world * batch_size
    base = min(n, world * batch_size)
    virt = list(range(n)) + [j % base for j in range(total - n)]
    return [virt[(k * world + rank) * batch_size:(k * world + rank + 1) * batch_size] for k in range(rounds)]


class DevicePrefetcher:
    """Host batches -> device batches, one batch AHEAD of the compute stream (the input pipeline of train_accel_gpu.py:70,111:
    the reference moves every batch with a synchronous ``move_to`` in the compute stream; at b = 32 that is 59 MB = 1.6 ms of
    PCIe time per 22 ms step).

    Two sets of device buffers; the copies of batch i + 1 run on a dedicated copy stream while the kernels of batch i run on the
    compute stream.  Ordering is by events only (no host sync): the compute stream waits for the copy event of the batch it is
    handed, and the copy stream waits, before refilling a buffer set, for the event the compute stream recorded when the consumer
    came back for its next batch (everything that read the set has been enqueued by then).  Host tensors are pinned on first
    sight unless they already are (``DataLoader(pin_memory=True)``), so the copies are true asynchronous DMA.  Batches must keep
    their shapes (the collators pad every modality to its ``pad_len``); a batch of another shape (the last partial one) is passed
    through with a plain copy on the compute stream."""

    def __init__(self, batches, device, depth: int = 2):
        self.it = iter(batches)
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.sets = [None] * depth                      # device buffer sets, allocated from the first batches
        self.copied = [torch.cuda.Event() for _ in range(depth)]
        self.released = [None] * depth                  # recorded on the compute stream when a set's consumer is done enqueueing
        self.slot = 0
        self.in_use = None                              # (slot handed out last, the stream it was handed out on)
        self.pending = None                             # (slot or None, batch) staged ahead
        self._stage()

    @staticmethod
    def _shape_sig(b):
        if torch.is_tensor(b):
            return (tuple(b.shape), b.dtype)
        if isinstance(b, dict):
            return tuple((k, DevicePrefetcher._shape_sig(v)) for k, v in b.items())
        if isinstance(b, list):
            return tuple(DevicePrefetcher._shape_sig(v) for v in b)
        raise TypeError("Invalid type for move_to")

    @staticmethod
    def _has_device_tensor(b):
        if torch.is_tensor(b):
            return b.device.type != "cpu"
        return any(DevicePrefetcher._has_device_tensor(v) for v in (b.values() if isinstance(b, dict) else b))

    def _alloc_like(self, b):
        if torch.is_tensor(b):
            return torch.empty(b.shape, dtype=b.dtype, device=self.device)
        if isinstance(b, dict):
            return {k: self._alloc_like(v) for k, v in b.items()}
        return [self._alloc_like(v) for v in b]

    def _copy_into(self, dst, src):
        if torch.is_tensor(src):
            if src.device.type == "cpu" and not src.is_pinned():
                src = src.pin_memory()
            dst.copy_(src, non_blocking=True)
            return
        for k in (src.keys() if isinstance(src, dict) else range(len(src))):
            self._copy_into(dst[k], src[k])

    def _stage(self):
        try:
            host = next(self.it)
        except StopIteration:
            self.pending = None
            return
        s = self.slot
        sig = self._shape_sig(host)
        if self.sets[s] is None:
            self.sets[s] = (sig, self._alloc_like(host))
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))          # allocated here, first written there
        if self.sets[s][0] != sig:                      # odd-shaped batch: no double buffering for it
            self.pending = (None, host)
            return
        if self._has_device_tensor(host):               # a device-resident source (synthetic_batch(device='cuda')): its producer
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))          # runs on the current stream
        with torch.cuda.stream(self.copy_stream):
            if self.released[s] is not None:
                self.copy_stream.wait_event(self.released[s])
            self._copy_into(self.sets[s][1], host)
            self.copied[s].record(self.copy_stream)
        self._keep = host                               # pinned source stays alive until the next staging call
        self.pending = (s, self.sets[s][1])
        self.slot = (s + 1) % len(self.sets)

    def __iter__(self):
        return self

    def __next__(self):
        cur = torch.cuda.current_stream(self.device)
        if self.in_use is not None:                     # the consumer is back: everything reading that set has been enqueued ...
            slot, handed_on = self.in_use               # ... on the stream the set was handed out on (which need not be `cur`)
            ev = torch.cuda.Event()
            ev.record(handed_on)
            self.released[slot] = ev
            self.in_use = None
        if self.pending is None:
            raise StopIteration
        s, batch = self.pending
        if s is None:
            out = _to_device(batch, self.device)
        else:
            cur.wait_event(self.copied[s])
            out, self.in_use = batch, (s, cur)
        self._stage()
        return out


def _to_device(obj, device):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_to_device(v, device) for v in obj]
    raise TypeError("Invalid type for move_to")

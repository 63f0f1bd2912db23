"""Embedding extraction with the reference's shape (infer_accel_gpu.py:97-136): load a checkpoint (``restart`` key of the
YAML), run the model in eval mode over the train and test splits, and write
``<output_dir>/{train,eval}_{embeddings,masks,labels}.pt`` in the reference's format (dict name -> (n, D) tensor;
dict modality -> (n,) bool; (n, ...) labels) — the files ``lp_accel_gpu.py`` of the reference consumes.

    python infer_accel_gpu.py <config.yaml> [--synthetic BATCHES]
"""
import importlib
import os
import sys
from collections import defaultdict

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
P = importlib.import_module("mca-paper_amd")


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    synthetic = int(sys.argv[sys.argv.index("--synthetic") + 1]) if "--synthetic" in sys.argv else 0
    assert torch.cuda.device_count() >= 1
    device = torch.device("cuda", 0)
    config = P.config.training_config(sys.argv[1])
    torch.manual_seed(0)                                                        # infer_accel_gpu.py:28 (seeds the predrop draws)
    model_config = P.config.get_model_config(config)
    model = P.build_model(model_config).to(device)          # infer_accel_gpu.py:50-53
    if config.restart:
        missing, unexpected = P.checkpoint.load_model(model, config.restart, strict=False)
        if missing or unexpected:          # embeddings from partly random weights are worse than no embeddings
            raise KeyError(f"checkpoint {config.restart} does not match the model: missing {missing[:8]} unexpected {unexpected[:8]}")
    elif not synthetic:
        raise AssertionError("config.restart must name a checkpoint")           # infer_accel_gpu.py:90
    model.eval()
    label_col = config.label_col
    if synthetic:
        def split(seed0):
            for i in range(synthetic):
                b = P.data.synthetic_batch(model_config, config.batch_size, seed=seed0 + i, p_drop=0.2)
                b[label_col] = {"data": torch.randn(config.batch_size, 7)}
                yield b
        splits = {"train": split(100), "eval": split(10_000)}
    else:
        from torch.utils.data import DataLoader
        # infer_accel_gpu.py:36-41 of the reference: the same setup_data call as training (ds_frac, predrop, split)
        ds = P.data.setup_data(config.dataset, split=config.split, ds_frac=config.ds_frac, ds_seed=config.ds_seed,
                               predrop=bool(config.get("predrop", False)), predrop_config=config.get("modality_config", {}))
        coll = P.MultimodalCollator(config.get("modality_config", {}), labels=label_col)
        splits = {"train": DataLoader(ds["train"], collate_fn=coll, batch_size=config.batch_size, drop_last=True, shuffle=False),
                  "eval": DataLoader(ds["test"], collate_fn=coll, batch_size=config.batch_size, drop_last=True, shuffle=False)}
    with torch.no_grad():
        for tv, dl in splits.items():
            embeddings, masks, labels = defaultdict(list), defaultdict(list), []
            for batch in dl:
                batch_labels = batch.pop(label_col)
                batch = {k: {kk: vv.to(device) for kk, vv in v.items()} for k, v in batch.items()}
                outputs = model(batch)
                outputs.pop("loss"); outputs.pop("losses")
                for extra in ("fcl_loss", "no-fcl_loss"):
                    outputs.pop(extra, None)
                for k, v in outputs.pop("modality_sample_mask").items():
                    masks[k].append(v.detach().cpu())
                for k, v in outputs.items():
                    embeddings[k].append(v.detach().cpu())
                labels.append(batch_labels["data"].detach().cpu())
            torch.save({k: torch.cat(v, 0) for k, v in masks.items()}, f"{config.output_dir}/{tv}_masks.pt")
            torch.save({k: torch.cat(v, 0) for k, v in embeddings.items()}, f"{config.output_dir}/{tv}_embeddings.pt")
            torch.save(torch.cat(labels, 0), f"{config.output_dir}/{tv}_labels.pt")
            print(f"{tv}: {sum(x.shape[0] for x in labels)} samples -> {config.output_dir}/{tv}_*.pt", flush=True)


if __name__ == "__main__":
    main()

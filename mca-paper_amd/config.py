"""Training configuration: the reference's YAML keys and defaults without yacs.

Reference: utils/config.py:9-61 (defaults), :76-93 (YAML overlay; unknown keys are accepted and kept),
:96-117 (``get_model_config``: the kwargs dict ``MCA(**model_config)`` is built from).
"""
from __future__ import annotations

import copy
import os
from datetime import datetime
from typing import Any, Dict

import yaml


class Config(dict):
    """dict with attribute access (the reference uses a yacs CfgNode the same way)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def default_train_config() -> Config:
    """Same keys and values as utils/config.py:9-61."""
    return Config(
        encoder_configs={}, modality_configs={},
        restart="", wandb_name="No Name", wandb_account_name="", wandb_restart="",
        epochs=3, start_epoch=0, batch_size=32, n_step_checkpoint=0, num_warmup_steps=3000,
        lr_scheduler_type="cosine", lr=1e-4, output_dir="", label_col="Labels", dataset="", split=0.1,
        ds_frac=1.0, ds_seed=42, clip=0.0,
        hidden_size=512, layers=10, heads=8, dim_head=64, ff_mult=4, num_fusion_tokens=256, seed=42,
        mean_pool=False, dropout=0.1, zorro=False, eao=False, run_eval_loop=True,
        bimodal_contrastive=True, non_fusion_fcl=True, fcl=True, no_fusion=False,
        fcl_root=[1, 2, 3, 4], fusion_combos=[4, 3, 2], return_logits=True,
    )


def training_config(filename: str, make_output_dir: bool = True) -> Config:
    """Defaults overlaid with the YAML (every YAML key is accepted, as with CfgNode(new_allowed=True));
    a timestamped output directory is created and the merged config dumped into it
    (utils/config.py:81-91)."""
    cfg = default_train_config()
    with open(filename, "r") as f:
        overlay = yaml.safe_load(f) or {}
    cfg.update(overlay)
    if not cfg.get("output_dir"):
        base = datetime.now().strftime("training_output_%H_%M_%d_%m_%Y")
        out, i = base, 1
        while os.path.isdir(out):
            out = f"{base}_{i}"
            i += 1
        cfg["output_dir"] = out
    if make_output_dir:
        os.makedirs(cfg["output_dir"], exist_ok=True)
        with open(os.path.join(cfg["output_dir"], "config.yaml"), "w") as f:
            yaml.safe_dump(dict(cfg), f)
    return cfg


def get_model_config(cfg: Dict[str, Any]) -> Dict[str, Any]:
    """utils/config.py:96-117."""
    return {
        "dim": cfg["hidden_size"], "depth": cfg["layers"], "heads": cfg["heads"], "dim_head": cfg["dim_head"],
        "ff_mult": cfg["ff_mult"], "num_fusion_tokens": cfg["num_fusion_tokens"],
        "encoder_configs": copy.deepcopy(cfg["encoder_configs"]), "batch_size": cfg["batch_size"],
        "fcl": cfg["fcl"], "fcl_root": cfg["fcl_root"], "bimodal_contrastive": cfg["bimodal_contrastive"],
        "non_fusion_fcl": cfg["non_fusion_fcl"], "fusion_combos": cfg["fusion_combos"], "zorro": cfg["zorro"],
        "eao": cfg["eao"], "no_fusion": cfg["no_fusion"], "mean_pool": cfg["mean_pool"],
    }


# ---- the BASELINE.json workloads, spelled out (configs/CMU_config1.yaml, CMU_config1_z_d40.yaml,
# ---- TCGA_config1.yaml of the reference) --------------------------------------------------------------
CMU_ENCODERS = {
    "COVAREP": {"type": "EmbeddedSequenceEncoder", "input_size": 74, "max_tokens": 1500},
    "FACET": {"type": "EmbeddedSequenceEncoder", "input_size": 35, "max_tokens": 450},
    "OpenFace": {"type": "EmbeddedSequenceEncoder", "input_size": 713, "max_tokens": 450},
    "glove_vectors": {"type": "EmbeddedSequenceEncoder", "input_size": 300, "max_tokens": 50},
}
TCGA_ENCODERS = {
    "gene": {"type": "TabularEncoder", "num_embeddings": 800, "max_tokens": 800, "max_value": 100},
    "protein": {"type": "TabularEncoder", "num_embeddings": 198, "max_tokens": 198, "max_value": 100},
    "methylation": {"type": "TabularEncoder", "num_embeddings": 800, "max_tokens": 800, "max_value": 100},
    "mirna": {"type": "TabularEncoder", "num_embeddings": 662, "max_tokens": 662, "max_value": 100},
}


def cmu_model_config(batch_size: int = 8, zorro: bool = False, long_seq: bool = False) -> Dict[str, Any]:
    enc = copy.deepcopy(CMU_ENCODERS)
    if long_seq:                       # BASELINE config 5: every modality padded to 1500 tokens
        for e in enc.values():
            e["max_tokens"] = 1500
    return dict(dim=512, depth=5, heads=8, dim_head=64, ff_mult=4, num_fusion_tokens=88, encoder_configs=enc,
                batch_size=batch_size, fcl=not zorro, fcl_root=[0, 1, 2, 3], bimodal_contrastive=False,
                non_fusion_fcl=False, fusion_combos=[4, 3, 2], zorro=zorro, eao=False, no_fusion=False,
                mean_pool=False)


def cmu_eao_model_config(batch_size: int = 8) -> Dict[str, Any]:
    """configs/CMU_config1_EAO.yaml of the reference: the EAO baseline on the CMU modalities (pairs of modalities as
    combinations, pairwise + fusion-channel losses, mean pooling)."""
    return dict(dim=512, depth=5, heads=8, dim_head=64, ff_mult=4, num_fusion_tokens=88, encoder_configs=copy.deepcopy(CMU_ENCODERS),
                batch_size=batch_size, fcl=True, fcl_root=[0, 1], bimodal_contrastive=True, non_fusion_fcl=True,
                fusion_combos=[2], zorro=False, eao=True, no_fusion=True, mean_pool=True)


def tcga_model_config(batch_size: int = 8) -> Dict[str, Any]:
    return dict(dim=512, depth=5, heads=8, dim_head=64, ff_mult=4, num_fusion_tokens=88,
                encoder_configs=copy.deepcopy(TCGA_ENCODERS), batch_size=batch_size, fcl=True, fcl_root=[0, 1, 2, 3],
                bimodal_contrastive=True, non_fusion_fcl=True, fusion_combos=[4, 3, 2], zorro=False, eao=False,
                no_fusion=False, mean_pool=False)

"""MI355X-native implementation of the MCA / MMA multimodal-fusion training step
(drop-in for the hot path of josiahbjorgaard/mca-paper: model.py / encoders.py / the step loop of
train_accel_gpu.py).  Import by name: ``importlib.import_module("mca-paper_amd")`` (the directory name
has a hyphen), or use the root-level shims ``model.py`` / ``encoders.py``.
"""
from . import checkpoint, config, data, encoders, metrics, params, structure  # noqa: F401
from .encoders import MultimodalCollator, collators, encoders_dict  # noqa: F401
from .model import MCA, EAO  # noqa: F401



def build_model(model_config: dict):
    """``EAO(**model_config) if model_config['eao'] else MCA(**model_config)`` (train_accel_gpu.py:51-54)."""
    return EAO(**model_config) if model_config.get("eao") else MCA(**model_config)


__all__ = ["build_model", "MCA", "EAO", "encoders_dict", "collators", "MultimodalCollator", "config", "data", "params", "structure"]

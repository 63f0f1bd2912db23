"""``utils.config`` of the reference (utils/config.py:76-117): ``training_config(yaml)``, ``get_model_config(config)``."""
import importlib as _il

_c = _il.import_module("mca-paper_amd.config")
training_config, get_model_config, default_train_config = _c.training_config, _c.get_model_config, _c.default_train_config
get_cfg_defaults_train = _c.default_train_config          # utils/config.py:9

__all__ = ["training_config", "get_model_config", "get_cfg_defaults_train", "default_train_config"]

"""Drop-in module with the reference's name: ``from model import MCA, EAO`` (reference: train_accel_gpu.py:12,
infer_accel_gpu.py:12; classes at model.py:24-126,282,481).  The implementation lives in mca-paper_amd/model.py (native HIP
path); every module class of the reference's model.py that has a native counterpart is re-exported under its name."""
import importlib as _il

_m = _il.import_module("mca-paper_amd.model")
MCA, EAO = _m.MCA, _m.EAO
LayerNorm, FeedForward, Attention, MCALayer, MCAPretrainingLoss = _m.LayerNorm, _m.FeedForward, _m.Attention, _m.MCALayer, _m.MCAPretrainingLoss
encoders_dict = _il.import_module("mca-paper_amd.encoders").encoders_dict          # model.py:8 imports it into this namespace

__all__ = ["MCA", "EAO", "LayerNorm", "FeedForward", "Attention", "MCALayer", "MCAPretrainingLoss", "encoders_dict"]

"""Drop-in module with the reference's name: ``from model import MCA`` (reference: model.py:282).
The implementation lives in mca-paper_amd/model.py (native HIP path)."""
import importlib as _il

_m = _il.import_module("mca-paper_amd.model")
MCA = _m.MCA
LayerNorm, FeedForward, Attention, MCALayer, MCAPretrainingLoss = _m.LayerNorm, _m.FeedForward, _m.Attention, _m.MCALayer, _m.MCAPretrainingLoss

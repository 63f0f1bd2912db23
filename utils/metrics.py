"""``utils.metrics`` of the reference (utils/metrics.py:20-70): Wang-Isola alignment / uniformity accumulators."""
import importlib as _il

_m = _il.import_module("mca-paper_amd.metrics")
Alignment, Uniformity, lalign, lunif = _m.Alignment, _m.Uniformity, _m.lalign, _m.lunif

__all__ = ["Alignment", "Uniformity", "lalign", "lunif"]

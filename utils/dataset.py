"""``utils.dataset`` of the reference (utils/dataset.py:29-84): ``setup_data`` (same signature), ``BatchPreDropout``."""
import importlib as _il

_d = _il.import_module("mca-paper_amd.data")
setup_data, BatchPreDropout, batch_predrop = _d.setup_data, _d.BatchPreDropout, _d.batch_predrop

__all__ = ["setup_data", "BatchPreDropout", "batch_predrop"]

"""Drop-in module with the reference's name: ``from encoders import encoders_dict, collators, MultimodalCollator``
(reference: encoders.py:277-283,367-403).  Implementation: mca-paper_amd/encoders.py."""
import importlib as _il

_e = _il.import_module("mca-paper_amd.encoders")
encoders_dict, collators, MultimodalCollator = _e.encoders_dict, _e.collators, _e.MultimodalCollator
EmbeddedSequenceEncoder, TabularEncoder, PositionalEncoder = _e.EmbeddedSequenceEncoder, _e.TabularEncoder, _e.PositionalEncoder
SequenceCollator, EmbeddedSequenceCollator, MatrixCollator = _e.SequenceCollator, _e.EmbeddedSequenceCollator, _e.MatrixCollator

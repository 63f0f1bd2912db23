"""Deterministic parameter initialisation shared by tests, bench and the golden generator."""
from __future__ import annotations

from typing import Dict

import torch


def init_state_dict(model_config: dict, seed: int = 43) -> Dict[str, torch.Tensor]:
    """state_dict (reference key names) of a freshly constructed ``MCA(**model_config)`` under
    ``torch.manual_seed(seed)`` on the CPU generator.  Because the module tree is created in the reference's
    order with the same torch constructors, this equals the reference's own initial weights for that seed
    (checked by tests/test_host_cpu.py::test_state_dict_keys_and_same_seed_init_as_reference against tests/golden/cmu_init_checksums.pt)."""
    from .model import MCA, EAO
    gen_state = torch.random.get_rng_state()
    try:
        torch.manual_seed(seed)
        m = (EAO if model_config.get("eao") else MCA)(**model_config)
        return {k: v.detach().clone() for k, v in m.state_dict().items()}
    finally:
        torch.random.set_rng_state(gen_state)

"""Stand-ins for the HF model objects the reference's callers hold (demo.py:125-132) when no
checkpoint is available: they only carry a state dict and a config, which is all the engine reads."""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import synth


class SyntheticLM:
    """Quacks like the `lm_model` of demo.py:125 for this repo's gen_utils: `.state_dict()`,
    `.eval()`, `.to(device)`; weights from conzic_amd.synth."""

    def __init__(self, cfg: synth.BertCfg, seed: int = 11, state: Dict[str, np.ndarray] = None):
        self.czc_cfg = cfg
        self._state = state if state is not None else synth.make_bert_weights(cfg, seed)

    def state_dict(self):
        return self._state

    def eval(self):
        return self

    def to(self, device):
        return self

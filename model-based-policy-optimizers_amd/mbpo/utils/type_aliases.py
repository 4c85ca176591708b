"""OptimizerState / OptimizerTrainingOutPut — mirrors mbpo/utils/type_aliases.py:10-19 (chex dataclasses there;
plain dataclasses with .replace here)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Any


class _Replace:
    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


@dataclass
class OptimizerState(_Replace):
    true_buffer_state: Any
    system_params: Any
    key: int


@dataclass
class OptimizerTrainingOutPut(_Replace):
    optimizer_state: OptimizerState

"""State — mirrors mbpo/optimizers/policy_optimizers/brax_utils/base.py:12-23 (brax env State carrying system_params).
All leaves are batched over envs ([N, ...]) — the reference gets the batch axis from VmapWrapper (training.py:50-74)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch


@dataclass
class State:
    pipeline_state: Optional[Any]
    obs: torch.Tensor          # [N, x_dim]
    reward: torch.Tensor       # [N]
    done: torch.Tensor         # [N] float flags
    system_params: Any
    metrics: Dict[str, Any] = field(default_factory=dict)
    info: Dict[str, Any] = field(default_factory=dict)   # 'steps' [N], 'truncation' [N], 'first_obs' [N, x_dim]

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

"""Dynamics ABC — mirrors mbpo/systems/dynamics/base_dynamics.py:11-24."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Generic, Tuple, TypeVar

import torch

DynamicsParams = TypeVar("DynamicsParams")


class Normal:
    """The only part of distrax.Distribution the reference consumes is .mean() (pendulum_system.py:32,34)."""

    def __init__(self, loc: torch.Tensor, scale: torch.Tensor):
        self.loc, self.scale = loc, scale

    def mean(self) -> torch.Tensor:
        return self.loc

    def stddev(self) -> torch.Tensor:
        return self.scale


class Dynamics(ABC, Generic[DynamicsParams]):
    def __init__(self, x_dim: int, u_dim: int):
        self.x_dim = x_dim
        self.u_dim = u_dim

    @abstractmethod
    def next_state(self, x: torch.Tensor, u: torch.Tensor, dynamics_params: DynamicsParams) -> Tuple[Normal, DynamicsParams]:
        pass

    @abstractmethod
    def init_params(self, key: int) -> DynamicsParams:
        pass

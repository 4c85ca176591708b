"""Reward ABC — mirrors mbpo/systems/rewards/base_rewards.py:11-25."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Generic, Optional, Tuple, TypeVar

import torch

from mbpo.systems.dynamics.base_dynamics import Normal

RewardParams = TypeVar("RewardParams")


class Reward(ABC, Generic[RewardParams]):
    def __init__(self, x_dim: int, u_dim: int):
        self.x_dim = x_dim
        self.u_dim = u_dim

    @abstractmethod
    def __call__(self, x: torch.Tensor, u: torch.Tensor, reward_params: RewardParams,
                 x_next: Optional[torch.Tensor] = None) -> Tuple[Normal, RewardParams]:
        pass

    @abstractmethod
    def init_params(self, key: int) -> RewardParams:
        pass

    # MI355X seam: how the fused rollout kernel evaluates this reward (kind id + device parameter vector)
    def kernel_spec(self, reward_params: RewardParams, device):
        raise NotImplementedError(f"{type(self).__name__} has no HIP kernel form; the fused rollout has no CPU/Python fallback")

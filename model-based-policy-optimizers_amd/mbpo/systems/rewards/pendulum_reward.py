"""PendulumReward — mirrors mbpo/systems/rewards/pendulum_reward.py:12-42; plus a build-defined QuadraticReward."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Optional, Sequence

import torch

from mbpo import _hip
from mbpo.systems.dynamics.base_dynamics import Normal
from mbpo.systems.rewards.base_rewards import Reward


@dataclass
class PendulumRewardParams:
    control_cost: float = 0.02
    angle_cost: float = 1.0
    target_angle: float = 0.0

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class PendulumReward(Reward[PendulumRewardParams]):
    def __init__(self):
        super().__init__(x_dim=3, u_dim=1)

    def init_params(self, key: int) -> PendulumRewardParams:
        return PendulumRewardParams()

    def kernel_spec(self, reward_params, device):
        p = reward_params
        return _hip.REWARD_PENDULUM, torch.tensor([p.angle_cost, p.control_cost, p.target_angle], dtype=torch.float32, device=device)

    def __call__(self, x, u, reward_params, x_next=None):
        from mbpo.systems.pendulum_system import _pendulum_step
        from mbpo.systems.dynamics.pendulum_dynamics import PendulumDynamicsParams
        _, r = _pendulum_step(x, u, PendulumDynamicsParams(), reward_params)
        return Normal(r, torch.zeros_like(r)), reward_params


@dataclass
class QuadraticRewardParams:
    target: Sequence[float]
    q: Sequence[float]
    r: Sequence[float]

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class QuadraticReward(Reward[QuadraticRewardParams]):
    """reward = -sum_d q_d (x_d - target_d)^2 - sum_d r_d u_d^2  (for non-Pendulum shapes; not in the reference)."""

    def __init__(self, x_dim: int, u_dim: int, target=None, q=None, r=None):
        super().__init__(x_dim, u_dim)
        self._default = QuadraticRewardParams(list(target if target is not None else [0.0] * x_dim),
                                              list(q if q is not None else [1.0] * x_dim),
                                              list(r if r is not None else [0.1] * u_dim))

    def init_params(self, key: int) -> QuadraticRewardParams:
        return self._default

    def kernel_spec(self, reward_params, device):
        p = reward_params
        return _hip.REWARD_QUADRATIC, torch.tensor(list(p.target) + list(p.q) + list(p.r), dtype=torch.float32, device=device)

    def __call__(self, x, u, reward_params, x_next=None):
        raise NotImplementedError("QuadraticReward is evaluated inside the fused System.step / rollout kernels")

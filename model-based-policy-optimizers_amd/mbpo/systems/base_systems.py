"""System / SystemParams / SystemState — mirrors mbpo/systems/base_systems.py:13-60.

`System.step(x, u, system_params)` keeps the reference signature; x may be a single state [x_dim] (what the reference
writes and vmaps) or a batch [N, x_dim] (batch-native here: one fused launch instead of vmap).
"""
from __future__ import annotations

import dataclasses
from abc import ABC
from dataclasses import dataclass, field
from typing import Any, Generic

import torch

from mbpo.systems.dynamics.base_dynamics import Dynamics, DynamicsParams
from mbpo.systems.rewards.base_rewards import Reward, RewardParams
from mbpo.utils import keys as K


@dataclass
class SystemParams(Generic[DynamicsParams, RewardParams]):
    dynamics_params: Any
    reward_params: Any
    key: int = 0

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


@dataclass
class SystemState(Generic[DynamicsParams, RewardParams]):
    x_next: torch.Tensor
    reward: torch.Tensor
    system_params: SystemParams
    done: Any = 0.0

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class System(ABC, Generic[DynamicsParams, RewardParams]):
    def __init__(self, dynamics: Dynamics, reward: Reward):
        self.dynamics = dynamics
        self.reward = reward
        self.x_dim = dynamics.x_dim
        self.u_dim = dynamics.u_dim

    @staticmethod
    def system_params_vmap_axes(axes: int = 0):
        return SystemParams(dynamics_params=None, reward_params=None, key=axes)

    def step(self, x: torch.Tensor, u: torch.Tensor, system_params: SystemParams) -> SystemState:
        """Systems that exist as device code (PendulumSystem, EnsembleSystem): one fused launch of the rollout kernel with
        open-loop actions and S=1 (csrc/rollout.hip).  A user-defined System overrides this with its own BATCHED torch code
        (x [N, x_dim], u [N, u_dim] on the device -> SystemState with x_next [N, x_dim], reward [N]); the trainers then step it
        between the HIP policy and bookkeeping kernels (ops.generic_rollout)."""
        from mbpo import ops
        if not self.fused:
            raise NotImplementedError(f"{type(self).__name__} must define step(x, u, system_params) itself (batched torch code)")
        dev = _device_of(x)
        single = x.dim() == 1
        xb = x.reshape(-1, self.x_dim).to(dev, torch.float32).contiguous().clone()
        ub = u.reshape(-1, self.u_dim).to(dev, torch.float32).contiguous()
        n = xb.shape[0]
        key, sub = K.split(system_params.key)
        rows = ops.model_rollout(x_dim=self.x_dim, u_dim=self.u_dim, actions=ub.reshape(1, n, self.u_dim), obs=xb,
                                 first_obs=xb.clone(), steps=torch.zeros(n, device=dev), done=torch.zeros(n, device=dev),
                                 n_steps=1, episode_length=2 ** 30, seed=sub, **self.rollout_spec(system_params, dev))
        X, U = self.x_dim, self.u_dim
        x_next, reward = rows[:, X + U + 2:2 * X + U + 2], rows[:, X + U]
        if single:
            x_next, reward = x_next[0], reward[0]
        # a stochastic System must split-and-return its key; the reference's Pendulum drops it (:38)
        return SystemState(x_next=x_next, reward=reward, system_params=system_params.replace(key=key))

    def init_params(self, key: int) -> SystemParams:
        keys = K.split(key, 3)
        return SystemParams(dynamics_params=self.dynamics.init_params(keys[0]),
                            reward_params=self.reward.init_params(keys[1]), key=keys[2])

    # MI355X seam: keyword arguments describing this system to ops.model_rollout.  PendulumSystem / EnsembleSystem return the
    # description of their device code (one fused launch per unroll); the default is the non-fused path for a user-defined
    # System: its own `step` between the HIP policy and episode-bookkeeping kernels.
    def rollout_spec(self, system_params: SystemParams, device) -> dict:
        from mbpo import _hip
        return dict(system_kind=_hip.SYS_GENERIC, system=self, system_params=system_params)

    @property
    def fused(self) -> bool:
        """True when the system runs inside the fused rollout kernel (and a training step can be hipGraph-captured)."""
        return type(self).rollout_spec is not System.rollout_spec


def _device_of(t: torch.Tensor) -> torch.device:
    return t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())

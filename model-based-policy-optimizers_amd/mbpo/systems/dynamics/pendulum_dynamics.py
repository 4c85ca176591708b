"""PendulumDynamics — mirrors mbpo/systems/dynamics/pendulum_dynamics.py:12-63; arithmetic runs in csrc/rollout.hip."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass

import torch

from mbpo.systems.dynamics.base_dynamics import Dynamics, Normal


@dataclass
class PendulumDynamicsParams:
    max_speed: float = 8.0
    max_torque: float = 2.0
    dt: float = 0.05
    g: float = 9.81
    m: float = 1.0
    l: float = 1.0

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    def vector(self, device) -> torch.Tensor:
        return torch.tensor([self.max_speed, self.max_torque, self.dt, self.g, self.m, self.l], dtype=torch.float32, device=device)


class PendulumDynamics(Dynamics[PendulumDynamicsParams]):
    def __init__(self):
        super().__init__(x_dim=3, u_dim=1)

    def init_params(self, key: int) -> PendulumDynamicsParams:
        return PendulumDynamicsParams()

    def next_state(self, x, u, dynamics_params):
        from mbpo.systems.pendulum_system import _pendulum_step   # one fused launch computes x' and reward together
        xn, _ = _pendulum_step(x, u, dynamics_params, None)
        return Normal(xn, torch.zeros_like(xn)), dynamics_params

"""Systems: analytic Pendulum (PINNED by closed-form known answers) and the learned ensemble (build-defined).

Batched torch-CPU restatement (the reference writes single-sample functions and vmaps them:
brax_utils/training.py:71-74).  Test infrastructure only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import nets


# ------------------------------------------------------------------------------------------ Pendulum
@dataclass
class PendulumParams:
    """PendulumDynamicsParams (dynamics/pendulum_dynamics.py:12-19) + PendulumRewardParams (rewards/pendulum_reward.py:12-16)."""
    max_speed: float = 8.0
    max_torque: float = 2.0
    dt: float = 0.05
    g: float = 9.81
    m: float = 1.0
    l: float = 1.0
    control_cost: float = 0.02
    angle_cost: float = 1.0
    target_angle: float = 0.0

    def sys_vector(self):
        return [self.max_speed, self.max_torque, self.dt, self.g, self.m, self.l]

    def reward_vector(self):
        return [self.angle_cost, self.control_cost, self.target_angle]


def pendulum_next_state(x: torch.Tensor, u: torch.Tensor, p: PendulumParams) -> torch.Tensor:
    """PendulumDynamics.next_state + ode (dynamics/pendulum_dynamics.py:29-63).  x: [N,3], u: [N,1] -> [N,3]."""
    dt_ = x.dtype
    c = lambda v: torch.tensor(v, dtype=dt_)
    th = torch.atan2(x[:, 1], x[:, 0])                                   # :35
    thdot = x[:, -1]                                                      # :36
    uc = torch.clamp(u[:, 0], -1.0, 1.0) * c(p.max_torque)                # :58
    newthddot = (c(3.0) * c(p.g)) / (c(2.0) * c(p.l)) * torch.sin(th) + c(3.0) / (c(p.m) * c(p.l) ** 2) * uc   # :59
    newthdot = thdot + newthddot * c(p.dt)                                # :60
    newthdot = torch.clamp(newthdot, -p.max_speed, p.max_speed)          # :61
    # dx = [newthdot, newthddot] (:62)
    newth = th + newthdot * c(p.dt)                                       # :40
    newthdot2 = thdot + newthddot * c(p.dt)                               # :41
    newthdot2 = torch.clamp(newthdot2, -p.max_speed, p.max_speed)        # :42
    return torch.stack([torch.cos(newth), torch.sin(newth), newthdot2], dim=1)   # :43


def pendulum_reward(x: torch.Tensor, u: torch.Tensor, p: PendulumParams) -> torch.Tensor:
    """PendulumReward.__call__ (rewards/pendulum_reward.py:27-42): pre-step x, unclipped u, ignores x_next. -> [N]."""
    dt_ = x.dtype
    pi = torch.tensor(math.pi, dtype=dt_)
    theta, omega = torch.atan2(x[:, 1], x[:, 0]), x[:, -1]                # :32
    diff = theta - torch.tensor(p.target_angle, dtype=dt_)                # :34
    diff = torch.remainder(diff + pi, 2 * pi) - pi                        # :35 (python % : sign of divisor)
    return -(p.angle_cost * diff ** 2 + 0.1 * omega ** 2) - p.control_cost * u[:, 0] ** 2   # :38-40


class PendulumSystem:
    """PendulumSystem.step (pendulum_system.py:18-39): x' = next_state(...).mean(); r = reward(x,u,.,x').mean()."""
    x_dim, u_dim = 3, 1

    def __init__(self, params: Optional[PendulumParams] = None):
        self.p = params or PendulumParams()

    def step(self, x, u, **_):
        return pendulum_next_state(x, u, self.p), self.reward(x, u)

    def reward(self, x, u):
        return pendulum_reward(x, u, self.p)


def quadratic_reward(x: torch.Tensor, u: torch.Tensor, target, q, r) -> torch.Tensor:
    """Build-defined generic reward for non-Pendulum shapes: -(sum q (x-t)^2) - sum r u^2."""
    return -(q * (x - target) ** 2).sum(dim=1) - (r * u ** 2).sum(dim=1)


# ------------------------------------------------------------------------------------------ Ensemble
class EnsembleSystem:
    """Learned-ensemble System behind the reference's Dynamics/System seam (base_dynamics.py:15-20,
    base_systems.py:40-52).  NOT IN THE REFERENCE (bsm is never imported — SURVEY §0.1): semantics are
    build-defined, parity-unpinned.

      out_e = MLP_e([x,u])  with out_e = [mu_e (x_dim), raw_std_e (x_dim)]
      mode 'mean' : x' = base + mean_e(mu_e)                  (what System.step's `.mean()` consumes)
      mode 'ts1'  : member m drawn per (env, step);  x' = base + mu_m (+ sigma_m*eps if sample_noise)
      mode 'tsinf': member m = env % E
      base = x if predict_delta else 0;  sigma = softplus(raw) + min_std
    """

    def __init__(self, params: torch.Tensor, dims: Sequence[int], n_members: int, x_dim: int, u_dim: int,
                 act: str = "swish", mode: str = "mean", predict_delta: bool = True, sample_noise: bool = False,
                 min_std: float = 1e-3, reward_fn=None):
        self.params, self.dims, self.E = params, list(dims), n_members
        self.x_dim, self.u_dim, self.act = x_dim, u_dim, act
        self.mode, self.predict_delta, self.sample_noise, self.min_std = mode, predict_delta, sample_noise, min_std
        self.reward_fn = reward_fn

    def reward(self, x, u):
        return self.reward_fn(x, u)

    def step(self, x, u, member_idx=None, model_noise=None, env_index=None):
        X = self.x_dim
        y = nets.ensemble_forward(self.params, self.dims, self.E, torch.cat([x, u], dim=1), self.act)  # [E,N,out]
        base = x if self.predict_delta else torch.zeros_like(x)
        if self.mode == "mean":
            acc = torch.zeros_like(x)
            for e in range(self.E):
                acc = acc + y[e, :, :X]
            xn = base + acc / self.E
        else:
            if self.mode == "tsinf":
                member_idx = (env_index % self.E).long()
            rows = torch.arange(x.shape[0])
            sel = y[member_idx.long(), rows]                        # [N,out]
            xn = base + sel[:, :X]
            if self.sample_noise:
                sigma = F.softplus(sel[:, X:2 * X]) + self.min_std
                xn = xn + sigma * model_noise
        return xn, self.reward(x, u)

"""Differentiable batched torch forms of the built-in Systems' `step`, for BPTT with networks wider than the fused BPTT kernel
takes (bptt_optimizer.py:183-186 accepts any `actor_features` / `critic_features`): the horizon is then walked on the host
(ops.BpttActorGradGeneric) and the model is a node of the torch autograd graph instead of a phase of k_bptt_actor.

    PendulumSystem   dynamics/pendulum_dynamics.py:29-63, rewards/pendulum_reward.py:27-42 (restated; the fused kernels hold the
                     same arithmetic in csrc/rollout.hip)
    EnsembleSystem   'mean' mode without sampled noise (what BPTT requires): x' = [x +] mean_e mu_e([x, u]) with the member MLPs as
                     HIP autograd nodes (ops.HipMlp), quadratic or Pendulum reward in torch
"""
from __future__ import annotations

import math

import torch

from mbpo import _hip
from mbpo.systems.base_systems import SystemState


def pendulum_next_state(x: torch.Tensor, u: torch.Tensor, dp) -> torch.Tensor:
    th = torch.atan2(x[:, 1], x[:, 0])                                                     # :35
    thdot = x[:, -1]                                                                        # :36
    uc = torch.clamp(u[:, 0], -1.0, 1.0) * dp.max_torque                                    # :58
    newthddot = (3.0 * dp.g) / (2.0 * dp.l) * torch.sin(th) + 3.0 / (dp.m * dp.l ** 2) * uc   # :59
    newthdot = torch.clamp(thdot + newthddot * dp.dt, -dp.max_speed, dp.max_speed)          # :60-61 (= :41-42)
    newth = th + newthdot * dp.dt                                                           # :40
    return torch.stack([torch.cos(newth), torch.sin(newth), newthdot], dim=1)               # :43


def pendulum_reward(x: torch.Tensor, u: torch.Tensor, rp) -> torch.Tensor:
    theta, omega = torch.atan2(x[:, 1], x[:, 0]), x[:, -1]                                  # :32
    diff = torch.remainder(theta - rp.target_angle + math.pi, 2 * math.pi) - math.pi        # :34-35
    return -(rp.angle_cost * diff ** 2 + 0.1 * omega ** 2) - rp.control_cost * u[:, 0] ** 2   # :38-40


def quadratic_reward(x: torch.Tensor, u: torch.Tensor, rvec: torch.Tensor, X: int, U: int) -> torch.Tensor:
    t, q, r = rvec[:X], rvec[X:2 * X], rvec[2 * X:2 * X + U]
    return -((x - t) ** 2 * q).sum(-1) - (u ** 2 * r).sum(-1)


class DifferentiableBuiltin:
    """`system.step(x [n, x], u [n, u], params)` -> SystemState with torch-differentiable x_next / reward, for a built-in System
    described by its rollout spec (System.rollout_spec)."""
    fused = False

    def __init__(self, system, spec: dict):
        self.system, self.spec = system, spec
        self.x_dim, self.u_dim = system.x_dim, system.u_dim
        kind = spec["system_kind"]
        if kind == _hip.SYS_ENSEMBLE:
            if spec.get("ens_mode", _hip.ENS_MEAN) != _hip.ENS_MEAN or spec.get("ens_sample_noise", False):
                raise _hip.MbpoHipError("BPTT needs a differentiable model: EnsembleSystem in 'mean' mode without sampled noise")
        elif kind != _hip.SYS_PENDULUM:
            raise _hip.MbpoHipError(f"no differentiable torch form for system kind {kind}")

    def step(self, x, u, system_params):
        from mbpo import ops
        spec, X, U = self.spec, self.x_dim, self.u_dim
        if spec["system_kind"] == _hip.SYS_PENDULUM:
            nxt = pendulum_next_state(x, u, system_params.dynamics_params)
        else:
            mu = ops.HipMlp.apply(spec["dyn_params"], torch.cat([x, u], dim=1), spec["dyn_spec"], None, None)[..., :X].mean(0)
            nxt = x + mu if spec.get("ens_predict_delta", True) else mu
        if spec["reward_kind"] == _hip.REWARD_PENDULUM:
            rew = pendulum_reward(x, u, system_params.reward_params)
        else:
            rew = quadratic_reward(x, u, spec["reward_params"], X, U)
        return SystemState(x_next=nxt, reward=rew, system_params=system_params)

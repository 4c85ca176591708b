"""PPO loss and update — torch-CPU restatement (test infrastructure); gradients via torch.autograd.

  PPOLoss.loss / compute_gae         mbpo/optimizers/policy_optimizers/ppo/losses.py:56-184
  minibatch_step / sgd_step          ppo/ppo.py:142-177  (single optax.adamw(lr, wd) over {policy, value}, no clip: ppo.py:128)
  make_inference_fn extras           ppo/ppo_network.py:59-84  (log_prob, raw_action stored with the rollout)
[3P, unverifiable here] brax make_value_network: MLP hidden + [1], output squeezed (ppo_network.py:41-45).

Flat state layout (the product's): params = [policy | value].  Minibatch rows are the PPO transition rows
[obs(x), action(u), reward, discount, next_obs(x), log_prob, raw_action(u), truncation], shaped [B, T, D].
Randomness: explicit standard-normal tensor for the entropy sample, [B, T, u].
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import nets, scans
from .sac import adamw_step


@dataclass
class PpoConfig:
    x_dim: int
    u_dim: int
    policy_dims: Sequence[int]
    value_dims: Sequence[int]
    policy_act: str = "swish"
    value_act: str = "swish"
    entropy_cost: float = 1e-4
    discounting: float = 0.9
    reward_scaling: float = 1.0
    gae_lambda: float = 0.95
    clipping_epsilon: float = 0.3
    normalize_advantage: bool = True
    lr: float = 1e-4
    wd: float = 1e-5

    @property
    def P(self):
        return nets.n_params(self.policy_dims)

    @property
    def V(self):
        return nets.n_params(self.value_dims)


def split_rows(data: torch.Tensor, X: int, U: int):
    o = X + U
    return dict(obs=data[..., :X], action=data[..., X:o], reward=data[..., o], discount=data[..., o + 1],
                next_obs=data[..., o + 2:o + 2 + X], log_prob=data[..., o + 2 + X], raw_action=data[..., o + 3 + X:o + 3 + X + U],
                truncation=data[..., o + 3 + X + U])


def loss(cfg: PpoConfig, params: torch.Tensor, data: torch.Tensor, ent_noise: torch.Tensor, norm_mean=None, norm_std=None):
    """PPOLoss.loss on data [B, T, D].  Returns (total, dict of the four reported terms, vs, advantages(normalised))."""
    X, U = cfg.x_dim, cfg.u_dim
    pol, val = params[:cfg.P], params[cfg.P:cfg.P + cfg.V]
    t = {k: v.transpose(0, 1) for k, v in split_rows(data, X, U).items()}          # time first (:79)
    obs = nets.normalize(t["obs"], norm_mean, norm_std)
    logits = nets.mlp_forward(pol, cfg.policy_dims, obs, cfg.policy_act)             # :80
    baseline = nets.mlp_forward(val, cfg.value_dims, obs, cfg.value_act)[..., 0]     # :82
    boot = nets.mlp_forward(val, cfg.value_dims, nets.normalize(t["next_obs"][-1], norm_mean, norm_std), cfg.value_act)[..., 0]
    rewards = t["reward"] * cfg.reward_scaling                                       # :87
    truncation = t["truncation"]
    termination = (1 - t["discount"]) * (1 - truncation)                             # :89
    target_lp = nets.log_prob(logits, t["raw_action"])                               # :91-92
    behaviour_lp = t["log_prob"]
    vs_np, adv_np = scans.compute_gae(truncation.detach().numpy(), termination.detach().numpy(), rewards.detach().numpy(),
                                      baseline.detach().numpy(), boot.detach().numpy(), cfg.discounting, cfg.gae_lambda,
                                      dtype=np.float64 if data.dtype == torch.float64 else np.float32)
    vs, adv = torch.from_numpy(vs_np).to(data.dtype), torch.from_numpy(adv_np).to(data.dtype)   # stop_gradient (:183)
    if cfg.normalize_advantage:
        adv = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)                  # :101-102 (jnp std = population)
    rho = torch.exp(target_lp - behaviour_lp)                                        # :103
    s1 = rho * adv
    s2 = torch.clamp(rho, 1 - cfg.clipping_epsilon, 1 + cfg.clipping_epsilon) * adv
    policy_loss = -torch.minimum(s1, s2).mean()                                      # :109
    v_error = vs - baseline
    v_loss = (v_error * v_error).mean() * 0.5                                        # :112-114
    noise = ent_noise.transpose(0, 1)
    entropy = nets.entropy(logits, noise).mean()                                     # :117
    entropy_loss = cfg.entropy_cost * -entropy
    total = policy_loss + v_loss + entropy_loss
    return total, dict(total_loss=total, policy_loss=policy_loss, v_loss=v_loss, entropy_loss=entropy_loss), vs, adv


def grads(cfg: PpoConfig, params, data, ent_noise, norm_mean=None, norm_std=None):
    p = params.clone().requires_grad_(True)
    total, terms, vs, adv = loss(cfg, p, data, ent_noise, norm_mean, norm_std)
    total.backward()
    return p.grad.detach(), {k: float(v.detach()) for k, v in terms.items()}, vs, adv


@dataclass
class PpoState:
    params: torch.Tensor
    adam_m: torch.Tensor
    adam_v: torch.Tensor
    count: int = 0


def init_state(cfg: PpoConfig, gen: torch.Generator, dtype=torch.float32) -> PpoState:
    """init_training_state (ppo.py:265-277)."""
    params = torch.cat([nets.init_mlp_flat(cfg.policy_dims, gen, dtype), nets.init_mlp_flat(cfg.value_dims, gen, dtype)])
    return PpoState(params, torch.zeros_like(params), torch.zeros_like(params), 0)


def minibatch_step(cfg: PpoConfig, st: PpoState, data, ent_noise, norm_mean=None, norm_std=None,
                   grad_override: Optional[torch.Tensor] = None):
    """ppo.py:142-156: one adamw update on one minibatch."""
    g, terms, _, _ = grads(cfg, st.params, data, ent_noise, norm_mean, norm_std)
    if grad_override is not None:
        g = grad_override
    count = st.count + 1
    p, m, v = adamw_step(st.params, g, st.adam_m, st.adam_v, count, cfg.lr, cfg.wd)
    return PpoState(p, m, v, count), terms, g

"""BPTT optimizer arithmetic — torch-CPU restatement (test infrastructure); gradients via torch.autograd THROUGH the model.

  Actor / Critic / act / get_log_prob      mbpo/optimizers/policy_optimizers/bptt_optimizer.py:123-172, 305-325
  Normalizer (batch merge of mean/std/size)  :38-77, 297-303
  actor_loss                                 :327-353
  _train_step (actor update, critic updates) :355-437
  rollout_policy (stop_grads=True)           mbpo/utils/optimizer_utils.py:62-116
  lambda_return                              mbpo/utils/optimizer_utils.py:119-152
  MLP(features, output_dim, swish)           mbpo/utils/network_utils.py:5-17
[3P, unverifiable here] flax Dense default init is lecun_normal (the BPTT nets use plain nn.Dense, network_utils.py:13-16);
  optax.apply_if_finite skips a non-finite update; optax.l2_loss(p, t) = 0.5*(p-t)^2.

Log-prob for action_dim > 1: the reference's `[H,A] - [H]` broadcast (:144-152) is only shape-valid for A == 1 (or A == H),
where its mean equals mean_t(logN_t) - mean_t(logdet_t).  SURVEY §8a B5 recommends, and this restatement (and the product)
uses, log_prob_t = sum_A logN - sum_A log(1 - a^2), which coincides with the reference for A == 1.

Flat layouts (the product's): actor params [P]; critic params [2*C] = [critic_1 | critic_2] (each MLP features + [1]).
Randomness: explicit standard-normal tensor act_noise [n, H, A].
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import nets
from .sac import adamw_step

EPS = 1e-8


@dataclass
class BpttConfig:
    x_dim: int
    u_dim: int
    actor_dims: Sequence[int]      # [x, features..., 2u]
    critic_dims: Sequence[int]     # [x, features..., 1]
    horizon: int = 20
    act: str = "swish"
    init_stddev: float = 1.0
    discount: float = 0.99
    lambda_: float = 0.97
    ent_coef: float = 0.005
    lr_actor: float = 1e-3
    wd_actor: float = 1e-5
    lr_critic: float = 1e-3
    wd_critic: float = 1e-5
    tau: float = 0.005

    @property
    def P(self):
        return nets.n_params(self.actor_dims)

    @property
    def C(self):
        return nets.n_params(self.critic_dims)


def inv_softplus(x: float) -> float:
    return math.log(math.exp(x) - 1.0) if x < 20.0 else x          # :107-108


def actor_forward(cfg: BpttConfig, params, obs_n):
    """Actor.__call__ (:131-142): mu, sig = split(MLP(obs)); sig = clip(softplus(sig + inv_softplus(init_std)), 1e-6, 1e2)."""
    out = nets.mlp_forward(params, cfg.actor_dims, obs_n, cfg.act)
    u = cfg.u_dim
    mu, raw = out[..., :u], out[..., u:]
    sig = torch.clamp(F.softplus(raw + inv_softplus(cfg.init_stddev)), 1e-6, 1e2)
    return mu, sig


def critic_forward(cfg: BpttConfig, params, obs_n):
    """Critic.__call__ (:155-172): two independent MLPs -> (v1, v2)."""
    C = cfg.C
    v1 = nets.mlp_forward(params[:C], cfg.critic_dims, obs_n, cfg.act)[..., 0]
    v2 = nets.mlp_forward(params[C:2 * C], cfg.critic_dims, obs_n, cfg.act)[..., 0]
    return v1, v2


def normalize(x, mean, std):
    return (x - mean) / std


def squash(x):
    return torch.clamp(torch.tanh(x), -0.999, 0.999)                # act :313-317


def lambda_return_t(reward, next_values, discount, lambda_):
    """lambda_return for [n, H] tensors (differentiable)."""
    H = reward.shape[1]
    inputs = reward + discount * next_values * (1 - lambda_)
    agg = next_values[:, -1]
    outs = [None] * H
    for t in range(H - 1, -1, -1):
        agg = inputs[:, t] + discount * lambda_ * agg
        outs[t] = agg
    return torch.stack(outs, dim=1)


def log_prob_steps(mu, sig, action):
    """Actor.get_log_prob (:144-152) per step: u = atanh(clip(a)); sum_A logN(u; mu, sig) - sum_A log(1 - a^2).
    At action_dim = 1 its mean over the horizon equals the mean of the reference's [H,1] - [H] broadcast (SURVEY §8a B5;
    tests/golden/semantics_kat.json: bptt_log_prob)."""
    a_c = torch.clamp(action, -1 + EPS, 1 - EPS)
    u = 0.5 * torch.log((1 + a_c) / (1 - a_c))
    log_l = (-0.5 * ((u - mu) / sig) ** 2 - torch.log(sig) - 0.5 * math.log(2 * math.pi)).sum(-1)
    return log_l - torch.log(1 - action ** 2).sum(-1)


def actor_loss(cfg: BpttConfig, system, actor_params, target_critic_params, init_states, act_noise, s_mean, s_std, r_mean, r_std):
    """vmap(actor_loss) + .mean() over initial states (:361-372).  `system.step` must be differentiable (oracle systems are).

    Returns (loss, aux) with aux = dict(entropy_loss, lambda_values [n,H], observation, action, reward, next_observation)."""
    n, H = init_states.shape[0], cfg.horizon
    obs = init_states
    obs_l, act_l, rew_l, nobs_l = [], [], [], []
    for t in range(H):                                               # rollout_policy :79-101
        mu, sig = actor_forward(cfg, actor_params, normalize(obs.detach(), s_mean, s_std))   # policy(stop_gradient(obs))
        a = squash(mu + act_noise[:, t] * sig)
        nxt, r = system.step(obs, a)
        obs_l.append(obs); act_l.append(a); rew_l.append(r); nobs_l.append(nxt)
        obs = nxt
    observation, action = torch.stack(obs_l, 1), torch.stack(act_l, 1)
    reward, next_observation = torch.stack(rew_l, 1), torch.stack(nobs_l, 1)
    next_n = normalize(next_observation, s_mean, s_std)             # :338-339
    reward_n = normalize(reward, r_mean, r_std)                     # :340-341
    v1, v2 = critic_forward(cfg, target_critic_params, next_n)      # :342
    bootstrap = torch.minimum(v1, v2)                               # :343
    lam = lambda_return_t(reward_n, bootstrap, cfg.discount, cfg.lambda_)   # :344
    obs_n = normalize(observation, s_mean, s_std)                   # :345  (NOT stop-gradiented)
    disc = torch.cat([torch.ones(1, dtype=lam.dtype), torch.full((H - 1,), cfg.discount, dtype=lam.dtype)]).cumprod(0)   # :346-348
    mu, sig = actor_forward(cfg, actor_params, obs_n)               # get_log_prob :144-152
    log_l = log_prob_steps(mu, sig, action)
    entropy_loss = -log_l.mean(dim=1)                               # per initial state :351
    loss = -(lam * disc).mean(dim=1) + entropy_loss * cfg.ent_coef   # :352
    aux = dict(entropy_loss=entropy_loss.mean(), lambda_values=lam, observation=observation, action=action, reward=reward,
               next_observation=next_observation)
    return loss.mean(), aux


def actor_grads(cfg, system, actor_params, target_critic_params, init_states, act_noise, s_mean, s_std, r_mean, r_std):
    p = actor_params.clone().requires_grad_(True)
    loss, aux = actor_loss(cfg, system, p, target_critic_params, init_states, act_noise, s_mean, s_std, r_mean, r_std)
    loss.backward()
    aux = {k: v.detach() for k, v in aux.items()}
    return p.grad.detach(), float(loss.detach()), aux


def critic_loss(cfg: BpttConfig, critic_params, obs, lamb, s_mean, s_std):
    """critic_loss_fn (:398-404): 0.5*(mean l2(v1,lamb) + mean l2(v2,lamb)), l2 = 0.5*(.)^2."""
    v1, v2 = critic_forward(cfg, critic_params, normalize(obs, s_mean, s_std))
    return 0.5 * ((0.5 * (v1 - lamb) ** 2).mean() + (0.5 * (v2 - lamb) ** 2).mean())


def critic_grads(cfg, critic_params, obs, lamb, s_mean, s_std):
    p = critic_params.clone().requires_grad_(True)
    loss = critic_loss(cfg, p, obs, lamb, s_mean, s_std)
    loss.backward()
    return p.grad.detach(), float(loss.detach())


def normalizer_update(x: torch.Tensor, mean, std, size):
    """Normalizer.update (:52-67)."""
    new_size = x.shape[0]
    total = new_size + size
    new_mean = (mean * size + x.sum(0)) / total
    new_s_n = std ** 2 * size + ((x - new_mean) ** 2).sum(0) + size * (mean - new_mean) ** 2
    new_std = torch.sqrt(new_s_n / total)
    return new_mean, torch.maximum(new_std, torch.full_like(new_std, EPS)), total


def apply_if_finite_adamw(p, g, m, v, count, lr, wd):
    """optax.apply_if_finite(adamw): skip the whole update (params, moments, count) when any gradient is non-finite."""
    if not bool(torch.isfinite(g).all()):
        return p, m, v, count
    count = count + 1
    p2, m2, v2 = adamw_step(p, g, m, v, count, lr, wd)
    return p2, m2, v2, count


# ------------------------------------------------------------------------------------------------ differentiable oracle systems
class TorchEnsembleSystem:
    """Differentiable twin of oracle.systems.EnsembleSystem ('mean' mode, predict_delta) with a quadratic reward."""

    def __init__(self, params, dims, n_members, x_dim, u_dim, target, q, r, act="swish"):
        self.params, self.dims, self.E, self.x_dim, self.u_dim, self.act = params, list(dims), n_members, x_dim, u_dim, act
        self.target, self.q, self.r = target, q, r

    def step(self, x, u):
        y = nets.ensemble_forward(self.params, self.dims, self.E, torch.cat([x, u], dim=1), self.act)
        acc = torch.zeros_like(x)
        for e in range(self.E):
            acc = acc + y[e, :, :self.x_dim]
        xn = x + acc / self.E
        rew = -(self.q * (x - self.target) ** 2).sum(1) - (self.r * u ** 2).sum(1)
        return xn, rew


class TorchPendulumSystem:
    """Differentiable PendulumSystem.step (pendulum_system.py:18-39)."""
    x_dim, u_dim = 3, 1

    def __init__(self, p=None):
        from .systems import PendulumParams
        self.p = p or PendulumParams()

    def step(self, x, u):
        from .systems import pendulum_next_state, pendulum_reward
        return pendulum_next_state(x, u, self.p), pendulum_reward(x, u, self.p)


# ------------------------------------------------------------------------------------------------ whole train steps
class CpuBpttLoop:
    """`train`'s scan body (bptt_optimizer.py:463-522) composed from the pieces above, with the product's Philox streams
    (seeds = (initial-state sampling, action noise, critic minibatch); Philox offset = train-step index)."""

    def __init__(self, cfg: BpttConfig, system, actor_params, critic_params, true_rows, n, critic_updates, seeds,
                 buffer_size=4096, normalize=True, insert=True):
        from . import replay as oreplay
        self.cfg, self.system, self.n, self.K = cfg, system, n, critic_updates
        self.seeds, self.do_norm, self.do_insert = seeds, normalize, insert
        self.ap, self.cp, self.tp = actor_params.clone(), critic_params.clone(), critic_params.clone()
        z = torch.zeros_like
        self.am, self.av, self.ac = z(self.ap), z(self.ap), 0
        self.cm, self.cv, self.cc = z(self.cp), z(self.cp), 0
        X = cfg.x_dim
        self.s_mean, self.s_std, self.s_size = torch.zeros(X), torch.ones(X), 0
        self.r_mean, self.r_std, self.r_size = torch.zeros(1), torch.ones(1), 0
        self.D = 2 * X + cfg.u_dim + 2
        self.queue = oreplay.UniformSamplingQueue(buffer_size, self.D, n)
        self.buf = self.queue.insert(self.queue.init(), true_rows.numpy())
        self.step_idx = 0

    def step(self):
        import numpy as np
        from . import philox
        cfg, n, H, X, U = self.cfg, self.n, self.cfg.horizon, self.cfg.x_dim, self.cfg.u_dim
        obs_seed, act_seed, critic_seed = self.seeds
        _, rows = self.queue.sample(self.buf, obs_seed, self.step_idx, n)
        x0 = torch.from_numpy(rows[:, :X].copy())
        noise = torch.from_numpy(philox.philox_normal(act_seed, self.step_idx, philox.STREAM_POLICY_NOISE,
                                                      np.arange(n * H * U, dtype=np.uint64))).reshape(n, H, U)
        g, loss, aux = actor_grads(cfg, self.system, self.ap, self.tp, x0, noise, self.s_mean, self.s_std, self.r_mean[0], self.r_std[0])
        self.ap, self.am, self.av, self.ac = apply_if_finite_adamw(self.ap, g, self.am, self.av, self.ac, cfg.lr_actor, cfg.wd_actor)
        R = n * H
        B = -(-R // self.K)
        idx = philox.philox_randint(critic_seed, self.step_idx, philox.STREAM_REPLAY, np.arange(self.K * B, dtype=np.uint64), 0, R)
        obs_flat, lam_flat = aux["observation"].reshape(R, X), aux["lambda_values"].reshape(R)
        closs = gnorm = None
        for k in range(self.K):
            j = torch.from_numpy(idx[k * B:(k + 1) * B].astype(np.int64))
            cg, closs = critic_grads(cfg, self.cp, obs_flat[j], lam_flat[j], self.s_mean, self.s_std)
            gnorm = float(cg.norm())
            before = self.cc
            self.cp, self.cm, self.cv, self.cc = apply_if_finite_adamw(self.cp, cg, self.cm, self.cv, self.cc, cfg.lr_critic, cfg.wd_critic)
            self.tp = (1 - cfg.tau) * self.tp + cfg.tau * self.cp
        rows_out = torch.cat([obs_flat, aux["action"].reshape(R, U), aux["reward"].reshape(R, 1), torch.ones(R, 1),
                              aux["next_observation"].reshape(R, X)], dim=1)
        if self.do_norm:
            self.s_mean, self.s_std, self.s_size = normalizer_update(obs_flat, self.s_mean, self.s_std, self.s_size)
            self.r_mean, self.r_std, self.r_size = normalizer_update(aux["reward"].reshape(R, 1), self.r_mean, self.r_std, self.r_size)
        if self.do_insert:
            self.buf = self.queue.insert(self.buf, rows_out.numpy())
        self.step_idx += 1
        return dict(actor_loss=loss, actor_grad_norm=float(g.norm()), critic_loss=closs, critic_grad_norm=gnorm, rows=rows_out)

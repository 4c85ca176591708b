"""SAC losses, sgd_step and optimizer — torch-CPU restatement (test infrastructure).

Gradients come from torch.autograd, i.e. they are derived independently of the product's hand-written backward.

  alpha_loss / critic_loss / actor_loss   mbpo/optimizers/policy_optimizers/sac/losses.py:61-125
  sgd_step (three updates at OLD params, Polyak with NEW q)   sac/sac.py:227-281
  gradient_update_fn                                          sac/utils.py:36-63
  optimizers: optax.chain(clip_by_global_norm(max_grad_norm), adamw(lr, weight_decay))   sac/sac.py:175-186
[3P, unverifiable here] optax semantics restated from upstream:
  clip_by_global_norm: g_norm = sqrt(sum g^2); g if g_norm < max_norm else (g / g_norm) * max_norm
  adamw: mu = b1 mu + (1-b1) g; nu = b2 nu + (1-b2) g^2; count += 1; mu_hat = mu/(1-b1^count); nu_hat = nu/(1-b2^count);
         u = mu_hat / (sqrt(nu_hat) + eps) + wd * p;  p <- p - lr * u      (b1=.9, b2=.999, eps=1e-8, eps_root=0)

Flat state layout = the product's (include/mbpo_hip.h): params = [policy | critic0 | critic1 | log_alpha].
Randomness: explicit standard-normal tensors noise_alpha / noise_critic / noise_actor, each [B,u].
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import nets


def floor_divide(x1: torch.Tensor, x2: float) -> torch.Tensor:
    """[3P] jnp.floor_divide on floats (the `//` of losses.py:95): remainder-based division, then rounded."""
    x2t = torch.as_tensor(x2, dtype=x1.dtype)
    mod = torch.fmod(x1, x2t)
    div = (x1 - mod) / x2t
    ind = (mod != 0) & ((x2t < 0) != (mod < 0))
    div = torch.where(ind, div - 1, div)
    return torch.round(div)


@dataclass
class SacConfig:
    x_dim: int
    u_dim: int
    policy_dims: Sequence[int]
    q_dims: Sequence[int]
    policy_act: str = "swish"
    q_act: str = "swish"
    discounting: float = 0.9
    reward_scaling: float = 1.0
    target_entropy: Optional[float] = None      # default -0.5*u_dim (losses.py:49-50)
    tau: float = 0.005
    lr_policy: float = 1e-4
    lr_q: float = 1e-4
    lr_alpha: float = 1e-4
    wd_policy: float = 0.0
    wd_q: float = 0.0
    wd_alpha: float = 0.0
    max_grad_norm: float = 1e5
    non_equidistant_time: bool = False          # losses.py:39-59, 90-98
    continuous_discounting: float = 0.0
    min_time_between_switches: float = 0.0
    max_time_between_switches: float = 0.0
    env_dt: float = 0.0

    @property
    def P(self):
        return nets.n_params(self.policy_dims)

    @property
    def Q(self):
        return nets.n_params(self.q_dims)

    @property
    def NP(self):
        return self.P + 2 * self.Q + 1

    @property
    def h_target(self):
        return -0.5 * self.u_dim if self.target_entropy is None else self.target_entropy


def split_batch(batch: torch.Tensor, X: int, U: int):
    """Flattened Transition row: [obs, action, reward, discount, next_obs, truncation] (sac/sac.py:194-200)."""
    return dict(obs=batch[:, :X], action=batch[:, X:X + U], reward=batch[:, X + U], discount=batch[:, X + U + 1],
                next_obs=batch[:, X + U + 2:2 * X + U + 2], truncation=batch[:, 2 * X + U + 2])


def losses(cfg: SacConfig, params: torch.Tensor, target_q: torch.Tensor, batch: torch.Tensor, noise_alpha, noise_critic,
           noise_actor, norm_mean=None, norm_std=None):
    """Returns (alpha_loss, critic_loss, actor_loss) as functions of the SAME params tensor, but each loss only lets
    gradient flow to its own variables — exactly like the three value_and_grad calls of sgd_step (sac.py:234-258)."""
    P, Q = cfg.P, cfg.Q
    pol, qp, log_alpha = params[:P], params[P:P + 2 * Q], params[P + 2 * Q]
    t = split_batch(batch, cfg.x_dim, cfg.u_dim)
    obs = nets.normalize(t["obs"], norm_mean, norm_std)
    nobs = nets.normalize(t["next_obs"], norm_mean, norm_std)
    alpha_c = torch.exp(log_alpha).detach()          # alpha = exp(training_state.alpha_params) passed as a constant (:240)

    # alpha_loss (losses.py:61-72): grads wrt log_alpha only
    dist = nets.mlp_forward(pol.detach(), cfg.policy_dims, obs, cfg.policy_act)
    z = nets.sample_no_postprocessing(dist, noise_alpha)
    lp = nets.log_prob(dist, z)
    alpha_loss = (torch.exp(log_alpha) * (-lp - cfg.h_target).detach()).mean()

    # critic_loss (losses.py:74-110): grads wrt q params only
    q_old = nets.q_forward(qp, cfg.q_dims, obs, t["action"], cfg.q_act)
    ndist = nets.mlp_forward(pol.detach(), cfg.policy_dims, nobs, cfg.policy_act)
    nz = nets.sample_no_postprocessing(ndist, noise_critic)
    nlp = nets.log_prob(ndist, nz)
    na = nets.postprocess(nz)
    next_q = nets.q_forward(target_q, cfg.q_dims, nobs, na, cfg.q_act)
    next_v = next_q.min(dim=-1).values - alpha_c * nlp
    if cfg.non_equidistant_time:                                                      # losses.py:90-96
        pseudo = t["action"][..., -1]
        t_lower, t_upper = cfg.min_time_between_switches, cfg.max_time_between_switches
        tfa = (t_upper - t_lower) / 2 * pseudo + (t_upper + t_lower) / 2
        tfa = floor_divide(tfa, cfg.env_dt) * cfg.env_dt
        discounting = torch.exp(-cfg.continuous_discounting * tfa)
    else:
        discounting = cfg.discounting
    target = (t["reward"] * cfg.reward_scaling + t["discount"] * discounting * next_v).detach()
    q_error = (q_old - target[:, None]) * (1 - t["truncation"])[:, None]
    critic_loss = 0.5 * (q_error ** 2).mean()

    # actor_loss (losses.py:112-125): grads wrt policy params only (q_params = OLD training_state.q_params, sac.py:253)
    adist = nets.mlp_forward(pol, cfg.policy_dims, obs, cfg.policy_act)
    az = nets.sample_no_postprocessing(adist, noise_actor)
    alp = nets.log_prob(adist, az)
    aa = nets.postprocess(az)
    q_act = nets.q_forward(qp.detach(), cfg.q_dims, obs, aa, cfg.q_act)
    actor_loss = (alpha_c * alp - q_act.min(dim=-1).values).mean()
    return alpha_loss, critic_loss, actor_loss


def grads(cfg: SacConfig, params, target_q, batch, noise_alpha, noise_critic, noise_actor, norm_mean=None, norm_std=None):
    """Flat gradient [NP] + the three loss values."""
    p = params.clone().requires_grad_(True)
    al, cl, ac = losses(cfg, p, target_q, batch, noise_alpha, noise_critic, noise_actor, norm_mean, norm_std)
    # the three losses touch disjoint variable groups, so one backward of the sum equals three separate value_and_grads
    (al + cl + ac).backward()
    return p.grad.detach(), (float(cl.detach()), float(ac.detach()), float(al.detach()))


def clip_by_global_norm(g: torch.Tensor, max_norm: float) -> torch.Tensor:
    n = torch.sqrt((g ** 2).sum())
    return g if bool(n < max_norm) else (g / n) * max_norm


def adamw_step(p, g, m, v, count: int, lr: float, wd: float, b1=0.9, b2=0.999, eps=1e-8):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    m_hat = m / (1 - b1 ** count)
    v_hat = v / (1 - b2 ** count)
    u = m_hat / (torch.sqrt(v_hat) + eps) + wd * p
    return p - lr * u, m, v


@dataclass
class SacState:
    params: torch.Tensor
    target_q: torch.Tensor
    adam_m: torch.Tensor
    adam_v: torch.Tensor
    count: int = 0

    def clone(self):
        return SacState(self.params.clone(), self.target_q.clone(), self.adam_m.clone(), self.adam_v.clone(), self.count)


def init_state(cfg: SacConfig, gen: torch.Generator, init_log_alpha: float = 0.0, dtype=torch.float32) -> SacState:
    """init_training_state (sac.py:376-402): target_q_params = q_params; optimizer states zero."""
    pol = nets.init_mlp_flat(cfg.policy_dims, gen, dtype)
    q0 = nets.init_mlp_flat(cfg.q_dims, gen, dtype)
    q1 = nets.init_mlp_flat(cfg.q_dims, gen, dtype)
    params = torch.cat([pol, q0, q1, torch.tensor([init_log_alpha], dtype=dtype)])
    return SacState(params, params[cfg.P:cfg.P + 2 * cfg.Q].clone(), torch.zeros_like(params), torch.zeros_like(params), 0)


def sgd_step(cfg: SacConfig, st: SacState, batch, noise_alpha, noise_critic, noise_actor, norm_mean=None, norm_std=None,
             grad_override: Optional[torch.Tensor] = None):
    """SAC.sgd_step (sac.py:227-281).  Returns (new state, metrics dict, flat grad)."""
    g, (cl, ac, al) = grads(cfg, st.params, st.target_q, batch, noise_alpha, noise_critic, noise_actor, norm_mean, norm_std)
    if grad_override is not None:   # multi-rank tests: the pmean'd gradient
        g = grad_override
    P, Q = cfg.P, cfg.Q
    count = st.count + 1
    new_p, new_m, new_v = st.params.clone(), st.adam_m.clone(), st.adam_v.clone()
    for sl, lr, wd in ((slice(0, P), cfg.lr_policy, cfg.wd_policy), (slice(P, P + 2 * Q), cfg.lr_q, cfg.wd_q),
                       (slice(P + 2 * Q, P + 2 * Q + 1), cfg.lr_alpha, cfg.wd_alpha)):
        gg = clip_by_global_norm(g[sl], cfg.max_grad_norm)
        new_p[sl], new_m[sl], new_v[sl] = adamw_step(st.params[sl], gg, st.adam_m[sl], st.adam_v[sl], count, lr, wd)
    new_tq = st.target_q * (1 - cfg.tau) + new_p[P:P + 2 * Q] * cfg.tau          # sac.py:260-261
    metrics = {"critic_loss": cl, "actor_loss": ac, "alpha_loss": al, "alpha": float(torch.exp(new_p[-1]))}
    return SacState(new_p, new_tq, new_m, new_v, count), metrics, g

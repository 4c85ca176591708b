"""EnsembleDynamics / EnsembleSystem — the learned-ensemble System behind the reference's Dynamics/System seam
(base_dynamics.py:15-20, base_systems.py:40-52).

NOT IN THE REFERENCE: `bsm` is declared in setup.py:22 but never imported; this is the slot MBPO's learned
model plugs into.  Semantics (build-defined, parity-unpinned; restated in oracle/systems.py):
    out_e = MLP_e([x, u]) = [mu_e (x_dim), raw_std_e (x_dim)],  sigma_e = softplus(raw_std_e) + min_std
    base  = x if predict_delta else 0
    'mean' : x' = base + mean_e(mu_e)               — what System.step's `.mean()` consumes
    'ts1'  : member drawn per (env, step);  x' = base + mu_m (+ sigma_m * eps when sample_noise)   — MBPO-style
    'tsinf': member = env % E
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from mbpo import _hip, ops
from mbpo.systems.base_systems import System, SystemParams
from mbpo.systems.dynamics.base_dynamics import Dynamics, Normal
from mbpo.systems.rewards.base_rewards import Reward
from mbpo.utils import keys as K

_MODES = {"mean": _hip.ENS_MEAN, "ts1": _hip.ENS_TS1, "tsinf": _hip.ENS_TSINF}


@dataclass
class EnsembleDynamicsParams:
    params: torch.Tensor          # flat [E * P] device tensor (layout: include/mbpo_hip.h)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


def lecun_uniform_flat(dims: Sequence[int], gen: torch.Generator) -> torch.Tensor:
    """flax lecun_uniform kernels U(+-sqrt(3/fan_in)), zero biases (sac/networks.py:23)."""
    parts = []
    for i in range(len(dims) - 1):
        bound = math.sqrt(3.0 / dims[i])
        parts.append(((torch.rand(dims[i], dims[i + 1], generator=gen, dtype=torch.float64) * 2 - 1) * bound).reshape(-1).float())
        parts.append(torch.zeros(dims[i + 1]))
    return torch.cat(parts)


class EnsembleDynamics(Dynamics[EnsembleDynamicsParams]):
    def __init__(self, x_dim: int, u_dim: int, n_members: int = 5, hidden_layer_sizes: Sequence[int] = (64, 64, 64),
                 activation: str = "swish", device=None):
        super().__init__(x_dim, u_dim)
        self.n_members = n_members
        self.dims = [x_dim + u_dim, *hidden_layer_sizes, 2 * x_dim]
        self.spec = ops.MlpSpec(self.dims, activation, n_members)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

    def init_params(self, key: int) -> EnsembleDynamicsParams:
        gen = torch.Generator().manual_seed(K.PRNGKey(key) % (2 ** 63))
        flat = torch.cat([lecun_uniform_flat(self.dims, gen) for _ in range(self.n_members)])
        return EnsembleDynamicsParams(params=flat.to(self.device))

    def fit(self, dynamics_params: EnsembleDynamicsParams, rows: torch.Tensor, num_steps: int, batch_size: int = 256,
            learning_rate: float = 1e-3, weight_decay: float = 0.0, key: int = 0, predict_delta: bool = True,
            min_std: float = 1e-3, n_rows: Optional[int] = None, next_obs_off: Optional[int] = None):
        """Model learning (N3 — not in the reference, whose model would come from `bsm`): `num_steps` AdamW steps on the
        members' Gaussian negative log-likelihood, each member on its own bootstrapped minibatch (sampling with replacement
        from rows[:n_rows]; Philox randint on the device).  `rows` are true-buffer transition rows (obs, action, reward,
        discount, next_obs, ...).  Updates dynamics_params.params in place; returns (dynamics_params, losses [num_steps, E])."""
        dev = self.device
        rows = rows.to(dev, torch.float32).contiguous()
        R = int(rows.shape[0] if n_rows is None else n_rows)
        if R <= 0:
            raise ValueError("no transitions to fit on")
        E = self.n_members
        if getattr(self, "_fit_cfg", None) != (batch_size, predict_delta, min_std, learning_rate, weight_decay):
            self._nll = ops.EnsembleNllGrad(x_dim=self.x_dim, u_dim=self.u_dim, spec=self.spec, batch=batch_size, device=dev,
                                            predict_delta=predict_delta, min_std=min_std)
            self._opt = ops.AdamW(E * self.spec.n_params, dev, learning_rate, weight_decay, apply_if_finite=True)
            self._fit_cfg = (batch_size, predict_delta, min_std, learning_rate, weight_decay)
            self._fit_state = torch.tensor([R, 0, 0, R], device=dev, dtype=torch.int32)
            self._fit_idx = torch.zeros(E * batch_size, device=dev, dtype=torch.int32)
            self._fit_scratch = torch.zeros(E * batch_size, 1, device=dev, dtype=torch.float32)
        self._fit_state[0] = R
        self._fit_state[3] = R
        losses = torch.zeros(num_steps, E, device=dev, dtype=torch.float32)
        seed = K.PRNGKey(key)
        col0 = rows[:, :1].contiguous()          # the sampler gathers something; one column keeps that cheap
        for it in range(num_steps):
            ops.replay_sample(col0, self._fit_state, E * batch_size, seed=seed, offset=it, out=self._fit_scratch, idx_out=self._fit_idx)
            g = self._nll(dynamics_params.params, rows, self._fit_idx.view(E, batch_size), next_obs_off=next_obs_off)
            self._opt.step(dynamics_params.params, g)
            losses[it].copy_(self._nll.metrics)
        return dynamics_params, losses

    def member_outputs(self, x: torch.Tensor, u: torch.Tensor, dynamics_params: EnsembleDynamicsParams) -> torch.Tensor:
        """[E, N, 2*x_dim] raw member outputs — mbpo_ensemble_mlp_forward."""
        xu = torch.cat([x.reshape(-1, self.x_dim), u.reshape(-1, self.u_dim)], dim=1).to(self.device, torch.float32).contiguous()
        return ops.ensemble_mlp_forward(dynamics_params.params, self.spec, xu)

    def next_state(self, x, u, dynamics_params, predict_delta: bool = True, min_std: float = 1e-3):
        """Mixture moments over members: mean = E_e[mu_e], std = sqrt(E_e[sigma_e^2] + Var_e[mu_e])."""
        y = self.member_outputs(x, u, dynamics_params)
        X = self.x_dim
        mu = y[..., :X] + (x.reshape(-1, X) if predict_delta else 0.0)
        sig = torch.nn.functional.softplus(y[..., X:]) + min_std
        mean = mu.mean(dim=0)
        std = torch.sqrt((sig ** 2).mean(dim=0) + mu.var(dim=0, unbiased=False))
        if x.dim() == 1:
            mean, std = mean[0], std[0]
        return Normal(mean, std), dynamics_params


class EnsembleSystem(System):
    def __init__(self, dynamics: EnsembleDynamics, reward: Reward, mode: str = "mean", predict_delta: bool = True,
                 sample_noise: bool = False, min_std: float = 1e-3):
        super().__init__(dynamics=dynamics, reward=reward)
        if mode not in _MODES:
            raise ValueError(f"mode must be one of {sorted(_MODES)}")
        self.mode, self.predict_delta, self.sample_noise, self.min_std = mode, predict_delta, sample_noise, min_std

    def rollout_spec(self, system_params: SystemParams, device) -> dict:
        rp = system_params.reward_params
        ck = (repr(rp), str(device))
        if getattr(self, "_rspec_key", None) != ck:    # cached: no H2D copy inside a captured graph
            self._rspec = self.reward.kernel_spec(rp, device)
            self._rspec_key = ck
        kind, rvec = self._rspec
        return dict(system_kind=_hip.SYS_ENSEMBLE, dyn_params=system_params.dynamics_params.params, dyn_spec=self.dynamics.spec,
                    ens_mode=_MODES[self.mode], ens_predict_delta=self.predict_delta, ens_sample_noise=self.sample_noise,
                    ens_min_std=self.min_std, reward_kind=kind, reward_params=rvec)

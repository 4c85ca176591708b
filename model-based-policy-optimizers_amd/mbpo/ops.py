"""Tensor-level wrappers over the C-ABI (one function per entry point of include/mbpo_hip.h).

These are the only callers of libmbpo_hip.so; everything in mbpo.systems / mbpo.optimizers goes through them.
Inputs must be contiguous CUDA(HIP) tensors — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _hip
from ._hip import check, current_stream_ptr, load, mlp_desc, ptr, require_device_tensor as _req


@dataclass
class MlpSpec:
    """Shape of an MLP (or of n_nets identical MLPs laid out consecutively in one flat tensor)."""
    dims: Sequence[int]
    activation: str = "swish"
    n_nets: int = 1

    @property
    def n_params(self) -> int:
        d = list(self.dims)
        return sum(d[i] * d[i + 1] + d[i + 1] for i in range(len(d) - 1))

    @property
    def total_params(self) -> int:
        return self.n_params * self.n_nets

    def desc(self, params: torch.Tensor) -> _hip.MlpDesc:
        _req(params, "params")
        return mlp_desc(params, self.dims, self.activation, self.n_nets)


def ensemble_mlp_forward(params: torch.Tensor, spec: MlpSpec, x: torch.Tensor, shared_input: bool = True) -> torch.Tensor:
    """y[e, n, :] = MLP_e(x[n]) — R2 of SURVEY §8a."""
    lib = load()
    _req(x, "x")
    d = spec.desc(params)
    din, dout = spec.dims[0], spec.dims[-1]
    if shared_input:
        if x.dim() != 2 or x.shape[1] != din:
            raise ValueError(f"x must be [N,{din}], got {tuple(x.shape)}")
        n = x.shape[0]
    else:
        if x.dim() != 3 or x.shape[0] != spec.n_nets or x.shape[2] != din:
            raise ValueError(f"x must be [{spec.n_nets},N,{din}], got {tuple(x.shape)}")
        n = x.shape[1]
    y = torch.empty((spec.n_nets, n, dout), device=x.device, dtype=torch.float32)
    check(lib.mbpo_ensemble_mlp_forward(C.byref(d), x.data_ptr(), int(shared_input), y.data_ptr(), n,
                                        current_stream_ptr()), "mbpo_ensemble_mlp_forward")
    return y


def transition_row_len(x_dim: int, u_dim: int, ppo_extras: bool = False) -> int:
    return 2 * x_dim + u_dim + 3 + ((1 + u_dim) if ppo_extras else 0)


def model_rollout(*, policy_params: torch.Tensor, policy_spec: MlpSpec, x_dim: int, u_dim: int,
                  obs: torch.Tensor, first_obs: torch.Tensor, steps: torch.Tensor, done: torch.Tensor,
                  n_steps: int, episode_length: int, action_repeat: int = 1,
                  system_kind: int = _hip.SYS_PENDULUM, dyn_params: Optional[torch.Tensor] = None,
                  dyn_spec: Optional[MlpSpec] = None, ens_mode: int = _hip.ENS_MEAN, ens_predict_delta: bool = True,
                  ens_sample_noise: bool = False, ens_min_std: float = 1e-3,
                  reward_kind: int = _hip.REWARD_PENDULUM, reward_params: torch.Tensor = None,
                  sys_params: Optional[torch.Tensor] = None,
                  norm_mean: Optional[torch.Tensor] = None, norm_std: Optional[torch.Tensor] = None,
                  deterministic: bool = False, ppo_extras: bool = False, env_major: bool = False,
                  policy_noise: Optional[torch.Tensor] = None, model_noise: Optional[torch.Tensor] = None,
                  member_idx: Optional[torch.Tensor] = None, seed: int = 0, offset: int = 0,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused S-step model rollout for N envs (R1-R8).  Updates obs/steps/done in place; returns rows [S*N, D]."""
    lib = load()
    n_envs = obs.shape[0]
    D = transition_row_len(x_dim, u_dim, ppo_extras)
    for t, nm in ((obs, "obs"), (first_obs, "first_obs"), (steps, "steps"), (done, "done"), (reward_params, "reward_params")):
        _req(t, nm)
    if obs.shape != (n_envs, x_dim) or first_obs.shape != (n_envs, x_dim):
        raise ValueError("obs/first_obs must be [N, x_dim]")
    if steps.shape != (n_envs,) or done.shape != (n_envs,):
        raise ValueError("steps/done must be [N]")
    if out is None:
        out = torch.empty((n_steps * n_envs, D), device=obs.device, dtype=torch.float32)
    else:
        _req(out, "out")
        if out.shape != (n_steps * n_envs, D):
            raise ValueError(f"out must be [{n_steps * n_envs},{D}]")
    d = _hip.RolloutDesc()
    d.policy = policy_spec.desc(policy_params)
    if system_kind == _hip.SYS_ENSEMBLE:
        if dyn_params is None or dyn_spec is None:
            raise ValueError("ensemble system needs dyn_params and dyn_spec")
        d.dynamics = dyn_spec.desc(dyn_params)
    d.x_dim, d.u_dim, d.n_envs, d.n_steps = x_dim, u_dim, n_envs, n_steps
    d.episode_length, d.action_repeat = episode_length, action_repeat
    d.system_kind, d.ens_mode = system_kind, ens_mode
    d.ens_predict_delta, d.ens_sample_noise, d.ens_min_std = int(ens_predict_delta), int(ens_sample_noise), ens_min_std
    d.reward_kind = reward_kind
    d.reward_params = reward_params.data_ptr()
    d.sys_params = ptr(_req(sys_params, "sys_params")) if sys_params is not None else None
    d.norm_mean = ptr(_req(norm_mean, "norm_mean")) if norm_mean is not None else None
    d.norm_std = ptr(_req(norm_std, "norm_std")) if norm_std is not None else None
    d.deterministic, d.ppo_extras, d.env_major = int(deterministic), int(ppo_extras), int(env_major)
    if policy_noise is not None:
        _req(policy_noise, "policy_noise")
        if policy_noise.numel() != n_steps * n_envs * u_dim:
            raise ValueError("policy_noise must be [S,N,u]")
    if model_noise is not None:
        _req(model_noise, "model_noise")
        if model_noise.numel() != n_steps * action_repeat * n_envs * x_dim:
            raise ValueError("model_noise must be [S,AR,N,x]")
    if member_idx is not None:
        _req(member_idx, "member_idx", torch.int32)
        if member_idx.numel() != n_steps * action_repeat * n_envs:
            raise ValueError("member_idx must be [S,AR,N]")
    d.policy_noise, d.model_noise, d.member_idx = ptr(policy_noise), ptr(model_noise), ptr(member_idx)
    d.seed, d.offset = seed, offset
    d.obs, d.first_obs, d.steps, d.done = obs.data_ptr(), first_obs.data_ptr(), steps.data_ptr(), done.data_ptr()
    d.transitions, d.row_len = out.data_ptr(), D
    check(lib.mbpo_model_rollout(C.byref(d), current_stream_ptr()), "mbpo_model_rollout")
    return out

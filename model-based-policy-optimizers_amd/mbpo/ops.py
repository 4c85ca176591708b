"""Tensor-level wrappers over the C-ABI (one function per entry point of include/mbpo_hip.h).

These are the only callers of libmbpo_hip.so; everything in mbpo.systems / mbpo.optimizers goes through them.
Inputs must be contiguous CUDA(HIP) tensors — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
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


def _uniform_hidden(spec: MlpSpec) -> Optional[int]:
    hid = [int(v) for v in spec.dims[1:-1]]
    return hid[0] if hid and all(h == hid[0] for h in hid) else None


def mlp_layered(params: torch.Tensor, spec: MlpSpec, xn: torch.Tensor, dy: Optional[torch.Tensor] = None, want_dx: bool = False,
                want_dw: bool = False, want_y: bool = True):
    """mbpo_mlp_layered_vjp: forward (and vector-Jacobian products) of an MLP of ANY hidden sizes, one fp32-MFMA GEMM launch per Dense
    layer (csrc/layered.hip) — what ensemble_mlp_forward / mlp_vjp fall back to outside the fused kernels' shapes.  xn [n, in] is
    the (already normalised) input shared by the nets.  Returns (dx wrt xn | None, dw | None, y | None)."""
    lib = load()
    _req(xn, "x")
    n, din, dout = xn.shape[0], spec.dims[0], spec.dims[-1]
    d = spec.desc(params)
    need = int(lib.mbpo_mlp_layered_workspace_floats(C.byref(d), n))
    if need < 0:
        check(need, "mbpo_mlp_layered_workspace_floats")
    ws = torch.empty(max(need, 1), device=xn.device, dtype=torch.float32)
    y = torch.empty((spec.n_nets, n, dout), device=xn.device, dtype=torch.float32) if want_y else None
    dx = torch.empty((spec.n_nets, n, din), device=xn.device, dtype=torch.float32) if (dy is not None and want_dx) else None
    dw = torch.empty(spec.n_nets * spec.n_params, device=xn.device, dtype=torch.float32) if (dy is not None and want_dw) else None
    check(lib.mbpo_mlp_layered_vjp(C.byref(d), xn.data_ptr(), n, ptr(dy), ptr(y), ptr(dx), ptr(dw), ws.data_ptr(), current_stream_ptr()),
          "mbpo_mlp_layered_vjp")
    return dx, dw, y


def ensemble_mlp_forward(params: torch.Tensor, spec: MlpSpec, x: torch.Tensor, shared_input: bool = True) -> torch.Tensor:
    """y[e, n, :] = MLP_e(x[n]) — R2 of SURVEY §8a.  Hidden layers of one width in {64, 128, 256}: the fused wave-chain kernel;
    any other sizes (shared input): layer by layer (mlp_layered)."""
    lib = load()
    _req(x, "x")
    d = spec.desc(params)
    din, dout = spec.dims[0], spec.dims[-1]
    if shared_input:
        if x.dim() != 2 or x.shape[1] != din:
            raise ValueError(f"x must be [N,{din}], got {tuple(x.shape)}")
        n = x.shape[0]
        if _uniform_hidden(spec) not in ROLLOUT_WIDTHS and len(spec.dims) > 2:
            return mlp_layered(params, spec, x.contiguous())[2]
    else:
        if x.dim() != 3 or x.shape[0] != spec.n_nets or x.shape[2] != din:
            raise ValueError(f"x must be [{spec.n_nets},N,{din}], got {tuple(x.shape)}")
        n = x.shape[1]
    y = torch.empty((spec.n_nets, n, dout), device=x.device, dtype=torch.float32)
    check(lib.mbpo_ensemble_mlp_forward(C.byref(d), x.data_ptr(), int(shared_input), y.data_ptr(), n,
                                        current_stream_ptr()), "mbpo_ensemble_mlp_forward")
    return y


def mlp_vjp(params: torch.Tensor, spec: MlpSpec, x: torch.Tensor, dy: torch.Tensor, norm_mean=None, norm_std=None,
            want_dx: bool = True, want_dw: bool = True, want_y: bool = False, workspace: Optional[torch.Tensor] = None):
    """mbpo_mlp_vjp: for the spec's 1 or 2 nets on the shared input x [n, in] with upstream gradients dy [n_nets, n, out]:
    (dx [n_nets, n, in] | None, dw [n_nets * n_params] | None, y [n_nets, n, out] | None).  64-wide hidden layers with input /
    output widths <= 32 and one or two nets: the fused kernel; any other shape (or an ensemble of more nets): layer by layer
    (mbpo_mlp_layered_vjp)."""
    lib = load()
    _req(x, "x"); _req(dy, "dy")
    din, dout = spec.dims[0], spec.dims[-1]
    n = x.shape[0]
    if x.dim() != 2 or x.shape[1] != din:
        raise ValueError(f"x must be [n,{din}], got {tuple(x.shape)}")
    if tuple(dy.shape) != (spec.n_nets, n, dout):
        raise ValueError(f"dy must be [{spec.n_nets},{n},{dout}], got {tuple(dy.shape)}")
    if len(spec.dims) > 2 and (_uniform_hidden(spec) != 64 or din > 32 or dout > 32 or spec.n_nets > 2):
        xn = x if norm_mean is None else ((x - norm_mean) / norm_std)
        dx, dw, y = mlp_layered(params, spec, xn.contiguous(), dy.contiguous(), want_dx=want_dx, want_dw=want_dw, want_y=want_y)
        if dx is not None and norm_mean is not None:
            dx = dx / norm_std               # d/dx of (x - mean) / std
        return dx, dw, y
    d = spec.desc(params)
    dx = torch.empty((spec.n_nets, n, din), device=x.device, dtype=torch.float32) if want_dx else None
    dw = torch.empty(spec.n_nets * spec.n_params, device=x.device, dtype=torch.float32) if want_dw else None
    y = torch.empty((spec.n_nets, n, dout), device=x.device, dtype=torch.float32) if want_y else None
    if want_dw:
        need = int(lib.mbpo_mlp_vjp_workspace_floats(C.byref(d), n))
        if need < 0:
            check(need, "mbpo_mlp_vjp_workspace_floats")
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(max(need, 1), device=x.device, dtype=torch.float32)
    check(lib.mbpo_mlp_vjp(C.byref(d), x.data_ptr(), n, ptr(norm_mean), ptr(norm_std), dy.data_ptr(), ptr(y), ptr(dx), ptr(dw),
                           ptr(workspace) if want_dw else None, current_stream_ptr()), "mbpo_mlp_vjp")
    return dx, dw, y


class HipMlp(torch.autograd.Function):
    """y [n_nets, n, out] = MLP_k((x - mean) / std): forward in mbpo_ensemble_mlp_forward, backward in mbpo_mlp_vjp — the node that
    puts the HIP networks into a torch autograd graph whose other nodes are a USER-DEFINED System.step (BPTT through a model
    that exists only as the user's torch code; utils/optimizer_utils.py:62-116, bptt_optimizer.py:327-378).  `params` gets a
    gradient when it requires one, `x` likewise (mean / std are constants of the graph, as the reference's normaliser state is)."""

    @staticmethod
    def forward(ctx, params, x, spec, norm_mean, norm_std):
        xn = x if norm_mean is None else ((x - norm_mean) / norm_std).contiguous()
        ctx.spec, ctx.norm = spec, (norm_mean, norm_std)
        ctx.save_for_backward(params, x)
        return ensemble_mlp_forward(params, spec, xn.contiguous())

    @staticmethod
    def backward(ctx, dy):
        params, x = ctx.saved_tensors
        need_w, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_w or need_x):
            return None, None, None, None, None
        dx, dw, _ = mlp_vjp(params, ctx.spec, x.contiguous(), dy.contiguous(), ctx.norm[0], ctx.norm[1], want_dx=need_x, want_dw=need_w)
        gx = dx.sum(0) if need_x else None               # the nets share the input: their input gradients add up
        gw = None
        if need_w:
            gw = torch.zeros_like(params)
            gw[:dw.numel()] = dw
        return gw, gx, None, None, None


# ------------------------------------------------------------------------------------------------ hidden-width padding
# The MFMA kernels are built for hidden layers of ONE width per launch (64 or 128; the rollout also 256).  The reference accepts
# any sizes (experiments/train_inverted_pendulum/exp_ppo.py: policy (32,)*4 beside critic (256,)*5).  Narrower or unequal hidden
# layers are therefore ZERO-PADDED to a common supported width: a padded unit has zero incoming weights and bias, so its
# pre-activation is 0 and swish/relu/tanh(0) = 0 — it contributes nothing forward; its outgoing weights are zero, so its delta and
# every gradient that touches a padded weight is 0, and AdamW leaves an all-zero (p, g, m, v) exactly where it is (update
# 0 / (0 + eps) + wd * 0).  The padding is a fixed point of training: the padded network IS the logical one, bit for bit in
# exact arithmetic (the k-ordered fp32 sums gain only +0.0 terms).
KERNEL_WIDTHS = (64, 128)
# Above that the SAC update runs layer by layer (csrc/layered.hip: one GEMM launch per Dense layer, any hidden sizes, no padding);
# a POLICY also runs inside the rollout / act kernels, which take one hidden width in {64, 128, 256}.
ROLLOUT_WIDTHS = (64, 128, 256)


def common_width(*hidden_size_lists, supported=KERNEL_WIDTHS, what: str = "networks") -> int:
    """Smallest supported width >= every hidden size given (each argument is a sequence of hidden sizes, possibly empty)."""
    m = max([int(h) for hs in hidden_size_lists for h in hs], default=supported[0])
    for w in supported:
        if m <= w:
            return w
    raise _hip.MbpoHipError(f"{what}: hidden width {m} exceeds what the MI355X kernels are built for (max {supported[-1]}; see "
                            "INTEGRATION.md, 'Network shapes')")


def padded_dims(dims: Sequence[int], width: Optional[int]) -> list:
    """`width` None: the network keeps its logical hidden sizes (the layered path takes any)."""
    d = [int(v) for v in dims]
    return d if width is None else [d[0]] + [width] * (len(d) - 2) + [d[-1]]


def embed_mlp_params(flat: torch.Tensor, dims: Sequence[int], width: int, n_nets: int = 1) -> torch.Tensor:
    """Flat params of `n_nets` MLPs with `dims` -> the same networks with every hidden layer zero-padded to `width`."""
    d, dp = [int(v) for v in dims], padded_dims(dims, width)
    if d == dp:
        return flat
    n, npad = sum(d[i] * d[i + 1] + d[i + 1] for i in range(len(d) - 1)), sum(dp[i] * dp[i + 1] + dp[i + 1] for i in range(len(d) - 1))
    out = torch.zeros(n_nets * npad, dtype=flat.dtype, device=flat.device)
    for e in range(n_nets):
        o, op = e * n, e * npad
        for i in range(len(d) - 1):
            w = flat[o:o + d[i] * d[i + 1]].reshape(d[i], d[i + 1])
            out[op:op + dp[i] * dp[i + 1]].reshape(dp[i], dp[i + 1])[:d[i], :d[i + 1]] = w
            o += d[i] * d[i + 1]; op += dp[i] * dp[i + 1]
            out[op:op + d[i + 1]] = flat[o:o + d[i + 1]]
            o += d[i + 1]; op += dp[i + 1]
    return out


def extract_mlp_params(flat_padded: torch.Tensor, dims: Sequence[int], width: int, n_nets: int = 1) -> torch.Tensor:
    """Inverse of embed_mlp_params: the logical networks' flat params out of the padded ones."""
    d, dp = [int(v) for v in dims], padded_dims(dims, width)
    if d == dp:
        return flat_padded
    n, npad = sum(d[i] * d[i + 1] + d[i + 1] for i in range(len(d) - 1)), sum(dp[i] * dp[i + 1] + dp[i + 1] for i in range(len(d) - 1))
    out = torch.zeros(n_nets * n, dtype=flat_padded.dtype, device=flat_padded.device)
    for e in range(n_nets):
        o, op = e * n, e * npad
        for i in range(len(d) - 1):
            out[o:o + d[i] * d[i + 1]] = flat_padded[op:op + dp[i] * dp[i + 1]].reshape(dp[i], dp[i + 1])[:d[i], :d[i + 1]].reshape(-1)
            o += d[i] * d[i + 1]; op += dp[i] * dp[i + 1]
            out[o:o + d[i + 1]] = flat_padded[op:op + d[i + 1]]
            o += d[i + 1]; op += dp[i + 1]
    return out


def make_rng(device, seed: int = 0, counter: int = 0) -> torch.Tensor:
    """Device RNG control words (include/mbpo_hip.h "randomness"): int64[2] holding the uint64 pair {seed word, step counter}."""
    t = torch.zeros(2, dtype=torch.int64, device=device)
    set_rng(t, seed, counter)
    return t


def _as_i64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def set_rng(rng: torch.Tensor, seed: Optional[int] = None, counter: Optional[int] = None) -> None:
    """Write the seed word and/or the counter (host -> device copy; outside any captured graph)."""
    if seed is not None and counter is not None:
        rng.copy_(torch.tensor([_as_i64(seed), _as_i64(counter)], dtype=torch.int64))
    elif seed is not None:
        rng[0:1].copy_(torch.tensor([_as_i64(seed)], dtype=torch.int64))
    elif counter is not None:
        rng[1:2].copy_(torch.tensor([_as_i64(counter)], dtype=torch.int64))


def rng_advance(rng: torch.Tensor, inc: int = 1) -> None:
    """rng[1] += inc on the device (graph-capturable)."""
    check(load().mbpo_rng_advance(rng_ptr(rng), int(inc), current_stream_ptr()), "mbpo_rng_advance")


def rng_ptr(rng: Optional[torch.Tensor]) -> Optional[int]:
    if rng is None:
        return None
    _req(rng, "rng_dev", torch.int64)
    if rng.numel() != 2:
        raise ValueError("rng_dev must be an int64[2] device tensor {seed word, step counter}")
    return rng.data_ptr()


def transition_row_len(x_dim: int, u_dim: int, ppo_extras: bool = False) -> int:
    return 2 * x_dim + u_dim + 3 + ((1 + u_dim) if ppo_extras else 0)


def model_rollout(*, policy_params: Optional[torch.Tensor] = None, policy_spec: Optional[MlpSpec] = None, x_dim: int, u_dim: int,
                  actions: Optional[torch.Tensor] = None,
                  obs: torch.Tensor, first_obs: torch.Tensor, steps: torch.Tensor, done: torch.Tensor,
                  n_steps: int, episode_length: int, action_repeat: int = 1,
                  system_kind: int = _hip.SYS_PENDULUM, dyn_params: Optional[torch.Tensor] = None,
                  dyn_spec: Optional[MlpSpec] = None, ens_mode: int = _hip.ENS_MEAN, ens_predict_delta: bool = True,
                  ens_sample_noise: bool = False, ens_min_std: float = 1e-3,
                  reward_kind: int = _hip.REWARD_PENDULUM, reward_params: torch.Tensor = None,
                  sys_params: Optional[torch.Tensor] = None,
                  norm_mean: Optional[torch.Tensor] = None, norm_std: Optional[torch.Tensor] = None,
                  deterministic: bool = False, ppo_extras: bool = False, env_major: bool = False, action_clip: float = 0.0,
                  policy_noise: Optional[torch.Tensor] = None, model_noise: Optional[torch.Tensor] = None,
                  member_idx: Optional[torch.Tensor] = None, seed: int = 0, offset: int = 0,
                  rng_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  system=None, system_params=None, system_params_out: Optional[list] = None) -> torch.Tensor:
    """Fused S-step model rollout for N envs (R1-R8).  Updates obs/steps/done in place; returns rows [S*N, D].
    system_kind == SYS_GENERIC (a user-defined `system`): the same contract through generic_rollout below."""
    if system_kind == _hip.SYS_GENERIC:
        if model_noise is not None or member_idx is not None:
            raise ValueError("model_noise / member_idx belong to the fused ensemble; a user-defined System draws its own randomness")
        return generic_rollout(system=system, system_params=system_params, system_params_out=system_params_out,
                               policy_params=policy_params, policy_spec=policy_spec, x_dim=x_dim, u_dim=u_dim, actions=actions,
                               obs=obs, first_obs=first_obs, steps=steps, done=done, n_steps=n_steps,
                               episode_length=episode_length, action_repeat=action_repeat, norm_mean=norm_mean, norm_std=norm_std,
                               deterministic=deterministic, ppo_extras=ppo_extras, env_major=env_major, action_clip=action_clip,
                               policy_noise=policy_noise, seed=seed, offset=offset, rng_dev=rng_dev, out=out)
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
    if actions is None:
        if policy_params is None or policy_spec is None:
            raise ValueError("need either a policy (policy_params, policy_spec) or open-loop actions")
        d.policy = policy_spec.desc(policy_params)
    else:
        _req(actions, "actions")
        if actions.numel() != n_steps * n_envs * u_dim:
            raise ValueError("actions must be [S,N,u]")
        d.actions = actions.data_ptr()
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
    d.action_clip = action_clip
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
    d.rng_dev = rng_ptr(rng_dev)
    d.obs, d.first_obs, d.steps, d.done = obs.data_ptr(), first_obs.data_ptr(), steps.data_ptr(), done.data_ptr()
    d.transitions, d.row_len = out.data_ptr(), D
    check(lib.mbpo_model_rollout(C.byref(d), current_stream_ptr()), "mbpo_model_rollout")
    return out


# ------------------------------------------------------------------------------------------------ user-defined System (non-fused)
def policy_act(policy_params: torch.Tensor, policy_spec: MlpSpec, obs: torch.Tensor, norm_mean=None, norm_std=None,
               deterministic: bool = False, action_clip: float = 0.0, noise: Optional[torch.Tensor] = None, seed: int = 0,
               offset: int = 0, rng_dev: Optional[torch.Tensor] = None, elem_base: int = 0, want_extras: bool = False,
               workspace: Optional[torch.Tensor] = None):
    """mbpo_policy_act: action [n,u] (and raw_action [n,u], log_prob [n] when want_extras) = policy(obs)."""
    lib = load()
    _req(obs, "obs")
    n, X = obs.shape
    U = policy_spec.dims[-1] // 2
    if X != policy_spec.dims[0]:
        raise ValueError(f"obs must be [n,{policy_spec.dims[0]}]")
    action = torch.empty(n, U, device=obs.device, dtype=torch.float32)
    raw = torch.empty(n, U, device=obs.device, dtype=torch.float32) if want_extras else None
    lp = torch.empty(n, device=obs.device, dtype=torch.float32) if want_extras else None
    need = n * (X + 2 * U)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), device=obs.device, dtype=torch.float32)
    if noise is not None:
        _req(noise, "noise")
        if noise.numel() != n * U:
            raise ValueError("noise must be [n,u]")
    d = policy_spec.desc(policy_params)
    check(lib.mbpo_policy_act(C.byref(d), obs.data_ptr(), n, ptr(norm_mean), ptr(norm_std), int(deterministic), float(action_clip),
                              ptr(noise), seed, offset, rng_ptr(rng_dev), int(elem_base), action.data_ptr(), ptr(raw), ptr(lp),
                              workspace.data_ptr(), current_stream_ptr()), "mbpo_policy_act")
    return (action, raw, lp) if want_extras else action


def episode_step(*, x_dim: int, u_dim: int, episode_length: int, action_repeat: int, ppo_extras: bool, env_major: bool,
                 step_index: int, n_steps: int, action, reward, x_next, first_obs, obs, steps, done, rows, raw_action=None,
                 log_prob=None, sys_done=None) -> None:
    """mbpo_episode_step: Episode/AutoReset bookkeeping + the Transition row of env step `step_index`."""
    lib = load()
    for t, nm in ((action, "action"), (reward, "reward"), (x_next, "x_next"), (first_obs, "first_obs"), (obs, "obs"), (steps, "steps"),
                  (done, "done"), (rows, "rows")):
        _req(t, nm)
    n = obs.shape[0]
    if x_next.shape != (n, x_dim) or action.shape != (n, u_dim) or reward.shape != (n,):
        raise ValueError(f"System.step must return x_next [{n},{x_dim}] and reward [{n}] for actions [{n},{u_dim}]; got "
                         f"{tuple(x_next.shape)}, {tuple(reward.shape)}, {tuple(action.shape)}")
    d = _hip.EpisodeStepDesc()
    d.x_dim, d.u_dim, d.n_envs, d.episode_length, d.action_repeat = x_dim, u_dim, n, episode_length, action_repeat
    d.ppo_extras, d.env_major, d.step_index, d.n_steps = int(ppo_extras), int(env_major), step_index, n_steps
    d.action, d.raw_action, d.log_prob = action.data_ptr(), ptr(raw_action), ptr(log_prob)
    d.reward, d.x_next, d.sys_done, d.first_obs = reward.data_ptr(), x_next.data_ptr(), ptr(sys_done), first_obs.data_ptr()
    d.obs, d.steps, d.done, d.transitions, d.row_len = obs.data_ptr(), steps.data_ptr(), done.data_ptr(), rows.data_ptr(), rows.shape[1]
    check(lib.mbpo_episode_step(C.byref(d), current_stream_ptr()), "mbpo_episode_step")


def generic_rollout(*, system, system_params, system_params_out: Optional[list] = None, policy_params=None, policy_spec=None,
                    x_dim: int, u_dim: int, actions=None, obs, first_obs, steps, done, n_steps: int, episode_length: int,
                    action_repeat: int = 1, norm_mean=None, norm_std=None, deterministic: bool = False, ppo_extras: bool = False,
                    env_major: bool = False, action_clip: float = 0.0, policy_noise=None, seed: int = 0, offset: int = 0,
                    rng_dev=None, out=None) -> torch.Tensor:
    """The contract of model_rollout for a USER-DEFINED System (the reference's plug-in seam, base_systems.py:40-52): per env step
    mbpo_policy_act (HIP) -> system.step(obs [N,x], action [N,u], system_params) x action_repeat (the user's batched torch code on
    the device; rewards summed, brax_utils/training.py:92-97) -> mbpo_episode_step (HIP).  Same Philox stream as the fused kernel.
    The final system_params are appended to `system_params_out` (the reference carries them in the env State)."""
    if system is None or system_params is None:
        raise ValueError("a user-defined System needs `system` and `system_params`")
    for t, nm in ((obs, "obs"), (first_obs, "first_obs"), (steps, "steps"), (done, "done")):
        _req(t, nm)
    N = obs.shape[0]
    D = transition_row_len(x_dim, u_dim, ppo_extras)
    if out is None:
        out = torch.empty((n_steps * N, D), device=obs.device, dtype=torch.float32)
    elif out.shape != (n_steps * N, D):
        raise ValueError(f"out must be [{n_steps * N},{D}]")
    if actions is None and (policy_params is None or policy_spec is None):
        raise ValueError("need either a policy (policy_params, policy_spec) or open-loop actions")
    if actions is not None:
        _req(actions, "actions")
        if ppo_extras:
            raise ValueError("ppo_extras needs a policy")
        actions = actions.reshape(n_steps, N, u_dim)
    if policy_noise is not None:
        policy_noise = policy_noise.reshape(n_steps, N, u_dim)
    ws = torch.empty(max(N * (x_dim + 2 * u_dim), 1), device=obs.device, dtype=torch.float32)
    sp = system_params
    for s in range(n_steps):
        raw = lp = None
        if actions is not None:
            act = actions[s].contiguous()
        else:
            res = policy_act(policy_params, policy_spec, obs, norm_mean, norm_std, deterministic, action_clip,
                             None if policy_noise is None else policy_noise[s].contiguous(), seed, offset, rng_dev,
                             elem_base=s * N * u_dim, want_extras=ppo_extras, workspace=ws)
            act, raw, lp = res if ppo_extras else (res, None, None)
        x, reward, sys_done = obs, None, None
        for _ in range(action_repeat):                              # EpisodeWrapper.step :92-97
            st = system.step(x, act, sp)
            x, sp = st.x_next, st.system_params
            r = torch.as_tensor(st.reward, device=obs.device, dtype=torch.float32).reshape(-1).expand(N)
            reward = r if reward is None else reward + r
            sys_done = st.done
        sys_done = None if isinstance(sys_done, (int, float)) and sys_done == 0 else \
            torch.as_tensor(sys_done, device=obs.device, dtype=torch.float32).reshape(-1).expand(N).contiguous()
        episode_step(x_dim=x_dim, u_dim=u_dim, episode_length=episode_length, action_repeat=action_repeat, ppo_extras=ppo_extras,
                     env_major=env_major, step_index=s, n_steps=n_steps, action=act, reward=reward.contiguous(),
                     x_next=x.to(torch.float32).contiguous(), first_obs=first_obs, obs=obs, steps=steps, done=done, rows=out,
                     raw_action=raw, log_prob=lp, sys_done=sys_done)
    if system_params_out is not None:
        system_params_out.append(sp)
    return out


# ------------------------------------------------------------------------------------------------ replay (R9)
def replay_insert(data: torch.Tensor, state: torch.Tensor, rows: torch.Tensor) -> None:
    """UniformSamplingQueue.insert on a ring-stored buffer.  data [max,D]; state int32[4]; rows [n,D]."""
    lib = load()
    _req(data, "data"); _req(state, "state", torch.int32); _req(rows, "rows")
    if rows.dim() != 2 or rows.shape[1] != data.shape[1]:
        raise ValueError(f"rows must be [n,{data.shape[1]}], got {tuple(rows.shape)}")
    check(lib.mbpo_replay_insert(data.data_ptr(), data.shape[0], data.shape[1], state.data_ptr(), rows.data_ptr(),
                                 rows.shape[0], current_stream_ptr()), "mbpo_replay_insert")


def replay_gather(data: torch.Tensor, state: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """jnp.take(data_logical, idx, axis=0, mode='wrap')."""
    lib = load()
    _req(data, "data"); _req(state, "state", torch.int32); _req(idx, "idx", torch.int32)
    if out is None:
        out = torch.empty((idx.numel(), data.shape[1]), device=data.device, dtype=torch.float32)
    else:
        _req(out, "out")
        if out.numel() != idx.numel() * data.shape[1]:
            raise ValueError(f"out must hold {idx.numel()} rows of {data.shape[1]} floats")
    check(lib.mbpo_replay_gather(data.data_ptr(), data.shape[0], data.shape[1], state.data_ptr(), idx.data_ptr(),
                                 idx.numel(), out.data_ptr(), current_stream_ptr()), "mbpo_replay_gather")
    return out


def replay_sample(data: torch.Tensor, state: torch.Tensor, n: int, seed: int, offset: int, return_idx: bool = False,
                  out: Optional[torch.Tensor] = None, rng_dev: Optional[torch.Tensor] = None,
                  idx_out: Optional[torch.Tensor] = None):
    """UniformSamplingQueue.sample: Philox randint in [sample_position, insert_position) + gather, one launch."""
    lib = load()
    _req(data, "data"); _req(state, "state", torch.int32)
    if out is None:
        out = torch.empty((n, data.shape[1]), device=data.device, dtype=torch.float32)
    idx = torch.empty((n,), device=data.device, dtype=torch.int32) if return_idx else None
    if idx_out is not None:
        _req(idx_out, "idx_out", torch.int32)
        if idx_out.numel() != n:
            raise ValueError(f"idx_out must have {n} entries")
        idx = idx_out
    check(lib.mbpo_replay_sample(data.data_ptr(), data.shape[0], data.shape[1], state.data_ptr(), seed, offset,
                                 rng_ptr(rng_dev), n, ptr(idx), out.data_ptr(), current_stream_ptr()), "mbpo_replay_sample")
    return (out, idx) if return_idx else out


def philox_permutation(n: int, seed: int, offset: int = 0, rng_dev: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Random permutation of range(n) as int32 (PPO.sgd_step's shared shuffle, ppo/ppo.py:166-171): stable argsort of Philox keys."""
    lib = load()
    dev = torch.device("cuda", torch.cuda.current_device())
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=dev)
    else:
        _req(out, "out", torch.int32)
    if workspace is None:
        workspace = torch.empty(n, dtype=torch.int32, device=out.device)
    else:
        _req(workspace, "workspace", torch.int32)
    if out.numel() != n or workspace.numel() < n:
        raise ValueError("out / workspace must hold n int32")
    check(lib.mbpo_philox_permutation(seed, offset, rng_ptr(rng_dev), n, out.data_ptr(), workspace.data_ptr(), current_stream_ptr()),
          "mbpo_philox_permutation")
    return out


# ------------------------------------------------------------------------------------------------ running statistics (R8)
def stats_workspace_floats(x_dim: int) -> int:
    """Floats mbpo_running_stats_reduce needs in `workspace` (one partial per workgroup and column)."""
    n = load().mbpo_running_stats_workspace_floats(int(x_dim))
    if n < 0:
        check(int(n), "mbpo_running_stats_workspace_floats")
    return int(n)


def running_stats_reduce(rows: torch.Tensor, col_off: int, x_dim: int, stats: torch.Tensor, pass_: int,
                         sums: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One pass of running_statistics.update's reductions (pass 0: n, sum d; pass 1: sum d*(d-upd))."""
    lib = load()
    _req(rows, "rows"); _req(stats, "stats")
    if sums is None:
        sums = torch.zeros(1 + 2 * x_dim, device=rows.device, dtype=torch.float32)
    if workspace is None:
        workspace = torch.empty(stats_workspace_floats(x_dim), device=rows.device, dtype=torch.float32)
    elif workspace.numel() < stats_workspace_floats(x_dim):
        raise ValueError(f"workspace must hold {stats_workspace_floats(x_dim)} floats, got {workspace.numel()}")
    check(lib.mbpo_running_stats_reduce(rows.data_ptr(), rows.shape[0], rows.shape[1], col_off, x_dim, stats.data_ptr(),
                                        sums.data_ptr(), workspace.data_ptr(), pass_, current_stream_ptr()),
          "mbpo_running_stats_reduce")
    return sums


def running_stats_update(rows: torch.Tensor, col_off: int, x_dim: int, stats: torch.Tensor, all_reduce=None,
                         sums: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                         std_min: float = 1e-6, std_max: float = 1e6) -> None:
    """running_statistics.update(state, rows[:, col_off:col_off+x_dim]); `all_reduce(t)` sums `t` over ranks in place
    (the reference's psum under pmap_axis_name, sac/sac.py:298-301)."""
    if all_reduce is None:
        # a single rank: nothing sits between the passes — three launches instead of five, the same bits
        lib = load()
        _req(rows, "rows"); _req(stats, "stats")
        if rows.dim() != 2 or stats.numel() != 1 + 3 * x_dim:
            raise ValueError("rows must be [n, row_len] and stats [1 + 3*x_dim]")
        if sums is None:
            sums = torch.zeros(1 + 2 * x_dim, device=rows.device, dtype=torch.float32)
        if workspace is None:
            workspace = torch.empty(stats_workspace_floats(x_dim), device=rows.device, dtype=torch.float32)
        _req(sums, "sums"); _req(workspace, "workspace")
        if workspace.numel() < stats_workspace_floats(x_dim):
            raise ValueError(f"workspace must hold {stats_workspace_floats(x_dim)} floats, got {workspace.numel()}")
        check(lib.mbpo_running_stats_update(rows.data_ptr(), rows.shape[0], rows.shape[1], col_off, x_dim, stats.data_ptr(),
                                            sums.data_ptr(), workspace.data_ptr(), std_min, std_max, current_stream_ptr()),
              "mbpo_running_stats_update")
        return
    sums = running_stats_reduce(rows, col_off, x_dim, stats, 0, sums=sums, workspace=workspace)
    if all_reduce is not None:
        all_reduce(sums)
    running_stats_reduce(rows, col_off, x_dim, stats, 1, sums=sums, workspace=workspace)
    if all_reduce is not None:
        all_reduce(sums[1 + x_dim:])
    running_stats_apply(stats, sums, x_dim, std_min, std_max)


def running_stats_apply(stats: torch.Tensor, sums: torch.Tensor, x_dim: int, std_min: float = 1e-6, std_max: float = 1e6) -> None:
    lib = load()
    _req(stats, "stats"); _req(sums, "sums")
    check(lib.mbpo_running_stats_apply(stats.data_ptr(), sums.data_ptr(), x_dim, std_min, std_max, current_stream_ptr()),
          "mbpo_running_stats_apply")


# ------------------------------------------------------------------------------------------------ scans (P4, B2)
def gae_scan(truncation, termination, rewards, values, bootstrap, gamma, lam: float, time_major: bool = False):
    """compute_gae (ppo/losses.py:128-184).  Arrays [B,T] (time_major=False) or [T,B]; bootstrap [B].
    `gamma` may be a tensor shaped like `rewards`: the per-step discount of non_equidistant_time (ppo/losses_new.py:181-226)."""
    lib = load()
    for t, nm in ((truncation, "truncation"), (termination, "termination"), (rewards, "rewards"), (values, "values"),
                  (bootstrap, "bootstrap")):
        _req(t, nm)
    B, T = (values.shape[1], values.shape[0]) if time_major else (values.shape[0], values.shape[1])
    if bootstrap.shape != (B,):
        raise ValueError("bootstrap must be [B]")
    vs = torch.empty_like(values)
    adv = torch.empty_like(values)
    if isinstance(gamma, torch.Tensor):
        _req(gamma, "gamma")
        if gamma.shape != values.shape:
            raise ValueError("a per-element discount must have the shape of rewards/values")
        check(lib.mbpo_gae_scan_discounts(truncation.data_ptr(), termination.data_ptr(), rewards.data_ptr(), values.data_ptr(),
                                          bootstrap.data_ptr(), gamma.data_ptr(), vs.data_ptr(), adv.data_ptr(), B, T, lam,
                                          int(time_major), current_stream_ptr()), "mbpo_gae_scan_discounts")
        return vs, adv
    check(lib.mbpo_gae_scan(truncation.data_ptr(), termination.data_ptr(), rewards.data_ptr(), values.data_ptr(),
                            bootstrap.data_ptr(), vs.data_ptr(), adv.data_ptr(), B, T, gamma, lam, int(time_major),
                            current_stream_ptr()), "mbpo_gae_scan")
    return vs, adv


def lambda_return_scan(rewards, next_values, gamma: float, lam: float, time_major: bool = False):
    """lambda_return (utils/optimizer_utils.py:119-152)."""
    lib = load()
    _req(rewards, "rewards"); _req(next_values, "next_values")
    B, T = (rewards.shape[1], rewards.shape[0]) if time_major else (rewards.shape[0], rewards.shape[1])
    out = torch.empty_like(rewards)
    check(lib.mbpo_lambda_return_scan(rewards.data_ptr(), next_values.data_ptr(), out.data_ptr(), B, T, gamma, lam,
                                      int(time_major), current_stream_ptr()), "mbpo_lambda_return_scan")
    return out


# ------------------------------------------------------------------------------------------------ SAC sgd_step (S3-S8)
class SacUpdater:
    """Owns the flat SAC train state on one GPU and drives mbpo_sac_grads / _grad_norms / _apply.

    Flat layout (include/mbpo_hip.h): params = [policy | critic0 | critic1 | log_alpha]; target_q; adam m/v; step count.
    `all_reduce(t)`, when given, must SUM `t` over ranks in place (torch.distributed.all_reduce) — it is called on the
    flat gradient between grads and apply, the position of the reference's jax.lax.pmean (sac/utils.py:29-33).
    """

    def __init__(self, *, x_dim: int, u_dim: int, policy_dims: Sequence[int], q_dims: Sequence[int], batch_size: int,
                 device, policy_activation: str = "swish", q_activation: str = "swish", discounting: float = 0.9,
                 reward_scaling: float = 1.0, target_entropy: Optional[float] = None, tau: float = 0.005,
                 lr_policy: float = 1e-4, lr_q: float = 1e-4, lr_alpha: float = 1e-4, wd_policy: float = 0.0,
                 wd_q: float = 0.0, wd_alpha: float = 0.0, max_grad_norm: float = 1e5, seed: int = 0,
                 all_reduce=None, world_size: int = 1, two_launch: Optional[bool] = None, p2p=None,
                 non_equidistant_time: bool = False, continuous_discounting: float = 0.0, min_time_between_switches: float = 0.0,
                 max_time_between_switches: float = 0.0, env_dt: float = 0.0):
        self.lib = load()
        self.x_dim, self.u_dim, self.batch_size = x_dim, u_dim, batch_size
        # two_launch (default on; MBPO_SAC_TWO_LAUNCH=0 or two_launch=False selects grads + apply): fwd/bwd, then ONE launch for slab
        # reduction [+ peer exchange] + optimizer step; the clip check is resolved by the next step's prologue or by finalize().
        # A library collective (`all_reduce` without p2p) sits BETWEEN reduction and optimizer step: that path keeps its launches.
        # (chosen explicitly — argument or environment — it stays; otherwise a trainer may switch between epochs on the clip rate)
        self.two_launch_explicit = two_launch is not None or "MBPO_SAC_TWO_LAUNCH" in os.environ
        if two_launch is None:
            two_launch = os.environ.get("MBPO_SAC_TWO_LAUNCH", "1") != "0"
        self.two_launch = bool(two_launch)
        self.policy_spec = MlpSpec(list(policy_dims), policy_activation, 1)
        self.q_spec = MlpSpec(list(q_dims), q_activation, 2)
        self.P, self.Q = self.policy_spec.n_params, self.q_spec.n_params
        self.NP = self.P + 2 * self.Q + 1
        self.device = torch.device(device)
        f = lambda n: torch.zeros(n, device=self.device, dtype=torch.float32)
        self.params, self.target_q, self.adam_m, self.adam_v = f(self.NP), f(2 * self.Q), f(self.NP), f(self.NP)
        self.step_count, self.grads, self.metrics, self.metrics_accum = f(1), f(self.NP), f(4), f(5)
        self.all_reduce, self.world_size = all_reduce, world_size
        self.p2p = p2p      # mbpo.parallel.P2PExchange: gradient exchange through peer memory instead of `all_reduce`
        self.p2p_fused = os.environ.get("MBPO_P2P_FUSED", "1") != "0"   # exchange inside the reduction kernel (default) or split
        # the fused forms wait INSIDE the reduction kernel for the peers' same kernel, so all of its ceil(NP / 256) workgroups must be
        # co-resident: the library refuses more than 1024 (mbpo_sac_step_p2p / mbpo_sac_grads_exchange_p2p).  Wide layered critics
        # (the reference's 256 x 5: NP ~ 530 k, 2070 workgroups) take the split form — push, gather, apply — instead.
        self._p2p_fused_fits = (self.NP + 255) // 256 <= 1024
        d = _hip.SacDesc()
        d.x_dim, d.u_dim = x_dim, u_dim
        d.policy_layers, d.q_layers = len(policy_dims) - 1, len(q_dims) - 1
        for i, v in enumerate(policy_dims):
            d.policy_dims[i] = int(v)
        for i, v in enumerate(q_dims):
            d.q_dims[i] = int(v)
        d.policy_activation, d.q_activation = _hip.ACT_IDS[policy_activation], _hip.ACT_IDS[q_activation]
        d.batch_size, d.row_len = batch_size, transition_row_len(x_dim, u_dim)
        d.discounting, d.reward_scaling, d.tau = discounting, reward_scaling, tau
        d.target_entropy = -0.5 * u_dim if target_entropy is None else target_entropy      # losses.py:49-50
        d.lr_policy, d.lr_q, d.lr_alpha = lr_policy, lr_q, lr_alpha
        d.wd_policy, d.wd_q, d.wd_alpha, d.max_grad_norm = wd_policy, wd_q, wd_alpha, max_grad_norm
        d.grad_scale = 1.0 / world_size
        d.non_equidistant_time = int(non_equidistant_time)          # N1 (sac/losses.py:90-98)
        d.continuous_discounting, d.env_dt = continuous_discounting, env_dt
        d.min_time_between_switches, d.max_time_between_switches = min_time_between_switches, max_time_between_switches
        d.seed, d.offset = seed, 0
        nws = self.lib.mbpo_sac_workspace_floats(C.byref(d))
        if nws < 0:
            check(int(nws), "mbpo_sac_workspace_floats")
        self.workspace = f(int(nws))
        d.params, d.target_q, d.adam_m, d.adam_v = (t.data_ptr() for t in (self.params, self.target_q, self.adam_m, self.adam_v))
        d.step_count, d.grads, d.workspace, d.metrics = (t.data_ptr() for t in (self.step_count, self.grads, self.workspace, self.metrics))
        d.metrics_accum = self.metrics_accum.data_ptr()
        self.desc = d
        off = int(self.lib.mbpo_sac_control_offset(C.byref(d)))
        if off < 0:
            check(off, "mbpo_sac_control_offset")
        self._control = self.workspace[off:off + 16].view(torch.int32)      # include/mbpo_hip.h: mbpo_sac_control_offset

    def clip_events(self) -> int:
        """Optimizer steps so far in which clip_by_global_norm scaled some group (device counter; this call synchronises)."""
        self.finalize()
        return int(self._control[13])

    def set_two_launch(self, flag: bool) -> None:
        """Choose the two-launch (speculative apply, clip resolved by the next launch) or the three-launch step.  Results are
        bit-identical; a step that clips costs the two-launch path a fix-up and a second pass (~95 vs ~36 us at B=256)."""
        self.finalize()
        self.two_launch = bool(flag)

    # views into the flat state
    @property
    def policy_params(self) -> torch.Tensor:
        return self.params[:self.P]

    @property
    def q_params(self) -> torch.Tensor:
        return self.params[self.P:self.P + 2 * self.Q]

    @property
    def log_alpha(self) -> torch.Tensor:
        return self.params[self.NP - 1:]

    def load_state(self, params, target_q=None, adam_m=None, adam_v=None, count: float = 0.0):
        self.finalize()          # a pending clip check belongs to the state that is being replaced
        self.params.copy_(params)
        self.target_q.copy_(self.params[self.P:self.P + 2 * self.Q] if target_q is None else target_q)
        self.adam_m.zero_() if adam_m is None else self.adam_m.copy_(adam_m)
        self.adam_v.zero_() if adam_v is None else self.adam_v.copy_(adam_v)
        self.step_count.fill_(count)

    def sgd_step(self, batch: torch.Tensor, norm_mean=None, norm_std=None, noise_alpha=None, noise_critic=None,
                 noise_actor=None, offset: int = 0, seed: Optional[int] = None, rng_dev: Optional[torch.Tensor] = None,
                 defer_clip_check: bool = False) -> None:
        """One SAC.sgd_step (sac/sac.py:227-281) on `batch` [B, 2x+u+3]; metrics land in self.metrics (device).
        Noise that is not given explicitly is Philox(seed, offset [+ rng_dev]): vary `offset` (or advance the device counter)
        between steps — the optimizer step count plays no part in the streams.
        defer_clip_check (two-launch path): leave the clip check of this step to the next sgd_step's first launch; the caller
        ends a run of steps with finalize() before reading params / target_q / moments / metrics[3]."""
        _req(batch, "batch")
        if batch.shape != (self.batch_size, self.desc.row_len):
            raise ValueError(f"batch must be [{self.batch_size},{self.desc.row_len}], got {tuple(batch.shape)}")
        d = self.desc
        d.batch = batch.data_ptr()
        d.norm_mean, d.norm_std = ptr(norm_mean), ptr(norm_std)
        for t, nm in ((noise_alpha, "noise_alpha"), (noise_critic, "noise_critic"), (noise_actor, "noise_actor")):
            if t is not None:
                _req(t, nm)
                if t.numel() != self.batch_size * self.u_dim:
                    raise ValueError(f"{nm} must be [B,u]")
        d.noise_alpha, d.noise_critic, d.noise_actor = ptr(noise_alpha), ptr(noise_critic), ptr(noise_actor)
        d.offset = offset
        if seed is not None:
            d.seed = seed
        d.rng_dev = rng_ptr(rng_dev)
        st = current_stream_ptr()
        if self.p2p is not None:
            p2p_fused = self.p2p_fused and self._p2p_fused_fits
            if self.two_launch and p2p_fused:
                check(self.lib.mbpo_sac_step_p2p(C.byref(d), C.byref(self.p2p.desc), st), "mbpo_sac_step_p2p")
                if not defer_clip_check:
                    self.finalize()
                return
            if p2p_fused:      # the slab reduction also exchanges: one launch less per sgd_step
                check(self.lib.mbpo_sac_grads_exchange_p2p(C.byref(d), C.byref(self.p2p.desc), st), "mbpo_sac_grads_exchange_p2p")
            else:
                check(self.lib.mbpo_sac_grads_p2p(C.byref(d), C.byref(self.p2p.desc), st), "mbpo_sac_grads_p2p")
                check(self.lib.mbpo_sac_gather_p2p(C.byref(d), C.byref(self.p2p.desc), st), "mbpo_sac_gather_p2p")
            check(self.lib.mbpo_sac_apply(C.byref(d), st), "mbpo_sac_apply")
            return
        if self.all_reduce is None and self.two_launch:
            check(self.lib.mbpo_sac_step(C.byref(d), st), "mbpo_sac_step")
            if not defer_clip_check:
                self.finalize()
            return
        check(self.lib.mbpo_sac_grads(C.byref(d), st), "mbpo_sac_grads")
        if self.all_reduce is not None:
            self.all_reduce(self.grads)
            check(self.lib.mbpo_sac_grad_norms(C.byref(d), st), "mbpo_sac_grad_norms")
        check(self.lib.mbpo_sac_apply(C.byref(d), st), "mbpo_sac_apply")

    def finalize(self, rng_dev: Optional[torch.Tensor] = None, rng_inc: int = 1) -> None:
        """Resolve the pending clip check of the last two-launch step (mbpo_sac_finalize; a no-op when nothing is pending).
        With `rng_dev` the same launch also advances the device RNG's step counter (the end of a training step)."""
        if rng_dev is not None:
            check(self.lib.mbpo_sac_finalize_advance(C.byref(self.desc), rng_ptr(rng_dev), int(rng_inc), current_stream_ptr()),
                  "mbpo_sac_finalize_advance")
        else:
            check(self.lib.mbpo_sac_finalize(C.byref(self.desc), current_stream_ptr()), "mbpo_sac_finalize")


# ------------------------------------------------------------------------------------------------ PPO minibatch update (P1-P3)
class PpoUpdater:
    """Owns the flat PPO train state ([policy | value] params, Adam moments, step count) on one GPU and drives
    mbpo_ppo_grads / mbpo_ppo_apply.  `all_reduce(t)` (SUM over ranks, in place) sits at ppo.py:149-154's pmean position."""

    def __init__(self, *, x_dim: int, u_dim: int, policy_dims: Sequence[int], value_dims: Sequence[int], batch_size: int,
                 unroll_length: int, device, policy_activation: str = "swish", value_activation: str = "swish",
                 entropy_cost: float = 1e-4, discounting: float = 0.9, reward_scaling: float = 1.0, gae_lambda: float = 0.95,
                 clipping_epsilon: float = 0.3, normalize_advantage: bool = True, lr: float = 1e-4, wd: float = 1e-5,
                 seed: int = 0, all_reduce=None, world_size: int = 1):
        self.lib = load()
        self.x_dim, self.u_dim, self.batch_size, self.unroll_length = x_dim, u_dim, batch_size, unroll_length
        self.policy_spec = MlpSpec(list(policy_dims), policy_activation, 1)
        self.value_spec = MlpSpec(list(value_dims), value_activation, 1)
        self.P, self.V = self.policy_spec.n_params, self.value_spec.n_params
        self.NPV = self.P + self.V
        self.device = torch.device(device)
        f = lambda n: torch.zeros(n, device=self.device, dtype=torch.float32)
        self.params, self.adam_m, self.adam_v, self.grads = f(self.NPV), f(self.NPV), f(self.NPV), f(self.NPV)
        self.step_count, self.metrics, self.metrics_accum = f(1), f(4), f(5)
        self.all_reduce, self.world_size = all_reduce, world_size
        self.fused_step = os.environ.get("MBPO_PPO_FUSED_STEP", "1") != "0"     # mbpo_ppo_step (single rank) vs grads + apply
        d = _hip.PpoDesc()
        d.x_dim, d.u_dim = x_dim, u_dim
        d.policy_layers, d.value_layers = len(policy_dims) - 1, len(value_dims) - 1
        for i, v in enumerate(policy_dims):
            d.policy_dims[i] = int(v)
        for i, v in enumerate(value_dims):
            d.value_dims[i] = int(v)
        d.policy_activation, d.value_activation = _hip.ACT_IDS[policy_activation], _hip.ACT_IDS[value_activation]
        d.batch_size, d.unroll_length, d.row_len = batch_size, unroll_length, transition_row_len(x_dim, u_dim, True)
        d.entropy_cost, d.discounting, d.reward_scaling = entropy_cost, discounting, reward_scaling
        d.gae_lambda, d.clipping_epsilon, d.normalize_advantage = gae_lambda, clipping_epsilon, int(normalize_advantage)
        d.lr, d.wd, d.grad_scale = lr, wd, 1.0 / world_size
        d.seed, d.offset = seed, 0
        nws = self.lib.mbpo_ppo_workspace_floats(C.byref(d))
        if nws < 0:
            check(int(nws), "mbpo_ppo_workspace_floats")
        self.workspace = f(int(nws))
        d.params, d.adam_m, d.adam_v, d.step_count, d.grads = (t.data_ptr() for t in (self.params, self.adam_m, self.adam_v, self.step_count, self.grads))
        d.workspace, d.metrics, d.metrics_accum = self.workspace.data_ptr(), self.metrics.data_ptr(), self.metrics_accum.data_ptr()
        self.desc = d

    @property
    def policy_params(self) -> torch.Tensor:
        return self.params[:self.P]

    @property
    def value_params(self) -> torch.Tensor:
        return self.params[self.P:]

    def load_state(self, params, adam_m=None, adam_v=None, count: float = 0.0):
        self.params.copy_(params)
        self.adam_m.zero_() if adam_m is None else self.adam_m.copy_(adam_m)
        self.adam_v.zero_() if adam_v is None else self.adam_v.copy_(adam_v)
        self.step_count.fill_(count)

    def minibatch_step(self, data: torch.Tensor, norm_mean=None, norm_std=None, entropy_noise=None, offset: int = 0,
                       seed: Optional[int] = None, rng_dev: Optional[torch.Tensor] = None) -> None:
        """One PPO.minibatch_step (ppo.py:142-156) on data [B, T, 2x+2u+4]."""
        _req(data, "data")
        if tuple(data.shape) != (self.batch_size, self.unroll_length, self.desc.row_len):
            raise ValueError(f"data must be [{self.batch_size},{self.unroll_length},{self.desc.row_len}], got {tuple(data.shape)}")
        d = self.desc
        d.data = data.data_ptr()
        d.norm_mean, d.norm_std = ptr(norm_mean), ptr(norm_std)
        if entropy_noise is not None:
            _req(entropy_noise, "entropy_noise")
            if entropy_noise.numel() != self.batch_size * self.unroll_length * self.u_dim:
                raise ValueError("entropy_noise must be [B,T,u]")
        d.entropy_noise = ptr(entropy_noise)
        d.offset = offset
        if seed is not None:
            d.seed = seed
        d.rng_dev = rng_ptr(rng_dev)
        st = current_stream_ptr()
        if self.all_reduce is None and self.fused_step:
            check(self.lib.mbpo_ppo_step(C.byref(d), st), "mbpo_ppo_step")      # the reduce launch applies AdamW: one launch less
            return
        check(self.lib.mbpo_ppo_grads(C.byref(d), st), "mbpo_ppo_grads")
        if self.all_reduce is not None:
            self.all_reduce(self.grads)
        check(self.lib.mbpo_ppo_apply(C.byref(d), st), "mbpo_ppo_apply")


# ------------------------------------------------------------------------------------------------ BPTT actor gradient (B1-B5)
class BpttActorGrad:
    """Drives mbpo_bptt_actor_grads: forward rollout through the model from `n` initial states, lambda-returns, and the
    backward sweep with respect to the actor parameters.  Buffers (transitions, lambda_values, grads, metrics) are owned here."""

    def __init__(self, *, x_dim: int, u_dim: int, horizon: int, actor_dims: Sequence[int], critic_dims: Sequence[int], n: int,
                 device, actor_activation: str = "swish", critic_activation: str = "swish", init_stddev: float = 1.0,
                 discount: float = 0.99, lambda_: float = 0.97, ent_coef: float = 0.005, seed: int = 0):
        self.lib = load()
        self.x_dim, self.u_dim, self.horizon, self.n = x_dim, u_dim, horizon, n
        self.device = torch.device(device)
        self.actor_spec = MlpSpec(list(actor_dims), actor_activation, 1)
        self.critic_spec = MlpSpec(list(critic_dims), critic_activation, 2)
        self.P, self.C = self.actor_spec.n_params, self.critic_spec.n_params
        f = lambda *s: torch.zeros(*s, device=self.device, dtype=torch.float32)
        self.row_len = 2 * x_dim + u_dim + 2
        self.transitions, self.lambda_values = f(n * horizon, self.row_len), f(n * horizon)
        self.grads, self.metrics = f(self.P), f(2)
        d = _hip.BpttDesc()
        d.x_dim, d.u_dim, d.horizon, d.n = x_dim, u_dim, horizon, n
        d.actor_layers, d.critic_layers = len(actor_dims) - 1, len(critic_dims) - 1
        for i, v in enumerate(actor_dims):
            d.actor_dims[i] = int(v)
        for i, v in enumerate(critic_dims):
            d.critic_dims[i] = int(v)
        d.actor_activation, d.critic_activation = _hip.ACT_IDS[actor_activation], _hip.ACT_IDS[critic_activation]
        d.init_stddev, d.discount, d.lambda_, d.ent_coef = init_stddev, discount, lambda_, ent_coef
        d.seed = seed
        self.desc = d
        self.workspace = None

    def __call__(self, *, actor_params, target_critic_params, init_states, state_mean, state_std, reward_mean_std,
                 system_kind: int, reward_kind: int, reward_params, sys_params=None, dyn_params=None, dyn_spec: Optional[MlpSpec] = None,
                 ens_predict_delta: bool = True, act_noise=None, offset: int = 0, rng_dev=None):
        d = self.desc
        for t, nm in ((actor_params, "actor_params"), (target_critic_params, "target_critic_params"), (init_states, "init_states"),
                      (state_mean, "state_mean"), (state_std, "state_std"), (reward_mean_std, "reward_mean_std"),
                      (reward_params, "reward_params")):
            _req(t, nm)
        if tuple(init_states.shape) != (self.n, self.x_dim):
            raise ValueError(f"init_states must be [{self.n},{self.x_dim}]")
        if actor_params.numel() != self.P or target_critic_params.numel() != 2 * self.C:
            raise ValueError("actor_params / target_critic_params have the wrong size")
        d.actor_params, d.target_critic_params = actor_params.data_ptr(), target_critic_params.data_ptr()
        d.system_kind, d.reward_kind, d.ens_predict_delta = system_kind, reward_kind, int(ens_predict_delta)
        if system_kind == _hip.SYS_ENSEMBLE:
            if dyn_params is None or dyn_spec is None:
                raise ValueError("ensemble system needs dyn_params and dyn_spec")
            d.dynamics = dyn_spec.desc(dyn_params)
        d.reward_params = reward_params.data_ptr()
        d.sys_params = ptr(_req(sys_params, "sys_params")) if sys_params is not None else None
        d.state_mean, d.state_std, d.reward_mean_std = state_mean.data_ptr(), state_std.data_ptr(), reward_mean_std.data_ptr()
        d.init_states = init_states.data_ptr()
        if act_noise is not None:
            _req(act_noise, "act_noise")
            if act_noise.numel() != self.n * self.horizon * self.u_dim:
                raise ValueError("act_noise must be [n,H,u]")
        d.act_noise = ptr(act_noise)
        d.offset = offset
        d.rng_dev = rng_ptr(rng_dev)
        d.transitions, d.lambda_values = self.transitions.data_ptr(), self.lambda_values.data_ptr()
        d.grads, d.metrics = self.grads.data_ptr(), self.metrics.data_ptr()
        if self.workspace is None:
            nws = self.lib.mbpo_bptt_workspace_floats(C.byref(d))
            if nws < 0:
                check(int(nws), "mbpo_bptt_workspace_floats")
            self.workspace = torch.zeros(int(nws), device=self.device, dtype=torch.float32)
        d.workspace = self.workspace.data_ptr()
        check(self.lib.mbpo_bptt_actor_grads(C.byref(d), current_stream_ptr()), "mbpo_bptt_actor_grads")
        return self.grads


def philox_normal(n: int, seed: int, offset: int = 0, stream: int = 1, rng_dev: Optional[torch.Tensor] = None, elem_base: int = 0,
                  device=None) -> torch.Tensor:
    """mbpo_philox_normal_fill: n standard normals of `stream` (1 = policy noise) — the numbers the fused kernels draw."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.empty(int(n), device=dev, dtype=torch.float32)
    check(load().mbpo_philox_normal_fill(int(seed) & ((1 << 64) - 1), int(offset) & ((1 << 64) - 1), rng_ptr(rng_dev), int(stream),
                                         int(elem_base), int(n), out.data_ptr(), current_stream_ptr()), "mbpo_philox_normal_fill")
    return out


class LambdaReturnFn(torch.autograd.Function):
    """lambda_return (utils/optimizer_utils.py:119-152) on [n, H] tensors as an autograd node: forward and backward are the HIP scan
    (mbpo_lambda_return_scan).  R_t = r_t + g(1-l) V_t + g l R_{t+1}, R_H = V_{H-1} is linear in (r, V): with
    G_t = sum_{s<=t} (g l)^{t-s} dL/dR_s  (the same recurrence run forward in time = the scan on the time-reversed upstream
    gradient with zero bootstrap):  dL/dr_t = G_t,  dL/dV_t = g(1-l) G_t  (t < H-1),  dL/dV_{H-1} = g G_{H-1}."""

    @staticmethod
    def forward(ctx, reward, next_values, discount, lambda_):
        ctx.cfg = (float(discount), float(lambda_))
        return lambda_return_scan(reward.contiguous(), next_values.contiguous(), discount, lambda_)

    @staticmethod
    def backward(ctx, g):
        discount, lambda_ = ctx.cfg
        G = lambda_return_scan(g.flip(1).contiguous(), torch.zeros_like(g), discount, lambda_).flip(1)
        dv = discount * (1.0 - lambda_) * G
        dv[:, -1] = discount * G[:, -1]
        return G, dv, None, None


class BpttActorGradGeneric:
    """BpttActorGrad's contract (grads [P], transitions [n*H, 2x+u+2], lambda_values [n*H], metrics) for a USER-DEFINED System —
    value_and_grad(vmap(actor_loss)) of bptt_optimizer.py:327-372 with rollout_policy (utils/optimizer_utils.py:62-116) walked on
    the host: per horizon step  HipMlp (actor on stop_gradient(obs), HIP) -> squash -> the user's batched, torch-differentiable
    System.step -> next step;  then HipMlp (twin target critics on the next states, HIP), LambdaReturnFn (HIP scans), the loss of
    :346-353, and one backward through the graph: torch autograd differentiates the user's step, mbpo_mlp_vjp the networks.
    Same Philox noise (stream POLICY_NOISE, element (traj*H + t)*u + d) and the same A > 1 log-prob form as the fused kernel.
    Not hipGraph-captured (user code runs between the kernels)."""

    def __init__(self, *, x_dim: int, u_dim: int, horizon: int, actor_dims: Sequence[int], critic_dims: Sequence[int], n: int,
                 device, actor_activation: str = "swish", critic_activation: str = "swish", init_stddev: float = 1.0,
                 discount: float = 0.99, lambda_: float = 0.97, ent_coef: float = 0.005, seed: int = 0):
        self.x_dim, self.u_dim, self.horizon, self.n = x_dim, u_dim, horizon, n
        self.device = torch.device(device)
        self.actor_spec = MlpSpec(list(actor_dims), actor_activation, 1)
        self.critic_spec = MlpSpec(list(critic_dims), critic_activation, 2)
        self.P, self.C = self.actor_spec.n_params, self.critic_spec.n_params
        self.init_stddev, self.discount, self.lambda_, self.ent_coef = float(init_stddev), float(discount), float(lambda_), float(ent_coef)
        f = lambda *s: torch.zeros(*s, device=self.device, dtype=torch.float32)
        self.row_len = 2 * x_dim + u_dim + 2
        self.transitions, self.lambda_values = f(n * horizon, self.row_len), f(n * horizon)
        self.grads, self.metrics = f(self.P), f(2)

        import types
        self.desc = types.SimpleNamespace(seed=seed)       # the field the optimizer sets on BpttActorGrad.desc

    def __call__(self, *, actor_params, target_critic_params, init_states, state_mean, state_std, reward_mean_std, system,
                 system_params, act_noise=None, offset: int = 0, rng_dev=None):
        import math
        n, H, X, U = self.n, self.horizon, self.x_dim, self.u_dim
        if tuple(init_states.shape) != (n, X):
            raise ValueError(f"init_states must be [{n},{X}]")
        if act_noise is None:
            act_noise = philox_normal(n * H * U, self.desc.seed, offset, 1, rng_dev, device=self.device)
        noise = act_noise.reshape(n, H, U)
        p = actor_params.detach().clone().requires_grad_(True)
        tcp = target_critic_params.detach()
        s_mean, s_std = state_mean.detach().contiguous(), state_std.detach().contiguous()
        inv_sp = math.log(math.exp(self.init_stddev) - 1.0) if self.init_stddev < 20.0 else self.init_stddev

        def actor(obs):           # Actor.__call__ (bptt_optimizer.py:131-142) on the state normaliser's view of obs
            out = HipMlp.apply(p, obs, self.actor_spec, s_mean, s_std)[0]
            mu, raw = out[:, :U], out[:, U:]
            return mu, torch.clamp(torch.nn.functional.softplus(raw + inv_sp), 1e-6, 1e2)

        obs, sp = init_states.detach(), system_params
        o_l, a_l, r_l, n_l = [], [], [], []
        for t in range(H):        # rollout_policy (optimizer_utils.py:79-101): policy(stop_gradient(obs)), then System.step
            mu, sig = actor(obs.detach())
            a = torch.clamp(torch.tanh(mu + noise[:, t] * sig), -0.999, 0.999)          # act (:313-317)
            st = system.step(obs, a, sp)
            nxt = st.x_next.reshape(n, X).to(torch.float32)
            r = torch.as_tensor(st.reward, device=self.device, dtype=torch.float32).reshape(-1).expand(n)
            o_l.append(obs); a_l.append(a); r_l.append(r); n_l.append(nxt)
            obs, sp = nxt, st.system_params
        observation, action = torch.stack(o_l, 1), torch.stack(a_l, 1)                 # [n, H, .]
        reward, next_observation = torch.stack(r_l, 1), torch.stack(n_l, 1)
        reward_n = (reward - reward_mean_std[0]) / reward_mean_std[1]                  # :340-341
        v = HipMlp.apply(tcp, next_observation.reshape(n * H, X), self.critic_spec, s_mean, s_std)   # :338-339, 342
        bootstrap = torch.minimum(v[0, :, 0], v[1, :, 0]).reshape(n, H)                # :343
        lam = LambdaReturnFn.apply(reward_n, bootstrap, self.discount, self.lambda_)   # :344
        disc = torch.cat([torch.ones(1, device=self.device), torch.full((H - 1,), self.discount, device=self.device)]).cumprod(0)   # :346-348
        mu, sig = actor(observation.reshape(n * H, X))                                 # get_log_prob (:144-152): obs NOT stop-gradiented
        a_flat = action.reshape(n * H, U)
        a_c = torch.clamp(a_flat, -1 + 1e-8, 1 - 1e-8)
        u_raw = 0.5 * torch.log((1 + a_c) / (1 - a_c))
        log_l = (-0.5 * ((u_raw - mu) / sig) ** 2 - torch.log(sig) - 0.5 * math.log(2 * math.pi)).sum(-1) - torch.log(1 - a_flat ** 2).sum(-1)
        entropy_loss = -log_l.reshape(n, H).mean(dim=1)                                # :351
        loss = (-(lam * disc).mean(dim=1) + entropy_loss * self.ent_coef).mean()       # :352, vmap + mean (:364-366)
        loss.backward()
        self.grads.copy_(p.grad)
        with torch.no_grad():
            self.metrics[0] = loss.detach()
            self.metrics[1] = entropy_loss.detach().mean()
            R = n * H
            self.transitions.copy_(torch.cat([observation.reshape(R, X), a_flat, reward.reshape(R, 1), torch.ones(R, 1, device=self.device),
                                              next_observation.reshape(R, X)], dim=1))
            self.lambda_values.copy_(lam.detach().reshape(R))
        self.system_params_out = sp
        return self.grads


class CriticGrad:
    """Drives mbpo_critic_grads: twin-V regression loss and gradient on a gathered minibatch (bptt_optimizer.py:385-404)."""

    def __init__(self, *, x_dim: int, critic_dims: Sequence[int], batch: int, device, activation: str = "swish"):
        self.lib = load()
        self.x_dim, self.batch, self.device = x_dim, int(batch), torch.device(device)
        self.spec = MlpSpec(list(critic_dims), activation, 2)
        self.C = self.spec.n_params
        self.dims = (C.c_int32 * (len(critic_dims)))(*[int(v) for v in critic_dims])
        self.layers = len(critic_dims) - 1
        self.act = _hip.ACT_IDS[activation]
        nws = self.lib.mbpo_critic_workspace_floats(x_dim, self.layers, self.dims, self.batch)
        if nws < 0:
            check(int(nws), "mbpo_critic_workspace_floats")
        self.workspace = torch.zeros(int(nws), device=self.device, dtype=torch.float32)
        self.grads = torch.zeros(2 * self.C, device=self.device, dtype=torch.float32)
        self.metrics = torch.zeros(1, device=self.device, dtype=torch.float32)

    def __call__(self, critic_params, transitions, lambda_values, idx, state_mean, state_std) -> torch.Tensor:
        for t, nm in ((critic_params, "critic_params"), (transitions, "transitions"), (lambda_values, "lambda_values"),
                      (state_mean, "state_mean"), (state_std, "state_std")):
            _req(t, nm)
        _req(idx, "idx", torch.int32)
        if critic_params.numel() != 2 * self.C:
            raise ValueError("critic_params must be [2*C]")
        if idx.numel() != self.batch:
            raise ValueError(f"idx must have {self.batch} entries")
        if transitions.dim() != 2 or lambda_values.numel() != transitions.shape[0]:
            raise ValueError("transitions must be [R, D] with lambda_values [R]")
        check(self.lib.mbpo_critic_grads(critic_params.data_ptr(), self.x_dim, self.layers, self.dims, self.act,
                                         transitions.data_ptr(), transitions.shape[1], lambda_values.data_ptr(), idx.data_ptr(),
                                         self.batch, state_mean.data_ptr(), state_std.data_ptr(), self.grads.data_ptr(),
                                         self.metrics.data_ptr(), self.workspace.data_ptr(), current_stream_ptr()),
              "mbpo_critic_grads")
        return self.grads


class CriticGradGeneric:
    """CriticGrad's contract (grads [2C], metrics [1]) for ANY critic hidden sizes: the twin-V regression of bptt_optimizer.py:385-404
    — critic_loss_fn = 0.5 * (mean l2(v1, lambda) + mean l2(v2, lambda)), l2 = 0.5 (.)^2 — with the two networks as HIP autograd
    nodes (HipMlp: forward and vector-Jacobian products layer by layer, csrc/layered.hip) and the loss head in torch."""

    def __init__(self, *, x_dim: int, critic_dims: Sequence[int], batch: int, device, activation: str = "swish"):
        self.x_dim, self.batch, self.device = x_dim, int(batch), torch.device(device)
        self.spec = MlpSpec(list(critic_dims), activation, 2)
        self.C = self.spec.n_params
        self.grads = torch.zeros(2 * self.C, device=self.device, dtype=torch.float32)
        self.metrics = torch.zeros(1, device=self.device, dtype=torch.float32)

    def __call__(self, critic_params, transitions, lambda_values, idx, state_mean, state_std) -> torch.Tensor:
        if idx.numel() != self.batch:
            raise ValueError(f"idx must have {self.batch} entries")
        rows = idx.long()
        obs = transitions[rows, :self.x_dim].contiguous()
        lam = lambda_values[rows]
        p = critic_params.detach().clone().requires_grad_(True)
        v = HipMlp.apply(p, obs, self.spec, state_mean.detach().contiguous(), state_std.detach().contiguous())      # [2, B, 1]
        loss = 0.5 * ((0.5 * (v[0, :, 0] - lam) ** 2).mean() + (0.5 * (v[1, :, 0] - lam) ** 2).mean())
        loss.backward()
        self.grads.copy_(p.grad)
        self.metrics[0] = loss.detach()
        return self.grads


class AdamW:
    """optax.adamw (optionally under optax.apply_if_finite) on a flat parameter vector, with an optional Polyak target.
    State (moments, count) lives on the device: graph-replayable."""

    def __init__(self, n: int, device, lr: float, weight_decay: float, apply_if_finite: bool = False):
        self.lib = load()
        self.n, self.lr, self.wd, self.apply_if_finite = int(n), float(lr), float(weight_decay), bool(apply_if_finite)
        dev = torch.device(device)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.count = torch.zeros(1, device=dev, dtype=torch.float32)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.workspace = torch.zeros(2 * ((n + 255) // 256) + 4, device=dev, dtype=torch.float32)

    def state_tensors(self):
        return self.m, self.v, self.count

    def load_state(self, m, v, count):
        self.m.copy_(m); self.v.copy_(v); self.count.copy_(torch.as_tensor(count, dtype=torch.float32).reshape(1))

    def step(self, params: torch.Tensor, grads: torch.Tensor, target: Optional[torch.Tensor] = None, tau: float = 0.0,
             grad_scale: float = 1.0) -> None:
        _req(params, "params"); _req(grads, "grads")
        if params.numel() != self.n or grads.numel() != self.n:
            raise ValueError("params/grads must have n elements")
        if target is not None:
            _req(target, "target")
            if target.numel() != self.n:
                raise ValueError("target must have n elements")
        check(self.lib.mbpo_adamw_step(params.data_ptr(), grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                       self.count.data_ptr(), self.n, self.lr, self.wd, grad_scale, int(self.apply_if_finite),
                                       ptr(target), tau, self.grad_norm.data_ptr(), self.workspace.data_ptr(),
                                       current_stream_ptr()), "mbpo_adamw_step")


class EnsembleNllGrad:
    """Drives mbpo_ens_nll_grads: per-member Gaussian NLL loss and gradient on per-member minibatches (N3)."""

    def __init__(self, *, x_dim: int, u_dim: int, spec: MlpSpec, batch: int, device, predict_delta: bool = True, min_std: float = 1e-3):
        self.lib = load()
        self.x_dim, self.u_dim, self.spec, self.batch = x_dim, u_dim, spec, int(batch)
        self.device = torch.device(device)
        self.E, self.n_params = spec.n_nets, spec.n_params
        d = _hip.EnsTrainDesc()
        d.x_dim, d.u_dim, d.batch = x_dim, u_dim, self.batch
        d.predict_delta, d.min_std = int(predict_delta), min_std
        self.desc = d
        self.grads = torch.zeros(self.E * self.n_params, device=self.device, dtype=torch.float32)
        self.metrics = torch.zeros(self.E, device=self.device, dtype=torch.float32)
        self.workspace = None

    def __call__(self, params: torch.Tensor, rows: torch.Tensor, idx: torch.Tensor, next_obs_off: Optional[int] = None) -> torch.Tensor:
        _req(params, "params"); _req(rows, "rows"); _req(idx, "idx", torch.int32)
        if params.numel() != self.E * self.n_params:
            raise ValueError("params must hold E * n_params floats")
        if rows.dim() != 2 or tuple(idx.shape) != (self.E, self.batch):
            raise ValueError(f"rows must be [R, D] and idx [{self.E},{self.batch}]")
        d = self.desc
        d.dynamics = self.spec.desc(params)
        d.rows, d.row_len = rows.data_ptr(), rows.shape[1]
        d.next_obs_off = self.x_dim + self.u_dim + 2 if next_obs_off is None else next_obs_off
        d.idx = idx.data_ptr()
        d.grads, d.metrics = self.grads.data_ptr(), self.metrics.data_ptr()
        if self.workspace is None:
            nws = self.lib.mbpo_ens_nll_workspace_floats(C.byref(d))
            if nws < 0:
                check(int(nws), "mbpo_ens_nll_workspace_floats")
            self.workspace = torch.zeros(int(nws), device=self.device, dtype=torch.float32)
        d.workspace = self.workspace.data_ptr()
        check(self.lib.mbpo_ens_nll_grads(C.byref(d), current_stream_ptr()), "mbpo_ens_nll_grads")
        return self.grads

"""mbpo.utils.optimizer_utils — the reference's free functions (mbpo/utils/optimizer_utils.py:11-161) with the same names and
argument order, over the kernels of libmbpo_hip.so.

  rollout_actions(system, system_params, init_state, actions, horizon)                       :12-59   -> mbpo_model_rollout(actions=...)
  rollout_policy(system, system_params, init_state, policy, policy_state, horizon, stop_grads) :63-116  -> policy callable + System.step per step
  lambda_return(reward, next_values, discount, lambda_)                                        :120-132 -> mbpo_lambda_return_scan
  static_scan(fn, inputs, start, reverse)                                                      :135-152 -> host loop (a Python `fn` cannot run in a kernel)
  soft_update(target_params, online_params, tau)                                               :155-161 -> mbpo_soft_update

Differences a reference user should know: the reference writes single-trajectory functions and vmaps them; here `init_state`
may be [x] or a batch [N, x] (actions [H, u] or [H, N, u]) and the leading axes of the returned Transition follow: [H, ...] or
[H, N, ...].  These functions are forward only: on this path the gradient through a rollout is taken inside mbpo_bptt_actor_grads
(BPTTOptimizer), not by differentiating Python code, so `stop_grads` is accepted and has nothing to stop.
"""
from __future__ import annotations

from typing import Any, Callable

import torch

from mbpo import _hip, ops
from mbpo.systems.base_systems import System, SystemParams
from mbpo.types import Transition


def _device(t: torch.Tensor) -> torch.device:
    return t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())


def rollout_actions(system: System, system_params: SystemParams, init_state: torch.Tensor, actions: torch.Tensor,
                    horizon: int) -> Transition:
    """optimizer_utils.py:12-59: open-loop actions through System.step, one fused launch for all steps (a user-defined
    System: one System.step per step).  observation = [init, x'_0 .. x'_{H-2}], discount = 1."""
    assert actions.shape[0] == horizon
    X, U = system.x_dim, system.u_dim
    dev = _device(init_state)
    single = init_state.dim() == 1
    obs = init_state.reshape(-1, X).to(dev, torch.float32).contiguous().clone()
    N = obs.shape[0]
    acts = actions.reshape(horizon, -1, U).to(dev, torch.float32)
    if acts.shape[1] != N:
        raise ValueError(f"actions must be [H,u] or [H,{N},u]")
    spec = system.rollout_spec(system_params, dev)
    if spec["system_kind"] == _hip.SYS_GENERIC:
        # a user-defined System: the reference's scan (:31-47) is a plain loop over System.step — it ignores SystemState.done (no
        # Episode / AutoReset bookkeeping on this path: a terminating System keeps propagating x_next) and carries system_params
        sp, o, r, n = system_params, [], [], []
        for h in range(horizon):
            out = system.step(x=obs, u=acts[h].contiguous(), system_params=sp)
            o.append(obs)
            r.append(torch.as_tensor(out.reward, device=dev, dtype=torch.float32).reshape(-1).expand(N))
            n.append(out.x_next.reshape(-1, X).to(torch.float32))
            obs, sp = n[-1], out.system_params
        rew = torch.stack(r)
        tr = Transition(observation=torch.stack(o), action=acts, reward=rew, discount=torch.ones_like(rew), next_observation=torch.stack(n))
        if single:
            tr = Transition(observation=tr.observation[:, 0], action=tr.action[:, 0], reward=tr.reward[:, 0], discount=tr.discount[:, 0],
                            next_observation=tr.next_observation[:, 0])
        return tr
    # fused systems (Pendulum, learned ensemble) never report done and the episode never ends: one launch for all steps
    z = torch.zeros(N, device=dev)
    rows = ops.model_rollout(x_dim=X, u_dim=U, actions=acts.contiguous(), obs=obs, first_obs=obs.clone(), steps=z, done=z.clone(),
                             n_steps=horizon, episode_length=2 ** 30, seed=system_params.key, **spec)
    return _transition_from_rows(rows, X, U, horizon, N, single)


def _transition_from_rows(rows: torch.Tensor, X: int, U: int, H: int, N: int, single: bool) -> Transition:
    r = rows.reshape(H, N, -1)
    tr = Transition(observation=r[..., :X], action=r[..., X:X + U], reward=r[..., X + U], discount=torch.ones_like(r[..., X + U]),
                    next_observation=r[..., X + U + 2:2 * X + U + 2])
    if single:
        tr = Transition(observation=tr.observation[:, 0], action=tr.action[:, 0], reward=tr.reward[:, 0], discount=tr.discount[:, 0],
                        next_observation=tr.next_observation[:, 0])
    return tr


def rollout_policy(system: System, system_params: SystemParams, init_state: torch.Tensor, policy: Callable, policy_state: Any,
                   horizon: int, stop_grads: bool = True) -> Transition:
    """optimizer_utils.py:63-116: `policy(obs, policy_state) -> (action, new_policy_state)` is an arbitrary callable (e.g.
    `optimizer.act`), so the steps are walked on the host: the callable, then System.step (one launch).  The trainers never
    take this route — they hand the whole unroll to the fused kernel."""
    X, U = system.x_dim, system.u_dim
    dev = _device(init_state)
    single = init_state.dim() == 1
    obs = init_state.reshape(-1, X).to(dev, torch.float32)
    sp, ps = system_params, policy_state
    o, a, r, n = [], [], [], []
    for _ in range(horizon):
        acs, ps = policy(obs[0] if single else obs, ps)
        acs = acs.reshape(-1, U).to(dev, torch.float32)
        out = system.step(x=obs, u=acs, system_params=sp)
        o.append(obs); a.append(acs); r.append(out.reward.reshape(-1)); n.append(out.x_next.reshape(-1, X))
        obs, sp = n[-1], out.system_params
    tr = Transition(observation=torch.stack(o), action=torch.stack(a), reward=torch.stack(r), discount=torch.ones_like(torch.stack(r)),
                    next_observation=torch.stack(n))
    if single:
        tr = Transition(observation=tr.observation[:, 0], action=tr.action[:, 0], reward=tr.reward[:, 0], discount=tr.discount[:, 0],
                        next_observation=tr.next_observation[:, 0])
    return tr


def lambda_return(reward: torch.Tensor, next_values: torch.Tensor, discount: float, lambda_: float) -> torch.Tensor:
    """optimizer_utils.py:120-132 (time on the leading axis: [H] or [H, B])."""
    assert reward.dim() == next_values.dim(), (reward.shape, next_values.shape)
    dev = _device(reward)
    r = reward.to(dev, torch.float32)
    v = next_values.to(dev, torch.float32)
    if r.dim() == 1:
        return ops.lambda_return_scan(r.reshape(-1, 1).contiguous(), v.reshape(-1, 1).contiguous(), discount, lambda_, time_major=True)[:, 0]
    H = r.shape[0]
    out = ops.lambda_return_scan(r.reshape(H, -1).contiguous(), v.reshape(H, -1).contiguous(), discount, lambda_, time_major=True)
    return out.reshape(r.shape)


def static_scan(fn: Callable, inputs: torch.Tensor, start: torch.Tensor, reverse: bool = False) -> torch.Tensor:
    """optimizer_utils.py:135-152 with a Python `fn` (API parity; the hot recurrences are the scan kernels)."""
    xs = list(inputs.flip(0) if reverse else inputs)
    outs, carry = [], start
    for x in xs:
        carry = fn(carry, x)
        outs.append(carry)
    out = torch.stack(outs)
    return out.flip(0) if reverse else out


def soft_update(target_params, online_params, tau: float = 0.005):
    """optimizer_utils.py:155-161: (1 - tau) * old + tau * new over a tensor or a (nested) tuple / list / dict of tensors."""
    if isinstance(target_params, dict):
        return {k: soft_update(v, online_params[k], tau) for k, v in target_params.items()}
    if isinstance(target_params, (tuple, list)):
        return type(target_params)(soft_update(t, o, tau) for t, o in zip(target_params, online_params))
    t = _hip.require_device_tensor(target_params.contiguous(), "target_params")
    o = _hip.require_device_tensor(online_params.contiguous(), "online_params")
    if t.shape != o.shape:
        raise ValueError("target and online parameters must have the same shape")
    out = torch.empty_like(t)
    _hip.check(_hip.load().mbpo_soft_update(t.data_ptr(), o.data_ptr(), out.data_ptr(), t.numel(), float(tau), _hip.current_stream_ptr()),
               "mbpo_soft_update")
    return out

"""Model rollout (HOT LOOP 1) — torch-CPU restatement of the env-adapter stack and actor_step.

Test infrastructure only.  Mirrors, layer by layer, what one `acting.actor_step` does under the reference's
wrapper stack AutoReset(Vmap(Episode(BraxWrapper))) (sac/sac.py:140-142, brax_utils/training.py:44-47):

  policy           sac_networks.py:58-73 (SAC) / ppo_network.py:59-84 (PPO extras log_prob, raw_action)
  AutoReset.step   brax_utils/training.py:119-137   (resets to the env's FIRST obs, no re-sample)
  Episode.step     brax_utils/training.py:91-107    (action_repeat inner scan, steps/done/truncation)
  BraxWrapper.step systems/brax_wrapper.py:40-50  -> System.step (pendulum_system.py:18-39)
  Transition       sac/acting.py:46-55              (next_observation is the POST-auto-reset obs)
  concat           sac/sac.py:296 (step-major rows)  /  ppo/ppo.py:210-213 (env-major [B*M, T])

Randomness enters as explicit tensors: policy_noise [S,N,u], model_noise [S,AR,N,x], member_idx [S,AR,N].
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import nets


@dataclass
class EnvState:
    """brax State fields the loop carries: obs, done, info['steps'], info['first_obs'] (brax_utils/base.py:12-23)."""
    obs: torch.Tensor        # [N,x]
    first_obs: torch.Tensor  # [N,x]
    steps: torch.Tensor      # [N] float
    done: torch.Tensor       # [N] float

    def clone(self):
        return EnvState(self.obs.clone(), self.first_obs.clone(), self.steps.clone(), self.done.clone())


def row_len(x_dim: int, u_dim: int, ppo_extras: bool) -> int:
    """ravel_pytree order of brax Transition(observation, action, reward, discount, next_observation, extras)
    with extras = {'policy_extras': {log_prob, raw_action}?, 'state_extras': {'truncation'}} (sac/sac.py:194-200)."""
    return 2 * x_dim + u_dim + 3 + ((1 + u_dim) if ppo_extras else 0)


def env_step(system, st: EnvState, action: torch.Tensor, episode_length: int, action_repeat: int, s: int,
             member_idx=None, model_noise=None):
    """AutoReset.step(Episode.step(BraxWrapper.step)).  Returns (new_state, reward, truncation)."""
    N = st.obs.shape[0]
    # AutoReset.step :120-124 — zero `steps` where previously done, clear done
    steps = torch.where(st.done != 0, torch.zeros_like(st.steps), st.steps)
    obs = st.obs
    reward = torch.zeros(N, dtype=obs.dtype)
    env_index = torch.arange(N)
    sys_done = torch.zeros(N, dtype=obs.dtype)                      # SystemState.done default 0.0 (base_systems.py:25)
    for ar in range(action_repeat):  # Episode.step :92-97 — scan over action_repeat, rewards summed
        kw = {}
        if member_idx is not None:
            kw["member_idx"] = member_idx[s, ar]
        if model_noise is not None:
            kw["model_noise"] = model_noise[s, ar]
        res = system.step(obs, action, env_index=env_index, **kw)
        obs, r = res[0], res[1]
        if len(res) > 2:                                            # a System that reports termination (SystemState.done)
            sys_done = res[2].to(obs.dtype)
        reward = reward + r
    steps = steps + action_repeat                                   # :98
    over = steps >= episode_length
    done = torch.where(over, torch.ones_like(sys_done), sys_done)   # :102
    trunc = torch.where(over, 1 - sys_done, torch.zeros_like(sys_done))   # :103-105
    # AutoReset.step :126-137 — obs <- first_obs where done
    obs = torch.where(done[:, None] != 0, st.first_obs, obs)
    return EnvState(obs, st.first_obs, steps, done), reward, trunc


def rollout(system, policy_params: torch.Tensor, policy_dims, st: EnvState, n_steps: int, episode_length: int,
            action_repeat: int = 1, act: str = "swish", norm_mean=None, norm_std=None, policy_noise=None,
            model_noise=None, member_idx=None, deterministic: bool = False, ppo_extras: bool = False,
            env_major: bool = False):
    """get_experience's scan (sac/sac.py:283-296) / generate_unroll (sac/acting.py:58-78).

    Returns (final EnvState, rows [S*N, D]) with rows laid out as the reference's flattened replay rows.
    """
    N, X = st.obs.shape
    U = policy_dims[-1] // 2
    D = row_len(X, U, ppo_extras)
    rows = torch.zeros(n_steps, N, D, dtype=st.obs.dtype)
    st = st.clone()
    for s in range(n_steps):
        obs = st.obs
        logits = nets.mlp_forward(policy_params, policy_dims, nets.normalize(obs, norm_mean, norm_std), act)
        if deterministic:
            z = nets.split_logits(logits)[0]
        else:
            z = nets.sample_no_postprocessing(logits, policy_noise[s])
        action = nets.postprocess(z)
        nst, reward, trunc = env_step(system, st, action, episode_length, action_repeat, s, member_idx, model_noise)
        r = rows[s]
        r[:, 0:X] = obs                        # observation = env_state.obs
        r[:, X:X + U] = action
        r[:, X + U] = reward
        r[:, X + U + 1] = 1 - nst.done         # discount = 1 - nstate.done
        r[:, X + U + 2:2 * X + U + 2] = nst.obs
        if ppo_extras:
            r[:, 2 * X + U + 2] = nets.log_prob(logits, z)
            r[:, 2 * X + U + 3:2 * X + 2 * U + 3] = z
        r[:, D - 1] = trunc
        st = nst
    if env_major:
        rows = rows.permute(1, 0, 2)           # ppo.py:210-213 swapaxes + reshape -> [N*T, D]
    return st, rows.reshape(n_steps * N, D).contiguous()


def brax_wrapper_reset(data_logical, insert_position: int, sample_position: int, keys, x_dim: int, u_dim: int):
    """BraxWrapper.reset under VmapWrapper.reset (systems/brax_wrapper.py:25-38, brax_utils/training.py:66-69): every env draws ONE
    transition uniformly from the true buffer and starts at its observation; reward = that row's reward, done = 0.
    The reference gives each env its own key (vmap over split keys); the build draws all N indices from the first key's
    first split with the env id as the Philox element index — N independent uniform draws either way.  An empty buffer yields
    index 0 (randint(0, 0) -> 0: the all-zero dummy row, base_optimizer.py:43-57).
    Returns (idx int32 [N], EnvState, reward [N], system key)."""
    import numpy as np
    from . import philox
    n = len(keys)
    k0, k1 = philox.split(keys[0])
    mx = data_logical.shape[0]
    if insert_position - sample_position > 0:
        idx = philox.philox_randint(k0, 0, philox.STREAM_REPLAY, np.arange(n, dtype=np.uint64), sample_position, insert_position)
    else:
        idx = np.zeros(n, np.int32)
    rows = data_logical[torch.from_numpy(np.mod(idx.astype(np.int64), mx))]
    obs = rows[:, :x_dim].clone()
    return idx, EnvState(obs, obs.clone(), torch.zeros(n), torch.zeros(n)), rows[:, x_dim + u_dim].clone(), k1


def evaluate(system, policy_params, policy_dims, first: EnvState, episode_length: int, action_repeat: int = 1, act: str = "swish",
             norm_mean=None, norm_std=None, deterministic: bool = True, policy_noise=None):
    """Evaluator.run_evaluation's unroll under EvalWrapper (sac/acting.py:82-145, brax_utils/training.py:156-199), step by step:
        episode_reward += reward * active;  episode_steps = where(active, info['steps'], episode_steps);  active *= 1 - done
    over unroll_length = episode_length // action_repeat steps of AutoReset(Vmap(Episode(env))).
    Returns (episode_reward [N], episode_steps [N])."""
    n_steps = episode_length // action_repeat
    st = first.clone()
    N = st.obs.shape[0]
    active, ep_reward, ep_steps = torch.ones(N), torch.zeros(N), torch.zeros(N)
    for s in range(n_steps):
        logits = nets.mlp_forward(policy_params, policy_dims, nets.normalize(st.obs, norm_mean, norm_std), act)
        z = nets.split_logits(logits)[0] if deterministic else nets.sample_no_postprocessing(logits, policy_noise[s])
        nst, reward, _ = env_step(system, st, nets.postprocess(z), episode_length, action_repeat, s)
        # info['steps'] of the state AFTER the step, before AutoReset zeroes it at the next step (:176-180)
        ep_steps = torch.where(active != 0, nst.steps, ep_steps)
        ep_reward = ep_reward + reward * active
        active = active * (1 - nst.done)
        st = nst
    return ep_reward, ep_steps

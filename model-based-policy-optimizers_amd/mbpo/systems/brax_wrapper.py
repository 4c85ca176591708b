"""BraxWrapper — mirrors mbpo/systems/brax_wrapper.py:14-62, already batched and already carrying what the reference's
Episode/AutoReset wrappers add (info['steps'], info['truncation'], info['first_obs'] — brax_utils/training.py:85-89,
113-117).  `step` exists for API parity; the trainers never call it per step — they hand the whole unroll to the fused
rollout kernel (ops.model_rollout), which contains BraxWrapper.step + Episode.step + AutoReset.step.
"""
from __future__ import annotations

from typing import TYPE_CHECKING, Sequence

import torch

from mbpo import ops
from mbpo.replay import ReplayBufferState, UniformSamplingQueue
from mbpo.systems.base_systems import System, SystemParams
from mbpo.utils import keys as K

if TYPE_CHECKING:      # a module-level import would be circular when mbpo.systems is imported before mbpo.optimizers
    from mbpo.optimizers.policy_optimizers.brax_utils.base import State


class BraxWrapper:
    def __init__(self, system: System, system_params: SystemParams, sample_buffer_state: ReplayBufferState,
                 sample_buffer: UniformSamplingQueue):
        self.system = system
        self.sample_buffer_state = sample_buffer_state
        self.sample_buffer = sample_buffer
        self.init_system_params = system_params

    def reset(self, rng: Sequence[int]) -> State:
        """One state per key, each drawn uniformly from the TRUE buffer (brax_wrapper.py:25-38).  The reference vmaps a
        batch-size-1 sample over the keys; here one launch draws all N rows (Philox index = env id)."""
        from mbpo.optimizers.policy_optimizers.brax_utils.base import State
        keys = list(rng) if isinstance(rng, (list, tuple)) else [rng]
        n = len(keys)
        bs = self.sample_buffer_state
        k0, k1 = K.split(keys[0])
        X = self.system.x_dim
        if self.sample_buffer.size(bs) > 0:
            rows = ops.replay_sample(bs.data, bs.state, n, seed=k0, offset=0)
        else:
            # randint(minval=0, maxval=0) yields 0 -> row 0 of the (all-zero dummy) buffer (base_optimizer.py:43-57)
            rows = self.sample_buffer.logical_data(bs)[0:1].expand(n, -1).contiguous()
        obs = rows[:, :X].contiguous()
        reward = rows[:, X + self.system.u_dim].contiguous()
        dev = obs.device
        z = lambda: torch.zeros(n, device=dev)
        return State(pipeline_state=None, obs=obs, reward=reward, done=z(),
                     system_params=self.init_system_params.replace(key=k1),
                     info={"steps": z(), "truncation": z(), "first_obs": obs.clone()})

    def step(self, state: State, action: torch.Tensor) -> State:
        """brax_wrapper.py:40-50 (no episode bookkeeping — that is the wrappers' job)."""
        nxt = self.system.step(state.obs, action, state.system_params)
        return state.replace(obs=nxt.x_next, reward=nxt.reward, done=torch.zeros_like(state.done) + nxt.done,
                             system_params=nxt.system_params)

    @property
    def action_size(self) -> int:
        return self.system.u_dim

    @property
    def observation_size(self) -> int:
        return self.system.x_dim

    @property
    def backend(self) -> str:
        return "hip"

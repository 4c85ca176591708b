"""BaseOptimizer — mirrors mbpo/optimizers/base_optimizer.py:14-57."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Generic, Optional, Tuple

import torch

from mbpo.replay import ReplayBufferState, UniformSamplingQueue
from mbpo.systems.base_systems import System
from mbpo.systems.dynamics.base_dynamics import DynamicsParams
from mbpo.systems.rewards.base_rewards import RewardParams
from mbpo.types import Transition
from mbpo.utils import keys as K
from mbpo.utils.type_aliases import OptimizerState, OptimizerTrainingOutPut


class BaseOptimizer(ABC, Generic[RewardParams, DynamicsParams]):
    def __init__(self, system: Optional[System] = None, key: int = K.PRNGKey(0)):
        self.system = system
        self.key = key

    def set_system(self, system: System):
        self.system = system

    @property
    def can_act_in_batches(self):
        return True

    @abstractmethod
    def act(self, obs: torch.Tensor, opt_state: OptimizerState, evaluate: bool = True) -> Tuple[torch.Tensor, OptimizerState]:
        pass

    def train(self, opt_state: OptimizerState) -> OptimizerTrainingOutPut:
        return OptimizerTrainingOutPut(optimizer_state=opt_state)

    def init(self, key: int, true_buffer_state: Optional[ReplayBufferState] = None) -> OptimizerState:
        pass

    def dummy_true_buffer_state(self, key: int) -> ReplayBufferState:
        """base_optimizer.py:43-57: a 10-row all-zero buffer of (obs, action, reward, discount, next_obs) transitions."""
        assert self.system is not None, "Base optimizer requires system to be defined."
        dev = torch.device("cuda", torch.cuda.current_device())
        dummy = Transition(observation=torch.zeros(self.system.x_dim, device=dev), action=torch.zeros(self.system.u_dim, device=dev),
                           next_observation=torch.zeros(self.system.x_dim, device=dev), reward=torch.zeros(1, device=dev),
                           discount=torch.zeros(1, device=dev))
        return UniformSamplingQueue(max_replay_size=10, dummy_data_sample=dummy, sample_batch_size=1, device=dev).init(key)

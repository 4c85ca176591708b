"""BraxOptimizer / SACOptimizer / PPOOptimizer — mirrors mbpo/optimizers/policy_optimizers/brax_optimizers.py:21-115
(same constructor signatures, `init` / `act` / `train` semantics and key-split structure)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Any, List, Optional, Tuple

import torch

from mbpo.optimizers.base_optimizer import BaseOptimizer
from mbpo.replay import ReplayBufferState, UniformSamplingQueue
from mbpo.systems.base_systems import System
from mbpo.systems.brax_wrapper import BraxWrapper
from mbpo.utils import keys as K
from mbpo.utils.type_aliases import OptimizerState, OptimizerTrainingOutPut


@dataclass
class BraxState(OptimizerState):
    policy_params: Any = None          # (normalizer_params, policy_params)  (brax_optimizers.py:69-72)


@dataclass
class BraxOutput(OptimizerTrainingOutPut):
    optimizer_state: BraxState
    summary: List[dict] = dataclasses.field(default_factory=list)


class BraxOptimizer(BaseOptimizer):
    def __init__(self, agent_class, true_buffer: UniformSamplingQueue, system: Optional[System] = None, **agent_kwargs):
        super().__init__(system)
        self.agent_class = agent_class
        self.agent_kwargs = agent_kwargs
        self.true_buffer = true_buffer
        if system is None:
            self.dummy_trainer = None
            self.make_policy = None
        else:
            self.set_system(system)

    def set_system(self, system: System):
        super().set_system(system)
        self.key, sys_key, buffer_key = K.split(self.key, 3)
        dummy_true_buffer_state = self.dummy_true_buffer_state(buffer_key)
        dummy_env = BraxWrapper(system=self.system, system_params=self.system.init_params(sys_key),
                                sample_buffer_state=dummy_true_buffer_state, sample_buffer=self.true_buffer)
        self.dummy_trainer = self.agent_class(environment=dummy_env, **self.agent_kwargs)
        self.make_policy = self.dummy_trainer.make_policy

    def init(self, key: int, true_buffer_state: Optional[ReplayBufferState] = None) -> BraxState:
        assert self.system is not None, "Brax optimizer requires system to be defined."
        if true_buffer_state is None:
            dummy_buffer_key, key = K.split(key, 2)
            true_buffer_state = self.dummy_true_buffer_state(dummy_buffer_key)
        keys = K.split(key, 3)
        system_params = self.system.init_params(keys[0])
        training_state = self.dummy_trainer.init_training_state(keys[1])
        norm, pol = training_state.get_policy_params()
        policy_params = (dataclasses.replace(norm, vec=norm.vec.clone()), pol.clone())
        return BraxState(system_params=system_params, true_buffer_state=true_buffer_state, policy_params=policy_params,
                         key=keys[2])

    def act(self, obs: torch.Tensor, opt_state: BraxState, evaluate: bool = True) -> Tuple[torch.Tensor, BraxState]:
        assert self.system is not None, "Brax optimizer requires system to be defined."
        policy = self.make_policy(opt_state.policy_params, evaluate)
        key, subkey = K.split(opt_state.key)
        action = policy(obs, subkey)[0]
        return action, opt_state.replace(key=key)

    def train(self, opt_state: BraxState) -> BraxOutput:
        """brax_optimizers.py:86-99: a NEW trainer per call (policy/critics re-initialised inside run_training)."""
        assert self.system is not None, "Brax optimizer requires system to be defined."
        env = BraxWrapper(system=self.system, system_params=opt_state.system_params,
                          sample_buffer_state=opt_state.true_buffer_state, sample_buffer=self.true_buffer)
        trainer = self.agent_class(environment=env, **self.agent_kwargs)
        key, new_key = K.split(opt_state.key)
        try:
            policy_params, metrics = trainer.run_training(key=new_key)
        finally:
            trainer.close()      # the captured graph and the peer-memory regions belong to this trainer (one per train())
        new_opt_state = opt_state.replace(policy_params=policy_params, key=new_key)
        return BraxOutput(optimizer_state=new_opt_state, summary=metrics)


class SACOptimizer(BraxOptimizer):
    def __init__(self, true_buffer: UniformSamplingQueue, system: Optional[System] = None, **sac_kwargs):
        from mbpo.optimizers.policy_optimizers.sac.sac import SAC
        super().__init__(agent_class=SAC, system=system, true_buffer=true_buffer, **sac_kwargs)


class PPOOptimizer(BraxOptimizer):
    def __init__(self, true_buffer: UniformSamplingQueue, system: Optional[System] = None, **ppo_kwargs):
        from mbpo.optimizers.policy_optimizers.ppo.ppo import PPO
        super().__init__(agent_class=PPO, system=system, true_buffer=true_buffer, **ppo_kwargs)

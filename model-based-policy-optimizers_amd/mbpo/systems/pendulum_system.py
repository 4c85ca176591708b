"""PendulumSystem — mirrors mbpo/systems/pendulum_system.py:12-46."""
from __future__ import annotations

import dataclasses

import torch

from mbpo import _hip
from mbpo.systems.base_systems import System, SystemParams, SystemState, _device_of
from mbpo.systems.dynamics.pendulum_dynamics import PendulumDynamics, PendulumDynamicsParams
from mbpo.systems.rewards.pendulum_reward import PendulumReward, PendulumRewardParams


class PendulumSystem(System[PendulumDynamicsParams, PendulumRewardParams]):
    def __init__(self):
        super().__init__(dynamics=PendulumDynamics(), reward=PendulumReward())
        self.min_action = -1.0
        self.max_action = 1.0

    def rollout_spec(self, system_params: SystemParams, device) -> dict:
        dp = system_params.dynamics_params or PendulumDynamicsParams()
        rp = system_params.reward_params or PendulumRewardParams()
        ck = (dataclasses.astuple(dp), dataclasses.astuple(rp), str(device))
        if getattr(self, "_spec_key", None) != ck:     # device vectors are cached: no H2D copy inside a captured graph
            kind, rvec = self.reward.kernel_spec(rp, device)
            self._spec = dict(system_kind=_hip.SYS_PENDULUM, sys_params=dp.vector(device), reward_kind=kind, reward_params=rvec)
            self._spec_key = ck
        return dict(self._spec)

    def reset(self, rng=None, device=None) -> SystemState:
        """pendulum_system.py:41-46: hanging-down start [-1, 0, 0], reward 0."""
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        return SystemState(x_next=torch.tensor([-1.0, 0.0, 0.0], device=dev), reward=torch.zeros((), device=dev),
                           system_params=SystemParams(dynamics_params=PendulumDynamicsParams(),
                                                      reward_params=PendulumRewardParams()))


def _pendulum_step(x, u, dyn_params, rew_params):
    sys = PendulumSystem()
    st = sys.step(x, u, SystemParams(dynamics_params=dyn_params, reward_params=rew_params or PendulumRewardParams()))
    return st.x_next, st.reward

"""iCEM trajectory optimizer — mirrors mbpo/optimizers/trajectory_optimizers/icem_optimizer.py:25-330 on the MI355X kernels.

One `optimize` = num_steps iterations of
    mbpo_icem_sample   coloured-noise candidates around (mean, std), previous elites appended                (:168-190)
    mbpo_model_rollout open-loop rollouts of every candidate x particle through System.step               (rollout_actions)
    mbpo_icem_update   objective, elites, soft mean/std update, best-so-far, elites carried over              (:193-232)
with all optimizer state in flat device vectors; the host only sequences launches.  Keys are integers (mbpo.utils.keys);
device noise is Philox(seed = split key, offset = iteration).  A user `cost_fn` (icem_optimizer.py:99,161-166: a callable over one
trajectory, `cost_fn(observation [H, x], action [H, u]) -> scalar`) cannot run inside the kernels: it is evaluated between the
rollout launch and the update launch, vmapped over all (candidate, particle) trajectories with torch.func.vmap on the device rows
(the reference vmaps it too), and enters the objective inside mbpo_icem_update_constrained: reward - lambda_constraint * relu(cost),
cost summarised over particles by mean (use_pessimism: max).  use_optimism -> max over particles of the reward.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from dataclasses import dataclass
from typing import Any, Generic, List, Mapping, NamedTuple, Optional, Sequence, Tuple, Union

import torch

from mbpo import _hip, ops
from mbpo.optimizers.base_optimizer import BaseOptimizer
from mbpo.replay import ReplayBufferState
from mbpo.systems.base_systems import System
from mbpo.systems.dynamics.base_dynamics import DynamicsParams
from mbpo.systems.rewards.base_rewards import RewardParams
from mbpo.utils import keys as K
from mbpo.utils.type_aliases import OptimizerState, OptimizerTrainingOutPut


class iCemParams(NamedTuple):
    """icem_optimizer.py:25-50 (same fields and defaults)."""
    num_particles: int = 10
    num_samples: int = 500
    num_elites: int = 50
    init_std: float = 0.5
    alpha: float = 0.0
    num_steps: int = 5
    exponent: float = 0.0
    elite_set_fraction: float = 0.3
    u_min: Union[float, Sequence[float]] = -1.0
    u_max: Union[float, Sequence[float]] = 1.0
    warm_start: bool = True
    lambda_constraint: float = 1e4


@dataclass
class iCemOptimizerState(OptimizerState, Generic[DynamicsParams, RewardParams]):
    best_sequence: torch.Tensor = None     # [horizon, action_dim]
    best_reward: torch.Tensor = None       # scalar

    @property
    def action(self):
        return self.best_sequence[0]


@dataclass
class iCemTrainingOutput(OptimizerTrainingOutPut, Generic[DynamicsParams, RewardParams]):
    optimizer_state: iCemOptimizerState = None
    summary: List[Mapping[str, Any]] = None


class iCemTO(BaseOptimizer, Generic[DynamicsParams, RewardParams]):
    def __init__(self, horizon: int, action_dim: int, key: int = K.PRNGKey(0), opt_params: iCemParams = iCemParams(), cost_fn=None,
                 use_optimism: bool = False, use_pessimism: bool = False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.cost_fn = cost_fn       # evaluated on the host side of the seam, between two launches (see the module docstring)
        self.lib = _hip.load()
        self.horizon, self.action_dim = int(horizon), int(action_dim)
        self.opt_params, self.key = opt_params, key
        self.opt_dim = (self.horizon, self.action_dim)
        self.use_optimism, self.use_pessimism = use_optimism, use_pessimism
        p = opt_params
        self.num_prev = max(int(p.elite_set_fraction * p.num_elites), 1)
        if not (0 < p.num_elites <= p.num_samples + self.num_prev):
            raise ValueError("num_elites must be in (0, num_samples + carried elites]")
        self._bufs = None

    # -- reference API ------------------------------------------------------------------------------------------------
    def init(self, key: int, true_buffer_state: Optional[ReplayBufferState] = None) -> iCemOptimizerState:
        assert self.system is not None, "iCem optimizer requires system to be defined."
        init_key, dummy_buffer_key, key = K.split(key, 3)
        dev = torch.device("cuda", torch.cuda.current_device())
        return iCemOptimizerState(true_buffer_state=self.dummy_true_buffer_state(dummy_buffer_key), system_params=self.system.init_params(init_key),
                                  best_sequence=torch.zeros(self.opt_dim, device=dev), best_reward=torch.zeros((), device=dev), key=key)

    def _buffers(self, dev):
        if self._bufs is None or self._bufs["dev"] != dev:
            p, H, U = self.opt_params, self.horizon, self.action_dim
            NC = p.num_samples + self.num_prev
            N = NC * p.num_particles
            f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)

            def vec(v):      # scalar or per-dimension bound -> [U] device vector
                t = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
                return (t.expand(U) if t.numel() == 1 else t.reshape(U)).contiguous().to(dev)
            self._bufs = dict(dev=dev, NC=NC, N=N, mean=f(H, U), std=f(H, U), best_value=f(1), best_seq=f(H, U), prev=f(self.num_prev, H, U),
                              actions=f(H, N, U), cand=f(NC, H, U), values=f(NC), rank=torch.zeros(NC, device=dev, dtype=torch.int32),
                              u_min=vec(p.u_min), u_max=vec(p.u_max), obs=f(N, self.system.x_dim), first=f(N, self.system.x_dim),
                              steps=f(N), done=f(N), rows=f(H * N, 2 * self.system.x_dim + U + 3))
        return self._bufs

    def optimize(self, initial_state: torch.Tensor, opt_state: iCemOptimizerState) -> iCemOptimizerState:
        assert self.system is not None, "iCem optimizer requires system to be defined."
        p, H, U = self.opt_params, self.horizon, self.action_dim
        dev = initial_state.device if initial_state.is_cuda else torch.device("cuda", torch.cuda.current_device())
        b = self._buffers(dev)
        X = self.system.x_dim
        x0 = initial_state.reshape(-1).to(dev, torch.float32)
        # initial distribution; warm start = previous best sequence shifted by one, last action repeated (:240-246)
        b["mean"].zero_()
        if p.warm_start:
            b["mean"][:-1].copy_(opt_state.best_sequence[1:])
            b["mean"][-1].copy_(opt_state.best_sequence[-1])
        b["std"].fill_(p.init_std)
        b["best_value"].fill_(float("-inf"))
        b["best_seq"].copy_(b["mean"])
        b["prev"].zero_()
        optimizer_key, key = K.split(opt_state.key, 2)
        spec = self.system.rollout_spec(opt_state.system_params, dev)
        st = _hip.current_stream_ptr()
        lib = self.lib
        carry_key = optimizer_key
        for it in range(p.num_steps):
            sampling_key, particles_key = K.split(carry_key, 2)          # :170-173 (the carried key is the first sampling split)
            carry_key = K.split(sampling_key, 2)[0]
            _hip.check(lib.mbpo_icem_sample(b["mean"].data_ptr(), b["std"].data_ptr(), b["prev"].data_ptr(), b["u_min"].data_ptr(),
                                            b["u_max"].data_ptr(), p.num_samples, self.num_prev, H, U, p.num_particles, float(p.exponent),
                                            sampling_key, it, None, b["actions"].data_ptr(), b["cand"].data_ptr(), st), "mbpo_icem_sample")
            b["obs"].copy_(x0.expand(b["N"], X))
            b["first"].copy_(b["obs"])
            b["steps"].zero_(); b["done"].zero_()
            ops.model_rollout(x_dim=X, u_dim=U, actions=b["actions"], obs=b["obs"], first_obs=b["first"], steps=b["steps"], done=b["done"],
                              n_steps=H, episode_length=2 ** 30, seed=particles_key, offset=it, out=b["rows"], **spec)
            cost_ptr = None
            if self.cost_fn is not None:
                rows3 = b["rows"].reshape(H, b["N"], -1)
                obs_t = rows3[:, :, :X].transpose(0, 1)                    # [N, H, x]: transitions.observation per (candidate, particle)
                act_t = rows3[:, :, X:X + U].transpose(0, 1)               # [N, H, u]
                cost = torch.func.vmap(self.cost_fn)(obs_t, act_t)         # :162  vmap(self.cost_fn)(observation, action)
                cost = torch.as_tensor(cost, device=dev, dtype=torch.float32).reshape(-1).contiguous()
                assert cost.numel() == b["N"], "cost_fn must return one scalar per trajectory"       # :163
                b["cost"] = cost
                cost_ptr = cost.data_ptr()
            _hip.check(lib.mbpo_icem_update_constrained(
                b["rows"].data_ptr(), b["rows"].shape[1], X + U, b["NC"], p.num_particles, H, U, b["cand"].data_ptr(), p.num_elites,
                self.num_prev, float(p.alpha), int(self.use_optimism), cost_ptr, float(p.lambda_constraint), int(self.use_pessimism),
                b["mean"].data_ptr(), b["std"].data_ptr(), b["best_value"].data_ptr(), b["best_seq"].data_ptr(), b["prev"].data_ptr(),
                b["values"].data_ptr(), b["rank"].data_ptr(), st), "mbpo_icem_update_constrained")
        return opt_state.replace(key=key, best_sequence=b["best_seq"].clone(), best_reward=b["best_value"][0].clone())

    def act(self, obs: torch.Tensor, opt_state: iCemOptimizerState, evaluate: bool = True) -> Tuple[torch.Tensor, iCemOptimizerState]:
        new_opt_state = self.optimize(initial_state=obs, opt_state=opt_state)
        return new_opt_state.action, new_opt_state


class iCEMOptimizer(BaseOptimizer):
    """iCEM wrapper with the SAC/PPO optimizers' interface (icem_optimizer.py:259-320)."""

    def __init__(self, horizon: int, opt_params: iCemParams = iCemParams(), system: Optional[System] = None, key: int = K.PRNGKey(0),
                 **agent_kwargs):
        super().__init__(system, key)
        self.horizon, self.key, self.opt_params = horizon, key, opt_params
        self.agent_class, self.agent_kwargs = iCemTO, agent_kwargs
        if system is not None:
            self.set_system(system)

    @property
    def can_act_in_batches(self):
        return False

    def init(self, key: int, true_buffer_state=None) -> iCemOptimizerState:
        assert self.system is not None, "iCEM optimizer requires system to be defined."
        self.agent = self.agent_class(horizon=self.horizon, action_dim=self.system.u_dim, key=self.key, opt_params=self.opt_params,
                                      **self.agent_kwargs)
        self.agent.set_system(self.system)
        if true_buffer_state is None:
            dummy_buffer_key, key = K.split(key, 2)
            true_buffer_state = self.dummy_true_buffer_state(dummy_buffer_key)
        agent_state = self.agent.init(key)
        return agent_state.replace(true_buffer_state=true_buffer_state)

    def act(self, obs: torch.Tensor, opt_state: iCemOptimizerState, evaluate: bool = True) -> Tuple[torch.Tensor, iCemOptimizerState]:
        assert self.system is not None, "iCEM optimizer requires system to be defined."
        action, opt_state = self.agent.act(obs.reshape(-1), opt_state, evaluate)
        return action.reshape(1, -1), opt_state

    def train(self, opt_state: iCemOptimizerState) -> iCemTrainingOutput:
        training_output = super().train(opt_state)
        return iCemTrainingOutput(optimizer_state=training_output.optimizer_state, summary=[])

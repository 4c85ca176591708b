"""BPTTOptimizer — mirrors mbpo/optimizers/policy_optimizers/bptt_optimizer.py:176-538 on the MI355X kernels.

One train step (reference `_train_step` + the `step` body of `train`, :355-437, :463-522) is this launch sequence:

  mbpo_replay_sample      initial states from the sampling buffer                         (:455-458)
  mbpo_bptt_actor_grads   rollout through the model, lambda-returns, backward sweep      (:361-372)
  mbpo_adamw_step         apply_if_finite(adamw) on the actor                            (:374-378)
  mbpo_replay_sample      critic minibatch indices, randint(0, n*H)                      (:380-386)
  K x (mbpo_critic_grads + mbpo_adamw_step with the Polyak target)                       (:388-419)
  mbpo_running_stats_*    state and reward normalisers on the simulated transitions      (:297-303, :475-476)
  mbpo_replay_insert      simulated transitions into the sampling buffer                 (:477-478)

All state lives on the device in flat fp32 vectors (layouts in include/mbpo_hip.h); the host only sequences launches.
Keys are integers (mbpo.utils.keys); the same split structure as the reference is kept, device noise comes from
Philox(seed = split key, offset = train-step counter).  No CPU fallback: the library must load and tensors must be on the GPU.
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass, field
from typing import Any, Generic, Optional, Sequence, Tuple

import torch

from mbpo import _hip, ops
from mbpo.optimizers.base_optimizer import BaseOptimizer
from mbpo.replay import ReplayBufferState, UniformSamplingQueue
from mbpo.systems.dynamics.base_dynamics import DynamicsParams
from mbpo.systems.rewards.base_rewards import RewardParams
from mbpo.types import Transition
from mbpo.utils import keys as K
from mbpo.utils.type_aliases import OptimizerState, OptimizerTrainingOutPut

EPS = 1e-8


@dataclass
class NormalizerState:
    """bptt_optimizer.py:31-35 as views into one device vector [size, mean[d], summed_variance[d], std[d]]
    (summed_variance = std^2 * size is what the reference re-forms on every update, :57)."""
    vec: torch.Tensor
    dim: int

    @property
    def size(self):
        return self.vec[0]

    @property
    def mean(self):
        return self.vec[1:1 + self.dim]

    @property
    def std(self):
        return self.vec[1 + 2 * self.dim:1 + 3 * self.dim]

    def clone(self):
        return NormalizerState(self.vec.clone(), self.dim)


class Normalizer:
    """bptt_optimizer.py:38-77.  `update` runs in mbpo_running_stats_* with the BPTT clip (std >= 1e-8, no ceiling)."""

    def __init__(self, input_shape):
        self.input_shape = tuple(input_shape)
        self.dim = int(self.input_shape[0])

    def initialize_normalizer_state(self, device=None) -> NormalizerState:
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        vec = torch.zeros(1 + 3 * self.dim, device=dev, dtype=torch.float32)
        vec[1 + 2 * self.dim:] = 1.0
        return NormalizerState(vec, self.dim)

    def update(self, x: torch.Tensor, state: NormalizerState) -> NormalizerState:
        rows = x.reshape(-1, self.dim).to(torch.float32).contiguous()
        new = state.clone()
        ops.running_stats_update(rows, 0, self.dim, new.vec, std_min=EPS, std_max=float("inf"))
        return new

    @staticmethod
    def normalize(x, state: NormalizerState):
        return (x - state.mean) / state.std

    @staticmethod
    def inverse(x, state: NormalizerState):
        return x * state.std + state.mean


@dataclass
class AdamWState:
    """optax.apply_if_finite(adamw) state: moments + count (device scalar)."""
    mu: torch.Tensor
    nu: torch.Tensor
    count: torch.Tensor

    def clone(self):
        return AdamWState(self.mu.clone(), self.nu.clone(), self.count.clone())


@dataclass
class BPTTState(OptimizerState, Generic[DynamicsParams, RewardParams]):
    """bptt_optimizer.py:80-88.  actor_params [P]; critic_params / target_critic_params [2*C] = [critic_1 | critic_2]."""
    actor_opt_state: AdamWState = None
    actor_params: torch.Tensor = None
    critic_opt_state: AdamWState = None
    critic_params: torch.Tensor = None
    target_critic_params: torch.Tensor = None
    state_normalizer_state: NormalizerState = None
    reward_normalizer_state: NormalizerState = None


@dataclass
class BPTTAgentSummary:
    """bptt_optimizer.py:91-98; after train() each field is a [train_steps] device tensor (the scan's stacked outputs)."""
    actor_grad_norm: Any = 0.0
    critic_grad_norm: Any = 0.0
    actor_loss: Any = 0.0
    critic_loss: Any = 0.0
    reward: Any = 0.0
    best_reward: Any = -math.inf

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


@dataclass
class BPTTTrainingOutput(OptimizerTrainingOutPut):
    optimizer_state: BPTTState = None
    bptt_summary: BPTTAgentSummary = None


def inv_softplus(x: float) -> float:
    return math.log(math.exp(x) - 1.0) if x < 20.0 else x


def lecun_normal_flat(dims: Sequence[int], gen: torch.Generator) -> torch.Tensor:
    """[3P] flax nn.Dense defaults (network_utils.py:13-16): kernel = lecun_normal (truncated normal at +-2 sigma with
    stddev sqrt(1/fan_in)/0.87962566), bias = 0.  Flat layout per layer: W[in][out] then b[out]."""
    parts = []
    for i in range(len(dims) - 1):
        fi, fo = int(dims[i]), int(dims[i + 1])
        w = torch.empty(fi, fo, dtype=torch.float32)
        std = math.sqrt(1.0 / fi) / 0.87962566103423978
        torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
        parts += [w.reshape(-1), torch.zeros(fo)]
    return torch.cat(parts)


class BPTTOptimizer(BaseOptimizer):
    def __init__(self,
                 obs_dim: int,
                 action_dim: int,
                 horizon: int = 20,
                 num_samples_per_gradient_update: int = 10,
                 train_steps: int = 20,
                 normalize: bool = True,
                 action_normalize: bool = True,
                 actor_features: Sequence[int] = (64, 64, 64),
                 policy_activation: str = "swish",
                 critic_features: Sequence[int] = (64, 64, 64),
                 critic_activation: str = "swish",
                 init_stddev: float = 1.0,
                 lr_actor: float = 1e-3,
                 weight_decay_actor: float = 1e-5,
                 lr_critic: float = 1e-3,
                 weight_decay_critic: float = 1e-5,
                 reset_optimizer: bool = True,
                 target_soft_update_tau: float = 0.005,
                 rng: int = K.PRNGKey(0),
                 evaluation_samples: int = 100,
                 evaluation_horizon: int = 100,
                 evaluation_frequency: int = -1,
                 critic_updates_per_policy_update: int = 1,
                 discount: float = 0.99,
                 lambda_: float = 0.97,
                 loss_ent_coefficient: float = 0.005,
                 use_best_trained_policy: bool = False,
                 sample_simulated_transitions: bool = True,
                 sampling_buffer_size: int = 10_000_000,
                 device=None,
                 process_group=None,
                 use_graph: bool = True,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        _hip.load()                                   # fail loudly without the HIP library
        self.use_graph = bool(use_graph)              # replay the train step as a hipGraph (single rank; see train())
        # data-parallel ranks: every rank runs num_samples_per_gradient_update trajectories from its own sampling
        # buffer; actor and critic gradients and the normalisers' sums are summed over the ranks (the mean through grad_scale),
        # parameters start identical (rank-0 broadcast in init) and stay identical
        from mbpo.parallel import DataParallel
        self.dp = DataParallel(process_group)
        self._all_reduce = self.dp.all_reduce_fn()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.obs_dim, self.action_dim = int(obs_dim), int(action_dim)
        self.state_normalizer = Normalizer((self.obs_dim,))
        self.reward_normalizer = Normalizer((1,))
        # logical shapes and kernel shapes.  The fused BPTT kernels (k_bptt_actor, k_critic_fwd_bwd) are built for 64-wide hidden
        # layers; narrower ones are zero-padded (ops.py "hidden-width padding").  WIDER networks (the reference takes any feature
        # tuple, bptt_optimizer.py:183-186) train on the non-fused path: the horizon walked on the host with the networks as HIP
        # autograd nodes — forward mbpo_ensemble_mlp_forward / mbpo_mlp_layered_vjp, vector-Jacobian products layer by layer
        # (csrc/layered.hip) — and the built-in System's step in its differentiable torch form (systems/torch_steps.py).  The actor is
        # padded to a rollout-kernel width (it also runs inside the evaluation rollout and `act`: <= 256); the critics keep any sizes.
        self.actor_dims_logical = [self.obs_dim, *[int(f) for f in actor_features], 2 * self.action_dim]
        self.critic_dims_logical = [self.obs_dim, *[int(f) for f in critic_features], 1]
        self.wide = max([int(f) for f in (*actor_features, *critic_features)], default=0) > 64
        if self.wide:
            self.actor_width = ops.common_width(actor_features, supported=ops.ROLLOUT_WIDTHS, what="BPTT actor")
            cmax = max([int(f) for f in critic_features], default=0)
            self.critic_width = ops.common_width(critic_features, supported=ops.ROLLOUT_WIDTHS) if cmax <= 256 else None
        else:
            self.actor_width = self.critic_width = ops.common_width(actor_features, critic_features, supported=(64,), what="BPTT")
        self.kernel_width = self.actor_width
        self.actor_dims = ops.padded_dims(self.actor_dims_logical, self.actor_width)
        self.critic_dims = ops.padded_dims(self.critic_dims_logical, self.critic_width)
        self.policy_activation, self.critic_activation = policy_activation, critic_activation
        self.actor_spec = ops.MlpSpec(self.actor_dims, policy_activation, 1)
        self.critic_spec = ops.MlpSpec(self.critic_dims, critic_activation, 2)
        self.init_stddev = float(init_stddev)
        self.lr_actor, self.weight_decay_actor = lr_actor, weight_decay_actor
        self.lr_critic, self.weight_decay_critic = lr_critic, weight_decay_critic
        actor_rng, critic_rng, rng = K.split(rng, 3)
        init_state_rng, rng = K.split(rng, 2)

        self.horizon = int(horizon)
        self.num_samples_per_gradient_update = int(num_samples_per_gradient_update)
        self.sample_simulated_transitions = sample_simulated_transitions
        self.normalize = normalize
        self.action_normalize = action_normalize
        self.train_steps = int(train_steps)
        self.reset_optimizer = reset_optimizer
        self.evaluate_agent = evaluation_frequency > 0
        self.evaluation_samples = int(evaluation_samples)
        self.evaluation_horizon = int(evaluation_horizon)
        self.evaluation_frequency = evaluation_frequency
        self.discount, self.lambda_ = discount, lambda_
        self.tau = target_soft_update_tau
        self.use_best_trained_policy = use_best_trained_policy
        self.loss_ent_coefficient = loss_ent_coefficient
        self.critic_updates_per_policy_updates = int(critic_updates_per_policy_update)
        self.train_policy = lambda obs, opt_state: self.act(obs, opt_state, evaluate=False)

        X, U = self.obs_dim, self.action_dim
        self.row_len = 2 * X + U + 2                 # ravel_pytree(dummy_transition): obs, action, reward, discount, next_obs
        dev = self.device
        dummy_transition = Transition(observation=torch.zeros(X, device=dev), action=torch.zeros(U, device=dev),
                                      next_observation=torch.zeros(X, device=dev), reward=torch.zeros(1, device=dev),
                                      discount=torch.zeros(1, device=dev))
        self.sampling_buffer = UniformSamplingQueue(max_replay_size=int(sampling_buffer_size), dummy_data_sample=dummy_transition,
                                                    sample_batch_size=self.num_samples_per_gradient_update, device=dev)
        self._init_buff_key = rng
        self._sampling_data: Optional[torch.Tensor] = None     # [sampling_buffer_size, D] allocated on first train()

        n, H, Kc = self.num_samples_per_gradient_update, self.horizon, self.critic_updates_per_policy_updates
        self.num_transitions = n * H
        self.critic_batch = math.ceil(self.num_transitions / Kc)
        self._actor_grad = None if self.wide else ops.BpttActorGrad(
            x_dim=X, u_dim=U, horizon=H, actor_dims=self.actor_dims, critic_dims=self.critic_dims, n=n, device=dev,
            actor_activation=policy_activation, critic_activation=critic_activation, init_stddev=self.init_stddev, discount=discount,
            lambda_=lambda_, ent_coef=loss_ent_coefficient)
        self._actor_grad_generic = ops.BpttActorGradGeneric(
            x_dim=X, u_dim=U, horizon=H, actor_dims=self.actor_dims, critic_dims=self.critic_dims, n=n, device=dev,
            actor_activation=policy_activation, critic_activation=critic_activation, init_stddev=self.init_stddev, discount=discount,
            lambda_=lambda_, ent_coef=loss_ent_coefficient)        # the same contract for a user-defined System (non-fused)
        self._critic_grad = (ops.CriticGradGeneric if self.wide else ops.CriticGrad)(
            x_dim=X, critic_dims=self.critic_dims, batch=self.critic_batch, device=dev, activation=critic_activation)
        self.P, self.C2 = self.actor_spec.n_params, 2 * self.critic_spec.n_params
        # Multi-GPU (as SAC / PPO): the actor and critic gradients and the normalisers' sums go through the one-shot
        # peer-memory exchange over xGMI (csrc/p2p.hpp) instead of a library collective — plain kernels, so the train step stays
        # hipGraph-capturable.  create() validates the mapped regions against torch.distributed.all_reduce on every rank and
        # returns None on every rank (keep the library collective) if anything is off; MBPO_P2P_ALLREDUCE=0 forces that.
        self.p2p = None
        if self.dp.group is not None and self.dp.world_size > 1:
            from mbpo.parallel import P2PExchange
            self.p2p = P2PExchange.create(self.dp, max(self.P, self.C2, 3 + 2 * self.obs_dim), dev)
            if self.p2p is not None:
                self._all_reduce = self.p2p.all_reduce_sum
        self._actor_opt = ops.AdamW(self.P, dev, lr_actor, weight_decay_actor, apply_if_finite=True)
        self._critic_opt = ops.AdamW(self.C2, dev, lr_critic, weight_decay_critic, apply_if_finite=True)
        # scratch (fixed addresses: a train step can be captured in a hipGraph)
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self._init_rows = f(n, self.row_len)
        self._init_obs = f(n, X)
        self._critic_idx = torch.zeros(Kc * self.critic_batch, device=dev, dtype=torch.int32)
        self._critic_rows = f(Kc * self.critic_batch, self.row_len)
        self._traj_state = torch.tensor([self.num_transitions, 0, 0, self.num_transitions], device=dev, dtype=torch.int32)
        self._reward_ms = f(2)
        self._rng = ops.make_rng(dev)      # device RNG words {seed word = 0, train-step index}: the Philox offset of every draw
        self._stats_sums_x, self._stats_ws_x = f(1 + 2 * X), f(ops.stats_workspace_floats(X))
        self._stats_sums_r, self._stats_ws_r = f(3), f(ops.stats_workspace_floats(1))

    # -- reference API --------------------------------------------------------------------------------------------
    def init(self, key: int, true_buffer_state: Optional[ReplayBufferState] = None) -> BPTTState:
        assert self.system is not None, "BPTT optimizer requires system to be defined."
        assert self.system.x_dim == self.obs_dim and self.system.u_dim == self.action_dim, \
            "input and action dimensions do not match with the system"
        critic_key, actor_key, system_key, key = K.split(key, 4)
        dev = self.device
        gen = torch.Generator().manual_seed(int(critic_key) % (2 ** 63))
        emb = lambda flat, dims, width: flat if width is None else ops.embed_mlp_params(flat, dims, width)
        critic_params = torch.cat([emb(lecun_normal_flat(self.critic_dims_logical, gen), self.critic_dims_logical, self.critic_width),
                                   emb(lecun_normal_flat(self.critic_dims_logical, gen), self.critic_dims_logical, self.critic_width)]).to(dev)
        gen = torch.Generator().manual_seed(int(actor_key) % (2 ** 63))
        actor_params = emb(lecun_normal_flat(self.actor_dims_logical, gen), self.actor_dims_logical, self.actor_width).to(dev)
        self.dp.broadcast(actor_params)
        self.dp.broadcast(critic_params)
        z = lambda n: torch.zeros(n, device=dev, dtype=torch.float32)
        system_params = self.system.init_params(system_key)
        if true_buffer_state is None:
            dummy_buffer_key, key = K.split(key, 2)
            true_buffer_state = self.dummy_true_buffer_state(dummy_buffer_key)
        return BPTTState(
            true_buffer_state=true_buffer_state,
            system_params=system_params,
            actor_opt_state=AdamWState(z(self.P), z(self.P), z(1)),
            actor_params=actor_params,
            critic_opt_state=AdamWState(z(self.C2), z(self.C2), z(1)),
            critic_params=critic_params,
            target_critic_params=critic_params.clone(),
            state_normalizer_state=self.state_normalizer.initialize_normalizer_state(dev),
            reward_normalizer_state=self.reward_normalizer.initialize_normalizer_state(dev),
            key=key)

    def update_normalizers(self, transition: Transition, bptt_state: BPTTState) -> BPTTState:
        return bptt_state.replace(
            state_normalizer_state=self.state_normalizer.update(transition.observation, bptt_state.state_normalizer_state),
            reward_normalizer_state=self.reward_normalizer.update(transition.reward, bptt_state.reward_normalizer_state))

    def act(self, obs: torch.Tensor, opt_state: BPTTState, evaluate: bool = True, *args, **kwargs) -> Tuple[torch.Tensor, BPTTState]:
        """bptt_optimizer.py:305-325.  obs [x] or [N, x]; the MLP runs in mbpo_ensemble_mlp_forward, the squash is elementwise."""
        single = obs.dim() == 1
        x = obs.reshape(-1, self.obs_dim).to(self.device, torch.float32)
        ns = opt_state.state_normalizer_state
        xn = ((x - ns.mean) / ns.std).contiguous()
        out = ops.ensemble_mlp_forward(opt_state.actor_params, self.actor_spec, xn)[0]
        U = self.action_dim
        mu, raw = out[:, :U], out[:, U:]
        new_state = opt_state
        if evaluate:
            z = mu
        else:
            sample_key, key = K.split(opt_state.key, 2)
            new_state = opt_state.replace(key=key)
            sig = torch.clamp(torch.nn.functional.softplus(raw + inv_softplus(self.init_stddev)), 1e-6, 1e2)
            gen = torch.Generator(device=self.device).manual_seed(int(sample_key) % (2 ** 63))
            z = mu + torch.randn(mu.shape, device=self.device, generator=gen) * sig
        a = torch.clamp(torch.tanh(z), -0.999, 0.999)
        return (a[0] if single else a), new_state

    # -- one train step on the working buffers ----------------------------------------------------------------------
    def _system_kwargs(self, system_params):
        spec = self.system.rollout_spec(system_params, self.device)
        if spec["system_kind"] == _hip.SYS_GENERIC:
            # a user-defined System (the reference's plug-in seam, base_systems.py:40-52): rollout_policy's scan
            # (utils/optimizer_utils.py:62-116) is walked on the host — ops.BpttActorGradGeneric: the networks' forward and VJP in HIP
            # (mbpo_ensemble_mlp_forward / mbpo_mlp_vjp), lambda-return in HIP, the user's step differentiated by torch autograd
            return spec, dict(system=self.system, system_params=system_params)
        if spec["system_kind"] == _hip.SYS_ENSEMBLE and spec.get("ens_mode", _hip.ENS_MEAN) != _hip.ENS_MEAN:
            raise _hip.MbpoHipError("BPTT needs a differentiable model: EnsembleSystem mode must be 'mean'")
        if spec.get("ens_sample_noise", False):
            raise _hip.MbpoHipError("BPTT through sampled model noise is not supported (ens_sample_noise must be False)")
        kw = dict(system_kind=spec["system_kind"], reward_kind=spec["reward_kind"], reward_params=spec["reward_params"],
                  sys_params=spec.get("sys_params"), dyn_params=spec.get("dyn_params"), dyn_spec=spec.get("dyn_spec"),
                  ens_predict_delta=spec.get("ens_predict_delta", True))
        return spec, kw

    def _train_step(self, w: "_Work", buff: ReplayBufferState, seeds: Tuple[int, int, int]) -> ReplayBufferState:
        """Mutates the working state `w` in place; returns the sampling-buffer state.  Summary scalars land in w.summary[:, i]."""
        X, U = self.obs_dim, self.action_dim
        obs_seed, act_seed, critic_seed = seeds
        # initial states: sampling_buffer.sample -> .observation   (:455-458)
        ops.replay_sample(buff.data, buff.state, self.num_samples_per_gradient_update, seed=obs_seed, offset=0,
                          rng_dev=self._rng, out=self._init_rows)
        self._init_obs.copy_(self._init_rows[:, :X])
        # actor: value_and_grad(vmap(actor_loss).mean)   (:361-372)
        self._reward_ms[0:1].copy_(w.reward_norm.vec[1:2])
        self._reward_ms[1:2].copy_(w.reward_norm.vec[3:4])
        ag = self._actor_grad_generic if w.generic else self._actor_grad
        ag.desc.seed = act_seed
        ag(actor_params=w.actor_params, target_critic_params=w.target_critic_params, init_states=self._init_obs,
           state_mean=w.state_norm.mean, state_std=w.state_norm.std, reward_mean_std=self._reward_ms, offset=0,
           rng_dev=self._rng, **w.sys_kw)
        gs = 1.0 / self.dp.world_size
        if self._all_reduce is not None:
            self._all_reduce(ag.grads)
        self._actor_opt.step(w.actor_params, ag.grads, grad_scale=gs)
        # critic: K minibatches drawn with replacement from the n*H simulated transitions   (:380-419)
        ops.replay_sample(ag.transitions, self._traj_state, self._critic_idx.numel(), seed=critic_seed, offset=0,
                          rng_dev=self._rng, out=self._critic_rows, idx_out=self._critic_idx)
        B = self.critic_batch
        for k in range(self.critic_updates_per_policy_updates):
            cg = self._critic_grad(w.critic_params, ag.transitions, ag.lambda_values, self._critic_idx[k * B:(k + 1) * B],
                                   w.state_norm.mean, w.state_norm.std)
            if self._all_reduce is not None:
                self._all_reduce(cg)
            self._critic_opt.step(w.critic_params, cg, target=w.target_critic_params, tau=self.tau, grad_scale=gs)
        # summary of this step (actor_grad_norm, critic_grad_norm, actor_loss, critic_loss)
        w.step_summary[0:1].copy_(self._actor_opt.grad_norm)
        w.step_summary[1:2].copy_(self._critic_opt.grad_norm)
        w.step_summary[2:3].copy_(ag.metrics[0:1])
        w.step_summary[3:4].copy_(self._critic_grad.metrics)
        # normalisers on the simulated transitions (:475-476), then into the sampling buffer (:477-478)
        if self.normalize:
            ops.running_stats_update(ag.transitions, 0, X, w.state_norm.vec, all_reduce=self._all_reduce, sums=self._stats_sums_x,
                                     workspace=self._stats_ws_x, std_min=EPS, std_max=float("inf"))
            ops.running_stats_update(ag.transitions, X + U, 1, w.reward_norm.vec, all_reduce=self._all_reduce, sums=self._stats_sums_r,
                                     workspace=self._stats_ws_r, std_min=EPS, std_max=float("inf"))
        if self.sample_simulated_transitions:
            buff = self.sampling_buffer.insert_rows(buff, ag.transitions)
        ops.rng_advance(self._rng)
        return buff

    def _capturable(self) -> bool:
        """As SAC._capturable: a train step can be captured when it holds plain kernels only — single rank, the peer-memory
        exchange, or an RCCL process group; never a host-side (gloo) collective (a failed capture is not recoverable)."""
        if self.dp.group is None or self.p2p is not None:
            return True
        import os
        import torch.distributed as dist
        return dist.get_backend(self.dp.group) == "nccl" and os.environ.get("MBPO_GRAPH_NCCL", "1") != "0"

    def close(self) -> None:
        """Release the peer-memory regions."""
        if self.p2p is not None:
            self._all_reduce = self.dp.all_reduce_fn()
            self.p2p.close()
            self.p2p = None

    def _evaluate(self, w: "_Work", eval_obs: torch.Tensor) -> torch.Tensor:
        """evaluate_policy (:480-493): deterministic rollouts of evaluation_horizon steps from eval_obs; mean summed reward."""
        X, U = self.obs_dim, self.action_dim
        n = eval_obs.shape[0]
        obs = eval_obs.clone()
        zeros = torch.zeros(n, device=self.device)
        rows = ops.model_rollout(policy_params=w.actor_params, policy_spec=self.actor_spec, x_dim=X, u_dim=U, obs=obs,
                                 first_obs=obs.clone(), steps=zeros, done=zeros.clone(), n_steps=self.evaluation_horizon,
                                 episode_length=2 ** 30, norm_mean=w.state_norm.mean.contiguous(), norm_std=w.state_norm.std.contiguous(),
                                 deterministic=True, action_clip=0.999, seed=0, **w.rollout_spec)
        return rows[:, X + U].reshape(self.evaluation_horizon, n).sum(0).mean()

    def train(self, bptt_state: BPTTState = None, opt_state: BPTTState = None) -> BPTTTrainingOutput:
        """bptt_optimizer.py:439-538."""
        bptt_state = bptt_state if bptt_state is not None else opt_state
        assert self.system is not None, "BPTT optimizer requires system to be defined."
        buffer_state = bptt_state.true_buffer_state
        if buffer_state.data.shape[1] != self.row_len:
            raise ValueError(f"true buffer rows have {buffer_state.data.shape[1]} columns; BPTT transitions "
                             f"(obs, action, reward, discount, next_obs) need {self.row_len}")
        train_key, key = K.split(bptt_state.key, 2)
        eval_rng, train_key = K.split(train_key, 2)
        eval_obs = None
        if self.evaluate_agent:
            if buffer_state.insert_position <= buffer_state.sample_position:
                raise ValueError("evaluation needs a non-empty true buffer")
            eval_rows = ops.replay_sample(buffer_state.data, buffer_state.state, self.evaluation_samples, seed=eval_rng, offset=0)
            eval_obs = eval_rows[:, :self.obs_dim].contiguous()
        eval_sim_key, buffer_key, train_key = K.split(train_key, 3)

        # the whole true buffer (max_replay_size rows, unused ones included — :453-454) seeds the sampling buffer
        mx = self.sampling_buffer.max_replay_size
        if self._sampling_data is None:
            self._sampling_data = torch.zeros(mx, self.row_len, device=self.device, dtype=torch.float32)
        buff = ReplayBufferState(data=self._sampling_data, state=torch.zeros(4, device=self.device, dtype=torch.int32),
                                 key=self._init_buff_key)
        true_rows = buffer_state.data
        if buffer_state.head != 0:
            idx = (torch.arange(true_rows.shape[0], device=true_rows.device) + buffer_state.head) % true_rows.shape[0]
            true_rows = true_rows[idx]
        buff = self.sampling_buffer.insert_rows(buff, true_rows)

        w = _Work(self, bptt_state.replace(key=train_key))
        self._actor_opt.load_state(bptt_state.actor_opt_state.mu, bptt_state.actor_opt_state.nu, bptt_state.actor_opt_state.count)
        self._critic_opt.load_state(bptt_state.critic_opt_state.mu, bptt_state.critic_opt_state.nu, bptt_state.critic_opt_state.count)
        ops.set_rng(self._rng, 0, 0)
        # per-train() Philox seeds; the step counter is the Philox offset (the reference re-splits a key every step)
        seeds = tuple(self.dp.rank_key(k) for k in K.split(train_key, 4)[:3])    # ranks draw different states / noise / minibatches
        self._last_seeds = seeds                       # (initial-state sampling, action noise, critic minibatch) — for parity tests
        summaries = torch.zeros(self.train_steps, 6, device=self.device, dtype=torch.float32)
        prev_reward = torch.zeros((), device=self.device)
        best_reward = torch.full((), -math.inf, device=self.device)
        best: Optional[BPTTState] = None
        state_key = train_key
        # Every launch of a train step reads its Philox offset / Adam counts / buffer positions from device memory, so the step
        # is captured once into a hipGraph (after one eager step that also sizes every workspace) and replayed: ~25 launches
        # per step cost more host time than GPU time at the reference's sizes (n = 50, H = 20).  Under a process group the step
        # is captured too when its exchanges are kernels: the peer-memory exchange, or RCCL collectives (stream-ordered kernels);
        # a host-side (gloo) collective is never put inside a capture (_capturable).  Evaluation runs eagerly between replays.
        graph = None
        can_capture = self.use_graph and self._capturable() and self.train_steps >= 2 and not w.generic   # (generic: user code between the kernels)
        n_rows = self.num_transitions
        for i in range(self.train_steps):
            sampling_key, state_key = K.split(state_key, 2)
            critic_training_key, state_key = K.split(state_key, 2)
            if can_capture and i >= 1:
                if graph is None:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self._train_step(w, buff, seeds)
                graph.replay()
                if self.sample_simulated_transitions:
                    buff = self.sampling_buffer.insert_mirror(buff, n_rows)
            else:
                buff = self._train_step(w, buff, seeds)
            if self.evaluate_agent:
                if i % self.evaluation_frequency == 0 or i == self.train_steps - 1:
                    reward = self._evaluate(w, eval_obs)
                    if best is None or bool(reward > best_reward):
                        best_reward, best = reward, w.snapshot(self, state_key)
                else:
                    reward = prev_reward
            else:
                reward = prev_reward
                best_reward, best = reward, None        # best == newest state (:515-516)
            summaries[i, :4].copy_(w.step_summary)
            summaries[i, 4] = reward
            summaries[i, 5] = best_reward
            prev_reward = reward
        if self.p2p is not None and self.p2p.status() != 0:
            raise _hip.MbpoHipError("BPTT: the peer-memory gradient exchange timed out on this rank; "
                                    "set MBPO_P2P_ALLREDUCE=0 to use the RCCL all-reduce")
        self._last_train_captured = graph is not None          # (for tests / diagnostics)
        final = w.snapshot(self, state_key)
        if self.use_best_trained_policy and best is not None:
            trained_state = best.replace(system_params=final.system_params)
        else:
            trained_state = final
        summary = BPTTAgentSummary(actor_grad_norm=summaries[:, 0], critic_grad_norm=summaries[:, 1], actor_loss=summaries[:, 2],
                                   critic_loss=summaries[:, 3], reward=summaries[:, 4], best_reward=summaries[:, 5])
        return BPTTTrainingOutput(optimizer_state=trained_state, bptt_summary=summary)


class _Work:
    """Mutable working copy of a BPTTState for one train() call (the reference's scan carry)."""

    def __init__(self, opt: BPTTOptimizer, st: BPTTState):
        self.base = st
        self.actor_params = st.actor_params.clone()
        self.critic_params = st.critic_params.clone()
        self.target_critic_params = st.target_critic_params.clone()
        self.state_norm = st.state_normalizer_state.clone()
        self.reward_norm = st.reward_normalizer_state.clone()
        self.rollout_spec, self.sys_kw = opt._system_kwargs(st.system_params)
        self.generic = self.rollout_spec["system_kind"] == _hip.SYS_GENERIC or opt.wide
        if opt.wide and self.rollout_spec["system_kind"] != _hip.SYS_GENERIC:
            # wider networks than the fused BPTT kernel takes: the built-in System as a node of the torch autograd graph
            from mbpo.systems.torch_steps import DifferentiableBuiltin
            self.sys_kw = dict(system=DifferentiableBuiltin(opt.system, self.rollout_spec), system_params=st.system_params)
        self.step_summary = torch.zeros(4, device=opt.device, dtype=torch.float32)

    def snapshot(self, opt: BPTTOptimizer, key: int) -> BPTTState:
        ao, co = opt._actor_opt, opt._critic_opt
        return self.base.replace(
            actor_params=self.actor_params.clone(), critic_params=self.critic_params.clone(),
            target_critic_params=self.target_critic_params.clone(),
            actor_opt_state=AdamWState(ao.m.clone(), ao.v.clone(), ao.count.clone()),
            critic_opt_state=AdamWState(co.m.clone(), co.v.clone(), co.count.clone()),
            state_normalizer_state=self.state_norm.clone(), reward_normalizer_state=self.reward_norm.clone(), key=key)

"""SAC trainer — host-side mirror of mbpo/optimizers/policy_optimizers/sac/sac.py (same constructor arguments,
derived quantities, method names, call order and metric keys); every numeric step runs in libmbpo_hip.so.

Where the reference has                      this file issues
  lax.scan of actor_step (:283-296)      ->  ONE mbpo_model_rollout launch per get_experience
  running_statistics.update (:298-301)   ->  mbpo_running_stats_reduce x2 + _apply (two-pass, psum positions kept)
  replay_buffer.insert / .sample (:303,318) -> mbpo_replay_insert / mbpo_replay_sample (device-resident positions)
  lax.scan of sgd_step (:324)            ->  G x (mbpo_sac_grads [+ all-reduce] + mbpo_sac_apply)
  jit(training_epoch) (:347-361)         ->  one captured hipGraph of a training_step, replayed per step

Randomness: the reference splits a threefry key per call (sac.py:289,309-311,354-356).  Here every draw of a training_step
is Philox(seed word, (call-site id << 32) + step index, stream, element) with the seed word (the epoch / prefill key) and
the step index in two DEVICE words (`self._rng`, include/mbpo_hip.h "randomness"): the captured hipGraph bakes only the
call-site constants, so a replayed step draws bit for bit what the same step draws when issued eagerly.
"""
from __future__ import annotations

import dataclasses
import math
import os
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import torch

from mbpo import _hip, ops
from mbpo.optimizers.policy_optimizers.brax_utils.base import State
from mbpo.parallel import DataParallel
from mbpo.replay import ReplayBufferState, UniformSamplingQueue
from mbpo.systems.brax_wrapper import BraxWrapper
from mbpo.systems.ensemble_system import lecun_uniform_flat
from mbpo.types import Transition
from mbpo.utils import keys as K

Metrics = Dict[str, Any]

# Philox call-site ids (high 32 bits of the offset; the low 32 bits count training steps on the device)
SITE_ROLLOUT, SITE_SAMPLE, SITE_SGD = 1, 2, 16


@dataclass
class RunningStatisticsState:
    """[3P] brax running_statistics.RunningStatisticsState as views into one device vector [count, mean, sv, std]."""
    vec: torch.Tensor
    x_dim: int

    @property
    def count(self):
        return self.vec[0]

    @property
    def mean(self):
        return self.vec[1:1 + self.x_dim]

    @property
    def summed_variance(self):
        return self.vec[1 + self.x_dim:1 + 2 * self.x_dim]

    @property
    def std(self):
        return self.vec[1 + 2 * self.x_dim:1 + 3 * self.x_dim]


@dataclass
class TrainingState:
    """sac.py:39-54.  The tensors are views into the updater's flat device state (include/mbpo_hip.h)."""
    policy_optimizer_state: Any
    policy_params: torch.Tensor
    q_optimizer_state: Any
    q_params: torch.Tensor
    target_q_params: torch.Tensor
    gradient_steps: torch.Tensor
    env_steps: int
    alpha_optimizer_state: Any
    alpha_params: torch.Tensor
    normalizer_params: RunningStatisticsState

    def get_policy_params(self):
        return self.normalizer_params, self.policy_params

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class SAC:
    def __init__(self,
                 environment: BraxWrapper,
                 num_timesteps: int,
                 episode_length: int,
                 action_repeat: int = 1,
                 num_env_steps_between_updates: int = 2,
                 num_envs: int = 1,
                 num_eval_envs: int = 128,
                 lr_alpha: float = 1e-4,
                 lr_policy: float = 1e-4,
                 lr_q: float = 1e-4,
                 wd_alpha: float = 0.,
                 wd_policy: float = 0.,
                 wd_q: float = 0.,
                 max_grad_norm: float = 1e5,
                 discounting: float = 0.9,
                 batch_size: int = 256,
                 num_evals: int = 1,
                 normalize_observations: bool = False,
                 reward_scaling: float = 1.,
                 tau: float = 0.005,
                 min_replay_size: int = 0,
                 max_replay_size: Optional[int] = None,
                 grad_updates_per_step: int = 1,
                 deterministic_eval: bool = True,
                 init_log_alpha: float = 0.,
                 target_entropy: Optional[float] = None,
                 policy_hidden_layer_sizes: Sequence[int] = (64, 64, 64),
                 policy_activation: str = "swish",
                 critic_hidden_layer_sizes: Sequence[int] = (64, 64, 64),
                 critic_activation: str = "swish",
                 wandb_logging: bool = False,
                 return_best_model: bool = False,
                 eval_environment: Optional[BraxWrapper] = None,
                 episode_length_eval: Optional[int] = None,
                 eval_key_fixed: bool = False,
                 non_equidistant_time: bool = False,
                 continuous_discounting: float = 0,
                 min_time_between_switches: float = 0,
                 max_time_between_switches: float = 0,
                 env_dt: float = 0,
                 # --- MI355X-side knobs (not in the reference) ---
                 use_graph: bool = True,
                 process_group=None,
                 ):
        if min_replay_size >= num_timesteps:
            raise ValueError('No training will happen because min_replay_size >= num_timesteps')     # sac.py:100-102
        if non_equidistant_time and not env_dt > 0:
            raise ValueError("non_equidistant_time needs env_dt > 0 (sac/losses.py:95 floors the switch time to multiples of it)")
        if wandb_logging:
            raise NotImplementedError("wandb is not available in this environment")
        self.eval_key_fixed = eval_key_fixed
        self.return_best_model = return_best_model
        self.target_entropy = target_entropy
        self.init_log_alpha = init_log_alpha
        self.min_replay_size = min_replay_size
        self.num_timesteps = num_timesteps
        self.num_envs = num_envs
        self.deterministic_eval = deterministic_eval
        self.num_eval_envs = num_eval_envs
        self.episode_length = episode_length
        self.action_repeat = action_repeat
        self.num_evals = num_evals
        self.num_env_steps_between_updates = num_env_steps_between_updates
        self.normalize_observations = normalize_observations
        self.batch_size = batch_size
        if max_replay_size is None:
            max_replay_size = num_timesteps
        self.max_replay_size = max_replay_size
        # sac.py:119-134 — identical derived quantities
        self.env_steps_per_actor_step = action_repeat * num_envs
        self.num_prefill_actor_steps = math.ceil(min_replay_size / num_envs)
        num_prefill_env_steps = self.num_prefill_actor_steps * self.env_steps_per_actor_step
        assert num_timesteps - num_prefill_env_steps >= 0
        self.num_evals_after_init = max(num_evals - 1, 1)
        num_env_steps_in_one_train_step = self.num_evals_after_init * self.env_steps_per_actor_step
        num_env_steps_in_one_train_step *= num_env_steps_between_updates
        self.num_training_steps_per_epoch = math.ceil(
            (num_timesteps - num_prefill_env_steps) / num_env_steps_in_one_train_step)
        self.grad_updates_per_step = grad_updates_per_step
        self.tau = tau
        self.env = environment
        if episode_length_eval is None:
            episode_length_eval = episode_length
        self.episode_length_eval = episode_length_eval
        self.eval_env = environment if eval_environment is None else eval_environment
        self.x_dim = self.env.observation_size
        self.u_dim = self.env.action_size
        self.device = torch.device("cuda", torch.cuda.current_device())
        # logical shapes (what the user asked for) and kernel shapes (hidden layers zero-padded to one supported width: ops.py
        # "hidden-width padding"; a learned ensemble inside the fused rollout must share that width)
        self.policy_dims_logical = [self.x_dim, *policy_hidden_layer_sizes, 2 * self.u_dim]
        self.q_dims_logical = [self.x_dim + self.u_dim, *critic_hidden_layer_sizes, 1]
        dyn_hidden = list(getattr(getattr(self.env.system, "dynamics", None), "dims", [])[1:-1]) if self.env.system.fused else []
        widest = max([int(h) for h in (*policy_hidden_layer_sizes, *critic_hidden_layer_sizes, *dyn_hidden)], default=0)
        if widest <= ops.KERNEL_WIDTHS[-1]:
            self.kernel_width = ops.common_width(policy_hidden_layer_sizes, critic_hidden_layer_sizes, dyn_hidden, what="SAC")
            self.q_width = self.kernel_width
        else:
            # wider than the fused update kernel takes (sac.py:84-88 accepts any tuple): mbpo_sac_step then runs its forward/backward
            # half layer by layer (csrc/sac_layered.hip) — the critics keep their logical sizes; the policy, which also runs inside
            # the rollout / act kernels, is padded to one of their widths
            self.kernel_width = ops.common_width(policy_hidden_layer_sizes, dyn_hidden, supported=ops.ROLLOUT_WIDTHS, what="SAC policy")
            self.q_width = None
        if dyn_hidden and any(h != self.kernel_width for h in dyn_hidden):
            raise _hip.MbpoHipError(f"SAC: the learned ensemble's hidden width {dyn_hidden} must equal the policy's kernel width "
                                    f"{self.kernel_width} inside the fused rollout (build the EnsembleDynamics with that width)")
        self.policy_dims = ops.padded_dims(self.policy_dims_logical, self.kernel_width)
        self.q_dims = ops.padded_dims(self.q_dims_logical, self.q_width)
        self.policy_spec = ops.MlpSpec(self.policy_dims, policy_activation, 1)
        # data-parallel ranks (the live form of _PMAP_AXIS_NAME, sac.py:188-189): one process per GPU
        self.process_group = process_group
        self.dp = DataParallel(process_group)
        self.world_size = self.dp.world_size
        all_reduce = self.dp.all_reduce_fn()
        self._all_reduce = all_reduce
        self.updater = ops.SacUpdater(
            x_dim=self.x_dim, u_dim=self.u_dim, policy_dims=self.policy_dims, q_dims=self.q_dims, batch_size=batch_size,
            device=self.device, policy_activation=policy_activation, q_activation=critic_activation,
            discounting=discounting, reward_scaling=reward_scaling, target_entropy=target_entropy, tau=tau,
            lr_policy=lr_policy, lr_q=lr_q, lr_alpha=lr_alpha, wd_policy=wd_policy, wd_q=wd_q, wd_alpha=wd_alpha,
            max_grad_norm=max_grad_norm, all_reduce=all_reduce, world_size=self.world_size,
            non_equidistant_time=non_equidistant_time, continuous_discounting=continuous_discounting,
            min_time_between_switches=min_time_between_switches, max_time_between_switches=max_time_between_switches, env_dt=env_dt)
        # Multi-GPU: exchange the flat gradient (and the normaliser's sums) through peer memory over xGMI instead of a
        # library collective (csrc/p2p.hpp) — plain kernels, graph-capturable.  create() validates the regions against
        # torch.distributed.all_reduce on every rank and returns None (keep RCCL) if anything is off.
        self.p2p = None
        if self.dp.group is not None and self.world_size > 1:
            from mbpo.parallel import P2PExchange
            self.p2p = P2PExchange.create(self.dp, self.updater.NP, self.device)
            if self.p2p is not None:
                self.updater.p2p = self.p2p
                self._all_reduce = self.p2p.all_reduce_sum
        # SAC's own buffer of model transitions (sac.py:191-205): rows carry state_extras.truncation
        z = lambda n: torch.zeros(n, device=self.device)
        dummy_transition = Transition(observation=z(self.x_dim), action=z(self.u_dim), reward=z(1), discount=z(1),
                                      next_observation=z(self.x_dim),
                                      extras={'state_extras': {'truncation': z(1)}, 'policy_extras': {}})
        self.replay_buffer = UniformSamplingQueue(max_replay_size=max_replay_size, dummy_data_sample=dummy_transition,
                                                  sample_batch_size=batch_size * grad_updates_per_step, device=self.device)
        self.row_len = self.replay_buffer.row_len
        # fixed device buffers (graph-replayable)
        S, N = num_env_steps_between_updates, num_envs
        self._rollout_rows = torch.empty(S * N, self.row_len, device=self.device)
        self._batch_rows = torch.empty(batch_size * grad_updates_per_step, self.row_len, device=self.device)
        self._stats_sums = torch.zeros(1 + 2 * self.x_dim, device=self.device)
        self._stats_ws = torch.empty(ops.stats_workspace_floats(self.x_dim), device=self.device)
        self._stats_vec = torch.zeros(1 + 3 * self.x_dim, device=self.device)
        self.use_graph = use_graph
        self._graph = None
        self._graph_key = None
        self._clip_events_seen = 0                 # _adapt_step_mode
        self._graph_refs = None
        self._rng = ops.make_rng(self.device)      # device uint64[2]: {key of the running epoch / prefill, step index}
        self._eval_calls = 0

    # ------------------------------------------------------------------------------------------------ policy
    def make_policy(self, params, deterministic: bool = False):
        """make_inference_fn (sac_networks.py:58-73): policy(observations, key) -> (action, extras)."""
        normalizer_params, policy_params = params
        nm, ns = self._norm(normalizer_params)

        def policy(observations: torch.Tensor, key_sample: int):
            obs = observations.reshape(-1, self.x_dim).to(self.device, torch.float32).contiguous()
            act = policy_act(policy_params, self.policy_spec, obs, nm, ns, deterministic, key_sample)
            return (act[0] if observations.dim() == 1 else act), {}

        return policy

    def _norm(self, normalizer_params: RunningStatisticsState):
        if not self.normalize_observations:
            return None, None
        return normalizer_params.mean.contiguous(), normalizer_params.std.contiguous()

    # ------------------------------------------------------------------------------------------------ state
    def init_training_state(self, key: int) -> TrainingState:
        """sac.py:376-402: fresh lecun-uniform policy and twin critics, target = critics, log_alpha, normalizer init."""
        key_policy, key_q = K.split(key)
        gp = torch.Generator().manual_seed(key_policy % (2 ** 63))
        gq = torch.Generator().manual_seed(key_q % (2 ** 63))
        # lecun-uniform at the LOGICAL fan-ins, then embedded into the (possibly wider) kernel shape
        pol = ops.embed_mlp_params(lecun_uniform_flat(self.policy_dims_logical, gp), self.policy_dims_logical, self.kernel_width)
        q = torch.cat([ops.embed_mlp_params(lecun_uniform_flat(self.q_dims_logical, gq), self.q_dims_logical, self.q_width)
                       for _ in range(2)])
        params = torch.cat([pol, q, torch.tensor([self.init_log_alpha], dtype=torch.float32)]).to(self.device)
        self.dp.broadcast(params, src=0)     # identical replicas: rank 0's initialisation everywhere
        self.updater.load_state(params)
        self._stats_vec.zero_()
        self._stats_vec[1 + 2 * self.x_dim:] = 1.0      # running_statistics.init_state: std = 1
        return self._training_state(env_steps=0)

    def _training_state(self, env_steps: int) -> TrainingState:
        u = self.updater
        return TrainingState(policy_optimizer_state=(u.adam_m[:u.P], u.adam_v[:u.P]), policy_params=u.policy_params,
                             q_optimizer_state=(u.adam_m[u.P:u.P + 2 * u.Q], u.adam_v[u.P:u.P + 2 * u.Q]),
                             q_params=u.q_params, target_q_params=u.target_q, gradient_steps=u.step_count,
                             env_steps=env_steps, alpha_optimizer_state=(u.adam_m[-1:], u.adam_v[-1:]),
                             alpha_params=u.log_alpha, normalizer_params=RunningStatisticsState(self._stats_vec, self.x_dim))

    # ------------------------------------------------------------------------------------------------ hot loops
    def rekey(self, key: int) -> None:
        """Start a fresh random stream: seed word <- key, step index <- 0 (host -> device write; never inside a capture)."""
        ops.set_rng(self._rng, K.PRNGKey(key), 0)

    def get_experience(self, normalizer_params: RunningStatisticsState, policy_params: torch.Tensor, env_state: State,
                       buffer_state: ReplayBufferState, key: Optional[int] = None):
        """sac.py:283-304.  `key` given: re-key the device stream first; None: continue it (the step index is advanced by
        the caller, once per training / prefill step)."""
        if key is not None:
            self.rekey(key)
        nm, ns = self._norm(normalizer_params)
        spec = self.env.system.rollout_spec(env_state.system_params, self.device)
        sp_out: list = []
        rows = ops.model_rollout(policy_params=policy_params, policy_spec=self.policy_spec, x_dim=self.x_dim, u_dim=self.u_dim,
                                 obs=env_state.obs, first_obs=env_state.info['first_obs'], steps=env_state.info['steps'],
                                 done=env_state.done, n_steps=self.num_env_steps_between_updates,
                                 episode_length=self.episode_length, action_repeat=self.action_repeat, norm_mean=nm,
                                 norm_std=ns, seed=0, offset=SITE_ROLLOUT << 32, rng_dev=self._rng,
                                 out=self._rollout_rows, system_params_out=sp_out, **spec)
        if sp_out:      # a user-defined System returns its (possibly updated) parameters: carried in the env State, as upstream
            env_state = env_state.replace(system_params=sp_out[-1])
        # running_statistics.update(normalizer_params, transitions.observation, pmap_axis_name)   (:298-301)
        ops.running_stats_update(rows, 0, self.x_dim, normalizer_params.vec, all_reduce=self._all_reduce,
                                 sums=self._stats_sums, workspace=self._stats_ws)
        buffer_state = self.replay_buffer.insert_rows(buffer_state, rows)                       # :303
        return normalizer_params, env_state, buffer_state

    def sgd_step(self, transitions_rows: torch.Tensor, normalizer_params: RunningStatisticsState, key: Optional[int] = None,
                 g: int = 0, defer_clip_check: bool = False) -> None:
        """sac.py:227-281 on one minibatch [B, D] (alpha, critic, actor updates at the old params + Polyak); `g` is the
        position inside training_step's scan (:324) — the reference hands every sgd_step its own split of the key."""
        if key is not None:
            self.rekey(key)
        nm, ns = self._norm(normalizer_params)
        self.updater.sgd_step(transitions_rows, nm, ns, seed=0, offset=(SITE_SGD + g) << 32, rng_dev=self._rng,
                              defer_clip_check=defer_clip_check)

    def training_step(self, training_state: TrainingState, env_state: State, buffer_state: ReplayBufferState,
                      key: Optional[int] = None):
        """sac.py:306-327.  `key` given: re-key the device stream (eager use); None: the next step of the running stream —
        what training_epoch issues, eagerly or as a hipGraph replay, with identical results."""
        if key is not None:
            self.rekey(key)
        normalizer_params, env_state, buffer_state = self.get_experience(
            training_state.normalizer_params, training_state.policy_params, env_state, buffer_state)
        training_state = training_state.replace(
            env_steps=training_state.env_steps + self.env_steps_per_actor_step * self.num_env_steps_between_updates)
        buffer_state, rows = self.replay_buffer.sample_rows(buffer_state, out=self._batch_rows, seed=0,
                                                            offset=SITE_SAMPLE << 32, rng_dev=self._rng)   # :318
        B = self.batch_size
        for g in range(self.grad_updates_per_step):                                                # scan :324
            # each step's clip check is resolved by the next step's first launch; the last one by finalize (ops.SacUpdater)
            self.sgd_step(rows[g * B:(g + 1) * B], normalizer_params, g=g, defer_clip_check=True)
        self.updater.finalize(rng_dev=self._rng)          # + the device RNG's step counter moves on (one launch)
        return training_state, env_state, buffer_state

    def prefill_replay_buffer(self, training_state: TrainingState, env_state: State, buffer_state: ReplayBufferState, key: int):
        """sac.py:329-345."""
        key, new_key = K.split(key)
        self.rekey(key)
        for _ in range(self.num_prefill_actor_steps):
            _, env_state, buffer_state = self.get_experience(training_state.normalizer_params, training_state.policy_params,
                                                             env_state, buffer_state)
            ops.rng_advance(self._rng)
            training_state = training_state.replace(env_steps=training_state.env_steps + self.env_steps_per_actor_step)
        return training_state, env_state, buffer_state, new_key

    def training_epoch(self, training_state: TrainingState, env_state: State, buffer_state: ReplayBufferState, key: int):
        """sac.py:347-361: num_training_steps_per_epoch training_steps; metrics averaged over the epoch.

        With use_graph the first step ever runs eagerly (it also warms every kernel up) and is captured into a hipGraph; all
        later steps, in this and in every following epoch, replay it — all launches read their positions / counters / RNG words
        from device memory."""
        self.updater.metrics_accum.zero_()
        n = self.num_training_steps_per_epoch
        env_steps_per = self.env_steps_per_actor_step * self.num_env_steps_between_updates
        done_steps = 0
        self.rekey(key)
        if self.use_graph and self._capturable():
            # The first training_step EVER issued against these buffers runs eagerly (it also warms every kernel up) and is then
            # captured; every later step — of this epoch and of all following ones, whatever num_training_steps_per_epoch is: the
            # reference's own acceptance runs have 1 or 2 — is a replay.
            gkey, refs = self._graph_signature(training_state, env_state, buffer_state)
            if self._graph is None or self._graph_key != gkey:
                training_state, env_state, buffer_state = self.training_step(training_state, env_state, buffer_state)
                done_steps = 1
                gkey, refs = self._graph_signature(training_state, env_state, buffer_state)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.training_step(training_state, env_state, buffer_state)
                # the graph holds raw device pointers: keep every tensor it was captured against alive with it
                self._graph, self._graph_key, self._graph_refs = graph, gkey, refs
            # capture does not execute: every step from here on is a replay
            replays = n - done_steps
            for _ in range(replays):
                self._graph.replay()
            done_steps = n
            # host mirrors of the replay positions (same integer arithmetic as the device)
            for _ in range(replays):
                buffer_state = _advance_mirror(self.replay_buffer, buffer_state, self._rollout_rows.shape[0])
            training_state = training_state.replace(env_steps=training_state.env_steps + replays * env_steps_per)
        while done_steps < n:
            training_state, env_state, buffer_state = self.training_step(training_state, env_state, buffer_state)
            done_steps += 1
        acc = self.updater.metrics_accum.cpu()
        self._adapt_step_mode(n * self.grad_updates_per_step)
        if self.p2p is not None and self.p2p.status() != 0:
            # a rank never arrived within the bounded wait (csrc/p2p.hpp): its gradients were poisoned with NaN, not skipped
            raise _hip.MbpoHipError("SAC: the peer-memory gradient exchange timed out on this rank; "
                                    "set MBPO_P2P_ALLREDUCE=0 to use the RCCL all-reduce")
        cnt = max(float(acc[4]), 1.0)
        metrics = {'critic_loss': float(acc[0]) / cnt, 'actor_loss': float(acc[1]) / cnt, 'alpha_loss': float(acc[2]) / cnt,
                   'alpha': float(acc[3]) / cnt, 'buffer_current_size': float(self.replay_buffer.size(buffer_state))}
        return training_state, env_state, buffer_state, metrics

    def _adapt_step_mode(self, steps: int) -> None:
        """Between epochs: pick the sgd_step flavour from how often clip_by_global_norm (sac.py:218-225) actually scaled a
        gradient.  The two-launch step applies the optimizer step unclipped and lets the next launch repair it — free when
        nothing clips (the reference default max_grad_norm = 1e5), a fix-up and a second pass over the phases when
        something does (~95 vs ~36 us per update at B = 256).  Both flavours give bit-identical parameters
        (tests/test_gpu_sac.py), every rank sees the same all-reduced gradient and so takes the same decision, and the
        hipGraph is keyed on the flavour.  An explicit choice (SacUpdater(two_launch=...), MBPO_SAC_TWO_LAUNCH) is left alone."""
        up = self.updater
        if up.two_launch_explicit or (up.all_reduce is not None and up.p2p is None):
            return
        events = up.clip_events()
        rate = (events - self._clip_events_seen) / max(int(steps), 1)
        self._clip_events_seen = events
        if up.two_launch and rate > 0.05:
            up.set_two_launch(False)
        elif not up.two_launch and rate < 0.01:
            up.set_two_launch(True)

    def _capturable(self) -> bool:
        """A training_step can be captured when it holds plain kernels only: single rank, the peer-memory exchange, or an
        RCCL process group (RCCL collectives are stream-ordered kernels and capture; a gloo collective is host code and
        would invalidate the capture — never attempted)."""
        if not self.env.system.fused:      # user code runs between the kernels: it may synchronise or allocate
            return False
        if self.dp.group is None or self.p2p is not None:
            return True
        import torch.distributed as dist
        return dist.get_backend(self.dp.group) == "nccl" and os.environ.get("MBPO_GRAPH_NCCL", "1") != "0"

    def _graph_signature(self, training_state: TrainingState, env_state: State, buffer_state: ReplayBufferState):
        """Every device address a captured training_step bakes in (`id()` of two tensors can be recycled) — the
        environment, replay and scratch buffers AND the train state the step reads and writes (a restored or cloned
        TrainingState must re-capture, not replay a graph that updates other tensors)."""
        spec = self.env.system.rollout_spec(env_state.system_params, self.device)
        u = self.updater
        tensors = [env_state.obs, env_state.info['first_obs'], env_state.info['steps'], env_state.done, buffer_state.data,
                   buffer_state.state, self._rollout_rows, self._batch_rows, self._stats_vec, self._rng,
                   training_state.policy_params, training_state.normalizer_params.vec, u.params, u.target_q, u.adam_m, u.adam_v,
                   u.step_count, u.workspace]
        tensors += [v for v in spec.values() if isinstance(v, torch.Tensor)]
        return tuple(t.data_ptr() for t in tensors) + (bool(self.updater.two_launch),), tensors

    def close(self) -> None:
        """Release the graph and the peer-memory regions (one P2PExchange per trainer: BraxOptimizer.train builds a trainer
        per call, brax_optimizers.py:95)."""
        self._graph = self._graph_key = self._graph_refs = None
        if self.p2p is not None:
            self.updater.p2p = None
            self.p2p.close()
            self.p2p = None

    def training_epoch_with_timing(self, training_state, env_state, buffer_state, key):
        """sac.py:363-374 (note: like the reference, `sps` omits the num_env_steps_between_updates factor)."""
        torch.cuda.synchronize()
        t = time.time()
        training_state, env_state, buffer_state, metrics = self.training_epoch(training_state, env_state, buffer_state, key)
        torch.cuda.synchronize()
        epoch_training_time = time.time() - t
        sps = (self.env_steps_per_actor_step * self.num_training_steps_per_epoch) / epoch_training_time
        metrics = {'training/sps': sps, **{f'training/{name}': value for name, value in metrics.items()}}
        return training_state, env_state, buffer_state, metrics

    # ------------------------------------------------------------------------------------------------ driver
    def reset_envs(self, env: BraxWrapper, key: int, n: int) -> State:
        return env.reset(K.split(key, n))

    def run_training(self, key: int, progress_fn: Callable[[int, Metrics], None] = lambda *args: None):
        """sac.py:404-494 — same order of key splits and phases."""
        key, subkey = K.split(key)
        training_state = self.init_training_state(subkey)
        key, rb_key, env_key, eval_key = K.split(key, 4)
        # the init key above is shared (rank 0's parameters are broadcast); everything that generates DATA is per rank, or
        # N ranks would roll out the same envs with the same noise and all-reduce N copies of one gradient
        rk = self.dp.rank_key
        env_state = self.reset_envs(self.env, rk(env_key), self.num_envs)
        buffer_state = self.replay_buffer.init(rk(rb_key))
        evaluator = Evaluator(self, self.eval_env, num_eval_envs=self.num_eval_envs, episode_length=self.episode_length_eval,
                              action_repeat=self.action_repeat, key=eval_key)
        all_metrics: List[Metrics] = []
        highest_eval_episode_reward = -float('inf')
        best_params = self._snapshot(training_state)
        if self.num_evals > 1:
            metrics = evaluator.run_evaluation(training_state.get_policy_params(), training_metrics={})
            if metrics['eval/episode_reward'] > highest_eval_episode_reward:
                highest_eval_episode_reward = metrics['eval/episode_reward']
                best_params = self._snapshot(training_state)
            all_metrics.append(metrics)
            progress_fn(0, metrics)
        key, prefill_key = K.split(key)
        training_state, env_state, buffer_state, _ = self.prefill_replay_buffer(training_state, env_state, buffer_state,
                                                                                 rk(prefill_key))
        if self.eval_key_fixed:
            key, eval_key = K.split(key)
        for _ in range(self.num_evals_after_init):
            key, epoch_key = K.split(key)
            training_state, env_state, buffer_state, training_metrics = self.training_epoch_with_timing(
                training_state, env_state, buffer_state, rk(epoch_key))
            if not self.eval_key_fixed:
                key, eval_key = K.split(key)
            metrics = evaluator.run_evaluation(training_state.get_policy_params(), training_metrics, unroll_key=eval_key)
            if metrics['eval/episode_reward'] > highest_eval_episode_reward:
                highest_eval_episode_reward = metrics['eval/episode_reward']
                best_params = self._snapshot(training_state)
            all_metrics.append(metrics)
            progress_fn(training_state.env_steps, metrics)
        last_params = self._snapshot(training_state)
        params_to_return = best_params if self.return_best_model else last_params
        return params_to_return, all_metrics

    def _snapshot(self, training_state: TrainingState):
        """(normalizer_params, policy_params) detached from the live flat state."""
        return (RunningStatisticsState(training_state.normalizer_params.vec.clone(), self.x_dim),
                training_state.policy_params.clone())


def _advance_mirror(q: UniformSamplingQueue, bs: ReplayBufferState, n: int) -> ReplayBufferState:
    mx = q.max_replay_size
    roll = min(0, mx - bs.insert_position - n)
    pos = bs.insert_position + roll
    k, _ = K.split(bs.key)
    return bs.replace(insert_position=(pos + n) % (mx + 1), sample_position=max(0, bs.sample_position + roll),
                      head=(bs.head - roll) % mx, key=k, sample_count=bs.sample_count + 1)


def policy_act(policy_params: torch.Tensor, policy_spec: ops.MlpSpec, obs: torch.Tensor, norm_mean, norm_std,
               deterministic: bool, key: int) -> torch.Tensor:
    """Policy inference for `act` (sac_networks.py:58-73 / ppo_network.py:59-84): normalise -> MLP -> NormalTanh mode / sample, ONE
    launch of mbpo_policy_act — the kernel and Philox stream the trainers' rollouts use (stream POLICY_NOISE, seed = the key split
    off for this call, element = row * u + column), so inference and training sample one and the same distribution code."""
    return ops.policy_act(policy_params, policy_spec, obs, norm_mean, norm_std, deterministic=deterministic,
                          seed=K.PRNGKey(key), offset=0)


class Evaluator:
    """sac/acting.py:82-145 on the fused rollout: num_eval_envs episodes of `episode_length` steps from fresh resets,
    deterministic (mode) or sampled actions; eval/episode_reward = mean over envs of the summed rewards."""

    def __init__(self, trainer: SAC, eval_env: BraxWrapper, num_eval_envs: int, episode_length: int, action_repeat: int, key: int):
        self.t, self.env = trainer, eval_env
        self.num_eval_envs, self.episode_length, self.action_repeat = num_eval_envs, episode_length, action_repeat
        self._key = key
        self._eval_walltime = 0.0
        self._steps_per_unroll = episode_length * num_eval_envs

    def run_evaluation(self, policy_params, training_metrics: Metrics, unroll_key: Optional[int] = None) -> Metrics:
        t = self.t
        if unroll_key is None:
            self._key, unroll_key = K.split(self._key)
        torch.cuda.synchronize()
        t0 = time.time()
        normalizer_params, pol = policy_params
        nm, ns = t._norm(normalizer_params)
        st = self.env.reset(K.split(unroll_key, self.num_eval_envs))
        n_steps = self.episode_length // self.action_repeat
        spec = self.env.system.rollout_spec(st.system_params, t.device)
        rows = ops.model_rollout(policy_params=pol, policy_spec=t.policy_spec, x_dim=t.x_dim, u_dim=t.u_dim, obs=st.obs,
                                 first_obs=st.info['first_obs'], steps=st.info['steps'], done=st.done, n_steps=n_steps,
                                 episode_length=self.episode_length, action_repeat=self.action_repeat, norm_mean=nm,
                                 norm_std=ns, deterministic=t.deterministic_eval, seed=unroll_key, offset=0, **spec)
        episode_reward, episode_steps = eval_metrics_from_rows(rows, t.x_dim, t.u_dim, n_steps, self.num_eval_envs, self.action_repeat)
        self.last_episode_rewards, self.last_episode_steps = episode_reward, episode_steps
        epoch_eval_time = time.time() - t0
        self._eval_walltime += epoch_eval_time
        return {'eval/walltime': self._eval_walltime, **training_metrics, 'eval/episode_reward': float(episode_reward.mean()),
                'eval/avg_episode_length': float(episode_steps.mean()), 'eval/epoch_eval_time': epoch_eval_time,
                'eval/sps': self._steps_per_unroll / max(epoch_eval_time, 1e-9)}


def eval_metrics_from_rows(rows: torch.Tensor, x_dim: int, u_dim: int, n_steps: int, n_envs: int, action_repeat: int):
    """EvalWrapper's bookkeeping (brax_utils/training.py:156-199) from the step-major transition rows of one unroll:
        active_0 = 1;  episode_reward += reward_t * active_t;  episode_steps = info['steps'] while active_t;
        active_{t+1} = active_t * (1 - done_t)                       (rows carry discount_t = 1 - done_t, sac/acting.py:50)
    info['steps'] counts action_repeat per env step from the reset (training.py:98), so while an episode is active it is
    (t + 1) * action_repeat.  Returns (episode_reward [N], episode_steps [N])."""
    rew = rows[:, x_dim + u_dim].reshape(n_steps, n_envs)
    disc = rows[:, x_dim + u_dim + 1].reshape(n_steps, n_envs)
    active = torch.ones_like(disc)
    if n_steps > 1:
        active[1:] = torch.cumprod(disc[:-1], dim=0)
    return (rew * active).sum(dim=0), active.sum(dim=0) * float(action_repeat)

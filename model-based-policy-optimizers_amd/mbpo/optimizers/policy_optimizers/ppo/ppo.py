"""PPO trainer — host-side mirror of mbpo/optimizers/policy_optimizers/ppo/ppo.py (same constructor arguments, derived
quantities, method names, call order, metric keys); every numeric step runs in libmbpo_hip.so.

Where the reference has                               this file issues
  scan of brax acting.generate_unroll (:194-208)  ->  K x mbpo_model_rollout (ppo_extras, env_major): rows land as
                                                      data[B*M, T, D] directly — no swapaxes/reshape (:210-213)
  running_statistics.update (:216-219)            ->  mbpo_running_stats_reduce x2 + _apply
  jr.permutation + reshape (:166-171)             ->  mbpo_philox_permutation + mbpo_replay_gather on [B*M] rows of T*D floats
  scan of minibatch_step (:172-176)               ->  M x (mbpo_ppo_grads [+ all-reduce] + mbpo_ppo_apply)

Randomness: as in the SAC trainer, every draw is Philox(seed word, (call-site id << 32) + step index, stream, element) with
the seed word (epoch key) and the training-step index in two device words (`self._rng`).
"""
from __future__ import annotations

import dataclasses
import math
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

from mbpo import _hip, ops
from mbpo.optimizers.policy_optimizers.brax_utils.base import State
from mbpo.optimizers.policy_optimizers.sac.sac import Evaluator, RunningStatisticsState, policy_act
from mbpo.parallel import DataParallel
from mbpo.systems.brax_wrapper import BraxWrapper
from mbpo.systems.ensemble_system import lecun_uniform_flat
from mbpo.utils import keys as K

Metrics = Dict[str, Any]

# Philox call-site ids (high 32 bits of the offset): unroll k, update epoch e (the permutation), minibatch (e, m)
SITE_UNROLL, SITE_PERM, SITE_MINIBATCH = 1, 1024, 65536


@dataclass
class PPONetworkParams:
    """ppo/losses.py:19-23."""
    policy: torch.Tensor
    value: torch.Tensor


@dataclass
class TrainingState:
    """ppo.py:36-44; tensors are views into the updater's flat device state."""
    optimizer_state: Any
    params: PPONetworkParams
    normalizer_params: RunningStatisticsState
    env_steps: int

    def get_policy_params(self):
        return self.normalizer_params, self.params.policy

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class PPO:
    def __init__(self,
                 environment: BraxWrapper,
                 num_timesteps: int,
                 episode_length: int,
                 action_repeat: int = 1,
                 num_envs: int = 1,
                 num_eval_envs: int = 128,
                 lr: float = 1e-4,
                 wd: float = 1e-5,
                 entropy_cost: float = 1e-4,
                 discounting: float = 0.9,
                 seed: int = 0,
                 unroll_length: int = 10,
                 batch_size: int = 32,
                 num_minibatches: int = 16,
                 num_updates_per_batch: int = 2,
                 num_evals: int = 1,
                 normalize_observations: bool = False,
                 reward_scaling: float = 1.,
                 clipping_epsilon: float = .3,
                 gae_lambda: float = .95,
                 deterministic_eval: bool = False,
                 normalize_advantage: bool = True,
                 policy_hidden_layer_sizes: Sequence[int] = (64, 64, 64),
                 policy_activation: str = "swish",
                 critic_hidden_layer_sizes: Sequence[int] = (64, 64, 64),
                 critic_activation: str = "swish",
                 wandb_logging: bool = False,
                 # --- MI355X-side knobs (not in the reference) ---
                 use_graph: bool = True,
                 process_group=None,
                 ):
        if wandb_logging:
            raise NotImplementedError("wandb is not available in this environment")
        self.episode_length = episode_length
        self.action_repeat = action_repeat
        self.num_timesteps = num_timesteps
        self.deterministic_eval = deterministic_eval
        self.normalize_observations = normalize_observations
        self.num_evals = num_evals
        self.num_updates_per_batch = num_updates_per_batch
        self.num_minibatches = num_minibatches
        self.batch_size = batch_size
        self.unroll_length = unroll_length
        self.num_eval_envs = num_eval_envs
        self.num_envs = num_envs
        assert batch_size * num_minibatches % num_envs == 0                                   # ppo.py:99
        self.env_step_per_training_step = batch_size * unroll_length * num_minibatches * action_repeat
        self.num_evals_after_init = max(num_evals - 1, 1)
        self.num_training_steps_per_epoch = math.ceil(num_timesteps / (self.num_evals_after_init * self.env_step_per_training_step))
        self.key = K.PRNGKey(seed)
        self.env = environment
        self.x_dim, self.u_dim = self.env.observation_size, self.env.action_size
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.policy_dims_logical = [self.x_dim, *policy_hidden_layer_sizes, 2 * self.u_dim]
        self.value_dims_logical = [self.x_dim, *critic_hidden_layer_sizes, 1]
        dyn_hidden = list(getattr(getattr(self.env.system, "dynamics", None), "dims", [])[1:-1]) if self.env.system.fused else []
        widest = max([int(h) for h in (*policy_hidden_layer_sizes, *critic_hidden_layer_sizes, *dyn_hidden)], default=0)
        if widest <= ops.KERNEL_WIDTHS[-1]:
            self.kernel_width = ops.common_width(policy_hidden_layer_sizes, critic_hidden_layer_sizes, dyn_hidden, what="PPO")
            self.value_width = self.kernel_width
        else:
            # wider than the fused update kernels take (ppo.py:60-63 accepts any tuple; exp_ppo.py: a 256x5 critic): the minibatch
            # update then runs layer by layer (csrc/ppo_layered.hip) — the value net keeps its logical sizes; the policy, which also
            # runs inside the rollout / act kernels, is padded to one of their widths
            self.kernel_width = ops.common_width(policy_hidden_layer_sizes, dyn_hidden, supported=ops.ROLLOUT_WIDTHS, what="PPO policy")
            self.value_width = None
        if dyn_hidden and any(h != self.kernel_width for h in dyn_hidden):
            raise _hip.MbpoHipError(f"PPO: the learned ensemble's hidden width {dyn_hidden} must equal the policy's kernel width "
                                    f"{self.kernel_width} inside the fused rollout (build the EnsembleDynamics with that width)")
        self.policy_dims = ops.padded_dims(self.policy_dims_logical, self.kernel_width)     # hidden layers zero-padded (ops.py)
        self.value_dims = ops.padded_dims(self.value_dims_logical, self.value_width)
        self.policy_spec = ops.MlpSpec(self.policy_dims, policy_activation, 1)
        self.dp = DataParallel(process_group)
        self._all_reduce = self.dp.all_reduce_fn()
        # multi-GPU: the flat gradient and the normaliser sums go through peer memory (csrc/p2p.hpp) when the exchange regions can
        # be set up and reproduce torch.distributed.all_reduce at start-up; otherwise the library collective stays
        self.p2p = None
        if self.dp.group is not None and self.dp.world_size > 1:
            from mbpo.parallel import P2PExchange
            n_pv = ops.MlpSpec(self.policy_dims).n_params + ops.MlpSpec(self.value_dims).n_params
            self.p2p = P2PExchange.create(self.dp, n_pv, self.device)
            if self.p2p is not None:
                self._all_reduce = self.p2p.all_reduce_sum
        self.updater = ops.PpoUpdater(
            x_dim=self.x_dim, u_dim=self.u_dim, policy_dims=self.policy_dims, value_dims=self.value_dims, batch_size=batch_size,
            unroll_length=unroll_length, device=self.device, policy_activation=policy_activation,
            value_activation=critic_activation, entropy_cost=entropy_cost, discounting=discounting,
            reward_scaling=reward_scaling, gae_lambda=gae_lambda, clipping_epsilon=clipping_epsilon,
            normalize_advantage=normalize_advantage, lr=lr, wd=wd, all_reduce=self._all_reduce, world_size=self.dp.world_size)
        self.row_len = ops.transition_row_len(self.x_dim, self.u_dim, True)
        n_traj = batch_size * num_minibatches
        self._data = torch.empty(n_traj, unroll_length, self.row_len, device=self.device)       # [B*M, T, D]
        self._shuffled = torch.empty_like(self._data)
        self._ring_state = torch.zeros(4, dtype=torch.int32, device=self.device)                # head = 0: plain row gather
        self._stats_vec = torch.zeros(1 + 3 * self.x_dim, device=self.device)
        self._stats_sums = torch.zeros(1 + 2 * self.x_dim, device=self.device)
        self._stats_ws = torch.empty(ops.stats_workspace_floats(self.x_dim), device=self.device)
        self._perm = torch.zeros(n_traj, dtype=torch.int32, device=self.device)
        self._perm_ws = torch.zeros(n_traj, dtype=torch.int32, device=self.device)
        self._rng = ops.make_rng(self.device)       # device uint64[2]: {epoch key, training-step index}
        self.use_graph = use_graph
        self._graph = None
        self._graph_key = None
        self._graph_refs = None

    # ------------------------------------------------------------------------------------------------ policy / state
    def _norm(self, normalizer_params: RunningStatisticsState):
        if not self.normalize_observations:
            return None, None
        return normalizer_params.mean.contiguous(), normalizer_params.std.contiguous()

    def rekey(self, key: int) -> None:
        """Start a fresh random stream: seed word <- key, step index <- 0."""
        ops.set_rng(self._rng, K.PRNGKey(key), 0)

    def make_policy(self, params, deterministic: bool = False):
        """make_inference_fn (ppo_network.py:59-84)."""
        normalizer_params, policy_params = params
        nm, ns = self._norm(normalizer_params)

        def policy(observations: torch.Tensor, key_sample: int):
            obs = observations.reshape(-1, self.x_dim).to(self.device, torch.float32).contiguous()
            act = policy_act(policy_params, self.policy_spec, obs, nm, ns, deterministic, key_sample)
            return (act[0] if observations.dim() == 1 else act), {}

        return policy

    def init_training_state(self, key: int) -> TrainingState:
        """ppo.py:265-277."""
        k0, k1 = K.split(key)
        pol = ops.embed_mlp_params(lecun_uniform_flat(self.policy_dims_logical, torch.Generator().manual_seed(k0 % (2 ** 63))),
                                   self.policy_dims_logical, self.kernel_width)
        val = ops.embed_mlp_params(lecun_uniform_flat(self.value_dims_logical, torch.Generator().manual_seed(k1 % (2 ** 63))),
                                   self.value_dims_logical, self.value_width)
        params = torch.cat([pol, val]).to(self.device)
        self.dp.broadcast(params, src=0)
        self.updater.load_state(params)
        self._stats_vec.zero_()
        self._stats_vec[1 + 2 * self.x_dim:] = 1.0
        return self._training_state(0)

    def _training_state(self, env_steps: int) -> TrainingState:
        u = self.updater
        return TrainingState(optimizer_state=(u.adam_m, u.adam_v, u.step_count),
                             params=PPONetworkParams(policy=u.policy_params, value=u.value_params),
                             normalizer_params=RunningStatisticsState(self._stats_vec, self.x_dim), env_steps=env_steps)

    # ------------------------------------------------------------------------------------------------ hot loops
    def minibatch_step(self, data: torch.Tensor, normalizer_params: RunningStatisticsState, key: Optional[int] = None,
                       e: int = 0, m: int = 0) -> None:
        """ppo.py:142-156 on one minibatch [B, T, D]; (e, m) = its position in the update-epoch / minibatch scans."""
        if key is not None:
            self.rekey(key)
        nm, ns = self._norm(normalizer_params)
        self.updater.minibatch_step(data, nm, ns, seed=0, offset=(SITE_MINIBATCH + e * self.num_minibatches + m) << 32,
                                    rng_dev=self._rng)

    def sgd_step(self, data: torch.Tensor, normalizer_params: RunningStatisticsState, key: Optional[int] = None, e: int = 0) -> None:
        """ppo.py:158-177: one shared permutation of the B*M trajectories, then M minibatch updates."""
        if key is not None:
            self.rekey(key)
        n = data.shape[0]
        perm = ops.philox_permutation(n, seed=0, offset=(SITE_PERM + e) << 32, rng_dev=self._rng, out=self._perm,
                                      workspace=self._perm_ws)
        # every leaf is permuted with the SAME key (ppo.py:166-169) == one row gather of whole trajectories
        flat = data.reshape(n, -1)
        shuffled = ops.replay_gather(flat, self._ring_state, perm, out=self._shuffled).reshape(
            self.num_minibatches, self.batch_size, self.unroll_length, self.row_len)
        for m in range(self.num_minibatches):
            self.minibatch_step(shuffled[m], normalizer_params, e=e, m=m)

    def training_step(self, training_state: TrainingState, state: State, key: Optional[int] = None):
        """ppo.py:179-233.  `key` given: re-key the device stream; None: the next step of the running stream."""
        new_key = None
        if key is not None:
            _, _, new_key = K.split(key, 3)
            self.rekey(key)
        nm, ns = self._norm(training_state.normalizer_params)
        spec = self.env.system.rollout_spec(state.system_params, self.device)
        n_unrolls = self.batch_size * self.num_minibatches // self.num_envs
        N, T = self.num_envs, self.unroll_length
        sp_out: list = []
        for k in range(n_unrolls):                                                               # scan :194-208
            ops.model_rollout(policy_params=training_state.params.policy, policy_spec=self.policy_spec, x_dim=self.x_dim,
                              u_dim=self.u_dim, obs=state.obs, first_obs=state.info['first_obs'], steps=state.info['steps'],
                              done=state.done, n_steps=T, episode_length=self.episode_length, action_repeat=self.action_repeat,
                              norm_mean=nm, norm_std=ns, ppo_extras=True, env_major=True, seed=0,
                              offset=(SITE_UNROLL + k) << 32, rng_dev=self._rng,
                              out=self._data[k * N:(k + 1) * N].reshape(N * T, self.row_len), system_params_out=sp_out, **spec)
            if sp_out:
                state = state.replace(system_params=sp_out[-1])
                spec = self.env.system.rollout_spec(state.system_params, self.device)
        # running_statistics.update(normalizer_params, data.observation)   (:216-219)
        rows = self._data.reshape(-1, self.row_len)
        ops.running_stats_update(rows, 0, self.x_dim, training_state.normalizer_params.vec, all_reduce=self._all_reduce,
                                 sums=self._stats_sums, workspace=self._stats_ws)
        for e in range(self.num_updates_per_batch):                                              # scan :222-226
            self.sgd_step(self._data, training_state.normalizer_params, e=e)
        ops.rng_advance(self._rng)
        training_state = training_state.replace(env_steps=training_state.env_steps + self.env_step_per_training_step)
        return training_state, state, new_key

    def training_epoch(self, training_state: TrainingState, state: State, key: int):
        """ppo.py:235-247.  The reference compiles the whole epoch scan into ONE XLA computation; here a training_step — K rollouts,
        3 statistics launches, E x (permutation, gather, M x minibatch_step) and the RNG advance, ~1.3 k launches at C3 — is captured
        once into a hipGraph and replayed (every launch reads its counters / RNG words from device memory, so a replayed step is bit
        for bit the eagerly issued one: tests/test_gpu_trainer_parity.py).  First step ever eager, all later ones replays, as in
        SAC.training_epoch."""
        self.updater.metrics_accum.zero_()
        self.rekey(key)
        n = self.num_training_steps_per_epoch
        done_steps = 0
        if self.use_graph and self._capturable():
            # first training_step ever against these buffers: eager, then captured; every later step of every epoch replays
            gkey, refs = self._graph_signature(training_state, state)
            if self._graph is None or self._graph_key != gkey:
                training_state, state, _ = self.training_step(training_state, state)
                done_steps = 1
                gkey, refs = self._graph_signature(training_state, state)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.training_step(training_state, state)
                self._graph, self._graph_key, self._graph_refs = graph, gkey, refs
            replays = n - done_steps
            for _ in range(replays):
                self._graph.replay()
            training_state = training_state.replace(env_steps=training_state.env_steps + replays * self.env_step_per_training_step)
            done_steps = n
        while done_steps < n:
            training_state, state, _ = self.training_step(training_state, state)
            done_steps += 1
        acc = self.updater.metrics_accum.cpu()
        if self.p2p is not None and self.p2p.status() != 0:
            raise _hip.MbpoHipError("PPO: the peer-memory gradient exchange timed out on this rank; "
                                    "set MBPO_P2P_ALLREDUCE=0 to use the RCCL all-reduce")
        cnt = max(float(acc[4]), 1.0)
        metrics = {'total_loss': float(acc[0]) / cnt, 'policy_loss': float(acc[1]) / cnt, 'v_loss': float(acc[2]) / cnt,
                   'entropy_loss': float(acc[3]) / cnt}
        return training_state, state, metrics

    def _capturable(self) -> bool:
        """As SAC._capturable: plain kernels only — single rank, the peer-memory exchange, or RCCL collectives (stream-ordered
        kernels); never a host-side (gloo) collective, never a user-defined System (user code runs between the kernels)."""
        if not self.env.system.fused:
            return False
        if self.dp.group is None or self.p2p is not None:
            return True
        import os
        import torch.distributed as dist
        return dist.get_backend(self.dp.group) == "nccl" and os.environ.get("MBPO_GRAPH_NCCL", "1") != "0"

    def _graph_signature(self, training_state: TrainingState, state: State):
        """Every device address a captured training_step bakes in — environment and scratch buffers AND the train state (ADVICE r3:
        a restored or cloned TrainingState must re-capture); the tensors are kept alive with the graph."""
        spec = self.env.system.rollout_spec(state.system_params, self.device)
        u = self.updater
        tensors = [state.obs, state.info['first_obs'], state.info['steps'], state.done, self._data, self._shuffled, self._stats_vec,
                   self._rng, self._perm, self._perm_ws, training_state.params.policy, training_state.params.value,
                   training_state.normalizer_params.vec, u.params, u.adam_m, u.adam_v, u.step_count, u.workspace]
        tensors += [v for v in spec.values() if isinstance(v, torch.Tensor)]
        return tuple(t.data_ptr() for t in tensors), tensors

    def training_epoch_with_timing(self, training_state, env_state, key):
        """ppo.py:249-263."""
        torch.cuda.synchronize()
        t = time.time()
        training_state, env_state, metrics = self.training_epoch(training_state, env_state, key)
        torch.cuda.synchronize()
        epoch_training_time = time.time() - t
        sps = (self.num_training_steps_per_epoch * self.env_step_per_training_step) / epoch_training_time
        metrics = {'training/sps': sps, **{f'training/{name}': value for name, value in metrics.items()}}
        return training_state, env_state, metrics

    def run_training(self, key: int, progress_fn: Callable[[int, Metrics], None] = lambda *args: None):
        """ppo.py:279-339."""
        key, subkey = K.split(key)
        training_state = self.init_training_state(subkey)
        key, rb_key, env_key, eval_key = K.split(key, 4)
        rk = self.dp.rank_key      # data-generating keys differ per rank; the init key above is shared (parameters are broadcast)
        env_state = self.env.reset(K.split(rk(env_key), self.num_envs))
        evaluator = Evaluator(self, self.env, num_eval_envs=self.num_eval_envs, episode_length=self.episode_length,
                              action_repeat=self.action_repeat, key=eval_key)
        all_metrics: List[Metrics] = []
        if self.num_evals > 1:
            metrics = evaluator.run_evaluation(self._snapshot(training_state), training_metrics={})
            all_metrics.append(metrics)
            progress_fn(0, metrics)
        key, prefill_key = K.split(key)
        current_step = 0
        for _ in range(self.num_evals_after_init):
            key, epoch_key = K.split(key)
            training_state, env_state, training_metrics = self.training_epoch_with_timing(training_state, env_state, rk(epoch_key))
            current_step = training_state.env_steps
            metrics = evaluator.run_evaluation(self._snapshot(training_state), training_metrics)
            all_metrics.append(metrics)
            progress_fn(current_step, metrics)
        return self._snapshot(training_state), all_metrics

    def close(self) -> None:
        self._graph = self._graph_key = self._graph_refs = None
        if self.p2p is not None:
            self.p2p.close()
            self.p2p = None

    def _snapshot(self, training_state: TrainingState):
        return (RunningStatisticsState(training_state.normalizer_params.vec.clone(), self.x_dim),
                training_state.params.policy.clone())

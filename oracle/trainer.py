"""Whole training steps assembled from the oracle pieces, on the CPU — test infrastructure only.

  CpuSacLoop   SAC.training_step   (sac/sac.py:283-327): get_experience -> running statistics -> replay insert -> sample
                                   -> G x sgd_step
  CpuPpoLoop   PPO.training_step   (ppo/ppo.py:179-233): K unrolls -> running statistics -> E x (permute, M x minibatch_step)

Both are driven by the SAME random streams as the product's trainers: every draw is
Philox(seed word, (call-site id << 32) + step index, stream, element) — oracle/philox.py restates csrc/common.hpp, and the
call-site ids are the trainers' own (sac/sac.py SITE_*, ppo/ppo.py SITE_* in the package) — so that
tests/test_gpu_trainer_parity.py can compare a training step of the HIP path (eager or replayed from a hipGraph) with
these loops value for value.  bench.py's cpu_baseline leg times CpuSacLoop.training_step ("port": the reference's JAX
path cannot run here).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import nets, philox, ppo, replay, rollout, sac

# call-site ids: must equal the constants of the product's trainers
SAC_SITE_ROLLOUT, SAC_SITE_SAMPLE, SAC_SITE_SGD = 1, 2, 16
PPO_SITE_UNROLL, PPO_SITE_PERM, PPO_SITE_MINIBATCH = 1, 1024, 65536


def _normal(seed: int, offset: int, stream: int, shape) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(philox.philox_normal(seed, offset, stream, np.arange(n, dtype=np.uint64))).reshape(*shape)


@dataclass
class CpuSacLoop:
    cfg: sac.SacConfig
    system: object
    n_envs: int
    n_steps: int            # S = num_env_steps_between_updates
    episode_length: int
    batch_size: int
    grad_updates: int       # G
    max_replay: int
    normalize: bool
    seed: int = 0           # the epoch key (seed word of the device RNG control)
    action_repeat: int = 1
    init_params: Optional[torch.Tensor] = None     # flat [policy | critic0 | critic1 | log_alpha]; None: fresh lecun-uniform
    init_obs: Optional[torch.Tensor] = None        # [N, x]; None: standard normal
    step_index: int = 0     # the device counter

    def __post_init__(self):
        g = torch.Generator().manual_seed(self.seed % (2 ** 63))
        X, U = self.cfg.x_dim, self.cfg.u_dim
        self.state = sac.init_state(self.cfg, g)
        if self.init_params is not None:
            p = self.init_params.clone()
            self.state = sac.SacState(p, p[self.cfg.P:self.cfg.P + 2 * self.cfg.Q].clone(), torch.zeros_like(p), torch.zeros_like(p), 0)
        obs = torch.randn(self.n_envs, X, generator=g) if self.init_obs is None else self.init_obs.clone()
        self.env = rollout.EnvState(obs, obs.clone(), torch.zeros(self.n_envs), torch.zeros(self.n_envs))
        self.queue = replay.UniformSamplingQueue(self.max_replay, 2 * X + U + 3, self.batch_size * self.grad_updates)
        self.qstate = self.queue.init()
        self.stats = replay.stats_init(X)
        self.last_rows = None
        self.last_idx = None

    def rekey(self, key: int):
        self.seed, self.step_index = key, 0

    def _norm(self):
        X = self.cfg.x_dim
        if not self.normalize:
            return None, None
        return torch.from_numpy(self.stats[1:1 + X].copy()), torch.from_numpy(self.stats[1 + 2 * X:].copy())

    def get_experience(self):
        """sac/sac.py:283-304."""
        X, U = self.cfg.x_dim, self.cfg.u_dim
        S, N = self.n_steps, self.n_envs
        nm, ns = self._norm()
        noise = _normal(self.seed, (SAC_SITE_ROLLOUT << 32) + self.step_index, philox.STREAM_POLICY_NOISE, (S, N, U))
        self.env, rows = rollout.rollout(self.system, self.state.params[:self.cfg.P], self.cfg.policy_dims, self.env, S,
                                         self.episode_length, self.action_repeat, self.cfg.policy_act, nm, ns, policy_noise=noise)
        self.stats = replay.stats_update(self.stats, rows[:, :X].numpy())
        self.qstate = self.queue.insert(self.qstate, rows.numpy())
        self.last_rows = rows
        return rows

    def prefill_step(self):
        self.get_experience()
        self.step_index += 1

    def training_step(self, n_sgd: Optional[int] = None):
        """sac/sac.py:306-327."""
        U = self.cfg.u_dim
        self.get_experience()
        idx, batch = self.queue.sample(self.qstate, self.seed, (SAC_SITE_SAMPLE << 32) + self.step_index)
        self.last_idx = idx
        batch = torch.from_numpy(batch)
        nm, ns = self._norm()
        B = self.batch_size
        met = None
        for gi in range(self.grad_updates if n_sgd is None else n_sgd):
            off = ((SAC_SITE_SGD + gi) << 32) + self.step_index
            noise = [_normal(self.seed, off, s, (B, U)) for s in (philox.STREAM_SAC_ALPHA, philox.STREAM_SAC_CRITIC, philox.STREAM_SAC_ACTOR)]
            self.state, met, _ = sac.sgd_step(self.cfg, self.state, batch[gi * B:(gi + 1) * B], *noise, nm, ns)
        self.step_index += 1
        return met


@dataclass
class CpuPpoLoop:
    cfg: ppo.PpoConfig
    system: object
    n_envs: int
    unroll_length: int      # T
    episode_length: int
    batch_size: int         # B
    num_minibatches: int    # M
    num_updates_per_batch: int   # E
    normalize: bool
    init_params: torch.Tensor    # flat [policy | value]
    init_obs: torch.Tensor       # [N, x]
    seed: int = 0
    action_repeat: int = 1
    step_index: int = 0
    last_data: Optional[torch.Tensor] = field(default=None, repr=False)

    def __post_init__(self):
        X = self.cfg.x_dim
        p = self.init_params.clone()
        self.state = ppo.PpoState(p, torch.zeros_like(p), torch.zeros_like(p), 0)
        obs = self.init_obs.clone()
        self.env = rollout.EnvState(obs, obs.clone(), torch.zeros(self.n_envs), torch.zeros(self.n_envs))
        self.stats = replay.stats_init(X)
        self.last_perms = []

    def rekey(self, key: int):
        self.seed, self.step_index = key, 0

    def _norm(self):
        X = self.cfg.x_dim
        if not self.normalize:
            return None, None
        return torch.from_numpy(self.stats[1:1 + X].copy()), torch.from_numpy(self.stats[1 + 2 * X:].copy())

    def training_step(self):
        """ppo/ppo.py:179-233."""
        X, U = self.cfg.x_dim, self.cfg.u_dim
        N, T, B, M = self.n_envs, self.unroll_length, self.batch_size, self.num_minibatches
        nm, ns = self._norm()       # the policy acts with the normaliser of the PREVIOUS step (:196-199)
        chunks = []
        for k in range(B * M // N):                                                               # scan :194-208
            noise = _normal(self.seed, ((PPO_SITE_UNROLL + k) << 32) + self.step_index, philox.STREAM_POLICY_NOISE, (T, N, U))
            self.env, rows = rollout.rollout(self.system, self.state.params[:self.cfg.P], self.cfg.policy_dims, self.env, T,
                                             self.episode_length, self.action_repeat, self.cfg.policy_act, nm, ns,
                                             policy_noise=noise, ppo_extras=True, env_major=True)
            chunks.append(rows.reshape(N, T, -1))
        data = torch.cat(chunks, 0)                                                                # [B*M, T, D]   (:210-213)
        self.last_data = data
        self.stats = replay.stats_update(self.stats, data.reshape(-1, data.shape[-1])[:, :X].numpy())   # :216-219
        nm, ns = self._norm()
        terms = None
        self.last_perms = []
        for e in range(self.num_updates_per_batch):                                                # scan :222-226
            perm = philox.philox_permutation(self.seed, ((PPO_SITE_PERM + e) << 32) + self.step_index, B * M)
            self.last_perms.append(perm)
            shuffled = data[torch.from_numpy(perm.astype(np.int64))].reshape(M, B, T, -1)         # :166-171
            for m in range(M):
                off = ((PPO_SITE_MINIBATCH + e * M + m) << 32) + self.step_index
                ent = _normal(self.seed, off, philox.STREAM_ENTROPY, (B, T, U))
                self.state, terms, _ = ppo.minibatch_step(self.cfg, self.state, shuffled[m], ent, nm, ns)
        self.step_index += 1
        return terms

"""One SAC.training_step (sac/sac.py:306-327) assembled from the oracle pieces, on the CPU — used as the
cpu_baseline leg of bench.py ("port": the build's own torch-CPU restatement; the reference's JAX path cannot run
here) and by the end-to-end parity test.  Test infrastructure only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import nets, replay, rollout, sac, systems


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
    seed: int = 0

    def __post_init__(self):
        g = torch.Generator().manual_seed(self.seed)
        self.gen = g
        X, U = self.cfg.x_dim, self.cfg.u_dim
        self.state = sac.init_state(self.cfg, g)
        obs = torch.randn(self.n_envs, X, generator=g)
        self.env = rollout.EnvState(obs, obs.clone(), torch.zeros(self.n_envs), torch.zeros(self.n_envs))
        self.queue = replay.UniformSamplingQueue(self.max_replay, 2 * X + U + 3, self.batch_size * self.grad_updates)
        self.qstate = self.queue.init()
        self.stats = replay.stats_init(X)
        self.sample_calls = 0

    def get_experience(self):
        X, U = self.cfg.x_dim, self.cfg.u_dim
        S, N = self.n_steps, self.n_envs
        nm = torch.from_numpy(self.stats[1:1 + X].copy()) if self.normalize else None
        ns = torch.from_numpy(self.stats[1 + 2 * X:].copy()) if self.normalize else None
        noise = torch.randn(S, N, U, generator=self.gen)
        self.env, rows = rollout.rollout(self.system, self.state.params[:self.cfg.P], self.cfg.policy_dims, self.env, S,
                                         self.episode_length, 1, self.cfg.policy_act, nm, ns, policy_noise=noise)
        self.stats = replay.stats_update(self.stats, rows[:, :X].numpy())
        self.qstate = self.queue.insert(self.qstate, rows.numpy())
        return rows

    def training_step(self, n_sgd: Optional[int] = None):
        X, U = self.cfg.x_dim, self.cfg.u_dim
        self.get_experience()
        self.sample_calls += 1
        _, batch = self.queue.sample(self.qstate, self.seed, self.sample_calls)
        batch = torch.from_numpy(batch)
        nm = torch.from_numpy(self.stats[1:1 + X].copy()) if self.normalize else None
        ns = torch.from_numpy(self.stats[1 + 2 * X:].copy()) if self.normalize else None
        B = self.batch_size
        met = None
        for gi in range(self.grad_updates if n_sgd is None else n_sgd):
            noise = [torch.randn(B, U, generator=self.gen) for _ in range(3)]
            self.state, met, _ = sac.sgd_step(self.cfg, self.state, batch[gi * B:(gi + 1) * B], *noise, nm, ns)
        return met

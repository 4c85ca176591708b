#!/usr/bin/env python
"""End-to-end MBPO on the Pendulum with every stage on the MI355X path:

    true transitions  ->  EnsembleDynamics.fit (N3, mbpo_ens_nll_grads + mbpo_adamw_step)
                      ->  SACOptimizer on EnsembleSystem (short model rollouts branched from true states + SAC updates)
                      ->  the policy acts on the TRUE PendulumSystem.

    python examples/mbpo_pendulum.py [--iters 2 --model-steps 1500 --sac-steps 40000]
"""
from __future__ import annotations

import argparse
import math
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))

import torch  # noqa: E402


def collect_uniform(system, n, gen, dev):
    """n true transitions from states / actions drawn uniformly over the Pendulum's range (full state coverage)."""
    th = (torch.rand(n, generator=gen) * 2 - 1) * math.pi
    x = torch.stack([torch.cos(th), torch.sin(th), (torch.rand(n, generator=gen) * 2 - 1) * 8], 1).to(dev)
    u = (torch.rand(n, 1, generator=gen) * 2 - 1).to(dev)
    nxt = system.step(x, u, system.reset().system_params)
    return x, u, nxt.reward, nxt.x_next


def true_return(system, optimizer, opt_state, steps=200):
    start = system.reset()
    x, total, true_params = start.x_next, 0.0, start.system_params      # the TRUE system's own parameters
    for _ in range(steps):
        u, opt_state = optimizer.act(x, opt_state, evaluate=True)
        nxt = system.step(x, u, true_params)
        x, total = nxt.x_next, total + float(nxt.reward)
    return total


def run(iters=2, n_true=4000, model_steps=1500, sac_steps=40_000, seed=0, verbose=True):
    from mbpo.optimizers import SACOptimizer
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import EnsembleDynamics, EnsembleSystem, PendulumReward, PendulumSystem
    from mbpo.types import Transition
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator().manual_seed(seed)
    true_system = PendulumSystem()
    s0 = true_system.reset()
    dummy = Transition(observation=s0.x_next, action=torch.zeros(1, device=dev), reward=s0.reward,
                       discount=torch.tensor(0.99, device=dev), next_observation=s0.x_next)
    true_buffer = UniformSamplingQueue(max_replay_size=iters * n_true, dummy_data_sample=dummy, sample_batch_size=1, device=dev)
    tbs = true_buffer.init(seed)
    dyn = EnsembleDynamics(3, 1, n_members=5)
    model = EnsembleSystem(dyn, PendulumReward(), mode="mean", predict_delta=True)
    dyn_params = dyn.init_params(seed + 1)
    history = []
    for it in range(iters):
        t0 = time.time()
        x, u, r, xn = collect_uniform(true_system, n_true, gen, dev)
        tbs = true_buffer.insert(tbs, Transition(observation=x, action=u, reward=r, discount=torch.ones(n_true, device=dev), next_observation=xn))
        n_rows = true_buffer.size(tbs)
        dyn_params, losses = dyn.fit(dyn_params, true_buffer.logical_data(tbs), num_steps=model_steps, batch_size=256, learning_rate=3e-3,
                                     key=seed + 10 * it, n_rows=n_rows)
        optimizer = SACOptimizer(system=model, true_buffer=true_buffer, num_timesteps=sac_steps, num_evals=2, reward_scaling=1,
                                 episode_length=10, episode_length_eval=10, normalize_observations=True, action_repeat=1, discounting=0.99,
                                 lr_policy=3e-4, lr_alpha=3e-4, lr_q=3e-4, num_envs=64, batch_size=128, grad_updates_per_step=64,
                                 max_replay_size=2 ** 15, min_replay_size=2 ** 9, num_eval_envs=16, deterministic_eval=True, tau=0.005,
                                 num_env_steps_between_updates=5)
        state = optimizer.init(key=seed + 3, true_buffer_state=tbs)
        state = state.replace(system_params=state.system_params.replace(dynamics_params=dyn_params))
        out = optimizer.train(opt_state=state)
        ret = true_return(true_system, optimizer, out.optimizer_state)
        history.append(dict(iteration=it, true_transitions=n_rows, model_nll=float(losses[-20:].mean()), true_return=ret,
                            seconds=time.time() - t0))
        if verbose:
            print(history[-1], flush=True)
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--model-steps", type=int, default=1500)
    ap.add_argument("--sac-steps", type=int, default=40_000)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    run(a.iters, model_steps=a.model_steps, sac_steps=a.sac_steps, seed=a.seed)

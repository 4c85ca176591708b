"""Generates tests/golden/ppo_step_small.npz and tests/golden/bptt_actor_small.npz: one PPO minibatch gradient (policy [3,64,2],
value [3,64,1], B=8, T=6) and one BPTT actor gradient through the analytic pendulum (actor/critic 64x2, horizon 6, 16
trajectories), every random input explicit, evaluated by the fp64 oracles (oracle/ppo.py, oracle/bptt.py).  Regenerate with
    python tests/golden/make_ppo_bptt_golden.py
The files pin the oracles (CPU test: reproduce to 1e-12) and the HIP path (GPU tests: fp32 tolerance)."""
import math
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import bptt as obptt, nets as onets, ppo as oppo  # noqa: E402

F64 = torch.float64


def build_ppo():
    X, U, B, T = 3, 1, 8, 6
    g = torch.Generator().manual_seed(20261005)
    cfg = oppo.PpoConfig(x_dim=X, u_dim=U, policy_dims=[X, 64, 2 * U], value_dims=[X, 64, 1], entropy_cost=1e-2, discounting=0.97,
                         reward_scaling=0.5, gae_lambda=0.9, clipping_epsilon=0.2, normalize_advantage=True, lr=1e-3, wd=1e-4)
    st = oppo.init_state(cfg, g)
    params = st.params.double() + 0.05 * torch.randn(st.params.shape, generator=g, dtype=F64)
    D = 2 * X + 2 * U + 4
    o = X + U
    data = torch.randn(B, T, D, generator=g, dtype=F64)
    data[..., X:o] = torch.tanh(data[..., o + 3 + X:o + 3 + X + U])
    data[..., o + 1] = (torch.rand(B, T, generator=g) > 0.15).double()
    data[..., D - 1] = (torch.rand(B, T, generator=g) < 0.2).double()
    data[..., o + 2 + X] = -1.0 + 0.5 * torch.randn(B, T, generator=g, dtype=F64)
    noise = torch.randn(B, T, U, generator=g, dtype=F64)
    nm, ns = torch.randn(X, generator=g, dtype=F64) * 0.3, torch.rand(X, generator=g, dtype=F64) + 0.5
    return cfg, params, data, noise, nm, ns


def eval_ppo(cfg, params, data, noise, nm, ns):
    grads, terms, vs, adv = oppo.grads(cfg, params, data, noise, nm, ns)
    return dict(grads=grads.numpy(), losses=np.array([float(terms[k]) for k in ("total_loss", "policy_loss", "v_loss", "entropy_loss")]),
                vs=vs.numpy(), adv=adv.numpy())


def build_bptt():
    X, U, H, n = 3, 1, 6, 16
    g = torch.Generator().manual_seed(20261006)
    cfg = obptt.BpttConfig(x_dim=X, u_dim=U, actor_dims=[X, 64, 64, 2 * U], critic_dims=[X, 64, 64, 1], horizon=H, discount=0.97,
                           lambda_=0.9, ent_coef=0.05, init_stddev=1.0)
    ap = onets.init_mlp_flat(cfg.actor_dims, g).double() + 0.02 * torch.randn(cfg.P, generator=g, dtype=F64)
    cp = torch.cat([onets.init_mlp_flat(cfg.critic_dims, g).double() + 0.02 * torch.randn(cfg.C, generator=g, dtype=F64) for _ in range(2)])
    th = (torch.rand(n, generator=g, dtype=F64) * 2 - 1) * math.pi
    x0 = torch.stack([torch.cos(th), torch.sin(th), (torch.rand(n, generator=g, dtype=F64) * 2 - 1) * 4], 1)
    noise = torch.randn(n, H, U, generator=g, dtype=F64)
    s_mean, s_std = torch.randn(X, generator=g, dtype=F64) * 0.2, torch.rand(X, generator=g, dtype=F64) + 0.6
    r_ms = torch.tensor([-1.3, 2.1], dtype=F64)
    return cfg, ap, cp, x0, noise, s_mean, s_std, r_ms


def eval_bptt(cfg, ap, cp, x0, noise, s_mean, s_std, r_ms):
    grads, loss, aux = obptt.actor_grads(cfg, obptt.TorchPendulumSystem(), ap, cp, x0, noise, s_mean, s_std, r_ms[0], r_ms[1])
    return dict(grads=grads.numpy(), losses=np.array([float(loss), float(aux["entropy_loss"])]), lambda_values=aux["lambda_values"].numpy(),
                next_observation=aux["next_observation"].numpy(), reward=aux["reward"].numpy())


if __name__ == "__main__":
    here = Path(__file__).parent
    cfg, params, data, noise, nm, ns = build_ppo()
    out = eval_ppo(cfg, params, data, noise, nm, ns)
    np.savez_compressed(here / "ppo_step_small.npz", params=params.numpy(), data=data.numpy(), noise=noise.numpy(), norm_mean=nm.numpy(),
                        norm_std=ns.numpy(), **out)
    print("ppo", {k: v.shape for k, v in out.items()})
    cfg, ap, cp, x0, noise, s_mean, s_std, r_ms = build_bptt()
    out = eval_bptt(cfg, ap, cp, x0, noise, s_mean, s_std, r_ms)
    np.savez_compressed(here / "bptt_actor_small.npz", actor_params=ap.numpy(), critic_params=cp.numpy(), x0=x0.numpy(), noise=noise.numpy(),
                        state_mean=s_mean.numpy(), state_std=s_std.numpy(), reward_mean_std=r_ms.numpy(), **out)
    print("bptt", {k: v.shape for k, v in out.items()})

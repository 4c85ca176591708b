"""Generates tests/golden/sac_step_small.npz: one SAC sgd_step on small networks (policy [3,64,2], critics [4,64,1], B = 16)
with every random input explicit, evaluated by the fp64 oracle (oracle/sac.py).  Regenerate with
    python tests/golden/make_sac_step_golden.py
The file pins the oracle (CPU test: the oracle must reproduce it to 1e-12) and the HIP path (GPU test: fp32 tolerance)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import sac as osac  # noqa: E402


def build():
    X, U, B = 3, 1, 16
    g = torch.Generator().manual_seed(20261004)
    cfg = osac.SacConfig(x_dim=X, u_dim=U, policy_dims=[X, 64, 2 * U], q_dims=[X + U, 64, 1], discounting=0.97, reward_scaling=2.0,
                         lr_policy=1e-3, lr_q=2e-3, lr_alpha=5e-4, wd_q=1e-3, max_grad_norm=0.5)
    st = osac.init_state(cfg, g, init_log_alpha=-0.7, dtype=torch.float64)
    st.params = st.params + 0.05 * torch.randn(st.params.shape, generator=g, dtype=torch.float64)
    st.target_q = st.params[cfg.P:cfg.P + 2 * cfg.Q] + 0.03 * torch.randn(2 * cfg.Q, generator=g, dtype=torch.float64)
    D = 2 * X + U + 3
    batch = torch.randn(B, D, generator=g, dtype=torch.float64)
    batch[:, X:X + U] = torch.tanh(batch[:, X:X + U])
    batch[:, X + U + 1] = (torch.rand(B, generator=g) > 0.2).double()
    batch[:, D - 1] = (torch.rand(B, generator=g) < 0.25).double()
    noise = [torch.randn(B, U, generator=g, dtype=torch.float64) for _ in range(3)]
    nm, ns = torch.randn(X, generator=g, dtype=torch.float64) * 0.3, torch.rand(X, generator=g, dtype=torch.float64) + 0.5
    return cfg, st, batch, noise, nm, ns


def evaluate(cfg, st, batch, noise, nm, ns):
    grads, (cl, ac, al) = osac.grads(cfg, st.params, st.target_q, batch, *noise, nm, ns)
    new, met, _ = osac.sgd_step(cfg, st, batch, *noise, nm, ns)
    return dict(grads=grads.numpy(), losses=np.array([cl, ac, al]), new_params=new.params.numpy(), new_target_q=new.target_q.numpy(),
                alpha=np.array(met["alpha"]))


if __name__ == "__main__":
    cfg, st, batch, noise, nm, ns = build()
    out = evaluate(cfg, st, batch, noise, nm, ns)
    np.savez_compressed(Path(__file__).with_name("sac_step_small.npz"), params=st.params.numpy(), target_q=st.target_q.numpy(),
                        batch=batch.numpy(), noise_alpha=noise[0].numpy(), noise_critic=noise[1].numpy(), noise_actor=noise[2].numpy(),
                        norm_mean=nm.numpy(), norm_std=ns.numpy(), **out)
    print("written", {k: v.shape for k, v in out.items()})

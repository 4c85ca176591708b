"""Ensemble model learning (N3) — torch-CPU restatement (test infrastructure); gradients via torch.autograd.

Not in the reference (its learned model would come from the external `bsm` package, setup.py:22): the loss is this
build's definition, the inverse of what the rollout consumes (oracle/systems.py EnsembleSystem):
    (mu, raw) = MLP_e([x, u]);  mean = mu (+ x if predict_delta);  sigma = softplus(raw) + min_std
    loss_e = mean_b sum_d [ 0.5 ((x'_d - mean_d) / sigma_d)^2 + log sigma_d ]
parity unpinned by construction.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn.functional as F

from . import nets


def member_nll(params_e: torch.Tensor, dims: Sequence[int], xu: torch.Tensor, x: torch.Tensor, x_next: torch.Tensor,
               predict_delta: bool = True, min_std: float = 1e-3, act: str = "swish") -> torch.Tensor:
    X = x.shape[1]
    out = nets.mlp_forward(params_e, dims, xu, act)
    mean = out[:, :X] + (x if predict_delta else 0.0)
    sigma = F.softplus(out[:, X:]) + min_std
    q = (x_next - mean) / sigma
    return (0.5 * q * q + torch.log(sigma)).sum(dim=1).mean()


def nll_grads(params: torch.Tensor, dims: Sequence[int], n_members: int, rows: torch.Tensor, idx: torch.Tensor, x_dim: int, u_dim: int,
              predict_delta: bool = True, min_std: float = 1e-3, next_obs_off: int | None = None, act: str = "swish"):
    """params [E*P]; rows [R, D]; idx [E, B] (long).  Returns (grads [E*P], losses [E])."""
    P = nets.n_params(dims)
    noff = x_dim + u_dim + 2 if next_obs_off is None else next_obs_off
    grads, losses = [], []
    for e in range(n_members):
        p = params[e * P:(e + 1) * P].clone().requires_grad_(True)
        b = rows[idx[e]]
        loss = member_nll(p, dims, b[:, :x_dim + u_dim], b[:, :x_dim], b[:, noff:noff + x_dim], predict_delta, min_std, act)
        loss.backward()
        grads.append(p.grad.detach())
        losses.append(loss.detach())
    return torch.cat(grads), torch.stack(losses)

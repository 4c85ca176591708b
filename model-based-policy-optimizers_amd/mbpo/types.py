"""Transition — the brax.training.types.Transition the reference passes around (NamedTuple there).

Flattening order is jax.flatten_util.ravel_pytree's: fields in declaration order, `extras` dict keys sorted
('policy_extras' < 'state_extras'; 'log_prob' < 'raw_action').  This is the row layout of every replay buffer and of
the rollout kernel's output (include/mbpo_hip.h).
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch


@dataclass
class Transition:
    observation: torch.Tensor
    action: torch.Tensor
    reward: torch.Tensor
    discount: torch.Tensor
    next_observation: torch.Tensor
    extras: Dict[str, Any] = field(default_factory=dict)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


def _has(extras, a, b):
    return bool(extras) and a in extras and b in extras[a]


def row_layout(x_dim: int, u_dim: int, truncation: bool, ppo_extras: bool):
    """name -> (offset, width) of each leaf inside a flattened row."""
    lay, off = {}, 0
    for name, w in (("observation", x_dim), ("action", u_dim), ("reward", 1), ("discount", 1), ("next_observation", x_dim)):
        lay[name] = (off, w)
        off += w
    if ppo_extras:
        lay["log_prob"] = (off, 1)
        off += 1
        lay["raw_action"] = (off, u_dim)
        off += u_dim
    if truncation:
        lay["truncation"] = (off, 1)
        off += 1
    return lay, off


def flatten(t: Transition) -> torch.Tensor:
    """[B, D] rows from a batched Transition (leaves [B,...])."""
    B = t.observation.shape[0]
    parts = [t.observation.reshape(B, -1), t.action.reshape(B, -1), t.reward.reshape(B, 1), t.discount.reshape(B, 1),
             t.next_observation.reshape(B, -1)]
    if _has(t.extras, "policy_extras", "log_prob"):
        parts += [t.extras["policy_extras"]["log_prob"].reshape(B, 1), t.extras["policy_extras"]["raw_action"].reshape(B, -1)]
    if _has(t.extras, "state_extras", "truncation"):
        parts.append(t.extras["state_extras"]["truncation"].reshape(B, 1))
    return torch.cat([p.to(torch.float32) for p in parts], dim=1).contiguous()


def unflatten(rows: torch.Tensor, x_dim: int, u_dim: int, truncation: bool, ppo_extras: bool) -> Transition:
    lay, D = row_layout(x_dim, u_dim, truncation, ppo_extras)
    if rows.shape[-1] != D:
        raise ValueError(f"rows have {rows.shape[-1]} columns, layout needs {D}")
    g = lambda n: rows[..., lay[n][0]:lay[n][0] + lay[n][1]]
    extras: Dict[str, Any] = {}
    if ppo_extras:
        extras["policy_extras"] = {"log_prob": g("log_prob")[..., 0], "raw_action": g("raw_action")}
    if truncation:
        extras.setdefault("policy_extras", {})
        extras["state_extras"] = {"truncation": g("truncation")[..., 0]}
    return Transition(observation=g("observation"), action=g("action"), reward=g("reward")[..., 0],
                      discount=g("discount")[..., 0], next_observation=g("next_observation"), extras=extras)

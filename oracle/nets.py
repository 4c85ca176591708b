"""Networks and the NormalTanh action distribution — torch-CPU restatement (test infrastructure).

Follows the in-tree text of the [3P] brax builders the reference calls:
  MLP / QModule / policy network: mbpo/optimizers/policy_optimizers/sac/networks.py:19-100 (dead copy of
    brax.training.networks; live call sites sac/sac_networks.py:31-43, ppo/ppo_network.py:34-47)
  NormalTanhDistribution: sac/parametric_distribution.py:66-124 (dead copy of brax.training.distribution)
[3P, unverifiable here]: flax Dense computes x @ kernel[in,out] + bias; lecun_uniform = U(+-sqrt(3/fan_in));
  bias init zeros; swish(x) = x*sigmoid(x); distrax.Tanh.forward_log_det_jacobian(z) = 2*(log2 - z - softplus(-2z)).

Flat parameter layout (build-defined, see include/mbpo_hip.h): per layer W[in][out] row-major then b[out].
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch
import torch.nn.functional as F

LOG_2PI_HALF = 0.5 * math.log(2.0 * math.pi)
MIN_STD = 0.001  # NormalTanhDistribution(min_std=0.001): sac/parametric_distribution.py:100


def n_params(dims: Sequence[int]) -> int:
    return sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))


def init_mlp_flat(dims: Sequence[int], gen: torch.Generator, dtype=torch.float32) -> torch.Tensor:
    """lecun_uniform kernels, zero biases (sac/networks.py:23,33-38) — [3P] init distribution."""
    parts = []
    for i in range(len(dims) - 1):
        fan_in = dims[i]
        bound = math.sqrt(3.0 / fan_in)
        w = (torch.rand(dims[i], dims[i + 1], generator=gen, dtype=torch.float64) * 2.0 - 1.0) * bound
        parts.append(w.reshape(-1).to(dtype))
        parts.append(torch.zeros(dims[i + 1], dtype=dtype))
    return torch.cat(parts)


def unflatten(params: torch.Tensor, dims: Sequence[int]) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    out = []
    off = 0
    for i in range(len(dims) - 1):
        w = params[off:off + dims[i] * dims[i + 1]].reshape(dims[i], dims[i + 1])
        off += dims[i] * dims[i + 1]
        b = params[off:off + dims[i + 1]]
        off += dims[i + 1]
        out.append((w, b))
    return out


def activation(x: torch.Tensor, act: str) -> torch.Tensor:
    if act in ("swish", "silu"):
        return x * torch.sigmoid(x)
    if act == "relu":
        return torch.relu(x)
    if act == "tanh":
        return torch.tanh(x)
    raise ValueError(act)


def mlp_forward(params: torch.Tensor, dims: Sequence[int], x: torch.Tensor, act: str = "swish") -> torch.Tensor:
    """MLP.__call__ (sac/networks.py:27-41): Dense + activation for all but the last layer."""
    h = x
    layers = unflatten(params, dims)
    for i, (w, b) in enumerate(layers):
        h = h @ w + b
        if i != len(layers) - 1:
            h = activation(h, act)
    return h


def ensemble_forward(params: torch.Tensor, dims: Sequence[int], n_nets: int, x: torch.Tensor, act: str = "swish",
                     shared_input: bool = True) -> torch.Tensor:
    """y[e] = MLP_e(x) for an ensemble stored as consecutive flat nets.  x: [N,in] or [E,N,in]."""
    p = n_params(dims)
    ys = []
    for e in range(n_nets):
        xe = x if shared_input else x[e]
        ys.append(mlp_forward(params[e * p:(e + 1) * p], dims, xe, act))
    return torch.stack(ys)


def q_forward(params: torch.Tensor, dims: Sequence[int], obs: torch.Tensor, action: torch.Tensor, act: str = "swish",
              n_critics: int = 2) -> torch.Tensor:
    """QModule.__call__ (sac/networks.py:58-68): concat(obs, act) -> n_critics independent MLPs -> [B, n_critics]."""
    hidden = torch.cat([obs, action], dim=-1)
    p = n_params(dims)
    res = [mlp_forward(params[k * p:(k + 1) * p], dims, hidden, act) for k in range(n_critics)]
    return torch.cat(res, dim=-1)


# ---------------------------------------------------------------------------------------- NormalTanh
def split_logits(logits: torch.Tensor):
    """create_dist (sac/parametric_distribution.py:117-120): loc, scale = split; scale = softplus(scale)+min_std."""
    u = logits.shape[-1] // 2
    loc, raw = logits[..., :u], logits[..., u:]
    return loc, F.softplus(raw) + MIN_STD


def tanh_log_det_jacobian(z: torch.Tensor) -> torch.Tensor:
    """[3P] distrax.Tanh.forward_log_det_jacobian: 2*(log 2 - z - softplus(-2z))."""
    return 2.0 * (math.log(2.0) - z - F.softplus(-2.0 * z))


def sample_no_postprocessing(logits: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """:55-56 — Normal(loc, scale).sample == loc + scale*eps with eps the explicit standard-normal draw."""
    loc, scale = split_logits(logits)
    return loc + scale * eps


def log_prob(logits: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """ParametricDistribution.log_prob (:66-73): Normal log-density minus tanh log-det-jacobian, summed over u."""
    loc, scale = split_logits(logits)
    lp = -0.5 * ((z - loc) / scale) ** 2 - torch.log(scale) - LOG_2PI_HALF
    lp = lp - tanh_log_det_jacobian(z)
    return lp.sum(dim=-1)


def entropy(logits: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """ParametricDistribution.entropy (:75-83): Normal entropy + log-det-jacobian at a fresh sample, summed."""
    loc, scale = split_logits(logits)
    ent = 0.5 + LOG_2PI_HALF + torch.log(scale)
    ent = ent + tanh_log_det_jacobian(loc + scale * eps)
    return ent.sum(dim=-1)


def postprocess(z: torch.Tensor) -> torch.Tensor:
    return torch.tanh(z)


def mode(logits: torch.Tensor) -> torch.Tensor:
    """:62-64 — tanh(loc)."""
    loc, _ = split_logits(logits)
    return torch.tanh(loc)


# ---------------------------------------------------------------------------------------- normalizer
def normalize(obs: torch.Tensor, mean, std) -> torch.Tensor:
    """[3P] brax running_statistics.normalize: (batch - mean) / std; identity when normalize_observations=False
    (sac/sac.py:158-163)."""
    if mean is None:
        return obs
    return (obs - mean) / std

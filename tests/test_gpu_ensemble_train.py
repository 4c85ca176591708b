"""GPU parity + learning test of the ensemble model-learning step (N3): mbpo_ens_nll_grads vs torch autograd
(oracle/ensemble.py), and EnsembleDynamics.fit on true Pendulum transitions.

Tolerance: gradients atol 2e-6 + rtol 5e-4 against the fp32 oracle, relative L2 < 5e-5 against fp64; losses 2e-5."""
import math

import numpy as np
import pytest
import torch

from oracle import ensemble as oens
from oracle import nets as onets

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("X,U,E,B,hidden,delta,seed", [
    (4, 1, 5, 256, (64, 64, 64), True, 0),     # the bench's ensemble
    (3, 1, 3, 70, (64, 64, 64), True, 1),      # Pendulum shape, ragged batch
    (8, 2, 2, 32, (64, 64), False, 2),         # widest output the one-tile fast path takes (2x = 16), absolute prediction
    (17, 6, 2, 48, (64, 64, 64), True, 3),     # config-5 shape: output 34 wide -> generic output-layer routines
    (4, 1, 4, 16 * 150, (64,), True, 4),       # one hidden layer; more tiles than slots: slabs accumulate
])
def test_ens_nll_grads_parity(dev, X, U, E, B, hidden, delta, seed):
    from mbpo import ops
    g = torch.Generator().manual_seed(seed)
    dims = [X + U, *hidden, 2 * X]
    P = onets.n_params(dims)
    params = torch.cat([onets.init_mlp_flat(dims, g) + 0.02 * torch.randn(P, generator=g) for _ in range(E)])
    R, D = 500, 2 * X + U + 2
    rows = torch.randn(R, D, generator=g)
    rows[:, X + U + 2:] = rows[:, :X] + 0.1 * torch.randn(R, X, generator=g)
    idx = torch.randint(0, R, (E, B), generator=g)
    ref_g, ref_l = oens.nll_grads(params, dims, E, rows, idx, X, U, delta, 1e-3)
    op = ops.EnsembleNllGrad(x_dim=X, u_dim=U, spec=ops.MlpSpec(dims, "swish", E), batch=B, device=dev, predict_delta=delta)
    got = op(params.to(dev), rows.to(dev), idx.to(torch.int32).to(dev))
    torch.cuda.synchronize()
    np.testing.assert_allclose(op.metrics.cpu().numpy(), ref_l.numpy(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(got.cpu(), ref_g, atol=2e-6, rtol=5e-4)
    g64, _ = oens.nll_grads(params.double(), dims, E, rows.double(), idx, X, U, delta, 1e-3)
    rel = float((got.cpu().double() - g64).norm() / g64.norm())
    assert rel < 5e-5, rel


def test_ensemble_fit_learns_pendulum_dynamics(dev):
    """EnsembleDynamics.fit on true Pendulum transitions: the NLL falls and the ensemble mean predicts held-out next states."""
    from mbpo.systems import EnsembleDynamics, PendulumSystem
    from mbpo.systems.base_systems import SystemParams
    system = PendulumSystem()
    g = torch.Generator().manual_seed(0)
    n = 6000
    th = (torch.rand(n, generator=g) * 2 - 1) * math.pi
    x = torch.stack([torch.cos(th), torch.sin(th), (torch.rand(n, generator=g) * 2 - 1) * 6], 1).to(dev)
    u = (torch.rand(n, 1, generator=g) * 2 - 1).to(dev)
    sp = system.reset().system_params
    nxt = system.step(x, u, sp)
    rows = torch.cat([x, u, nxt.reward[:, None], torch.ones(n, 1, device=dev), nxt.x_next], 1)
    train, test = rows[:5000], rows[5000:]
    dyn = EnsembleDynamics(3, 1, n_members=5)
    params = dyn.init_params(1)
    params, losses = dyn.fit(params, train, num_steps=1500, batch_size=256, learning_rate=3e-3, key=7)
    l0, l1 = float(losses[:20].mean()), float(losses[-20:].mean())
    assert l1 < l0 - 3.0, (l0, l1)                       # starts near sum_d log(softplus(0)+1e-3) + errors, ends strongly negative
    dist, _ = dyn.next_state(test[:, :3], test[:, 3:4], params)
    err = float((dist.mean() - test[:, 6:9]).abs().mean())
    base = float((test[:, :3] - test[:, 6:9]).abs().mean())    # predicting "no change"
    assert err < 0.25 * base, (err, base)

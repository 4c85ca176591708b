"""GPU: mbpo.utils.optimizer_utils (reference: mbpo/utils/optimizer_utils.py:11-161) — same names and argument order over the
kernels; checked against the oracle and the hand KATs."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import scans as oscans, systems as osys

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def test_rollout_actions_and_rollout_policy(dev):
    from mbpo.systems import PendulumSystem
    from mbpo.utils.optimizer_utils import rollout_actions, rollout_policy
    system = PendulumSystem()
    sp = system.init_params(0)
    H = 12
    g = torch.Generator().manual_seed(0)
    actions = torch.rand(H, 1, generator=g) * 2 - 1
    x0 = torch.tensor([-1.0, 0.0, 0.0])
    tr = rollout_actions(system, sp, x0.to(dev), actions.to(dev), H)
    assert tr.observation.shape == (H, 3) and tr.action.shape == (H, 1) and tr.reward.shape == (H,) and tr.next_observation.shape == (H, 3)
    ref = osys.PendulumSystem()
    x, obs, nxt, rew = x0[None], [], [], []
    for t in range(H):
        xn, r = ref.step(x, actions[t][None])
        obs.append(x[0]); nxt.append(xn[0]); rew.append(r[0]); x = xn
    torch.testing.assert_close(tr.observation.cpu(), torch.stack(obs), atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(tr.next_observation.cpu(), torch.stack(nxt), atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(tr.reward.cpu(), torch.stack(rew), atol=5e-5, rtol=2e-5)
    assert torch.equal(tr.discount.cpu(), torch.ones(H)) and torch.equal(tr.action.cpu(), actions)
    assert torch.equal(tr.observation[1:], tr.next_observation[:-1])          # state[1:] = next_state[:-1]  (:48-50)
    # batched initial states + the same actions replayed through a policy callable
    N = 7
    xb = torch.stack([torch.cos(torch.linspace(0, 3, N)), torch.sin(torch.linspace(0, 3, N)), torch.linspace(-1, 1, N)], 1).to(dev)
    ab = (torch.rand(H, N, 1, generator=g) * 2 - 1).to(dev)
    trb = rollout_actions(system, sp, xb, ab, H)
    assert trb.observation.shape == (H, N, 3) and trb.reward.shape == (H, N)

    def policy(obs, t):
        return ab[t], t + 1
    trp = rollout_policy(system, sp, xb, policy, 0, H)
    torch.testing.assert_close(trp.next_observation, trb.next_observation, atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(trp.reward, trb.reward, atol=1e-6, rtol=1e-6)
    with pytest.raises(AssertionError):
        rollout_actions(system, sp, x0.to(dev), actions.to(dev), H + 1)


def test_lambda_return_static_scan_soft_update(dev):
    from mbpo.utils.optimizer_utils import lambda_return, soft_update, static_scan
    kat = json.loads((GOLD / "scan_kat.json").read_text())
    for c in kat["lambda_return"]:
        got = lambda_return(torch.tensor(c["reward"], device=dev), torch.tensor(c["next_values"], device=dev), c["discount"], c["lambda"])
        np.testing.assert_allclose(got.cpu().numpy(), c["returns"], rtol=1e-6, err_msg=c["why"])
    g = torch.Generator().manual_seed(1)
    r, v = torch.randn(20, 33, generator=g), torch.randn(20, 33, generator=g)
    got = lambda_return(r.to(dev), v.to(dev), 0.97, 0.9)
    ref = np.stack([oscans.lambda_return(r[:, b].numpy(), v[:, b].numpy(), 0.97, 0.9, dtype=np.float64) for b in range(33)], 1)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    # static_scan with the same recurrence reproduces lambda_return (:127-131)
    inputs = (r + 0.97 * v * (1 - 0.9)).to(dev)
    ss = static_scan(lambda agg, inp: inp + 0.97 * 0.9 * agg, inputs, v[-1].to(dev), reverse=True)
    torch.testing.assert_close(ss, got, atol=1e-5, rtol=1e-5)
    t, o = torch.randn(1000, generator=g).to(dev), torch.randn(1000, generator=g).to(dev)
    torch.testing.assert_close(soft_update(t, o, 0.005), (1 - 0.005) * t + 0.005 * o, atol=1e-7, rtol=1e-6)


def test_systems_exports_match_the_reference():
    """mbpo/systems/__init__.py:1-4 and mbpo/optimizers/__init__.py:1-6, name for name."""
    import mbpo.optimizers as mo
    import mbpo.systems as ms
    for n in ("System", "SystemState", "SystemParams", "PendulumSystem", "PendulumDynamics", "PendulumReward", "DynamicsParams",
              "Dynamics", "RewardParams", "Reward"):
        assert hasattr(ms, n), n
    for n in ("SAC", "iCemTO", "iCemOptimizerState", "iCemParams", "iCEMOptimizer", "BaseOptimizer", "PPOOptimizer", "SACOptimizer",
              "BraxOptimizer", "BraxState", "BraxOutput", "BPTTOptimizer", "BPTTState"):
        assert hasattr(mo, n), n
    from mbpo.utils.optimizer_utils import lambda_return, rollout_actions, rollout_policy, soft_update, static_scan  # noqa: F401

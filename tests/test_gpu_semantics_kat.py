"""GPU: the HIP path against tests/golden/semantics_kat.json (hand-derived known answers; generator
tests/golden/make_semantics_kat.py imports neither oracle/ nor the product).  fp32 tolerances stated per check."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KAT = json.loads((Path(__file__).parent / "golden" / "semantics_kat.json").read_text())


def _bias_policy(logits, x_dim, dev):
    """A one-Dense-layer 'policy' with zero kernel and bias = logits: the network output IS the KAT's logits."""
    from mbpo import ops
    dims = [x_dim, len(logits)]
    params = torch.cat([torch.zeros(x_dim * len(logits)), torch.tensor(logits, dtype=torch.float32)]).to(dev)
    return params, ops.MlpSpec(dims)


def test_normal_tanh_kat_through_policy_act_and_fused_rollout(dev):
    """sample / log_prob / mode at the hand points, through mbpo_policy_act AND through the fused rollout kernel's PPO extras
    (log_prob, raw_action columns).  atol 2e-6 on z and the action; log_prob 2e-5 relative (hardware exp/log in the fused kernel)."""
    from mbpo import _hip, ops
    for c in KAT["normal_tanh"]:
        U = len(c["eps"])
        X = 3 if U == 1 else 4
        params, spec = _bias_policy(c["logits"], X, dev)
        obs = torch.zeros(1, X, device=dev)
        eps = torch.tensor(c["eps"], device=dev).reshape(1, U)
        act, raw, lp = ops.policy_act(params, spec, obs, noise=eps, want_extras=True)
        np.testing.assert_allclose(raw.cpu().numpy()[0], c["z"], atol=2e-6, rtol=2e-6, err_msg=c["why"])
        np.testing.assert_allclose(act.cpu().numpy()[0], c["action"], atol=2e-6, rtol=2e-6, err_msg=c["why"])
        np.testing.assert_allclose(float(lp), c["log_prob"], rtol=2e-5, atol=2e-5, err_msg=c["why"])
        mode = ops.policy_act(params, spec, obs, deterministic=True)
        np.testing.assert_allclose(mode.cpu().numpy()[0], c["mode"], atol=2e-6, rtol=2e-6, err_msg=c["why"])
        # fused kernel: one env, one step, PPO row layout [obs, action, reward, discount, next_obs, log_prob, raw_action, trunc]
        if U == 1:
            kw = dict(system_kind=_hip.SYS_PENDULUM, sys_params=torch.tensor([8.0, 2.0, 0.05, 9.81, 1.0, 1.0], device=dev),
                      reward_kind=_hip.REWARD_PENDULUM, reward_params=torch.tensor([1.0, 0.02, 0.0], device=dev))
            o = torch.tensor([[1.0, 0.0, 0.0]], device=dev)
        else:
            dd = [X + U, X]
            kw = dict(system_kind=_hip.SYS_ENSEMBLE, dyn_params=torch.zeros(dd[0] * dd[1] + dd[1], device=dev),
                      dyn_spec=ops.MlpSpec(dd, "swish", 1), reward_kind=_hip.REWARD_QUADRATIC,
                      reward_params=torch.cat([torch.zeros(X), torch.ones(X), torch.ones(U)]).to(dev))
            o = torch.zeros(1, X, device=dev)
        params_f, spec_f = _bias_policy(c["logits"], X, dev)
        rows = ops.model_rollout(policy_params=params_f, policy_spec=spec_f, x_dim=X, u_dim=U, obs=o, first_obs=o.clone(),
                                 steps=torch.zeros(1, device=dev), done=torch.zeros(1, device=dev), n_steps=1, episode_length=10,
                                 ppo_extras=True, policy_noise=eps.reshape(1, 1, U), **kw).cpu().numpy()[0]
        np.testing.assert_allclose(rows[X:X + U], c["action"], atol=2e-6, rtol=2e-6, err_msg=c["why"])
        np.testing.assert_allclose(rows[2 * X + U + 2], c["log_prob"], rtol=2e-5, atol=2e-5, err_msg=c["why"])
        np.testing.assert_allclose(rows[2 * X + U + 3:2 * X + 2 * U + 3], c["z"], atol=2e-6, rtol=2e-6, err_msg=c["why"])


def test_queue_kat_through_the_ring(dev):
    """insert / roll / wrap-gather hand cases through mbpo_replay_insert / _gather: the ring's LOGICAL content and positions."""
    from mbpo.replay import UniformSamplingQueue
    from mbpo import ops
    from mbpo.types import Transition
    q = KAT["queue"]
    # a 1-column row is not a Transition: drive the row-level API directly
    data = torch.zeros(q["max_replay_size"], 1, device=dev)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    for step in q["steps"]:
        ops.replay_insert(data, state, torch.tensor(step["insert"], device=dev).reshape(-1, 1))
        st = state.cpu().tolist()
        assert (st[0], st[1]) == (step["insert_position"], step["sample_position"]), step["why"]
        logical = ops.replay_gather(data, state, torch.arange(q["max_replay_size"], dtype=torch.int32, device=dev))
        assert logical[:, 0].cpu().tolist() == step["data"], step["why"]
    got = ops.replay_gather(data, state, torch.tensor(q["gather"]["idx"], dtype=torch.int32, device=dev))
    assert got[:, 0].cpu().tolist() == q["gather"]["rows"]


def test_running_statistics_and_normalizer_kat(dev):
    """brax running_statistics.update (clip 1e-6..1e6) and BPTT's Normalizer.update (floor 1e-8) are the same kernels with
    different clips (include/mbpo_hip.h): both hand-case sets through mbpo_running_stats_*."""
    from mbpo import ops
    for c in KAT["running_stats"]:
        stats = torch.tensor([c["count"], c["mean"], c["summed_variance"], 1.0], device=dev)
        ops.running_stats_update(torch.tensor(c["batch"], device=dev).reshape(-1, 1), 0, 1, stats)
        np.testing.assert_allclose(stats.cpu().numpy(), [c["new_count"], c["new_mean"], c["new_summed_variance"], c["new_std"]],
                                   rtol=1e-6, atol=1e-7, err_msg=c["why"])
    for c in KAT["normalizer"]:
        stats = torch.tensor([float(c["size"]), c["mean"], c["std"] ** 2 * c["size"], c["std"]], device=dev)
        ops.running_stats_update(torch.tensor(c["x"], device=dev).reshape(-1, 1), 0, 1, stats, std_min=1e-8, std_max=float("inf"))
        got = stats.cpu().numpy()
        assert got[0] == c["new_size"], c["why"]
        np.testing.assert_allclose([got[1], got[3]], [c["new_mean"], c["new_std"]], rtol=1e-6, err_msg=c["why"])


def test_adamw_and_soft_update_kat(dev):
    from mbpo import ops
    from mbpo.utils.optimizer_utils import soft_update
    for c in KAT["adamw"]:
        opt = ops.AdamW(1, dev, c["lr"], c["wd"])
        opt.load_state(torch.tensor([c["m"]]), torch.tensor([c["v"]]), float(c["count"] - 1))
        p = torch.tensor([c["p"]], device=dev)
        opt.step(p, torch.tensor([c["g"]], device=dev))
        np.testing.assert_allclose([float(p), float(opt.m), float(opt.v)], [c["new_p"], c["new_m"], c["new_v"]], rtol=2e-6, err_msg=c["why"])
        assert float(opt.count) == c["count"]
    for c in KAT["soft_update"]:
        out = soft_update(torch.tensor(c["target"], device=dev), torch.tensor(c["online"], device=dev), c["tau"])
        np.testing.assert_allclose(out.cpu().numpy(), c["out"], rtol=1e-6, err_msg=c["why"])
    nested = soft_update({"a": (torch.ones(3, device=dev), torch.zeros(2, device=dev))}, {"a": (torch.zeros(3, device=dev), torch.ones(2, device=dev))}, 0.5)
    assert nested["a"][0].tolist() == [0.5] * 3 and nested["a"][1].tolist() == [0.5] * 2


def test_clip_by_global_norm_kat_through_sac_apply(dev):
    """optax.clip_by_global_norm inside mbpo_sac_apply: with lr tiny and one Adam step from zero moments the update direction
    is sign(g); instead check the clip through the first moment: m = 0.1 * clipped g."""
    from mbpo import ops
    X, U, B = 3, 1, 16
    for c in KAT["clip_by_global_norm"]:
        up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, 64, 64, 2], q_dims=[X + U, 64, 64, 1], batch_size=B, device=dev,
                            max_grad_norm=c["max_norm"])
        up.load_state(torch.zeros(up.NP, device=dev))
        # hand the optimizer a gradient whose policy group is the KAT vector (rest zero) via the all-reduce seam
        g = torch.zeros(up.NP, device=dev)
        g[0], g[1] = c["g"]

        def inject(t):
            t.copy_(g)
        up.all_reduce = inject
        batch = torch.zeros(B, 2 * X + U + 3, device=dev)
        z = torch.zeros(B, U, device=dev)
        up.sgd_step(batch, None, None, z, z, z)
        np.testing.assert_allclose(up.adam_m[:2].cpu().numpy() / 0.1, c["out"], rtol=1e-6, err_msg=c["why"])

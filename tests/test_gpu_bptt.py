"""GPU parity: BPTT actor gradient (B1-B5) — forward rollout through the model, target critics, lambda-return and the
hand-written reverse sweep, vs torch autograd THROUGH the model (oracle/bptt.py).

Tolerances (fp32): transitions / lambda-values 2e-4 (H-step rollouts feed rounding back through the dynamics); actor
gradient atol 5e-6 + rtol 2e-3 against the fp32 oracle and rtol 1e-3 against fp64 (the gradient is a sum over H chained
Jacobian products); losses 2e-5.
"""
import math

import numpy as np
import pytest
import torch

from oracle import bptt as obptt
from oracle import nets as onets
from oracle import systems as osys

pytestmark = pytest.mark.gpu


def _setup(X, U, H, n, system, E, seed, hidden=(64, 64, 64)):
    g = torch.Generator().manual_seed(seed)
    cfg = obptt.BpttConfig(x_dim=X, u_dim=U, actor_dims=[X, *hidden, 2 * U], critic_dims=[X, *hidden, 1], horizon=H,
                           discount=0.97, lambda_=0.9, ent_coef=0.05, init_stddev=1.0)
    ap = onets.init_mlp_flat(cfg.actor_dims, g) + 0.02 * torch.randn(cfg.P, generator=g)
    cp = torch.cat([onets.init_mlp_flat(cfg.critic_dims, g) + 0.02 * torch.randn(cfg.C, generator=g) for _ in range(2)])
    if X == 3:
        th = (torch.rand(n, generator=g) * 2 - 1) * math.pi
        x0 = torch.stack([torch.cos(th), torch.sin(th), (torch.rand(n, generator=g) * 2 - 1) * 4], 1)
    else:
        x0 = torch.randn(n, X, generator=g)
    noise = torch.randn(n, H, U, generator=g)
    s_mean, s_std = torch.randn(X, generator=g) * 0.2, torch.rand(X, generator=g) + 0.6
    r_ms = torch.tensor([-1.3, 2.1])
    extra = {}
    if system == "pendulum":
        tsys = obptt.TorchPendulumSystem()
    else:
        dd = [X + U, *hidden, 2 * X]
        dp = torch.cat([onets.init_mlp_flat(dd, g) * 0.5 + 0.01 * torch.randn(onets.n_params(dd), generator=g) for _ in range(E)])
        tgt, q, r = torch.randn(X, generator=g) * 0.3, torch.rand(X, generator=g), torch.rand(U, generator=g) * 0.2
        tsys = obptt.TorchEnsembleSystem(dp, dd, E, X, U, tgt, q, r)
        extra = dict(dd=dd, dp=dp, tgt=tgt, q=q, r=r)
    return cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, tsys, extra


def _run_hip(dev, cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, system, extra, n, explicit_noise=True, seed=0, offset=0):
    from mbpo import _hip, ops
    op = ops.BpttActorGrad(x_dim=cfg.x_dim, u_dim=cfg.u_dim, horizon=cfg.horizon, actor_dims=cfg.actor_dims,
                           critic_dims=cfg.critic_dims, n=n, device=dev, init_stddev=cfg.init_stddev, discount=cfg.discount,
                           lambda_=cfg.lambda_, ent_coef=cfg.ent_coef, seed=seed)
    kw = {}
    if system == "pendulum":
        pp = osys.PendulumParams()
        kw.update(system_kind=_hip.SYS_PENDULUM, reward_kind=_hip.REWARD_PENDULUM,
                  reward_params=torch.tensor(pp.reward_vector()).to(dev), sys_params=torch.tensor(pp.sys_vector()).to(dev))
    else:
        kw.update(system_kind=_hip.SYS_ENSEMBLE, reward_kind=_hip.REWARD_QUADRATIC,
                  reward_params=torch.cat([extra["tgt"], extra["q"], extra["r"]]).to(dev), dyn_params=extra["dp"].to(dev),
                  dyn_spec=ops.MlpSpec(extra["dd"], "swish", extra["dp"].numel() // onets.n_params(extra["dd"])))
    op(actor_params=ap.to(dev), target_critic_params=cp.to(dev), init_states=x0.to(dev), state_mean=s_mean.to(dev),
       state_std=s_std.to(dev), reward_mean_std=r_ms.to(dev), act_noise=noise.to(dev) if explicit_noise else None, offset=offset, **kw)
    torch.cuda.synchronize()
    return op


@pytest.mark.parametrize("X,U,H,n,system,E", [
    (3, 1, 10, 20, "pendulum", 0),       # the reference's BPTT test system (tests/test_bptt.py), ragged n
    (3, 1, 20, 16, "pendulum", 0),       # reference horizon 20
    (4, 1, 5, 48, "ensemble", 5),        # north-star shape, horizon 5
    (4, 2, 6, 17, "ensemble", 3),        # u=2 (sum-over-A log-prob), ragged n
    (17, 6, 8, 16, "ensemble", 10),      # BASELINE config 5 shape (shorter horizon for the oracle)
    (17, 6, 32, 16, "ensemble", 10),     # BASELINE config 5 at its FULL horizon H=32 (E=10, x=17, u=6), one tile of trajectories
])
def test_bptt_actor_grad_parity(dev, X, U, H, n, system, E):
    cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, tsys, extra = _setup(X, U, H, n, system, E, 0)
    g_ref, loss_ref, aux = obptt.actor_grads(cfg, tsys, ap, cp, x0, noise, s_mean, s_std, r_ms[0], r_ms[1])
    d = lambda t: t.double()
    tsys64 = obptt.TorchPendulumSystem() if system == "pendulum" else obptt.TorchEnsembleSystem(
        d(extra["dp"]), extra["dd"], E, X, U, d(extra["tgt"]), d(extra["q"]), d(extra["r"]))
    g64, loss64, aux64 = obptt.actor_grads(cfg, tsys64, d(ap), d(cp), d(x0), d(noise), d(s_mean), d(s_std), d(r_ms[0]), d(r_ms[1]))
    op = _run_hip(dev, cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, system, extra, n)
    rows = op.transitions.cpu().reshape(n, H, -1)
    torch.testing.assert_close(rows[..., :X], aux["observation"], atol=2e-4, rtol=2e-4)
    torch.testing.assert_close(rows[..., X:X + U], aux["action"], atol=2e-4, rtol=2e-4)
    torch.testing.assert_close(rows[..., X + U], aux["reward"], atol=5e-4, rtol=5e-4)
    assert torch.all(rows[..., X + U + 1] == 1.0)
    torch.testing.assert_close(rows[..., X + U + 2:], aux["next_observation"], atol=2e-4, rtol=2e-4)
    torch.testing.assert_close(op.lambda_values.cpu().reshape(n, H), aux["lambda_values"], atol=5e-4, rtol=5e-4)
    m = op.metrics.cpu().tolist()
    np.testing.assert_allclose(m[0], loss64, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(m[1], float(aux64["entropy_loss"]), rtol=1e-4, atol=2e-5)
    g = op.grads.cpu()
    torch.testing.assert_close(g, g_ref, atol=5e-6, rtol=2e-3)
    torch.testing.assert_close(g.double(), g64, atol=5e-6, rtol=1e-3)


def test_bptt_philox_noise_path(dev):
    from oracle import philox
    X, U, H, n = 3, 1, 6, 16
    cfg, ap, cp, x0, _, s_mean, s_std, r_ms, tsys, extra = _setup(X, U, H, n, "pendulum", 0, 1)
    seed, offset = 4242, 7
    noise = torch.from_numpy(philox.philox_normal(seed, offset, philox.STREAM_POLICY_NOISE, np.arange(n * H * U, dtype=np.uint64))).reshape(n, H, U)
    a = _run_hip(dev, cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, "pendulum", extra, n, explicit_noise=True)
    b = _run_hip(dev, cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, "pendulum", extra, n, explicit_noise=False, seed=seed, offset=offset)
    torch.testing.assert_close(a.grads, b.grads, atol=1e-6, rtol=1e-4)


def test_bptt_many_tiles_accumulate(dev):
    """n = 16 * 300 trajectories > number of workgroups: workgroups walk several tiles and accumulate into their slab;
    linearity check: the gradient of the mean over 2 copies of a batch equals the gradient of one copy."""
    X, U, H, n = 3, 1, 4, 16
    cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, tsys, extra = _setup(X, U, H, n, "pendulum", 0, 2)
    one = _run_hip(dev, cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, "pendulum", extra, n)
    reps = 300
    many = _run_hip(dev, cfg, ap, cp, x0.repeat(reps, 1), noise.repeat(reps, 1, 1), s_mean, s_std, r_ms, "pendulum", extra, n * reps)
    torch.testing.assert_close(many.grads, one.grads, atol=2e-6, rtol=2e-4)
    torch.testing.assert_close(many.metrics, one.metrics, atol=1e-5, rtol=1e-4)


def test_bptt_config5_full_size_replication(dev):
    """BASELINE configs[4] at full size — E=10, H=32, x=17, u=6, n=4096 initial states per GPU — through the size-independent
    property: the loss is a mean over trajectories, so 4096 = 256 copies of a 16-trajectory batch must give the gradient, the
    metrics, the simulated transitions and the lambda-values of ONE copy (which test_bptt_actor_grad_parity's H=32 case
    compares with the oracle).  Copies land in different workgroups / slabs: the cross-tile reduction is what is exercised."""
    X, U, H, n, E = 17, 6, 32, 16, 10
    cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, tsys, extra = _setup(X, U, H, n, "ensemble", E, 0)
    one = _run_hip(dev, cfg, ap, cp, x0, noise, s_mean, s_std, r_ms, "ensemble", extra, n)
    reps = 256
    many = _run_hip(dev, cfg, ap, cp, x0.repeat(reps, 1), noise.repeat(reps, 1, 1), s_mean, s_std, r_ms, "ensemble", extra, n * reps)
    assert many.transitions.shape[0] == 4096 * 32
    torch.testing.assert_close(many.grads, one.grads, atol=5e-6, rtol=5e-4)
    torch.testing.assert_close(many.metrics, one.metrics, atol=1e-5, rtol=1e-4)
    D = one.transitions.shape[1]
    assert torch.equal(many.transitions.reshape(reps, n * H, D), one.transitions.reshape(1, n * H, D).expand(reps, -1, -1))
    assert torch.equal(many.lambda_values.reshape(reps, n * H), one.lambda_values.reshape(1, n * H).expand(reps, -1))
    assert bool(torch.isfinite(many.grads).all())


# ------------------------------------------------------------------------------------------------ B4: critic regression
@pytest.mark.parametrize("X,R,batch,hidden,seed", [
    (3, 1000, 1000, (64, 64, 64), 0),     # the reference test's shape: n=50 x H=20 transitions, one critic update
    (3, 200, 67, (64, 64, 64), 1),        # ragged batch (ceil(200/3)), sampling with replacement
    (11, 320, 160, (64, 64), 2),          # wider observation, two hidden layers
    (5, 40, 7, (64,), 3),                 # less than one tile, one hidden layer
    (3, 16 * 700, 16 * 700, (64, 64, 64), 4),   # more tiles than workgroups: slabs accumulate
])
def test_critic_grads_parity(dev, X, R, batch, hidden, seed):
    """Loss and gradient vs torch autograd (oracle/bptt.py:critic_grads).  Tolerance: atol 2e-6 + rtol 5e-4 (fp32 sums over the
    batch in a different association order)."""
    from mbpo import ops
    g = torch.Generator().manual_seed(seed)
    cfg = obptt.BpttConfig(x_dim=X, u_dim=1, actor_dims=[X, 64, 2], critic_dims=[X, *hidden, 1])
    cp = torch.cat([onets.init_mlp_flat(cfg.critic_dims, g) + 0.02 * torch.randn(cfg.C, generator=g) for _ in range(2)])
    D = 2 * X + 1 + 2
    rows = torch.randn(R, D, generator=g)
    lam = torch.randn(R, generator=g) * 3
    idx = torch.randint(0, R, (batch,), generator=g, dtype=torch.int32)
    s_mean, s_std = torch.randn(X, generator=g) * 0.2, torch.rand(X, generator=g) + 0.6
    ref_g, ref_l = obptt.critic_grads(cfg, cp, rows[idx.long(), :X], lam[idx.long()], s_mean, s_std)
    op = ops.CriticGrad(x_dim=X, critic_dims=cfg.critic_dims, batch=batch, device=dev)
    got = op(cp.to(dev), rows.to(dev), lam.to(dev), idx.to(dev), s_mean.to(dev), s_std.to(dev))
    torch.cuda.synchronize()
    assert abs(float(op.metrics[0]) - ref_l) <= 1e-5 + 2e-5 * abs(ref_l)
    torch.testing.assert_close(got.cpu(), ref_g, atol=2e-6, rtol=5e-4)
    # fp64 oracle: bounds the error of the fp32 oracle itself
    g64, _ = obptt.critic_grads(cfg, cp.double(), rows[idx.long(), :X].double(), lam[idx.long()].double(), s_mean.double(), s_std.double())
    rel = float((got.cpu().double() - g64).norm() / g64.norm())
    assert rel < 2e-5, rel


def test_critic_grads_rejects_bad_shapes(dev):
    from mbpo import _hip, ops
    with pytest.raises(_hip.MbpoHipError):
        ops.CriticGrad(x_dim=3, critic_dims=[3, 32, 1], batch=8, device=dev)      # hidden width must be 64
    op = ops.CriticGrad(x_dim=3, critic_dims=[3, 64, 1], batch=8, device=dev)
    with pytest.raises(ValueError):
        op(torch.zeros(op.C * 2, device=dev), torch.zeros(4, 9, device=dev), torch.zeros(4, device=dev),
           torch.zeros(5, device=dev, dtype=torch.int32), torch.zeros(3, device=dev), torch.ones(3, device=dev))


# ------------------------------------------------------------------------------------------------ generic AdamW step
@pytest.mark.parametrize("n", [1, 255, 4673, 100_003])
def test_adamw_step_parity(dev, n):
    """5 chained apply_if_finite(adamw) steps with a Polyak target vs oracle.sac.adamw_step; the 3rd gradient holds a NaN and
    must be skipped entirely (params, moments, count).  Tolerance: moments 2e-6 rel (+1 ulp of the largest), params atol 3e-7 (fp32 pow/sqrt)."""
    from mbpo import ops
    g = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=g)
    tgt = p.clone()
    m, v, cnt = torch.zeros(n), torch.zeros(n), 0
    opt = ops.AdamW(n, dev, lr=3e-3, weight_decay=1e-2, apply_if_finite=True)
    dp, dt = p.to(dev), tgt.to(dev)
    tau = 0.05
    for it in range(5):
        gr = torch.randn(n, generator=g) * (10.0 ** (it - 2))
        if it == 2:
            gr[n // 2] = float("nan")
        p, m, v, cnt2 = obptt.apply_if_finite_adamw(p, gr, m, v, cnt, 3e-3, 1e-2)
        if cnt2 != cnt:
            tgt = (1 - tau) * tgt + tau * p
        cnt = cnt2
        opt.step(dp, gr.to(dev), target=dt, tau=tau)
        torch.cuda.synchronize()
        if it == 2:
            assert math.isnan(float(opt.grad_norm))
        else:
            assert abs(float(opt.grad_norm) - float(gr.norm())) <= 1e-5 * float(gr.norm())
        assert float(opt.count) == cnt
        # b1*m + 0.1*g may cancel: absolute slack of one fp32 ulp of the largest moment (hipcc contracts to FMA, torch does not)
        torch.testing.assert_close(opt.m.cpu(), m, atol=2e-7 * float(m.abs().max()), rtol=2e-6)
        torch.testing.assert_close(opt.v.cpu(), v, atol=2e-7 * float(v.abs().max()), rtol=2e-6)
        torch.testing.assert_close(dp.cpu(), p, atol=3e-7, rtol=1e-6)
        torch.testing.assert_close(dt.cpu(), tgt, atol=3e-7, rtol=1e-6)
    assert cnt == 4


def test_adamw_step_without_finite_guard_and_scale(dev):
    """grad_scale (the 1/world_size of a data-parallel mean) and the unguarded path."""
    from mbpo import ops
    from oracle.sac import adamw_step
    n = 1000
    g = torch.Generator().manual_seed(5)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    opt = ops.AdamW(n, dev, lr=1e-3, weight_decay=1e-5, apply_if_finite=False)
    dp = p.to(dev)
    opt.step(dp, (gr * 4).to(dev), grad_scale=0.25)
    p2, m2, v2 = adamw_step(p, gr, torch.zeros(n), torch.zeros(n), 1, 1e-3, 1e-5)
    torch.testing.assert_close(dp.cpu(), p2, atol=2e-7, rtol=1e-6)
    torch.testing.assert_close(opt.v.cpu(), v2, atol=1e-12, rtol=2e-6)


# ------------------------------------------------------------------------------------------------ BPTT Normalizer
def test_bptt_normalizer_matches_reference_form(dev):
    """Normalizer.update (bptt_optimizer.py:52-67) via mbpo_running_stats_* with the BPTT clip.  Three chained batches,
    the last one constant (std floor 1e-8 is NOT reached because earlier variance remains)."""
    import sys
    from mbpo.optimizers.policy_optimizers.bptt_optimizer import Normalizer
    g = torch.Generator().manual_seed(0)
    nz = Normalizer((4,))
    st = nz.initialize_normalizer_state(dev)
    mean, std, size = torch.zeros(4), torch.ones(4), 0
    for k, nrows in enumerate((1000, 37, 64)):
        x = torch.randn(nrows, 4, generator=g) * torch.tensor([1.0, 5.0, 0.1, 30.0]) + torch.tensor([0.0, 2.0, -1.0, 100.0])
        if k == 2:
            x = x[:1].repeat(nrows, 1)
        st = nz.update(x.to(dev), st)
        mean, std, size = obptt.normalizer_update(x, mean, std, size)
        assert float(st.size) == size
        torch.testing.assert_close(st.mean.cpu(), mean, atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(st.std.cpu(), std, atol=1e-5, rtol=2e-5)
    # degenerate: constant data from the start -> std clamps at 1e-8 (brax's clip would give 1e-6)
    st = nz.update(torch.full((8, 4), 2.5, device=dev), nz.initialize_normalizer_state(dev))
    assert torch.allclose(st.std.cpu(), torch.full((4,), 1e-8))
    assert torch.allclose(st.mean.cpu(), torch.full((4,), 2.5))


def test_bptt_actor_against_committed_golden(dev):
    """HIP actor gradient vs tests/golden/bptt_actor_small.npz (fp64 oracle on fixed inputs: pendulum, horizon 6, 16 trajectories,
    64x2 nets).  Rollout quantities 2e-4, lambda-returns 5e-4, losses rtol 1e-4, gradients atol 5e-6 + rtol 1e-3."""
    from pathlib import Path
    from mbpo import _hip, ops
    gold = np.load(Path(__file__).resolve().parent / "golden" / "bptt_actor_small.npz")
    X, U, H, n = 3, 1, 6, 16
    op = ops.BpttActorGrad(x_dim=X, u_dim=U, horizon=H, actor_dims=[X, 64, 64, 2 * U], critic_dims=[X, 64, 64, 1], n=n, device=dev,
                           init_stddev=1.0, discount=0.97, lambda_=0.9, ent_coef=0.05)
    f = lambda k: torch.from_numpy(gold[k]).float().to(dev).contiguous()
    pp = osys.PendulumParams()
    op(actor_params=f("actor_params"), target_critic_params=f("critic_params"), init_states=f("x0"), state_mean=f("state_mean"),
       state_std=f("state_std"), reward_mean_std=f("reward_mean_std"), act_noise=f("noise"), system_kind=_hip.SYS_PENDULUM,
       reward_kind=_hip.REWARD_PENDULUM, reward_params=torch.tensor(pp.reward_vector()).to(dev), sys_params=torch.tensor(pp.sys_vector()).to(dev))
    torch.cuda.synchronize()
    rows = op.transitions.cpu().double().reshape(n, H, -1)
    np.testing.assert_allclose(rows[..., X + U + 2:].numpy(), gold["next_observation"], atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(rows[..., X + U].numpy(), gold["reward"], atol=5e-4, rtol=5e-4)
    np.testing.assert_allclose(op.lambda_values.cpu().double().reshape(n, H).numpy(), gold["lambda_values"], atol=5e-4, rtol=5e-4)
    np.testing.assert_allclose(op.metrics.cpu().numpy(), gold["losses"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(op.grads.cpu().double(), torch.from_numpy(gold["grads"]), atol=5e-6, rtol=1e-3)

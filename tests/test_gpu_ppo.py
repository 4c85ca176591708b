"""GPU parity: PPO minibatch update (P1-P3) — HIP value pre-pass + GAE + advantage normalisation + hand-written
loss backward + AdamW, vs the torch-autograd oracle (oracle/ppo.py).

Tolerances (fp32): vs / advantages 2e-5; flat gradients atol 2e-6 + rtol 5e-4 vs the fp32 oracle and rtol 2e-4 vs fp64
(advantage normalisation divides by a minibatch std, which amplifies rounding slightly more than in SAC);
loss terms 1e-5; optimizer step 1e-7/1e-6 GIVEN the device gradient (see tests/test_gpu_sac.py for why).
"""
import numpy as np
import pytest
import torch

from oracle import nets as onets
from oracle import ppo as oppo

pytestmark = pytest.mark.gpu


def _make(X, U, hidden, B, T, seed, normalize, v_hidden=None, **kw):
    g = torch.Generator().manual_seed(seed)
    cfg = oppo.PpoConfig(x_dim=X, u_dim=U, policy_dims=[X, *hidden, 2 * U], value_dims=[X, *(hidden if v_hidden is None else v_hidden), 1], **kw)
    st = oppo.init_state(cfg, g)
    st.params = st.params + 0.03 * torch.randn(st.params.shape, generator=g)
    D = 2 * X + 2 * U + 4
    data = torch.randn(B, T, D, generator=g)
    o = X + U
    data[..., X:o] = torch.tanh(data[..., o + 3 + X:o + 3 + X + U])                 # action = tanh(raw_action)
    data[..., o + 1] = (torch.rand(B, T, generator=g) > 0.1).float()                # discount
    data[..., D - 1] = (torch.rand(B, T, generator=g) < 0.15).float()               # truncation
    data[..., o + 2 + X] = -1.0 + 0.5 * torch.randn(B, T, generator=g)              # behaviour log-prob
    noise = torch.randn(B, T, U, generator=g)
    nm = torch.randn(X, generator=g) * 0.3 if normalize else None
    ns = torch.rand(X, generator=g) + 0.5 if normalize else None
    return cfg, st, data, noise, nm, ns


def _updater(dev, cfg, B, T, **kw):
    from mbpo import ops
    return ops.PpoUpdater(x_dim=cfg.x_dim, u_dim=cfg.u_dim, policy_dims=cfg.policy_dims, value_dims=cfg.value_dims,
                          batch_size=B, unroll_length=T, device=dev, entropy_cost=cfg.entropy_cost, discounting=cfg.discounting,
                          reward_scaling=cfg.reward_scaling, gae_lambda=cfg.gae_lambda, clipping_epsilon=cfg.clipping_epsilon,
                          normalize_advantage=cfg.normalize_advantage, lr=cfg.lr, wd=cfg.wd, **kw)


@pytest.mark.parametrize("X,U,hidden,B,T,normalize,norm_adv", [
    (3, 1, (64, 64), 128, 40, True, True),        # reference test config (tests/test_ppo.py:30-56)
    (3, 1, (64, 64, 64), 32, 5, False, True),     # defaults (ppo.py:69-72), horizon-5 rollouts
    (4, 1, (64, 64, 64), 16, 10, True, False),    # no advantage normalisation
    (4, 2, (128, 128), 24, 7, True, True),        # 128-wide, ragged M = 168 (not a multiple of 16), u=2
    (17, 6, (64, 64), 8, 3, False, True),
    (3, 1, (64, 64, 64), 512, 40, True, True),    # BASELINE configs[2] at FULL size: B=512, T=40 (20 480 samples per minibatch)
    (4, 1, (64, 64, 64), 512, 5, True, True),     # the same at horizon-5 unrolls
    (3, 1, (64, 64), 2000, 3, True, True),        # k_ppo_values_gae with two trajectories per workgroup (B > 4 x CUs)
    (3, 1, (64, 64), 3, 1023, False, True),       # the longest trajectory a workgroup's LDS arrays hold (64 tiles, four at a time)
    (3, 1, (64, 64), 2, 1024, False, True),       # one step longer: values, GAE scan and moments as separate launches
])
def test_ppo_gradients_and_step(dev, X, U, hidden, B, T, normalize, norm_adv):
    cfg, st, data, noise, nm, ns = _make(X, U, hidden, B, T, 0, normalize, entropy_cost=1e-2, discounting=0.99,
                                          reward_scaling=0.5, gae_lambda=0.95, clipping_epsilon=0.3,
                                          normalize_advantage=norm_adv, lr=3e-4, wd=1e-5)
    g_ref, terms, vs_ref, adv_ref = oppo.grads(cfg, st.params, data, noise, nm, ns)
    d64 = lambda t: None if t is None else t.double()
    g_ref64, terms64, vs64, adv64 = oppo.grads(cfg, st.params.double(), data.double(), noise.double(), d64(nm), d64(ns))
    up = _updater(dev, cfg, B, T)
    up.load_state(st.params.to(dev))
    dd = lambda t: None if t is None else t.to(dev)
    up.minibatch_step(data.to(dev), dd(nm), dd(ns), noise.to(dev))
    torch.cuda.synchronize()
    g = up.grads.cpu()
    torch.testing.assert_close(g, g_ref, atol=2e-6, rtol=5e-4)
    torch.testing.assert_close(g.double(), g_ref64, atol=2e-6, rtol=2e-4)
    m = up.metrics.cpu().tolist()
    np.testing.assert_allclose(m, [terms64["total_loss"], terms64["policy_loss"], terms64["v_loss"], terms64["entropy_loss"]],
                               rtol=2e-5, atol=1e-5)
    st_new, _, _ = oppo.minibatch_step(cfg, st, data, noise, nm, ns, grad_override=g)
    torch.testing.assert_close(up.params.cpu(), st_new.params, atol=1e-7, rtol=1e-6)
    torch.testing.assert_close(up.adam_m.cpu(), st_new.adam_m, atol=1e-9, rtol=1e-5)
    torch.testing.assert_close(up.adam_v.cpu(), st_new.adam_v, atol=1e-12, rtol=1e-5)
    assert float(up.step_count.cpu()) == 1.0


@pytest.mark.parametrize("X,U,hidden,v_hidden,B,T,normalize,norm_adv", [
    (3, 1, (32, 32, 32, 32), (256,) * 5, 64, 10, True, True),      # experiments/train_inverted_pendulum/exp_ppo.py:36-38
    (4, 2, (48, 80), (200, 72, 40), 20, 7, True, False),           # unequal hidden layers, ragged M = 140
    (17, 6, (300,), (96,), 256, 12, False, True),                  # M = 3072 rows: weight gradients split over the rows
])
def test_ppo_layered_path_any_widths(dev, X, U, hidden, v_hidden, B, T, normalize, norm_adv):
    """Shapes outside the fused kernels' range (ppo.py:60-63 accepts any tuple): values pre-pass and loss forward/backward run layer
    by layer (csrc/ppo_layered.hip); GAE scan, moments, reduction and AdamW are the fused path's.  mbpo_ppo_grads + mbpo_ppo_apply
    and mbpo_ppo_step give the same bits."""
    cfg, st, data, noise, nm, ns = _make(X, U, hidden, B, T, 4, normalize, v_hidden=v_hidden, entropy_cost=1e-2, discounting=0.99,
                                          reward_scaling=0.5, gae_lambda=0.95, clipping_epsilon=0.3, normalize_advantage=norm_adv,
                                          lr=3e-4, wd=1e-5)
    d64 = lambda t: None if t is None else t.double()
    g_ref64, terms64, vs64, adv64 = oppo.grads(cfg, st.params.double(), data.double(), noise.double(), d64(nm), d64(ns))
    dd = lambda t: None if t is None else t.to(dev)
    outs = []
    for fused in ("0", "1"):
        up = _updater(dev, cfg, B, T)
        up.fused_step = fused == "1"
        up.load_state(st.params.to(dev))
        up.minibatch_step(data.to(dev), dd(nm), dd(ns), noise.to(dev))
        torch.cuda.synchronize()
        outs.append((up.grads.cpu().clone(), up.params.cpu().clone(), up.metrics.cpu().clone()))
        assert float(up.step_count.cpu()) == 1.0
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    g = outs[0][0]
    scale = float(g_ref64.abs().max())
    torch.testing.assert_close(g.double(), g_ref64, atol=2e-6 + 2e-6 * scale, rtol=5e-4)
    np.testing.assert_allclose(outs[0][2].tolist(), [terms64["total_loss"], terms64["policy_loss"], terms64["v_loss"], terms64["entropy_loss"]],
                               rtol=5e-5, atol=1e-5)
    st_new, _, _ = oppo.minibatch_step(cfg, st, data, noise, nm, ns, grad_override=g)
    torch.testing.assert_close(outs[0][1], st_new.params, atol=1e-7, rtol=1e-6)


def test_ppo_clip_branches(dev):
    """Behaviour log-probs far from the target ones push rho outside [1-eps, 1+eps] on both sides: the clipped branch
    (zero gradient) and the unclipped-but-smaller branch must both match autograd."""
    cfg, st, data, noise, nm, ns = _make(3, 1, (64, 64), 32, 8, 3, False, clipping_epsilon=0.1)
    X, U = 3, 1
    data[..., X + U + 2 + X] += torch.randn(32, 8, generator=torch.Generator().manual_seed(1)) * 1.5
    g_ref, _, _, _ = oppo.grads(cfg, st.params, data, noise)
    up = _updater(dev, cfg, 32, 8)
    up.load_state(st.params.to(dev))
    up.minibatch_step(data.to(dev), None, None, noise.to(dev))
    torch.testing.assert_close(up.grads.cpu(), g_ref, atol=2e-6, rtol=5e-4)


def test_ppo_chained_minibatches(dev):
    cfg, st, _, _, _, _ = _make(3, 1, (64, 64), 32, 10, 5, False, lr=1e-3, entropy_cost=1e-3)
    up = _updater(dev, cfg, 32, 10)
    up.load_state(st.params.to(dev))
    for k in range(10):
        _, _, data, noise, _, _ = _make(3, 1, (64, 64), 32, 10, 100 + k, False)
        st, terms, _ = oppo.minibatch_step(cfg, st, data, noise)
        up.minibatch_step(data.to(dev), None, None, noise.to(dev))
        np.testing.assert_allclose(up.metrics.cpu().tolist()[0], terms["total_loss"], rtol=2e-3, atol=2e-3)
    rel = float((up.params.cpu() - st.params).norm() / st.params.norm())
    assert rel < 1e-3 and float(up.step_count.cpu()) == 10.0


def test_ppo_bad_args(dev):
    from mbpo import ops, _hip
    with pytest.raises(_hip.MbpoHipError):      # the value net ends in one output
        ops.PpoUpdater(x_dim=3, u_dim=1, policy_dims=[3, 64, 64, 2], value_dims=[3, 64, 64, 2], batch_size=8, unroll_length=4, device=dev)
    # unequal hidden layers are not an error: the layered path (test_ppo_layered_path_any_widths)
    ops.PpoUpdater(x_dim=3, u_dim=1, policy_dims=[3, 64, 128, 2], value_dims=[3, 64, 64, 1], batch_size=8, unroll_length=4, device=dev)


def test_ppo_step_against_committed_golden(dev):
    """HIP minibatch gradient vs tests/golden/ppo_step_small.npz (fp64 oracle outputs on fixed inputs; one-hidden-layer nets,
    B=8, T=6).  Gradients atol 2e-6 + rtol 3e-4, loss terms 2e-5."""
    from pathlib import Path
    from mbpo import ops
    gold = np.load(Path(__file__).resolve().parent / "golden" / "ppo_step_small.npz")
    X, U, B, T = 3, 1, 8, 6
    up = ops.PpoUpdater(x_dim=X, u_dim=U, policy_dims=[X, 64, 2 * U], value_dims=[X, 64, 1], batch_size=B, unroll_length=T, device=dev,
                        entropy_cost=1e-2, discounting=0.97, reward_scaling=0.5, gae_lambda=0.9, clipping_epsilon=0.2,
                        normalize_advantage=True, lr=1e-3, wd=1e-4)
    f = lambda k: torch.from_numpy(gold[k]).float().to(dev).contiguous()
    up.load_state(f("params"))
    up.minibatch_step(f("data"), f("norm_mean"), f("norm_std"), f("noise"))
    torch.cuda.synchronize()
    torch.testing.assert_close(up.grads.cpu().double(), torch.from_numpy(gold["grads"]), atol=2e-6, rtol=3e-4)
    np.testing.assert_allclose(up.metrics.cpu().numpy(), gold["losses"], rtol=2e-5, atol=1e-5)


def test_ppo_fused_step_equals_grads_plus_apply(dev):
    """mbpo_ppo_step (the reduce launch applies AdamW to what it has just summed) against mbpo_ppo_grads + mbpo_ppo_apply on the same
    state and minibatches: four chained steps, BIT-identical parameters, moments, count, gradients and metrics — for the few-slab
    (one-stage sum) and the many-slab (two-stage sum, two workgroups per CU) shapes."""
    from mbpo import ops
    X, U = 4, 1
    for B, T in ((32, 5), (512, 40)):
        g = torch.Generator().manual_seed(B)
        pd, vd = [X, 64, 64, 2 * U], [X, 64, 64, 1]
        ups = []
        for fused in (True, False):
            up = ops.PpoUpdater(x_dim=X, u_dim=U, policy_dims=pd, value_dims=vd, batch_size=B, unroll_length=T, device=dev, lr=1e-3, wd=1e-4)
            up.fused_step = fused
            ups.append(up)
        init = torch.randn(ups[0].NPV, generator=g) * 0.1
        for up in ups:
            up.load_state(init.to(dev))
        D = 2 * X + 2 * U + 4
        for it in range(4):
            data = torch.randn(B, T, D, generator=g)
            data[..., X + U + 1] = 1.0                                   # discount
            data[..., D - 1] = (torch.rand(B, T, generator=g) < 0.05).float()
            data = data.to(dev)
            for up in ups:
                up.minibatch_step(data, offset=it << 32, seed=9)
        torch.cuda.synchronize()
        a, b = ups
        assert float(a.step_count) == float(b.step_count) == 4
        for name in ("params", "adam_m", "adam_v", "grads", "metrics"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (B, T, name)


@pytest.mark.parametrize("X,B,T,normalize,norm_adv,given_noise", [
    (3, 512, 40, True, True, True),       # BASELINE configs[2] at full size: 1280 tiles, 5 per workgroup
    (4, 512, 5, True, True, False),       # horizon-5 unrolls, in-kernel Philox entropy noise, one tile per workgroup
    (4, 37, 3, False, False, True),       # ragged: M = 111 rows, the last tile holds 15
    (2, 64, 10, True, True, False),       # the other observation widths the kernels are instantiated for
    (5, 128, 8, True, True, True),
    (6, 50, 7, False, True, False),
])
def test_ppo_lean_kernel_matches_generic_kernel(dev, X, B, T, normalize, norm_adv, given_noise, hidden=(64, 64, 64)):
    """k_ppo_lean (csrc/ppo_lean.hip: the loss forward/backward specialised for the 64x3 benchmark networks — weights resident in
    registers for all of a workgroup's tiles, weight gradients summed in registers, one slab per CU) against the generic k_ppo_fwd_bwd.
    Every per-tile number is formed by the same MFMA / FMA sequences; what differs is which tiles a slab sums (256 slabs instead of 512)
    — the cross-tile summation order — so gradients agree to fp32 summation-order accuracy, not bit for bit.  Oracle parity of the lean
    kernel itself: test_ppo_gradients_and_step above runs on it for these shapes."""
    import ctypes as C
    from mbpo import _hip
    lib = _hip.load()
    lib.mbpo_debug_set_ppo_lean.argtypes = [C.c_int]
    cfg, st, data, noise, nm, ns = _make(X, 1, hidden, B, T, 2, normalize, entropy_cost=1e-2, discounting=0.99, reward_scaling=0.5,
                                          gae_lambda=0.95, clipping_epsilon=0.3, normalize_advantage=norm_adv, lr=3e-4, wd=1e-5)
    d = lambda t: None if t is None else t.to(dev)
    outs = []
    try:
        for lean in (0, 1):
            lib.mbpo_debug_set_ppo_lean(lean)
            up = _updater(dev, cfg, B, T)
            up.load_state(st.params.to(dev))
            from mbpo import ops
            rng = ops.make_rng(dev, 3)
            if given_noise:
                up.minibatch_step(data.to(dev), d(nm), d(ns), noise.to(dev))
            else:
                up.minibatch_step(data.to(dev), d(nm), d(ns), seed=5, offset=9 << 32, rng_dev=rng)
            torch.cuda.synchronize()
            outs.append((up.grads.cpu().clone(), up.metrics.cpu().clone(), up.params.cpu().clone()))
    finally:
        lib.mbpo_debug_set_ppo_lean(-1)
    (g0, m0, p0), (g1, m1, p1) = outs
    assert float(g0.abs().sum()) > 0 and not torch.equal(g0, torch.zeros_like(g0))
    P = cfg.P
    for name, sl in (("policy", slice(0, P)), ("value", slice(P, None))):
        scale = float(g0[sl].abs().max())
        torch.testing.assert_close(g1[sl], g0[sl], atol=2e-6 * max(scale, 1e-3), rtol=2e-5, msg=lambda m: f"{name}: {m}")
    np.testing.assert_allclose(m1.numpy(), m0.numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("X,B,T", [(3, 128, 40), (4, 2000, 3), (5, 33, 9)])
def test_ppo_lean_kernels_at_two_hidden_layers(dev, X, B, T):
    """The reference's own PPO shapes (64 x 2: tests/test_ppo.py, experiments/playground_ppo_mpbo.py): k_ppo_vg_lean and k_ppo_lean<X, 1> against
    the generic launches, as at 64 x 3 (same per-tile arithmetic, another cross-tile summation order)."""
    test_ppo_lean_kernel_matches_generic_kernel(dev, X, B, T, True, True, False, hidden=(64, 64))

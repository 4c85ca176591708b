"""GPU parity: SAC sgd_step (S3-S8) — hand-written HIP forward/backward + clip/AdamW/Polyak vs torch-autograd oracle.

Tolerance (fp32): flat gradients agree to atol 2e-6 + rtol 2e-4 against the fp32 oracle and atol 2e-6 + rtol 1e-4 against the
fp64 oracle (gradient magnitudes are O(1e-3..1)); after one optimizer step parameters agree to 2e-6 absolute
(lr*O(1) updates); after 20 chained steps to 2e-4.
"""
import numpy as np
import pytest
import torch

from oracle import nets as onets
from oracle import sac as osac

pytestmark = pytest.mark.gpu


def _make(X, U, hidden, B, seed, normalize, trunc_p=0.2, q_hidden=None, **cfgkw):
    g = torch.Generator().manual_seed(seed)
    cfg = osac.SacConfig(x_dim=X, u_dim=U, policy_dims=[X, *hidden, 2 * U], q_dims=[X + U, *(hidden if q_hidden is None else q_hidden), 1],
                         **cfgkw)
    st = osac.init_state(cfg, g, init_log_alpha=-0.3)
    st.params = st.params + 0.03 * torch.randn(st.params.shape, generator=g)      # non-zero biases
    st.target_q = st.params[cfg.P:cfg.P + 2 * cfg.Q] + 0.02 * torch.randn(2 * cfg.Q, generator=g)
    D = 2 * X + U + 3
    batch = torch.randn(B, D, generator=g)
    batch[:, X:X + U] = torch.tanh(batch[:, X:X + U])
    batch[:, X + U + 1] = (torch.rand(B, generator=g) > 0.1).float()      # discount
    batch[:, D - 1] = (torch.rand(B, generator=g) < trunc_p).float()      # truncation
    noise = [torch.randn(B, U, generator=g) for _ in range(3)]
    nm = torch.randn(X, generator=g) * 0.3 if normalize else None
    ns = torch.rand(X, generator=g) + 0.5 if normalize else None
    return cfg, st, batch, noise, nm, ns


def _updater(dev, cfg, B, **kw):
    from mbpo import ops
    return ops.SacUpdater(x_dim=cfg.x_dim, u_dim=cfg.u_dim, policy_dims=cfg.policy_dims, q_dims=cfg.q_dims, batch_size=B,
                          device=dev, discounting=cfg.discounting, reward_scaling=cfg.reward_scaling,
                          target_entropy=cfg.target_entropy, tau=cfg.tau, lr_policy=cfg.lr_policy, lr_q=cfg.lr_q,
                          lr_alpha=cfg.lr_alpha, wd_policy=cfg.wd_policy, wd_q=cfg.wd_q, wd_alpha=cfg.wd_alpha,
                          max_grad_norm=cfg.max_grad_norm, **kw)


@pytest.mark.parametrize("X,U,hidden,B,normalize", [
    (4, 1, (64, 64, 64), 256, False),     # BASELINE config 2 networks (defaults sac.py:84-88)
    (3, 1, (64, 64, 64), 256, True),
    (3, 1, (128, 128, 128), 64, True),    # reference test config (tests/test_sac.py:30-57)
    (4, 1, (64, 64), 32, False),          # exp.py-like B=32
    (17, 6, (64, 64, 64), 48, True),      # config 5 shape
    (4, 2, (64, 64, 64), 40, False),      # ragged: B not a multiple of 16
])
def test_sac_gradients_and_step(dev, X, U, hidden, B, normalize):
    cfg, st, batch, noise, nm, ns = _make(X, U, hidden, B, 0, normalize, discounting=0.99, reward_scaling=1.5,
                                           lr_policy=3e-4, lr_q=3e-4, lr_alpha=3e-4, wd_q=1e-3)
    g_ref, (cl, ac, al) = osac.grads(cfg, st.params, st.target_q, batch, *noise, nm, ns)
    to64 = lambda t: None if t is None else t.double()
    g_ref64, (cl64, ac64, al64) = osac.grads(cfg, st.params.double(), st.target_q.double(), batch.double(),
                                             *[n.double() for n in noise], to64(nm), to64(ns))
    up = _updater(dev, cfg, B)
    up.load_state(st.params.to(dev), st.target_q.to(dev))
    d = lambda t: None if t is None else t.to(dev)
    up.sgd_step(batch.to(dev), d(nm), d(ns), *[n.to(dev) for n in noise])
    torch.cuda.synchronize()
    g = up.grads.cpu()
    # optimizer parity is checked GIVEN the device gradient: Adam's first step g/(|g|+eps) turns a 1e-9 difference on a
    # near-zero gradient element into an O(lr) parameter difference, which says nothing about either implementation
    st_new, met, _ = osac.sgd_step(cfg, st, batch, *noise, nm, ns, grad_override=g)
    P, Q = cfg.P, cfg.Q
    for name, sl in (("policy", slice(0, P)), ("critic", slice(P, P + 2 * Q)), ("alpha", slice(P + 2 * Q, None))):
        torch.testing.assert_close(g[sl], g_ref[sl], atol=2e-6, rtol=2e-4, msg=lambda m: f"{name} grad vs fp32 oracle: {m}")
        torch.testing.assert_close(g[sl].double(), g_ref64[sl], atol=2e-6, rtol=1e-4,
                                   msg=lambda m: f"{name} grad vs fp64 oracle: {m}")
    m = up.metrics.cpu().tolist()
    np.testing.assert_allclose(m[:3], [cl64, ac64, al64], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(m[3], met["alpha"], rtol=1e-6)
    torch.testing.assert_close(up.params.cpu(), st_new.params, atol=1e-7, rtol=1e-6)
    torch.testing.assert_close(up.target_q.cpu(), st_new.target_q, atol=1e-7, rtol=1e-6)
    torch.testing.assert_close(up.adam_m.cpu(), st_new.adam_m, atol=1e-9, rtol=1e-5)
    torch.testing.assert_close(up.adam_v.cpu(), st_new.adam_v, atol=1e-12, rtol=1e-5)
    assert float(up.step_count.cpu()) == 1.0


@pytest.mark.parametrize("X,U,hidden,q_hidden,B,normalize", [
    (3, 1, (256, 256, 256), (256, 256, 256), 256, True),     # wider than the fused kernels take
    (4, 2, (48, 80), (200, 72, 40), 100, True),              # unequal hidden layers, ragged batch, different depths
    (17, 6, (96,), (300,), 3000, False),                     # one hidden layer; 3000 rows: weight gradients split over the rows
    (4, 1, (256,) * 5, (256,) * 5, 64, True),                # the depth experiments/train_inverted_pendulum/exp_ppo.py gives its critic
])
def test_sac_layered_path_any_widths(dev, X, U, hidden, q_hidden, B, normalize):
    """Shapes outside the fused kernel's range (sac.py:84-88 accepts any tuple): the forward/backward half runs layer by layer
    (csrc/sac_layered.hip, one fp32-MFMA GEMM launch per Dense layer), everything behind it is the fused path's code.  Same
    tolerances as test_sac_gradients_and_step."""
    cfg, st, batch, noise, nm, ns = _make(X, U, hidden, B, 5, normalize, q_hidden=q_hidden, discounting=0.99, reward_scaling=1.5,
                                           lr_policy=3e-4, lr_q=3e-4, lr_alpha=3e-4, wd_q=1e-3)
    to64 = lambda t: None if t is None else t.double()
    g_ref64, (cl64, ac64, al64) = osac.grads(cfg, st.params.double(), st.target_q.double(), batch.double(),
                                             *[n.double() for n in noise], to64(nm), to64(ns))
    d = lambda t: None if t is None else t.to(dev)
    outs = []
    for two_launch in (False, True):        # mbpo_sac_grads + mbpo_sac_apply, and mbpo_sac_step (defined to equal them bit for bit)
        up = _updater(dev, cfg, B, two_launch=two_launch)
        up.load_state(st.params.to(dev), st.target_q.to(dev))
        up.sgd_step(batch.to(dev), d(nm), d(ns), *[n.to(dev) for n in noise])
        up.finalize()
        torch.cuda.synchronize()
        outs.append((up.grads.cpu().clone(), up.params.cpu().clone(), up.target_q.cpu().clone(), up.metrics.cpu().clone()))
        assert float(up.step_count.cpu()) == 1.0
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    g = outs[0][0]
    P, Q = cfg.P, cfg.Q
    for name, sl in (("policy", slice(0, P)), ("critic", slice(P, P + 2 * Q)), ("alpha", slice(P + 2 * Q, None))):
        scale = float(g_ref64[sl].abs().max())
        torch.testing.assert_close(g[sl].double(), g_ref64[sl], atol=2e-6 + 2e-6 * scale, rtol=2e-4,
                                   msg=lambda m: f"{name} grad vs fp64 oracle: {m}")
    st_new, met, _ = osac.sgd_step(cfg, st, batch, *noise, nm, ns, grad_override=g)
    np.testing.assert_allclose(outs[0][3].tolist()[:3], [cl64, ac64, al64], rtol=5e-5, atol=2e-6)
    np.testing.assert_allclose(outs[0][3].tolist()[3], met["alpha"], rtol=1e-6)
    torch.testing.assert_close(outs[0][1], st_new.params, atol=1e-7, rtol=1e-6)
    torch.testing.assert_close(outs[0][2], st_new.target_q, atol=1e-7, rtol=1e-6)


def test_sac_layered_path_equals_fused_path_on_a_padded_network(dev):
    """The same logical 60-wide networks twice: zero-padded to 64 through the fused kernel (what the trainer does, ops.py 'hidden-width
    padding') and unpadded through the layered path (60 is not a kernel width).  Gradients of the logical entries agree."""
    from mbpo import ops
    X, U, B = 4, 1, 96
    cfg, st, batch, noise, nm, ns = _make(X, U, (60, 60, 60), B, 8, True, discounting=0.95)
    cfg_pad = osac.SacConfig(x_dim=X, u_dim=U, policy_dims=ops.padded_dims(cfg.policy_dims, 64), q_dims=ops.padded_dims(cfg.q_dims, 64),
                             discounting=0.95)
    P, Q = cfg.P, cfg.Q
    pol = ops.embed_mlp_params(st.params[:P], cfg.policy_dims, 64)
    q = ops.embed_mlp_params(st.params[P:P + 2 * Q], cfg.q_dims, 64, n_nets=2)
    tq = ops.embed_mlp_params(st.target_q, cfg.q_dims, 64, n_nets=2)
    up_l, up_f = _updater(dev, cfg, B), _updater(dev, cfg_pad, B)
    up_l.load_state(st.params.to(dev), st.target_q.to(dev))
    up_f.load_state(torch.cat([pol, q, st.params[-1:]]).to(dev), tq.to(dev))
    for up in (up_l, up_f):
        up.sgd_step(batch.to(dev), nm.to(dev), ns.to(dev), *[n.to(dev) for n in noise])
        up.finalize()
    torch.cuda.synchronize()
    gf = up_f.grads.cpu()
    g_fused = torch.cat([ops.extract_mlp_params(gf[:cfg_pad.P], cfg.policy_dims, 64),
                         ops.extract_mlp_params(gf[cfg_pad.P:cfg_pad.P + 2 * cfg_pad.Q], cfg.q_dims, 64, n_nets=2), gf[-1:]])
    torch.testing.assert_close(up_l.grads.cpu(), g_fused, atol=1e-6, rtol=1e-4)
    torch.testing.assert_close(up_l.metrics.cpu(), up_f.metrics.cpu(), atol=1e-6, rtol=2e-5)


@pytest.mark.parametrize("X,U,hidden,B", [
    (4, 1, (64, 64, 64), 40),          # ragged tile (B not a multiple of 16)
    (7, 1, (64, 64), 256),             # critic input x + u = 8: the widest layer the VALU form takes; one H x H layer per net
    (3, 1, (64, 64, 64, 64), 48),      # an odd number of H x H layers (the runners' other register image holds the last one)
])
def test_sac_thin_layer_variant_matches_mfma_layers(dev, monkeypatch, X, U, hidden, B):
    """k_sac_fwd_bwd<64,4,false,2,true> ('thin layers by VALU': layer 0 of the forward chains, the output layer's dgrad/wgrad and
    layer 0's wgrad as plain FMAs around the runners) against the same launch with every layer on MFMA
    (MBPO_SAC_THIN=0) and against the fp32 oracle: gradients, metrics and the optimizer step."""
    cfg, st, batch, noise, nm, ns = _make(X, U, hidden, B, 2, True, discounting=0.97, reward_scaling=0.7, lr_policy=3e-4, lr_q=3e-4,
                                           lr_alpha=3e-4, wd_q=1e-3)
    g_ref, (cl, ac, al) = osac.grads(cfg, st.params, st.target_q, batch, *noise, nm, ns)
    out = {}
    for thin in ("1", "0"):
        monkeypatch.setenv("MBPO_SAC_THIN", thin)
        up = _updater(dev, cfg, B)
        up.load_state(st.params.to(dev), st.target_q.to(dev))
        up.sgd_step(batch.to(dev), nm.to(dev), ns.to(dev), *[n.to(dev) for n in noise])
        torch.cuda.synchronize()
        out[thin] = (up.grads.cpu().clone(), up.params.cpu().clone(), up.metrics.cpu().clone())
    torch.testing.assert_close(out["1"][0], out["0"][0], atol=5e-7, rtol=2e-5)       # same arithmetic, another summation order
    torch.testing.assert_close(out["1"][1], out["0"][1], atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(out["1"][2], out["0"][2], atol=1e-6, rtol=2e-5)
    torch.testing.assert_close(out["1"][0], g_ref, atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(out["1"][2].tolist()[:3], [cl, ac, al], rtol=5e-5, atol=2e-6)


def test_sac_clip_triggers(dev):
    """max_grad_norm small enough to clip every group (the reference default 1e5 never does)."""
    cfg, st, batch, noise, nm, ns = _make(4, 1, (64, 64, 64), 64, 1, False, max_grad_norm=1e-3, lr_policy=1e-3, lr_q=1e-3,
                                           lr_alpha=1e-3)
    up = _updater(dev, cfg, 64)
    up.load_state(st.params.to(dev), st.target_q.to(dev))
    up.sgd_step(batch.to(dev), None, None, *[n.to(dev) for n in noise])
    g = up.grads.cpu()
    for sl in (slice(0, cfg.P), slice(cfg.P, cfg.P + 2 * cfg.Q), slice(cfg.P + 2 * cfg.Q, None)):
        assert float(torch.sqrt((g[sl] ** 2).sum())) > 1e-3          # every group really clips
    st_new, _, _ = osac.sgd_step(cfg, st, batch, *noise, grad_override=g)
    torch.testing.assert_close(up.params.cpu(), st_new.params, atol=1e-7, rtol=1e-6)
    torch.testing.assert_close(up.adam_m.cpu(), st_new.adam_m, atol=1e-10, rtol=1e-5)


def test_sac_all_truncated_rows_give_zero_critic_grad(dev):
    cfg, st, batch, noise, nm, ns = _make(3, 1, (64, 64, 64), 32, 2, False, trunc_p=1.1)
    up = _updater(dev, cfg, 32)
    up.load_state(st.params.to(dev), st.target_q.to(dev))
    up.sgd_step(batch.to(dev), None, None, *[n.to(dev) for n in noise])
    g = up.grads.cpu()
    assert torch.count_nonzero(g[cfg.P:cfg.P + 2 * cfg.Q]) == 0
    assert float(up.metrics.cpu()[0]) == 0.0


def test_sac_chained_steps(dev):
    """20 chained sgd_steps (fresh minibatch + noise each) against the independent oracle trajectory: relative L2
    distance of the parameter vectors < 1e-3 and loss metrics within 1e-3 (element-wise closeness is not meaningful:
    Adam's g/(sqrt(v)+eps) amplifies rounding on near-zero-gradient elements)."""
    X, U, B = 4, 1, 64
    cfg, st, _, _, _, _ = _make(X, U, (64, 64, 64), B, 3, False, lr_policy=1e-3, lr_q=1e-3, lr_alpha=1e-3, discounting=0.95)
    up = _updater(dev, cfg, B)
    up.load_state(st.params.to(dev), st.target_q.to(dev))
    g = torch.Generator().manual_seed(11)
    D = 2 * X + U + 3
    for k in range(20):
        batch = torch.randn(B, D, generator=g)
        batch[:, X + U + 1] = 1.0
        batch[:, D - 1] = (torch.rand(B, generator=g) < 0.2).float()
        noise = [torch.randn(B, U, generator=g) for _ in range(3)]
        st, met, _ = osac.sgd_step(cfg, st, batch, *noise)
        up.sgd_step(batch.to(dev), None, None, *[n.to(dev) for n in noise])
        m = up.metrics.cpu().tolist()
        np.testing.assert_allclose(m, [met["critic_loss"], met["actor_loss"], met["alpha_loss"], met["alpha"]],
                                   rtol=1e-3, atol=1e-3)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(up.params.cpu(), st.params) < 1e-3
    assert rel(up.target_q.cpu(), st.target_q) < 1e-3
    assert float(up.step_count.cpu()) == 20.0


def test_sac_philox_noise_path(dev):
    """NULL noise pointers -> device Philox; must equal feeding oracle/philox.py draws explicitly."""
    from oracle import philox
    cfg, st, batch, _, _, _ = _make(4, 1, (64, 64, 64), 64, 4, False)
    seed, offset, B, U = 777, 13, 64, 1
    idx = np.arange(B * U, dtype=np.uint64)
    noise = [torch.from_numpy(philox.philox_normal(seed, offset, s, idx)).reshape(B, U)
             for s in (philox.STREAM_SAC_ALPHA, philox.STREAM_SAC_CRITIC, philox.STREAM_SAC_ACTOR)]
    outs = []
    for explicit in (True, False):
        up = _updater(dev, cfg, B, seed=seed)
        up.load_state(st.params.to(dev), st.target_q.to(dev))
        if explicit:
            up.sgd_step(batch.to(dev), None, None, *[n.to(dev) for n in noise])
        else:
            up.sgd_step(batch.to(dev), offset=offset)
        outs.append(up.grads.cpu())
    torch.testing.assert_close(outs[0], outs[1], atol=1e-6, rtol=1e-4)


def test_sac_bad_args(dev):
    from mbpo import ops, _hip
    with pytest.raises(_hip.MbpoHipError):      # the policy must end in 2u outputs
        ops.SacUpdater(x_dim=3, u_dim=1, policy_dims=[3, 64, 64, 3], q_dims=[4, 64, 64, 1], batch_size=32, device=dev)
    with pytest.raises(_hip.MbpoHipError):      # a critic takes x + u inputs
        ops.SacUpdater(x_dim=3, u_dim=1, policy_dims=[3, 64, 64, 2], q_dims=[3, 64, 64, 1], batch_size=32, device=dev)
    with pytest.raises(_hip.MbpoHipError):      # at least one hidden layer
        ops.SacUpdater(x_dim=3, u_dim=1, policy_dims=[3, 2], q_dims=[4, 1], batch_size=32, device=dev)
    # unequal or wide hidden layers are NOT errors: they take the layered path (test_sac_layered_path_any_widths)
    ops.SacUpdater(x_dim=3, u_dim=1, policy_dims=[3, 64, 32, 2], q_dims=[4, 256, 256, 1], batch_size=32, device=dev)


@pytest.mark.parametrize("max_norm,mode", [(0.05, "finalize_each"), (0.05, "deferred"), (1e5, "deferred"), (0.4, "deferred")])
def test_two_launch_step_matches_three_launch_path(dev, max_norm, mode):
    """mbpo_sac_step (fwd/bwd + ONE launch: slab reduction + unclipped optimizer step + undo log; the clip check resolved by the
    next step's prologue or by mbpo_sac_finalize) vs mbpo_sac_grads + mbpo_sac_apply on the same state and batches: 6 chained
    steps, BIT-identical state.  max_norm 0.05: every group clips every step (the fix-up path runs each time, in the next
    launch's prologue when deferred, in k_sac_finalize otherwise); 0.4: only some groups / steps clip; 1e5: never."""
    from mbpo import ops
    X, U, B = 4, 1, 256
    g = torch.Generator().manual_seed(3)
    mk = lambda two: ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, 64, 64, 64, 2 * U], q_dims=[X + U, 64, 64, 64, 1], batch_size=B,
                                    device=dev, seed=5, max_grad_norm=max_norm, lr_policy=1e-3, lr_q=1e-3, lr_alpha=1e-3, wd_q=1e-3,
                                    two_launch=two)
    ups = [mk(True), mk(False)]
    init = torch.randn(ups[0].NP, generator=g) * 0.1
    for up in ups:
        up.load_state(init.to(dev))
    D = 2 * X + U + 3
    clipped = 0
    for it in range(6):
        batch = torch.randn(B, D, generator=g).to(dev)
        batch[:, -1] = (torch.rand(B, generator=g) < 0.1).float().to(dev)
        ups[0].sgd_step(batch, offset=it, defer_clip_check=(mode == "deferred"))
        ups[1].sgd_step(batch, offset=it)
        gn = float(ups[1].grads[:ups[1].P].norm())
        clipped += gn >= max_norm
    ups[0].finalize()
    ups[0].finalize()          # idempotent
    torch.cuda.synchronize()
    a, b = ups
    assert float(a.step_count) == float(b.step_count) == 6
    for name in ("params", "target_q", "adam_m", "adam_v", "grads", "metrics"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert clipped == (6 if max_norm == 0.05 else 0 if max_norm == 1e5 else clipped)
    print('policy-group clips:', clipped, 'of 6 at max_norm', max_norm)
    # the device counter of clip events (mbpo_sac_control_offset word 13): both flavours count the same steps — those in which
    # SOME group clipped (>= the policy-group count formed above)
    ev = [up.clip_events() for up in ups]
    assert ev[0] == ev[1], ev
    assert ev[0] == (6 if max_norm == 0.05 else 0 if max_norm == 1e5 else ev[0]) and ev[0] >= clipped
    torch.testing.assert_close(a.metrics_accum, b.metrics_accum, atol=0, rtol=0)


@pytest.mark.parametrize("max_norm", [0.05, 0.4, float("inf")])
def test_mixed_two_and_three_launch_steps_on_one_state(dev, max_norm):
    """the header allows mbpo_sac_step -> mbpo_sac_grads + mbpo_sac_apply -> ... -> mbpo_sac_finalize on ONE
    state without a finalize in between.  The three-launch step's fwd/bwd prologue resolves the pending speculative step; its
    reduce launch must RECORD that (seq[1] = seq[0]) because it overwrites the clip-norm partials — otherwise the final finalize
    repeats the old step's check on the new step's partials and reverts parameters from a stale undo log.  Compared bit for bit
    with the pure three-launch path; max_norm 0.05 clips every step, 0.4 some, inf none (and inf makes the quick-check limits
    overflow: the flag must be +inf for a non-finite gradient to fail `flag < limit`)."""
    from mbpo import ops, _hip
    import ctypes as C
    X, U, B = 4, 1, 256
    g = torch.Generator().manual_seed(11)
    mk = lambda two: ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, 64, 64, 64, 2 * U], q_dims=[X + U, 64, 64, 64, 1], batch_size=B,
                                    device=dev, seed=5, max_grad_norm=max_norm, lr_policy=1e-3, lr_q=1e-3, lr_alpha=1e-3, two_launch=two)
    mixed, ref = mk(True), mk(False)
    init = torch.randn(ref.NP, generator=g) * 0.1
    for up in (mixed, ref):
        up.load_state(init.to(dev))
    D = 2 * X + U + 3
    # flavour per step on the mixed updater: T = two-launch deferred, 3 = grads + apply through the raw entry points (no finalize)
    plan = "T3T33TT3"
    for it, kind in enumerate(plan):
        batch = torch.randn(B, D, generator=g).to(dev)
        if max_norm == float("inf") and it == 4:
            batch[0, 0] = float("inf")          # an overflowing gradient: clip_by_global_norm gives g/inf*inf = NaN on both paths
        batch[:, -1] = (torch.rand(B, generator=g) < 0.1).float().to(dev)
        ref.sgd_step(batch, offset=it)
        if kind == "T":
            mixed.sgd_step(batch, offset=it, defer_clip_check=True)
        else:
            d = mixed.desc
            d.batch, d.offset = batch.data_ptr(), it
            d.norm_mean = d.norm_std = d.noise_alpha = d.noise_critic = d.noise_actor = None
            st = _hip.current_stream_ptr()
            _hip.check(mixed.lib.mbpo_sac_grads(C.byref(d), st), "mbpo_sac_grads")
            _hip.check(mixed.lib.mbpo_sac_apply(C.byref(d), st), "mbpo_sac_apply")
    mixed.finalize()
    torch.cuda.synchronize()
    assert float(mixed.step_count) == float(ref.step_count) == len(plan)
    for name in ("params", "target_q", "adam_m", "adam_v", "grads"):
        a, b = getattr(mixed, name), getattr(ref, name)
        assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0)), name
        assert torch.equal(torch.isnan(a), torch.isnan(b)), name
    if max_norm != float("inf"):
        assert bool(torch.isfinite(ref.params).all())
        assert mixed.clip_events() == ref.clip_events()


@pytest.mark.parametrize("U", [2, 1])        # u = 1: the forward-mode actor / thin-layer kernel variant
def test_sac_non_equidistant_time_target(dev, U):
    """N1 (sac/losses.py:90-98): the critic target's discount is exp(-continuous_discounting * t) per sample, t decoded from the
    last action component, affinely mapped to [min, max]_time_between_switches and floored to multiples of env_dt."""
    X, B = 4, 96
    kw = dict(non_equidistant_time=True, continuous_discounting=0.9, min_time_between_switches=0.05, max_time_between_switches=0.75,
              env_dt=0.05)
    cfg, st, batch, noise, nm, ns = _make(X, U, (64, 64, 64), B, 4, True, reward_scaling=1.5, lr_policy=3e-4, lr_q=3e-4,
                                           lr_alpha=3e-4, **kw)
    g_ref, (cl, ac, al) = osac.grads(cfg, st.params, st.target_q, batch, *noise, nm, ns)
    cfg_eq = osac.SacConfig(**{**cfg.__dict__, "non_equidistant_time": False})
    g_eq, _ = osac.grads(cfg_eq, st.params, st.target_q, batch, *noise, nm, ns)
    P, Q = cfg.P, cfg.Q
    assert float((g_ref[P:P + 2 * Q] - g_eq[P:P + 2 * Q]).abs().max()) > 1e-4          # the option really changes the critic loss
    up = _updater(dev, cfg, B, **kw)
    up.load_state(st.params.to(dev), st.target_q.to(dev))
    up.sgd_step(batch.to(dev), nm.to(dev), ns.to(dev), *[n.to(dev) for n in noise])
    torch.cuda.synchronize()
    g = up.grads.cpu()
    torch.testing.assert_close(g, g_ref, atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(up.metrics.cpu().tolist()[:3], [cl, ac, al], rtol=5e-5, atol=2e-6)


def test_sac_step_against_committed_golden(dev):
    """HIP sgd_step vs tests/golden/sac_step_small.npz (fp64 oracle outputs on fixed inputs; one-hidden-layer nets, B = 16).
    Gradients atol 2e-6 + rtol 2e-4; the optimizer step is checked GIVEN the device gradient elsewhere, here end to end with
    a tolerance that allows Adam's first-step sign amplification on near-zero gradient elements (lr = 1e-3..2e-3)."""
    from pathlib import Path
    from mbpo import ops
    gold = np.load(Path(__file__).resolve().parent / "golden" / "sac_step_small.npz")
    X, U, B = 3, 1, 16
    up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, 64, 2 * U], q_dims=[X + U, 64, 1], batch_size=B, device=dev, discounting=0.97,
                        reward_scaling=2.0, lr_policy=1e-3, lr_q=2e-3, lr_alpha=5e-4, wd_q=1e-3, max_grad_norm=0.5)
    f = lambda k: torch.from_numpy(gold[k]).float().to(dev).contiguous()
    up.load_state(f("params"), f("target_q"))
    up.sgd_step(f("batch"), f("norm_mean"), f("norm_std"), f("noise_alpha"), f("noise_critic"), f("noise_actor"))
    torch.cuda.synchronize()
    torch.testing.assert_close(up.grads.cpu().double(), torch.from_numpy(gold["grads"]), atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(up.metrics.cpu().numpy()[:3], gold["losses"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(float(up.metrics[3]), float(gold["alpha"]), rtol=1e-5)
    d = (up.params.cpu().double() - torch.from_numpy(gold["new_params"])).abs()
    assert float(d.max()) <= 4.1e-3 and float((d > 1e-5).float().mean()) < 0.01      # at most a few sign flips of size 2*lr
    torch.testing.assert_close(up.target_q.cpu().double(), torch.from_numpy(gold["new_target_q"]), atol=5e-5, rtol=0)


# ------------------------------------------------------------------------------------------------ hidden-width padding
def test_embed_extract_roundtrip_cpu_side():
    from mbpo import ops
    g = torch.Generator().manual_seed(0)
    dims = [5, 32, 48, 32, 3]
    flat = torch.randn(2 * sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(4)), generator=g)
    pad = ops.embed_mlp_params(flat, dims, 64, n_nets=2)
    assert pad.numel() == 2 * ops.MlpSpec(ops.padded_dims(dims, 64)).n_params
    assert torch.equal(ops.extract_mlp_params(pad, dims, 64, n_nets=2), flat)
    assert ops.common_width((32, 32), (64,)) == 64 and ops.common_width((100,), ()) == 128
    with pytest.raises(Exception):
        ops.common_width((256,))


@pytest.mark.parametrize("ph,qh", [((32, 32, 32, 32), (64, 64, 64)), ((48, 48), (100, 100, 100)), ((64, 64, 64), (32, 32))])
def test_sac_padded_widths_match_the_logical_networks(dev, ph, qh):
    """the reference accepts any hidden sizes (exp_ppo.py: policy (32,)*4 beside critic (256,)*5).  Unequal / narrow
    hidden layers are zero-padded to one kernel width; the padded networks ARE the logical ones: 5 chained sgd_steps on the
    padded HIP state vs the oracle on the LOGICAL shapes (rel L2 like test_sac_chained_steps), and every padded entry of params,
    moments and target stays exactly zero."""
    from mbpo import ops
    X, U, B = 4, 1, 64
    pd, qd = [X, *ph, 2 * U], [X + U, *qh, 1]
    W = ops.common_width(ph, qh)
    cfg = osac.SacConfig(X, U, pd, qd, lr_policy=1e-3, lr_q=1e-3, lr_alpha=1e-3, wd_q=1e-3)
    g = torch.Generator().manual_seed(1)
    st = osac.init_state(cfg, g)
    P, Q = cfg.P, cfg.Q
    emb = lambda flat, dims: ops.embed_mlp_params(flat, dims, W)
    padded = torch.cat([emb(st.params[:P], pd), emb(st.params[P:P + Q], qd), emb(st.params[P + Q:P + 2 * Q], qd), st.params[-1:]])
    pdp, qdp = ops.padded_dims(pd, W), ops.padded_dims(qd, W)
    up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=pdp, q_dims=qdp, batch_size=B, device=dev, lr_policy=1e-3, lr_q=1e-3,
                        lr_alpha=1e-3, wd_q=1e-3)
    up.load_state(padded.to(dev))
    D = 2 * X + U + 3
    for it in range(5):
        batch = torch.randn(B, D, generator=g)
        batch[:, X + U + 1] = 1.0
        batch[:, D - 1] = (torch.rand(B, generator=g) < 0.1).float()
        noise = [torch.randn(B, U, generator=g) for _ in range(3)]
        st, _, _ = osac.sgd_step(cfg, st, batch, *noise)
        up.sgd_step(batch.to(dev), None, None, *[n.to(dev) for n in noise])
    Pp, Qp = up.P, up.Q
    ext = lambda flat, dims: ops.extract_mlp_params(flat, dims, W)
    for name, ref in (("params", st.params), ("adam_m", st.adam_m), ("adam_v", st.adam_v)):
        t = getattr(up, name).cpu()
        logical = torch.cat([ext(t[:Pp], pd), ext(t[Pp:Pp + Qp], qd), ext(t[Pp + Qp:Pp + 2 * Qp], qd), t[-1:]])
        rel = float((logical - ref).norm() / ref.norm().clamp_min(1e-30))
        assert rel < (2e-3 if name == "adam_v" else 1e-3), (name, rel)
        # what is NOT a logical entry is exactly zero
        back = torch.cat([emb(logical[:P], pd), emb(logical[P:P + Q], qd), emb(logical[P + Q:P + 2 * Q], qd), logical[-1:]])
        assert torch.equal(back, t), name
    tq = up.target_q.cpu()
    assert torch.equal(torch.cat([emb(ext(tq[:Qp], qd), qd), emb(ext(tq[Qp:], qd), qd)]), tq)


def test_sac_trainer_accepts_reference_experiment_shapes(dev):
    """exp.py / exp_ppo.py-like shapes through the trainers: policy (32,)*4 with critic (48,)*3 trains on the fused kernels (zero-
    padded to 64); SAC with exp_ppo.py's 256x5 critic, or with more 128-wide layers than a tile's LDS holds, trains on the layered
    path (INTEGRATION.md "Network shapes"), and so does PPO with that critic; a POLICY wider than the rollout kernels take (256) is
    refused by name."""
    from mbpo import _hip
    from mbpo.optimizers.policy_optimizers.ppo.ppo import PPO
    from mbpo.optimizers.policy_optimizers.sac.sac import SAC
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import PendulumSystem
    from mbpo.systems.brax_wrapper import BraxWrapper
    from mbpo.types import Transition
    X, U = 3, 1
    system = PendulumSystem()
    dummy = Transition(observation=torch.zeros(X), action=torch.zeros(U), reward=torch.zeros(1), discount=torch.zeros(1),
                       next_observation=torch.zeros(X))
    tb = UniformSamplingQueue(16, dummy, 1, device=dev)
    tbs = tb.insert_rows(tb.init(0), torch.randn(16, 2 * X + U + 2, generator=torch.Generator().manual_seed(0)).to(dev))
    env = BraxWrapper(system, system.init_params(0), tbs, tb)
    tr = SAC(environment=env, num_timesteps=32 + 32 * 2 * 3, episode_length=10, num_env_steps_between_updates=2, num_envs=32, batch_size=32,
             grad_updates_per_step=2, min_replay_size=32, policy_hidden_layer_sizes=(32,) * 4, critic_hidden_layer_sizes=(48,) * 3)
    assert tr.kernel_width == 64 and tr.policy_dims == [3, 64, 64, 64, 64, 2] and tr.policy_dims_logical == [3, 32, 32, 32, 32, 2]
    params, metrics = tr.run_training(key=1)
    assert bool(torch.isfinite(params[1]).all()) and "eval/episode_reward" in metrics[-1]
    pp = PPO(environment=env, num_timesteps=2000, episode_length=10, num_envs=32, unroll_length=5, batch_size=16, num_minibatches=2,
             policy_hidden_layer_sizes=(32,) * 4, critic_hidden_layer_sizes=(48,) * 5)
    assert pp.kernel_width == 64
    params, metrics = pp.run_training(key=2)
    assert bool(torch.isfinite(params[1]).all())
    small = dict(environment=env, num_timesteps=32 + 32 * 2 * 3, episode_length=10, num_env_steps_between_updates=2, num_envs=32,
                 batch_size=32, grad_updates_per_step=2, min_replay_size=32)
    wide = SAC(**small, policy_hidden_layer_sizes=(32,) * 4, critic_hidden_layer_sizes=(256,) * 5)      # exp_ppo.py's critic
    assert wide.kernel_width == 64 and wide.q_width is None and wide.q_dims == [4, 256, 256, 256, 256, 256, 1]
    params, metrics = wide.run_training(key=3)
    assert bool(torch.isfinite(params[1]).all()) and bool(torch.isfinite(wide.updater.params).all())
    # four 128-wide hidden layers in the policy: the stored activations of a 16-row tile exceed the 160 KiB of LDS -> layered as well
    deep = SAC(**small, policy_hidden_layer_sizes=(32,) * 4, critic_hidden_layer_sizes=(128,) * 3)
    assert deep.kernel_width == 128
    params, metrics = deep.run_training(key=4)
    assert bool(torch.isfinite(deep.updater.params).all())
    wide_pol = SAC(**small, policy_hidden_layer_sizes=(200, 200), critic_hidden_layer_sizes=(300, 100))      # policy padded to 256 (rollout kernels)
    assert wide_pol.kernel_width == 256 and wide_pol.q_dims == [4, 300, 100, 1]
    params, metrics = wide_pol.run_training(key=5)
    assert bool(torch.isfinite(wide_pol.updater.params).all())
    wide_ppo = PPO(environment=env, num_timesteps=2000, episode_length=10, num_envs=32, unroll_length=5, batch_size=16, num_minibatches=2,
                   policy_hidden_layer_sizes=(32,) * 4, critic_hidden_layer_sizes=(256,) * 5)          # exp_ppo.py:36-38 verbatim
    assert wide_ppo.kernel_width == 64 and wide_ppo.value_width is None and wide_ppo.value_dims == [3, 256, 256, 256, 256, 256, 1]
    params, metrics = wide_ppo.run_training(key=6)
    assert bool(torch.isfinite(params[1]).all())
    with pytest.raises(_hip.MbpoHipError, match="exceeds"):       # a policy also runs inside the rollout kernels: one width up to 256
        PPO(environment=env, num_timesteps=2000, episode_length=10, num_envs=32, unroll_length=5, batch_size=16, num_minibatches=2,
            policy_hidden_layer_sizes=(512, 512))
    ok = SAC(environment=env, num_timesteps=1000, episode_length=10, policy_hidden_layer_sizes=(100, 100, 100), critic_hidden_layer_sizes=(128,) * 3)
    assert ok.kernel_width == 128


def test_sac_layered_gemm_tile_variants_agree(dev, tmp_path):
    """k_layered_gemm<MODE, T>: the 64 x 64 (T = 2) and 32 x 32 (T = 1) tile variants of the layered path's GEMM, forced in turn through
    MBPO_LAYERED_T2_MIN (read once per process: two child processes), give the same gradients within summation order."""
    import os
    import subprocess
    import sys
    code = '''
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "model-based-policy-optimizers_amd"); sys.path.insert(0, "tests")
import test_gpu_sac as T
dev = torch.device("cuda", 0)
cfg, st, batch, noise, nm, ns = T._make(5, 2, (200, 72), 300, 11, True, q_hidden=(136, 264, 40))
up = T._updater(dev, cfg, 300, two_launch=False)
up.load_state(st.params.to(dev), st.target_q.to(dev))
up.sgd_step(batch.to(dev), nm.to(dev), ns.to(dev), *[n.to(dev) for n in noise])
torch.cuda.synchronize()
torch.save({"g": up.grads.cpu(), "p": up.params.cpu(), "m": up.metrics.cpu()}, sys.argv[1])
'''
    outs = []
    for t2_min in ("1", "1000000000"):
        f = tmp_path / f"out_{t2_min}.pt"
        env = dict(os.environ, MBPO_LAYERED_T2_MIN=t2_min)
        r = subprocess.run([sys.executable, "-c", code, str(f)], env=env, capture_output=True, text=True, cwd=str(__import__("pathlib").Path(__file__).resolve().parents[1]))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f, weights_only=True))
    scale = float(outs[0]["g"].abs().max())
    torch.testing.assert_close(outs[0]["g"], outs[1]["g"], atol=1e-6 + 1e-6 * scale, rtol=1e-4)
    torch.testing.assert_close(outs[0]["p"], outs[1]["p"], atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(outs[0]["m"], outs[1]["m"], atol=1e-6, rtol=2e-5)
    cfg, st, batch, noise, nm, ns = _make(5, 2, (200, 72), 300, 11, True, q_hidden=(136, 264, 40))
    g_ref, _ = osac.grads(cfg, st.params.double(), st.target_q.double(), batch.double(), *[n.double() for n in noise], nm.double(), ns.double())
    torch.testing.assert_close(outs[0]["g"].double(), g_ref, atol=2e-6 + 2e-6 * float(g_ref.abs().max()), rtol=2e-4)


@pytest.mark.parametrize("pol_act,q_act,neq", [("relu", "tanh", False), ("tanh", "relu", True), ("swish", "swish", True)])
def test_sac_layered_path_activations_and_per_sample_discount(dev, pol_act, q_act, neq):
    """The layered path's epilogues for every activation (act and act' inside the GEMM epilogues) and its loss head with the
    per-sample discount of N1 (sac/losses.py:90-98), on shapes the fused kernel does not take; B = 1 included (one row, one tile)."""
    X, U = 5, 2
    kw = dict(non_equidistant_time=True, continuous_discounting=0.9, min_time_between_switches=0.05, max_time_between_switches=0.75,
              env_dt=0.05) if neq else {}
    for B in (70, 1):
        cfg, st, batch, noise, nm, ns = _make(X, U, (40, 72), B, 21, True, q_hidden=(136, 24), policy_act=pol_act, q_act=q_act,
                                               reward_scaling=1.5, **kw)
        g_ref, (cl, ac, al) = osac.grads(cfg, st.params.double(), st.target_q.double(), batch.double(), *[n.double() for n in noise],
                                         nm.double(), ns.double())
        up = _updater(dev, cfg, B, policy_activation=pol_act, q_activation=q_act, **kw)
        up.load_state(st.params.to(dev), st.target_q.to(dev))
        up.sgd_step(batch.to(dev), nm.to(dev), ns.to(dev), *[n.to(dev) for n in noise])
        up.finalize()
        torch.cuda.synchronize()
        g = up.grads.cpu().double()
        torch.testing.assert_close(g, g_ref, atol=2e-6 + 2e-6 * float(g_ref.abs().max()), rtol=3e-4)
        np.testing.assert_allclose(up.metrics.cpu().tolist()[:3], [cl, ac, al], rtol=1e-4, atol=2e-6)

"""GPU: the SAC forward/backward kernel specialised for the benchmark networks (csrc/sac_lean.hip, k_sac_lean<X>) against the
generic chain-table kernel (csrc/sac.hip, k_sac_fwd_bwd) — BIT FOR BIT.

The specialised kernel forms every dot product with the same MFMA sequence over the same k groups as the generic one (operands
swapped: a*b = b*a), the thin layers with the same FMA chains and the elementwise sections with the same expressions, so the
per-tile gradient slabs — and with them gradients, metrics, parameters, moments, target critics — must be identical, not close.
Oracle parity of the specialised kernel itself is tests/test_gpu_sac.py (it is the kernel those tests run on the 64x3 shapes).
"""
import ctypes as C

import pytest
import torch

from test_gpu_sac import _make, _updater

pytestmark = pytest.mark.gpu


def _set_lean(mode: int) -> None:
    from mbpo import _hip
    lib = _hip.load()
    lib.mbpo_debug_set_sac_lean.argtypes = [C.c_int]
    lib.mbpo_debug_set_sac_lean.restype = C.c_int
    assert lib.mbpo_debug_set_sac_lean(mode) == 0


@pytest.fixture(autouse=True)
def _restore_default():
    yield
    _set_lean(-1)


def _state(up):
    return {k: getattr(up, k).detach().clone() for k in ("grads", "params", "target_q", "adam_m", "adam_v", "metrics", "metrics_accum",
                                                          "step_count", "workspace")}


def _slab_floats(up):
    """workspace = [policy slabs | critic slabs | loss partials | ...]: everything the fwd/bwd launch writes."""
    n_tiles = (up.batch_size + 15) // 16
    return n_tiles * (up.P + 2 * up.Q + 4)


def _run(dev, lean, cfg, st, batch, noise, nm, ns, B, steps=1, given_noise=True, **kw):
    from mbpo import ops
    _set_lean(1 if lean else 0)
    up = _updater(dev, cfg, B, **kw)
    up.load_state(st.params.to(dev), st.target_q.to(dev))
    d = lambda t: None if t is None else t.to(dev)
    rng = ops.make_rng(dev, 11)
    for i in range(steps):
        bt = torch.roll(batch, i, 0).to(dev)
        if given_noise:
            up.sgd_step(bt, d(nm), d(ns), *[n.to(dev) for n in noise], defer_clip_check=steps > 1)
        else:
            up.sgd_step(bt, d(nm), d(ns), seed=3, offset=(7 + i) << 32, rng_dev=rng, defer_clip_check=steps > 1)
    up.finalize()
    torch.cuda.synchronize()
    return up, _state(up)


@pytest.mark.parametrize("X,B,normalize,given_noise", [
    (4, 256, True, True),        # the north-star shape (BASELINE configs[1])
    (4, 256, False, False),      # in-kernel Philox noise
    (3, 256, True, False),       # Pendulum observations (tests/test_sac.py of the reference at 64x3)
    (3, 40, True, True),         # ragged: the last tile holds 8 rows
    (4, 16, False, True),        # one tile
    (4, 2048, True, False),      # 128 tiles x 3 roles = 384 workgroups: more than the chip holds at once
    (2, 96, True, False),        # the other observation widths the kernel is instantiated for
    (5, 256, True, True),
    (6, 200, False, False),      # x + 2 = all eight chain waves in layer 0's weight gradient; ragged last tile
])
def test_lean_kernel_equals_generic_kernel_bit_for_bit(dev, X, B, normalize, given_noise):
    cfg, st, batch, noise, nm, ns = _make(X, 1, (64, 64, 64), B, 3, normalize, discounting=0.97, reward_scaling=1.5,
                                           lr_policy=3e-4, lr_q=3e-4, lr_alpha=3e-4, wd_q=1e-3)
    up_g, g = _run(dev, False, cfg, st, batch, noise, nm, ns, B, given_noise=given_noise)
    up_l, l = _run(dev, True, cfg, st, batch, noise, nm, ns, B, given_noise=given_noise)
    n = _slab_floats(up_g)
    assert torch.equal(g["workspace"][:n], l["workspace"][:n]), "per-tile slabs / loss partials differ"
    assert float(g["grads"].abs().sum()) > 0
    for k in ("grads", "metrics", "metrics_accum", "params", "target_q", "adam_m", "adam_v", "step_count"):
        assert torch.equal(g[k], l[k]), k


def test_lean_kernel_chain_of_steps_and_clip_fixup(dev):
    """Twelve chained two-launch steps with the clip check deferred to the next launch's prologue, at a max_grad_norm that clips
    some steps and not others: the specialised kernel's rare path (canonical norms, fix-up from the undo log, second pass) gives
    the generic kernel's parameters bit for bit, and counts the same clip events."""
    X, B = 4, 64
    cfg, st, batch, noise, nm, ns = _make(X, 1, (64, 64, 64), B, 4, True, discounting=0.95)
    # per-step gradient norms of the unclipped run place the thresholds: one that never clips, one that always does, and a few
    # around the median norm, at which some steps clip and others do not
    from mbpo import ops
    _set_lean(1)
    up = _updater(dev, cfg, B)
    up.load_state(st.params.to(dev), st.target_q.to(dev))
    rng = ops.make_rng(dev, 11)
    norms = []
    for i in range(12):
        up.sgd_step(torch.roll(batch, i, 0).to(dev), nm.to(dev), ns.to(dev), seed=3, offset=(7 + i) << 32, rng_dev=rng)
        g = up.grads
        norms.append(max(float(g[:up.P].norm()), float(g[up.P:up.P + 2 * up.Q].norm()), float(g[-1:].norm())))
    med = sorted(norms)[6]
    outs = {}
    for max_norm in (1e5, 1.25 * med, med, 0.8 * med, 1e-3):
        cfg.max_grad_norm = max_norm
        up_g, g = _run(dev, False, cfg, st, batch, noise, nm, ns, B, steps=12, given_noise=False)
        up_l, l = _run(dev, True, cfg, st, batch, noise, nm, ns, B, steps=12, given_noise=False)
        for k in ("grads", "metrics", "metrics_accum", "params", "target_q", "adam_m", "adam_v", "step_count"):
            assert torch.equal(g[k], l[k]), (max_norm, k)
        assert up_g.clip_events() == up_l.clip_events()
        outs[max_norm] = up_l.clip_events()
    assert outs[1e5] == 0 and outs[1e-3] == 12, outs
    assert any(0 < n < 12 for n in outs.values()), (outs, norms)


def test_lean_kernel_non_equidistant_discount(dev):
    """N1 (sac/losses.py:90-98): the per-sample discount inside the specialised kernel's target section."""
    X, B = 3, 48
    kw = dict(non_equidistant_time=True, continuous_discounting=0.7, min_time_between_switches=0.05, max_time_between_switches=0.4, env_dt=0.05)
    cfg, st, batch, noise, nm, ns = _make(X, 1, (64, 64, 64), B, 6, True, discounting=0.9, **kw)
    _, g = _run(dev, False, cfg, st, batch, noise, nm, ns, B, **kw)
    _, l = _run(dev, True, cfg, st, batch, noise, nm, ns, B, **kw)
    for k in ("grads", "metrics", "params", "target_q"):
        assert torch.equal(g[k], l[k]), k

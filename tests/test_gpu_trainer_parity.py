"""GPU: whole training steps of the HIP trainers against the CPU oracle loops on identical Philox streams — the path
bench.py times (SAC.training_step replayed from a hipGraph) and PPO.training_step.

  SAC.training_step  (sac/sac.py:283-327)   vs  oracle/trainer.py:CpuSacLoop
  PPO.training_step  (ppo/ppo.py:158-233)   vs  oracle/trainer.py:CpuPpoLoop

Tolerances (fp32): rollout rows of a 5-step unroll 2e-4 (rounding differences feed back through the dynamics);
replay positions, sampled indices and permutations bit-exact; running statistics 1e-5 relative; parameters after the
first training step by relative L2 (the first Adam steps move every weight by ~lr*sign(g): an element whose gradient is at
rounding level may flip, so element-wise comparison is meaningless) 5e-4, after 3 chained steps 5e-3.
The hipGraph-replayed epoch must equal the eagerly issued epoch BIT FOR BIT: all that changes between steps lives in device
words (include/mbpo_hip.h "randomness").
"""
import numpy as np
import pytest
import torch

from oracle import philox, ppo as oppo, sac as osac, systems as osys, trainer as otr

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _true_buffer(dev, X, U, rows, seed=0):
    from mbpo.replay import UniformSamplingQueue
    from mbpo.types import Transition
    dummy = Transition(observation=torch.zeros(X), action=torch.zeros(U), reward=torch.zeros(1), discount=torch.zeros(1),
                       next_observation=torch.zeros(X))
    tb = UniformSamplingQueue(rows, dummy, 1, device=dev)
    g = torch.Generator().manual_seed(seed)
    data = torch.randn(rows, 2 * X + U + 2, generator=g)
    if X == 3:       # Pendulum-shaped observations (cos, sin, omega)
        th = (torch.rand(rows, generator=g) * 2 - 1) * np.pi
        data[:, 0], data[:, 1], data[:, 2] = torch.cos(th), torch.sin(th), (torch.rand(rows, generator=g) * 2 - 1) * 8
    return tb, tb.insert_rows(tb.init(0), data.to(dev))


def _make_system(dev, kind):
    from mbpo.systems import EnsembleDynamics, EnsembleSystem, PendulumSystem, QuadraticReward
    if kind == "pendulum":
        system = PendulumSystem()
        return system, system.init_params(1), osys.PendulumSystem(), 3, 1
    X, U, E = 4, 1, 5
    dyn = EnsembleDynamics(X, U, n_members=E, device=dev)
    rew = QuadraticReward(X, U, target=[0.1, 0, 0, 0], q=[1, 2, 0.5, 0.1], r=[0.3])
    system = EnsembleSystem(dyn, rew)
    sp = system.init_params(1)
    sp.dynamics_params.params.mul_(0.5)
    rp = sp.reward_params
    osystem = osys.EnsembleSystem(sp.dynamics_params.params.cpu().clone(), dyn.dims, E, X, U,
                                  reward_fn=lambda a, b: osys.quadratic_reward(a, b, torch.tensor(rp.target), torch.tensor(rp.q),
                                                                               torch.tensor(rp.r)))
    return system, sp, osystem, X, U


SAC_KW = dict(num_envs=64, batch_size=256, grad_updates_per_step=4, num_env_steps_between_updates=5, episode_length=5,
              normalize_observations=True, max_replay_size=1500, min_replay_size=64, discounting=0.95, lr_policy=3e-4,
              lr_q=3e-4, lr_alpha=3e-4, wd_q=1e-4)


def _sac_setup(dev, kind, use_graph, n_steps=4):
    from mbpo.optimizers.policy_optimizers.sac.sac import SAC
    from mbpo.systems.brax_wrapper import BraxWrapper
    system, sp, osystem, X, U = _make_system(dev, kind)
    tb, tbs = _true_buffer(dev, X, U, 512)
    env = BraxWrapper(system, sp, tbs, tb)
    N, S = SAC_KW["num_envs"], SAC_KW["num_env_steps_between_updates"]
    tr = SAC(environment=env, num_timesteps=64 + N * S * n_steps, use_graph=use_graph, **SAC_KW)
    assert tr.num_training_steps_per_epoch == n_steps and tr.num_prefill_actor_steps == 1
    ts = tr.init_training_state(7)
    es = tr.reset_envs(env, 11, N)
    bs = tr.replay_buffer.init(13)
    return tr, ts, es, bs, osystem, X, U


def _sac_oracle(tr, es, osystem, X, U):
    cfg = osac.SacConfig(X, U, tr.policy_dims, tr.q_dims, discounting=SAC_KW["discounting"], lr_policy=3e-4, lr_q=3e-4,
                         lr_alpha=3e-4, wd_q=1e-4)
    return otr.CpuSacLoop(cfg, osystem, SAC_KW["num_envs"], SAC_KW["num_env_steps_between_updates"], SAC_KW["episode_length"],
                          SAC_KW["batch_size"], SAC_KW["grad_updates_per_step"], SAC_KW["max_replay_size"], True,
                          init_params=tr.updater.params.cpu().clone(), init_obs=es.obs.cpu().clone())


@pytest.mark.parametrize("kind", ["pendulum", "ensemble"])
def test_sac_training_step_matches_cpu_oracle(dev, kind):
    """Eager SAC.training_step, step by step, against CpuSacLoop (prefill included)."""
    from mbpo.utils import keys as K
    tr, ts, es, bs, osystem, X, U = _sac_setup(dev, kind, use_graph=False)
    loop = _sac_oracle(tr, es, osystem, X, U)
    # prefill (sac.py:329-345): one get_experience per prefill step under the first split of the prefill key
    ts, es, bs, _ = tr.prefill_replay_buffer(ts, es, bs, 17)
    loop.rekey(K.split(17)[0])
    loop.prefill_step()
    torch.testing.assert_close(tr._rollout_rows.cpu(), loop.last_rows, atol=2e-4, rtol=2e-4)
    assert bs.state.cpu().tolist()[:2] == [int(loop.qstate["insert_position"]), int(loop.qstate["sample_position"])]
    # three training steps of one epoch stream
    tr.rekey(19)
    loop.rekey(19)
    D = 2 * X + U + 3
    for step in range(3):
        ts, es, bs = tr.training_step(ts, es, bs)
        loop.training_step()
        torch.cuda.synchronize()
        tol = 2e-4 if step == 0 else 2e-3        # later steps act with parameters that already differ by rounding
        torch.testing.assert_close(tr._rollout_rows.cpu(), loop.last_rows, atol=tol, rtol=tol)
        # integer-valued bookkeeping columns are exact whatever the floats did: discount, truncation
        for col in (X + U + 1, D - 1):
            assert torch.equal(tr._rollout_rows[:, col].cpu(), loop.last_rows[:, col])
        st = bs.state.cpu().tolist()
        assert st[0] == int(loop.qstate["insert_position"]) == bs.insert_position
        assert st[1] == int(loop.qstate["sample_position"]) == bs.sample_position
        # the sampled minibatch rows: indices are bit-exact, so every gathered row is the oracle's row
        ref_batch = torch.from_numpy(loop.queue.gather(loop.qstate, loop.last_idx))
        torch.testing.assert_close(tr._batch_rows.cpu(), ref_batch, atol=tol, rtol=tol)
        sv = tr._stats_vec.cpu().numpy()
        assert sv[0] == loop.stats[0]
        np.testing.assert_allclose(sv[1:], loop.stats[1:], rtol=2e-5 if step == 0 else 2e-4, atol=2e-5)
        lim = 5e-4 if step == 0 else 5e-3
        P, Q2 = tr.updater.P, 2 * tr.updater.Q
        assert _rel(tr.updater.params[:P], loop.state.params[:P]) < lim
        assert _rel(tr.updater.params[P:P + Q2], loop.state.params[P:P + Q2]) < lim
        assert _rel(tr.updater.target_q, loop.state.target_q) < lim
        assert abs(float(tr.updater.params[-1]) - float(loop.state.params[-1])) < 1e-5 * (step + 1)
        assert float(tr.updater.step_count) == loop.state.count == 4 * (step + 1)
        torch.testing.assert_close(es.obs.cpu(), loop.env.obs, atol=tol, rtol=tol)
        assert torch.equal(es.info["steps"].cpu(), loop.env.steps) and torch.equal(es.done.cpu(), loop.env.done)


@pytest.mark.parametrize("kind", ["pendulum", "ensemble"])
def test_sac_graph_epoch_is_bit_identical_to_eager_and_matches_oracle(dev, kind):
    """training_epoch through the captured hipGraph (what bench.py times) == the same epoch issued eagerly, bit for bit;
    and both agree with CpuSacLoop after the 4 steps."""
    from mbpo.utils import keys as K
    out = []
    for use_graph in (False, True):
        tr, ts, es, bs, osystem, X, U = _sac_setup(dev, kind, use_graph=use_graph)
        loop = _sac_oracle(tr, es, osystem, X, U) if use_graph else None
        ts, es, bs, _ = tr.prefill_replay_buffer(ts, es, bs, 17)
        ts, es, bs, metrics = tr.training_epoch(ts, es, bs, 19)
        torch.cuda.synchronize()
        assert (tr._graph is not None) == use_graph
        st = bs.state.cpu().tolist()
        assert st[0] == bs.insert_position and st[1] == bs.sample_position and st[2] == bs.head
        out.append(dict(params=tr.updater.params.cpu().clone(), tq=tr.updater.target_q.cpu().clone(), m=tr.updater.adam_m.cpu().clone(),
                        v=tr.updater.adam_v.cpu().clone(), obs=es.obs.cpu().clone(), stats=tr._stats_vec.cpu().clone(),
                        rows=tr._rollout_rows.cpu().clone(), batch=tr._batch_rows.cpu().clone(), data=bs.data.cpu().clone(),
                        state=st, count=float(tr.updater.step_count), metrics=metrics, rng=tr._rng.cpu().tolist(),
                        env_steps=ts.env_steps))
    a, b = out
    for k in ("params", "tq", "m", "v", "obs", "stats", "rows", "batch", "data"):
        assert torch.equal(a[k], b[k]), f"graph replay differs from eager in {k}"
    assert a["state"] == b["state"] and a["count"] == b["count"] == 16 and a["rng"] == b["rng"] and a["env_steps"] == b["env_steps"]
    assert a["rng"][1] == 4
    assert a["metrics"] == b["metrics"]
    loop.rekey(K.split(17)[0])
    loop.prefill_step()
    loop.rekey(19)
    for _ in range(4):
        met = loop.training_step()
    P = tr.updater.P
    assert _rel(b["params"][:P], loop.state.params[:P]) < 5e-3 and _rel(b["params"][P:-1], loop.state.params[P:-1]) < 5e-3
    assert b["state"][0] == int(loop.qstate["insert_position"]) and b["state"][1] == int(loop.qstate["sample_position"])
    np.testing.assert_allclose(b["stats"].numpy(), loop.stats, rtol=5e-4, atol=5e-5)


def test_sac_short_epochs_replay_the_cached_graph(dev):
    """The reference's acceptance runs have 1 or 2 training steps per epoch (tests/test_sac.py:30-57).  The first step ever is issued
    eagerly and captured; every later step — the rest of that epoch AND the whole of every following epoch — is a replay.  Three
    2-step epochs through the graph equal the same six steps issued eagerly, bit for bit (parameters, moments, ring, RNG words)."""
    out = []
    for use_graph in (False, True):
        tr, ts, es, bs, osystem, X, U = _sac_setup(dev, "ensemble", use_graph=use_graph, n_steps=2)
        ts, es, bs, _ = tr.prefill_replay_buffer(ts, es, bs, 17)
        graphs = []
        for epoch, key in enumerate((19, 23, 29)):
            ts, es, bs, metrics = tr.training_epoch(ts, es, bs, key)
            graphs.append(tr._graph)
        torch.cuda.synchronize()
        if use_graph:
            assert graphs[0] is not None and graphs[1] is graphs[0] and graphs[2] is graphs[0]      # captured once, replayed since
        else:
            assert graphs == [None, None, None]
        st = bs.state.cpu().tolist()
        assert st[0] == bs.insert_position and st[1] == bs.sample_position and st[2] == bs.head
        out.append(dict(params=tr.updater.params.cpu().clone(), tq=tr.updater.target_q.cpu().clone(), m=tr.updater.adam_m.cpu().clone(),
                        v=tr.updater.adam_v.cpu().clone(), obs=es.obs.cpu().clone(), stats=tr._stats_vec.cpu().clone(),
                        data=bs.data.cpu().clone(), state=st, count=float(tr.updater.step_count), rng=tr._rng.cpu().tolist(),
                        env_steps=ts.env_steps, metrics=metrics))
    a, b = out
    for k in ("params", "tq", "m", "v", "obs", "stats", "data"):
        assert torch.equal(a[k], b[k]), f"graph replay differs from eager in {k}"
    assert a["state"] == b["state"] and a["count"] == b["count"] == 24 and a["rng"] == b["rng"] and a["env_steps"] == b["env_steps"]
    assert a["metrics"] == b["metrics"]


def test_sac_trainer_picks_the_step_flavour_from_the_clip_rate(dev, monkeypatch):
    """max_grad_norm small enough that every sgd_step clips: after the first epoch the trainer switches from the two-launch step
    (a clip costs it a fix-up and a second pass) to the three-launch step, re-captures its hipGraph, and its state stays
    bit-identical to a trainer pinned to the three-launch step; with the default max_grad_norm it keeps the two-launch step."""
    from mbpo.optimizers.policy_optimizers.sac.sac import SAC
    from mbpo.systems.brax_wrapper import BraxWrapper
    monkeypatch.delenv("MBPO_SAC_TWO_LAUNCH", raising=False)

    def run(max_norm, pinned):
        if pinned:
            monkeypatch.setenv("MBPO_SAC_TWO_LAUNCH", "0")
        else:
            monkeypatch.delenv("MBPO_SAC_TWO_LAUNCH", raising=False)
        system, sp, _, X, U = _make_system(dev, "pendulum")
        tb, tbs = _true_buffer(dev, X, U, 512)
        env = BraxWrapper(system, sp, tbs, tb)
        N, S = SAC_KW["num_envs"], SAC_KW["num_env_steps_between_updates"]
        tr = SAC(environment=env, num_timesteps=64 + N * S * 4, use_graph=True, max_grad_norm=max_norm, **SAC_KW)
        ts, es, bs = tr.init_training_state(7), tr.reset_envs(env, 11, N), tr.replay_buffer.init(13)
        ts, es, bs, _ = tr.prefill_replay_buffer(ts, es, bs, 17)
        flavours = [tr.updater.two_launch]
        for key in (19, 23, 29):
            ts, es, bs, _ = tr.training_epoch(ts, es, bs, key)
            flavours.append(tr.updater.two_launch)
        torch.cuda.synchronize()
        return tr, flavours

    a, fa = run(1e-3, pinned=False)
    assert fa == [True, False, False, False], fa                 # one epoch observed, then the three-launch step
    assert a.updater.clip_events() == 3 * 4 * SAC_KW["grad_updates_per_step"]
    b, fb = run(1e-3, pinned=True)
    assert fb == [False] * 4 and b.updater.two_launch_explicit
    for name in ("params", "target_q", "adam_m", "adam_v"):
        assert torch.equal(getattr(a.updater, name), getattr(b.updater, name)), name
    c, fc = run(1e5, pinned=False)
    assert fc == [True] * 4 and c.updater.clip_events() == 0


def test_sac_steps_draw_fresh_noise_past_2_pow_24(dev):
    """the random streams must not depend on a float counter.  With the device step index at 2^24 - 1, 2^24 and
    2^24 + 1 (where a float32 `x + 1` stops moving) and beyond 2^32, consecutive replays still draw different numbers."""
    from mbpo import ops
    tr, ts, es, bs, _, X, U = _sac_setup(dev, "pendulum", use_graph=False)
    ts, es, bs, _ = tr.prefill_replay_buffer(ts, es, bs, 17)
    tr.updater.step_count.fill_(float(2 ** 24))          # Adam's count saturated: bias corrections are 1 either way
    seen = []
    for counter in (2 ** 24 - 1, 2 ** 24, 2 ** 24 + 1, 2 ** 32 + 5, 2 ** 32 + 6):
        ops.set_rng(tr._rng, 99, counter)
        es.obs.copy_(es.info["first_obs"]); es.info["steps"].zero_(); es.done.zero_()
        tr.get_experience(ts.normalizer_params, ts.policy_params, es, bs)
        seen.append(tr._rollout_rows[:, X].cpu().clone())          # the sampled actions
    for i in range(len(seen)):
        for j in range(i + 1, len(seen)):
            assert not torch.equal(seen[i], seen[j])
    # and the counter itself advances exactly by one per training step, as an integer
    ops.set_rng(tr._rng, 99, 2 ** 24)
    ts, es, bs = tr.training_step(ts, es, bs)
    assert tr._rng.cpu().tolist() == [99, 2 ** 24 + 1]


PPO_KW = dict(num_envs=32, unroll_length=8, batch_size=16, num_minibatches=4, num_updates_per_batch=2, episode_length=20,
              normalize_observations=True, discounting=0.97, lr=3e-4, wd=1e-5, entropy_cost=1e-2, gae_lambda=0.95,
              clipping_epsilon=0.3, policy_hidden_layer_sizes=(64, 64), critic_hidden_layer_sizes=(64, 64))


@pytest.mark.parametrize("kind", ["pendulum", "ensemble"])
def test_ppo_training_step_matches_cpu_oracle(dev, kind):
    """PPO.training_step (2 unrolls of 32 envs -> 64 trajectories, 2 update epochs x 4 minibatches) against CpuPpoLoop."""
    from mbpo.optimizers.policy_optimizers.ppo.ppo import PPO
    from mbpo.systems.brax_wrapper import BraxWrapper
    system, sp, osystem, X, U = _make_system(dev, kind)
    tb, tbs = _true_buffer(dev, X, U, 256)
    env = BraxWrapper(system, sp, tbs, tb)
    kw = PPO_KW
    tr = PPO(environment=env, num_timesteps=3 * 16 * 8 * 4, **kw)
    ts = tr.init_training_state(5)
    es = env.reset([101 + i for i in range(kw["num_envs"])])
    cfg = oppo.PpoConfig(X, U, tr.policy_dims, tr.value_dims, entropy_cost=kw["entropy_cost"], discounting=kw["discounting"],
                         gae_lambda=kw["gae_lambda"], clipping_epsilon=kw["clipping_epsilon"], lr=kw["lr"], wd=kw["wd"])
    loop = otr.CpuPpoLoop(cfg, osystem, kw["num_envs"], kw["unroll_length"], kw["episode_length"], kw["batch_size"],
                          kw["num_minibatches"], kw["num_updates_per_batch"], True, init_params=tr.updater.params.cpu().clone(),
                          init_obs=es.obs.cpu().clone())
    tr.rekey(23)
    loop.rekey(23)
    for step in range(3):
        ts, es, _ = tr.training_step(ts, es)
        terms = loop.training_step()
        torch.cuda.synchronize()
        tol = 2e-4 if step == 0 else 3e-3
        torch.testing.assert_close(tr._data.cpu(), loop.last_data, atol=tol, rtol=tol)
        assert torch.equal(tr._perm.cpu(), torch.from_numpy(loop.last_perms[-1]))      # the last update epoch's shuffle
        sv = tr._stats_vec.cpu().numpy()
        assert sv[0] == loop.stats[0]
        np.testing.assert_allclose(sv[1:], loop.stats[1:], rtol=2e-5 if step == 0 else 5e-4, atol=2e-5)
        lim = 1e-3 if step == 0 else 1e-2
        P = tr.updater.P
        assert _rel(tr.updater.params[:P], loop.state.params[:P]) < lim
        assert _rel(tr.updater.params[P:], loop.state.params[P:]) < lim
        assert float(tr.updater.step_count) == loop.state.count == 8 * (step + 1)
        m = tr.updater.metrics.cpu().tolist()       # the last minibatch's four loss terms
        for got, key in zip(m, ("total_loss", "policy_loss", "v_loss", "entropy_loss")):
            assert abs(got - terms[key]) <= (2e-3 if step == 0 else 2e-2) * max(1.0, abs(terms[key])), key
    assert ts.env_steps == 3 * tr.env_step_per_training_step


@pytest.mark.parametrize("kind", ["pendulum", "ensemble"])
def test_ppo_training_epoch_graph_equals_eager(dev, kind):
    """PPO.training_epoch through the captured hipGraph (the reference compiles the epoch scan into one XLA computation,
    ppo/ppo.py:235-247) == the same epoch issued eagerly, bit for bit: parameters, moments, step count, env state, statistics, the
    last collected batch, the last permutation, the RNG words and the epoch's metrics."""
    from mbpo.optimizers.policy_optimizers.ppo.ppo import PPO
    from mbpo.systems.brax_wrapper import BraxWrapper
    out = {}
    for use_graph in (False, True):
        system, sp, osystem, X, U = _make_system(dev, kind)
        tb, tbs = _true_buffer(dev, X, U, 256)
        env = BraxWrapper(system, sp, tbs, tb)
        tr = PPO(environment=env, num_timesteps=5 * 16 * 8 * 4, use_graph=use_graph, **PPO_KW)
        assert tr.num_training_steps_per_epoch == 5
        ts = tr.init_training_state(5)
        es = env.reset([101 + i for i in range(PPO_KW["num_envs"])])
        metrics = None
        for key in (19, 31):                   # the second epoch re-uses the captured graph (same addresses)
            ts, es, metrics = tr.training_epoch(ts, es, key)
        torch.cuda.synchronize()
        assert (tr._graph is not None) == use_graph
        assert ts.env_steps == 10 * tr.env_step_per_training_step
        u = tr.updater
        out[use_graph] = dict(params=u.params.clone(), m=u.adam_m.clone(), v=u.adam_v.clone(), count=u.step_count.clone(),
                              obs=es.obs.clone(), steps=es.info["steps"].clone(), done=es.done.clone(), stats=tr._stats_vec.clone(),
                              data=tr._data.clone(), perm=tr._perm.clone(), rng=tr._rng.clone(), metrics=metrics)
        tr.close()
    for name, eager in out[False].items():
        if name == "metrics":
            assert eager == out[True][name]
        else:
            assert torch.equal(eager, out[True][name]), name
    assert float(out[True]["count"]) == 10 * 8


def test_philox_permutation_bit_exact(dev):
    from mbpo import ops
    # n <= 1024: one workgroup's LDS sort; 1024 < n <= 16384: one workgroup per key bucket (k_perm_bucket_sort); above: rank count
    for n, seed, off in ((1, 3, 0), (64, 5, 7), (1000, 2 ** 40 + 3, (1024 << 32) + 9), (1025, 4, 1), (4096, 9, 5 << 32), (5000, 13, 77),
                         (16384, 11, 1 << 33), (20000, 21, 3)):
        ref = philox.philox_permutation(seed, off, n)
        got = ops.philox_permutation(n, seed=seed, offset=off).cpu().numpy()
        assert np.array_equal(got, ref), n
        assert np.array_equal(np.sort(got), np.arange(n))
    # the bucket path's fallback (a bucket overflowing its LDS list raises workspace[0]; uniform keys never do): a raised flag makes the
    # one-workgroup sort behind it redo the permutation and clear the flag — the result is the same either way
    for stale in (1, 0):
        ws = torch.zeros(16384, dtype=torch.int32, device=dev)
        ws[0] = stale
        got = ops.philox_permutation(16384, seed=11, offset=1 << 33, workspace=ws).cpu().numpy()
        assert np.array_equal(got, philox.philox_permutation(11, 1 << 33, 16384))
        assert int(ws[0]) == 0
    # the device words are ADDED to the host (seed, offset)
    rng = ops.make_rng(dev, seed=100, counter=5)
    got = ops.philox_permutation(500, seed=7, offset=2 << 32, rng_dev=rng).cpu().numpy()
    assert np.array_equal(got, philox.philox_permutation(107, (2 << 32) + 5, 500))
    ops.rng_advance(rng, 3)
    assert rng.cpu().tolist() == [100, 8]

"""GPU parity of the iCEM device kernels (N4) against oracle/icem.py, the whole optimize() on the analytic Pendulum against the
oracle loop, and the reference's acceptance test (tests/test_icemopt.py: 200 MPC steps, horizon 20, default iCemParams,
summed reward >= -400).

Tolerances: coloured noise / candidates 2e-5 (fp32 sums of <= 65 harmonics against the fp64 irfft); values 1e-5; the
optimize() comparison uses loose tolerances because an elite set can flip on rounding-level ties."""
import math

import numpy as np
import pytest
import torch

from oracle import icem as oicem
from oracle import systems as osys

pytestmark = pytest.mark.gpu


def _lib():
    from mbpo import _hip
    return _hip, _hip.load()


@pytest.mark.parametrize("H,U,S,Kp,P,beta", [(20, 1, 500, 15, 10, 0.0), (21, 2, 64, 3, 2, 2.0), (8, 3, 17, 1, 1, 0.25), (64, 1, 33, 5, 3, 1.0), (128, 2, 40, 4, 2, 1.0), (127, 1, 9, 0, 1, 0.5)])
def test_icem_sample_parity(dev, H, U, S, Kp, P, beta):
    _hip, lib = _lib()
    g = torch.Generator().manual_seed(H + U)
    mean = (torch.randn(H, U, generator=g) * 0.3)
    std = torch.rand(H, U, generator=g) * 0.5 + 0.1
    prev = torch.rand(Kp, H, U, generator=g) * 2 - 1
    umin, umax = torch.full((U,), -1.0), torch.linspace(0.5, 1.0, U)
    NC, N = S + Kp, (S + Kp) * P
    actions = torch.zeros(H, N, U, device=dev)
    cand = torch.zeros(NC, H, U, device=dev)
    d = lambda t: t.to(dev).contiguous()
    dm, ds, dp, dlo, dhi = d(mean), d(std), d(prev), d(umin), d(umax)
    seed, off = 0x1234567, 3
    _hip.check(lib.mbpo_icem_sample(dm.data_ptr(), ds.data_ptr(), dp.data_ptr(), dlo.data_ptr(), dhi.data_ptr(), S, Kp, H, U, P, beta, seed, off,
                                    None, actions.data_ptr(), cand.data_ptr(), None), "mbpo_icem_sample")
    torch.cuda.synchronize()
    ref = oicem.sample_candidates(mean.double().numpy(), std.double().numpy(), prev.double().numpy(), umin.double().numpy(),
                                  umax.double().numpy(), S, H, U, beta, seed, off)
    np.testing.assert_allclose(cand.cpu().numpy(), ref, atol=2e-5, rtol=2e-5)
    # every particle of a candidate gets the candidate's sequence: actions[t, c*P + p] == cand[c, t]
    exp = cand.permute(1, 0, 2).repeat_interleave(P, dim=1)
    assert torch.equal(actions, exp)
    if beta == 0.0 and S >= 500:      # white noise: unit variance before clipping is what `sigma` normalises to
        raw = (ref[:S] - mean.double().numpy()[None]) / std.double().numpy()[None]
        inside = np.abs(ref[:S]) < 0.49
        assert abs(raw[inside].std() - 1.0) < 0.25


def test_icem_update_parity(dev):
    _hip, lib = _lib()
    rng = np.random.default_rng(0)
    H, U, NC, P, X, ne, nprev, alpha = 12, 2, 157, 3, 4, 20, 6, 0.3
    D = 2 * X + U + 3
    rows = rng.standard_normal((H * NC * P, D)).astype(np.float32)
    rows[:, X + U] = np.round(rows[:, X + U], 1)           # coarse rewards: ties between candidates do occur
    cand = rng.standard_normal((NC, H, U)).astype(np.float32)
    mean, std = rng.standard_normal((H, U)).astype(np.float32), (rng.random((H, U)) + 0.2).astype(np.float32)
    for use_max in (0, 1):
        rew = rows[:, X + U].reshape(H, NC, P)
        per_p = rew.mean(axis=0, dtype=np.float64)
        values = per_p.max(axis=1) if use_max else per_p.mean(axis=1)
        for bv0 in (-np.inf, 10.0):
            bs0 = rng.standard_normal((H, U)).astype(np.float32)
            m2, s2, bv, bs, pe = oicem.update(values.astype(np.float32).astype(np.float64), cand.astype(np.float64), mean.astype(np.float64),
                                              std.astype(np.float64), bv0, bs0.astype(np.float64), ne, nprev, alpha)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            dmean, dstd, dbv, dbs, dprev = t(mean.copy()), t(std.copy()), torch.tensor([bv0], device=dev, dtype=torch.float32), t(bs0.copy()), \
                torch.zeros(nprev, H, U, device=dev)
            dvals, drank = torch.zeros(NC, device=dev), torch.zeros(NC, device=dev, dtype=torch.int32)
            drows, dcand = t(rows), t(cand)
            _hip.check(lib.mbpo_icem_update(drows.data_ptr(), D, X + U, NC, P, H, U, dcand.data_ptr(), ne, nprev, alpha, use_max, dmean.data_ptr(),
                                            dstd.data_ptr(), dbv.data_ptr(), dbs.data_ptr(), dprev.data_ptr(), dvals.data_ptr(), drank.data_ptr(),
                                            None), "mbpo_icem_update")
            torch.cuda.synchronize()
            np.testing.assert_allclose(dvals.cpu().numpy(), values, atol=1e-5, rtol=1e-5)
            # ranks from the device values (ties broken by index, as np.argsort(kind='stable'))
            order = np.argsort(dvals.cpu().numpy(), kind="stable")
            assert np.array_equal(np.argsort(drank.cpu().numpy()), order)
            m2, s2, bv, bs, pe = oicem.update(dvals.cpu().numpy().astype(np.float64), cand.astype(np.float64), mean.astype(np.float64),
                                              std.astype(np.float64), bv0, bs0.astype(np.float64), ne, nprev, alpha)
            np.testing.assert_allclose(dmean.cpu().numpy(), m2, atol=1e-5, rtol=1e-5)
            np.testing.assert_allclose(dstd.cpu().numpy(), s2, atol=1e-5, rtol=1e-5)
            np.testing.assert_allclose(dprev.cpu().numpy(), pe, atol=0, rtol=0)
            np.testing.assert_allclose(dbs.cpu().numpy(), bs, atol=0, rtol=0)
            assert float(dbv) == np.float32(bv)


def test_icem_optimize_matches_oracle_loop_on_pendulum(dev):
    """iCemTO.optimize on the analytic Pendulum vs the numpy loop (same Philox candidates, oracle rollouts)."""
    from mbpo.optimizers import iCemParams, iCemTO
    from mbpo.systems import PendulumSystem
    from mbpo.utils import keys as K
    params = iCemParams(num_particles=2, num_samples=120, num_elites=12, num_steps=3, exponent=1.0, alpha=0.1, init_std=0.6)
    H = 10
    system = PendulumSystem()
    opt = iCemTO(horizon=H, action_dim=1, opt_params=params, key=5)
    opt.set_system(system)
    st = opt.init(7)
    st = st.replace(best_sequence=(torch.rand(H, 1, device=dev) - 0.5))
    x0 = torch.tensor([-0.8, 0.6, 0.5], device=dev)
    new = opt.optimize(x0, st)
    torch.cuda.synchronize()
    # oracle loop with the same keys
    osystem = osys.PendulumSystem()

    def step(x, u):
        xn, r = osystem.step(torch.from_numpy(x), torch.from_numpy(u))
        return xn.numpy(), r.numpy()

    mean = np.zeros((H, 1)); mean[:-1] = st.best_sequence.cpu().double().numpy()[1:]; mean[-1] = st.best_sequence.cpu().double().numpy()[-1]
    std = np.full((H, 1), params.init_std)
    best_v, best_s = -np.inf, mean.copy()
    nprev = max(int(params.elite_set_fraction * params.num_elites), 1)
    prev = np.zeros((nprev, H, 1))
    optimizer_key, _ = K.split(st.key, 2)
    carry = optimizer_key
    for it in range(params.num_steps):
        sampling_key, _pk = K.split(carry, 2)
        carry = K.split(sampling_key, 2)[0]
        cand = oicem.sample_candidates(mean, std, prev, -1.0, 1.0, params.num_samples, H, 1, params.exponent, sampling_key, it)
        vals = oicem.objective(step, x0.cpu().double().numpy(), cand, params.num_particles)
        mean, std, best_v, best_s, prev = oicem.update(vals, cand, mean, std, best_v, best_s, params.num_elites, nprev, params.alpha)
    assert abs(float(new.best_reward) - best_v) <= 2e-3 * max(1.0, abs(best_v))
    np.testing.assert_allclose(new.best_sequence.cpu().numpy(), best_s, atol=5e-3)


@pytest.mark.timeout(600)
def test_icem_mpc_solves_pendulum(dev):
    """tests/test_icemopt.py on the HIP path: MPC with the default iCemParams (500 samples, 10 particles, 5 steps), horizon 20."""
    from mbpo.optimizers import iCemParams, iCemTO
    from mbpo.systems import PendulumSystem
    system = PendulumSystem()
    state = system.reset()
    opt = iCemTO(horizon=20, action_dim=1, system=None, opt_params=iCemParams(), key=1)
    opt.set_system(system)
    ost = opt.init(2)
    x, total = state.x_next, 0.0
    for _ in range(200):
        u, ost = opt.act(obs=x, opt_state=ost)
        nxt = system.step(x=x, u=u, system_params=state.system_params)
        ost = ost.replace(system_params=nxt.system_params)
        x, total = nxt.x_next, total + float(nxt.reward)
    print("icem MPC return:", total)
    assert total >= -400


def test_icem_constraint_term_values(dev):
    """mbpo_icem_update_constrained (icem_optimizer.py:157-166): objective = summarize(mean_t reward) - lambda * relu(summarize_cost(cost_p)),
    summarize_cost = mean over particles, or max under pessimism; a NULL cost vector gives mbpo_icem_update's values."""
    rng = np.random.default_rng(4)
    X, U, H, NC, P = 3, 1, 6, 40, 3
    D = 2 * X + U + 3
    N = NC * P
    rows = rng.standard_normal((H * N, D)).astype(np.float32)
    cand = rng.standard_normal((NC, H, U)).astype(np.float32)
    cost = rng.standard_normal(N).astype(np.float32)
    from mbpo import _hip
    lib = _hip.load()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rew = rows[:, X + U].reshape(H, NC, P).mean(axis=0, dtype=np.float64)
    drows, dcand, dcost = t(rows), t(cand), t(cost)          # (kept alive: a temporary's block would be handed to the next allocation)
    for use_max in (0, 1):
        for cost_max in (0, 1):
            reward = rew.max(axis=1) if use_max else rew.mean(axis=1)
            c = cost.reshape(NC, P).astype(np.float64)
            c = c.max(axis=1) if cost_max else c.mean(axis=1)
            want = reward - 7.5 * np.maximum(c, 0.0)
            dvals, drank = torch.zeros(NC, device=dev), torch.zeros(NC, device=dev, dtype=torch.int32)
            mean, std = torch.zeros(H, U, device=dev), torch.ones(H, U, device=dev)
            bv, bs, prev = torch.full((1,), -np.inf, device=dev), torch.zeros(H, U, device=dev), torch.zeros(2, H, U, device=dev)
            _hip.check(lib.mbpo_icem_update_constrained(drows.data_ptr(), D, X + U, NC, P, H, U, dcand.data_ptr(), 5, 2, 0.1, use_max,
                                                        dcost.data_ptr(), 7.5, cost_max, mean.data_ptr(), std.data_ptr(), bv.data_ptr(),
                                                        bs.data_ptr(), prev.data_ptr(), dvals.data_ptr(), drank.data_ptr(), None),
                       "mbpo_icem_update_constrained")
            torch.cuda.synchronize()
            np.testing.assert_allclose(dvals.cpu().numpy(), want, atol=2e-5, rtol=2e-5)
            assert float(bv) == float(dvals.max())


def test_icem_cost_fn_steers_the_plan(dev):
    """iCemTO(cost_fn=...) (icem_optimizer.py:99,161-166): a torch cost over one trajectory, vmapped over candidates x particles.
    With the constraint "never push with u < -0.2" the best sequence obeys it; without it the same keys produce a plan that does
    not — and the constrained optimum matches the numpy loop that applies the same penalty to the oracle's objective."""
    from mbpo.optimizers import iCemParams, iCemTO
    from mbpo.systems import PendulumSystem
    from mbpo.utils import keys as K
    params = iCemParams(num_particles=2, num_samples=150, num_elites=15, num_steps=4, exponent=1.0, alpha=0.1, init_std=0.6, lambda_constraint=50.0)
    H = 10
    system = PendulumSystem()
    x0 = torch.tensor([-0.8, 0.6, 0.5], device=dev)

    def cost_fn(observation, action):          # one trajectory: [H, 3], [H, 1] -> scalar; positive where the constraint is violated
        return torch.clamp(-0.2 - action, min=0.0).sum()

    def run(cf):
        opt = iCemTO(horizon=H, action_dim=1, opt_params=params, key=5, cost_fn=cf)
        opt.set_system(system)
        st = opt.init(7)
        return opt.optimize(x0, st), st

    free, _ = run(None)
    con, st = run(cost_fn)
    assert float(free.best_sequence.min()) < -0.3                # the unconstrained plan does push hard (u = -1 from this state)
    assert float(con.best_sequence.min()) >= -0.2 - 1e-6         # the constrained one does not
    # numpy loop with the same keys and the same penalty
    osystem = osys.PendulumSystem()

    def step(x, u):
        xn, r = osystem.step(torch.from_numpy(x), torch.from_numpy(u))
        return xn.numpy(), r.numpy()

    mean = np.zeros((H, 1)); std = np.full((H, 1), params.init_std)
    best_v, best_s = -np.inf, mean.copy()
    nprev = max(int(params.elite_set_fraction * params.num_elites), 1)
    prev = np.zeros((nprev, H, 1))
    carry = K.split(st.key, 2)[0]
    for it in range(params.num_steps):
        sampling_key, _pk = K.split(carry, 2)
        carry = K.split(sampling_key, 2)[0]
        cand = oicem.sample_candidates(mean, std, prev, -1.0, 1.0, params.num_samples, H, 1, params.exponent, sampling_key, it)
        vals = oicem.objective(step, x0.cpu().double().numpy(), cand, params.num_particles)
        vals = vals - params.lambda_constraint * np.maximum(np.clip(-0.2 - cand[:, :, 0], 0.0, None).sum(axis=1), 0.0)
        mean, std, best_v, best_s, prev = oicem.update(vals, cand, mean, std, best_v, best_s, params.num_elites, nprev, params.alpha)
    assert abs(float(con.best_reward) - best_v) <= 2e-3 * max(1.0, abs(best_v))
    np.testing.assert_allclose(con.best_sequence.cpu().numpy(), best_s, atol=5e-3)


@pytest.mark.parametrize("N,S,L,AR,env_major", [(5500, 20, 2 ** 30, 1, False), (333, 12, 5, 1, True), (100, 6, 4, 2, False)])
def test_open_loop_pendulum_thread_kernel_equals_tile_kernel(dev, N, S, L, AR, env_major):
    """The planner's candidate evaluation on the analytic Pendulum (rollout_actions) runs one THREAD per env (k_openloop_pendulum) instead
    of a 768-thread workgroup per 16 envs: same device functions in the same order — the tile kernel's rows, obs, steps, done bit for bit
    (incl. auto-reset at a short episode length and action_repeat 2)."""
    import ctypes as C
    from mbpo import ops, _hip
    from mbpo.systems import PendulumSystem
    lib = _hip.load()
    lib.mbpo_debug_set_rollout_lean.argtypes = [C.c_int]
    system = PendulumSystem()
    spec = system.rollout_spec(system.reset(device=dev).system_params, dev)
    g = torch.Generator().manual_seed(3)
    th = (torch.rand(N, generator=g) * 2 - 1) * math.pi
    obs0 = torch.stack([torch.cos(th), torch.sin(th), (torch.rand(N, generator=g) * 2 - 1) * 8], dim=1)
    first = torch.tensor([-1.0, 0.0, 0.0]).expand(N, 3).contiguous()
    actions = (torch.rand(S, N, 1, generator=g) * 2.4 - 1.2).to(dev)           # some outside [-1, 1]: the torque clip
    steps0 = torch.randint(0, max(1, min(L, 5)), (N,), generator=g).float()
    done0 = (torch.rand(N, generator=g) < 0.2).float()
    res = {}
    try:
        for mode in (0, -1):
            lib.mbpo_debug_set_rollout_lean(mode)
            obs, steps, done = obs0.to(dev), steps0.to(dev), done0.to(dev)
            rows = ops.model_rollout(x_dim=3, u_dim=1, actions=actions, obs=obs, first_obs=first.to(dev), steps=steps, done=done, n_steps=S,
                                     episode_length=L, action_repeat=AR, env_major=env_major, seed=5, offset=0, **spec)
            torch.cuda.synchronize()
            res[mode] = (rows.clone(), obs.clone(), steps.clone(), done.clone())
    finally:
        lib.mbpo_debug_set_rollout_lean(-1)
    for a, b in zip(res[0], res[-1]):
        assert torch.equal(a, b)
    assert float(res[-1][0].abs().sum()) > 0

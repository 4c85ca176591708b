"""CPU suite: the oracle against the golden vectors, oracle self-consistency, Philox known answers."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import nets as onets
from oracle import philox
from oracle import replay as orep
from oracle import rollout as oro
from oracle import scans as oscan
from oracle import systems as osys

GOLD = Path(__file__).parent / "golden"


# ------------------------------------------------------------------------------------------------ Pendulum KATs (pinned)
def test_pendulum_kat_oracle():
    kat = json.loads((GOLD / "pendulum_kat.json").read_text())
    p = osys.PendulumParams()
    for dt_, tol in ((torch.float64, 1e-12), (torch.float32, 2e-6)):
        for c in kat["cases"]:
            x = torch.tensor([c["x"]], dtype=dt_)
            u = torch.tensor([c["u"]], dtype=dt_)
            xn = osys.pendulum_next_state(x, u, p)[0].double().numpy()
            r = float(osys.pendulum_reward(x, u, p)[0])
            np.testing.assert_allclose(xn, c["x_next"], atol=max(tol, 2e-7 if dt_ == torch.float32 else 2e-15), rtol=tol,
                                       err_msg=c["why"])
            assert abs(r - c["reward"]) <= tol * max(1.0, abs(c["reward"])) * 4, c["why"]


def test_pendulum_shapes_like_reference_test():
    """tests/test_sys_pendulum.py:12-20 — 20 envs, uniform actions -> x_next (20,3), reward (20,)."""
    x = torch.tensor([[-1.0, 0.0, 0.0]] * 20)
    u = torch.rand(20, 1, generator=torch.Generator().manual_seed(0))
    xn, r = osys.PendulumSystem().step(x, u)
    assert xn.shape == (20, 3) and r.shape == (20,)


# ------------------------------------------------------------------------------------------------ scans KATs (pinned)
def test_scan_kat_oracle():
    kat = json.loads((GOLD / "scan_kat.json").read_text())
    for c in kat["gae"]:
        col = lambda k: np.asarray(c[k], np.float64).reshape(-1, 1)
        vs, adv = oscan.compute_gae(col("truncation"), col("termination"), col("rewards"), col("values"),
                                    np.asarray([c["bootstrap"]]), c["discounting"], c["gae_lambda"])
        np.testing.assert_allclose(vs.reshape(-1), c["vs"], rtol=1e-12, atol=1e-12, err_msg=c["why"])
        np.testing.assert_allclose(adv.reshape(-1), c["advantages"], rtol=1e-12, atol=1e-12, err_msg=c["why"])
    for c in kat["lambda_return"]:
        out = oscan.lambda_return(np.asarray(c["reward"]).reshape(-1, 1), np.asarray(c["next_values"]).reshape(-1, 1),
                                  c["discount"], c["lambda"])
        np.testing.assert_allclose(out.reshape(-1), c["returns"], rtol=1e-12, atol=1e-12, err_msg=c["why"])


def test_gae_lambda1_no_masks_is_discounted_return():
    """Property: lambda=1, no masks -> vs_t = sum_k g^k r_{t+k} + g^{T-t} boot."""
    rng = np.random.default_rng(0)
    T, B, g = 6, 4, 0.9
    r, v, boot = rng.standard_normal((T, B)), rng.standard_normal((T, B)), rng.standard_normal(B)
    z = np.zeros((T, B))
    vs, _ = oscan.compute_gae(z, z, r, v, boot, g, 1.0)
    ret = boot.copy()
    for t in range(T - 1, -1, -1):
        ret = r[t] + g * ret
        np.testing.assert_allclose(vs[t], ret, rtol=1e-12)


# ------------------------------------------------------------------------------------------------ Philox
def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors):
    ctr=0,key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8;  ctr=ff..,key=ff.. -> 408f276d 41c83b0e a20bc7c6 6d5451fd;
    ctr=243f6a88 85a308d3 13198a2e 03707344, key=a4093822 299f31d0 -> d16cfe09 94fdcceb 5001e420 24126ea1."""
    r = philox.philox4x32_10(np.array([0], np.uint32), 0, 0, 0, 0, 0)
    assert [int(v[0]) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xFFFFFFFF
    r = philox.philox4x32_10(np.array([f], np.uint32), f, f, f, f, f)
    assert [int(v[0]) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox.philox4x32_10(np.array([0x243f6a88], np.uint32), 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(v[0]) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_normal_moments_and_randint_range():
    n = philox.philox_normal(7, 3, philox.STREAM_POLICY_NOISE, np.arange(200000, dtype=np.uint64))
    assert abs(float(n.mean())) < 0.01 and abs(float(n.std()) - 1.0) < 0.01
    k = philox.philox_randint(7, 3, philox.STREAM_REPLAY, np.arange(100000, dtype=np.uint64), 5, 17)
    assert k.min() == 5 and k.max() == 16
    counts = np.bincount(k - 5, minlength=12)
    assert counts.min() > 100000 / 12 * 0.9


# ------------------------------------------------------------------------------------------------ replay buffer
def test_queue_semantics():
    q = orep.UniformSamplingQueue(10, 2, 1)
    st = q.init()
    rows = lambda a, b: np.stack([np.arange(a, b), np.arange(a, b)], 1).astype(np.float32)
    st = q.insert(st, rows(0, 4))
    assert (int(st["insert_position"]), int(st["sample_position"]), q.size(st)) == (4, 0, 4)
    st = q.insert(st, rows(4, 10))
    assert int(st["insert_position"]) == 10 and q.size(st) == 10
    # overflow by 3: roll=-3, oldest 3 rows dropped, positions stay at the end
    st = q.insert(st, rows(10, 13))
    assert int(st["insert_position"]) == 10 and int(st["sample_position"]) == 0
    assert np.array_equal(st["data"][:, 0], np.arange(3, 13))
    # wrap-mode gather
    assert np.array_equal(q.gather(st, np.array([-1, 10, 3]))[:, 0], [12, 3, 6])


def test_queue_sample_position_after_roll_when_not_full():
    q = orep.UniformSamplingQueue(8, 1, 1)
    st = q.init()
    st = q.insert(st, np.zeros((6, 1), np.float32))
    st = q.insert(st, np.ones((5, 1), np.float32))          # 6+5 > 8 -> roll = -3
    assert int(st["insert_position"]) == 8 and int(st["sample_position"]) == 0
    assert np.array_equal(st["data"][:, 0], [0, 0, 0, 1, 1, 1, 1, 1])


def test_running_stats_matches_batch_moments():
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((100, 3)) * 2 + 1, rng.standard_normal((50, 3)) - 2
    s = orep.stats_update(orep.stats_update(orep.stats_init(3).astype(np.float64), a, dtype=np.float64), b, dtype=np.float64)
    allx = np.concatenate([a, b])
    assert s[0] == 150
    np.testing.assert_allclose(s[1:4], allx.mean(0), rtol=1e-6)
    np.testing.assert_allclose(s[7:10], allx.std(0), rtol=1e-6)   # population std (summed_variance / count)


# ------------------------------------------------------------------------------------------------ NormalTanh / nets
def test_normal_tanh_log_prob_matches_change_of_variables():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(50, 4, generator=g, dtype=torch.float64)
    eps = torch.randn(50, 2, generator=g, dtype=torch.float64)
    z = onets.sample_no_postprocessing(logits, eps)
    loc, scale = onets.split_logits(logits)
    normal_lp = torch.distributions.Normal(loc, scale).log_prob(z)
    a = torch.tanh(z)
    ref = (normal_lp - torch.log(1 - a ** 2)).sum(-1)          # textbook tanh-squash correction
    torch.testing.assert_close(onets.log_prob(logits, z), ref, rtol=1e-7, atol=1e-7)
    assert torch.all(scale >= onets.MIN_STD)


def test_mlp_forward_matches_torch_linear():
    g = torch.Generator().manual_seed(1)
    dims = [5, 16, 16, 3]
    p = onets.init_mlp_flat(dims, g, dtype=torch.float64) + 0.1
    layers = onets.unflatten(p, dims)
    x = torch.randn(7, 5, generator=g, dtype=torch.float64)
    h = x
    for i, (w, b) in enumerate(layers):
        h = torch.nn.functional.linear(h, w.T, b)
        if i < 2:
            h = torch.nn.functional.silu(h)
    torch.testing.assert_close(onets.mlp_forward(p, dims, x), h)
    # lecun_uniform bound
    w0 = onets.unflatten(onets.init_mlp_flat(dims, g), dims)[0][0]
    assert float(w0.abs().max()) <= math.sqrt(3 / 5) + 1e-6


# ------------------------------------------------------------------------------------------------ rollout bookkeeping
def test_rollout_episode_autoreset_bookkeeping():
    """Episode/AutoReset semantics (brax_utils/training.py:98-137): done & truncation at steps>=L, next_obs is the
    FIRST obs after a reset, steps restart from 0 on the step after done."""
    g = torch.Generator().manual_seed(0)
    pd = [3, 64, 64, 2]
    pp = onets.init_mlp_flat(pd, g)
    N, S, L = 4, 7, 3
    obs = torch.tensor([[1.0, 0.0, 0.0]] * N)
    first = torch.tensor([[0.0, 1.0, 0.5]] * N)
    st = oro.EnvState(obs, first, torch.zeros(N), torch.zeros(N))
    st2, rows = oro.rollout(osys.PendulumSystem(), pp, pd, st, S, L, policy_noise=torch.randn(S, N, 1, generator=g))
    rows = rows.reshape(S, N, 10)
    disc, trunc = rows[:, 0, 5], rows[:, 0, 9]
    assert disc.tolist() == [1, 1, 0, 1, 1, 0, 1]
    assert trunc.tolist() == [0, 0, 1, 0, 0, 1, 0]
    assert torch.equal(rows[2, :, 6:9], first) and torch.equal(rows[3, :, 0:3], first)
    assert torch.equal(st2.steps, torch.full((N,), 1.0))
    # observation chain: obs[s+1] == next_obs[s]
    assert torch.equal(rows[1:, :, 0:3], rows[:-1, :, 6:9])


def test_ensemble_nll_oracle_matches_finite_differences():
    """oracle/ensemble.py (N3, build-defined loss): autograd gradient vs central differences in fp64, and the closed form of the
    loss at a known point (zero weights: mean = x, sigma = softplus(0) + min_std)."""
    import math
    import torch
    from oracle import ensemble as oens
    from oracle import nets as onets
    g = torch.Generator().manual_seed(0)
    X, U, E, B = 3, 1, 2, 8
    dims = [X + U, 64, 2 * X]
    P = onets.n_params(dims)
    params = torch.cat([onets.init_mlp_flat(dims, g, torch.float64) for _ in range(E)])
    rows = torch.randn(40, 2 * X + U + 2, generator=g, dtype=torch.float64)
    idx = torch.randint(0, 40, (E, B), generator=g)
    grads, losses = oens.nll_grads(params, dims, E, rows, idx, X, U, True, 1e-3)
    for j in (0, 5, P - 1, P + 7, 2 * P - 3):
        e = j // P
        hp = params.clone(); hp[j] += 1e-6
        hm = params.clone(); hm[j] -= 1e-6
        fd = (oens.nll_grads(hp, dims, E, rows, idx, X, U)[1][e] - oens.nll_grads(hm, dims, E, rows, idx, X, U)[1][e]) / 2e-6
        assert abs(float(fd) - float(grads[j])) <= 1e-6 + 1e-5 * abs(float(grads[j]))
    zero = torch.zeros(E * P, dtype=torch.float64)
    _, l0 = oens.nll_grads(zero, dims, E, rows, idx, X, U, True, 1e-3)
    sig = math.log(2.0) + 1e-3
    for e in range(E):
        b = rows[idx[e]]
        want = (0.5 * ((b[:, X + U + 2:] - b[:, :X]) / sig) ** 2 + math.log(sig)).sum(1).mean()
        assert abs(float(l0[e]) - float(want)) < 1e-12


def test_icem_oracle_noise_and_update_properties():
    """oracle/icem.py: powerlaw_psd_gaussian is normalised to unit variance for every exponent and series length parity, its
    spectrum falls as f^-beta; the elite update follows np.argsort (best last, ties by index) and the soft update formula."""
    import numpy as np
    from oracle import icem as oicem
    rng = np.random.default_rng(0)
    for H in (20, 21):
        K = H // 2 + 1
        for beta in (0.0, 1.0, 2.0):
            y = oicem.powerlaw_psd_gaussian(beta, H, rng.standard_normal((20000, K)), rng.standard_normal((20000, K)))
            assert y.shape == (20000, H)
            # `sigma` (general_utils.py:177-180) normalises the non-DC power to one: unit variance about each series' own mean
            nd = y - y.mean(axis=-1, keepdims=True)
            assert abs(np.sqrt((nd ** 2).mean()) - 1.0) < 0.03, (H, beta)
            spec = (np.abs(np.fft.rfft(y, axis=-1)) ** 2).mean(axis=0)
            if beta > 0:
                assert spec[1] > spec[K // 2] > spec[K - 2]          # coloured: power decreases with frequency
    # deterministic candidates from the Philox stream: same (seed, offset) -> same samples, different offset -> different
    a = oicem.sample_candidates(np.zeros((8, 2)), np.ones((8, 2)), np.zeros((1, 8, 2)), -1.0, 1.0, 16, 8, 2, 1.0, 11, 0)
    b = oicem.sample_candidates(np.zeros((8, 2)), np.ones((8, 2)), np.zeros((1, 8, 2)), -1.0, 1.0, 16, 8, 2, 1.0, 11, 0)
    c = oicem.sample_candidates(np.zeros((8, 2)), np.ones((8, 2)), np.zeros((1, 8, 2)), -1.0, 1.0, 16, 8, 2, 1.0, 11, 1)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and a.shape == (17, 8, 2) and np.abs(a).max() <= 1.0
    vals = np.array([0.0, 3.0, 1.0, 3.0, -2.0])
    cand = np.arange(5 * 2 * 1, dtype=np.float64).reshape(5, 2, 1)
    m, s, bv, bs, pe = oicem.update(vals, cand, np.zeros((2, 1)), np.ones((2, 1)), -np.inf, np.zeros((2, 1)), 3, 2, 0.5)
    assert bv == 3.0 and np.array_equal(bs, cand[3])                 # stable argsort: of the tied bests the later index is last
    assert np.array_equal(pe, cand[[1, 3]])
    assert np.allclose(m, 0.5 * cand[[2, 1, 3]].mean(0)) and np.allclose(s, np.sqrt(0.5 + 0.5 * cand[[2, 1, 3]].var(0)))


def test_oracle_reproduces_sac_step_golden():
    """tests/golden/sac_step_small.npz (tests/golden/make_sac_step_golden.py): the fp64 oracle must reproduce the committed
    gradients, losses and updated state — any edit to oracle/sac.py / oracle/nets.py that changes the arithmetic shows up here."""
    import importlib.util
    from pathlib import Path
    import numpy as np
    gdir = Path(__file__).resolve().parent / "golden"
    spec = importlib.util.spec_from_file_location("make_sac_step_golden", gdir / "make_sac_step_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold = np.load(gdir / "sac_step_small.npz")
    cfg, st, batch, noise, nm, ns = mod.build()
    np.testing.assert_array_equal(st.params.numpy(), gold["params"])          # the generator's inputs are what the file holds
    out = mod.evaluate(cfg, st, batch, noise, nm, ns)
    for k, v in out.items():
        np.testing.assert_allclose(v, gold[k], rtol=1e-12, atol=1e-13, err_msg=k)


def test_oracle_reproduces_ppo_and_bptt_goldens():
    """tests/golden/ppo_step_small.npz and bptt_actor_small.npz (tests/golden/make_ppo_bptt_golden.py): the fp64 oracles must
    reproduce the committed gradients / losses / scans — an edit to oracle/ppo.py, oracle/bptt.py, oracle/scans.py or
    oracle/nets.py that changes the arithmetic shows up here."""
    import importlib.util
    from pathlib import Path
    import numpy as np
    gdir = Path(__file__).resolve().parent / "golden"
    spec = importlib.util.spec_from_file_location("make_ppo_bptt_golden", gdir / "make_ppo_bptt_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold = np.load(gdir / "ppo_step_small.npz")
    cfg, params, data, noise, nm, ns = mod.build_ppo()
    np.testing.assert_array_equal(params.numpy(), gold["params"])
    for k, v in mod.eval_ppo(cfg, params, data, noise, nm, ns).items():
        np.testing.assert_allclose(v, gold[k], rtol=1e-12, atol=1e-13, err_msg=f"ppo {k}")
    gold = np.load(gdir / "bptt_actor_small.npz")
    args = mod.build_bptt()
    np.testing.assert_array_equal(args[1].numpy(), gold["actor_params"])
    for k, v in mod.eval_bptt(*args).items():
        np.testing.assert_allclose(v, gold[k], rtol=1e-12, atol=1e-13, err_msg=f"bptt {k}")

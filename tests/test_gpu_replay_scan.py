"""GPU parity through the C-ABI: replay buffer (bit-exact), running statistics, GAE / lambda-return scans."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import replay as orep
from oracle import scans as oscan

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def _logical_view(data, state):
    """logical row i = physical row (i + head) % max."""
    mx = data.shape[0]
    head = int(state[2])
    idx = (torch.arange(mx) + head) % mx
    return data.cpu()[idx].numpy()


@pytest.mark.parametrize("mx,D,batches", [
    (10, 9, [1, 1, 3, 5, 4, 10, 2]),        # the reference's true buffer: max_replay_size=10 (base_optimizer.py:54)
    (64, 12, [20, 20, 20, 20, 7, 64, 1]),   # overflow -> roll path, ragged sizes, full-size insert
    (1000, 10, [333, 333, 333, 333, 333]),
    (7, 3, [0, 7, 0, 3]),                   # empty inserts
])
def test_replay_insert_gather_bit_exact(dev, mx, D, batches):
    from mbpo import ops
    q = orep.UniformSamplingQueue(mx, D, 8)
    st = q.init()
    data = torch.zeros(mx, D, device=dev)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(0)
    for n in batches:
        rows = rng.standard_normal((n, D)).astype(np.float32)
        st = q.insert(st, rows)
        ops.replay_insert(data, state, torch.from_numpy(rows).to(dev))
        s = state.cpu()
        assert int(s[0]) == int(st["insert_position"])
        assert int(s[1]) == int(st["sample_position"])
        assert np.array_equal(_logical_view(data, s), st["data"])   # bit-exact logical contents
        # gather incl. out-of-range / negative indices (mode='wrap')
        idx = rng.integers(-3 * mx, 3 * mx, size=37).astype(np.int32)
        got = ops.replay_gather(data, state, torch.from_numpy(idx).to(dev)).cpu().numpy()
        assert np.array_equal(got, q.gather(st, idx))


def test_replay_insert_too_large(dev):
    from mbpo import ops, _hip
    data = torch.zeros(8, 4, device=dev)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(_hip.MbpoHipError):
        ops.replay_insert(data, state, torch.zeros(9, 4, device=dev))


def test_replay_sample_bit_exact(dev):
    from mbpo import ops
    mx, D = 2 ** 14, 12
    q = orep.UniformSamplingQueue(mx, D, 64 * 640)
    st = q.init()
    data = torch.zeros(mx, D, device=dev)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(1)
    for k in range(30):   # 30*640 > 2**14 -> several rolls
        rows = rng.standard_normal((640, D)).astype(np.float32)
        st = q.insert(st, rows)
        ops.replay_insert(data, state, torch.from_numpy(rows).to(dev))
        if k % 7 == 3:
            seed, offset = 99 + k, 5 * k
            idx_ref, batch_ref = q.sample(st, seed, offset)
            out, idx = ops.replay_sample(data, state, 64 * 640, seed, offset, return_idx=True)
            assert np.array_equal(idx.cpu().numpy(), idx_ref)          # sampling indices bit-exact
            assert np.array_equal(out.cpu().numpy(), batch_ref)        # gathered rows bit-exact
            assert idx_ref.min() >= int(st["sample_position"]) and idx_ref.max() < int(st["insert_position"])


@pytest.mark.parametrize("X,D,off,n", [(3, 10, 0, 100), (4, 12, 0, 20480), (17, 43, 0, 777), (3, 10, 6, 5), (4, 12, 0, 1)])
def test_running_stats(dev, X, D, off, n):
    from mbpo import ops
    rng = np.random.default_rng(2)
    stats = orep.stats_init(X)
    stats_d = torch.from_numpy(stats.copy()).to(dev)
    for it in range(3):
        rows = (rng.standard_normal((n, D)) * 3 + 1.5).astype(np.float32)
        ref64 = orep.stats_update(stats.astype(np.float64), rows[:, off:off + X].astype(np.float64), dtype=np.float64)
        ops.running_stats_update(torch.from_numpy(rows).to(dev), off, X, stats_d)
        got = stats_d.cpu().numpy()
        assert got[0] == ref64[0]
        np.testing.assert_allclose(got, ref64, rtol=2e-5, atol=2e-5)
        stats = got.copy()


@pytest.mark.parametrize("X,D,off,n", [(4, 12, 0, 20480), (3, 10, 2, 1), (17, 40, 5, 100000), (128, 128, 0, 700)])
def test_running_stats_fused_update_is_bit_identical_to_the_split_path(dev, X, D, off, n):
    """mbpo_running_stats_update (single rank: 3 launches) vs reduce(pass 0) -> reduce(pass 1) -> apply (what a process group
    runs, with its all-reduces in between): the same bits in stats and sums, over several chained updates."""
    from mbpo import ops
    g = torch.Generator().manual_seed(5)
    a = torch.from_numpy(orep.stats_init(X)).to(dev)
    b = a.clone()
    sums_a, sums_b = torch.zeros(1 + 2 * X, device=dev), torch.zeros(1 + 2 * X, device=dev)
    for it in range(3):
        rows = (torch.randn(n, D, generator=g) * 2.5 + 0.7).to(dev)
        ops.running_stats_update(rows, off, X, a, sums=sums_a)                                   # fused
        ops.running_stats_update(rows, off, X, b, all_reduce=lambda t: None, sums=sums_b)        # split (identity all-reduce)
        assert torch.equal(a, b), it
        assert torch.equal(sums_a, sums_b), it


def test_running_stats_first_update_from_init(dev):
    """init_state then one update == plain batch mean/std (count=n), incl. the std clip at 1e-6 for constant columns."""
    from mbpo import ops
    X = 3
    rows = torch.tensor([[1.0, 2.0, 5.0]] * 4)
    rows[:, 1] = torch.tensor([0.0, 1.0, 2.0, 3.0])
    stats_d = torch.from_numpy(orep.stats_init(X)).to(dev)
    ops.running_stats_update(rows.to(dev), 0, X, stats_d)
    got = stats_d.cpu().numpy()
    assert got[0] == 4
    np.testing.assert_allclose(got[1:4], [1.0, 1.5, 5.0], rtol=1e-6)
    np.testing.assert_allclose(got[7:10], [1e-6, np.sqrt(1.25), 1e-6], rtol=1e-6)


# ------------------------------------------------------------------------------------------------ scans
def _gae_inputs(T, B, seed, p_trunc=0.15, p_term=0.1):
    rng = np.random.default_rng(seed)
    trunc = (rng.random((T, B)) < p_trunc).astype(np.float32)
    discount = (rng.random((T, B)) > p_term).astype(np.float32)
    term = (1 - discount) * (1 - trunc)            # ppo/losses.py:89
    rew = rng.standard_normal((T, B)).astype(np.float32)
    val = rng.standard_normal((T, B)).astype(np.float32)
    boot = rng.standard_normal(B).astype(np.float32)
    return trunc, term.astype(np.float32), rew, val, boot


@pytest.mark.parametrize("T,B", [(1, 5), (3, 7), (5, 1000), (8, 64), (10, 33), (40, 513), (64, 10), (65, 9), (200, 21)])
@pytest.mark.parametrize("time_major", [False, True])
def test_gae_parity(dev, T, B, time_major):
    from mbpo import ops
    trunc, term, rew, val, boot = _gae_inputs(T, B, T * 1000 + B)
    vs_ref, adv_ref = oscan.compute_gae(trunc, term, rew, val, boot, 0.99, 0.95)
    lay = (lambda a: a) if time_major else (lambda a: np.ascontiguousarray(a.T))
    to = lambda a: torch.from_numpy(lay(a)).to(dev)
    vs, adv = ops.gae_scan(to(trunc), to(term), to(rew), to(val), torch.from_numpy(boot).to(dev), 0.99, 0.95, time_major)
    vs, adv = vs.cpu().numpy(), adv.cpu().numpy()
    if not time_major:
        vs, adv = vs.T, adv.T
    np.testing.assert_allclose(vs, vs_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv, adv_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("T,B", [(1, 3), (5, 100), (20, 50), (32, 4096), (100, 3)])
@pytest.mark.parametrize("time_major", [False, True])
def test_lambda_return_parity(dev, T, B, time_major):
    from mbpo import ops
    rng = np.random.default_rng(T + B)
    rew = rng.standard_normal((T, B)).astype(np.float32)
    nv = rng.standard_normal((T, B)).astype(np.float32)
    ref = oscan.lambda_return(rew, nv, 0.99, 0.95)
    lay = (lambda a: a) if time_major else (lambda a: np.ascontiguousarray(a.T))
    out = ops.lambda_return_scan(torch.from_numpy(lay(rew)).to(dev), torch.from_numpy(lay(nv)).to(dev), 0.99, 0.95,
                                 time_major).cpu().numpy()
    if not time_major:
        out = out.T
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)


def test_scan_golden_kat(dev):
    """Hand-derived known answers (tests/golden/scan_kat.json) through the HIP path."""
    from mbpo import ops
    kat = json.loads((GOLD / "scan_kat.json").read_text())
    for case in kat["gae"]:
        a = lambda k: torch.tensor(case[k], dtype=torch.float32).reshape(-1, 1).to(dev)   # [T,1] time-major
        for tm in (True, False):
            f = (lambda t: t) if tm else (lambda t: t.T.contiguous())
            vs, adv = ops.gae_scan(f(a("truncation")), f(a("termination")), f(a("rewards")), f(a("values")),
                                   torch.tensor([case["bootstrap"]], dtype=torch.float32).to(dev),
                                   case["discounting"], case["gae_lambda"], tm)
            np.testing.assert_allclose(vs.cpu().numpy().reshape(-1), case["vs"], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(adv.cpu().numpy().reshape(-1), case["advantages"], rtol=1e-6, atol=1e-6)
    for case in kat["lambda_return"]:
        r = torch.tensor(case["reward"], dtype=torch.float32).reshape(-1, 1).to(dev)
        nv = torch.tensor(case["next_values"], dtype=torch.float32).reshape(-1, 1).to(dev)
        out = ops.lambda_return_scan(r, nv, case["discount"], case["lambda"], True)
        np.testing.assert_allclose(out.cpu().numpy().reshape(-1), case["returns"], rtol=1e-6, atol=1e-6)


def test_gae_full_size_properties(dev):
    """BASELINE config 3 size (16384 envs x T=40): size-independent properties instead of the slow oracle loop:
    (i) lambda=0 -> vs = r + g(1-term)*V_{t+1} masked (1-step TD target); (ii) batch-major == time-major."""
    from mbpo import ops
    T, B = 40, 16384
    trunc, term, rew, val, boot = _gae_inputs(T, B, 7)
    to = lambda a: torch.from_numpy(a).to(dev)
    vs_tm, adv_tm = ops.gae_scan(to(trunc), to(term), to(rew), to(val), to(boot), 0.99, 0.95, True)
    bm = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)
    vs_bm, adv_bm = ops.gae_scan(bm(trunc), bm(term), bm(rew), bm(val), to(boot), 0.99, 0.95, False)
    torch.testing.assert_close(vs_bm.T, vs_tm, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(adv_bm.T, adv_tm, rtol=1e-5, atol=1e-5)
    vs0, _ = ops.gae_scan(bm(trunc), bm(term), bm(rew), bm(val), to(boot), 0.99, 0.0, False)
    v_next = np.concatenate([val[1:], boot[None]], 0)
    td = (rew + 0.99 * (1 - term) * v_next - val) * (1 - trunc) + val
    np.testing.assert_allclose(vs0.cpu().numpy().T, td, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("T,B,time_major", [(5, 37, False), (40, 130, False), (5, 300, True), (100, 9, False)])
def test_gae_per_element_discount_parity(dev, T, B, time_major):
    """N1: compute_gae with a per-step discount array (ppo/losses_new.py:181-226) vs the numpy restatement (1e-5)."""
    from mbpo import ops
    from oracle import scans
    rng = np.random.default_rng(T * 1000 + B)
    trunc = (rng.random((T, B)) < 0.1).astype(np.float32)
    term = ((rng.random((T, B)) < 0.05) * (1 - trunc)).astype(np.float32)
    rew, val = rng.standard_normal((T, B)).astype(np.float32), rng.standard_normal((T, B)).astype(np.float32)
    boot = rng.standard_normal(B).astype(np.float32)
    disc = np.exp(-0.9 * 0.05 * rng.integers(1, 16, (T, B))).astype(np.float32)
    vs_ref, adv_ref = scans.compute_gae(trunc, term, rew, val, boot, disc.astype(np.float64), 0.95)
    lay = (lambda a: torch.from_numpy(a).to(dev).contiguous()) if time_major else (lambda a: torch.from_numpy(a.T.copy()).to(dev))
    vs, adv = ops.gae_scan(lay(trunc), lay(term), lay(rew), lay(val), torch.from_numpy(boot).to(dev), lay(disc), 0.95, time_major)
    back = (lambda t: t.cpu().numpy()) if time_major else (lambda t: t.cpu().numpy().T)
    np.testing.assert_allclose(back(vs), vs_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(back(adv), adv_ref, rtol=1e-5, atol=1e-5)

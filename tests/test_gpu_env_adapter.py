"""GPU: the env adapter either side of the fused rollout — BraxWrapper.reset (R4: systems/brax_wrapper.py:25-38) and the
Evaluator / EvalWrapper (N2: sac/acting.py:82-145, brax_utils/training.py:156-199) — against their oracle restatements."""
import numpy as np
import pytest
import torch

from oracle import nets as onets, philox, rollout as oro, systems as osys

pytestmark = pytest.mark.gpu


def test_host_key_split_restatement():
    from mbpo.utils import keys as K
    for key in (0, 1, 12345, 2 ** 63 + 17, 2 ** 64 - 1):
        for num in (2, 3, 7):
            assert K.split(key, num) == philox.split(key, num)


@pytest.mark.parametrize("N,rows_in,cap", [(4096, 2 ** 16, 2 ** 16), (4096, 1, 10), (33, 300, 256), (64, 0, 10)])
def test_brax_wrapper_reset_parity(dev, N, rows_in, cap):
    """N independent draws from the true buffer (bit-exact indices vs oracle/philox.py), first_obs, the key split, the reset
    reward; full buffer of 2^16 rows at N = 4096 (BASELINE configs[1]), the reference tests' 1-row buffer, a wrapped ring
    (300 rows into 256), and the EMPTY buffer (row 0 of the zero dummy data)."""
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import EnsembleDynamics, EnsembleSystem, QuadraticReward
    from mbpo.systems.brax_wrapper import BraxWrapper
    from mbpo.types import Transition
    from mbpo.utils import keys as K
    X, U = 4, 1
    dummy = Transition(observation=torch.zeros(X), action=torch.zeros(U), reward=torch.zeros(1), discount=torch.zeros(1),
                       next_observation=torch.zeros(X))
    tb = UniformSamplingQueue(cap, dummy, 1, device=dev)
    tbs = tb.init(0)
    g = torch.Generator().manual_seed(3)
    data = torch.randn(rows_in, 2 * X + U + 2, generator=g)
    for a in range(0, rows_in, 200):                       # several inserts: exercises the ring head
        tbs = tb.insert_rows(tbs, data[a:a + 200].to(dev))
    system = EnsembleSystem(EnsembleDynamics(X, U, n_members=2, device=dev), QuadraticReward(X, U))
    env = BraxWrapper(system, system.init_params(1), tbs, tb)
    keys = K.split(77, N)
    st = env.reset(keys)
    logical = tb.logical_data(tbs).cpu()
    idx, est, rew, k1 = oro.brax_wrapper_reset(logical, tbs.insert_position, tbs.sample_position, keys, X, U)
    assert torch.equal(st.obs.cpu(), est.obs) and torch.equal(st.info["first_obs"].cpu(), est.obs)       # gathered rows: exact
    assert torch.equal(st.reward.cpu(), rew)
    assert st.system_params.key == k1
    assert float(st.done.abs().sum()) == 0 and float(st.info["steps"].abs().sum()) == 0 and float(st.info["truncation"].abs().sum()) == 0
    assert st.info["first_obs"].data_ptr() != st.obs.data_ptr()                                           # a copy, not an alias
    if rows_in == 0:
        assert float(st.obs.abs().sum()) == 0
    if rows_in == 2 ** 16:
        # the draws really cover the buffer: 4096 draws from 65536 rows -> ~3970 distinct rows expected
        assert len(np.unique(idx)) > 3800 and idx.min() >= 0 and idx.max() < 2 ** 16
    if rows_in == 300:
        assert tbs.head != 0 and idx.max() < 256


def _policy(seed, X, U):
    g = torch.Generator().manual_seed(seed)
    pd = [X, 64, 64, 2 * U]
    return pd, onets.init_mlp_flat(pd, g) + 0.05 * torch.randn(onets.n_params(pd), generator=g)


class _Trainer:
    """What Evaluator needs from a trainer."""

    def __init__(self, dev, pd, deterministic, nm, ns):
        from mbpo import ops
        self.device, self.policy_spec, self.x_dim, self.u_dim = dev, ops.MlpSpec(pd), pd[0], pd[-1] // 2
        self.deterministic_eval = deterministic
        self._nm, self._ns = nm, ns

    def _norm(self, _):
        return self._nm, self._ns


@pytest.mark.parametrize("L,AR,kind", [(10, 3, "fused"), (12, 3, "fused"), (40, 1, "fused"), (40, 1, "user_done"), (30, 2, "user_done")])
def test_evaluator_matches_eval_wrapper_oracle(dev, L, AR, kind):
    """eval/episode_reward sums rewards until the FIRST done; eval/avg_episode_length is info['steps'] at that point (or at the
    end of the unroll) — from the rows, not a constant.  'user_done': a System that terminates episodes itself."""
    from mbpo.optimizers.policy_optimizers.sac.sac import Evaluator
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import PendulumSystem
    from mbpo.systems.brax_wrapper import BraxWrapper
    from mbpo.types import Transition
    from mbpo.utils import keys as K
    import test_gpu_generic_system as tg
    X, U, N = 3, 1, 50
    system = PendulumSystem() if kind == "fused" else tg._user_pendulum()(fall_done=True)
    osystem = osys.PendulumSystem() if kind == "fused" else tg._OracleFall()
    dummy = Transition(observation=torch.zeros(X), action=torch.zeros(U), reward=torch.zeros(1), discount=torch.zeros(1),
                       next_observation=torch.zeros(X))
    tb = UniformSamplingQueue(128, dummy, 1, device=dev)
    g = torch.Generator().manual_seed(2)
    data = torch.randn(128, 2 * X + U + 2, generator=g)
    th = (torch.rand(128, generator=g) * 2 - 1) * np.pi
    data[:, 0], data[:, 1], data[:, 2] = torch.cos(th), torch.sin(th), torch.randn(128, generator=g)
    tbs = tb.insert_rows(tb.init(0), data.to(dev))
    env = BraxWrapper(system, system.init_params(0), tbs, tb)
    pd, ppar = _policy(9, X, U)
    nm, ns = torch.randn(X, generator=g) * 0.1, torch.rand(X, generator=g) + 0.5
    tr = _Trainer(dev, pd, True, nm.to(dev), ns.to(dev))
    ev = Evaluator(tr, env, num_eval_envs=N, episode_length=L, action_repeat=AR, key=5)
    m = ev.run_evaluation((None, ppar.to(dev)), {"training/x": 1.0}, unroll_key=31)
    keys = K.split(31, N)
    _, first, _, _ = oro.brax_wrapper_reset(tb.logical_data(tbs).cpu(), tbs.insert_position, tbs.sample_position, keys, X, U)
    er, es = oro.evaluate(osystem, ppar, pd, first, L, AR, "swish", nm, ns, deterministic=True)
    torch.testing.assert_close(ev.last_episode_rewards.cpu(), er, atol=2e-3, rtol=2e-4)
    assert torch.equal(ev.last_episode_steps.cpu(), es)
    assert abs(m["eval/episode_reward"] - float(er.mean())) <= 2e-4 * abs(float(er.mean())) + 1e-3
    assert m["eval/avg_episode_length"] == float(es.mean()) and m["training/x"] == 1.0
    for k in ("eval/walltime", "eval/epoch_eval_time", "eval/sps"):
        assert k in m
    if kind == "user_done":
        assert float(es.min()) < float(es.max())          # episodes of different lengths: the mask matters
    elif L % AR:
        assert m["eval/avg_episode_length"] == float((L // AR) * AR)     # never reaches episode_length: no done inside the unroll

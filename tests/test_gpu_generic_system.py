"""GPU: a USER-DEFINED System (the reference's plug-in seam: base_systems.py:40-52, base_dynamics.py:15-20) trained against
through the non-fused path — mbpo_policy_act (HIP) -> the user's batched torch System.step -> mbpo_episode_step (HIP).

The user systems below are written in plain torch (they reuse the oracle's Pendulum formulas on device tensors), so the
same physics exists in both forms: the fused rollout kernel (PendulumSystem) and the non-fused path must produce the same
transition rows from the same Philox stream (atol 2e-5: the two evaluate atan2/sin/cos/tanh with different libm paths), and
both must agree with the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import nets as onets, philox, rollout as oro, systems as osys

pytestmark = pytest.mark.gpu


def _user_pendulum():
    from mbpo.systems import Dynamics, Reward, System, SystemState
    from mbpo.systems.dynamics.base_dynamics import Normal

    class MyDynamics(Dynamics):
        def __init__(self):
            super().__init__(x_dim=3, u_dim=1)

        def init_params(self, key):
            return osys.PendulumParams()

        def next_state(self, x, u, dynamics_params):
            xn = osys.pendulum_next_state(x, u, dynamics_params)
            return Normal(xn, torch.zeros_like(xn)), dynamics_params

    class MyReward(Reward):
        def __init__(self):
            super().__init__(x_dim=3, u_dim=1)

        def init_params(self, key):
            return osys.PendulumParams()

        def __call__(self, x, u, reward_params, x_next=None):
            r = osys.pendulum_reward(x, u, reward_params)
            return Normal(r, torch.zeros_like(r)), reward_params

    class MySystem(System):
        """pendulum_system.py:18-39, batched: x' = next_state(...).mean(); r = reward(x, u, ., x').mean()."""

        def __init__(self, fall_done: bool = False):
            super().__init__(dynamics=MyDynamics(), reward=MyReward())
            self.fall_done = fall_done
            self.calls = 0

        def step(self, x, u, system_params):
            self.calls += 1
            d, dp = self.dynamics.next_state(x, u, system_params.dynamics_params)
            x_next = d.mean()
            r, rp = self.reward(x, u, system_params.reward_params, x_next)
            done = (x_next[:, 2].abs() > 3.0).float() if self.fall_done else 0.0      # a termination condition of the user's own
            return SystemState(x_next=x_next, reward=r.mean(), system_params=system_params.replace(dynamics_params=dp, reward_params=rp),
                               done=done)

    return MySystem


class _OracleFall:
    """oracle counterpart of MySystem(fall_done=True)"""
    x_dim, u_dim = 3, 1

    def __init__(self):
        self.s = osys.PendulumSystem()

    def step(self, x, u, **_):
        xn, r = self.s.step(x, u)
        return xn, r, (xn[:, 2].abs() > 3.0).float()


def _inputs(N, S, seed=0):
    g = torch.Generator().manual_seed(seed)
    th = (torch.rand(N, generator=g) * 2 - 1) * np.pi
    obs = torch.stack([torch.cos(th), torch.sin(th), (torch.rand(N, generator=g) * 2 - 1) * 4], 1)
    pd = [3, 64, 64, 2]
    ppar = onets.init_mlp_flat(pd, g) + 0.05 * torch.randn(onets.n_params(pd), generator=g)
    nm, ns = torch.randn(3, generator=g) * 0.1, torch.rand(3, generator=g) + 0.5
    return obs, pd, ppar, nm, ns


@pytest.mark.parametrize("ppo,env_major,ar,fall", [(False, False, 1, False), (True, True, 1, False), (False, False, 2, False),
                                                   (False, False, 1, True), (True, True, 2, True)])
def test_generic_rollout_equals_fused_and_oracle(dev, ppo, env_major, ar, fall):
    from mbpo import ops
    from mbpo.systems import PendulumSystem
    N, S, L = 100, 7, 5
    obs0, pd, ppar, nm, ns = _inputs(N, S)
    seed, offset = 4242, (3 << 32) + 11
    user = _user_pendulum()(fall_done=fall)
    sp = user.init_params(0)
    spec = ops.MlpSpec(pd)

    def run(system, sysp):
        o = obs0.to(dev).clone()
        st, dn = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        rows = ops.model_rollout(policy_params=ppar.to(dev), policy_spec=spec, x_dim=3, u_dim=1, obs=o, first_obs=obs0.to(dev),
                                 steps=st, done=dn, n_steps=S, episode_length=L, action_repeat=ar, norm_mean=nm.to(dev),
                                 norm_std=ns.to(dev), ppo_extras=ppo, env_major=env_major, seed=seed, offset=offset,
                                 **system.rollout_spec(sysp, dev))
        return rows.cpu(), o.cpu(), st.cpu(), dn.cpu()

    rows_g, o_g, st_g, dn_g = run(user, sp)
    assert user.calls == S * ar and not user.fused
    noise = torch.from_numpy(philox.philox_normal(seed, offset, philox.STREAM_POLICY_NOISE, np.arange(S * N, dtype=np.uint64))).reshape(S, N, 1)
    est, rows_ref = oro.rollout(_OracleFall() if fall else osys.PendulumSystem(), ppar, pd,
                                oro.EnvState(obs0, obs0.clone(), torch.zeros(N), torch.zeros(N)), S, L, ar, "swish", nm, ns,
                                policy_noise=noise, ppo_extras=ppo, env_major=env_major)
    torch.testing.assert_close(rows_g, rows_ref, atol=2e-4, rtol=2e-4)
    D = rows_g.shape[1]
    assert torch.equal(rows_g[:, 3 + 1 + 1], rows_ref[:, 3 + 1 + 1]) and torch.equal(rows_g[:, D - 1], rows_ref[:, D - 1])   # discount, truncation
    assert torch.equal(st_g, est.steps) and torch.equal(dn_g, est.done)
    torch.testing.assert_close(o_g, est.obs, atol=2e-4, rtol=2e-4)
    if fall:
        assert float((1 - rows_g[:, 5]).sum()) > float(rows_g[:, D - 1].sum()) > 0      # terminations AND truncations happened
    else:
        fused = PendulumSystem()
        rows_f, o_f, st_f, dn_f = run(fused, fused.init_params(0))
        torch.testing.assert_close(rows_g, rows_f, atol=2e-5, rtol=2e-5)
        assert torch.equal(st_g, st_f) and torch.equal(dn_g, dn_f)


def test_sac_and_ppo_train_against_a_user_defined_system(dev):
    """SAC.training_epoch / PPO.training_step on the user's System == the same on the fused PendulumSystem (same streams),
    to rounding: parameters by relative L2 after 3 steps."""
    from mbpo.optimizers.policy_optimizers.ppo.ppo import PPO
    from mbpo.optimizers.policy_optimizers.sac.sac import SAC
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import PendulumSystem
    from mbpo.systems.brax_wrapper import BraxWrapper
    from mbpo.types import Transition
    X, U = 3, 1
    dummy = Transition(observation=torch.zeros(X), action=torch.zeros(U), reward=torch.zeros(1), discount=torch.zeros(1),
                       next_observation=torch.zeros(X))
    g = torch.Generator().manual_seed(1)
    data = torch.randn(64, 2 * X + U + 2, generator=g)
    th = (torch.rand(64, generator=g) * 2 - 1) * np.pi
    data[:, 0], data[:, 1] = torch.cos(th), torch.sin(th)
    res = {}
    for name in ("user", "fused"):
        system = _user_pendulum()() if name == "user" else PendulumSystem()
        tb = UniformSamplingQueue(64, dummy, 1, device=dev)
        tbs = tb.insert_rows(tb.init(0), data.to(dev))
        env = BraxWrapper(system, system.init_params(0), tbs, tb)
        tr = SAC(environment=env, num_timesteps=32 + 32 * 4 * 3, episode_length=6, num_env_steps_between_updates=4, num_envs=32,
                 batch_size=64, grad_updates_per_step=3, normalize_observations=True, max_replay_size=512, min_replay_size=32)
        ts = tr.init_training_state(7)
        es = tr.reset_envs(env, 11, 32)
        bs = tr.replay_buffer.init(13)
        ts, es, bs, _ = tr.prefill_replay_buffer(ts, es, bs, 17)
        ts, es, bs, met = tr.training_epoch(ts, es, bs, 19)
        torch.cuda.synchronize()
        assert (tr._graph is not None) == (name == "fused")          # user code is never captured into a hipGraph
        pp = PPO(environment=env, num_timesteps=1000, episode_length=6, num_envs=32, unroll_length=4, batch_size=16,
                 num_minibatches=2, num_updates_per_batch=2, normalize_observations=True, policy_hidden_layer_sizes=(64, 64),
                 critic_hidden_layer_sizes=(64, 64))
        pts = pp.init_training_state(3)
        pes = env.reset(list(range(50, 82)))
        pp.rekey(29)
        for _ in range(2):
            pts, pes, _ = pp.training_step(pts, pes)
        torch.cuda.synchronize()
        res[name] = (tr.updater.params.cpu().clone(), tr._stats_vec.cpu().clone(), bs.state.cpu().tolist(), pp.updater.params.cpu().clone())
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(res["user"][0], res["fused"][0]) < 2e-3 and rel(res["user"][3], res["fused"][3]) < 5e-3
    assert res["user"][2] == res["fused"][2]
    torch.testing.assert_close(res["user"][1], res["fused"][1], atol=1e-4, rtol=1e-3)


def test_policy_act_matches_oracle_and_bad_shapes(dev):
    from mbpo import ops
    N = 33
    obs0, pd, ppar, nm, ns = _inputs(N, 1, seed=5)
    pd2 = [3, 64, 64, 6]                    # u_dim = 3
    g = torch.Generator().manual_seed(2)
    ppar2 = onets.init_mlp_flat(pd2, g)
    noise = torch.randn(N, 3, generator=g)
    act, raw, lp = ops.policy_act(ppar2.to(dev), ops.MlpSpec(pd2), obs0.to(dev), nm.to(dev), ns.to(dev), noise=noise.to(dev),
                                  want_extras=True)
    logits = onets.mlp_forward(ppar2, pd2, onets.normalize(obs0, nm, ns), "swish")
    z = onets.sample_no_postprocessing(logits, noise)
    torch.testing.assert_close(raw.cpu(), z, atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(act.cpu(), onets.postprocess(z), atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(lp.cpu(), onets.log_prob(logits, z), atol=5e-5, rtol=5e-5)
    mode = ops.policy_act(ppar2.to(dev), ops.MlpSpec(pd2), obs0.to(dev), nm.to(dev), ns.to(dev), deterministic=True, action_clip=0.5)
    torch.testing.assert_close(mode.cpu(), torch.tanh(logits[:, :3]).clamp(-0.5, 0.5), atol=2e-5, rtol=2e-5)
    with pytest.raises(ValueError):
        ops.policy_act(ppar2.to(dev), ops.MlpSpec(pd2), obs0[:, :2].contiguous().to(dev))


def test_rollout_actions_ignores_done_of_a_terminating_system(dev):
    """the reference's rollout_actions (utils/optimizer_utils.py:12-59) is a plain scan of System.step — it ignores
    SystemState.done and carries system_params; no Episode / AutoReset bookkeeping.  A user System that reports done mid-rollout
    must keep propagating x_next (compared with a plain step loop of the oracle's Pendulum)."""
    from mbpo.utils.optimizer_utils import rollout_actions
    user = _user_pendulum()(fall_done=True)
    sp = user.init_params(0)
    H, N = 25, 9
    g = torch.Generator().manual_seed(5)
    actions = torch.rand(H, N, 1, generator=g) * 2 - 1
    th = torch.linspace(0.5, 3.0, N)
    x0 = torch.stack([torch.cos(th), torch.sin(th), torch.linspace(-2.5, 2.5, N)], 1)
    tr = rollout_actions(user, sp, x0.to(dev), actions.to(dev), H)
    assert user.calls == H
    ref = osys.PendulumSystem()
    x, nxt, rew, fell = x0, [], [], 0
    for t in range(H):
        xn, r = ref.step(x, actions[t])
        fell += int((xn[:, 2].abs() > 3.0).sum())
        nxt.append(xn); rew.append(r); x = xn
    assert fell > 0                                              # the System did report done along the way
    torch.testing.assert_close(tr.next_observation.cpu(), torch.stack(nxt), atol=2e-4, rtol=2e-4)
    torch.testing.assert_close(tr.reward.cpu(), torch.stack(rew), atol=5e-4, rtol=2e-4)
    assert torch.equal(tr.observation[1:], tr.next_observation[:-1]) and torch.equal(tr.observation[0].cpu(), x0)
    assert torch.equal(tr.discount.cpu(), torch.ones(H, N))


def test_optimizer_act_runs_the_trainers_policy_head(dev):
    """SACOptimizer.act / PPOOptimizer.act (brax_optimizers.py:74-83, sac_networks.py:58-73) go through
    mbpo_policy_act — the normalise -> chain -> NormalTanh kernel and Philox stream of the training rollouts — not a torch
    restatement with a torch.Generator.  mode (evaluate=True) and sample (evaluate=False) against oracle/nets.py with the Philox
    noise the call's key selects; the key is split exactly once per call (:81-83)."""
    from mbpo.optimizers import PPOOptimizer, SACOptimizer
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import PendulumSystem
    from mbpo.types import Transition
    from mbpo.utils import keys as K
    system = PendulumSystem()
    s0 = system.reset()
    dummy = Transition(observation=s0.x_next, action=torch.zeros(1, device=dev), reward=s0.reward, discount=torch.tensor(0.99, device=dev),
                       next_observation=s0.x_next)
    buf = UniformSamplingQueue(10, dummy, 1, device=dev)
    for cls, kw in ((SACOptimizer, dict(num_timesteps=1000, episode_length=20, num_envs=4, batch_size=16, normalize_observations=True,
                                        policy_hidden_layer_sizes=(64, 64), critic_hidden_layer_sizes=(64, 64))),
                    (PPOOptimizer, dict(num_timesteps=1000, episode_length=20, num_envs=4, batch_size=4, num_minibatches=1,
                                        unroll_length=5, normalize_observations=True, policy_hidden_layer_sizes=(32, 32),
                                        critic_hidden_layer_sizes=(32, 32)))):
        opt = cls(system=system, true_buffer=buf, **kw)
        st = opt.init(key=7)
        norm, pol = st.policy_params
        # a non-trivial normaliser so that the normalise step is exercised
        norm.vec[1:4] = torch.tensor([0.1, -0.2, 0.3], device=dev)
        norm.vec[7:10] = torch.tensor([0.7, 1.3, 2.0], device=dev)
        tr = opt.dummy_trainer
        dims = tr.policy_dims
        g = torch.Generator().manual_seed(3)
        obs = torch.randn(33, 3, generator=g)
        logits = onets.mlp_forward(pol.cpu(), dims, onets.normalize(obs, norm.mean.cpu(), norm.std.cpu()))
        a_mode, st1 = opt.act(obs.to(dev), st, evaluate=True)
        torch.testing.assert_close(a_mode.cpu(), onets.mode(logits), atol=2e-5, rtol=2e-5)
        key, subkey = K.split(st.key)
        assert st1.key == key
        a_smp, st2 = opt.act(obs.to(dev), st, evaluate=False)
        eps = torch.from_numpy(philox.philox_normal(K.PRNGKey(subkey), 0, philox.STREAM_POLICY_NOISE, np.arange(33, dtype=np.uint64))).reshape(33, 1)
        torch.testing.assert_close(a_smp.cpu(), onets.postprocess(onets.sample_no_postprocessing(logits, eps)), atol=5e-5, rtol=5e-5)
        # a single observation [x] gives a single action [u], the same numbers as row 0 of the batch of one
        a1, _ = opt.act(obs[0].to(dev), st, evaluate=True)
        assert a1.shape == (1,) and torch.allclose(a1.cpu(), onets.mode(logits)[0], atol=2e-5)

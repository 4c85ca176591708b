"""GPU parity: ensemble MLP forward (R2) and fused model rollout (R1-R8) vs the CPU oracle, through the C-ABI.

Tolerances (fp32, north_star "within a stated fp32 tolerance"): the MFMA path is an exact k-ordered fmaf chain
per layer; the oracle sums in torch's order.  Single forward: atol 2e-5 + rtol 2e-5 (vs fp32 oracle) and
vs the fp64 oracle atol 5e-5.  Multi-step rollouts feed rounding differences back through the dynamics, so rows are
compared at atol 2e-4 + rtol 2e-4 for S<=5.
"""
import math

import numpy as np
import pytest
import torch

from oracle import nets as onets
from oracle import rollout as oro
from oracle import systems as osys

pytestmark = pytest.mark.gpu


def _ens_params(dims, E, seed):
    g = torch.Generator().manual_seed(seed)
    ps = []
    for _ in range(E):
        p = onets.init_mlp_flat(dims, g)
        # non-zero biases so the bias path is exercised
        p = p + 0.05 * torch.randn(p.shape, generator=g)
        ps.append(p)
    return torch.cat(ps)


@pytest.mark.parametrize("dims,E,N,act", [
    ([5, 64, 64, 64, 8], 5, 4096, "swish"),
    ([4, 64, 64, 6], 1, 37, "swish"),
    ([23, 64, 64, 64, 34], 10, 300, "swish"),
    ([5, 128, 128, 8], 3, 100, "relu"),
    ([4, 256, 256, 6], 2, 50, "tanh"),
    ([7, 3], 2, 20, "swish"),          # single Dense layer
    ([5, 64, 64, 64, 8], 5, 1, "swish"),
    # the lean throughput kernel's shapes (csrc/ens_lean.hip: 4 or 5 inputs, 64 x 3): ragged last tile, odd tile count, 6 outputs (partial
    # 16-byte output store), more tile pairs than workgroups per member
    ([4, 64, 64, 64, 6], 3, 1000 * 16 + 7, "swish"),
    ([5, 64, 64, 64, 8], 7, 3 * 16, "swish"),
    ([5, 64, 64, 64, 16], 2, 33, "swish"),
    ([5, 64, 64, 64, 8], 5, 32768, "swish"),       # C4's global env count on one GPU
    ([3, 64, 64, 64, 4], 4, 555, "swish"),         # the other input widths the lean kernel is instantiated for
    ([6, 64, 64, 64, 10], 2, 70, "swish"),
    ([7, 64, 64, 64, 12], 3, 129, "swish"),
    ([4, 64, 64, 6], 5, 2000, "swish"),            # two hidden layers (the reference experiments' 64 x 2)
])
def test_ensemble_forward_parity(dev, dims, E, N, act):
    from mbpo import ops
    params = _ens_params(dims, E, 0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, dims[0], generator=g)
    y_ref = onets.ensemble_forward(params, dims, E, x, act)
    y_ref64 = onets.ensemble_forward(params.double(), dims, E, x.double(), act)
    spec = ops.MlpSpec(dims, act, E)
    y = ops.ensemble_mlp_forward(params.to(dev), spec, x.to(dev)).cpu()
    assert y.shape == (E, N, dims[-1])
    torch.testing.assert_close(y, y_ref, atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(y.double(), y_ref64, atol=5e-5, rtol=5e-5)


def test_ensemble_forward_per_member_input(dev):
    from mbpo import ops
    dims, E, N = [5, 64, 64, 8], 4, 70
    params = _ens_params(dims, E, 3)
    x = torch.randn(E, N, dims[0], generator=torch.Generator().manual_seed(2))
    y_ref = onets.ensemble_forward(params, dims, E, x, "swish", shared_input=False)
    y = ops.ensemble_mlp_forward(params.to(dev), ops.MlpSpec(dims, "swish", E), x.to(dev), shared_input=False).cpu()
    torch.testing.assert_close(y, y_ref, atol=2e-5, rtol=2e-5)
    # the lean kernel's shape with a per-member input
    dims, E, N = [5, 64, 64, 64, 8], 3, 1234
    params = _ens_params(dims, E, 4)
    x = torch.randn(E, N, dims[0], generator=torch.Generator().manual_seed(5))
    y_ref = onets.ensemble_forward(params, dims, E, x, "swish", shared_input=False)
    y = ops.ensemble_mlp_forward(params.to(dev), ops.MlpSpec(dims, "swish", E), x.to(dev), shared_input=False).cpu()
    torch.testing.assert_close(y, y_ref, atol=2e-5, rtol=2e-5)


def test_mfma_layout_identity(dev):
    """A = I check with an ASYMMETRIC weight: y = x @ W must reproduce W's rows when x = one-hot rows."""
    from mbpo import ops
    K, Nn = 16, 16
    W = torch.arange(K * Nn, dtype=torch.float32).reshape(K, Nn) * 0.01 + torch.arange(Nn) * 1.0
    b = torch.zeros(Nn)
    params = torch.cat([W.reshape(-1), b])
    x = torch.eye(K)
    y = ops.ensemble_mlp_forward(params.to(dev), ops.MlpSpec([K, Nn], "swish", 1), x.to(dev)).cpu()[0]
    torch.testing.assert_close(y, W, atol=0, rtol=0)


def _pendulum_obs(N, gen):
    th = (torch.rand(N, generator=gen) * 2 - 1) * math.pi
    thd = (torch.rand(N, generator=gen) * 2 - 1) * 8
    return torch.stack([torch.cos(th), torch.sin(th), thd], dim=1)


def _run_rollout_case(dev, *, N, S, L, AR, X, U, system, E=0, mode="mean", sample_noise=False, ppo=False,
                      env_major=False, normalize=False, deterministic=False, hidden=(64, 64, 64), seed=0,
                      init_steps=None, atol=2e-4, replicate=0):
    from mbpo import ops, _hip
    g = torch.Generator().manual_seed(seed)
    pdims = [X, *hidden, 2 * U]
    ppar = onets.init_mlp_flat(pdims, g) + 0.02 * torch.randn(onets.n_params(pdims), generator=g)
    if X == 3:
        obs0 = _pendulum_obs(N, g)
        first = _pendulum_obs(N, g)
    else:
        obs0 = torch.randn(N, X, generator=g)
        first = torch.randn(N, X, generator=g)
    steps0 = torch.randint(0, L, (N,), generator=g).float() if init_steps is None else init_steps.clone()
    done0 = (torch.rand(N, generator=g) < 0.2).float()
    pnoise = torch.randn(S, N, U, generator=g)
    mnoise = torch.randn(S, AR, N, X, generator=g) if sample_noise else None
    midx = torch.randint(0, max(E, 1), (S, AR, N), generator=g, dtype=torch.int32) if mode == "ts1" else None
    nm = torch.randn(X, generator=g) * 0.3 if normalize else None
    ns = torch.rand(X, generator=g) + 0.5 if normalize else None
    pp = osys.PendulumParams()
    kw = {}
    if system == "pendulum":
        osystem = osys.PendulumSystem(pp)
        rparams = torch.tensor(pp.reward_vector())
        kw.update(system_kind=_hip.SYS_PENDULUM, sys_params=torch.tensor(pp.sys_vector()).to(dev),
                  reward_kind=_hip.REWARD_PENDULUM)
    else:
        ddims = [X + U, *hidden, 2 * X]
        dpar = torch.cat([onets.init_mlp_flat(ddims, g) * 0.5 + 0.01 * torch.randn(onets.n_params(ddims), generator=g)
                          for _ in range(E)])
        if X == 3:
            rfn = lambda x, u: osys.pendulum_reward(x, u, pp)
            rparams = torch.tensor(pp.reward_vector())
            rk = _hip.REWARD_PENDULUM
        else:
            tgt, q, r = torch.randn(X, generator=g), torch.rand(X, generator=g), torch.rand(U, generator=g) * 0.1
            rfn = lambda x, u: osys.quadratic_reward(x, u, tgt, q, r)
            rparams = torch.cat([tgt, q, r])
            rk = _hip.REWARD_QUADRATIC
        osystem = osys.EnsembleSystem(dpar, ddims, E, X, U, mode=mode, predict_delta=True, sample_noise=sample_noise,
                                      min_std=1e-3, reward_fn=rfn)
        kw.update(system_kind=_hip.SYS_ENSEMBLE, dyn_params=dpar.to(dev), dyn_spec=ops.MlpSpec(ddims, "swish", E),
                  ens_mode={"mean": _hip.ENS_MEAN, "ts1": _hip.ENS_TS1, "tsinf": _hip.ENS_TSINF}[mode],
                  ens_predict_delta=True, ens_sample_noise=sample_noise, ens_min_std=1e-3, reward_kind=rk)
    st0 = oro.EnvState(obs0, first, steps0, done0)
    st_ref, rows_ref = oro.rollout(osystem, ppar, pdims, st0, S, L, AR, norm_mean=nm, norm_std=ns, policy_noise=pnoise,
                                   model_noise=mnoise, member_idx=midx, deterministic=deterministic, ppo_extras=ppo,
                                   env_major=env_major)
    obs_d, steps_d, done_d = obs0.to(dev), steps0.to(dev), done0.to(dev)
    rows = ops.model_rollout(policy_params=ppar.to(dev), policy_spec=ops.MlpSpec(pdims, "swish", 1), x_dim=X, u_dim=U,
                             obs=obs_d, first_obs=first.to(dev), steps=steps_d, done=done_d, n_steps=S, episode_length=L,
                             action_repeat=AR, reward_params=rparams.to(dev),
                             norm_mean=None if nm is None else nm.to(dev), norm_std=None if ns is None else ns.to(dev),
                             deterministic=deterministic, ppo_extras=ppo, env_major=env_major,
                             policy_noise=pnoise.to(dev), model_noise=None if mnoise is None else mnoise.to(dev),
                             member_idx=None if midx is None else midx.to(dev), **kw)
    torch.cuda.synchronize()
    rows = rows.cpu()
    D = rows.shape[1]
    # integer-valued bookkeeping columns are exact: discount, truncation, steps, done
    assert torch.equal(rows[:, X + U + 1], rows_ref[:, X + U + 1])
    assert torch.equal(rows[:, D - 1], rows_ref[:, D - 1])
    assert torch.equal(steps_d.cpu(), st_ref.steps)
    assert torch.equal(done_d.cpu(), st_ref.done)
    torch.testing.assert_close(rows, rows_ref, atol=atol, rtol=atol)
    torch.testing.assert_close(obs_d.cpu(), st_ref.obs, atol=atol, rtol=atol)
    if replicate:
        # full-size property: envs are independent, so `replicate` copies of the N envs (in other tiles / workgroups) must
        # reproduce the N-env rows bit for bit
        k = replicate
        rep = lambda t, dim=0: None if t is None else t.repeat_interleave(1, 0).repeat(*[k if d == dim else 1 for d in range(t.dim())]).to(dev)
        obs_k, steps_k, done_k = rep(obs0), rep(steps0), rep(done0)
        rows_k = ops.model_rollout(policy_params=ppar.to(dev), policy_spec=ops.MlpSpec(pdims, "swish", 1), x_dim=X, u_dim=U,
                                   obs=obs_k, first_obs=rep(first), steps=steps_k, done=done_k, n_steps=S, episode_length=L,
                                   action_repeat=AR, reward_params=rparams.to(dev),
                                   norm_mean=None if nm is None else nm.to(dev), norm_std=None if ns is None else ns.to(dev),
                                   deterministic=deterministic, ppo_extras=ppo, env_major=env_major,
                                   policy_noise=rep(pnoise, 1), model_noise=rep(mnoise, 2), member_idx=rep(midx, 2), **kw)
        torch.cuda.synchronize()
        assert rows_k.shape[0] == S * N * k
        if env_major:
            want = rows.reshape(N, S, D).repeat(k, 1, 1)
        else:
            want = rows.reshape(S, N, D).repeat(1, k, 1)
        assert torch.equal(rows_k.cpu().reshape(want.shape), want)
        assert torch.equal(obs_k.cpu(), obs_d.cpu().repeat(k, 1)) and torch.equal(steps_k.cpu(), steps_d.cpu().repeat(k))
    return rows


def test_rollout_pendulum_sac(dev):
    _run_rollout_case(dev, N=100, S=7, L=5, AR=1, X=3, U=1, system="pendulum")


def test_rollout_pendulum_normalized_actionrepeat(dev):
    _run_rollout_case(dev, N=33, S=6, L=6, AR=2, X=3, U=1, system="pendulum", normalize=True)


def test_rollout_pendulum_ppo_env_major(dev):
    _run_rollout_case(dev, N=48, S=10, L=4, AR=1, X=3, U=1, system="pendulum", ppo=True, env_major=True)


def test_rollout_pendulum_deterministic_128(dev):
    _run_rollout_case(dev, N=20, S=5, L=200, AR=1, X=3, U=1, system="pendulum", deterministic=True,
                      hidden=(128, 128, 128))


def test_rollout_ensemble_mean_c2_shape(dev):
    # BASELINE config 2 shape: x=4,u=1, E=5, H=5, S=5 (smaller N for the oracle)
    _run_rollout_case(dev, N=512, S=5, L=5, AR=1, X=4, U=1, system="ensemble", E=5, mode="mean")


def test_rollout_c2_full_size_replication(dev):
    """BASELINE configs[1] at FULL size, N = 4096 envs (x=4, u=1, E=5, H=S=5): the 512-env rows are compared with the oracle,
    then 8 copies of those envs (4096) must reproduce them exactly."""
    _run_rollout_case(dev, N=512, S=5, L=5, AR=1, X=4, U=1, system="ensemble", E=5, mode="mean", normalize=True, replicate=8)


def test_rollout_c3_full_size_replication(dev):
    """BASELINE configs[2]: PPO collection, N = 16384 envs, unroll T = 40, PPO row layout, env-major — 128 oracle-checked envs x 128."""
    _run_rollout_case(dev, N=128, S=40, L=40, AR=1, X=3, U=1, system="pendulum", ppo=True, env_major=True, normalize=True,
                      replicate=128, atol=2e-3)


def test_rollout_ensemble_pendulum_shape(dev):
    _run_rollout_case(dev, N=130, S=5, L=5, AR=1, X=3, U=1, system="ensemble", E=5, mode="mean", normalize=True)


def test_rollout_ensemble_ts1_noise(dev):
    _run_rollout_case(dev, N=77, S=5, L=3, AR=1, X=4, U=1, system="ensemble", E=5, mode="ts1", sample_noise=True)


def test_rollout_ensemble_tsinf_ppo(dev):
    _run_rollout_case(dev, N=64, S=8, L=5, AR=1, X=4, U=2, system="ensemble", E=3, mode="tsinf", ppo=True, env_major=True)


def test_rollout_ensemble_bptt_shape(dev):
    # BASELINE config 5 shape: x=17,u=6,E=10
    _run_rollout_case(dev, N=40, S=4, L=32, AR=1, X=17, U=6, system="ensemble", E=10, mode="mean", atol=5e-4)


def test_rollout_single_env(dev):
    # BASELINE config 1: N=1
    _run_rollout_case(dev, N=1, S=10, L=1000, AR=1, X=4, U=1, system="ensemble", E=5, mode="mean")


def test_rollout_empty(dev):
    from mbpo import ops, _hip
    pdims = [3, 64, 64, 2]
    ppar = torch.zeros(onets.n_params(pdims), device=dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    rows = ops.model_rollout(policy_params=ppar, policy_spec=ops.MlpSpec(pdims), x_dim=3, u_dim=1, obs=z(0, 3),
                             first_obs=z(0, 3), steps=z(0), done=z(0), n_steps=5, episode_length=5,
                             reward_params=z(3), sys_params=z(6))
    assert rows.shape == (0, 10)


def test_rollout_bad_args(dev):
    from mbpo import ops, _hip
    pdims = [3, 64, 64, 2]
    ppar = torch.zeros(onets.n_params(pdims), device=dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    with pytest.raises(_hip.MbpoHipError):
        ops.model_rollout(policy_params=ppar, policy_spec=ops.MlpSpec(pdims), x_dim=3, u_dim=1, obs=z(4, 3),
                          first_obs=z(4, 3), steps=z(4), done=z(4), n_steps=5, episode_length=0,
                          reward_params=z(3), sys_params=z(6))
    with pytest.raises(_hip.MbpoHipError):  # CPU tensor: no fallback
        ops.model_rollout(policy_params=ppar.cpu(), policy_spec=ops.MlpSpec(pdims), x_dim=3, u_dim=1, obs=z(4, 3),
                          first_obs=z(4, 3), steps=z(4), done=z(4), n_steps=5, episode_length=5,
                          reward_params=z(3), sys_params=z(6))


def test_rollout_philox_noise_matches_oracle(dev):
    """Device Philox normal draws == oracle/philox.py draws fed as explicit noise (integer part bit-exact)."""
    from mbpo import ops, _hip
    from oracle import philox
    N, S, X, U = 50, 3, 3, 1
    g = torch.Generator().manual_seed(5)
    pdims = [X, 64, 64, 2 * U]
    ppar = onets.init_mlp_flat(pdims, g)
    obs0 = _pendulum_obs(N, g)
    pp = osys.PendulumParams()
    seed, offset = 1234567890123, 42
    idx = np.arange(S * N * U, dtype=np.uint64)
    noise = torch.from_numpy(philox.philox_normal(seed, offset, philox.STREAM_POLICY_NOISE, idx)).reshape(S, N, U)
    outs = []
    for use_explicit in (True, False):
        obs_d = obs0.to(dev)
        rows = ops.model_rollout(policy_params=ppar.to(dev), policy_spec=ops.MlpSpec(pdims), x_dim=X, u_dim=U, obs=obs_d,
                                 first_obs=obs0.to(dev), steps=torch.zeros(N, device=dev), done=torch.zeros(N, device=dev),
                                 n_steps=S, episode_length=100, reward_params=torch.tensor(pp.reward_vector()).to(dev),
                                 sys_params=torch.tensor(pp.sys_vector()).to(dev),
                                 policy_noise=noise.to(dev) if use_explicit else None, seed=seed, offset=offset)
        outs.append(rows.cpu())
    torch.testing.assert_close(outs[0], outs[1], atol=1e-5, rtol=1e-5)


def _set_rollout_lean(mode: int) -> None:
    import ctypes as C
    from mbpo import _hip
    lib = _hip.load()
    lib.mbpo_debug_set_rollout_lean.argtypes = [C.c_int]
    lib.mbpo_debug_set_rollout_lean.restype = C.c_int
    assert lib.mbpo_debug_set_rollout_lean(mode) == 0


@pytest.mark.parametrize("kw", [
    dict(N=512, S=5, L=5, X=4, system="ensemble", E=5, mode="mean", normalize=True),            # BASELINE configs[1]
    dict(N=4800, S=5, L=3, X=4, system="ensemble", E=5, mode="mean"),                           # more tiles than CUs, episode wrap
    dict(N=77, S=6, L=4, X=4, system="ensemble", E=5, mode="ts1", sample_noise=True),           # ragged tile, sampled member + noise
    dict(N=130, S=5, L=5, X=3, system="ensemble", E=3, mode="tsinf", ppo=True, env_major=True, normalize=True),
    dict(N=100, S=7, L=5, X=3, system="pendulum"),
    dict(N=128, S=40, L=40, X=3, system="pendulum", ppo=True, env_major=True, normalize=True, atol=2e-3),   # BASELINE configs[2] layout
    dict(N=33, S=4, L=9, X=4, system="ensemble", E=1, mode="mean", deterministic=True),
    dict(N=9000, S=3, L=2, X=4, system="ensemble", E=5, mode="mean", normalize=True),          # 563 tiles: the default picks two in flight
    dict(N=16, S=1, L=5, X=4, system="ensemble", E=2, mode="mean"),                           # one tile, one step
    dict(N=300, S=6, L=4, X=2, system="ensemble", E=4, mode="ts1", sample_noise=True, normalize=True),   # two observation elements    dict(N=1000, S=5, L=3, X=3, system="pendulum", ppo=True, env_major=True, normalize=True, hidden=(64, 64)),      # tests/test_ppo.py's policy
    dict(N=9000, S=4, L=3, X=4, system="ensemble", E=5, mode="mean", hidden=(64, 64)),                          # 64 x 2 policy and members, both forms
])
def test_rollout_lean_equals_generic_kernel(dev, kw):
    """k_rollout_lean (csrc/rollout_lean.hip: weights resident in registers, 16-byte activation stores, one bookkeeping section per
    step) forms every dot product with the generic k_model_rollout64's MFMA sequence and every elementwise value with its expressions:
    rows identical bit for bit (each run is also checked against the oracle inside _run_rollout_case)."""
    try:
        _set_rollout_lean(0)
        rows_g = _run_rollout_case(dev, AR=1, U=1, **kw)
        _set_rollout_lean(3)                      # one tile per workgroup at a time
        rows_l = _run_rollout_case(dev, AR=1, U=1, **kw)
        _set_rollout_lean(2)                      # two tiles in flight per workgroup (policy of one beside the members of the other)
        rows_p = _run_rollout_case(dev, AR=1, U=1, **kw)
    finally:
        _set_rollout_lean(-1)
    assert torch.equal(rows_g, rows_l)
    assert torch.equal(rows_g, rows_p)

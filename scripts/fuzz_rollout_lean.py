"""Random-shape fuzz of k_rollout_lean (both forms: one tile per workgroup at a time, two tiles in flight) against the generic
k_model_rollout64: rows, obs, steps, done must be identical bit for bit.  Usage: python scripts/fuzz_rollout_lean.py [n_cases] [seed]"""
import ctypes as C, math, random, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.load()
lib.mbpo_debug_set_rollout_lean.argtypes = [C.c_int]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(rnd.randrange(1 << 30))


def lecun(dims, n=1):
    parts = []
    for _ in range(n):
        for i, o in zip(dims[:-1], dims[1:]):
            parts += [((torch.rand(i, o, generator=g) * 2 - 1) * math.sqrt(3.0 / i)).reshape(-1), 0.1 * torch.randn(o, generator=g)]
    return torch.cat(parts)


bad = 0
for case in range(n_cases):
    X = rnd.choice([3, 4])
    system = rnd.choice(["ensemble", "ensemble", "pendulum"]) if X == 3 else "ensemble"
    E = rnd.randint(1, 5) if system == "ensemble" else 0
    N = rnd.choice([1, 15, 16, 17, 100, 513, 4096, 4100, 9001])
    S = rnd.randint(1, 7)
    L = rnd.choice([1, 2, 3, 5, 1000])
    ppo, env_major, normalize = rnd.random() < 0.4, rnd.random() < 0.4, rnd.random() < 0.5
    deterministic = rnd.random() < 0.2
    mode = rnd.choice(["mean", "ts1", "tsinf"]) if system == "ensemble" else "mean"
    sample_noise = mode != "mean" and rnd.random() < 0.5
    n_out = 2 * X if (sample_noise or rnd.random() < 0.7) else X
    pd = [X, 64, 64, 64, 2]
    pp = lecun(pd).to(dev)
    kw = dict(x_dim=X, u_dim=1, n_steps=S, episode_length=L, policy_params=pp, policy_spec=ops.MlpSpec(pd), ppo_extras=ppo, env_major=env_major,
              deterministic=deterministic, seed=rnd.randrange(1 << 20), offset=rnd.randrange(1 << 20))
    if normalize:
        kw.update(norm_mean=(torch.randn(X, generator=g) * 0.3).to(dev), norm_std=(torch.rand(X, generator=g) + 0.5).to(dev))
    if system == "ensemble":
        dd = [X + 1, 64, 64, 64, n_out]
        kw.update(system_kind=_hip.SYS_ENSEMBLE, dyn_params=(lecun(dd, E) * 0.5).to(dev), dyn_spec=ops.MlpSpec(dd, "swish", E),
                  ens_mode={"mean": _hip.ENS_MEAN, "ts1": _hip.ENS_TS1, "tsinf": _hip.ENS_TSINF}[mode], ens_predict_delta=rnd.random() < 0.8,
                  ens_sample_noise=sample_noise, ens_min_std=1e-3)
        if X == 3 and rnd.random() < 0.5:
            kw.update(reward_kind=_hip.REWARD_PENDULUM, reward_params=torch.tensor([1.0, 0.001, 0.0]).to(dev))
        else:
            kw.update(reward_kind=_hip.REWARD_QUADRATIC, reward_params=torch.cat([torch.randn(X, generator=g), torch.rand(X, generator=g),
                                                                                  torch.rand(1, generator=g) * 0.1]).to(dev))
    else:
        from mbpo.systems import PendulumSystem
        ps = PendulumSystem()
        kw.update(ps.rollout_spec(ps.reset(device=dev).system_params, dev))
    obs0 = torch.randn(N, X, generator=g)
    first = torch.randn(N, X, generator=g)
    steps0 = torch.randint(0, max(1, min(L, 4)), (N,), generator=g).float()
    done0 = (torch.rand(N, generator=g) < 0.2).float()
    res = {}
    for lm in (0, 3, 2):
        lib.mbpo_debug_set_rollout_lean(lm)
        obs, steps, done = obs0.to(dev), steps0.to(dev), done0.to(dev)
        rows = ops.model_rollout(obs=obs, first_obs=first.to(dev), steps=steps, done=done, **kw)
        torch.cuda.synchronize()
        res[lm] = (rows.clone(), obs.clone(), steps.clone(), done.clone())
    ok = all(torch.equal(a, b) for a, b in zip(res[0], res[3])) and all(torch.equal(a, b) for a, b in zip(res[0], res[2]))
    finite = bool(torch.isfinite(res[0][0]).all())
    if not ok:
        bad += 1
    print(f"case {case}: X={X} {system} E={E} N={N} S={S} L={L} ppo={ppo} env_major={env_major} norm={normalize} det={deterministic} mode={mode} "
          f"noise={sample_noise} n_out={n_out}: {'identical' if ok else 'MISMATCH'} finite={finite}", flush=True)
lib.mbpo_debug_set_rollout_lean(-1)
print(f"{n_cases - bad} / {n_cases} cases identical")
sys.exit(1 if bad else 0)

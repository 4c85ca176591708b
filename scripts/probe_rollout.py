import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip


def init_flat(dims, g):
    # LeCun-style init, flat [w0, b0, w1, b1, ...] layout (timing probe only)
    parts = []
    for i, o in zip(dims[:-1], dims[1:]):
        parts += [(torch.randn(i, o, generator=g) / i ** 0.5).reshape(-1), torch.zeros(o)]
    return torch.cat(parts)

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (N, X, U, E, S) in [(4096, 4, 1, 5, 5), (32768, 4, 1, 5, 5), (4096, 3, 1, 5, 5)]:
    pd = [X, 64, 64, 64, 2*U]; dd = [X+U, 64, 64, 64, 2*X]
    pp = init_flat(pd, g).to(dev)
    dp = torch.cat([init_flat(dd, g) for _ in range(E)]).to(dev)
    obs = torch.randn(N, X, generator=g).to(dev); first = obs.clone()
    steps = torch.zeros(N, device=dev); done = torch.zeros(N, device=dev)
    rp = torch.cat([torch.zeros(X), torch.ones(X), torch.ones(U)*0.1]).to(dev)
    out = torch.empty(S*N, 2*X+U+3, device=dev)
    def run():
        ops.model_rollout(policy_params=pp, policy_spec=ops.MlpSpec(pd), x_dim=X, u_dim=U, obs=obs, first_obs=first,
                          steps=steps, done=done, n_steps=S, episode_length=S, system_kind=_hip.SYS_ENSEMBLE, dyn_params=dp,
                          dyn_spec=ops.MlpSpec(dd, 'swish', E), reward_kind=_hip.REWARD_QUADRATIC, reward_params=rp,
                          seed=1, offset=0, out=out)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/50
    flop = S*N*(2*E*(sum(dd[i]*dd[i+1] for i in range(4))) + 2*sum(pd[i]*pd[i+1] for i in range(4)))
    print(f"N={N} x={X} E={E} S={S}: {ms*1e3:.1f} us/rollout  {S*N/ms/1e3:.2f} M transitions/s  {flop/ms/1e9:.2f} TFLOP/s")

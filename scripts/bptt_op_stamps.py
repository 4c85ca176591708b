"""Cycles per op kind of k_bptt_actor (block 0) from in-kernel s_memtime stamps: where a horizon step's time goes."""
import ctypes as C, math, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip
dev = torch.device('cuda:0')
NAMES = ["pi fwd | sample", "member fwd | mean", "- | pendulum step", "- | reward, normalise", "V fwd | store, advance", "- | reload",
         "recompute | dL/dV", "V dgrad | dL/dx'", "member refwd | dL/dy", "member dgrad | dL/dxa", "- | pendulum vjp",
         "- | reward vjp, logits", "pi bwd | dL/dx"]


def lecun(dims, g, n=1):
    parts = []
    for _ in range(n):
        for i, o in zip(dims[:-1], dims[1:]):
            parts += [((torch.rand(i, o, generator=g) * 2 - 1) * math.sqrt(3.0 / i)).reshape(-1), torch.zeros(o)]
    return torch.cat(parts)


def run(X, U, E, H, n):
    g = torch.Generator().manual_seed(0)
    hid = (64, 64, 64)
    ad, cd, dd = [X, *hid, 2 * U], [X, *hid, 1], [X + U, *hid, 2 * X]
    op = ops.BpttActorGrad(x_dim=X, u_dim=U, horizon=H, actor_dims=ad, critic_dims=cd, n=n, device=dev, seed=3)
    kw = dict(actor_params=lecun(ad, g).to(dev), target_critic_params=lecun(cd, g, 2).to(dev), init_states=torch.randn(n, X, generator=g).to(dev),
              state_mean=torch.zeros(X, device=dev), state_std=torch.ones(X, device=dev), reward_mean_std=torch.tensor([0.0, 1.0], device=dev),
              system_kind=_hip.SYS_ENSEMBLE, reward_kind=_hip.REWARD_QUADRATIC,
              reward_params=torch.cat([torch.zeros(X), torch.ones(X), torch.ones(U) * 0.1]).to(dev),
              dyn_params=(lecun(dd, g, E) * 0.5).to(dev), dyn_spec=ops.MlpSpec(dd, "swish", E))
    lib = _hip.load()
    for _ in range(2):
        op(**kw)
    torch.cuda.synchronize()
    stamps = torch.zeros(32, dtype=torch.int64, device=dev)
    lib.mbpo_debug_set_bptt_stamps(C.c_void_p(stamps.data_ptr()))
    reps = 3
    for _ in range(reps):
        op(**kw)
    torch.cuda.synchronize()
    lib.mbpo_debug_set_bptt_stamps(C.c_void_p(0))
    s = stamps.cpu().double()
    tot = float(s[:16].sum())
    print(f"x={X} u={U} E={E} H={H} n={n}: block 0, {tot / reps / 1e3:.0f} k cycles per call, per horizon step {tot / reps / H / 1e3:.1f} k")
    for i, nm in enumerate(NAMES):
        if s[16 + i] > 0:
            print(f"  {nm:28s} {s[i] / tot * 100:5.1f}%   {s[i] / s[16 + i]:8.0f} cycles/op  x{int(s[16 + i] / reps / H)} per step")


run(17, 6, 10, 32, 4096)
run(4, 1, 5, 5, 4096)

"""Section timeline of the generic k_model_rollout64 (tile 0, env step 1) from in-kernel s_memtime stamps; bench-shaped rollout.
The benchmark networks run k_rollout_lean by default (its timeline: scripts/rollout_lean_dev.py); this script pins the generic kernel."""
import ctypes as C, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip
dev = torch.device('cuda:0')
X, U, E, N, S = 4, 1, 5, 4096, 5
g = torch.Generator().manual_seed(0)
pol = ops.MlpSpec([X, 64, 64, 64, 2 * U], "swish", 1)
dyn = ops.MlpSpec([X + U, 64, 64, 64, 2 * X], "swish", E)
pp = (torch.randn(pol.total_params, generator=g) * 0.1).to(dev)
dp = (torch.randn(dyn.total_params, generator=g) * 0.05).to(dev)
rp = torch.cat([torch.zeros(X), torch.ones(X), 0.1 * torch.ones(U)]).to(dev)
lib = _hip.load()
lib.mbpo_debug_set_rollout_lean.argtypes = [C.c_int]
lib.mbpo_debug_set_rollout_lean(0)
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
names = ["A: inputs (normalise, first-layer request)", "policy chain", "B: sample action, member first-layer request", "model chains (all members, one round)",
         "C: member mean, reward, next state", "D: episode bookkeeping, auto-reset", "row write-out"]
acc = None
for it in range(12):
    obs = torch.randn(N, X, generator=g).to(dev)
    if it == 2:
        lib.mbpo_debug_set_rollout_stamps(C.c_void_p(stamps.data_ptr()))
    ops.model_rollout(policy_params=pp, policy_spec=pol, x_dim=X, u_dim=U, obs=obs, first_obs=obs.clone(), steps=torch.zeros(N, device=dev),
                      done=torch.zeros(N, device=dev), n_steps=S, episode_length=5, system_kind=_hip.SYS_ENSEMBLE, dyn_params=dp, dyn_spec=dyn,
                      reward_kind=_hip.REWARD_QUADRATIC, reward_params=rp, seed=1, offset=it)
    torch.cuda.synchronize()
    if it >= 2:
        s = stamps.cpu().clone()
        acc = s if acc is None else acc + s
lib.mbpo_debug_set_rollout_stamps(C.c_void_p(0))
acc = acc.double() / 10
t = acc[:8]
print("one env step (cycles):", round(float(t[7] - t[0])))
for i in range(1, 8):
    print(f"  {names[i - 1]:58s} {float(t[i] - t[i - 1]):8.0f}")

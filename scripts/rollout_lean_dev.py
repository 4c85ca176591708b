"""A/B of mbpo_model_rollout with the generic k_model_rollout64 and with k_rollout_lean (csrc/rollout_lean.hip), same process: device
time per launch under hipGraph replay at N = 4096 and N = 32768 (BASELINE configs[1] networks: x = 4, u = 1, E = 5, S = 5), rows compared
bit for bit, and the lean kernel's s_memtime timeline of env step 1 of workgroup 0."""
import ctypes as C, math, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.load()
lib.mbpo_debug_set_rollout_lean.argtypes = [C.c_int]
g = torch.Generator().manual_seed(0)
X, U, E, S, hid = 4, 1, 5, 5, (64, 64, 64)


def lecun(dims, n=1):
    parts = []
    for _ in range(n):
        for i, o in zip(dims[:-1], dims[1:]):
            parts += [((torch.rand(i, o, generator=g) * 2 - 1) * math.sqrt(3.0 / i)).reshape(-1), torch.zeros(o)]
    return torch.cat(parts)


pd, dd = [X, *hid, 2 * U], [X + U, *hid, 2 * X]
pp, dp = lecun(pd).to(dev), lecun(dd, E).to(dev)
rp = torch.cat([torch.zeros(X), torch.ones(X), torch.ones(U) * 0.1]).to(dev)


def timed_graph(fn, inner=20, reps=10):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(inner):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * inner) * 1e3


for N in (4096, 32768):
    obs0 = torch.randn(N, X, generator=g).to(dev)
    res, tms = {}, {0: [], 1: []}
    for lean in (0, 1, 0, 1):
        lib.mbpo_debug_set_rollout_lean(lean)
        obs, first = obs0.clone(), obs0.clone()
        steps, done = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        rows = torch.empty(S * N, 2 * X + U + 3, device=dev)

        def run():
            ops.model_rollout(policy_params=pp, policy_spec=ops.MlpSpec(pd), x_dim=X, u_dim=U, obs=obs, first_obs=first, steps=steps, done=done,
                              n_steps=S, episode_length=S, system_kind=_hip.SYS_ENSEMBLE, dyn_params=dp, dyn_spec=ops.MlpSpec(dd, "swish", E),
                              reward_kind=_hip.REWARD_QUADRATIC, reward_params=rp, seed=1, offset=0, out=rows)
        run(); torch.cuda.synchronize()
        res[lean] = (rows.clone(), obs.clone(), steps.clone())
        tms[lean].append(timed_graph(run))
    same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    flop = N * S * 2 * (E * sum(i * o for i, o in zip(dd[:-1], dd[1:])) + sum(i * o for i, o in zip(pd[:-1], pd[1:])))
    print("N=%d  us per launch: generic %s  lean %s   (lean: %.1f TFLOP/s = %.1f %% of 157.3, %.0f M transitions/s)   rows identical %s" %
          (N, " ".join("%.1f" % t for t in tms[0]), " ".join("%.1f" % t for t in tms[1]), flop / min(tms[1]) / 1e6,
           flop / min(tms[1]) / 1e6 / 157.3 * 100, N * S / min(tms[1]), same), flush=True)

# ---- timeline of env step 1, workgroup 0
lib.mbpo_debug_set_rollout_lean(1)
lib.mbpo_debug_set_rollout_stamps.argtypes = [C.c_void_p]
N = 4096
obs, first = obs0[:N].clone(), obs0[:N].clone()
steps, done = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
rows = torch.empty(S * N, 2 * X + U + 3, device=dev)
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
acc = torch.zeros(16, dtype=torch.float64)
for it in range(30):
    if it == 10:
        lib.mbpo_debug_set_rollout_stamps(C.c_void_p(stamps.data_ptr()))
    ops.model_rollout(policy_params=pp, policy_spec=ops.MlpSpec(pd), x_dim=X, u_dim=U, obs=obs, first_obs=first, steps=steps, done=done,
                      n_steps=S, episode_length=S, system_kind=_hip.SYS_ENSEMBLE, dyn_params=dp, dyn_spec=ops.MlpSpec(dd, "swish", E),
                      reward_kind=_hip.REWARD_QUADRATIC, reward_params=rp, seed=1, offset=0, out=rows)
    torch.cuda.synchronize()
    if it >= 10:
        acc += stamps.cpu().double()
lib.mbpo_debug_set_rollout_stamps(C.c_void_p(0))
acc /= 20
names = ["policy input layer | noise, previous rows", "policy hidden 1", "policy hidden 2", "policy output + sample", "member input layer",
         "member hidden 1", "member hidden 2", "member output", "reward, next state, bookkeeping, next input"]
print("k_rollout_lean<4>, env step 1 of workgroup 0 (cycles):")
for i, n in enumerate(names):
    print("   %-46s %7.0f" % (n, float(acc[i + 1] - acc[i])))
print("   %-46s %7.0f" % ("step total", float(acc[9] - acc[0])))

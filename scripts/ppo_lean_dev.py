"""A/B of the PPO minibatch step (BASELINE config 3: B = 512, T = 40 and T = 5) with the generic k_ppo_fwd_bwd and with k_ppo_lean,
same process: device time per minibatch_step under hipGraph replay, and the gradient difference between the two."""
import ctypes as C, math, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.load()
lib.mbpo_debug_set_ppo_lean.argtypes = [C.c_int]
g = torch.Generator().manual_seed(0)
X, U, hid = 4, 1, (64, 64, 64)


def lecun(dims):
    parts = []
    for i, o in zip(dims[:-1], dims[1:]):
        parts += [((torch.rand(i, o, generator=g) * 2 - 1) * math.sqrt(3.0 / i)).reshape(-1), torch.zeros(o)]
    return torch.cat(parts)


pd, vd = [X, *hid, 2 * U], [X, *hid, 1]
p0 = torch.cat([lecun(pd), lecun(vd)]).to(dev)
D = ops.transition_row_len(X, U, True)
for T in (40, 5):
    data = torch.randn(512, T, D, generator=g) * 0.5
    data[..., X + U + 1] = 1.0
    data[..., -1] = 0.0
    data = data.to(dev)
    res = {}
    for lean in (0, 1, 0, 1):
        lib.mbpo_debug_set_ppo_lean(lean)
        up = ops.PpoUpdater(x_dim=X, u_dim=U, policy_dims=pd, value_dims=vd, batch_size=512, unroll_length=T, device=dev)
        up.load_state(p0)
        up.minibatch_step(data, seed=1, offset=5 << 32)
        torch.cuda.synchronize()
        grads = up.grads.clone()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                up.minibatch_step(data, seed=1, offset=5 << 32)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(lean, []).append((e0.elapsed_time(e1) / 200 * 1e3, grads))
    lib.mbpo_debug_set_ppo_lean(-1)
    g0, g1 = res[0][0][1], res[1][0][1]
    print("T=%d  minibatch_step us: generic %s  lean %s   max |dg| %.3e (max |g| %.3e)  finite %s" %
          (T, " ".join("%.1f" % t for t, _ in res[0]), " ".join("%.1f" % t for t, _ in res[1]), float((g0 - g1).abs().max()),
           float(g0.abs().max()), bool(torch.isfinite(g1).all())), flush=True)

# ---- per-section timeline of k_ppo_lean (workgroup 0, its first 5 tiles): s_memtime stamps
lib.mbpo_debug_set_ppo_stamps.argtypes = [C.c_void_p]
T = 40
data = torch.randn(512, T, D, generator=g) * 0.5
data[..., X + U + 1] = 1.0
data[..., -1] = 0.0
data = data.to(dev)
lib.mbpo_debug_set_ppo_lean(1)
up = ops.PpoUpdater(x_dim=X, u_dim=U, policy_dims=pd, value_dims=vd, batch_size=512, unroll_length=T, device=dev)
up.load_state(p0)
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
acc = torch.zeros(64, dtype=torch.float64)
for it in range(30):
    if it == 10:
        lib.mbpo_debug_set_ppo_stamps(C.c_void_p(stamps.data_ptr()))
    up.minibatch_step(data, seed=1, offset=5 << 32)
    torch.cuda.synchronize()
    if it >= 10:
        acc += stamps.cpu().double()
lib.mbpo_debug_set_ppo_stamps(C.c_void_p(0))
acc /= 20
names = ["tile -> LDS", "barrier", "thin 0", "hidden 1", "hidden 2", "out + loss (+ noise)", "barrier", "thin dgrad", "dgrad L2", "dgrad L1",
         "weight gradients"]
# slot 0 = kernel top; tile t stamps slots 12 t + 1 .. 12 t + 11 (the end of each section); a tile starts where the previous one ended
for t in range(5):
    b = 12 * t
    start = float(acc[0] if t == 0 else acc[b - 1])
    row, prev = [], start
    for i in range(1, 12):
        row.append(float(acc[b + i]) - prev)
        prev = float(acc[b + i])
    print("tile %d: " % t + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, row)) + "   | total %.0f" % (prev - start))

"""Back-to-back time of the SAC fwd/bwd kernel and of a whole sgd_step for a given hidden width (HIP events)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip
dev = torch.device('cuda:0')
X, U, B = 4, 1, 256
hid = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,64,64").split(","))
up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, *hid, 2 * U], q_dims=[X + U, *hid, 1], batch_size=B, device=dev, seed=1)
g = torch.Generator().manual_seed(0)
up.load_state((torch.randn(up.params.numel(), generator=g) * 0.1).to(dev))
batch = torch.randn(B, 2 * X + U + 3, generator=g).to(dev)
lib = _hip.load()
d = up.desc
d.batch = batch.data_ptr()
st = torch.cuda.current_stream()
for name, fn in (("fwd_bwd kernel alone", lambda: _hip.check(lib.mbpo_sac_grads_phase(C.byref(d), 1, st.cuda_stream), "p")),
                 ("whole sgd_step (3 launches, eager)", lambda: up.sgd_step(batch))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"hidden {hid}: {name}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us")

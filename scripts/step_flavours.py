"""Device time per SAC update of each step flavour on ONE rank (no exchange), hipGraph of 64 chained updates: the local cost under
DESIGN section 6's latency-budget table.
  two-launch   mbpo_sac_step                                   (fwd/bwd, reduce + speculative apply)
  three-launch mbpo_sac_grads + mbpo_sac_apply                 (fwd/bwd, reduce, apply)              — the peer-fused form's local part
  four-launch  mbpo_sac_grads + [all-reduce] + mbpo_sac_grad_norms + mbpo_sac_apply — the split / library-collective forms' local part
"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops

dev = torch.device('cuda:0')
X, U, B, G = 4, 1, 256, 64
hid = (64, 64, 64)
g = torch.Generator().manual_seed(0)
D = 2 * X + U + 3
batches = torch.randn(G, B, D, generator=g).to(dev)
nm, ns = (torch.randn(X, generator=g) * 0.3).to(dev), (torch.rand(X, generator=g) + 0.5).to(dev)


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / G * 1e3


for name, kw in (("two-launch", dict(two_launch=True)), ("three-launch", dict(two_launch=False)),
                 ("four-launch (identity all-reduce)", dict(two_launch=False, all_reduce=lambda t: None, world_size=1))):
    up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, *hid, 2 * U], q_dims=[X + U, *hid, 1], batch_size=B, device=dev, seed=1, **kw)
    gg = torch.Generator().manual_seed(0)
    up.load_state((torch.randn(up.params.numel(), generator=gg) * 0.1).to(dev))
    rng = ops.make_rng(dev, 5)

    def scan():
        for i in range(G):
            up.sgd_step(batches[i], nm, ns, seed=0, offset=(16 + i) << 32, rng_dev=rng, defer_clip_check=True)
        up.finalize()

    ts = [timed(scan) for _ in range(3)]
    print("%-36s %s us per update" % (name, " ".join("%.2f" % t for t in ts)), flush=True)

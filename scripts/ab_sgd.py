"""Same-box A/B of library variants on the SAC update: graph of 64 chained sgd_steps (deferred clip check + finalize), device time."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
    from mbpo import ops
    dev = torch.device('cuda:0')
    X, U, B, G = 4, 1, 256, 64
    hid = (64, 64, 64)
    up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, *hid, 2 * U], q_dims=[X + U, *hid, 1], batch_size=B, device=dev, seed=1,
                        max_grad_norm=float(os.environ.get("MBPO_AB_MAXNORM", "1e5")))
    g = torch.Generator().manual_seed(0)
    up.load_state((torch.randn(up.params.numel(), generator=g) * 0.1).to(dev))
    batches = torch.randn(G, B, 2 * X + U + 3, generator=g).to(dev)
    rng = ops.make_rng(dev, 5)
    if os.environ.get("MBPO_AB_NO_ACCUM"):
        up.desc.metrics_accum = None      # timing experiment: no running metric sums
    def scan():
        for i in range(G):
            up.sgd_step(batches[i], seed=0, offset=(16 + i) << 32, rng_dev=rng, defer_clip_check=True)
        up.finalize()
    scan(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        scan()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"us_per_update": e0.elapsed_time(e1) / 50 / G * 1e3,
                      "ptrs": [hex(t.data_ptr() & 0xffffffffff) for t in (up.params, up.workspace, batches, up.grads, up.adam_m)]}))
else:
    libs = sys.argv[1:]
    for r in range(int(os.environ.get("ROUNDS", "3"))):
        for lib in libs:
            env = dict(os.environ, MBPO_HIP_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print(lib, line[-1] if line else out.stderr[-300:], flush=True)

"""Development loop of the specialised SAC kernel (csrc/sac_lean.hip) on one GPU box, one process:
  1. slabs / gradients / parameters bit-identical to the generic kernel (quick form of tests/test_gpu_sac_lean.py);
  2. same-process A/B: fwd/bwd launch alone back to back, and the two-launch update inside a graph of 64 chained updates;
  3. the kernel's own phase timeline from s_memtime stamps (tile 0, critic and actor role).
Usage: python scripts/lean_dev.py [--no-stamps]"""
import ctypes as C, json, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip

dev = torch.device('cuda:0')
lib = _hip.load()
lib.mbpo_debug_set_sac_lean.argtypes = [C.c_int]
X, U, B, G = 4, 1, 256, 64
hid = (64, 64, 64)
g = torch.Generator().manual_seed(0)
D = 2 * X + U + 3


def updater():
    up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, *hid, 2 * U], q_dims=[X + U, *hid, 1], batch_size=B, device=dev, seed=1)
    gg = torch.Generator().manual_seed(0)
    up.load_state((torch.randn(up.params.numel(), generator=gg) * 0.1).to(dev))
    return up


batches = torch.randn(G, B, D, generator=g).to(dev)
nm, ns = (torch.randn(X, generator=g) * 0.3).to(dev), (torch.rand(X, generator=g) + 0.5).to(dev)

# ---- 1. bit identity
res = {}
for lean in (0, 1):
    lib.mbpo_debug_set_sac_lean(lean)
    up = updater()
    rng = ops.make_rng(dev, 5)
    for i in range(6):
        up.sgd_step(batches[i], nm, ns, seed=0, offset=(16 + i) << 32, rng_dev=rng, defer_clip_check=True)
    up.finalize()
    torch.cuda.synchronize()
    res[lean] = {k: getattr(up, k).clone() for k in ("workspace", "grads", "params", "target_q", "adam_m", "adam_v", "metrics")}
n_slab = (B // 16) * (up.P + 2 * up.Q + 4)
ident = {k: bool(torch.equal(res[0][k] if k != "workspace" else res[0][k][:n_slab], res[1][k] if k != "workspace" else res[1][k][:n_slab]))
         for k in res[0]}
print("bit-identical to the generic kernel:", ident, flush=True)
if not all(ident.values()):
    for k in ("workspace", "grads"):
        a, b = res[0][k][:n_slab] if k == "workspace" else res[0][k], res[1][k][:n_slab] if k == "workspace" else res[1][k]
        bad = (a != b).nonzero().flatten()
        print(k, "mismatches", bad.numel(), "first", bad[:12].tolist(), "max abs diff", float((a - b).abs().max()),
              "nan", int(torch.isnan(b).sum()))
        if k == "workspace" and bad.numel():
            P, Q = up.P, up.Q
            nt = B // 16
            for idx in bad[:6].tolist():
                if idx < nt * P:
                    print("   policy slab tile", idx // P, "elem", idx % P, float(a[idx]), float(b[idx]))
                elif idx < nt * (P + 2 * Q):
                    j = idx - nt * P
                    print("   critic slab tile", j // (2 * Q), "critic", (j % (2 * Q)) // Q, "elem", j % Q, float(a[idx]), float(b[idx]))
                else:
                    print("   loss partial", idx - nt * (P + 2 * Q), float(a[idx]), float(b[idx]))


# ---- 2. timing
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


out = {}
for rnd in range(3):
    for lean in (0, 1):
        lib.mbpo_debug_set_sac_lean(lean)
        up = updater()
        rng = ops.make_rng(dev, 5)
        d = up.desc

        def scan():
            for i in range(G):
                up.sgd_step(batches[i], nm, ns, seed=0, offset=(16 + i) << 32, rng_dev=rng, defer_clip_check=True)
            up.finalize()

        def alone():
            d.batch = batches[0].data_ptr()
            d.norm_mean, d.norm_std = nm.data_ptr(), ns.data_ptr()
            d.rng_dev = rng.data_ptr()
            for i in range(G):
                _hip.check(lib.mbpo_sac_grads_phase(C.byref(d), 1, torch.cuda.current_stream().cuda_stream), "fwd_bwd")

        out.setdefault(("pair", lean), []).append(timed(scan))
        out.setdefault(("alone", lean), []).append(timed(alone))
for k, v in out.items():
    print("%-6s %-8s us: %s" % (k[0], "lean" if k[1] else "generic", " ".join("%.2f" % t for t in v)), flush=True)

# ---- 3. stamps of the lean kernel
if "--no-stamps" not in sys.argv:
    lib.mbpo_debug_set_sac_lean(1)
    up = updater()
    stamps = torch.zeros(96, dtype=torch.int64, device=dev)
    names = ["top -> tile in LDS", "thin 0 + requests F1", "hidden 1", "hidden 2", "out + sample", "thin (F1) + requests B",
             "hidden 1 (F1)", "hidden 2 (F1)", "out (F1)", "loss section", "thin dgrad / wgrad-out", "L2 dgrad || wgrad",
             "L1 dgrad || wgrad", "wgrad first"]
    acc = None
    for it in range(30):
        if it == 10:
            lib.mbpo_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
        up.sgd_step(batches[it % G], nm, ns, offset=it << 32)
        torch.cuda.synchronize()
        if it >= 10:
            s = stamps.cpu()[:32].reshape(2, 16).clone()
            acc = s if acc is None else acc + s
            f = stamps.cpu()[32:64].clone()
            fine = f if it == 10 else fine + f
    lib.mbpo_debug_set_stamps(C.c_void_p(0))
    acc = acc.double() / 20
    fine = fine.double() / 20
    print("critic L2 step, wave 0 (dgrad): reads issued %d  MFMAs %d  epilogue+write %d  to barrier %d  barrier %d" %
          tuple(float(fine[i + 1] - fine[i]) for i in (0, 1, 2, 3, 4)))
    print("critic L2 step, wave 4 (wgrad): reads issued %d  MFMAs %d  stores %d  to barrier %d  barrier %d" %
          tuple(float(fine[8 + i + 1] - fine[8 + i]) for i in (0, 1, 2, 3, 4)))
    print("   wave 4 start minus wave 0 start: %d" % float(fine[8] - fine[0]))
    b12 = float(fine[5])
    print("critic L1 step (cycles after barrier 12): aux copy starts %d, ends %d | wave 0 (dgrad) at barrier 13 %d, wave 4 (wgrad) %d, released %d | aux final copy %d..%d" %
          tuple(float(fine[i]) - b12 for i in (16, 17, 20, 21, 22, 18, 19)))
    for role, nm_ in ((0, "critic 0"), (1, "actor")):
        t = acc[role, :15] - acc[role, 0]
        print("role", nm_, "total cycles %.0f" % float(t[14]))
        for i in range(1, 15):
            print("    %-28s %8.0f" % (names[i - 1], float(t[i] - t[i - 1])))

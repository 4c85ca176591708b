"""Phase timeline of the generic k_sac_fwd_bwd (tile 0, both roles) from in-kernel s_memtime stamps; bench-shaped SAC step.
The 64x3 benchmark shape runs k_sac_lean by default (its timeline: scripts/lean_dev.py); this script pins the generic kernel."""
import ctypes as C, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops, _hip
dev = torch.device('cuda:0')
X, U, B = 4, 1, 256
hid = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,64,64").split(","))
up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, *hid, 2 * U], q_dims=[X + U, *hid, 1], batch_size=B, device=dev, seed=1)
g = torch.Generator().manual_seed(0)
up.load_state((torch.randn(up.params.numel(), generator=g) * 0.1).to(dev))
D = 2 * X + U + 3
batch = torch.randn(B, D, generator=g).to(dev)
lib = _hip.load()
lib.mbpo_debug_set_sac_lean.argtypes = [C.c_int]
lib.mbpo_debug_set_sac_lean(0)
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
names = {0: ["start", "setup", "E load tile", "F0 pi(s')||Q1||Q2 fwd", "E sample a'", "F1 Qtgt fwd", "E targets", "B2 Q dgrad||wgrad", "E partials"],
         1: ["start", "setup", "E load tile", "F0 pi(s) fwd", "E sample a", "F1 Q1||Q2 fwd", "E dL/dq", "B2 Q input-grad", "E dL/dlogits",
             "B3 pi dgrad||wgrad", "E partials"]}
import os
if X and U == 1 and os.environ.get("MBPO_SAC_JVP", "1") != "0" and hid[0] in (64, 128):
    # forward-mode actor role (the default at u = 1): no critic input-gradient phase
    names[1] = ["start", "setup", "E load tile", "F0 pi(s) fwd", "E sample a", "F1 Q1||Q2 fwd + tangent", "E dL/dq, dL/dlogits",
                "B2 pi dgrad||wgrad", "E partials"]
for mode in ("warm (same params re-read)", "cold (params rewritten by apply)"):
    acc = None
    for it in range(20):
        if it == 10:
            lib.mbpo_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
        up.sgd_step(batch) if mode.startswith("cold") else up.grads_only(batch) if hasattr(up, "grads_only") else up.sgd_step(batch)
        torch.cuda.synchronize()
        if it >= 10:
            s = stamps.cpu()[:32].reshape(2, 16).clone(); print('  section after phase 1: chain set-up + first-layer request %d / %d cycles (critic / actor role), rest of the section %d / %d' % (int(s[0,14]-s[0,5]), int(s[1,14]-s[1,5]), int(s[0,6]-s[0,14]), int(s[1,6]-s[1,14]))); fine = stamps.cpu()[40:47].clone(); wg = stamps.cpu()[48:52].clone(); print('  wgrad hidden step (critic B2, chain 2 wave 0): wgrad %d  bgrad %d  barrier %d' % tuple(int(wg[i+1]-wg[i]) for i in range(3))); print('  fwd hidden step (actor F0, wave 0): request %d  compute %d  barrier %d | request %d  compute %d  barrier %d' % tuple(int(fine[i+1]-fine[i]) for i in range(6)))
            acc = s if acc is None else acc + s
    lib.mbpo_debug_set_stamps(C.c_void_p(0))
    acc = acc.double() / 10
    print(mode)
    for role in (0, 1):
        n = len(names[role])
        t = acc[role, :n] - acc[role, 0]
        print("  role", role, " total cycles", round(float(t[-1])))
        for i in range(1, n):
            print(f"    {names[role][i]:28s} {float(t[i] - t[i-1]):9.0f} cyc")
    break

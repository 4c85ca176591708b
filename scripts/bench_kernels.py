"""Per-kernel rooflines for every row of SURVEY §8a (the secondary measurements of SURVEY §8d).

    python scripts/bench_kernels.py [--out profiles/r02_kernel_rooflines.json]

Each entry: the C-ABI entry point, the configuration, average launch time from HIP events on the launch stream, the
ALGORITHMIC work per unit (SURVEY §8d's figures, restated next to each case), achieved GB/s or TFLOP/s and the fraction of the
bounding roof (HBM 8 TB/s, fp32 MFMA 157.3 TFLOP/s; MI355X_MICROARCH.md).  Small cases are the BASELINE.json configurations
(launch-latency-bound: their time is the ~4.5 us kernel floor); the "large" case of every HBM-bound kernel is sized so the
launch floor is negligible and shows what the kernel reaches against the roof.
Inputs are synthetic (seeded); nothing under oracle/ or /root/reference is touched.
"""
import argparse
import json
import math
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0
LAST_EAGER = None     # eager (Python + launch API) time per call of the case measured last
MFMA_F32_PEAK_TF = 157.3


def mlp_macs(dims):
    return sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def lecun_flat(dims, g, n_nets=1):
    parts = []
    for _ in range(n_nets):
        for i, o in zip(dims[:-1], dims[1:]):
            lim = math.sqrt(3.0 / i)
            parts += [((torch.rand(i, o, generator=g) * 2 - 1) * lim).reshape(-1), torch.zeros(o)]
    return torch.cat(parts)


def timed(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def timed_graph(fn, inner=20, reps=10):
    """Device-side time per call: `inner` calls captured into one hipGraph and replayed (no Python / launch-API time between
    kernels).  Returns None when the op cannot be captured."""
    try:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(inner):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            graph.replay()
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / (reps * inner)
    except Exception as e:      # noqa: BLE001
        print(f"  (graph capture failed: {type(e).__name__}: {str(e)[:120]})", file=sys.stderr)
        torch.cuda.synchronize()
        return None


def both(fn, reps, warm=5, inner=20):
    """(device time per call under graph replay if capturable else eager, eager time per call)"""
    global LAST_EAGER
    te = timed(fn, reps, warm)
    LAST_EAGER = te
    tg = timed_graph(fn, inner=inner, reps=max(2, reps // inner))
    return (tg if tg is not None else te), te


def hbm_entry(name, entry, cfg, t, nbytes, unit_note):
    gbs = nbytes / t / 1e9
    return {"kernel": name, "entry": entry, "config": cfg, "device_us": t * 1e6, "eager_us": LAST_EAGER * 1e6, "bound": "hbm", "algorithmic_bytes": nbytes,
            "per_unit": unit_note, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}


def mfma_entry(name, entry, cfg, t, flop, unit_note, extra=None):
    tf = flop / t / 1e12
    e = {"kernel": name, "entry": entry, "config": cfg, "device_us": t * 1e6, "eager_us": LAST_EAGER * 1e6, "bound": "mfma", "algorithmic_flop": flop,
         "per_unit": unit_note, "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF}
    if extra:
        e.update(extra)
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="run only the cases whose function name contains this (rollout, ens_fwd, sac, ppo, bptt, icem, ...)")
    args = ap.parse_args()
    want = lambda name: args.only is None or args.only in name
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the HIP path has no CPU fallback")
    from mbpo import _hip, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    out = []

    def log(s):
        print(s, file=sys.stderr, flush=True)

    # ---------------------------------------------------------------- P4: GAE scan (24 B / element)
    for (B, T, reps) in [(16384, 5, 200), (16384, 40, 200), (1 << 20, 40, 20)]:
        f = lambda: torch.rand(B, T, generator=g).to(dev)
        trunc, term = (f() < 0.05).float(), (f() < 0.02).float()
        rew, val, boot = f(), f(), torch.rand(B, generator=g).to(dev)
        t, te = both(lambda: ops.gae_scan(trunc, term, rew, val, boot, 0.99, 0.95), reps)
        out.append(hbm_entry("k_gae_scan", "mbpo_gae_scan", {"B": B, "T": T}, t, 24 * B * T, "24 B per element (4 reads + 2 writes)"))
        out[-1]["elements_per_s"] = B * T / t
        log(f"gae {B}x{T}: {t * 1e6:.1f} us (eager {te * 1e6:.1f})")
        del trunc, term, rew, val, boot
    # ---------------------------------------------------------------- B2: lambda-return scan (12 B / element)
    for (B, T, reps) in [(4096, 32, 200), (1 << 20, 32, 20)]:
        rew, nv = torch.rand(B, T, generator=g).to(dev), torch.rand(B, T, generator=g).to(dev)
        t, te = both(lambda: ops.lambda_return_scan(rew, nv, 0.99, 0.97), reps)
        out.append(hbm_entry("k_lambda_return_scan", "mbpo_lambda_return_scan", {"B": B, "T": T}, t, 12 * B * T,
                             "12 B per element (2 reads + 1 write)"))
        log(f"lambda {B}x{T}: {t * 1e6:.1f} us")
        del rew, nv
    # ---------------------------------------------------------------- R6/S1: replay insert / sample (48 B rows)
    D = 12
    for (mx, n, reps) in [(1 << 20, 20480, 200), (1 << 22, 1 << 20, 20)]:
        data = torch.zeros(mx, D, device=dev)
        state = torch.zeros(4, dtype=torch.int32, device=dev)
        rows = torch.rand(n, D, generator=g).to(dev)
        t, te = both(lambda: ops.replay_insert(data, state, rows), reps)
        out.append(hbm_entry("k_replay_insert", "mbpo_replay_insert", {"max_replay": mx, "rows": n, "D": D}, t, 2 * 48 * n,
                             "48 B read + 48 B write per row (SURVEY counts the 48 B write only)"))
        log(f"insert {n}: {t * 1e6:.1f} us")
        for (ns, r2) in ([(256, 200)] if n == 20480 else [(1 << 20, 20)]):
            buf = torch.empty(ns, D, device=dev)
            t, te = both(lambda: ops.replay_sample(data, state, ns, 1, 0, out=buf), r2)
            out.append(hbm_entry("k_replay_sample", "mbpo_replay_sample", {"max_replay": mx, "rows": ns, "D": D}, t, 2 * 48 * ns,
                                 "48 B read + 48 B write per sampled row (random rows: 64 B sectors)"))
            log(f"sample {ns}: {t * 1e6:.1f} us")
        del data, rows
    # ---------------------------------------------------------------- R8: running statistics (2 passes over x columns)
    for (n, reps) in [(20480, 200), (1 << 22, 10)]:
        X = 4
        rows = torch.rand(n, D, generator=g).to(dev)
        stats = torch.cat([torch.zeros(1 + 2 * X), torch.ones(X)]).to(dev)
        sums = torch.zeros(1 + 2 * X, device=dev)
        ws = torch.empty(ops.stats_workspace_floats(X), device=dev)
        t, te = both(lambda: ops.running_stats_update(rows, 0, X, stats, sums=sums, workspace=ws), reps)
        out.append(hbm_entry("k_running_stats (2 reduce passes + apply)", "mbpo_running_stats_reduce x2 + _apply",
                             {"rows": n, "x": X, "row_len": D}, t, 2 * 4 * X * n,
                             "2 passes x 4x B per row (the obs columns; rows are 48 B apart, so sectors fetch 3x that)"))
        log(f"stats {n}: {t * 1e6:.1f} us (3 launches)")
        del rows
    # ---------------------------------------------------------------- S8/B4: AdamW (+ Polyak)  28 / 36 B per parameter
    for (n, reps) in [(26309, 200), (1 << 24, 20)]:
        opt = ops.AdamW(n, dev, lr=3e-4, weight_decay=0.0)
        p, gr, tgt = torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev), torch.zeros(n, device=dev)
        t, te = both(lambda: opt.step(p, gr, target=tgt, tau=0.005), reps)
        out.append(hbm_entry("k_adamw_step (+Polyak)", "mbpo_adamw_step", {"params": n}, t, 36 * n,
                             "36 B per parameter (read g,p,m,v,target; write p,m,v,target)"))
        log(f"adamw {n}: {t * 1e6:.1f} us")
        del p, gr, tgt, opt

    # ---------------------------------------------------------------- R1-R7: fused model rollout
    def rollout_case(N, X, U, E, S, hid_pi, hid_dyn, reps):
        pd, dd = [X, *hid_pi, 2 * U], [X + U, *hid_dyn, 2 * X]
        pp = lecun_flat(pd, g).to(dev)
        dp = lecun_flat(dd, g, E).to(dev)
        pd_k, dd_k = pd, dd
        if max(hid_pi) != max(hid_dyn):
            # the fused kernel walks policy and members at ONE hidden width: the host zero-pads the narrower net (as the trainers
            # do, INTEGRATION.md "Network shapes"); FLOP are counted on the logical shapes
            w = max(max(hid_pi), max(hid_dyn))
            pp, dp = ops.embed_mlp_params(pp, pd, w), ops.embed_mlp_params(dp, dd, w, E)
            pd_k, dd_k = ops.padded_dims(pd, w), ops.padded_dims(dd, w)
        obs = torch.randn(N, X, generator=g).to(dev)
        first = obs.clone()
        steps, done = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        rp = torch.cat([torch.zeros(X), torch.ones(X), torch.ones(U) * 0.1]).to(dev)
        rows = torch.empty(S * N, 2 * X + U + 3, device=dev)

        def run():
            ops.model_rollout(policy_params=pp, policy_spec=ops.MlpSpec(pd_k), x_dim=X, u_dim=U, obs=obs, first_obs=first, steps=steps,
                              done=done, n_steps=S, episode_length=S, system_kind=_hip.SYS_ENSEMBLE, dyn_params=dp,
                              dyn_spec=ops.MlpSpec(dd_k, "swish", E), reward_kind=_hip.REWARD_QUADRATIC, reward_params=rp, seed=1,
                              offset=0, out=rows)
        t, te = both(run, reps)
        flop = N * S * (2 * E * mlp_macs(dd) + 2 * mlp_macs(pd))
        e = mfma_entry("k_model_rollout", "mbpo_model_rollout",
                       {"N": N, "x": X, "u": U, "E": E, "S": S, "policy": list(hid_pi), "member": list(hid_dyn)}, t, flop,
                       "2*E*M + 2*P FLOP per transition", {"transitions_per_s": N * S / t})
        out.append(e)
        log(f"rollout N={N} x={X} E={E} {hid_dyn}: {t * 1e6:.1f} us  {N * S / t / 1e6:.1f} M transitions/s")

    def attempt(fn, *a):
        try:
            fn(*a)
        except Exception as e:      # noqa: BLE001 — a configuration the kernels reject is reported, not hidden
            out.append({"kernel": fn.__name__, "config": [list(x) if isinstance(x, tuple) else x for x in a], "error": str(e)[:300]})
            log(f"{fn.__name__}{a}: {e}")

    if want("rollout_case"): rollout_case(4096, 4, 1, 5, 5, (64, 64, 64), (64, 64, 64), 50)       # C2
    if want("rollout_case"): rollout_case(32768, 4, 1, 5, 5, (64, 64, 64), (64, 64, 64), 20)      # C4's global env count on one GPU
    if want("rollout_case"): attempt(rollout_case, 4096, 4, 1, 5, 5, (64, 64, 64), (256, 256), 20)   # "also report 256x2" members
    if want("rollout_case"): attempt(rollout_case, 4096, 4, 1, 5, 5, (256, 256), (256, 256), 20)
    if want("rollout_case"): attempt(rollout_case, 4096, 17, 6, 10, 5, (64, 64, 64), (64, 64, 64), 20)     # C5 shape through the rollout kernel

    # ---------------------------------------------------------------- R2: vmapped ensemble forward (System.step outside the fused rollout)
    def ens_fwd_case(N, X, U, E, hid, reps):
        dd = [X + U, *hid, 2 * X]
        spec = ops.MlpSpec(dd, "swish", E)
        dp = lecun_flat(dd, g, E).to(dev)
        xu = torch.randn(N, X + U, generator=g).to(dev)
        t, te = both(lambda: ops.ensemble_mlp_forward(dp, spec, xu), reps)
        out.append(mfma_entry("k_ensemble_forward", "mbpo_ensemble_mlp_forward", {"N": N, "x": X, "u": U, "E": E, "member": list(hid)}, t,
                              N * 2 * E * mlp_macs(dd), "2*E*M FLOP per row (shared input)", {"rows_per_s": N / t}))
        log(f"ensemble forward N={N} E={E} {hid}: {t * 1e6:.1f} us")

    if want("ens_fwd_case"): attempt(ens_fwd_case, 4096, 4, 1, 5, (64, 64, 64), 100)
    if want("ens_fwd_case"): attempt(ens_fwd_case, 32768, 4, 1, 5, (64, 64, 64), 50)

    # ---------------------------------------------------------------- S3-S8: SAC sgd_step
    def sac_case(X, U, hidden, B, reps):
        pd, qd = [X, *hidden, 2 * U], [X + U, *hidden, 1]
        up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=pd, q_dims=qd, batch_size=B, device=dev)
        params = torch.cat([lecun_flat(pd, g), lecun_flat(qd, g, 2), torch.zeros(1)]).to(dev)
        up.load_state(params)
        batch = torch.randn(B, 2 * X + U + 3, generator=g)
        batch[:, X + U + 1] = 1.0
        batch[:, -1] = 0.0
        batch = batch.to(dev)
        # as the trainer issues it: each step's clip check is resolved by the next step's first launch (a scan of sgd_steps ends
        # in ONE mbpo_sac_finalize, not timed here)
        t, te = both(lambda: up.sgd_step(batch, defer_clip_check=True), reps)
        up.finalize()
        flop = B * 2 * (5 * mlp_macs(pd) + 12 * mlp_macs(qd))
        layered = not (len(set(hidden)) == 1 and hidden[0] in (64, 128))
        out.append(mfma_entry("k_layered_gemm x ~46 + heads + k_sac_reduce + k_sac_apply (layered path)" if layered
                              else "k_sac_fwd_bwd + k_sac_reduce_apply", "mbpo_sac_step",
                              {"x": X, "u": U, "hidden": list(hidden), "B": B}, t, flop, "2*(5P + 12Q) FLOP per sample",
                              {"updates_per_s": 1.0 / t}))
        log(f"sac sgd_step x={X} {hidden} B={B}: {t * 1e6:.1f} us")

    if want("sac_case"): sac_case(4, 1, (64, 64, 64), 256, 200)
    if want("sac_case"): sac_case(3, 1, (128, 128, 128), 256, 200)      # the reference tests' width (tests/test_sac.py)
    if want("sac_case"): attempt(sac_case, 4, 1, (64, 64, 64), 2048, 100)        # C4's global batch on one GPU
    if want("sac_case"): attempt(sac_case, 17, 6, (64, 64, 64), 256, 200)
    if want("sac_case"): attempt(sac_case, 4, 1, (256, 256, 256), 256, 50)        # wider than the fused kernels take: one GEMM launch per Dense layer
    if want("sac_case"): attempt(sac_case, 4, 1, (256, 256, 256), 4096, 20)

    # ---------------------------------------------------------------- P3-P6: PPO minibatch_step (C3)
    def ppo_case(X, U, hidden, B, T, reps):
        pd, vd = [X, *hidden, 2 * U], [X, *hidden, 1]
        up = ops.PpoUpdater(x_dim=X, u_dim=U, policy_dims=pd, value_dims=vd, batch_size=B, unroll_length=T, device=dev)
        up.load_state(torch.cat([lecun_flat(pd, g), lecun_flat(vd, g)]).to(dev))
        Dp = ops.transition_row_len(X, U, True)
        data = torch.randn(B, T, Dp, generator=g) * 0.5
        data[..., X + U + 1] = 1.0                       # discount
        data[..., -1] = 0.0                              # truncation
        data = data.to(dev)
        t, te = both(lambda: up.minibatch_step(data), reps)
        flop = B * T * 2 * 3 * (mlp_macs(pd) + mlp_macs(vd)) + B * 2 * mlp_macs(vd)
        out.append(mfma_entry("k_ppo_values + k_ppo_fwd_bwd + reduce/apply", "mbpo_ppo_grads + mbpo_ppo_apply",
                              {"x": X, "u": U, "hidden": list(hidden), "B": B, "T": T}, t, flop,
                              "3*2*(P+V) FLOP per (sample, step) + bootstrap value; includes the in-kernel GAE",
                              {"gae_elements_per_s": B * T / t}))
        log(f"ppo minibatch_step B={B} T={T}: {t * 1e6:.1f} us")

    if want("ppo_case"): ppo_case(4, 1, (64, 64, 64), 512, 5, 100)
    if want("ppo_case"): attempt(ppo_case, 4, 1, (64, 64, 64), 512, 40, 50)
    if want("ppo_case"): attempt(ppo_case, 3, 1, (64, 64), 128, 40, 100)        # the reference's own PPO test shape (tests/test_ppo.py:30-56)

    # ---------------------------------------------------------------- B1-B5: BPTT actor gradient (C5)
    def bptt_case(X, U, E, H, n, reps):
        hidden = (64, 64, 64)
        ad, cd, dd = [X, *hidden, 2 * U], [X, *hidden, 1], [X + U, *hidden, 2 * X]
        op = ops.BpttActorGrad(x_dim=X, u_dim=U, horizon=H, actor_dims=ad, critic_dims=cd, n=n, device=dev, seed=3)
        apar, cpar = lecun_flat(ad, g).to(dev), lecun_flat(cd, g, 2).to(dev)
        dpar = (lecun_flat(dd, g, E) * 0.5).to(dev)
        x0 = torch.randn(n, X, generator=g).to(dev)
        kw = dict(actor_params=apar, target_critic_params=cpar, init_states=x0, state_mean=torch.zeros(X, device=dev),
                  state_std=torch.ones(X, device=dev), reward_mean_std=torch.tensor([0.0, 1.0], device=dev),
                  system_kind=_hip.SYS_ENSEMBLE, reward_kind=_hip.REWARD_QUADRATIC,
                  reward_params=torch.cat([torch.zeros(X), torch.ones(X), torch.ones(U) * 0.1]).to(dev), dyn_params=dpar,
                  dyn_spec=ops.MlpSpec(dd, "swish", E))
        t, te = both(lambda: op(**kw), reps, warm=2, inner=2)
        P_, M_, V_ = mlp_macs(ad), mlp_macs(dd), mlp_macs(cd)
        flop = n * H * (3 * 2 * P_ + 2 * 2 * E * M_ + 2 * 2 * 2 * V_)
        out.append(mfma_entry("k_bptt_actor + reduce", "mbpo_bptt_actor_grads", {"x": X, "u": U, "E": E, "H": H, "n": n}, t, flop,
                              "3*(2P) + 2*(2*E*M) + 2*(2*2V) FLOP per (state, step), fwd+bwd",
                              {"state_steps_per_s": n * H / t, "grads_finite": bool(torch.isfinite(op.grads).all())}))
        log(f"bptt x={X} u={U} E={E} H={H} n={n}: {t * 1e3:.2f} ms  {n * H / t / 1e6:.2f} M state-steps/s")

    if want("bptt_case"): attempt(bptt_case, 17, 6, 10, 32, 4096, 5)          # C5, one GPU's share
    if want("bptt_case"): bptt_case(4, 1, 5, 5, 4096, 20)            # north-star shape

    # ---------------------------------------------------------------- N3: ensemble NLL fwd+bwd
    X, U, E, Bn = 4, 1, 5, 256
    dd = [X + U, 64, 64, 64, 2 * X]
    spec = ops.MlpSpec(dd, "swish", E)
    nll = ops.EnsembleNllGrad(x_dim=X, u_dim=U, spec=spec, batch=Bn, device=dev)
    dpar = lecun_flat(dd, g, E).to(dev)
    rows = torch.randn(8192, 2 * X + U + 2, generator=g).to(dev)
    idx = torch.randint(0, 8192, (E, Bn), generator=g, dtype=torch.int32).to(dev)
    t, te = both(lambda: nll(dpar, rows, idx), 100)
    out.append(mfma_entry("k_ens_nll_fwd_bwd + k_ens_reduce", "mbpo_ens_nll_grads", {"x": X, "u": U, "E": E, "B": Bn}, t,
                          E * Bn * 3 * 2 * mlp_macs(dd), "3*(2M) FLOP per (member, sample)"))
    log(f"ensemble nll: {t * 1e6:.1f} us")

    # ---------------------------------------------------------------- N4: iCEM planner at the reference's defaults (icem_optimizer.py:25-50)
    def icem_case(H, reps):
        from mbpo.optimizers.trajectory_optimizers.icem_optimizer import iCemParams, iCemTO
        from mbpo.systems import PendulumSystem
        from mbpo.utils import keys as K
        p = iCemParams()
        system = PendulumSystem()
        X, U = system.x_dim, system.u_dim
        opt = iCemTO(horizon=H, action_dim=U, opt_params=p, system=system)
        st0 = opt.init(K.PRNGKey(0))
        b = opt._buffers(dev)
        spec = system.rollout_spec(st0.system_params, dev)
        lib, stp = _hip.load(), _hip.current_stream_ptr
        b["std"].fill_(p.init_std)
        x0 = torch.tensor([-1.0, 0.0, 0.0], device=dev)
        NC, N, D = b["NC"], b["N"], b["rows"].shape[1]

        def sample():
            _hip.check(lib.mbpo_icem_sample(b["mean"].data_ptr(), b["std"].data_ptr(), b["prev"].data_ptr(), b["u_min"].data_ptr(),
                                            b["u_max"].data_ptr(), p.num_samples, opt.num_prev, H, U, p.num_particles, float(p.exponent),
                                            11, 0, None, b["actions"].data_ptr(), b["cand"].data_ptr(), stp()), "mbpo_icem_sample")

        def rollout():
            b["obs"].copy_(x0.expand(N, X)); b["first"].copy_(b["obs"]); b["steps"].zero_(); b["done"].zero_()
            ops.model_rollout(x_dim=X, u_dim=U, actions=b["actions"], obs=b["obs"], first_obs=b["first"], steps=b["steps"], done=b["done"],
                              n_steps=H, episode_length=2 ** 30, seed=5, offset=0, out=b["rows"], **spec)

        def update():
            b["best_value"].fill_(float("-inf"))
            _hip.check(lib.mbpo_icem_update_constrained(
                b["rows"].data_ptr(), D, X + U, NC, p.num_particles, H, U, b["cand"].data_ptr(), p.num_elites, opt.num_prev, float(p.alpha),
                0, None, float(p.lambda_constraint), 0, b["mean"].data_ptr(), b["std"].data_ptr(), b["best_value"].data_ptr(),
                b["best_seq"].data_ptr(), b["prev"].data_ptr(), b["values"].data_ptr(), b["rank"].data_ptr(), stp()), "mbpo_icem_update")

        cfg = {"num_samples": p.num_samples, "num_particles": p.num_particles, "num_elites": p.num_elites, "horizon": H, "x": X, "u": U,
               "system": "Pendulum"}
        sample(); rollout(); update()
        t, te = both(sample, reps)
        out.append(hbm_entry("k_icem_sample", "mbpo_icem_sample", cfg, t, 4 * (NC * H * U + H * N * U),
                             "4 B per candidate element written once + once per particle (actions [H][N][u])"))
        log(f"icem sample: {t * 1e6:.1f} us")
        t, te = both(rollout, reps)
        out.append(hbm_entry("k_model_rollout (open loop, Pendulum)", "mbpo_model_rollout", cfg, t, 4 * H * N * (U + D),
                             "4 B per action read + one transition row written per (step, trajectory)",))
        out[-1]["transitions_per_s"] = H * N / t
        log(f"icem open-loop rollout: {t * 1e6:.1f} us  {H * N / t / 1e6:.1f} M transitions/s")
        t, te = both(update, reps)
        out.append(hbm_entry("k_icem_update", "mbpo_icem_update_constrained", cfg, t, 4 * H * N * 1 + 4 * NC * H * U,
                             "4 B per reward element of the rows + the candidates read once"))
        log(f"icem update: {t * 1e6:.1f} us")
        ost = st0
        t, te = both(lambda: opt.optimize(x0, ost), max(reps // 10, 3), warm=1, inner=1)
        out.append({"kernel": "iCemTO.optimize (one MPC step)", "entry": "mbpo_icem_sample + mbpo_model_rollout + mbpo_icem_update x num_steps",
                    "config": dict(cfg, num_steps=p.num_steps), "device_us": t * 1e6, "eager_us": te * 1e6, "bound": "launch",
                    "note": "num_steps iterations of sample -> open-loop rollout -> update, host loop as icem_optimizer.py:135-252"})
        log(f"icem optimize: {t * 1e6:.1f} us (eager {te * 1e6:.1f})")

    if want("icem_case"): attempt(icem_case, 20, 100)

    res = {"device": torch.cuda.get_device_name(0), "roofs": {"hbm_GBs": HBM_PEAK_GBS, "mfma_f32_TFLOPs": MFMA_F32_PEAK_TF},
           "note": "device_us = HIP-event average per call with the calls captured into a hipGraph and replayed (device time, inputs resident in HBM; multi-launch ops timed whole); eager_us = the same call issued from Python",
           "kernels": out}
    txt = json.dumps(res, indent=1)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()

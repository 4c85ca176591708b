"""One-factor ablation of the restated [3P] semantics against the reference's learning thresholds (VERDICT r2 task 1).

The reference's only pass/fail signals are its end-to-end thresholds (tests/test_sac.py:84-89, tests/test_ppo.py:84-89: last
eval/episode_reward >= -400 and |reward of step 200| <= 0.1), at ONE JAX key that cannot be replayed here.  The CPU oracle loop
(oracle/trainer.py) run on the reference's configuration meets them on a minority of keys.  This script asks whether ONE of the
semantics SURVEY §8c lists as unverifiable — restated from knowledge of brax / optax / flax — explains that: every flip below is
applied alone, inside THIS process only (monkeypatches of the oracle modules; the oracle files themselves stay the faithful
restatement), over the same keys, and the pass count is tabulated.

    python scripts/learning_ablation.py sac  <flip> <first_key> <n_keys>      # one (flip, keys) cell; prints one JSON line per key
    python scripts/learning_ablation.py ppo  <flip> <first_key> <n_keys> [num_timesteps]
    python scripts/learning_ablation.py list
    python scripts/learning_ablation.py table <results.jsonl>                 # markdown table from collected lines

CPU only; about 2-4 minutes per SAC key on one thread.  Driver: scripts/run_learning_ablation.sh.
"""
from __future__ import annotations

import json
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))

from mbpo.systems.ensemble_system import lecun_uniform_flat  # noqa: E402  (host-side init helper: pure torch-CPU)
from mbpo.utils import keys as K  # noqa: E402
from oracle import nets, philox, ppo as oppo, replay as orep, rollout as oro, sac as osac, systems as osys, trainer as otr  # noqa: E402

SAC_FLIPS = {
    "base": "the restatement as committed (oracle/)",
    "std_floor_1e-2": "running_statistics std floor 1e-6 -> 1e-2 ([3P] std_min_value)",
    "normalize_off": "normalize_observations=False (diagnostic: is the normaliser involved at all)",
    "next_obs_pre_reset": "Transition.next_observation = x_next BEFORE AutoReset (sac/acting.py:46-55 stores the post-reset obs)",
    "no_truncation_mask": "critic error NOT masked by (1 - truncation) (sac/losses.py:104-108): time limit = termination",
    "truncation_bootstrap": "time limit bootstraps: discount 1 at truncation + pre-reset next_obs + no mask",
    "min_std_0": "NormalTanh min_std 0.001 -> 0",
    "min_std_1e-1": "NormalTanh min_std 0.001 -> 0.1",
    "target_entropy_-u": "target entropy -0.5*u -> -u (sac/losses.py:49-50 pins -0.5*u in-tree; diagnostic)",
    "init_bound_1_over_fan_in": "kernel init U(+-sqrt(1/fan_in)) instead of lecun_uniform U(+-sqrt(3/fan_in))",
    "init_lecun_normal": "kernel init N(0, 1/fan_in) instead of lecun_uniform",
    "twin_same_init": "both critics start from the same kernel draw",
    "adam_eps_in_sqrt": "Adam update m_hat / sqrt(v_hat + eps) instead of m_hat / (sqrt(v_hat) + eps)",
    "adam_no_bias_correction": "Adam without bias correction",
    "new_alpha": "critic and actor losses see the UPDATED alpha (sac.py:240 passes exp(OLD log_alpha))",
    "actor_new_q": "actor loss sees the UPDATED critics (sac.py:253 passes the OLD q_params)",
    "torch_randn_noise": "all normal draws from torch.randn instead of the build's Philox + Box-Muller",
    "relu": "relu activations (brax's builder default; the reference passes swish explicitly, sac.py:85-88)",
}
PPO_FLIPS = {
    "base": "the restatement as committed (oracle/)",
    "adv_std_sample": "advantage normalisation with the sample std (ddof=1) instead of jnp.std (population)",
    "no_adv_norm": "normalize_advantage=False (diagnostic)",
    "entropy_no_sample": "entropy bonus = Normal entropy + log-det at the MODE instead of at a fresh sample (ppo/losses.py:117)",
    "std_floor_1e-2": "running_statistics std floor 1e-6 -> 1e-2",
    "normalize_off": "normalize_observations=False (diagnostic)",
    "min_std_0": "NormalTanh min_std 0.001 -> 0",
    "init_bound_1_over_fan_in": "kernel init U(+-sqrt(1/fan_in))",
    "adam_eps_in_sqrt": "Adam update m_hat / sqrt(v_hat + eps)",
    "no_truncation": "truncation column zeroed: the time limit terminates (ppo/losses.py:89)",
    "torch_randn_noise": "all normal draws from torch.randn instead of Philox + Box-Muller",
    "relu": "relu activations",
}


# ------------------------------------------------------------------------------------------------ shared patches
def patch_common(flip: str):
    if flip == "std_floor_1e-2":
        orig = orep.stats_update
        orep.stats_update = lambda stats, batch, **kw: orig(stats, batch, std_min=1e-2, **kw)
    if flip == "min_std_0":
        nets.MIN_STD = 0.0
    if flip == "min_std_1e-1":
        nets.MIN_STD = 0.1
    if flip in ("adam_eps_in_sqrt", "adam_no_bias_correction"):
        def adamw_step(p, g, m, v, count, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
            m = b1 * m + (1 - b1) * g
            v = b2 * v + (1 - b2) * g * g
            if flip == "adam_no_bias_correction":
                u = m / (torch.sqrt(v) + eps) + wd * p
            else:
                u = (m / (1 - b1 ** count)) / torch.sqrt(v / (1 - b2 ** count) + eps) + wd * p
            return p - lr * u, m, v
        osac.adamw_step = adamw_step
        oppo.adamw_step = adamw_step
    if flip == "torch_randn_noise":
        def _normal(seed, offset, stream, shape):
            g = torch.Generator().manual_seed((hash((int(seed), int(offset), int(stream))) & (2 ** 62 - 1)))
            return torch.randn(*shape, generator=g)
        otr._normal = _normal


def init_flat(dims, gen, flip):
    w = lecun_uniform_flat(dims, gen)
    if flip == "init_bound_1_over_fan_in":
        w = w / math.sqrt(3.0)
    if flip == "init_lecun_normal":
        parts = []
        for i in range(len(dims) - 1):
            parts.append((torch.randn(dims[i], dims[i + 1], generator=gen) / math.sqrt(dims[i])).reshape(-1))
            parts.append(torch.zeros(dims[i + 1]))
        w = torch.cat(parts)
    return w


# ------------------------------------------------------------------------------------------------ SAC (tests/test_sac.py:30-57)
class AblSacLoop(otr.CpuSacLoop):
    flip = "base"

    def get_experience(self):
        X, U = self.cfg.x_dim, self.cfg.u_dim
        S, N = self.n_steps, self.n_envs
        nm, ns = self._norm()
        noise = otr._normal(self.seed, (otr.SAC_SITE_ROLLOUT << 32) + self.step_index, philox.STREAM_POLICY_NOISE, (S, N, U))
        self.env, rows = oro.rollout(self.system, self.state.params[:self.cfg.P], self.cfg.policy_dims, self.env, S,
                                     self.episode_length, self.action_repeat, self.cfg.policy_act, nm, ns, policy_noise=noise)
        if self.flip in ("next_obs_pre_reset", "truncation_bootstrap"):
            ended = rows[:, X + U + 1] == 0
            xn, _ = self.system.step(rows[:, :X], rows[:, X:X + U])
            rows[ended, X + U + 2:2 * X + U + 2] = xn[ended]
        if self.flip in ("no_truncation_mask", "truncation_bootstrap"):
            if self.flip == "truncation_bootstrap":
                rows[rows[:, -1] != 0, X + U + 1] = 1.0
            rows[:, -1] = 0.0
        self.stats = orep.stats_update(self.stats, rows[:, :X].numpy())
        self.qstate = self.queue.insert(self.qstate, rows.numpy())
        self.last_rows = rows
        return rows

    def training_step(self, n_sgd=None):
        if self.flip not in ("new_alpha", "actor_new_q"):
            return super().training_step(n_sgd)
        # sequential flavour: alpha update, then critic (new alpha), then actor (new alpha [+ new critics])
        U, cfg = self.cfg.u_dim, self.cfg
        self.get_experience()
        idx, batch = self.queue.sample(self.qstate, self.seed, (otr.SAC_SITE_SAMPLE << 32) + self.step_index)
        batch = torch.from_numpy(batch)
        nm, ns = self._norm()
        B, P, Q = self.batch_size, cfg.P, cfg.Q
        for gi in range(self.grad_updates):
            off = ((otr.SAC_SITE_SGD + gi) << 32) + self.step_index
            noise = [otr._normal(self.seed, off, s, (B, U)) for s in (philox.STREAM_SAC_ALPHA, philox.STREAM_SAC_CRITIC, philox.STREAM_SAC_ACTOR)]
            st = self.state
            mb = batch[gi * B:(gi + 1) * B]
            count = st.count + 1
            new_p, new_m, new_v = st.params.clone(), st.adam_m.clone(), st.adam_v.clone()
            groups = {"alpha": (slice(P + 2 * Q, P + 2 * Q + 1), cfg.lr_alpha), "q": (slice(P, P + 2 * Q), cfg.lr_q), "pi": (slice(0, P), cfg.lr_policy)}

            def upd(name, at):
                g, _ = osac.grads(cfg, at, st.target_q, mb, *noise, nm, ns)
                sl, lr = groups[name]
                new_p[sl], new_m[sl], new_v[sl] = osac.adamw_step(st.params[sl], g[sl], st.adam_m[sl], st.adam_v[sl], count, lr, 0.0)

            upd("alpha", st.params)
            if self.flip == "new_alpha":
                mixed = st.params.clone(); mixed[-1] = new_p[-1]
                upd("q", mixed); upd("pi", mixed)
            else:
                upd("q", st.params)
                mixed = st.params.clone(); mixed[P:P + 2 * Q] = new_p[P:P + 2 * Q]
                upd("pi", mixed)
            new_tq = st.target_q * (1 - cfg.tau) + new_p[P:P + 2 * Q] * cfg.tau
            self.state = osac.SacState(new_p, new_tq, new_m, new_v, count)
        self.step_index += 1


def run_sac(key: int, flip: str):
    """SACOptimizer.init/train + SAC.run_training's key structure around CpuSacLoop (same as scripts/sac_pendulum_seeds.py)."""
    X, U, H = 3, 1, 128
    act = "relu" if flip == "relu" else "swish"
    pd, qd = [X, H, H, H, 2 * U], [X + U, H, H, H, 1]
    state_key = K.split(key, 3)[2]
    _, run_key = K.split(state_key)
    key, subkey = K.split(run_key)
    kp, kq = K.split(subkey)
    pol = init_flat(pd, torch.Generator().manual_seed(kp % (2 ** 63)), flip)
    gq = torch.Generator().manual_seed(kq % (2 ** 63))
    q0 = init_flat(qd, gq, flip)
    q1 = q0.clone() if flip == "twin_same_init" else init_flat(qd, gq, flip)
    params = torch.cat([pol, q0, q1, torch.zeros(1)])
    key, rb_key, env_key, eval_key = K.split(key, 4)
    cfg = osac.SacConfig(X, U, pd, qd, policy_act=act, q_act=act, discounting=0.99, lr_policy=3e-4, lr_q=3e-4, lr_alpha=3e-4,
                         target_entropy=(-1.0 * U if flip == "target_entropy_-u" else None))
    N = 32
    obs0 = torch.tensor([[-1.0, 0.0, 0.0]]).repeat(N, 1)
    loop = AblSacLoop(cfg, osys.PendulumSystem(), N, 20, 200, 64, 640, 2 ** 14, flip != "normalize_off", init_params=params, init_obs=obs0)
    loop.flip = flip

    def evaluate():
        nm, ns = loop._norm()
        first = oro.EnvState(obs0[:1].clone(), obs0[:1].clone(), torch.zeros(1), torch.zeros(1))
        er, _ = oro.evaluate(loop.system, loop.state.params[:cfg.P], pd, first, 200, 1, act, nm, ns, deterministic=True)
        return float(er[0])

    evals = [evaluate()]
    key, prefill_key = K.split(key)
    loop.rekey(K.split(prefill_key)[0])
    for _ in range(4):
        loop.prefill_step()
    for _ in range(19):
        key, epoch_key = K.split(key)
        loop.rekey(epoch_key)
        for _ in range(2):
            loop.training_step()
        key, _ = K.split(key)
        evals.append(evaluate())
    nm, ns = loop._norm()
    x, r = obs0[:1].clone(), 0.0
    sysm = osys.PendulumSystem()
    for _ in range(200):
        a = torch.tanh(nets.mlp_forward(loop.state.params[:cfg.P], pd, nets.normalize(x, nm, ns), act)[:, :U])
        x, rr = sysm.step(x, a)
        r = float(rr[0])
    return evals, r


# ------------------------------------------------------------------------------------------------ PPO (tests/test_ppo.py:30-56)
def patch_ppo(flip: str):
    if flip in ("adv_std_sample", "entropy_no_sample"):
        src_loss = oppo.loss

        def loss(cfg, params, data, ent_noise, norm_mean=None, norm_std=None):
            if flip == "entropy_no_sample":
                return src_loss(cfg, params, data, torch.zeros_like(ent_noise), norm_mean, norm_std)
            # ddof=1: redo the normalisation around the restatement's own GAE
            import dataclasses
            total, terms, vs, adv = src_loss(dataclasses.replace(cfg, normalize_advantage=False), params, data, ent_noise, norm_mean, norm_std)
            X, U = cfg.x_dim, cfg.u_dim
            adv_n = (adv - adv.mean()) / (adv.std(unbiased=True) + 1e-8)
            t = {k: v.transpose(0, 1) for k, v in oppo.split_rows(data, X, U).items()}
            logits = nets.mlp_forward(params[:cfg.P], cfg.policy_dims, nets.normalize(t["obs"], norm_mean, norm_std), cfg.policy_act)
            rho = torch.exp(nets.log_prob(logits, t["raw_action"]) - t["log_prob"])
            pl = -torch.minimum(rho * adv_n, torch.clamp(rho, 1 - cfg.clipping_epsilon, 1 + cfg.clipping_epsilon) * adv_n).mean()
            total = total - terms["policy_loss"] + pl
            terms = dict(terms, policy_loss=pl, total_loss=total)
            return total, terms, vs, adv_n
        oppo.loss = loss
    if flip == "no_truncation":
        src_roll = oro.rollout

        def rollout(*a, **kw):
            st, rows = src_roll(*a, **kw)
            rows[:, -1] = 0.0
            return st, rows
        oro.rollout = rollout


def run_ppo(key: int, flip: str, steps: int):
    import dataclasses
    X, U, N, T, B, M, E = 3, 1, 256, 40, 128, 32, 8
    act = "relu" if flip == "relu" else "swish"
    pd, vd = [X, 64, 64, 2 * U], [X, 64, 64, 1]
    per_epoch = math.ceil(steps / (19 * B * T * M))
    state_key = K.split(key, 3)[2]
    _, run_key = K.split(state_key)
    k, subkey = K.split(run_key)
    k0, k1 = K.split(subkey)
    params = torch.cat([init_flat(pd, torch.Generator().manual_seed(k0 % (2 ** 63)), flip),
                        init_flat(vd, torch.Generator().manual_seed(k1 % (2 ** 63)), flip)])
    k, rb_key, env_key, eval_key = K.split(k, 4)
    cfg = oppo.PpoConfig(X, U, pd, vd, policy_act=act, value_act=act, entropy_cost=1e-1, discounting=0.99, reward_scaling=1.0,
                         gae_lambda=0.95, clipping_epsilon=0.3, normalize_advantage=(flip != "no_adv_norm"), lr=3e-3, wd=0.0)
    obs0 = torch.tensor([[-1.0, 0.0, 0.0]]).repeat(N, 1)
    loop = otr.CpuPpoLoop(cfg, osys.PendulumSystem(), N, T, 200, B, M, E, flip != "normalize_off", init_params=params, init_obs=obs0)

    def evaluate():
        nm, ns = loop._norm()
        first = oro.EnvState(obs0[:1].clone(), obs0[:1].clone(), torch.zeros(1), torch.zeros(1))
        return float(oro.evaluate(loop.system, loop.state.params[:cfg.P], pd, first, 200, 1, act, nm, ns, deterministic=True)[0][0])

    evals = [evaluate()]
    k, prefill_key = K.split(k)
    for _ in range(19):
        k, epoch_key = K.split(k)
        loop.rekey(epoch_key)
        for _ in range(per_epoch):
            loop.training_step()
        evals.append(evaluate())
    nm, ns = loop._norm()
    x, r = obs0[:1].clone(), 0.0
    sysm = osys.PendulumSystem()
    for _ in range(200):
        a = torch.tanh(nets.mlp_forward(loop.state.params[:cfg.P], pd, nets.normalize(x, nm, ns), act)[:, :U])
        x, rr = sysm.step(x, a)
        r = float(rr[0])
    return evals, r


# ------------------------------------------------------------------------------------------------ table
def table(path: str):
    rows = [json.loads(l) for l in Path(path).read_text().splitlines() if l.startswith("{")]
    out = []
    for algo, flips in (("sac", SAC_FLIPS), ("ppo", PPO_FLIPS)):
        cells = {}
        for r in rows:
            if r["algo"] == algo:
                cells.setdefault((r["flip"], r.get("steps")), {})[r["key"]] = r
        if not cells:
            continue
        out.append(f"### {algo.upper()} — reference configuration tests/test_{algo}.py:30-57, CPU oracle loop\n")
        out.append("| flip | what changes | keys | pass (eval >= -400 and |r_200| <= 0.1) | final eval per key |")
        out.append("|---|---|---|---|---|")
        for (flip, steps), by_key in sorted(cells.items(), key=lambda kv: (list(flips).index(kv[0][0]) if kv[0][0] in flips else 99, kv[0][1] or 0)):
            ks = sorted(by_key)
            npass = sum(by_key[k]["pass"] for k in ks)
            finals = " ".join(f"{by_key[k]['final']:.0f}{'*' if by_key[k]['pass'] else ''}" for k in ks)
            name = flip if not steps or algo == "sac" else f"{flip} @ {steps:,} steps"
            out.append(f"| `{name}` | {flips.get(flip, '')} | {ks[0]}..{ks[-1]} ({len(ks)}) | **{npass}/{len(ks)}** | {finals} |")
        out.append("")
    print("\n".join(out))


def main():
    mode = sys.argv[1]
    if mode == "list":
        print("sac:", " ".join(SAC_FLIPS)); print("ppo:", " ".join(PPO_FLIPS)); return
    if mode == "table":
        return table(sys.argv[2])
    flip, first_key, n_keys = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 1_000_000
    assert flip in (SAC_FLIPS if mode == "sac" else PPO_FLIPS), flip
    torch.set_num_threads(1)
    patch_common(flip)
    if mode == "ppo":
        patch_ppo(flip)
    for key in range(first_key, first_key + n_keys):
        t0 = time.time()
        evals, r = run_sac(key, flip) if mode == "sac" else run_ppo(key, flip, steps)
        ok = bool(evals[-1] >= -400 and abs(r) <= 0.1)          # (NaN compares False: a diverged run fails)
        print(json.dumps({"algo": mode, "flip": flip, "key": key, "steps": (steps if mode == "ppo" else 20_000), "final": evals[-1],
                          "best": max(evals), "r200": abs(r), "pass": ok, "seconds": round(time.time() - t0), "curve": [round(e) if e == e else None for e in evals]}),
              flush=True)


if __name__ == "__main__":
    main()

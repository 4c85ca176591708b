"""Writes semantics_kat.json: hand-derived known answers for the pieces of the hot path whose semantics come from
third-party packages the reference depends on (brax / optax / distrax) or from in-tree formulas that no reference test pins.

Every expected value below is written out with plain `math` arithmetic from the cited in-tree text for that specific input —
this script imports neither oracle/ nor the product, so the fixture is an independent pin for both:

  normal_tanh   sac/parametric_distribution.py:66-124 (the in-tree copy of brax's NormalTanhDistribution: min_std = 0.001,
                log_prob = sum_d [log N(z; loc, sigma) - log|d tanh/dz|], entropy = sum_d [H(N) + log|d tanh/dz|(z_fresh)],
                mode = tanh(loc)); log|d tanh/dz| is evaluated here as log(1 - tanh(z)^2), NOT with the numerically stable
                2(log 2 - z - softplus(-2z)) form the implementations use — the two must agree.
  bptt_log_prob policy_optimizers/bptt_optimizer.py:123-152 (Actor + get_log_prob) at action_dim = 1, including the
                [H,1] - [H] broadcast whose mean equals mean(logN) - mean(logdet).
  normalizer    bptt_optimizer.py:52-67 (Normalizer.update, EPS = 1e-8 floor).
  queue         UniformSamplingQueue insert / roll / wrap-gather as used at sac/sac.py:303,318, restated in SURVEY §8a R9.
  running_stats brax running_statistics.update as called at sac/sac.py:298-301 (two-pass form, std clip 1e-6..1e6).
  adamw / clip  optax.adamw + clip_by_global_norm as chained at sac/sac.py:175-186.
  soft_update   utils/optimizer_utils.py:155-161 and sac/sac.py:260-261.
"""
import json
import math
from pathlib import Path

LOG_SQRT_2PI = 0.5 * math.log(2 * math.pi)


def softplus(x):
    return math.log1p(math.exp(x))


def logdet(z):
    return math.log(1.0 - math.tanh(z) ** 2)


def log_normal(z, loc, sigma):
    return -0.5 * ((z - loc) / sigma) ** 2 - math.log(sigma) - LOG_SQRT_2PI


# ---------------------------------------------------------------------------------------------- NormalTanh
nt = []


def nt_case(why, loc, raw, eps, eps_entropy):
    """logits = [loc..., raw...]; z = loc + sigma*eps; entropy sample drawn with eps_entropy."""
    sigma = [softplus(r) + 0.001 for r in raw]
    z = [l + s * e for l, s, e in zip(loc, sigma, eps)]
    lp = sum(log_normal(zz, l, s) - logdet(zz) for zz, l, s in zip(z, loc, sigma))
    zf = [l + s * e for l, s, e in zip(loc, sigma, eps_entropy)]
    ent = sum(0.5 + LOG_SQRT_2PI + math.log(s) + logdet(zz) for s, zz in zip(sigma, zf))
    nt.append({"why": why, "logits": loc + raw, "eps": eps, "eps_entropy": eps_entropy, "sigma": sigma, "z": z,
               "action": [math.tanh(v) for v in z], "mode": [math.tanh(l) for l in loc], "log_prob": lp, "entropy": ent})


nt_case("loc=0, raw=0, eps=0: sigma=ln2+0.001, z=0, tanh'(0)=1 -> log_prob=-ln(sigma)-0.5 ln(2 pi); entropy=0.5+0.5 ln(2 pi)+ln(sigma)",
        [0.0], [0.0], [0.0], [0.0])
nt_case("loc=0.5, raw=1, eps=-1.2: sigma=ln(1+e)+0.001; the quadratic term is exactly -0.72", [0.5], [1.0], [-1.2], [0.3])
nt_case("two action dims: log_prob and entropy SUM over the event axis (:84-85, :93-94)", [-0.3, 0.8], [-2.0, 0.5], [0.7, -0.4], [-1.1, 0.9])
nt_case("large |z| (z ~ 4.3): tanh saturates, log(1-tanh^2) ~ -7.2 — the stable form must still agree", [3.0], [0.2], [1.6], [-0.2])
nt_case("raw=-20: softplus underflows to ~2e-9, min_std=0.001 is what is left of sigma", [0.1], [-20.0], [2.0], [1.0])

# ---------------------------------------------------------------------------------------------- BPTT Actor.get_log_prob, A = 1
# mu, sig are the Actor's outputs for H=3 steps; squashed actions a_t = clip(tanh(mu + sig*eps), +-0.999) (:313-317).
# get_log_prob (:144-152): u = atanh(a); log_l[H,1] = logN(u; mu, sig); log_l -= sum_A log(1 - a^2) -> [H] broadcast against [H,1]
# gives an [H,H] matrix M[i][j] = logN_i - logdet_j; its mean (what actor_loss uses, :350-351) = mean_i logN_i - mean_j logdet_j.
mu, sig, eps = [0.2, -0.5, 1.0], [0.7, 1.3, 0.4], [0.5, -1.0, 2.5]
a = [max(-0.999, min(0.999, math.tanh(m + s * e))) for m, s, e in zip(mu, sig, eps)]
u = [math.atanh(x) for x in a]
logn = [log_normal(uu, m, s) for uu, m, s in zip(u, mu, sig)]
ld = [math.log(1.0 - x * x) for x in a]
matrix = [[logn[i] - ld[j] for j in range(3)] for i in range(3)]
bptt = {"why": "A=1, H=3; third step: tanh(2.0)=0.964 (no clip); mean over the [H,H] broadcast = mean(logN) - mean(logdet)",
        "mu": mu, "sig": sig, "eps": eps, "squashed_action": a, "log_normal": logn, "log_det": ld, "broadcast_matrix": matrix,
        "mean_log_prob": sum(sum(r) for r in matrix) / 9.0,
        "per_step_sum_semantic": [logn[i] - ld[i] for i in range(3)],
        "note": "mean(per_step_sum_semantic) == mean_log_prob: the build's A>1 definition (sum over A per step) reduces to the reference at A=1"}
# sigma parameterisation (:135-139): sig = clip(softplus(raw + inv_softplus(init_stddev)), 1e-6, 1e2)
inv_sp = lambda y: math.log(math.expm1(y))
bptt["sigma_cases"] = [{"raw": r, "init_stddev": s0, "sig": max(1e-6, min(1e2, softplus(r + inv_sp(s0))))}
                       for r, s0 in ((0.0, 1.0), (0.0, 2.0), (-30.0, 1.0), (200.0, 1.0), (0.7, 0.3))]

# ---------------------------------------------------------------------------------------------- Normalizer.update
norm = []
# from the initial state (mean 0, std 1, size 0) with x = [1, 3]: total=2, mean=2, s_n = 0 + (1+1) + 0 = 2, std = 1
norm.append({"why": "init state, x=[1,3]: mean=(0+4)/2=2; s_n=1*0+[(1-2)^2+(3-2)^2]+0=2; std=sqrt(2/2)=1",
             "mean": 0.0, "std": 1.0, "size": 0, "x": [1.0, 3.0], "new_mean": 2.0, "new_std": 1.0, "new_size": 2})
# then x = [5]: total=3, mean=(2*2+5)/3=3, s_n = 1*2 + (5-3)^2 + 2*(2-3)^2 = 2+4+2 = 8, std = sqrt(8/3)
norm.append({"why": "state (2,1,2), x=[5]: mean=3; s_n=1*2+4+2*1=8; std=sqrt(8/3)",
             "mean": 2.0, "std": 1.0, "size": 2, "x": [5.0], "new_mean": 3.0, "new_std": math.sqrt(8.0 / 3.0), "new_size": 3})
norm.append({"why": "constant data from the init state: variance 0 -> std floored at EPS=1e-8 (:62)",
             "mean": 0.0, "std": 1.0, "size": 0, "x": [4.0, 4.0, 4.0], "new_mean": 4.0, "new_std": 1e-8, "new_size": 3})

# ---------------------------------------------------------------------------------------------- queue
# max_replay_size = 5, rows are single numbers.  a = [10,11,12], b = [20,21,22], c = [30,31]
queue = {"max_replay_size": 5, "steps": [
    {"why": "insert 3 into empty: roll=min(0,5-0-3)=0; data=[10,11,12,0,0]; insert=3; sample=0", "insert": [10.0, 11.0, 12.0],
     "data": [10.0, 11.0, 12.0, 0.0, 0.0], "insert_position": 3, "sample_position": 0},
    {"why": "insert 3 more: roll=min(0,5-3-3)=-1 -> data rolled left by 1 = [11,12,0,0,10]; position 3-1=2; write at 2..4; "
            "insert=(2+3)%6=5; sample=max(0,0-1)=0", "insert": [20.0, 21.0, 22.0],
     "data": [11.0, 12.0, 20.0, 21.0, 22.0], "insert_position": 5, "sample_position": 0},
    {"why": "insert 2 when full: roll=min(0,5-5-2)=-2 -> [20,21,22,11,12]; position 3; write at 3..4; insert=5; sample=0",
     "insert": [30.0, 31.0], "data": [20.0, 21.0, 22.0, 30.0, 31.0], "insert_position": 5, "sample_position": 0}],
    "gather": {"why": "jnp.take(mode='wrap') on the final data: idx 7 -> 7 mod 5 = 2; -1 -> 4; 5 -> 0",
               "idx": [0, 4, 7, -1, 5], "rows": [20.0, 31.0, 22.0, 31.0, 20.0]}}

# ---------------------------------------------------------------------------------------------- running statistics
rs = []
rs.append({"why": "init (count 0, mean 0, sv 0), batch [1,3]: count=2; d_old=[1,3]; mean=4/2=2; d_new=[-1,1]; sv=1*-1+3*1=2; std=sqrt(2/2)=1",
           "count": 0.0, "mean": 0.0, "summed_variance": 0.0, "batch": [1.0, 3.0],
           "new_count": 2.0, "new_mean": 2.0, "new_summed_variance": 2.0, "new_std": 1.0})
rs.append({"why": "then batch [5]: count=3; d_old=3; mean=2+3/3=3; d_new=2; sv=2+6=8; std=sqrt(8/3)",
           "count": 2.0, "mean": 2.0, "summed_variance": 2.0, "batch": [5.0],
           "new_count": 3.0, "new_mean": 3.0, "new_summed_variance": 8.0, "new_std": math.sqrt(8.0 / 3.0)})
rs.append({"why": "constant batch from init: sv=0 -> std clipped up to std_min_value=1e-6",
           "count": 0.0, "mean": 0.0, "summed_variance": 0.0, "batch": [7.0, 7.0],
           "new_count": 2.0, "new_mean": 7.0, "new_summed_variance": 0.0, "new_std": 1e-6})

# ---------------------------------------------------------------------------------------------- adamw, clip, soft update
p, g, lr, wd = 1.0, 0.5, 0.1, 0.01
m1, v1 = 0.1 * g, 0.001 * g * g
upd = (m1 / (1 - 0.9)) / (math.sqrt(v1 / (1 - 0.999)) + 1e-8) + wd * p
adam = [{"why": "first step from zero moments: m_hat=g, v_hat=g^2 -> update = g/(|g|+eps) + wd*p = 1/(1+2e-8)+0.01; p' = p - lr*update",
         "p": p, "g": g, "m": 0.0, "v": 0.0, "count": 1, "lr": lr, "wd": wd, "new_m": m1, "new_v": v1, "new_p": p - lr * upd}]
m2, v2 = 0.9 * m1 + 0.1 * (-1.0), 0.999 * v1 + 0.001 * 1.0
p1 = p - lr * upd
upd2 = (m2 / (1 - 0.9 ** 2)) / (math.sqrt(v2 / (1 - 0.999 ** 2)) + 1e-8) + wd * p1
adam.append({"why": "second step, g=-1: bias corrections 1-0.9^2=0.19 and 1-0.999^2=0.001999",
             "p": p1, "g": -1.0, "m": m1, "v": v1, "count": 2, "lr": lr, "wd": wd, "new_m": m2, "new_v": v2, "new_p": p1 - lr * upd2})
clip = [{"why": "norm 5 >= max_norm 1: g/5*1", "g": [3.0, 4.0], "max_norm": 1.0, "out": [0.6, 0.8]},
        {"why": "norm 5 < max_norm 10: unchanged", "g": [3.0, 4.0], "max_norm": 10.0, "out": [3.0, 4.0]}]
soft = [{"why": "tau=0.25: 0.75*[1,2] + 0.25*[5,-2]", "target": [1.0, 2.0], "online": [5.0, -2.0], "tau": 0.25, "out": [2.0, 1.0]},
        {"why": "tau=0.005 (the default): 0.995*4 + 0.005*0", "target": [4.0], "online": [0.0], "tau": 0.005, "out": [3.98]}]

out = {"_doc": __doc__, "normal_tanh": nt, "bptt_log_prob": bptt, "normalizer": norm, "queue": queue, "running_stats": rs,
       "adamw": adam, "clip_by_global_norm": clip, "soft_update": soft}
Path(__file__).with_name("semantics_kat.json").write_text(json.dumps(out, indent=1))
print("wrote semantics_kat.json:", {k: (len(v) if isinstance(v, list) else "-") for k, v in out.items() if k != "_doc"})

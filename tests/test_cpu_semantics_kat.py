"""CPU: the oracle against tests/golden/semantics_kat.json — hand-derived known answers (generator:
tests/golden/make_semantics_kat.py, plain `math`, imports neither oracle/ nor the product) for the pieces whose semantics
come from brax / optax / distrax or from in-tree formulas no reference test pins.  fp64 throughout: rtol 1e-12."""
import json
import math
from pathlib import Path

import numpy as np
import torch

from oracle import bptt as obptt, nets as onets, replay as orep, sac as osac

KAT = json.loads((Path(__file__).parent / "golden" / "semantics_kat.json").read_text())
T = lambda v: torch.tensor(v, dtype=torch.float64)


def test_normal_tanh_kat():
    for c in KAT["normal_tanh"]:
        logits, eps, eps_e = T(c["logits"])[None], T(c["eps"])[None], T(c["eps_entropy"])[None]
        z = onets.sample_no_postprocessing(logits, eps)
        np.testing.assert_allclose(z[0].numpy(), c["z"], rtol=1e-12, atol=1e-14, err_msg=c["why"])
        np.testing.assert_allclose(onets.postprocess(z)[0].numpy(), c["action"], rtol=1e-12, err_msg=c["why"])
        np.testing.assert_allclose(onets.mode(logits)[0].numpy(), c["mode"], rtol=1e-12, err_msg=c["why"])
        # the stable log-det form vs log(1 - tanh^2): the saturated case loses digits in the KAT's own naive form
        tol = 1e-9 if max(abs(v) for v in c["z"]) > 3 else 1e-12
        np.testing.assert_allclose(float(onets.log_prob(logits, z)), c["log_prob"], rtol=tol, err_msg=c["why"])
        np.testing.assert_allclose(float(onets.entropy(logits, eps_e)), c["entropy"], rtol=1e-9, err_msg=c["why"])


def test_bptt_log_prob_kat():
    c = KAT["bptt_log_prob"]
    mu, sig, a = T(c["mu"])[:, None], T(c["sig"])[:, None], T(c["squashed_action"])[:, None]
    per_step = obptt.log_prob_steps(mu, sig, a)
    np.testing.assert_allclose(per_step.numpy(), c["per_step_sum_semantic"], rtol=1e-10)
    # the reference's [H,1] - [H] broadcast, written out, has the same mean
    m = np.asarray(c["broadcast_matrix"])
    assert m.shape == (3, 3)
    np.testing.assert_allclose(m.mean(), c["mean_log_prob"], rtol=1e-12)
    np.testing.assert_allclose(float(per_step.mean()), c["mean_log_prob"], rtol=1e-10)
    for s in c["sigma_cases"]:
        got = float(torch.clamp(torch.nn.functional.softplus(T(s["raw"]) + obptt.inv_softplus(s["init_stddev"])), 1e-6, 1e2))
        np.testing.assert_allclose(got, s["sig"], rtol=1e-12)


def test_normalizer_update_kat():
    for c in KAT["normalizer"]:
        mean, std, size = obptt.normalizer_update(T(c["x"])[:, None], T([c["mean"]]), T([c["std"]]), c["size"])
        assert size == c["new_size"], c["why"]
        np.testing.assert_allclose([float(mean), float(std)], [c["new_mean"], c["new_std"]], rtol=1e-12, err_msg=c["why"])


def test_queue_kat():
    q = KAT["queue"]
    queue = orep.UniformSamplingQueue(q["max_replay_size"], 1, 1)
    st = queue.init()
    for step in q["steps"]:
        st = queue.insert(st, np.asarray(step["insert"], np.float32)[:, None])
        assert st["data"][:, 0].tolist() == step["data"], step["why"]
        assert (int(st["insert_position"]), int(st["sample_position"])) == (step["insert_position"], step["sample_position"]), step["why"]
    assert queue.gather(st, np.asarray(q["gather"]["idx"]))[:, 0].tolist() == q["gather"]["rows"]


def test_running_statistics_kat():
    for c in KAT["running_stats"]:
        s = np.array([c["count"], c["mean"], c["summed_variance"], 1.0], np.float64)
        out = orep.stats_update(s, np.asarray(c["batch"], np.float64)[:, None], dtype=np.float64)
        np.testing.assert_allclose(out, [c["new_count"], c["new_mean"], c["new_summed_variance"], c["new_std"]], rtol=1e-12, err_msg=c["why"])


def test_adamw_clip_soft_update_kat():
    for c in KAT["adamw"]:
        p, m, v = osac.adamw_step(T([c["p"]]), T([c["g"]]), T([c["m"]]), T([c["v"]]), c["count"], c["lr"], c["wd"])
        np.testing.assert_allclose([float(p), float(m), float(v)], [c["new_p"], c["new_m"], c["new_v"]], rtol=1e-12, err_msg=c["why"])
    for c in KAT["clip_by_global_norm"]:
        np.testing.assert_allclose(osac.clip_by_global_norm(T(c["g"]), c["max_norm"]).numpy(), c["out"], rtol=1e-12, err_msg=c["why"])
    # soft_update as sgd_step applies it (sac.py:260-261): target*(1-tau) + new*tau
    for c in KAT["soft_update"]:
        out = T(c["target"]) * (1 - c["tau"]) + T(c["online"]) * c["tau"]
        np.testing.assert_allclose(out.numpy(), c["out"], rtol=1e-12, err_msg=c["why"])

"""Pass rates of the HIP trainers at the reference's own acceptance configurations (tests/test_sac.py:30-89, tests/test_ppo.py:30-89 of
the reference), over keys — the GPU counterpart of scripts/learning_ablation.py's `base` rows.  One JSON line per run on stdout.
    python scripts/hip_learning_rates.py > profiles/r03_hip_learning.jsonl"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd")); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_host_api as T  # noqa: E402  (the tests' own configuration helpers: the reference's kwargs verbatim)


def sac_run(dev, key):
    from mbpo.optimizers import SACOptimizer
    system, sampling_buffer, sbs = T._one_row_true_buffer(dev)
    optimizer = SACOptimizer(system=system, true_buffer=sampling_buffer, num_timesteps=20_000, num_evals=20, reward_scaling=1,
                             episode_length=200, normalize_observations=True, action_repeat=1, discounting=0.99,
                             lr_policy=3e-4, lr_alpha=3e-4, lr_q=3e-4, num_envs=32, batch_size=64,
                             grad_updates_per_step=20 * 32, max_replay_size=2 ** 14, min_replay_size=2 ** 7, num_eval_envs=1,
                             deterministic_eval=True, tau=0.005, wd_policy=0, wd_q=0, wd_alpha=0,
                             num_env_steps_between_updates=20, policy_hidden_layer_sizes=(128, 128, 128),
                             critic_hidden_layer_sizes=(128, 128, 128))
    out = optimizer.train(opt_state=optimizer.init(key=key, true_buffer_state=sbs))
    r_last = T._closed_loop_last_reward(system, optimizer, out.optimizer_state)
    evals = [round(m["eval/episode_reward"]) for m in out.summary]
    return evals, r_last, evals[-1] >= -400 and abs(r_last) <= 0.1


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for key in range(20):
        evals, r_last, good = sac_run(dev, key)
        print(json.dumps({"path": "hip", "algo": "sac", "key": key, "pass": bool(good), "final_eval": evals[-1], "abs_r200": abs(r_last), "evals": evals}), flush=True)
    for steps, keys in ((1_000_000, range(12)), (4_000_000, range(6))):
        for key in keys:
            evals, r_last, good = T._ppo_reference_run(dev, key, steps)
            print(json.dumps({"path": "hip", "algo": "ppo", "num_timesteps": steps, "key": key, "pass": bool(good), "final_eval": evals[-1],
                              "abs_r200": abs(r_last), "evals": evals}), flush=True)


if __name__ == "__main__":
    main()

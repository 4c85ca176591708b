"""GAE and lambda-return — plain sequential loops following the reference line by line (test infrastructure).

compute_gae:   mbpo/optimizers/policy_optimizers/ppo/losses.py:128-184  (time-major [T,B])
lambda_return: mbpo/utils/optimizer_utils.py:119-152                     ([T] per trajectory, here [T,B])
PINNED by hand-derived cases in tests/golden/scan_kat.json (T<=3 incl. truncation/termination masks).
"""
from __future__ import annotations

import numpy as np


def compute_gae(truncation, termination, rewards, values, bootstrap_value, discounting, gae_lambda, dtype=np.float64):
    truncation = np.asarray(truncation, dtype)
    termination = np.asarray(termination, dtype)
    rewards = np.asarray(rewards, dtype)
    values = np.asarray(values, dtype)
    bootstrap_value = np.asarray(bootstrap_value, dtype)
    T = truncation.shape[0]
    discounting = np.asarray(discounting, dtype)                                         # scalar, or [T,B]: the per-step discount
    per_step = discounting.ndim > 0                                                      # of non_equidistant_time (losses_new.py:181-226)
    truncation_mask = 1 - truncation                                                     # :153
    values_t_plus_1 = np.concatenate([values[1:], bootstrap_value[None]], axis=0)       # :155-156
    deltas = rewards + discounting * (1 - termination) * values_t_plus_1 - values       # :157
    deltas = deltas * truncation_mask                                                    # :158
    acc = np.zeros_like(bootstrap_value)                                                 # :160
    vs_minus_v_xs = np.zeros_like(values)
    for t in range(T - 1, -1, -1):                                                       # reverse scan :169-174
        disc_t = discounting[t] if per_step else discounting
        acc = deltas[t] + disc_t * (1 - termination[t]) * truncation_mask[t] * gae_lambda * acc   # :166
        vs_minus_v_xs[t] = acc
    vs = vs_minus_v_xs + values                                                          # :176
    vs_t_plus_1 = np.concatenate([vs[1:], bootstrap_value[None]], axis=0)               # :178-179
    advantages = (rewards + discounting * (1 - termination) * vs_t_plus_1 - values) * truncation_mask   # :180-181
    return vs, advantages


def lambda_return(reward, next_values, discount, lambda_, dtype=np.float64):
    reward = np.asarray(reward, dtype)
    next_values = np.asarray(next_values, dtype)
    inputs = reward + discount * next_values * (1 - lambda_)                             # :128
    agg = next_values[-1]                                                                # start (:131)
    out = np.zeros_like(inputs)
    for t in range(inputs.shape[0] - 1, -1, -1):                                         # static_scan reverse :135-152
        agg = inputs[t] + discount * lambda_ * agg                                       # :130
        out[t] = agg
    return out

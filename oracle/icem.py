"""iCEM trajectory optimizer — numpy restatement (test infrastructure).

  powerlaw_psd_gaussian          mbpo/utils/general_utils.py:81-208  (Timmer & Koenig coloured noise, unit variance)
  iCemTO.optimize / step         mbpo/optimizers/trajectory_optimizers/icem_optimizer.py:135-257
  rollout_actions                mbpo/utils/optimizer_utils.py:11-59
Randomness: the build's Philox stream ICEM — element ((sample*U + dim)*K + k)*2 + {0: real, 1: imaginary} — instead of
jax.random; the spectrum scaling, the DC/Nyquist corrections, the irfft and the normalisation follow the reference line by line.
"""
from __future__ import annotations

import numpy as np

from . import philox


def powerlaw_psd_gaussian(exponent: float, samples: int, sr: np.ndarray, si: np.ndarray) -> np.ndarray:
    """sr, si: standard-normal draws [..., K], K = samples//2 + 1.  Returns [..., samples]."""
    f = np.fft.rfftfreq(samples)                               # :134
    fmin = max(0.0, 1.0 / samples)                             # :137-138
    s_scale = f.copy()
    ix = int(np.sum(s_scale < fmin))                           # :144
    if ix and ix < len(s_scale):                               # :158-164
        s_scale[:ix] = s_scale[ix]
    s_scale = s_scale ** (-exponent / 2.0)                     # :165
    w = s_scale[1:].copy()                                     # :168-170
    w[-1] *= (1 + (samples % 2)) / 2.0
    sigma = 2 * np.sqrt(np.sum(w ** 2)) / samples
    sr = sr * s_scale                                          # :183-184
    si = si * s_scale
    if not (samples % 2):                                      # :188-190
        si[..., -1] = 0
        sr[..., -1] = sr[..., -1] * np.sqrt(2)
    si[..., 0] = 0                                             # :193-194
    sr[..., 0] = sr[..., 0] * np.sqrt(2)
    s = sr + 1j * si                                           # :197
    return np.fft.irfft(s, n=samples, axis=-1) / sigma         # :200


def sample_candidates(mean, std, prev_elites, u_min, u_max, n_samples, horizon, u_dim, exponent, seed, offset, dtype=np.float64):
    """[n_samples + n_prev, H, U] candidates (icem_optimizer.py:176-190)."""
    K = horizon // 2 + 1
    idx = np.arange(n_samples * u_dim * K * 2, dtype=np.uint64)
    z = philox.philox_normal(seed, offset, philox.STREAM_ICEM, idx).astype(dtype).reshape(n_samples, u_dim, K, 2)
    colored = powerlaw_psd_gaussian(exponent, horizon, z[..., 0].copy(), z[..., 1].copy())      # [S, U, H]
    colored = np.transpose(colored, (0, 2, 1))                                                   # [S, H, U]
    a = np.clip(mean[None] + colored * std[None], u_min, u_max)
    return np.concatenate([a, prev_elites], axis=0).astype(dtype)


def update(values, candidates, mean, std, best_value, best_sequence, n_elites, n_prev, alpha):
    """icem_optimizer.py:196-232.  Returns (mean, std, best_value, best_sequence, prev_elites)."""
    order = np.argsort(values, axis=0, kind="stable")[-n_elites:]
    elites, elite_values = candidates[order], values[order]
    elite_mean, elite_var = elites.mean(axis=0), elites.var(axis=0)
    new_mean = mean * alpha + (1 - alpha) * elite_mean
    new_std = np.sqrt(std ** 2 * alpha + (1 - alpha) * elite_var)
    if best_value <= elite_values[-1]:
        best_value, best_sequence = elite_values[-1], elites[-1]
    return new_mean, new_std, best_value, best_sequence, elites[-n_prev:]


def objective(system_step, x0, candidates, n_particles, use_max=False):
    """values[c] = summarize_particles(mean_t reward) for a deterministic system (all particles coincide)."""
    NC, H, _ = candidates.shape
    x = np.repeat(x0[None], NC, axis=0)
    tot = np.zeros(NC, candidates.dtype)
    for t in range(H):
        x, r = system_step(x, candidates[:, t])
        tot += r
    return tot / H

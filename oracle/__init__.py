"""oracle/ — CPU restatement of the reference's MBPO inner-loop arithmetic.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
path (model-based-policy-optimizers_amd/) never does and has no CPU fallback.

PARITY STATUS — read before trusting a number checked against this package:
  * The reference (lasgroup/Model-based-policy-optimizers @ 2024-12-18) is pure Python/JAX and CANNOT be
    imported here: jax, jaxlib, flax, optax, chex, distrax, brax are absent (ordinary ModuleNotFoundError,
    SURVEY.md §8c) and there is no network.  The reference ships NO golden vectors, known-answer tests or
    fixtures for this path (its tests are learning-outcome thresholds).
  * PINNED: Pendulum dynamics/reward (closed-form known answers derived by hand from
    mbpo/systems/dynamics/pendulum_dynamics.py:35-63 and rewards/pendulum_reward.py:32-41 —
    tests/golden/pendulum_kat.json), GAE / lambda-return hand-derived cases (tests/golden/scan_kat.json).
  * PARITY UNPINNED: everything restated from third-party semantics that are not in the reference tree —
    brax (UniformSamplingQueue, running_statistics, networks, NormalTanhDistribution, wrappers),
    optax (adamw, clip_by_global_norm), flax (Dense, swish, lecun_uniform), distrax (Normal, Tanh) — and the
    learned ensemble Dynamics, which does not exist in the reference at all.  Each such function says
    "[3P, unverifiable here]" in its docstring and cites the in-tree call site / dead-code text it follows.
  * JAX threefry PRNG streams are not reproducible here: parity is defined on identical EXPLICIT random
    tensors (noise, indices, permutations); the build's own counter-based Philox generator is restated
    bit-exactly in oracle/philox.py.

Every function cites the reference file:line it follows (paths relative to the reference repo root).
"""

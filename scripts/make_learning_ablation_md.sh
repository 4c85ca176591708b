#!/bin/bash
# profiles/r03_learning_ablation.md = header + the tables of scripts/learning_ablation.py over profiles/r03_learning_ablation.jsonl + conclusion
cd "$(dirname "$0")/.."
OUT=profiles/r03_learning_ablation.md
cat > $OUT <<'HDR'
# Learning thresholds of the reference's own tests — one-factor ablation (round 3, VERDICT r02 task 1)

The only reference-held pass/fail signals for the hot path are `tests/test_sac.py:84-89` and `tests/test_ppo.py:84-89` of the reference:
after training at the configuration those files hold, `eval/episode_reward >= -400` and `|reward at step 200| <= 0.1`, both with
`PRNGKey(0)`.  The reference cannot run here (no jax / brax / optax / flax: SURVEY §8c), so its pass RATE over keys is unknown; what can
be measured is the rate of this repository's restatement, and whether any single restated third-party semantic moves it.

* Every row below is the CPU oracle loop (`oracle/trainer.py`: `CpuSacLoop` / `CpuPpoLoop`) at the reference's configuration verbatim,
  with ONE thing changed, over the listed keys (`scripts/learning_ablation.py`, `scripts/run_learning_ablation.sh`; raw rows:
  `profiles/r03_learning_ablation.jsonl`, one JSON line per (algorithm, flip, key) with the evaluation curve).  `*` marks a key that
  passes BOTH thresholds.  A key is the integer handed to the trainer (`PRNGKey(k)`); the restatement's random streams are the
  build's Philox generator, not jax's threefry, so key k here and key k upstream are different draws of the same distribution.
* The HIP trainers follow the oracle loop curve for curve (DESIGN §5 table; `tests/test_gpu_trainer_parity.py`).  Their own pass
  rates at the same configurations (`scripts/hip_learning_rates.py` on one MI355X, raw rows `profiles/r03_hip_learning.jsonl`):
  **SAC 9/20 on keys 0..19 — keys 3, 6, 9, 10, 11, 12, 13, 17, 18: exactly the nine keys the CPU oracle loop passes**, with final
  returns within a few units of the oracle's on every key; PPO at the reference's 1 M steps 2/12 (keys 0, 10), at 4 M steps 4/6
  (keys 0, 3, 4, 5) — PPO's outcome at the reference's budget is chaotic in last-bit arithmetic (HIP and oracle agree on the
  distribution, not on which key passes: the oracle passes 0, 6, 9 of 0..9).  `tests/test_gpu_host_api.py` runs the reference's
  configurations verbatim over several keys and reports each key's outcome (xfail on a miss at the reference budget) instead of
  pinning one lucky key.

HDR
python scripts/learning_ablation.py table profiles/r03_learning_ablation.jsonl >> $OUT
cat >> $OUT <<'FTR'

## What the tables say

1. **No restated third-party semantic explains the pass rate.**  Every flip of a [3P] semantic the survey lists as unverifiable — the
   normaliser's std floor, pre- vs post-reset `next_observation`, the truncation mask and time-limit bootstrapping, `min_std`, the target
   entropy, Adam's epsilon placement and bias correction, which alpha / critics the losses see, the advantage std's ddof, the
   generator behind the normal draws — leaves SAC at 3/10 on keys 0..9 with the SAME three keys passing (3, 6, 9) and nearly the same
   final returns, and leaves PPO between 1/10 and 4/10.  The outcome is bimodal (≈ −350: swings up and balances; ≈ −1600: never leaves
   the hanging position) and decided early.
2. **What does move it is not a restated semantic but the task's conditioning.**  Switching observation normalisation OFF (the
   reference's tests switch it ON) takes SAC to 19/20 and PPO to 10/10: every episode starts from the single state the reference's
   `PendulumSystem.reset` returns, `[-1, 0, 0]`, so the running statistics are fitted on almost constant cos θ / sin θ columns first
   and blow those features up by their tiny std once the pendulum moves.  A smaller kernel initialisation (`U(±sqrt(1/fan_in))`, or
   `lecun_normal`) takes SAC to 15/20 and 12/20 and PPO to 6/10: the same sensitivity from the other side.  Both are in-tree facts of
   the reference (its test configuration, its `lecun_uniform` in `sac/networks.py:23,64,91`), restated here as they are written.
3. **Over 20 keys the restatement's SAC passes 9/20** (3/10 on keys 0..9, 6/10 on keys 10..19): the "2/10" of round 2 was a small
   sample of a ≈ 45 % rate.  Upstream CI runs ONE key; a 45 % (SAC) / 30 % (PPO) rate is compatible with that key passing there.
4. Two semantics ARE load-bearing and are restated as upstream has them: PPO's entropy bonus evaluated at a fresh sample
   (`ppo/losses.py:117`; at the mode: 0/10) and the swish activations the reference passes explicitly (relu, brax's builder default:
   SAC 4/10, PPO 0/10).

**Conclusion: reference pass rate unpinned.**  The restatement reaches the reference's thresholds on 9 of 20 (SAC) and 3 of 10 (PPO, at
the reference's 1 M steps) keys; no single-factor change of a third-party semantic moves those numbers, and the two factors that do are
the reference's own configuration.  Whether upstream's rate differs cannot be decided without running upstream.
FTR
echo "wrote $OUT"

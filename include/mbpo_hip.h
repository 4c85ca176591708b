/*
 * mbpo_hip.h — C-ABI of libmbpo_hip.so, the MI355X (gfx950) implementation of the
 * MBPO inner-loop hot path of lasgroup/Model-based-policy-optimizers.
 *
 * The reference has no FFI: its hot path is jax.numpy traced under jit/scan/vmap.  Each
 * entry point below replaces one traced computation; the "replaces" line cites the
 * reference file:line (relative to the reference repo root) whose arithmetic it computes.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; every pointer is a DEVICE pointer unless named h_*;
 *   - the caller owns all memory; no entry point allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value: MBPO_OK (0) or a negative MBPO_ERR_*; mbpo_last_error() gives text;
 *   - all floating point is IEEE fp32 (the reference's dtype: sac/sac.py:379,389);
 *     indices and positions are int32/int64 and bit-exact.
 *
 * Parameter layout of an MLP ("flat params"): for layer l = 0..n_layers-1
 *   W_l[dims[l]][dims[l+1]] row-major (flax Dense kernel layout [in,out]), then b_l[dims[l+1]].
 * n_nets networks of identical shape sit net_stride floats apart (ensemble members, twin critics).
 */
#ifndef MBPO_HIP_H
#define MBPO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBPO_OK 0
#define MBPO_ERR_ARG (-1)         /* invalid argument / shape */
#define MBPO_ERR_UNSUPPORTED (-2) /* valid but outside what the kernels are built for */
#define MBPO_ERR_LAUNCH (-3)      /* HIP runtime error at launch */

#define MBPO_MAX_LAYERS 8

/* activation ids (reference default: flax.linen.swish — sac/sac.py:86-88, utils/network_utils.py:8) */
#define MBPO_ACT_SWISH 0
#define MBPO_ACT_RELU 1
#define MBPO_ACT_TANH 2

typedef struct mbpo_mlp_desc {
  const float *params;               /* flat params of net 0 */
  int64_t net_stride;                /* floats between consecutive nets */
  int32_t n_nets;                    /* >= 1 (ensemble members / critics) */
  int32_t n_layers;                  /* Dense layers incl. the output layer, 1..MBPO_MAX_LAYERS */
  int32_t dims[MBPO_MAX_LAYERS + 1]; /* dims[0] = input size, dims[n_layers] = output size */
  int32_t activation;                /* MBPO_ACT_*; applied after every layer except the last */
} mbpo_mlp_desc;

/* ---- library ---- */
int mbpo_version(void);
const char *mbpo_last_error(void);

/* ---- randomness ---------------------------------------------------------------------------------
 * The reference threads jax.random keys through every call and splits them (sac/sac.py:289,309-311; ppo/ppo.py:190-193).
 * JAX's threefry streams cannot be reproduced without JAX, so every drawing entry point here is keyed by
 * (seed, offset, stream id, element index) -> Philox4x32-10 (csrc/common.hpp; stream ids are fixed per consumer).
 * `rng_dev` (optional, device uint64[2] = {seed word, step counter}) is ADDED to the host-side (seed, offset):
 *   seed_eff = seed + rng_dev[0],   offset_eff = offset + rng_dev[1].
 * A training step captured into a hipGraph bakes its host-side seed/offset constants; everything that changes from step
 * to step lives in the two device words, so a replayed graph draws bit for bit what the eagerly issued step draws.
 * Convention used by the trainers: offset = (call-site id << 32), rng_dev = {epoch key, training-step index}.
 * mbpo_rng_advance: rng_dev[1] += inc (one tiny launch, graph-capturable). */
int mbpo_rng_advance(uint64_t *rng_dev, uint64_t inc, void *stream);
/* out[i] = standard normal of element (elem_base + i) of `stream` (1 = policy noise ... 10 = iCEM; csrc/common.hpp) under
 * (seed + rng_dev[0], offset + rng_dev[1]) — exactly the number a fused kernel draws for that element.  For host-side horizon
 * loops around a user-defined System (BPTT: the noise of jr.normal in act(), bptt_optimizer.py:305-325). */
int mbpo_philox_normal_fill(uint64_t seed, uint64_t offset, const uint64_t *rng_dev, uint32_t stream, uint64_t elem_base, int64_t n,
                            float *out, void *stream_);

/* ---- R2: ensemble MLP forward -------------------------------------------------------------
 * replaces: the (new) learned Dynamics.next_state evaluated under vmap —
 *           mbpo/systems/dynamics/base_dynamics.py:15-20 called from
 *           mbpo/systems/pendulum_system.py:31-32 / brax_utils/training.py:71-74;
 *           MLP semantics: sac/networks.py:19-41, utils/network_utils.py:5-17.
 * x: [n_rows, dims[0]] if shared_input else [n_nets, n_rows, dims[0]];  y: [n_nets, n_rows, dims[n_layers]].
 */
int mbpo_ensemble_mlp_forward(const mbpo_mlp_desc *mlp, const float *x, int32_t shared_input,
                              float *y, int64_t n_rows, void *stream);

/* ---- R1-R8: fused model rollout -----------------------------------------------------------
 * replaces: SAC.get_experience's scan of acting.actor_step (sac/sac.py:283-296, sac/acting.py:35-55)
 *           and brax generate_unroll as used by PPO.training_step (ppo/ppo.py:194-208), i.e. per env and step:
 *           policy inference (sac_networks.py:58-73 / ppo_network.py:59-84, NormalTanh:
 *           sac/parametric_distribution.py:66-124), AutoReset/Episode bookkeeping
 *           (brax_utils/training.py:77-137), BraxWrapper.step -> System.step
 *           (systems/brax_wrapper.py:40-50, pendulum_system.py:18-39), Transition assembly.
 */
#define MBPO_SYS_PENDULUM 0 /* analytic PendulumDynamics (dynamics/pendulum_dynamics.py:29-63) */
#define MBPO_SYS_ENSEMBLE 1 /* learned ensemble MLP behind Dynamics.next_state */

#define MBPO_ENS_MEAN 0  /* x' = E_members[mean_e]  (what System.step consumes: .mean()) */
#define MBPO_ENS_TS1 1   /* member drawn per (env, step) — MBPO-style trajectory sampling */
#define MBPO_ENS_TSINF 2 /* env i bound to member i % E */

#define MBPO_REWARD_PENDULUM 0  /* rewards/pendulum_reward.py:27-42; params [angle_cost, control_cost, target_angle] */
#define MBPO_REWARD_QUADRATIC 1 /* -(sum q_d (x_d - t_d)^2) - sum r_d u_d^2; params [t[x], q[x], r[u]] */

typedef struct mbpo_rollout_desc {
  mbpo_mlp_desc policy;   /* [x_dim] -> [2*u_dim] */
  mbpo_mlp_desc dynamics; /* [x_dim+u_dim] -> [2*x_dim] (mean, raw std); ignored for MBPO_SYS_PENDULUM */
  int32_t x_dim, u_dim;
  int64_t n_envs;          /* N */
  int32_t n_steps;         /* S = num_env_steps_between_updates (SAC) or unroll_length (PPO) */
  int32_t episode_length;  /* model-env horizon H (EpisodeWrapper) */
  int32_t action_repeat;   /* EpisodeWrapper.action_repeat */
  int32_t system_kind;     /* MBPO_SYS_* */
  int32_t ens_mode;        /* MBPO_ENS_* */
  int32_t ens_predict_delta; /* 1: x' = x + f(x,u) */
  int32_t ens_sample_noise;  /* 1 (TS modes): x' += sigma_e * eps */
  float ens_min_std;         /* sigma = softplus(raw) + ens_min_std */
  int32_t reward_kind;       /* MBPO_REWARD_* */
  const float *reward_params;
  const float *sys_params;   /* pendulum: [max_speed, max_torque, dt, g, m, l] */
  const float *norm_mean;    /* [x_dim] or NULL (normalize_observations=False) */
  const float *norm_std;     /* [x_dim] or NULL */
  int32_t deterministic;     /* 1: action = tanh(loc) (mode) */
  int32_t ppo_extras;        /* 1: rows carry log_prob and raw_action (ppo_network.py:72-80) */
  int32_t env_major;         /* 0: row = s*N + i (SAC concat order, sac.py:296); 1: row = i*S + s (PPO [B*M,T]) */
  float action_clip;         /* > 0: action = clip(tanh(z), +-action_clip) — BPTT's squash_action (bptt_optimizer.py:313-317) */
  const float *actions;        /* optional [S, N, u_dim] open-loop actions: the policy is skipped (`policy` may be zeroed) —
                                  rollout_actions (utils/optimizer_utils.py:11-59) and, with S=1, System.step itself */
  /* randomness: explicit tensors when non-NULL, else counter-based Philox4x32-10 keyed by (seed, offset) */
  const float *policy_noise;   /* [S, N, u_dim] standard normal */
  const float *model_noise;    /* [S, action_repeat, N, x_dim] standard normal */
  const int32_t *member_idx;   /* [S, action_repeat, N] in [0, E) for MBPO_ENS_TS1 */
  uint64_t seed, offset;
  const uint64_t *rng_dev;     /* optional device uint64[2] {seed word, step counter} added to (seed, offset): see "randomness" */
  /* env state, updated in place (brax State.obs / info['steps'] / done / info['first_obs']) */
  float *obs;             /* [N, x_dim] */
  const float *first_obs; /* [N, x_dim] */
  float *steps;           /* [N] (float flags, as in the reference) */
  float *done;            /* [N] */
  /* output: flattened Transition rows (brax UniformSamplingQueue ravel order)
   *   [obs(x), action(u), reward, discount, next_obs(x), {log_prob, raw_action(u)}, truncation] */
  float *transitions;     /* [S*N, row_len] */
  int32_t row_len;        /* 2x+u+3 (+1+u with ppo_extras) */
} mbpo_rollout_desc;

int mbpo_model_rollout(const mbpo_rollout_desc *d, void *stream);

/* ---- R1/R4-R7 for a USER-DEFINED System: the non-fused env step ------------------------------------
 * replaces: the same reference lines as mbpo_model_rollout, split at the reference's own plug-in seam — System.step
 *           (systems/base_systems.py:40-52 -> Dynamics.next_state, dynamics/base_dynamics.py:15-20 / Reward.__call__,
 *           rewards/base_rewards.py:15-21) stays the caller's code (a batched torch function on the device); per env step the
 *           host issues  mbpo_policy_act -> System.step (x action_repeat) -> mbpo_episode_step.
 * mbpo_policy_act   : make_inference_fn (sac_networks.py:58-73 / ppo_network.py:59-84): logits = MLP(normalize(obs)),
 *                     NormalTanh sample / mode (sac/parametric_distribution.py:66-124); raw_action [n,u] and log_prob [n] are
 *                     optional (PPO's policy extras).  Noise: explicit [n,u] or Philox(seed, offset [+ rng_dev], stream 1,
 *                     element elem_base + i*u + d) — with elem_base = s*N*u this IS the fused kernel's stream at step s.
 *                     workspace: n * (x_dim + 2*u_dim) floats.
 * mbpo_episode_step : EpisodeWrapper.step + AutoResetWrapper.step bookkeeping (brax_utils/training.py:91-137) and the
 *                     Transition row of actor_step (sac/acting.py:46-55) for step `step_index` of an unroll of `n_steps`;
 *                     `reward` is already summed over action_repeat, `sys_done` (optional) is SystemState.done.
 *                     Updates obs/steps/done in place, exactly like mbpo_model_rollout. */
int mbpo_policy_act(const mbpo_mlp_desc *policy, const float *obs, int64_t n, const float *norm_mean, const float *norm_std,
                    int32_t deterministic, float action_clip, const float *noise, uint64_t seed, uint64_t offset,
                    const uint64_t *rng_dev, uint64_t elem_base, float *action, float *raw_action, float *log_prob,
                    float *workspace, void *stream);

typedef struct mbpo_episode_step_desc {
  int32_t x_dim, u_dim;
  int64_t n_envs;
  int32_t episode_length, action_repeat;
  int32_t ppo_extras, env_major;   /* as in mbpo_rollout_desc */
  int32_t step_index, n_steps;     /* s and S: the row is s*N + i (or i*S + s with env_major) */
  const float *action;             /* [N, u] */
  const float *raw_action;         /* [N, u] (ppo_extras) */
  const float *log_prob;           /* [N]    (ppo_extras) */
  const float *reward;             /* [N] */
  const float *x_next;             /* [N, x] */
  const float *sys_done;           /* [N] or NULL (= 0) */
  const float *first_obs;          /* [N, x] */
  float *obs, *steps, *done;       /* env state, updated in place */
  float *transitions;              /* [S*N, row_len] */
  int32_t row_len;
} mbpo_episode_step_desc;

int mbpo_episode_step(const mbpo_episode_step_desc *d, void *stream);

/* ---- R9: replay buffer (brax UniformSamplingQueue semantics, INT32-exact) --------------------
 * replaces: replay_buffer.insert / .sample as called at sac/sac.py:303,318 and systems/brax_wrapper.py:29,
 *           bptt_optimizer.py:459,479 ([3P] brax.training.replay_buffers.UniformSamplingQueue; field use
 *           evidenced by bptt_optimizer.py:447-456).
 * The logical array data[max_size][row_len] of the reference is stored as a ring: logical row i lives at
 * physical row (i + head) % max_size, so the reference's jnp.roll on overflow is an O(1) head move.
 * state (device, int32[4]) = {insert_position, sample_position, head, total_inserted_low32}.
 */
int mbpo_replay_insert(float *data, int64_t max_size, int32_t row_len, int32_t *state, const float *rows,
                       int64_t n_rows, void *stream);
/* out[j] = data_logical[wrap(idx[j])]   (jnp.take(..., mode='wrap')); idx are LOGICAL indices */
int mbpo_replay_gather(const float *data, int64_t max_size, int32_t row_len, const int32_t *state, const int32_t *idx,
                       int64_t n, float *out, void *stream);
/* idx[j] = randint(sample_position, insert_position) from Philox(seed, offset, stream=REPLAY, j), then gather.
 * idx_out may be NULL; rng_dev (optional device uint64[2], see "randomness") is added to (seed, offset) at run time.
 * Fused sample+gather: sac/sac.py:318 (UniformSamplingQueue.sample). */
int mbpo_replay_sample(const float *data, int64_t max_size, int32_t row_len, const int32_t *state, uint64_t seed,
                       uint64_t offset, const uint64_t *rng_dev, int64_t n, int32_t *idx_out, float *out, void *stream);

/* perm = stable argsort of key_i = Philox(seed, offset [+ rng_dev], stream PERM, i).word0, i in [0, n): the ONE shared
 * permutation of PPO.sgd_step (ppo/ppo.py:166-171: jr.permutation with the same key for every leaf); gather whole
 * trajectories with mbpo_replay_gather(idx = perm).  workspace: n uint32 of scratch (keys for n > 16384; word 0 is an overflow flag of
 * the bucketed sort for 1024 < n <= 16384 — its initial value does not matter, it is left at 0).  n <= 2^20. */
int mbpo_philox_permutation(uint64_t seed, uint64_t offset, const uint64_t *rng_dev, int64_t n, int32_t *perm,
                            uint32_t *workspace, void *stream);

/* ---- R8: running_statistics.update ([3P] brax.training.acme.running_statistics; call sites
 * sac/sac.py:298-301, ppo/ppo.py:216-219), in the reference's own two-pass form, split so that a multi-GPU
 * host can all-reduce `sums` after each pass (the live form of `pmap_axis_name`'s two psums):
 *   pass 0: sums[0] = n, sums[1..x] = sum(d),  d = obs - mean_old  over rows[:, col_off : col_off+x_dim]
 *   pass 1: sums[1+x..1+2x) = sum(d * (d - upd)),  upd = sums[1..x] / (count + sums[0])
 *   apply : count += n; mean += sum_d/count; summed_variance += pass-1 sums;
 *           std = clip(sqrt(max(summed_variance,0)/count), std_min, std_max)   (brax: 1e-6, 1e6)
 * The same update IS BPTT's Normalizer.update (bptt_optimizer.py:52-67) with summed_variance = std^2 * size:
 *   sum (x - new_mean)^2 + size*(mean - new_mean)^2 == sum d*(d - upd); its floor is std_min = 1e-8, no ceiling.
 * stats (device, fp32) = [count, mean[x], summed_variance[x], std[x]];
 * workspace >= mbpo_running_stats_workspace_floats(x_dim) floats (one partial per workgroup, column and pass).
 */
int64_t mbpo_running_stats_workspace_floats(int32_t x_dim);
int mbpo_running_stats_reduce(const float *rows, int64_t n_rows, int32_t row_len, int32_t col_off, int32_t x_dim,
                              const float *stats, float *sums, float *workspace, int32_t pass, void *stream);
int mbpo_running_stats_apply(float *stats, const float *sums, int32_t x_dim, float std_min, float std_max, void *stream);
/* The same update for a SINGLE rank (nothing to all-reduce between the passes) in three launches instead of five: pass 0, then a
 * pass 1 whose workgroups add the first pass's partials up themselves, then one workgroup that sums the second pass and applies.
 * Bit-identical to reduce(pass 0) -> reduce(pass 1) -> apply; `sums` receives the same values. */
int mbpo_running_stats_update(const float *rows, int64_t n_rows, int32_t row_len, int32_t col_off, int32_t x_dim, float *stats,
                              float *sums, float *workspace, float std_min, float std_max, void *stream);

/* ---- P4: GAE (ppo/losses.py:128-184) and B2: lambda-return (utils/optimizer_utils.py:119-152) -------
 * Reverse first-order linear recurrences evaluated as wavefront-shuffle segmented scans.
 * Layout: time_major=1 -> arrays are [T,B] (the reference's layout after its swapaxes, losses.py:79);
 *         time_major=0 -> [B,T] (PPO's native data layout, ppo.py:210-213; no transpose needed).
 * gae:  vs, adv (both stop_gradient in the reference) from truncation, termination, rewards, values, bootstrap[B].
 * lambda_return: returns[t] = r_t + gamma*(1-lam)*v'_t + gamma*lam*returns[t+1], returns[T] = v'_{T-1}.
 */
int mbpo_gae_scan(const float *truncation, const float *termination, const float *rewards, const float *values,
                  const float *bootstrap, float *vs, float *advantages, int64_t B, int32_t T, float gamma,
                  float lam, int32_t time_major, void *stream);
int mbpo_lambda_return_scan(const float *rewards, const float *next_values, float *returns, int64_t B, int32_t T,
                            float gamma, float lam, int32_t time_major, void *stream);
/* N1: GAE with a per-element discount array (same layout as rewards) — the non_equidistant_time form of compute_gae
 * (ppo/losses_new.py:181-226): gamma is replaced element-wise in deltas, in the scan coefficient and in the advantages. */
int mbpo_gae_scan_discounts(const float *truncation, const float *termination, const float *rewards, const float *values,
                            const float *bootstrap, const float *discounts, float *vs, float *advantages, int64_t B,
                            int32_t T, float lam, int32_t time_major, void *stream);

/* ---- S3-S8: SAC sgd_step (sac/sac.py:227-281) -------------------------------------------------
 * replaces: SAC.sgd_step = alpha_update + critic_update + actor_update (all three evaluated at the OLD
 *           (alpha, policy, q) parameters, sac.py:234-258) + Polyak target update (sac.py:260-261);
 *           losses sac/losses.py:61-125; gradient_update_fn + optax chain(clip_by_global_norm, adamw)
 *           sac/utils.py:36-63, sac/sac.py:175-186.
 *
 * Flat train state (device, fp32), NP = P + 2*Q + 1 with P = policy params, Q = params of one critic:
 *   params   [NP] = [ policy | critic 0 | critic 1 | log_alpha ]
 *   target_q [2Q];  adam_m [NP];  adam_v [NP];  step_count [1] (optax's count, kept as a float: it only feeds Adam's bias
 *                   corrections 1 - b^count, which are exactly 1.0f long before a float stops counting at 2^24; it plays NO part
 *                   in the random streams)
 *   grads    [NP]   written by mbpo_sac_grads, read by mbpo_sac_apply (all-reduce it in between for N>1 ranks —
 *                   the live form of the reference's dead jax.lax.pmean, sac/utils.py:29-33)
 * Three entry points so that the multi-GPU exchange sits at the reference's pmean position:
 *   mbpo_sac_grads      : per-sample forward/backward of the three losses on one minibatch -> grads, metrics[0..2]
 *                         (critic_loss, actor_loss, alpha_loss); also bumps step_count (at the start of the forward/backward
 *                         launch) and leaves per-group
 *                         sum-of-squares partials of `grads` in the workspace.
 *   mbpo_sac_grad_norms : recompute those partials from `grads` (call after an all-reduce changed it).
 *   mbpo_sac_apply      : grads *= grad_scale; clip_by_global_norm per optimizer; AdamW; target <- (1-tau) target + tau q_new;
 *                         metrics[3] = exp(new log_alpha).
 * Noise: explicit standard-normal tensors [B,u] or NULL -> Philox(seed, offset [+ rng_dev]), streams 5/6/7: the caller
 *        advances `offset` (or the device counter) between sgd_steps.
 * Network shapes (sac.py:84-88 takes any tuple): policy and critics with hidden layers of ONE common width in {64, 128} whose 16-row
 *        tile fits the LDS run on the fused wave-chain kernel; every other shape (any hidden sizes, up to MBPO_MAX_LAYERS Dense layers)
 *        runs its forward/backward layer by layer — one GEMM launch per Dense layer — behind the same entry points, with the same
 *        results contract (mbpo_sac_step == mbpo_sac_grads + mbpo_sac_apply bit for bit).
 */
typedef struct mbpo_sac_desc {
  int32_t x_dim, u_dim;
  int32_t policy_layers;                     /* Dense layers of the policy: dims policy_dims[0..policy_layers] */
  int32_t policy_dims[MBPO_MAX_LAYERS + 1];  /* [x_dim, hidden..., 2*u_dim] */
  int32_t q_layers;
  int32_t q_dims[MBPO_MAX_LAYERS + 1];       /* [x_dim+u_dim, hidden..., 1] */
  int32_t policy_activation, q_activation;   /* MBPO_ACT_* */
  float *params, *target_q, *adam_m, *adam_v, *step_count, *grads;
  float *workspace;                          /* >= mbpo_sac_workspace_floats() floats */
  float *metrics;                            /* [4] critic_loss, actor_loss, alpha_loss, alpha */
  float *metrics_accum;                      /* optional [5]: running sums of the four metrics + step count, for the
                                                epoch means of sac/sac.py:360 without a host round trip per step */
  const float *batch;                        /* [batch_size, row_len] SAC transition rows (2x+u+3) */
  int32_t batch_size, row_len;
  const float *norm_mean, *norm_std;         /* [x_dim] or NULL */
  const float *noise_alpha, *noise_critic, *noise_actor; /* [batch_size, u_dim] or NULL */
  uint64_t seed, offset;
  const uint64_t *rng_dev;                   /* optional device uint64[2], see "randomness" */
  float discounting, reward_scaling, target_entropy, tau;
  float lr_policy, lr_q, lr_alpha, wd_policy, wd_q, wd_alpha, max_grad_norm;
  float grad_scale;                          /* 1/world_size when grads were all-reduced with SUM, else 1 */
  /* N1 (sac/losses.py:90-98): per-sample discount exp(-continuous_discounting * t), t = the switch time encoded in the last
   * action component, affinely mapped to [min,max]_time_between_switches and floored to a multiple of env_dt */
  int32_t non_equidistant_time;
  float continuous_discounting, min_time_between_switches, max_time_between_switches, env_dt;
} mbpo_sac_desc;

int64_t mbpo_sac_workspace_floats(const mbpo_sac_desc *d);
/* Offset (in floats) inside `workspace` of the 16-word control block the optimizer launches keep (uint32 words unless noted):
 * [0] two-launch steps issued, [1] two-launch steps whose clip check is resolved, [13] clip events = optimizer steps in which
 * clip_by_global_norm actually scaled some group (sac.py:218-225 'optax.clip_by_global_norm'), counted by every path
 * (mbpo_sac_apply, the next mbpo_sac_step, mbpo_sac_finalize).  The other words are private.  A host reads [13] between epochs
 * to choose between the two- and the three-launch step: a step that clips costs the two-launch path a second pass (see
 * INTEGRATION.md, "Gradient clipping").  Negative: error code. */
int64_t mbpo_sac_control_offset(const mbpo_sac_desc *d);
int mbpo_sac_grads(const mbpo_sac_desc *d, void *stream);
/* measurement hook: run only part of mbpo_sac_grads — phase_mask bit0 = forward/backward kernel (k_sac_fwd_bwd),
 * bit1 = slab reduce (k_sac_reduce).  mbpo_sac_grads == phase_mask 3.  Used by bench.py to time the dominant kernel. */
int mbpo_sac_grads_phase(const mbpo_sac_desc *d, int32_t phase_mask, void *stream);
int mbpo_sac_grad_norms(const mbpo_sac_desc *d, void *stream);
int mbpo_sac_apply(const mbpo_sac_desc *d, void *stream);
/* The default path — ONE sgd_step in TWO launches (mbpo_sac_grads + mbpo_sac_apply is three):
 *   mbpo_sac_step     : the forward/backward kernel, then ONE launch that reduces the per-tile slabs and applies the optimizer step
 *                       UNCLIPPED, saving the previous (params, adam_m, adam_v, target_q) in an undo log inside `workspace`.
 *                       clip_by_global_norm needs the global gradient norm, i.e. every block's partial: instead of a device-wide
 *                       meeting point inside the launch (measured slower than the kernel boundary it removes), the check is done
 *                       by the NEXT consumer of the parameters — the prologue of the next mbpo_sac_step / mbpo_sac_grads launch, or
 *                       mbpo_sac_finalize — which, if a group's norm reached max_grad_norm, recomputes that group's step from the
 *                       undo log with the clipped gradient: bit-identical to mbpo_sac_grads + mbpo_sac_apply in every case.
 *   mbpo_sac_finalize : resolves the pending check when no further mbpo_sac_step follows (end of training_step's scan of
 *                       sgd_steps, sac/sac.py:324; before anything else reads params / target_q / adam state).  Idempotent.
 * metrics[3] ('alpha') of a step becomes valid when its check has been resolved.  `workspace` must be zero before the first call. */
int mbpo_sac_step(const mbpo_sac_desc *d, void *stream);
int mbpo_sac_finalize(const mbpo_sac_desc *d, void *stream);
/* mbpo_sac_finalize and mbpo_rng_advance(rng_dev, inc) in ONE launch: the end of a training step (sac.py:306-327 — after the scan
 * of sgd_steps the step's randomness is used up and the device counter moves on). */
int mbpo_sac_finalize_advance(const mbpo_sac_desc *d, uint64_t *rng_dev, uint64_t inc, void *stream);

/* ---- P1-P3: PPO minibatch update (ppo/ppo.py:142-156, ppo/losses.py:56-126) -----------------------
 * replaces: PPO.minibatch_step = value_and_grad(PPOLoss.loss) + optax.adamw(lr, wd) over {policy, value} (ppo.py:128,139-140;
 *           no gradient clipping in this variant).  Inside the loss: policy logits and value baseline on [B,T] samples,
 *           bootstrap value, compute_gae (stop-gradient), advantage normalisation over the whole minibatch
 *           (losses.py:101-102), clipped surrogate, 0.5*MSE value loss, entropy bonus with a fresh NormalTanh sample.
 * Flat state: params [P+V] = [ policy | value ]; adam_m/adam_v [P+V]; step_count [1]; grads [P+V].
 *   mbpo_ppo_grads : grads, metrics[0..3] = total_loss, policy_loss, v_loss, entropy_loss (losses.py:121-126); bumps step_count
 *   mbpo_ppo_apply : grads *= grad_scale; AdamW.   All-reduce `grads` in between for N>1 ranks (ppo.py:149-154's pmean).
 * data: one minibatch [batch_size, unroll_length, row_len] of PPO rows (row_len = 2x+2u+4), i.e. the rollout kernel's
 *       env_major output after the permutation gather.  entropy_noise [B,T,u] or NULL -> Philox(seed, offset [+ rng_dev], stream 9).
 * Network shapes (ppo.py:60-63 takes any tuple): as for SAC — one common hidden width in {64, 128} -> fused kernels, anything else
 *       (e.g. experiments/train_inverted_pendulum/exp_ppo.py's 256x5 critic) -> layer by layer behind the same entry points.
 */
typedef struct mbpo_ppo_desc {
  int32_t x_dim, u_dim;
  int32_t policy_layers;
  int32_t policy_dims[MBPO_MAX_LAYERS + 1];  /* [x_dim, hidden..., 2*u_dim] */
  int32_t value_layers;
  int32_t value_dims[MBPO_MAX_LAYERS + 1];   /* [x_dim, hidden..., 1] */
  int32_t policy_activation, value_activation;
  float *params, *adam_m, *adam_v, *step_count, *grads;
  float *workspace;                          /* >= mbpo_ppo_workspace_floats() floats */
  float *metrics;                            /* [4] */
  float *metrics_accum;                      /* optional [5]: running sums of the four metrics + count */
  const float *data;
  int32_t batch_size, unroll_length, row_len;
  const float *norm_mean, *norm_std;         /* [x_dim] or NULL */
  const float *entropy_noise;                /* [B,T,u] or NULL */
  uint64_t seed, offset;
  const uint64_t *rng_dev;                   /* optional device uint64[2], see "randomness" */
  float entropy_cost, discounting, reward_scaling, gae_lambda, clipping_epsilon;
  int32_t normalize_advantage;
  float lr, wd, grad_scale;
} mbpo_ppo_desc;

int64_t mbpo_ppo_workspace_floats(const mbpo_ppo_desc *d);
int mbpo_ppo_grads(const mbpo_ppo_desc *d, void *stream);
int mbpo_ppo_apply(const mbpo_ppo_desc *d, void *stream);
/* mbpo_ppo_grads + mbpo_ppo_apply with no seam for a collective between them (a single rank): the launch that sums the gradient
 * also applies AdamW to it — the same bits, one launch less per minibatch_step (ppo/ppo.py:142-156). */
int mbpo_ppo_step(const mbpo_ppo_desc *d, void *stream);

/* ---- B1-B5: BPTT actor gradient (bptt_optimizer.py:327-378) -----------------------------------------
 * replaces: value_and_grad(vmap(actor_loss)) of BPTTOptimizer._train_step, i.e. rollout_policy with stop_grads=True
 *           (utils/optimizer_utils.py:62-116) through System.step, target-critic min(v1,v2) on the normalised next states,
 *           lambda_return (:119-152), discount cumprod, Actor.get_log_prob (:144-152), and the backward pass through ALL of
 *           it with respect to the actor parameters (gradient flows a -> x' -> ... through the model and the target
 *           critic; the policy's own observation input is stop-gradiented when acting but NOT inside get_log_prob).
 * One launch: forward rollout (checkpoints x_t, a_t, eps_t in `workspace`), lambda-returns, reverse sweep with per-step
 * recomputation of the activations in LDS.  Outputs: actor `grads` [P]; the simulated transitions
 * [n*horizon, 2x+u+2] = [obs, action, reward(raw), discount=1, next_obs] (what :478-479 inserts into the sampling buffer);
 * lambda_values [n*horizon] (the critic targets of :385-419); metrics = {actor_loss, entropy_loss}.
 * Log-prob for u_dim > 1 uses sum_A logN - sum_A log(1-a^2) per step (identical to the reference at u_dim = 1).
 */
typedef struct mbpo_bptt_desc {
  int32_t x_dim, u_dim, horizon;
  int32_t actor_layers;
  int32_t actor_dims[MBPO_MAX_LAYERS + 1];   /* [x_dim, features..., 2*u_dim] */
  int32_t critic_layers;
  int32_t critic_dims[MBPO_MAX_LAYERS + 1];  /* [x_dim, features..., 1] */
  int32_t actor_activation, critic_activation;
  float init_stddev;                         /* Actor.init_stddev (:127) */
  const float *actor_params;                 /* [P] */
  const float *target_critic_params;         /* [2*C] = [critic_1 | critic_2] */
  int32_t system_kind;                       /* MBPO_SYS_* */
  mbpo_mlp_desc dynamics;                    /* ensemble (MBPO_ENS_MEAN semantics), ignored for MBPO_SYS_PENDULUM */
  int32_t ens_predict_delta;
  int32_t reward_kind;                       /* MBPO_REWARD_* */
  const float *reward_params, *sys_params;
  const float *state_mean, *state_std;       /* [x_dim] state normaliser */
  const float *reward_mean_std;              /* [2] reward normaliser {mean, std} */
  const float *init_states;                  /* [n, x_dim] */
  int64_t n;
  const float *act_noise;                    /* [n, horizon, u_dim] or NULL -> Philox(seed, offset [+ rng_dev], stream 1) */
  uint64_t seed, offset;
  const uint64_t *rng_dev;                   /* optional device uint64[2], see "randomness" */
  float discount, lambda_, ent_coef;
  float *transitions, *lambda_values, *grads, *metrics;
  float *workspace;                          /* >= mbpo_bptt_workspace_floats() floats */
} mbpo_bptt_desc;

int64_t mbpo_bptt_workspace_floats(const mbpo_bptt_desc *d);
int mbpo_bptt_actor_grads(const mbpo_bptt_desc *d, void *stream);

/* ---- B4: twin-V critic regression (bptt_optimizer.py:385-419) ----------------------------------------
 * replaces: value_and_grad(critic_loss_fn) of update_critic: loss = 0.5*(mean l2(v1, lamb) + mean l2(v2, lamb)), l2 = 0.5(.)^2,
 *           on a minibatch gathered (with replacement) from the flattened simulated transitions:
 *           obs_j = transitions[idx[j], 0:x_dim] (normalised with the state normaliser), target_j = lambda_values[idx[j]].
 * grads [2*C] in the [critic_1 | critic_2] layout; metrics[0] = critic loss.  workspace >= mbpo_critic_workspace_floats().
 */
int64_t mbpo_critic_workspace_floats(int32_t x_dim, int32_t critic_layers, const int32_t *critic_dims, int64_t batch);
int mbpo_critic_grads(const float *critic_params, int32_t x_dim, int32_t critic_layers, const int32_t *critic_dims,
                      int32_t activation, const float *transitions, int32_t row_len, const float *lambda_values,
                      const int32_t *idx, int64_t batch, const float *state_mean, const float *state_std, float *grads,
                      float *metrics, float *workspace, void *stream);

/* ---- B1/B3 around a USER-DEFINED System: vector-Jacobian product of one or two 64-wide MLPs ------------------------
 * replaces: the reverse-mode pieces of value_and_grad(vmap(actor_loss)) (bptt_optimizer.py:361-372) that are NOT the user's
 *           System.step when the horizon loop of rollout_policy (utils/optimizer_utils.py:62-116) has to run on the host:
 *           the actor MLP (weights only: act(stop_gradient(obs)), :85-86) and the twin target critics on the next state
 *           (input only: bptt_optimizer.py:342-343).  MLP semantics: utils/network_utils.py:5-17.
 * For every net k < n_nets (1 or 2; shared input):  y_k = MLP_k((x - norm_mean) / norm_std)   [norm_* NULL: y_k = MLP_k(x)]
 *   dx[k][j][:] = d <dy[k][j], y_k[j]> / d x[j]        (optional, [n_nets][n][dims[0]], with respect to the RAW x)
 *   dw[k][:]    = sum_j d <dy[k][j], y_k[j]> / d params_k   (optional, [n_nets][params per net]; needs net_stride == params per net)
 *   y[k][j][:]  = the recomputed outputs (optional, [n_nets][n][dims[n_layers]])
 * dy: [n_nets][n][dims[n_layers]].  Hidden layers 64 wide, input / output width <= 32.  workspace (only for dw)
 * >= mbpo_mlp_vjp_workspace_floats(mlp, n) floats.  Fixed-order reductions: bit-reproducible.
 */
int64_t mbpo_mlp_vjp_workspace_floats(const mbpo_mlp_desc *mlp, int64_t n);
int mbpo_mlp_vjp(const mbpo_mlp_desc *mlp, const float *x, int64_t n, const float *norm_mean, const float *norm_std,
                 const float *dy, float *y, float *dx, float *dw, float *workspace, void *stream);

/* The same contract for ANY hidden sizes (bptt_optimizer.py:183-186 `actor_features` / `critic_features` accept any tuple; the fused
 * kernel above is built for 64-wide hidden layers): one fp32-MFMA GEMM launch per Dense layer over all n rows (csrc/layered.hip),
 * forward recomputed with its pre-activations kept in `workspace`, then input and weight gradients layer by layer.
 *   x: [n][dims[0]], ALREADY normalised (shared by the nets);  dy: [n_nets][n][dims[n_layers]] or NULL (forward only -> y);
 *   y (optional with dy): [n_nets][n][dims[n_layers]];  dx (optional): [n_nets][n][dims[0]] with respect to the x given;
 *   dw (optional): [n_nets][params per net].  workspace >= mbpo_mlp_layered_workspace_floats(mlp, n) floats.  Fixed-order sums. */
int64_t mbpo_mlp_layered_workspace_floats(const mbpo_mlp_desc *mlp, int64_t n);
int mbpo_mlp_layered_vjp(const mbpo_mlp_desc *mlp, const float *x, int64_t n, const float *dy, float *y, float *dx, float *dw,
                         float *workspace, void *stream);

/* ---- generic optimizer step: [optax.apply_if_finite(] optax.adamw(lr, wd) [)] + optional Polyak target ----------------
 * replaces: actor_optimizer.update/apply_updates (bptt_optimizer.py:218-225, 374-378), critic_optimizer + soft_update
 *           (:406-410; utils/optimizer_utils.py:155-161).
 * grads *= grad_scale; if apply_if_finite and any gradient is non-finite the whole update (params, moments, count) is
 * skipped; count (device scalar) advances by one otherwise; grad_norm_out (optional) = optax.global_norm(grads);
 * target (optional) <- (1 - tau) * target + tau * new_params.   workspace >= 2 * ceil(n / 256) + 4 floats.
 */
int mbpo_adamw_step(float *params, const float *grads, float *adam_m, float *adam_v, float *step_count, int64_t n, float lr,
                    float wd, float grad_scale, int32_t apply_if_finite, float *target, float tau, float *grad_norm_out,
                    float *workspace, void *stream);

/* soft_update alone (utils/optimizer_utils.py:155-161): out = (1 - tau) * target + tau * online; out may alias target. */
int mbpo_soft_update(const float *target, const float *online, float *out, int64_t n, float tau, void *stream);

/* ---- N3: ensemble model learning (the step before the rollout path) ------------------------------------
 * Not in the reference (its model would come from the external `bsm` package, setup.py:22).  Gradient of every member's
 * Gaussian negative log-likelihood on its own minibatch of true transitions, with exactly the parameterisation the rollout
 * kernel consumes: (mu, raw) = MLP_e([x, u]); mean = mu (+ x if predict_delta); sigma = softplus(raw) + min_std;
 *   loss_e = mean_b sum_d [ 0.5 ((x'_d - mean_d)/sigma_d)^2 + log sigma_d ].
 * rows: [R, row_len] transitions (x at column 0, u at x_dim, next_obs at next_obs_off); idx: [E, batch] row indices (one
 * bootstrapped minibatch per member); grads [E * n_params] in the dynamics' flat layout; metrics [E] = loss_e.
 * Apply with mbpo_adamw_step on the whole flat vector (members are independent, AdamW is elementwise).  Hidden width 64. */
typedef struct mbpo_ens_train_desc {
  int32_t x_dim, u_dim;
  mbpo_mlp_desc dynamics;
  const float *rows;
  int32_t row_len, next_obs_off;
  const int32_t *idx;
  int64_t batch;
  int32_t predict_delta;
  float min_std;
  float *grads, *metrics, *workspace;
} mbpo_ens_train_desc;

int64_t mbpo_ens_nll_workspace_floats(const mbpo_ens_train_desc *d);
int mbpo_ens_nll_grads(const mbpo_ens_train_desc *d, void *stream);

/* ---- N4: iCEM trajectory optimizer, device side (trajectory_optimizers/icem_optimizer.py:135-252) ---------------------
 * One iteration = mbpo_icem_sample -> mbpo_model_rollout(actions = the sampled sequences) -> mbpo_icem_update.
 * mbpo_icem_sample: coloured noise (utils/general_utils.py:81-208, powerlaw_psd_gaussian, as a direct inverse real DFT; Philox
 *   stream ICEM) per (sample, action dim) series of length horizon; candidate = clip(mean + noise*std, u_min, u_max) (:186-187);
 *   the n_prev previous elites are appended (:190); every candidate is replicated over n_particles envs:
 *   actions [horizon][(n_samples+n_prev)*n_particles][u_dim] (env = candidate*n_particles + particle),
 *   candidates [(n_samples+n_prev)][horizon][u_dim].  u_min/u_max are [u_dim] device vectors.
 * mbpo_icem_update: values[c] = mean (use_max: max) over particles of mean_t reward (:146-163) from the rollout's step-major
 *   transition rows; elites = the n_elites best in np.argsort order; mean <- alpha*mean + (1-alpha)*elite mean, std likewise on
 *   the population variance (:199-209); best_value/best_sequence keep the best elite seen (:212-221); the n_prev best elites
 *   go to prev_elites (:227).  workspace: n_candidates int32.  State vectors are [horizon*u_dim] device floats. */
int mbpo_icem_sample(const float *mean, const float *std, const float *prev_elites, const float *u_min, const float *u_max,
                     int32_t n_samples, int32_t n_prev, int32_t horizon, int32_t u_dim, int32_t n_particles, float exponent,
                     uint64_t seed, uint64_t offset, const uint64_t *rng_dev, float *actions, float *candidates, void *stream);
int mbpo_icem_update(const float *rows, int32_t row_len, int32_t reward_col, int32_t n_candidates, int32_t n_particles,
                     int32_t horizon, int32_t u_dim, const float *candidates, int32_t n_elites, int32_t n_prev, float alpha,
                     int32_t use_max, float *mean, float *std, float *best_value, float *best_sequence, float *prev_elites,
                     float *values, int32_t *workspace, void *stream);
/* The same with the reference's constraint term (icem_optimizer.py:99,157-166): particle_cost[c * n_particles + p] = the user's
 * cost_fn on the trajectory of candidate c, particle p (evaluated by the host between the rollout and this launch — a Python
 * callable cannot run in a kernel); objective[c] = reward[c] - lambda_constraint * relu(cost[c]), cost[c] = mean (cost_use_max:
 * max, `use_pessimism`) over the particles.  particle_cost NULL: mbpo_icem_update. */
int mbpo_icem_update_constrained(const float *rows, int32_t row_len, int32_t reward_col, int32_t n_candidates, int32_t n_particles,
                                 int32_t horizon, int32_t u_dim, const float *candidates, int32_t n_elites, int32_t n_prev, float alpha,
                                 int32_t use_max, const float *particle_cost, float lambda_constraint, int32_t cost_use_max, float *mean,
                                 float *std, float *best_value, float *best_sequence, float *prev_elites, float *values,
                                 int32_t *workspace, void *stream);

/* ---- one-shot all-reduce over xGMI peer memory (multi-GPU SAC gradient exchange) ------------------------
 * replaces: the live form of the reference's jax.lax.pmean(grad) (sac/utils.py:29-33) for vectors small enough that a
 *           collective is pure latency.  Every rank owns an exchange REGION (mbpo_p2p_alloc -> 64-byte IPC handle, passed to
 *           the other ranks out of band, mbpo_p2p_open there); a producer kernel stores its vector into slot[rank] of every
 *           rank's region and publishes per-block arrival counts; the consumer waits for all ranks (bounded spin) and adds the
 *           slots in rank order, so every rank holds bit-identical sums.  Regions are zero-initialised by mbpo_p2p_alloc.
 * regions[r] : base of rank r's region as mapped in THIS process (regions[rank] is the own allocation);
 * n_max      : floats per slot the region was sized for (mbpo_p2p_region_bytes).
 * mbpo_p2p_all_reduce_sum: buf[0..n) <- sum over ranks, in place (3 small launches; every rank must call it the same number
 *   of times).  On a consumer timeout buf is filled with NaN and mbpo_p2p_status reports 1 — nothing hangs.
 */
#define MBPO_P2P_MAX_RANKS 16
typedef struct mbpo_p2p_desc {
  int32_t world, rank;
  int64_t n_max;
  void *regions[MBPO_P2P_MAX_RANKS];
} mbpo_p2p_desc;

int64_t mbpo_p2p_region_bytes(int32_t world, int64_t n_max);
int mbpo_p2p_alloc(int64_t bytes, void **ptr, void *handle64);
int mbpo_p2p_open(const void *handle64, int32_t peer_device, void **ptr);   /* peer_device < 0: same device */
int mbpo_p2p_close(void *ptr);
int mbpo_p2p_free(void *ptr);
int mbpo_p2p_all_reduce_sum(const mbpo_p2p_desc *d, float *buf, int64_t n, void *stream);
/* SAC sgd_step over N ranks without a collective launch:  mbpo_sac_grads_p2p (= mbpo_sac_grads whose reduction kernel also
 * stores the rank's gradient into every rank's exchange region)  ->  mbpo_sac_gather_p2p (waits for all ranks, grads <- sum
 * over ranks in rank order, clip-norm partials)  ->  mbpo_sac_apply (grad_scale = 1/N).  n_max >= NP. */
int mbpo_sac_grads_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream);
int mbpo_sac_gather_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream);
/* The same exchange in ONE reduction launch (= mbpo_sac_grads_p2p + mbpo_sac_gather_p2p): the reduction kernel stores the rank's
 * gradient into every region, waits inside the kernel for every rank's arrivals and leaves grads = sum over ranks and the
 * clip-norm partials.  Then mbpo_sac_apply (grad_scale = 1/N).  Needs the reduction's workgroups (NP/256) co-resident. */
int mbpo_sac_grads_exchange_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream);
/* mbpo_sac_step over N ranks: the second launch also exchanges (stores the rank's gradient into every region, waits for every rank,
 * sums the world slots in rank order) before the unclipped optimizer step (grad_scale = 1/N); mbpo_sac_finalize as above. */
int mbpo_sac_step_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream);
int mbpo_p2p_status(const mbpo_p2p_desc *d, int32_t *status_out);

#ifdef __cplusplus
}
#endif
#endif /* MBPO_HIP_H */

// generic_env.hip — the NON-fused env step for a user-defined System (the reference's plug-in seam, base_systems.py:40-52,
// base_dynamics.py:15-20).  The fused rollout kernel (rollout.hip) contains the policy, the system and the wrapper stack in one
// launch, which only works for systems that exist as device code (Pendulum, the learned ensemble).  For any other System the
// host walks the env steps and calls, per step:
//     mbpo_policy_act      policy inference: normalise -> MLP (k_ensemble_forward) -> NormalTanh head          [HIP, this file]
//     System.step          the user's batched torch code                                                          [user]
//     mbpo_episode_step    Episode / AutoReset bookkeeping + Transition row assembly                            [HIP, this file]
// with the SAME random stream as the fused kernel (Philox element index (s*N + env)*U + d of stream POLICY_NOISE), so a System
// that exists in both forms produces the same rows either way (tests/test_gpu_generic_system.py).
#include "common.hpp"

#define LOG_SQRT_2PI_G 0.91893853320467274178f
#define LOG_2_G 0.69314718055994530942f

__global__ void __launch_bounds__(256) k_normalize_rows(const float *x, long long n, int d, const float *mean, const float *std, float *out) {
  const long long total = n * d;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % d);
    out[i] = mean ? (x[i] - mean[c]) / std[c] : x[i];      // running_statistics.normalize: (batch - mean) / std
  }
}

struct HeadArgs {
  const float *logits;   // [n, 2u]
  long long n;
  int U, deterministic;
  float action_clip;
  const float *noise;    // [n, u] or NULL
  unsigned long long seed, offset, elem_base;
  const unsigned long long *rng_dev;
  float *action, *raw_action, *log_prob;
};

// NormalTanhDistribution (sac/parametric_distribution.py:66-124) exactly as the fused rollout kernel evaluates it:
//   sigma = softplus(raw) + 0.001; z = loc + sigma*eps; a = tanh(z); log_prob = sum_d [logN(z) - 2(log2 - z - softplus(-2z))]
__global__ void __launch_bounds__(256) k_policy_head(HeadArgs A) {
  const RngKey rk = rng_resolve(A.seed, A.offset, A.rng_dev);
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n) return;
  const int U = A.U;
  float lp = 0.f;
  for (int d = 0; d < U; ++d) {
    const float loc = A.logits[i * 2 * U + d], raw = A.logits[i * 2 * U + U + d];
    float z, a;
    if (A.deterministic) {
      z = loc;                                                         // mode: tanh(loc)   (:121-124)
    } else {
      const long long nidx = i * U + d;
      const float eps = A.noise ? A.noise[nidx] : philox_normal(rk.seed, rk.offset, MBPO_STREAM_POLICY_NOISE, A.elem_base + (unsigned long long)nidx);
      const float sigma = softplus_f(raw) + 0.001f;
      z = loc + sigma * eps;
      const float ldj = 2.0f * (LOG_2_G - z - softplus_f(-2.0f * z));
      lp += -0.5f * eps * eps - logf(sigma) - LOG_SQRT_2PI_G - ldj;
    }
    a = tanhf(z);
    if (A.action_clip > 0.f) a = fminf(fmaxf(a, -A.action_clip), A.action_clip);
    A.action[i * U + d] = a;
    if (A.raw_action) A.raw_action[i * U + d] = z;
  }
  if (A.log_prob) A.log_prob[i] = lp;
}

extern "C" int mbpo_policy_act(const mbpo_mlp_desc *policy, const float *obs, int64_t n, const float *norm_mean, const float *norm_std,
                               int32_t deterministic, float action_clip, const float *noise, uint64_t seed, uint64_t offset,
                               const uint64_t *rng_dev, uint64_t elem_base, float *action, float *raw_action, float *log_prob,
                               float *workspace, void *stream) {
  MBPO_REQUIRE(policy && obs && action && workspace, MBPO_ERR_ARG, "policy_act: null pointer");
  MBPO_REQUIRE(n >= 0, MBPO_ERR_ARG, "policy_act: negative n");
  MBPO_REQUIRE((norm_mean == nullptr) == (norm_std == nullptr), MBPO_ERR_ARG, "policy_act: norm_mean/norm_std mismatch");
  MBPO_REQUIRE(policy->n_nets == 1 && policy->n_layers >= 1 && policy->n_layers <= MBPO_MAX_LAYERS, MBPO_ERR_ARG, "policy_act: bad policy descriptor");
  const int X = policy->dims[0], U2 = policy->dims[policy->n_layers];
  MBPO_REQUIRE(U2 > 0 && U2 % 2 == 0, MBPO_ERR_ARG, "policy_act: the policy must end in 2*u_dim logits");
  if (n == 0) return MBPO_OK;
  hipStream_t st = (hipStream_t)stream;
  float *xn = workspace, *logits = workspace + n * X;
  const long long total = (long long)n * X;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_normalize_rows, dim3(grid), dim3(256), 0, st, obs, (long long)n, X, norm_mean, norm_std, xn);
  int rc = mbpo_ensemble_mlp_forward(policy, xn, 1, logits, n, stream);
  if (rc != MBPO_OK) return rc;
  HeadArgs A{logits, (long long)n, U2 / 2, deterministic, action_clip, noise, (unsigned long long)seed, (unsigned long long)offset,
             (unsigned long long)elem_base, (const unsigned long long *)rng_dev, action, raw_action, log_prob};
  hipLaunchKernelGGL(k_policy_head, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A);
  MBPO_CHECK_LAUNCH("policy_act");
  return MBPO_OK;
}

struct EpisodeArgs {
  int X, U, D;
  long long N;
  int episode_length, action_repeat, ppo_extras, env_major, s, S;
  const float *action, *raw_action, *log_prob, *reward, *x_next, *sys_done, *first_obs;
  float *obs, *steps, *done, *rows;
};

// AutoResetWrapper.step(EpisodeWrapper.step(...)) around an ALREADY evaluated System.step (brax_utils/training.py:91-137) and the
// Transition of actor_step (sac/acting.py:46-55); one thread per env.
__global__ void __launch_bounds__(256) k_episode_step(EpisodeArgs A) {
  const long long env = (long long)blockIdx.x * 256 + threadIdx.x;
  if (env >= A.N) return;
  const int X = A.X, U = A.U, D = A.D;
  float steps = A.done[env] != 0.f ? 0.f : A.steps[env];          // AutoReset.step :120-124
  steps += (float)A.action_repeat;                                // Episode.step :98
  const float sd = A.sys_done ? A.sys_done[env] : 0.f;            // SystemState.done (base_systems.py:25)
  const bool over = steps >= (float)A.episode_length;
  const float done = over ? 1.f : sd;                             // :102
  const float trunc = over ? 1.f - sd : 0.f;                      // :103-105
  const long long row = A.env_major ? env * A.S + A.s : (long long)A.s * A.N + env;
  float *r = A.rows + row * D;
  for (int c = 0; c < X; ++c) {
    r[c] = A.obs[env * X + c];                                    // observation = env_state.obs
    const float nx = done != 0.f ? A.first_obs[env * X + c] : A.x_next[env * X + c];   // AutoReset :126-137
    r[X + U + 2 + c] = nx;
    A.obs[env * X + c] = nx;
  }
  for (int d = 0; d < U; ++d) r[X + d] = A.action[env * U + d];
  r[X + U] = A.reward[env];
  r[X + U + 1] = 1.f - done;                                      // discount = 1 - nstate.done
  if (A.ppo_extras) {
    r[2 * X + U + 2] = A.log_prob[env];
    for (int d = 0; d < U; ++d) r[2 * X + U + 3 + d] = A.raw_action[env * U + d];
  }
  r[D - 1] = trunc;
  A.steps[env] = steps;
  A.done[env] = done;
}

extern "C" int mbpo_episode_step(const mbpo_episode_step_desc *d, void *stream) {
  MBPO_REQUIRE(d, MBPO_ERR_ARG, "episode_step: null descriptor");
  MBPO_REQUIRE(d->x_dim > 0 && d->u_dim > 0 && d->n_envs >= 0, MBPO_ERR_ARG, "episode_step: bad sizes");
  MBPO_REQUIRE(d->episode_length > 0 && d->action_repeat > 0, MBPO_ERR_ARG, "episode_step: episode_length and action_repeat must be positive");
  MBPO_REQUIRE(d->step_index >= 0 && d->step_index < d->n_steps, MBPO_ERR_ARG, "episode_step: step_index %d outside [0, %d)", d->step_index, d->n_steps);
  const int want = 2 * d->x_dim + d->u_dim + 3 + (d->ppo_extras ? 1 + d->u_dim : 0);
  MBPO_REQUIRE(d->row_len == want, MBPO_ERR_ARG, "episode_step: row_len %d != expected %d", d->row_len, want);
  if (d->n_envs == 0) return MBPO_OK;
  MBPO_REQUIRE(d->action && d->reward && d->x_next && d->first_obs && d->obs && d->steps && d->done && d->transitions, MBPO_ERR_ARG,
               "episode_step: null pointer");
  MBPO_REQUIRE(!d->ppo_extras || (d->raw_action && d->log_prob), MBPO_ERR_ARG, "episode_step: ppo_extras needs raw_action and log_prob");
  EpisodeArgs A{d->x_dim, d->u_dim, d->row_len, (long long)d->n_envs, d->episode_length, d->action_repeat, d->ppo_extras, d->env_major,
                d->step_index, d->n_steps, d->action, d->raw_action, d->log_prob, d->reward, d->x_next, d->sys_done, d->first_obs,
                d->obs, d->steps, d->done, d->transitions};
  hipLaunchKernelGGL(k_episode_step, dim3((unsigned)((d->n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
  MBPO_CHECK_LAUNCH("episode_step");
  return MBPO_OK;
}

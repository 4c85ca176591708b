// rollout_shared.hpp — what the rollout kernels share (csrc/rollout.hip: the generic kernels; csrc/rollout_lean.hip: the kernel specialised
// for the benchmark networks): the argument block, the analytic Pendulum step / reward, the hardware-transcendental helpers.
#pragma once
#include "common.hpp"

struct RolloutArgs {
  MlpDev policy, dyn;
  int x_dim, u_dim;
  long long n_envs;
  int n_steps, episode_length, action_repeat;
  int system_kind, ens_mode, ens_predict_delta, ens_sample_noise;
  float ens_min_std;
  int reward_kind;
  const float *reward_params, *sys_params, *norm_mean, *norm_std;
  int deterministic, ppo_extras, env_major;
  float action_clip;
  const float *actions;
  const float *policy_noise, *model_noise;
  const int *member_idx;
  unsigned long long seed, offset;
  const unsigned long long *rng_dev;
  float *obs;
  const float *first_obs;
  float *steps, *done;
  float *transitions;
  int row_len;
  // LDS geometry
  int ld_x, ld_xu, ld_h, ld_y, n_chains, n_out;
};

// PendulumDynamics.next_state (dynamics/pendulum_dynamics.py:29-63), fp32, same operation order.
__device__ __forceinline__ void pendulum_step(const float *x, float u, const float *sp, float *xn) {
  const float max_speed = sp[0], max_torque = sp[1], dt = sp[2], g = sp[3], mm = sp[4], l = sp[5];
  const float th = atan2f(x[1], x[0]);
  const float thdot = x[2];
  const float uc = fminf(fmaxf(u, -1.0f), 1.0f) * max_torque;
  const float thdd = (3.0f * g) / (2.0f * l) * sinf(th) + 3.0f / (mm * (l * l)) * uc;
  float nthdot = thdot + thdd * dt;
  nthdot = fminf(fmaxf(nthdot, -max_speed), max_speed);
  const float nth = th + nthdot * dt;  // dx[0] = clipped newthdot (ode :61-63)
  float nthdot2 = thdot + thdd * dt;   // dx[-1] = newthddot (:41)
  nthdot2 = fminf(fmaxf(nthdot2, -max_speed), max_speed);
  xn[0] = cosf(nth);
  xn[1] = sinf(nth);
  xn[2] = nthdot2;
}

// PendulumReward.__call__ (rewards/pendulum_reward.py:32-41): uses pre-step x and the unclipped action.
__device__ __forceinline__ float pendulum_reward(const float *x, float u, const float *rp) {
  const float angle_cost = rp[0], control_cost = rp[1], target = rp[2];
  const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
  const float theta = atan2f(x[1], x[0]), omega = x[2];
  float d = theta - target;
  float t = d + PI_F;
  float mpy = fmodf(t, TWO_PI_F);  // python/jnp % : result takes the sign of the divisor
  if (mpy < 0.0f) mpy += TWO_PI_F;
  d = mpy - PI_F;
  return -(angle_cost * (d * d) + 0.1f * (omega * omega)) - control_cost * (u * u);
}

__device__ __forceinline__ float ro_fexp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
__device__ __forceinline__ float ro_flog(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float ro_fsoftplus(float x) { return fmaxf(x, 0.0f) + ro_flog(1.0f + ro_fexp(-fabsf(x))); }
__device__ __forceinline__ float ro_ftanh(float x) {
  const float e = ro_fexp(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
  return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}


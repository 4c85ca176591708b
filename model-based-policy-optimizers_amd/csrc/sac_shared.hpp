// sac_shared.hpp — device helpers shared by the SAC forward/backward kernels (sac.hip: the generic chain-table kernel,
// sac_lean.hip: the kernel specialised for the benchmark networks) and the optimizer launches of sac.hip.
#pragma once
#include "common.hpp"

#define LOG_SQRT_2PI 0.91893853320467274178f
#define LOG_2 0.69314718055994530942f

// Flat optimizer state + what the clip check needs (shared by k_sac_apply, k_sac_reduce_apply, the fix-up and k_sac_finalize).
struct SacOptArgs {
  float *params, *target_q, *adam_m, *adam_v, *grads, *metrics, *metrics_accum;
  float *undo;                  // [3*NP + Q2]: params | adam_m | adam_v | target_q BEFORE the last speculative step
  const float *step_count, *ss_part;
  unsigned int *seq;            // [0] speculative steps issued, [1] ... resolved by k_sac_finalize; then, in the same 8 dwords (ONE scalar
                                // load in k_sac_fwd_bwd): [2..4] and [5..7] two slots of per-group sums of squares formed with float
                                // atomics by k_sac_reduce_apply (order not fixed: a QUICK, conservative clip test only), step k adds
                                // to slot k & 1
  unsigned int *slot_word;      // the slot k & 1 of the step in flight, published by its fwd/bwd launch for its reduce launch
  float *undo_count;            // [0] optax count of the last speculative step (step_count itself is bumped again by the next fwd/bwd
                                //     launch), [1] metrics_accum[3] before that step added its 'alpha', [2..3] its Adam bias corrections
  int n_parts, P, Q2;
  float lr[3], wd[3];
  float max_norm, tau, one_minus_tau, grad_scale;
};

// Control block (16 words in the workspace, mbpo_sac_control_offset): [0] speculative steps issued, [1] resolved, [2..7] quick sums,
// [8] slot word, [9..12] undo_count, [13] clip events = optimizer steps in which some group's gradient was clipped (every path).
#define SAC_CTL_CLIP_EVENTS 13

struct AdamOut {
  float p, m, v;
};
// [3P optax.adamw] scale_by_adam(b1=.9,b2=.999,eps=1e-8) -> add_decayed_weights(wd) -> scale(-lr); optax forms (1 - decay) in Python
// double and only then casts: f32(0.1), f32(0.001) — not 1.f - 0.999f.  corr0/corr1 = 1 - b^count.
__device__ __forceinline__ AdamOut sac_adam(float p, float m, float v, float g, float corr0, float corr1, float lr, float wd) {
  AdamOut o;
  o.m = 0.9f * m + 0.1f * g;
  o.v = 0.999f * v + 0.001f * (g * g);
  const float mu_hat = o.m / corr0;
  const float nu_hat = o.v / corr1;
  float upd = mu_hat / (sqrtf(nu_hat) + 1e-8f);
  upd = upd + wd * p;
  o.p = p + (-lr) * upd;  // optax.apply_updates: p + u, u = -lr * upd
  return o;
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Waves 0..2 of the calling workgroup: gnorm[w] = sqrt(sum of optimizer group w's partials) * grad_scale — the fixed order of
// k_sac_apply (lane-strided sum, then the shuffle tree): every caller forms bit-identical norms.  Result in s_gn[3] (LDS), valid
// after the caller's next barrier.
__device__ __forceinline__ void sac_group_norms(const SacOptArgs &O, float *s_gn, int tid) {
  const int w = tid >> 6, lane = tid & 63;
  if (w < 3) {
    float ss = 0.f;
    for (int p = lane; p < O.n_parts; p += 64) ss += O.ss_part[p * 3 + w];
    ss = wave_sum64(ss);
    if (lane == 0) s_gn[w] = sqrtf(ss) * O.grad_scale;
  }
}
// The clip fix-up: recompute, from the undo log, the optimizer step of every element whose group's norm reached max_norm
// ([3P optax.clip_by_global_norm] g <- g if g_norm < max_norm else (g / g_norm) * max_norm).  Every workgroup that runs this writes
// the SAME values to the same addresses (inputs: grads, undo, step_count, ss_part — all final since the previous launch), so
// concurrent callers are benign and the result does not depend on who ran last.
// (Tried out of line, `noinline`: a call inside k_sac_fwd_bwd made hipcc cap the kernel at 128 VGPRs with 96 of them spilled —
// 148 us per launch.  It stays inline and k_sac_fwd_bwd keeps it OUTSIDE its phase loop.)
__device__ __forceinline__ void sac_clip_fixup(const SacOptArgs &O, const float *s_gn, int tid, int nthreads) {
  const int NP = O.P + O.Q2 + 1;
  const float corr0 = O.undo_count[2], corr1 = O.undo_count[3];   // 1 - b^count as the speculative step itself formed them
  const float *u_p = O.undo, *u_m = O.undo + NP, *u_v = O.undo + 2 * NP, *u_tq = O.undo + 3 * NP;
  for (int i = tid; i < NP; i += nthreads) {
    const int grp = (i < O.P) ? 0 : (i < O.P + O.Q2 ? 1 : 2);
    const float gnorm = s_gn[grp];
    if (gnorm < O.max_norm) continue;                 // this group's speculative (unclipped) step stands
    float g = O.grads[i] * O.grad_scale;
    g = (g / gnorm) * O.max_norm;
    const AdamOut o = sac_adam(u_p[i], u_m[i], u_v[i], g, corr0, corr1, O.lr[grp], O.wd[grp]);
    O.params[i] = o.p;
    O.adam_m[i] = o.m;
    O.adam_v[i] = o.v;
    if (grp == 1) {
      O.target_q[i - O.P] = u_tq[i - O.P] * O.one_minus_tau + o.p * O.tau;
    } else if (grp == 2) {
      const float al = expf(o.p);                                    // 'alpha' of the step (sac.py:267), now from the clipped update
      O.metrics[3] = al;
      if (O.metrics_accum) O.metrics_accum[3] = O.undo_count[1] + al;
    }
  }
}

// jnp.floor_divide for floats ([3P] jax.numpy: remainder-based, then rounded): x1 // x2
__device__ __forceinline__ float floor_divide_f(float x1, float x2) {
  const float mod = fmodf(x1, x2);
  float div = (x1 - mod) / x2;
  if (mod != 0.0f && ((x2 < 0.0f) != (mod < 0.0f))) div -= 1.0f;
  return roundf(div);
}

// per action-dim pieces of NormalTanh (sac/parametric_distribution.py:66-73,117-120)
struct ActSample {
  float z, a, sigma, lp;
};
// The elementwise sections run on ONE wave while 15 wait at the barrier, and a lone wave issues one instruction per 4 cycles:
// libm's expf/log1pf/tanhf/logf (~30-60 instructions each) made one sample cost ~1400 cycles.  These forms use the hardware
// v_exp_f32 / v_log_f32 / v_rcp_f32 (~1 ulp each); absolute errors stay ~1e-7, far inside the parity tolerances.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
__device__ __forceinline__ float flog(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fsoftplus(float x) { return fmaxf(x, 0.0f) + flog(1.0f + fexp(-fabsf(x))); }
__device__ __forceinline__ float ftanh(float x) {
  const float e = fexp(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
  return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ ActSample normal_tanh_sample(float loc, float raw, float eps) {
  ActSample o;
  o.sigma = fsoftplus(raw) + 0.001f;
  o.z = loc + o.sigma * eps;
  o.a = ftanh(o.z);
  // log N(z; loc, sigma) with (z-loc)/sigma == eps, minus Tanh.forward_log_det_jacobian(z)
  const float ldj = 2.0f * (LOG_2 - o.z - fsoftplus(-2.0f * o.z));
  o.lp = -0.5f * eps * eps - flog(o.sigma) - LOG_SQRT_2PI - ldj;
  return o;
}

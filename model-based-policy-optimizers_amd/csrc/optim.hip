// optim.hip — generic optimizer step: [optax.apply_if_finite(] optax.adamw [)] + optional Polyak target (see mbpo_hip.h).
// HBM-bound elementwise work: 28 B per parameter (36 B with a target).  Three tiny launches:
//   k_grad_stats   per-block sum g^2 and count of non-finite gradients (fixed order, deterministic)
//   k_adamw_step   every block re-reduces the partials (wave shuffle), decides skip/apply, updates its 256 parameters
//   k_adamw_bump   count += 1 unless the update was skipped (a separate launch: every block of k_adamw_step reads count)
#include "common.hpp"

__device__ __forceinline__ float opt_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// at most ADAMW_MAX_PARTS partials however long the vector is: block b sums the chunks b, b + gridDim.x, ... in that order, so
// the result depends on n only (and equals the one-chunk-per-block sum for n <= 256 * ADAMW_MAX_PARTS)
#define ADAMW_MAX_PARTS 1024
#define ADAMW_MAX_BLOCKS 4096

__global__ void __launch_bounds__(256) k_grad_stats(const float *grads, long long n, float scale, float *part) {
  __shared__ float s_a[4], s_b[4];
  float ss_t = 0.f, nf_t = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float g = grads[i] * scale;
    const bool fin = isfinite(g);
    ss_t += fin ? g * g : 0.f;
    nf_t += fin ? 0.f : 1.f;
  }
  float ss = opt_wave_sum(ss_t), nf = opt_wave_sum(nf_t);
  if ((threadIdx.x & 63) == 0) {
    s_a[threadIdx.x >> 6] = ss;
    s_b[threadIdx.x >> 6] = nf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x + 0] = s_a[0] + s_a[1] + s_a[2] + s_a[3];
    part[2 * blockIdx.x + 1] = s_b[0] + s_b[1] + s_b[2] + s_b[3];
  }
}

struct AdamwArgs {
  float *params, *m, *v, *target, *grad_norm_out;
  const float *grads, *count, *part;
  long long n;
  int n_parts, apply_if_finite;
  float lr, wd, scale, tau, one_minus_tau;
};

__device__ __forceinline__ void reduce_parts(const float *part, int n_parts, float *ss_out, float *nf_out) {
  __shared__ float s_r[2];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (w < 2) {
    float a = 0.f;
    for (int p = lane; p < n_parts; p += 64) a += part[2 * p + w];
    a = opt_wave_sum(a);
    if (lane == 0) s_r[w] = a;
  }
  __syncthreads();
  *ss_out = s_r[0];
  *nf_out = s_r[1];
}

__global__ void __launch_bounds__(256) k_adamw_step(AdamwArgs A) {
  float ss, nf;
  reduce_parts(A.part, A.n_parts, &ss, &nf);
  if (blockIdx.x == 0 && threadIdx.x == 0 && A.grad_norm_out) A.grad_norm_out[0] = nf > 0.f ? NAN : sqrtf(ss);   // optax.global_norm
  if (A.apply_if_finite && nf > 0.f) return;   // [3P optax.apply_if_finite] skip params, moments and count
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  const float count = A.count[0] + 1.0f;
  const float c1 = 1.f - powf(b1, count), c2 = 1.f - powf(b2, count);
  auto elem = [&](float g_in, float m_in, float v_in, float p, float t_in, float &m_o, float &v_o, float &p_o, float &t_o) {
    const float g = g_in * A.scale;
    m_o = b1 * m_in + 0.1f * g;                              // optax forms (1 - b) in double: f32(0.1), f32(0.001)
    v_o = b2 * v_in + 0.001f * (g * g);
    const float mu_hat = m_o / c1;
    const float nu_hat = v_o / c2;
    p_o = p + (-A.lr) * (mu_hat / (sqrtf(nu_hat) + eps) + A.wd * p);
    t_o = A.one_minus_tau * t_in + A.tau * p_o;              // soft_update (optimizer_utils.py:155-161)
  };
  const long long gtid = (long long)blockIdx.x * 256 + threadIdx.x, gsz = (long long)gridDim.x * 256;
  // 16 bytes per lane over the 16-byte-aligned body (7 streams in, 4 out: the kernel is pure HBM traffic), scalars for the tail
  const unsigned long long al = (unsigned long long)A.params | (unsigned long long)A.m | (unsigned long long)A.v | (unsigned long long)A.grads |
                                (unsigned long long)A.target;
  const long long nq = (al & 15ull) == 0 ? (A.n >> 2) : 0;
  for (long long q = gtid; q < nq; q += gsz) {
    const f32x4 g4 = reinterpret_cast<const f32x4 *>(A.grads)[q], m4 = reinterpret_cast<const f32x4 *>(A.m)[q],
                v4 = reinterpret_cast<const f32x4 *>(A.v)[q], p4 = reinterpret_cast<const f32x4 *>(A.params)[q];
    f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
    if (A.target) t4 = reinterpret_cast<const f32x4 *>(A.target)[q];
    f32x4 mo, vo, po, to;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a, b, c, d;
      elem(g4[k], m4[k], v4[k], p4[k], t4[k], a, b, c, d);
      mo[k] = a; vo[k] = b; po[k] = c; to[k] = d;
    }
    reinterpret_cast<f32x4 *>(A.m)[q] = mo;
    reinterpret_cast<f32x4 *>(A.v)[q] = vo;
    reinterpret_cast<f32x4 *>(A.params)[q] = po;
    if (A.target) reinterpret_cast<f32x4 *>(A.target)[q] = to;
  }
  for (long long i = (nq << 2) + gtid; i < A.n; i += gsz) {
    float mo, vo, po, to;
    elem(A.grads[i], A.m[i], A.v[i], A.params[i], A.target ? A.target[i] : 0.f, mo, vo, po, to);
    A.m[i] = mo;
    A.v[i] = vo;
    A.params[i] = po;
    if (A.target) A.target[i] = to;
  }
}

__global__ void __launch_bounds__(64) k_adamw_bump(float *count, const float *part, int n_parts, int apply_if_finite) {
  float nf = 0.f;   // a count of non-finite entries: small integers, exact in any order
  for (int p = threadIdx.x; p < n_parts; p += 64) nf += part[2 * p + 1];
  nf = opt_wave_sum(nf);
  if (threadIdx.x == 0 && !(apply_if_finite && nf > 0.f)) count[0] = count[0] + 1.0f;
}

extern "C" int mbpo_adamw_step(float *params, const float *grads, float *adam_m, float *adam_v, float *step_count, int64_t n,
                               float lr, float wd, float grad_scale, int32_t apply_if_finite, float *target, float tau,
                               float *grad_norm_out, float *workspace, void *stream) {
  MBPO_REQUIRE(params && grads && adam_m && adam_v && step_count && workspace, MBPO_ERR_ARG, "adamw_step: null pointer");
  MBPO_REQUIRE(n > 0 && n < (1LL << 31), MBPO_ERR_ARG, "adamw_step: bad n");
  const long long chunks = (n + 255) / 256;
  const int parts = (int)(chunks < ADAMW_MAX_PARTS ? chunks : ADAMW_MAX_PARTS);
  const int blocks = (int)(chunks < ADAMW_MAX_BLOCKS ? chunks : ADAMW_MAX_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_grad_stats, dim3(parts), dim3(256), 0, st, grads, (long long)n, grad_scale, workspace);
  AdamwArgs A;
  A.params = params; A.m = adam_m; A.v = adam_v; A.target = target; A.grad_norm_out = grad_norm_out;
  A.grads = grads; A.count = step_count; A.part = workspace; A.n = n; A.n_parts = parts; A.apply_if_finite = apply_if_finite;
  A.lr = lr; A.wd = wd; A.scale = grad_scale; A.tau = tau; A.one_minus_tau = (float)(1.0 - (double)tau);
  hipLaunchKernelGGL(k_adamw_step, dim3(blocks), dim3(256), 0, st, A);
  hipLaunchKernelGGL(k_adamw_bump, dim3(1), dim3(64), 0, st, step_count, (const float *)workspace, parts, apply_if_finite);
  MBPO_CHECK_LAUNCH("adamw_step");
  return MBPO_OK;
}

// ------------------------------------------------------------------------------------------------ soft_update (Polyak) alone
// replaces: mbpo/utils/optimizer_utils.py:155-161 — out = (1 - tau) * target + tau * online, with (1 - tau) formed in double on
// the host exactly as the fused optimizer kernels do (k_sac_apply, k_adamw_step).  out may alias target.
__global__ void __launch_bounds__(256) k_soft_update(const float *target, const float *online, float *out, long long n, float one_minus_tau, float tau) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = one_minus_tau * target[i] + tau * online[i];
}

extern "C" int mbpo_soft_update(const float *target, const float *online, float *out, int64_t n, float tau, void *stream) {
  MBPO_REQUIRE(n >= 0, MBPO_ERR_ARG, "soft_update: negative n");
  if (n == 0) return MBPO_OK;
  MBPO_REQUIRE(target && online && out, MBPO_ERR_ARG, "soft_update: null pointer");
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_soft_update, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, target, online, out,
                     (long long)n, (float)(1.0 - (double)tau), tau);
  MBPO_CHECK_LAUNCH("soft_update");
  return MBPO_OK;
}

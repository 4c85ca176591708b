// ppo_layered.hip — PPO's minibatch update for network shapes outside the fused kernels' range (layered.hpp): the same loss
// (ppo/losses.py:56-126), flat gradient layout and random stream as k_ppo_values / k_ppo_fwd_bwd, every Dense layer one GEMM launch
// over the minibatch.  It leaves ONE slab + one set of loss sums; GAE scan, advantage moments, the slab reduction, metrics and AdamW
// are the fused path's code.
#include "layered.hpp"
#include "ppo_layered.hpp"

namespace {
constexpr float P_LOG_2 = 0.69314718055994530942f;
constexpr float P_LOG_SQRT_2PI = 0.91893853320467274178f;
__device__ __forceinline__ float p_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
__device__ __forceinline__ float p_log(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float p_softplus(float x) { return fmaxf(x, 0.0f) + p_log(1.0f + p_exp(-fabsf(x))); }
__device__ __forceinline__ float p_tanh(float x) {
  const float e = p_exp(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
  return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}

struct PrepArgs {
  const float *data, *mean, *std;
  int B, T, D, X, U;
  float reward_scaling;
  float *x_all, *trunc, *term, *rew, *step_count;
};
__global__ void __launch_bounds__(256) k_ppol_prep(PrepArgs A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long M = (long long)A.B * A.T, rows = M + A.B;
  if (i == 0) A.step_count[0] = A.step_count[0] + 1.0f;
  if (i >= rows) return;
  const int X = A.X, U = A.U, D = A.D;
  const float *src = i < M ? A.data + i * D : A.data + ((i - M) * A.T + (A.T - 1)) * D + X + U + 2;   // observation | next_observation[-1] (:84-85)
  for (int c = 0; c < X; ++c) {
    float o = src[c];
    if (A.mean) o = (o - A.mean[c]) / A.std[c];
    A.x_all[i * X + c] = o;
  }
  if (i < M) {
    const float *row = A.data + i * D;
    const float tr = row[D - 1], disc = row[X + U + 1];
    A.trunc[i] = tr;
    A.term[i] = (1.f - disc) * (1.f - tr);       // :89
    A.rew[i] = row[X + U] * A.reward_scaling;    // :87
  }
}

struct HeadArgs {
  const float *data, *outp, *values, *vs, *adv, *mom, *ent_noise;
  unsigned long long seed, offset;
  const unsigned long long *rng_dev;
  int B, T, D, X, U, normalize_advantage;
  float entropy_cost, clip_eps;
  float *dy_pi, *dy_v;       // [M][2U], [M + B]
  float *extras;             // [3]
};
// 256 rows per workgroup (one thread each), a fixed LDS tree over the workgroup's loss partials -> extras[4 * workgroup + k]; the
// reduction launch adds the workgroups' partials in order — deterministic.  (One workgroup walking all M rows was 70 us at C3's
// M = 20 480: as long as a whole 256-wide layer.)
__global__ void __launch_bounds__(256) k_ppol_heads(HeadArgs A) {
  __shared__ float s_red[3][256];
  const int tid = threadIdx.x;
  const long long M = (long long)A.B * A.T;
  const int X = A.X, U = A.U, D = A.D;
  const float invM = 1.0f / (float)M;
  const float adv_mean = A.normalize_advantage ? A.mom[0] : 0.f, adv_istd = A.normalize_advantage ? 1.0f / (A.mom[1] + 1e-8f) : 1.f;
  const RngKey rk = rng_resolve(A.seed, A.offset, A.rng_dev);
  float l_pol = 0.f, l_v = 0.f, l_ent = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + tid; i < M + A.B; i += (long long)gridDim.x * 256) {
    if (i >= M) {
      A.dy_v[i] = 0.f;       // the bootstrap rows' values carry no gradient (stop_gradient inside compute_gae)
      continue;
    }
    const float *row = A.data + i * D;
    float lp_t = 0.f, ent = 0.f;
    for (int d = 0; d < U; ++d) {
      const float loc = A.outp[i * 2 * U + d], raw = A.outp[i * 2 * U + U + d];
      const float sg = p_softplus(raw) + 0.001f;
      const float z = row[2 * X + U + 3 + d];
      const float q = (z - loc) / sg, lsg = p_log(sg);
      lp_t += -0.5f * q * q - lsg - P_LOG_SQRT_2PI - 2.0f * (P_LOG_2 - z - p_softplus(-2.0f * z));      // :91-92
      const long long nidx = i * U + d;
      const float eps = A.ent_noise ? A.ent_noise[nidx] : philox_normal(rk.seed, rk.offset, MBPO_STREAM_ENTROPY, (unsigned long long)nidx);
      const float zf = loc + sg * eps;
      ent += 0.5f + P_LOG_SQRT_2PI + lsg + 2.0f * (P_LOG_2 - zf - p_softplus(-2.0f * zf));              // :117
    }
    const float lp_b = row[2 * X + U + 2];
    const float adv = (A.adv[i] - adv_mean) * adv_istd, vs = A.vs[i], v = A.values[i];
    const float rho = p_exp(lp_t - lp_b);                                                                // :103
    const float lo = 1.f - A.clip_eps, hi = 1.f + A.clip_eps;
    const float s1 = rho * adv, s2 = fminf(fmaxf(rho, lo), hi) * adv;
    const bool inside = (rho >= lo) && (rho <= hi);
    const float w = inside ? 1.f : (s1 < s2 ? 1.f : 0.f);
    const float g_lp = -invM * rho * adv * w, g_ent = -A.entropy_cost * invM;
    l_pol += -fminf(s1, s2);
    l_v += 0.5f * (vs - v) * (vs - v);
    l_ent += ent;
    A.dy_v[i] = -(vs - v) * invM;                                                                        // :112-114
    for (int d = 0; d < U; ++d) {
      const float loc = A.outp[i * 2 * U + d], raw = A.outp[i * 2 * U + U + d];
      const float sg = p_softplus(raw) + 0.001f;
      const float z = row[2 * X + U + 3 + d];
      const float q = (z - loc) / sg;
      const long long nidx = i * U + d;
      const float eps = A.ent_noise ? A.ent_noise[nidx] : philox_normal(rk.seed, rk.offset, MBPO_STREAM_ENTROPY, (unsigned long long)nidx);
      const float th = p_tanh(loc + sg * eps);
      A.dy_pi[i * 2 * U + d] = g_lp * (q / sg) + g_ent * (-2.f * th);
      A.dy_pi[i * 2 * U + U + d] = (g_lp * ((q * q - 1.f) / sg) + g_ent * (1.f / sg - 2.f * th * eps)) * fast_sigmoid(raw);
    }
  }
  s_red[0][tid] = l_pol; s_red[1][tid] = l_v; s_red[2][tid] = l_ent;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 3; ++k) s_red[k][tid] += s_red[k][tid + s];
    }
    __syncthreads();
  }
  if (tid < 3) A.extras[4 * blockIdx.x + tid] = s_red[tid][0];
}

struct Carve {
  float *base;
  long long off;
  float *take(long long n) {
    float *p = base ? base + off : nullptr;
    off += (n + 3) & ~3LL;
    return p;
  }
};
struct Bufs {
  float *x_all, *values, *outp, *dy_pi, *dy_v, *pp[2], *pq[2], *part, *part2, *extras;   // pp/part: policy pass, pq/part2: value pass
  int n_heads;
  float *Zv[MBPO_MAX_LAYERS + 1], *Hv[MBPO_MAX_LAYERS + 1], *Zp[MBPO_MAX_LAYERS + 1], *Hp[MBPO_MAX_LAYERS + 1];
};
long long carve_all(float *base, const mbpo_ppo_desc *d, const LayeredNet &pi, const LayeredNet &v, Bufs *b) {
  Carve c{base, 0};
  const long long M = (long long)d->batch_size * d->unroll_length, R = M + d->batch_size, X = d->x_dim, U = d->u_dim;
  b->x_all = c.take(R * X); b->values = c.take(R); b->outp = c.take(M * 2 * U); b->dy_pi = c.take(M * 2 * U); b->dy_v = c.take(R);
  for (int l = 1; l < v.L; ++l) { b->Zv[l] = c.take(R * v.dims[l]); b->Hv[l] = c.take(R * v.dims[l]); }
  for (int l = 1; l < pi.L; ++l) { b->Zp[l] = c.take(M * pi.dims[l]); b->Hp[l] = c.take(M * pi.dims[l]); }
  int mh = layered_max_hidden(pi);
  const int mv = layered_max_hidden(v);
  mh = mv > mh ? mv : mh;
  b->pp[0] = c.take(M * mh); b->pp[1] = c.take(M * mh);
  b->pq[0] = c.take(R * mh); b->pq[1] = c.take(R * mh);
  b->part = c.take(layered_part_floats(pi, (int)M));
  b->part2 = c.take(layered_part_floats(v, (int)R));
  b->n_heads = (int)((R + 255) / 256);
  b->extras = c.take(4LL * b->n_heads);
  return c.off;
}
}  // namespace

long long ppo_layered_floats(const mbpo_ppo_desc *d, const MlpDev &pi, const MlpDev &v) {
  Bufs b;
  return carve_all(nullptr, d, layered_net(pi, nullptr, 0, 1), layered_net(v, nullptr, 0, 1), &b);
}

int ppo_layered_values(const mbpo_ppo_desc *d, const MlpDev &pi, const MlpDev &v, float *ws, float *trunc, float *term, float *rew,
                       float **values_out, hipStream_t st) {
  const LayeredNet npi = layered_net(pi, d->params, 0, 1), nv = layered_net(v, d->params + pi.n_params, 0, 1);
  Bufs b;
  carve_all(ws, d, npi, nv, &b);
  const long long M = (long long)d->batch_size * d->unroll_length, R = M + d->batch_size;
  PrepArgs A = {d->data, d->norm_mean, d->norm_std, d->batch_size, d->unroll_length, d->row_len, d->x_dim, d->u_dim, d->reward_scaling,
                b.x_all, trunc, term, rew, d->step_count};
  hipLaunchKernelGGL(k_ppol_prep, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, A);
  // the value net on all R rows and — it depends on nothing the GAE / moments launches produce — the policy on the M loss rows, level by
  // level in one launch each (ppo_layered_fwd_bwd of the same step finds outp / Zp / Hp in the workspace)
  const LayeredFwd f[2] = {{nv, b.x_all, 0, (int)R, b.Zv, b.Hv, b.values}, {npi, b.x_all, 0, (int)M, b.Zp, b.Hp, b.outp}};
  int rc = layered_forward_multi(f, 2, st);
  if (rc != MBPO_OK) return rc;
  *values_out = b.values;
  MBPO_CHECK_LAUNCH("ppo_layered_values");
  return MBPO_OK;
}

int ppo_layered_fwd_bwd(const mbpo_ppo_desc *d, const MlpDev &pi, const MlpDev &v, float *ws, const float *vs, const float *adv,
                        const float *mom, float *slab, float **extras_out, int *n_extras_out, hipStream_t st) {
  const LayeredNet npi = layered_net(pi, d->params, 0, 1), nv = layered_net(v, d->params + pi.n_params, 0, 1);
  Bufs b;
  carve_all(ws, d, npi, nv, &b);
  const long long M = (long long)d->batch_size * d->unroll_length, R = M + d->batch_size;
  int rc;       // the policy's forward pass ran beside the value net's in ppo_layered_values
  HeadArgs A = {d->data, b.outp, b.values, vs, adv, mom, d->entropy_noise, d->seed, d->offset, (const unsigned long long *)d->rng_dev,
                d->batch_size, d->unroll_length, d->row_len, d->x_dim, d->u_dim, d->normalize_advantage, d->entropy_cost,
                d->clipping_epsilon, b.dy_pi, b.dy_v, b.extras};
  hipLaunchKernelGGL(k_ppol_heads, dim3(b.n_heads), dim3(256), 0, st, A);
  *extras_out = b.extras;
  *n_extras_out = b.n_heads;
  const LayeredBwd g[2] = {{npi, b.x_all, 0, (int)M, b.Zp, b.Hp, b.dy_pi, slab, 0, nullptr, b.pp[0], b.pp[1], b.part},
                           {nv, b.x_all, 0, (int)R, b.Zv, b.Hv, b.dy_v, slab + pi.n_params, 0, nullptr, b.pq[0], b.pq[1], b.part2}};
  if ((rc = layered_backward_multi(g, 2, st)) != MBPO_OK) return rc;
  MBPO_CHECK_LAUNCH("ppo_layered_fwd_bwd");
  return MBPO_OK;
}

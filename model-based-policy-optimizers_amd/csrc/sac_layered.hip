// sac_layered.hip — the forward/backward half of SAC's sgd_step for network shapes outside the fused kernel's range (layered.hpp):
// same losses (sac/losses.py:61-125), same flat gradient layout, same random streams as k_sac_fwd_bwd — it leaves ONE "tile" of
// slabs (the whole minibatch's gradient, already divided by B) + the four loss sums, and everything downstream of the fwd/bwd
// launch (slab reduction, metrics, clip partials, peer exchange, optimizer step) is the fused path's code, unchanged.
#include "layered.hpp"
#include "sac_layered.hpp"

namespace {
constexpr float L_LOG_2 = 0.69314718055994530942f;
constexpr float L_LOG_SQRT_2PI = 0.91893853320467274178f;

__device__ __forceinline__ float l_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
__device__ __forceinline__ float l_log(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float l_softplus(float x) { return fmaxf(x, 0.0f) + l_log(1.0f + l_exp(-fabsf(x))); }
__device__ __forceinline__ float l_tanh(float x) {
  const float e = l_exp(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
  return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}
// NormalTanh pieces per action dimension (sac/parametric_distribution.py:66-73,117-120) — the forms of k_sac_fwd_bwd
struct LSample {
  float a, sigma, lp;
};
__device__ __forceinline__ LSample l_sample(float loc, float raw, float eps) {
  LSample o;
  o.sigma = l_softplus(raw) + 0.001f;
  const float z = loc + o.sigma * eps;
  o.a = l_tanh(z);
  const float ldj = 2.0f * (L_LOG_2 - z - l_softplus(-2.0f * z));
  o.lp = -0.5f * eps * eps - l_log(o.sigma) - L_LOG_SQRT_2PI - ldj;
  return o;
}
__device__ __forceinline__ float l_floor_divide(float x1, float x2) {
  const float mod = fmodf(x1, x2);
  float div = (x1 - mod) / x2;
  if (mod != 0.0f && ((x2 < 0.0f) != (mod < 0.0f))) div -= 1.0f;
  return roundf(div);
}

struct PrepArgs {
  const float *batch, *mean, *std;
  int B, D, X, U;
  float *xo, *xn, *qin_d, *qin_p, *qin_n;
  SacLayeredBegin bg;
};
// normalised observations, the critics' input rows; thread (0, 0) does what block 0 of k_sac_fwd_bwd does at its top
__global__ void __launch_bounds__(256) k_sacl_prep(PrepArgs A) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b == 0) {
    A.bg.step_count[0] = A.bg.step_count[0] + 1.0f;
    const unsigned int slot = A.bg.seq[0] & 1u;
    A.bg.slot_word[0] = slot;
    float *q = reinterpret_cast<float *>(A.bg.seq) + 2 + 3 * slot;
    q[0] = 0.f; q[1] = 0.f; q[2] = 0.f;
    if (A.bg.p2p_epoch) {
      A.bg.p2p_epoch[0] = A.bg.p2p_epoch[0] + 1u;
      A.bg.p2p_epoch[1] = A.bg.p2p_epoch[1] + A.bg.p2p_blocks;
    }
  }
  if (b >= A.B) return;
  const int X = A.X, U = A.U, XU = X + U;
  const float *row = A.batch + (long long)b * A.D;
  for (int j = 0; j < X; ++j) {
    const float m = A.mean ? A.mean[j] : 0.f, s = A.std ? A.std[j] : 1.f;
    const float o = (row[j] - m) / s, n = (row[X + U + 2 + j] - m) / s;
    A.xo[(long long)b * X + j] = o;
    A.xn[(long long)b * X + j] = n;
    A.qin_d[(long long)b * XU + j] = o;
    A.qin_p[(long long)b * XU + j] = o;
    A.qin_n[(long long)b * XU + j] = n;
  }
  for (int j = 0; j < U; ++j) A.qin_d[(long long)b * XU + X + j] = row[X + j];
}

struct SampleArgs {
  const float *outp, *outn;                       // policy(obs), policy(next_obs): [B][2U]
  const float *noise_alpha, *noise_critic, *noise_actor;
  unsigned long long seed, offset;
  const unsigned long long *rng_dev;
  int B, X, U;
  float *qin_p, *qin_n, *lp_alpha, *lp_next, *lp_actor, *eps_actor;
};
__global__ void __launch_bounds__(256) k_sacl_sample(SampleArgs A) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= A.B) return;
  const int U = A.U, XU = A.X + U;
  const RngKey rk = rng_resolve(A.seed, A.offset, A.rng_dev);
  float la = 0.f, ln = 0.f, lc = 0.f;
  for (int d = 0; d < U; ++d) {
    const long long nidx = (long long)b * U + d;
    const float e_al = A.noise_alpha ? A.noise_alpha[nidx] : philox_normal(rk.seed, rk.offset, MBPO_STREAM_SAC_ALPHA, (unsigned long long)nidx);
    const float e_cr = A.noise_critic ? A.noise_critic[nidx] : philox_normal(rk.seed, rk.offset, MBPO_STREAM_SAC_CRITIC, (unsigned long long)nidx);
    const float e_ac = A.noise_actor ? A.noise_actor[nidx] : philox_normal(rk.seed, rk.offset, MBPO_STREAM_SAC_ACTOR, (unsigned long long)nidx);
    const float loc = A.outp[(long long)b * 2 * U + d], raw = A.outp[(long long)b * 2 * U + U + d];
    la += l_sample(loc, raw, e_al).lp;                                                 // alpha loss sample (:66-68)
    const LSample sa = l_sample(loc, raw, e_ac);                                       // actor loss sample (:117-120)
    lc += sa.lp;
    A.qin_p[(long long)b * XU + A.X + d] = sa.a;
    A.eps_actor[nidx] = e_ac;
    const LSample sn = l_sample(A.outn[(long long)b * 2 * U + d], A.outn[(long long)b * 2 * U + U + d], e_cr);   // next action (:80-87)
    ln += sn.lp;
    A.qin_n[(long long)b * XU + A.X + d] = sn.a;
  }
  A.lp_alpha[b] = la;
  A.lp_next[b] = ln;
  A.lp_actor[b] = lc;
}

struct LossArgs {
  const float *batch, *qd, *qp, *qn;              // q*: [2][B]
  const float *lp_alpha, *lp_next, *lp_actor, *log_alpha;
  int B, D, X, U;
  float discounting, reward_scaling, target_entropy;
  int neq;
  float neq_cd, neq_tl, neq_tu, neq_dt;
  float *dqd, *dqp;                               // [2][B]
  float *slab_ex;                                 // [4]: sum err0^2, sum actor, sum alpha, sum err1^2
};
// one workgroup: thread t takes rows t, t + 1024, ... (fixed order), then an LDS tree — deterministic
__global__ void __launch_bounds__(1024) k_sacl_losses(LossArgs A) {
  __shared__ float s_red[4][1024];
  const int tid = threadIdx.x;
  const float alpha = expf(A.log_alpha[0]);
  const float invB = 1.0f / (float)A.B;
  float e0 = 0.f, e1 = 0.f, ac = 0.f, al = 0.f;
  const int X = A.X, U = A.U, B = A.B;
  for (int b = tid; b < B; b += 1024) {
    const float *row = A.batch + (long long)b * A.D;
    const float nq = fminf(A.qn[b], A.qn[B + b]);
    const float next_v = nq - alpha * A.lp_next[b];                                      // :89
    float gamma = A.discounting;
    if (A.neq) {                                                                          // :90-96
      const float pseudo = row[X + U - 1];
      float tfa = (A.neq_tu - A.neq_tl) / 2.0f * pseudo + (A.neq_tu + A.neq_tl) / 2.0f;
      tfa = l_floor_divide(tfa, A.neq_dt) * A.neq_dt;
      gamma = expf(-A.neq_cd * tfa);
    }
    const float target = row[X + U] * A.reward_scaling + row[X + U + 1] * gamma * next_v;   // :101-103
    const float trunc = row[A.D - 1];
    const float r0 = (A.qd[b] - target) * (1.f - trunc), r1 = (A.qd[B + b] - target) * (1.f - trunc);   // q_error :104-108
    e0 += r0 * r0;
    e1 += r1 * r1;
    A.dqd[b] = r0 * (1.f - trunc) * (0.5f * invB);          // loss = 0.5 * mean(err^2) over [B,2]
    A.dqd[B + b] = r1 * (1.f - trunc) * (0.5f * invB);
    al += alpha * (-A.lp_alpha[b] - A.target_entropy);      // :70-72
    const float q0 = A.qp[b], q1 = A.qp[B + b];
    ac += alpha * A.lp_actor[b] - fminf(q0, q1);            // :123-124
    float g0 = 0.f, g1 = 0.f;                               // ties split evenly, as jnp.min's gradient does
    if (q0 < q1) g0 = -invB;
    else if (q1 < q0) g1 = -invB;
    else g0 = g1 = -0.5f * invB;
    A.dqp[b] = g0;
    A.dqp[B + b] = g1;
  }
  s_red[0][tid] = e0; s_red[1][tid] = ac; s_red[2][tid] = al; s_red[3][tid] = e1;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s_red[k][tid] += s_red[k][tid + s];
    }
    __syncthreads();
  }
  if (tid < 4) A.slab_ex[tid] = s_red[tid][0];
}

struct ActorHeadArgs {
  const float *dqin;                              // [2][B][X+U]: d(actor loss)/d(critic input) per critic
  const float *outp, *eps_actor, *log_alpha;
  int B, X, U;
  float *doutp;                                   // [B][2U]
};
__global__ void __launch_bounds__(256) k_sacl_actor_head(ActorHeadArgs A) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= A.B) return;
  const int U = A.U, XU = A.X + U;
  const float alpha = expf(A.log_alpha[0]);
  const float invB = 1.0f / (float)A.B;
  for (int d = 0; d < U; ++d) {
    const float loc = A.outp[(long long)b * 2 * U + d], raw = A.outp[(long long)b * 2 * U + U + d], eps = A.eps_actor[(long long)b * U + d];
    const LSample s = l_sample(loc, raw, eps);
    const float dLda = A.dqin[(long long)b * XU + A.X + d] + A.dqin[((long long)A.B + b) * XU + A.X + d];
    // z = loc + sigma*eps;  log_prob = const - log(sigma) - log(1 - tanh(z)^2)  =>  dlp/dz = 2a, dlp/dsigma = -1/sigma
    const float gz = dLda * (1.f - s.a * s.a) + alpha * invB * 2.f * s.a;
    const float gsig = gz * eps - alpha * invB / s.sigma;
    A.doutp[(long long)b * 2 * U + d] = gz;
    A.doutp[(long long)b * 2 * U + U + d] = gsig * fast_sigmoid(raw);
  }
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
  float *xo, *xn, *qin_d, *qin_p, *qin_n, *outp, *outn;
  float *Zp[MBPO_MAX_LAYERS + 1], *Hp[MBPO_MAX_LAYERS + 1];
  float *Zqd[MBPO_MAX_LAYERS + 1], *Hqd[MBPO_MAX_LAYERS + 1], *Zqp[MBPO_MAX_LAYERS + 1];
  float *pp[2], *pq[2], *pr[2];      // per-pass scratch: passes of one level run side by side in one launch
  float *qd, *qp, *qn, *lp_alpha, *lp_next, *lp_actor, *eps_actor, *dqd, *dqp, *dqin, *doutp, *part, *part2;
};

long long carve_all(float *base, const mbpo_sac_desc *d, const LayeredNet &pi, const LayeredNet &q, Bufs *b) {
  Carve c{base, 0};
  const long long B = d->batch_size, X = d->x_dim, U = d->u_dim, XU = X + U;
  b->xo = c.take(B * X); b->xn = c.take(B * X);
  b->qin_d = c.take(B * XU); b->qin_p = c.take(B * XU); b->qin_n = c.take(B * XU);
  b->outp = c.take(B * 2 * U); b->outn = c.take(B * 2 * U);
  for (int l = 1; l < pi.L; ++l) { b->Zp[l] = c.take(B * pi.dims[l]); b->Hp[l] = c.take(B * pi.dims[l]); }
  for (int l = 1; l < q.L; ++l) {
    b->Zqd[l] = c.take(2 * B * q.dims[l]); b->Hqd[l] = c.take(2 * B * q.dims[l]); b->Zqp[l] = c.take(2 * B * q.dims[l]);
  }
  int mh = layered_max_hidden(pi);
  const int mq = layered_max_hidden(q);
  mh = mq > mh ? mq : mh;
  b->pp[0] = c.take(2 * B * mh); b->pp[1] = c.take(2 * B * mh);
  b->pq[0] = c.take(2 * B * mh); b->pq[1] = c.take(2 * B * mh);
  b->pr[0] = c.take(2 * B * mh); b->pr[1] = c.take(2 * B * mh);
  b->qd = c.take(2 * B); b->qp = c.take(2 * B); b->qn = c.take(2 * B);
  b->lp_alpha = c.take(B); b->lp_next = c.take(B); b->lp_actor = c.take(B); b->eps_actor = c.take(B * U);
  b->dqd = c.take(2 * B); b->dqp = c.take(2 * B); b->dqin = c.take(2 * B * XU); b->doutp = c.take(B * 2 * U);
  const long long pa = layered_part_floats(pi, (int)B), pb = layered_part_floats(q, (int)B);
  b->part = c.take(pa > pb ? pa : pb);
  b->part2 = c.take(pb);
  return c.off;
}
}  // namespace

long long sac_layered_floats(const mbpo_sac_desc *d, const MlpDev &pi, const MlpDev &q) {
  Bufs b;
  return carve_all(nullptr, d, layered_net(pi, nullptr, 0, 1), layered_net(q, nullptr, 0, 2), &b);
}

int sac_layered_fwd_bwd(const mbpo_sac_desc *d, const MlpDev &pi, const MlpDev &q, const MlpDev &qt, float *ws, float *slab_pi,
                        float *slab_q, float *slab_ex, const SacLayeredBegin &bg, hipStream_t st) {
  const int B = d->batch_size, X = d->x_dim, U = d->u_dim, XU = X + U, P = pi.n_params, Q = q.n_params;
  const LayeredNet npi = layered_net(pi, d->params, 0, 1);
  const LayeredNet nq = layered_net(q, d->params + P, Q, 2);
  const LayeredNet nqt = layered_net(qt, d->target_q, Q, 2);
  Bufs b;
  carve_all(ws, d, npi, nq, &b);
  const unsigned rb = (unsigned)((B + 255) / 256);
  int rc;
  {
    PrepArgs A = {d->batch, d->norm_mean, d->norm_std, B, d->row_len, X, U, b.xo, b.xn, b.qin_d, b.qin_p, b.qin_n, bg};
    hipLaunchKernelGGL(k_sacl_prep, dim3(rb), dim3(256), 0, st, A);
  }
  // policy(obs) with stored activations and policy(next_obs) without: the two passes of a level in one launch
  float *ping[MBPO_MAX_LAYERS + 1], *pinq[MBPO_MAX_LAYERS + 1], *pinr[MBPO_MAX_LAYERS + 1];
  for (int l = 0; l <= MBPO_MAX_LAYERS; ++l) { ping[l] = b.pp[l & 1]; pinq[l] = b.pq[l & 1]; pinr[l] = b.pr[l & 1]; }
  {
    const LayeredFwd f[2] = {{npi, b.xo, 0, B, b.Zp, b.Hp, b.outp}, {npi, b.xn, 0, B, nullptr, ping, b.outn}};
    if ((rc = layered_forward_multi(f, 2, st)) != MBPO_OK) return rc;
  }
  {
    SampleArgs A = {b.outp, b.outn, d->noise_alpha, d->noise_critic, d->noise_actor, d->seed, d->offset,
                    (const unsigned long long *)d->rng_dev, B, X, U, b.qin_p, b.qin_n, b.lp_alpha, b.lp_next, b.lp_actor, b.eps_actor};
    hipLaunchKernelGGL(k_sacl_sample, dim3(rb), dim3(256), 0, st, A);
  }
  // the two critics on (s, a) [stored], on (s, a~pi) [pre-activations stored], the two target critics on (s', a'): three passes per level
  {
    const LayeredFwd f[3] = {{nq, b.qin_d, 0, B, b.Zqd, b.Hqd, b.qd}, {nq, b.qin_p, 0, B, b.Zqp, pinq, b.qp}, {nqt, b.qin_n, 0, B, nullptr, pinr, b.qn}};
    if ((rc = layered_forward_multi(f, 3, st)) != MBPO_OK) return rc;
  }
  {
    LossArgs A = {d->batch, b.qd, b.qp, b.qn, b.lp_alpha, b.lp_next, b.lp_actor, d->params + P + 2 * Q, B, d->row_len, X, U,
                  d->discounting, d->reward_scaling, d->target_entropy, d->non_equidistant_time, d->continuous_discounting,
                  d->min_time_between_switches, d->max_time_between_switches, d->env_dt, b.dqd, b.dqp, slab_ex};
    hipLaunchKernelGGL(k_sacl_losses, dim3(1), dim3(1024), 0, st, A);
  }
  // critic loss -> critic parameters, and (side by side) actor loss -> the action through the (old) critics
  {
    const LayeredBwd g[2] = {{nq, b.qin_d, 0, B, b.Zqd, b.Hqd, b.dqd, slab_q, Q, nullptr, b.pp[0], b.pp[1], b.part},
                             {nq, b.qin_p, 0, B, b.Zqp, nullptr, b.dqp, nullptr, 0, b.dqin, b.pq[0], b.pq[1], b.part2}};
    if ((rc = layered_backward_multi(g, 2, st)) != MBPO_OK) return rc;
  }
  // -> the policy's outputs -> policy parameters
  {
    ActorHeadArgs A = {b.dqin, b.outp, b.eps_actor, d->params + P + 2 * Q, B, X, U, b.doutp};
    hipLaunchKernelGGL(k_sacl_actor_head, dim3(rb), dim3(256), 0, st, A);
  }
  if ((rc = layered_backward(npi, b.xo, 0, B, b.Zp, b.Hp, b.doutp, slab_pi, P, nullptr, b.pp[0], b.pp[1], b.part, st)) != MBPO_OK) return rc;
  MBPO_CHECK_LAUNCH("sac_layered_fwd_bwd");
  (void)XU;
  return MBPO_OK;
}

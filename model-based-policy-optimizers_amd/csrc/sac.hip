// sac.hip — S3-S8: SAC sgd_step as three kernels (see include/mbpo_hip.h for the flat state layout).
//
//   k_sac_fwd_bwd   2 workgroups per 16-sample tile: a CRITIC role (pi(s') fwd, target-Q fwd, Q fwd+bwd) and an
//                   ACTOR role (pi(s) fwd+bwd, alpha loss, Q fwd + input-gradient bwd).  All three losses are taken
//                   at the OLD parameters (sac/sac.py:234-258), so the roles are independent.  fp32 MFMA; activations,
//                   pre-activations and deltas live in LDS; each workgroup writes its weight gradients to a private
//                   SLAB laid out like the flat params (no atomics: the cross-tile sum is a fixed-order reduce ->
//                   bitwise reproducible).
//   k_sac_reduce    grads[i] = sum over tiles of slab[t][i]; loss metrics; per-group sum-of-squares partials.
//   k_sac_apply     clip_by_global_norm + AdamW per optimizer group + Polyak on the critics.
//
// Algorithmic work per sample per sgd_step: 2*(5P + 12Q) FLOP — latency-bound at B=256, hence the
// few fat launches and the slab scheme instead of a tree of small kernels.
#include "common.hpp"
#include "chain_run.hpp"
#include "p2p.hpp"
#include <string.h>
int mbpo_p2p_make_dev(const mbpo_p2p_desc *d, P2pDev *P);

#define LOG_SQRT_2PI 0.91893853320467274178f
#define LOG_2 0.69314718055994530942f

// What one wave group walks in one phase, filled in by the host (sac_chain_table): the kernel fetches its entry with one
// scalar load instead of ~200 scalar instructions of role/phase/chain case analysis per phase.
struct SacChainDesc {
  int mode;        // CH_*
  int netid;       // 0 = policy shape, 1 = critic shape
  int base_sel;    // parameters relative to: 0 = params, 1 = target_q
  int param_off;   // floats
  int x, ldx;      // LDS offsets (floats from the dynamic-LDS base), -1 = none
  int pp0, pp1, zb, hb, y, dx;
  int slab_sel;    // 0 none, 1 = policy slab, 2 = critic slab
  int slab_off;    // floats inside the tile's slab
  int pad0, pad1;
};

struct SacArgs {
  MlpDev pi, q, qt;
  NetShape sh_pi, sh_q;
  SacChainDesc tab[2][5][4];   // [role][phase (index nph = idle)][chain]
  int X, U, B, D;
  const float *batch, *norm_mean, *norm_std, *log_alpha;
  const float *noise_alpha, *noise_critic, *noise_actor;
  unsigned long long seed, offset;
  const unsigned long long *rng_dev;
  float discounting, reward_scaling, target_entropy;
  int neq;                      // non_equidistant_time (losses.py:90-98)
  float neq_cd, neq_tl, neq_tu, neq_dt;
  float *slab_pi, *slab_q, *slab_ex;
  int ld_x, ld_xu, ld_h, ld_y, LH;
  unsigned int *p2p_epoch;      // multi-GPU peer exchange: [0] += 1, [1] += p2p_blocks at the start of every step (or NULL)
  unsigned int p2p_blocks;
  unsigned long long *stamps;   // measurement hook (mbpo_debug_set_stamps): [2 roles][16] s_memtime values of tile 0, or NULL
};

// Timeline stamps for DESIGN.md's phase breakdown: one s_memtime per phase boundary, written by thread 0 of tile 0.
#define SAC_STAMP(i)                                                                   \
  if (A.stamps && tile == 0 && tid == 0) {                                             \
    unsigned long long t_;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
    A.stamps[role * 16 + (i)] = t_;                                                    \
  }

static unsigned long long *g_sac_stamps = nullptr;
// Measurement hook (not part of include/mbpo_hip.h): device buffer of >= 32 uint64 that k_sac_fwd_bwd fills with
// s_memtime stamps at its phase boundaries (tile 0, both roles); NULL switches the stamps off.
extern "C" int mbpo_debug_set_stamps(void *buf) {
  g_sac_stamps = (unsigned long long *)buf;
  return MBPO_OK;
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

// Workgroup = 8 waves = one 16-sample tile in one ROLE.  Wave w = (chain c = w/2, sub = w%2): each network chain is
// shared by 2 waves (column slices of H/2), chains advance side by side in lockstep, one workgroup barrier per layer
// (wave_mlp.hpp "Lockstep groups").  The per-sgd_step critical path is ~16 layer-steps of ~32 MFMAs per wave.
// (16 waves x 4 per chain halves the MFMAs per step again but caps a wave at 128 VGPRs: hipcc spilled ~850 registers.)
//   CRITIC role                                     ACTOR(+alpha) role
//   P1  c0: pi(s') fwd  c1: Q1(s,a) fwd+store       P1  c0: pi(s) fwd+store
//                       c2: Q2(s,a) fwd+store
//   -- sample a', log pi'                           -- sample alpha/actor actions
//   P2  c0: Qtgt1(s',a')  c1: Qtgt2(s',a')          P2  c0: Q1(s,a~) fwd (z kept)  c1: Q2(s,a~)
//   -- target y, errors, dL/dq                      -- min_q, dL/dq
//   P3  per layer: c0/c1: Q1/Q2 dgrad               P3  per layer: c0/c1: Q1/Q2 input-gradient
//                  c2/c3: Q1/Q2 wgrad               -- dL/dlogits
//                                                   P4  per layer: c0: pi dgrad   c1: pi wgrad
template <int H, int SP, bool WIDE>   // SP = waves per chain: 4 chains x SP waves; WIDE: chain_run.hpp fast_shape
__global__ void __launch_bounds__(256 * SP) k_sac_fwd_bwd(SacArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid_ = threadIdx.x, nthreads = 256 * SP;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int chain = wave / SP, sub = wave % SP;
  const int role = blockIdx.x & 1;  // 0 = critic, 1 = actor(+alpha)
  const int tile = blockIdx.x >> 1;
  const int X = A.X, U = A.U, D = A.D, B = A.B;
  const int row0 = tile * 16;
  const int ld_x = A.ld_x, ld_xu = A.ld_xu, ld_h = A.ld_h, ld_y = A.ld_y, LH = A.LH;
  const int T = 16 * ld_h;  // one hidden tile

  // ---- LDS carve (every region a multiple of 4 floats: rows stay 16-byte aligned) ----
  float *s_row = smem;                      // [16][D] (flat copy of the tile's transitions)
  const int D4 = D;
  float *s_sn = s_row + 16 * ((D + 3) & ~3);  // [16][ld_x]   normalised obs
  float *s_sn2 = s_sn + 16 * ld_x;          // [16][ld_x]   normalised next obs
  float *s_qin = s_sn2 + 16 * ld_x;         // [16][ld_xu]  [sn, a]
  float *s_qin2 = s_qin + 16 * ld_xu;       // [16][ld_xu]  [s'n, a']
  float *s_pp = s_qin2 + 16 * ld_xu;        // 4 tiles: ping-pong hidden / delta tiles
  float *s_store = s_pp + 4 * T;            // 4*LH tiles: stored z / h
  float *s_y = s_store + 4 * LH * T;        // [3][16][ld_y]  network outputs
  float *s_dy = s_y + 3 * 16 * ld_y;        // [2][16][ld_y]  output gradients
  float *s_dx = s_dy + 2 * 16 * ld_y;       // [2][16][ld_xu] critic input gradients (actor role)
  const int U4 = (16 * U + 3) & ~3;
  float *s_eps = s_dx + 2 * 16 * ld_xu;     // [16][U] ...
  float *s_a = s_eps + U4;
  float *s_sig = s_a + U4;
  float *s_raw = s_sig + U4;
  float *s_lp = s_raw + U4;
  float *s_lpa = s_lp + U4;
  float *s_scal = s_lpa + U4;               // [4][16]

  {
    const int tid = tid_;
    SAC_STAMP(0);
    if (A.p2p_epoch && blockIdx.x == 0 && tid == 0) {   // the exchange's epoch is stable while the reduce/gather kernels read it
      A.p2p_epoch[0] = A.p2p_epoch[0] + 1u;
      A.p2p_epoch[1] = A.p2p_epoch[1] + A.p2p_blocks;
    }
  }
  // requested now, consumed after the first layer phase: the two scalar loads overlap the tile load instead of preceding it
  const float log_alpha_v = A.log_alpha[0];
  const RngKey rk_ = rng_resolve(A.seed, A.offset, A.rng_dev);
  const float invB = 1.0f / (float)B;
  const float *pi_p = A.pi.params, *q1_p = A.q.params, *q2_p = A.q.params + A.q.net_stride;
  const float *t1_p = A.qt.params, *t2_p = A.qt.params + A.qt.net_stride;
  const int QL = A.q.n_layers, PL = A.pi.n_layers;
  const int Lmax = QL > PL ? QL : PL;
  float *y_pi = s_y, *y_q1 = s_y + 16 * ld_y, *y_q2 = s_y + 2 * 16 * ld_y;
  // stored activations.  critic role: Q1 (z,h), Q2 (z,h);  actor role: policy (z,h), Q1 z, Q2 z
  const int o_sn = (int)(s_sn - smem), o_sn2 = (int)(s_sn2 - smem), o_qin = (int)(s_qin - smem), o_qin2 = (int)(s_qin2 - smem);
  const int o_pp = (int)(s_pp - smem), o_y = (int)(s_y - smem), o_dy = (int)(s_dy - smem), o_dx = (int)(s_dx - smem);
  const int o_st0 = (int)(s_store - smem), o_st1 = o_st0 + LH * T, o_st2 = o_st0 + 2 * LH * T, o_st3 = o_st0 + 3 * LH * T;

  // Phases (one barrier per layer step inside a phase, one after each elementwise section E):
  //   critic role:  E load | F0: c0 pi(s') fwd, c1/c2 Q1/Q2(s,a) fwd | E sample a' | F1: c0/c1 Qtgt1/2(s',a') fwd | E targets
  //                 | B2: c0/c1 Q1/Q2 dgrad, c2/c3 Q1/Q2 wgrad | E loss partials
  //   actor role:   E load | F0: c0 pi(s) fwd | E sample a | F1: c0/c1 Q1/Q2(s,a) fwd | E dL/dq | B2: c0/c1 Q input-grad
  //                 | E dL/dlogits | B3: c0 pi dgrad, c1 pi wgrad | E loss partials
  // What this wave walks in the current phase (wave-uniform scalars; LDS operands as offsets from smem, -1 = none).
  WSet<HT, SP> R;
  int mode = CH_IDLE, netid = 0;
  const float *cparams = pi_p;
  float *cslab = nullptr;
  int cx = -1, cldx = ld_x, cpp0 = -1, cpp1 = -1, czb = -1, chb = -1, cy = -1, cdx = -1;
#define P(off) ((off) < 0 ? (float *)nullptr : smem + (off))
  const int nph = role == 0 ? 3 : 4;
#pragma nounroll
  for (int ph = -1; ph < nph; ++ph) {
    const int tid = opaque(tid_), lane = tid & 63;   // keeps per-lane addresses of all phases from being hoisted and spilled
    if (ph >= 0) {
      const int len = (role == 0) ? (ph == 0 ? Lmax : QL) : ((ph == 0 || ph == 3) ? PL : QL);
      const NetShape sh = netid == 0 ? A.sh_pi : A.sh_q;
      if (mode == CH_FWD)
        chain_fwd_run<HT, SP, WIDE>(sh, cparams, P(cx), cldx, P(cpp0), P(cpp1), P(czb), P(chb), P(cy), ld_y, ld_h, len, sub, lane, R,
                              (A.stamps && tile == 0 && role == 1 && ph == 0 && wave == 0) ? A.stamps + 40 : nullptr);
      else if (mode == CH_DGRAD)
        chain_dgrad_run<HT, SP, WIDE>(sh, cparams, P(cy), ld_y, P(czb), P(cpp0), P(cpp1), P(cdx), ld_xu, ld_h, len, sub, lane, R);
      else if (mode == CH_WGRAD)
        chain_wgrad_run<HT, SP, WIDE>(sh, P(cx), cldx, P(chb), P(cy), ld_y, P(cpp0), P(cpp1), cslab, false, ld_h, len, sub, lane,
                                (A.stamps && tile == 0 && role == 0 && sub == 0 && chain == 2) ? A.stamps + 48 : nullptr);
      else
        chain_idle_run(len);
    }
    SAC_STAMP(2 * ph + 3);
    // ---- the chain this wave walks in the NEXT phase; its first layer's weights are requested now ----
    {
      const int nx = ph + 1;
      const SacChainDesc &cd = A.tab[role][nx][chain];
      mode = cd.mode;
      netid = cd.netid;
      cparams = (cd.base_sel ? A.qt.params : A.pi.params) + cd.param_off;
      cx = cd.x; cldx = cd.ldx; cpp0 = cd.pp0; cpp1 = cd.pp1; czb = cd.zb; chb = cd.hb; cy = cd.y; cdx = cd.dx;
      cslab = cd.slab_sel == 1 ? A.slab_pi + (long long)tile * A.pi.n_params + cd.slab_off
                               : (cd.slab_sel == 2 ? A.slab_q + (long long)tile * (2 * A.q.n_params) + cd.slab_off : nullptr);
      const NetShape shn = netid == 0 ? A.sh_pi : A.sh_q;
      if (mode == CH_FWD) chain_fwd_prefetch<HT, SP, WIDE>(R, shn, cparams, sub, lane);
      else if (mode == CH_DGRAD) chain_dgrad_prefetch<HT, SP, WIDE>(R, shn, cparams, sub, lane);
    }
    if (A.stamps && tile == 0 && tid == 0 && ph == 1) {
      unsigned long long t_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
      A.stamps[role * 16 + 14] = t_;
    }
    // ---- elementwise section after phase ph ----
    if (ph == -1) {
      // load the tile's transitions; normalise observations (q and policy both preprocess obs: sac/networks.py:76-78,96-98).
      // Every global load of this section is independent: one latency, one barrier.
      const long long nvalid = (long long)(B - row0 < 16 ? B - row0 : 16) * D;
      for (int idx = tid; idx < 16 * D; idx += nthreads) s_row[idx] = idx < nvalid ? A.batch[(long long)row0 * D + idx] : 0.f;
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        const int r = idx & 15, cc = idx >> 4;
        float o = 0.f, o2 = 0.f;
        if (row0 + r < B) {
          o = A.batch[(long long)(row0 + r) * D + cc];
          o2 = A.batch[(long long)(row0 + r) * D + X + U + 2 + cc];
        }
        if (A.norm_mean) {
          const float mu = A.norm_mean[cc], sd = A.norm_std[cc];
          o = (o - mu) / sd;
          o2 = (o2 - mu) / sd;
        }
        s_sn[r * ld_x + cc] = o;
        s_sn2[r * ld_x + cc] = o2;
        s_qin[r * ld_xu + cc] = o;
        s_qin2[r * ld_xu + cc] = o2;
      }
      if (role == 0) {
        for (int idx = tid; idx < 16 * U; idx += nthreads) {
          const int r = idx & 15, d = idx >> 4;
          s_qin[r * ld_xu + X + d] = (row0 + r < B) ? A.batch[(long long)(row0 + r) * D + X + d] : 0.f;  // transitions.action
        }
      }
    }
    const float alpha = expf(log_alpha_v);
    const unsigned long long rng_off = rk_.offset, rng_seed = rk_.seed;
    if (ph == -1) {
    } else if (role == 0) {
      // ============================== CRITIC (sac/losses.py:74-110) ==============================
      if (ph == 0) {
        // next_action ~ policy(next_observation); next_log_prob (:80-87)
        for (int i2 = tid; i2 < 16 * U; i2 += nthreads) {
          const int r = i2 & 15, d = i2 >> 4, idx = r * U + d;
          long long nidx = (long long)(row0 + r) * U + d;
          float eps = 0.f;
          if (row0 + r < B)
            eps = A.noise_critic ? A.noise_critic[nidx] : philox_normal(rng_seed, rng_off, MBPO_STREAM_SAC_CRITIC, (unsigned long long)nidx);
          ActSample sm = normal_tanh_sample(y_pi[r * ld_y + d], y_pi[r * ld_y + U + d], eps);
          s_qin2[r * ld_xu + X + d] = sm.a;  // postprocess(next_action)
          s_lp[idx] = sm.lp;
        }
        if (tid < 32) {  // keep q_old_action (:78-79) out of the way of the target critics' outputs
          const int k = tid >> 4, r = tid & 15;
          s_scal[32 + tid] = (k == 0 ? y_q1 : y_q2)[r * ld_y];
        }
      } else if (ph == 1) {
        if (tid < 32) {
          const int k = tid >> 4, r = tid & 15;
          const bool ok = row0 + r < B;
          float nlp = 0.f;
          for (int d = 0; d < U; ++d) nlp += s_lp[r * U + d];
          const float nq = fminf(y_q1[r * ld_y], y_q2[r * ld_y]);
          const float next_v = nq - alpha * nlp;                                                   // :89
          const float rew = s_row[r * D4 + X + U], disc = s_row[r * D4 + X + U + 1];
          float gamma = A.discounting;
          if (A.neq) {                                                                             // :90-96
            const float pseudo = s_row[r * D4 + X + U - 1];                                        // transitions.action[..., -1]
            float tfa = (A.neq_tu - A.neq_tl) / 2.0f * pseudo + (A.neq_tu + A.neq_tl) / 2.0f;
            tfa = floor_divide_f(tfa, A.neq_dt) * A.neq_dt;
            gamma = expf(-A.neq_cd * tfa);
          }
          const float target = rew * A.reward_scaling + disc * gamma * next_v;                     // :101-103
          const float trunc = s_row[r * D4 + D - 1];
          const float err = ok ? (s_scal[32 + tid] - target) * (1.f - trunc) : 0.f;               // q_error :104-108
          s_scal[tid] = err * err;
          // loss = 0.5*mean(err^2) over [B,2]  ->  dL/dq = err*(1-trunc)/(2B)
          s_dy[(k * 16 + r) * ld_y] = err * (1.f - trunc) * (0.5f * invB);
        }
      } else {
        if (tid == 0) {
          float acc = 0.f;
          for (int i = 0; i < 32; ++i) acc += s_scal[i];
          A.slab_ex[tile * 4 + 0] = acc;
        }
      }
    } else {
      // ============================== ACTOR + ALPHA (sac/losses.py:61-72, 112-125) ==============================
      if (ph == 0) {
        // two independent samples per element: the first half of the workgroup draws the actor-loss sample, the second half
        // the alpha-loss sample (different waves: the two instruction streams run side by side)
        const int half = nthreads / 2;
        const bool second = tid >= half;
        for (int i2 = second ? tid - half : tid; i2 < 16 * U; i2 += half) {
          const int r = i2 & 15, d = i2 >> 4, idx = r * U + d;
          long long nidx = (long long)(row0 + r) * U + d;
          const bool ok = row0 + r < B;
          const float loc = y_pi[r * ld_y + d], raw = y_pi[r * ld_y + U + d];
          if (second) {
            float e_al = 0.f;
            if (ok) e_al = A.noise_alpha ? A.noise_alpha[nidx] : philox_normal(rng_seed, rng_off, MBPO_STREAM_SAC_ALPHA, (unsigned long long)nidx);
            ActSample sal = normal_tanh_sample(loc, raw, e_al);   // alpha loss sample (:66-68)
            s_lpa[idx] = sal.lp;
          } else {
            float e_ac = 0.f;
            if (ok) e_ac = A.noise_actor ? A.noise_actor[nidx] : philox_normal(rng_seed, rng_off, MBPO_STREAM_SAC_ACTOR, (unsigned long long)nidx);
            ActSample sac = normal_tanh_sample(loc, raw, e_ac);   // actor loss sample (:117-119)
            s_lp[idx] = sac.lp;
            s_eps[idx] = e_ac;
            s_a[idx] = sac.a;
            s_sig[idx] = sac.sigma;
            s_raw[idx] = raw;
            s_qin[r * ld_xu + X + d] = sac.a;  // postprocess(action) (:120)
          }
        }
      } else if (ph == 1) {
        if (tid < 16) {
          const int r = tid;
          const bool ok = row0 + r < B;
          float lp_al = 0.f, lp_ac = 0.f;
          for (int d = 0; d < U; ++d) {
            lp_al += s_lpa[r * U + d];
            lp_ac += s_lp[r * U + d];
          }
          // alpha_loss = alpha * stop_gradient(-log_prob - target_entropy); d/dlog_alpha = the same value   (:70-72)
          s_scal[r] = ok ? alpha * (-lp_al - A.target_entropy) : 0.f;
          const float q0 = y_q1[r * ld_y], q1 = y_q2[r * ld_y];
          const float mq = fminf(q0, q1);
          s_scal[32 + r] = ok ? (alpha * lp_ac - mq) : 0.f;   // actor_loss = alpha*log_prob - min_q   (:123-124)
          // d(mean(-min_q))/dq_k: -1/B on the arg-min critic (ties split evenly, as jnp.min's gradient does)
          float g0 = 0.f, g1 = 0.f;
          if (ok) {
            if (q0 < q1) g0 = -invB;
            else if (q1 < q0) g1 = -invB;
            else g0 = g1 = -0.5f * invB;
          }
          s_dy[r * ld_y] = g0;
          s_dy[(16 + r) * ld_y] = g1;
        }
      } else if (ph == 2) {
        for (int i2 = tid; i2 < 16 * U; i2 += nthreads) {
          const int r = i2 & 15, d = i2 >> 4, idx = r * U + d;
          const bool ok = row0 + r < B;
          const float a = s_a[idx], sg = s_sig[idx], eps = s_eps[idx], raw = s_raw[idx];
          const float dLda = s_dx[r * ld_xu + X + d] + s_dx[(16 + r) * ld_xu + X + d];
          // z = loc + sigma*eps;  log_prob = const - log(sigma) - log(1 - tanh(z)^2)  =>  dlp/dz = 2a, dlp/dsigma = -1/sigma
          const float gz = dLda * (1.f - a * a) + alpha * invB * 2.f * a;
          const float gsig = gz * eps - alpha * invB / sg;
          s_dy[r * ld_y + d] = ok ? gz : 0.f;                              // d/dloc
          s_dy[r * ld_y + U + d] = ok ? gsig * fast_sigmoid(raw) : 0.f;    // d/draw = d/dsigma * softplus'(raw)
        }
      } else {
        if (tid == 0) {
          float al = 0.f, ac = 0.f;
          for (int i = 0; i < 16; ++i) {
            al += s_scal[i];
            ac += s_scal[32 + i];
          }
          A.slab_ex[tile * 4 + 1] = ac;
          A.slab_ex[tile * 4 + 2] = al;
        }
      }
    }
    __syncthreads();
    SAC_STAMP(2 * ph + 4);
  }
}

// ------------------------------------------------------------------------------------------------
struct SacReduceArgs {
  const float *slab_pi, *slab_q, *slab_ex;
  int n_tiles, P, Q2, B;
  float *grads, *metrics, *metrics_accum, *ss_part, *step_count;
};

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum-of-squares partials per workgroup and optimizer group (0 policy, 1 critics, 2 alpha), fixed order
__device__ __forceinline__ void group_sumsq(float g, int i, int P, int Q2, int NP, float *ss_part) {
  __shared__ float s_ss[3][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float gg = (i < NP) ? g * g : 0.f;
  const int grp = (i < P) ? 0 : (i < P + Q2 ? 1 : 2);
  float v0 = wave_sum64(grp == 0 ? gg : 0.f), v1 = wave_sum64(grp == 1 ? gg : 0.f), v2 = wave_sum64(grp == 2 ? gg : 0.f);
  if (lane == 0) {
    s_ss[0][wave] = v0;
    s_ss[1][wave] = v1;
    s_ss[2][wave] = v2;
  }
  __syncthreads();
  if (tid < 3) ss_part[blockIdx.x * 3 + tid] = s_ss[tid][0] + s_ss[tid][1] + s_ss[tid][2] + s_ss[tid][3];
}

__global__ void __launch_bounds__(256) k_sac_reduce(SacReduceArgs A) {
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float g = 0.f;
  if (i < A.P) {
    g = slab_sum<16>(A.slab_pi, A.P, A.n_tiles, i);
  } else if (i < A.P + A.Q2) {
    const int j = i - A.P;
    g = slab_sum<16>(A.slab_q, A.Q2, A.n_tiles, j);
  } else if (i == NP - 1) {
    // (all loads of the three sums in flight together: this one thread is the kernel's critical path)
    const float ce = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 0), ac = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 1),
                al = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 2);
    const float invB = 1.0f / (float)A.B;
    g = al * invB;                          // d alpha_loss / d log_alpha
    A.metrics[0] = 0.5f * ce * (0.5f * invB);  // critic_loss = 0.5 * mean over [B,2]
    A.metrics[1] = ac * invB;
    A.metrics[2] = al * invB;
    if (A.metrics_accum) {
      A.metrics_accum[0] += A.metrics[0];
      A.metrics_accum[1] += A.metrics[1];
      A.metrics_accum[2] += A.metrics[2];
      A.metrics_accum[4] += 1.0f;
    }
    A.step_count[0] = A.step_count[0] + 1.0f;  // optimizer count (read by apply; fwd_bwd of this step already ran)
  }
  if (i < NP) A.grads[i] = g;
  group_sumsq(g, i, A.P, A.Q2, NP, A.ss_part);
}

// multi-GPU: the reduced local gradient goes straight into every rank's exchange region (p2p.hpp) instead of a collective
__global__ void __launch_bounds__(256) k_sac_reduce_push(SacReduceArgs A, P2pDev X) {
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned epoch = X.epoch[0];
  float g = 0.f;
  if (i < A.P) {
    g = slab_sum<16>(A.slab_pi, A.P, A.n_tiles, i);
  } else if (i < A.P + A.Q2) {
    const int j = i - A.P;
    g = slab_sum<16>(A.slab_q, A.Q2, A.n_tiles, j);
  } else if (i == NP - 1) {
    // (all loads of the three sums in flight together: this one thread is the kernel's critical path)
    const float ce = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 0), ac = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 1),
                al = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 2);
    const float invB = 1.0f / (float)A.B;
    g = al * invB;
    A.metrics[0] = 0.5f * ce * (0.5f * invB);
    A.metrics[1] = ac * invB;
    A.metrics[2] = al * invB;
    if (A.metrics_accum) {
      A.metrics_accum[0] += A.metrics[0];
      A.metrics_accum[1] += A.metrics[1];
      A.metrics_accum[2] += A.metrics[2];
      A.metrics_accum[4] += 1.0f;
    }
    A.step_count[0] = A.step_count[0] + 1.0f;
  }
  p2p_push(X, epoch, i, NP, g);
}

// multi-GPU: wait for every rank's gradient, add the world slots in rank order, form the clip-norm partials
__global__ void __launch_bounds__(256) k_sac_gather(P2pDev X, float *grads, int P, int Q2, float *ss_part) {
  const int NP = P + Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned epoch = X.epoch[0];
  const bool ok = p2p_wait(X, X.epoch[1]);
  float g = 0.f;
  if (i < NP) {
    g = ok ? p2p_sum(X, epoch, i) : NAN;
    grads[i] = g;
  }
  group_sumsq(g, i, P, Q2, NP, ss_part);
}

// multi-GPU, one launch: local slab reduction -> stores into every rank's region -> wait for every rank's arrivals -> sum of the
// world slots in rank order -> clip-norm partials.  All workgroups of the grid (NP/256 ~ 100) are co-resident, so waiting inside
// the producing kernel for the peers' same kernel cannot deadlock; compared with k_sac_reduce_push + k_sac_gather this removes a
// kernel boundary from every sgd_step and leaves the xGMI store latency as the only cost of the exchange.
__global__ void __launch_bounds__(256) k_sac_reduce_exchange(SacReduceArgs A, P2pDev X) {
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned epoch = X.epoch[0], want = X.epoch[1];
  float g = 0.f;
  if (i < A.P) {
    g = slab_sum<16>(A.slab_pi, A.P, A.n_tiles, i);
  } else if (i < A.P + A.Q2) {
    const int j = i - A.P;
    g = slab_sum<16>(A.slab_q, A.Q2, A.n_tiles, j);
  } else if (i == NP - 1) {
    const float ce = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 0), ac = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 1),
                al = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 2);
    const float invB = 1.0f / (float)A.B;
    g = al * invB;
    A.metrics[0] = 0.5f * ce * (0.5f * invB);
    A.metrics[1] = ac * invB;
    A.metrics[2] = al * invB;
    if (A.metrics_accum) {
      A.metrics_accum[0] += A.metrics[0];
      A.metrics_accum[1] += A.metrics[1];
      A.metrics_accum[2] += A.metrics[2];
      A.metrics_accum[4] += 1.0f;
    }
    A.step_count[0] = A.step_count[0] + 1.0f;
  }
  p2p_push(X, epoch, i, NP, g);
  const bool ok = p2p_wait(X, want);
  float gs = 0.f;
  if (i < NP) {
    gs = ok ? p2p_sum(X, epoch, i) : NAN;
    A.grads[i] = gs;
  }
  group_sumsq(gs, i, A.P, A.Q2, NP, A.ss_part);
}

__global__ void __launch_bounds__(256) k_sac_sumsq(const float *grads, int P, int Q2, float *ss_part) {
  const int NP = P + Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float g = i < NP ? grads[i] : 0.f;
  group_sumsq(g, i, P, Q2, NP, ss_part);
}

struct SacApplyArgs {
  float *params, *target_q, *adam_m, *adam_v, *grads, *metrics, *metrics_accum;
  const float *step_count, *ss_part;
  int n_parts, P, Q2;
  float lr[3], wd[3];
  float max_norm, tau, one_minus_tau, grad_scale;
};

__global__ void __launch_bounds__(256) k_sac_apply(SacApplyArgs A) {
  __shared__ float s_scale[3];
  __shared__ float s_corr[2];
  const int tid = threadIdx.x;
  // the element's own operands are requested first: their latency overlaps the norm reduction below
  const int NP_ = A.P + A.Q2 + 1;
  const int i_ = blockIdx.x * 256 + tid;
  const bool in_ = i_ < NP_;
  const float g_in = in_ ? A.grads[i_] : 0.f, m_in = in_ ? A.adam_m[i_] : 0.f, v_in = in_ ? A.adam_v[i_] : 0.f,
              p_in = in_ ? A.params[i_] : 0.f;
  const float count_in = A.step_count[0];
  const bool crit_ = in_ && i_ >= A.P && i_ < A.P + A.Q2;
  const float tq_in = crit_ ? A.target_q[i_ - A.P] : 0.f;
  {
    // wave w < 3 reduces optimizer group w's sum-of-squares partials (fixed shuffle tree -> deterministic)
    const int w = tid >> 6, lane = tid & 63;
    if (w < 3) {
      float ss = 0.f;
      for (int p = lane; p < A.n_parts; p += 64) ss += A.ss_part[p * 3 + w];
      ss = wave_sum64(ss);
      // [3P optax.clip_by_global_norm] g_norm = sqrt(sum g^2); g <- g if g_norm < max_norm else (g / g_norm) * max_norm
      if (lane == 0) s_scale[w] = sqrtf(ss) * A.grad_scale;
    } else if (lane == 0) {
      // the Adam bias corrections are the same for every element: the fourth wave forms them (two powf, ~300 instructions)
      // beside the three norm reductions instead of every wave after the barrier
      s_corr[0] = 1.f - powf(0.9f, count_in);
      s_corr[1] = 1.f - powf(0.999f, count_in);
    }
  }
  __syncthreads();
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + tid;
  if (i >= NP) return;
  const int grp = (i < A.P) ? 0 : (i < A.P + A.Q2 ? 1 : 2);
  const float gnorm = s_scale[grp];
  float g = g_in * A.grad_scale;
  if (!(gnorm < A.max_norm)) g = (g / gnorm) * A.max_norm;
  // [3P optax.adamw] scale_by_adam(b1=.9,b2=.999,eps=1e-8) -> add_decayed_weights(wd) -> scale(-lr)
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  // (count_in is already incremented for this step)  optax forms (1 - decay) in Python double and only then casts: f32(0.1), f32(0.001) — not 1.f - 0.999f
  const float mu = b1 * m_in + 0.1f * g;
  const float nu = b2 * v_in + 0.001f * (g * g);
  A.adam_m[i] = mu;
  A.adam_v[i] = nu;
  const float mu_hat = mu / s_corr[0];   // 1 - b1^count, 1 - b2^count
  const float nu_hat = nu / s_corr[1];
  float upd = mu_hat / (sqrtf(nu_hat) + eps);
  const float p = p_in;
  upd = upd + A.wd[grp] * p;
  const float pn = p + (-A.lr[grp]) * upd;  // optax.apply_updates: p + u, u = -lr * upd
  A.params[i] = pn;
  if (grp == 1) {
    const int j = i - A.P;
    A.target_q[j] = tq_in * A.one_minus_tau + pn * A.tau;           // sac.py:260-261 ((1 - tau) formed in double on the host)
  } else if (grp == 2) {
    A.metrics[3] = expf(pn);                                      // 'alpha': exp(alpha_params) (sac.py:267)
    if (A.metrics_accum) A.metrics_accum[3] += A.metrics[3];
  }
}

// ------------------------------------------------------------------------------------------------ host side

// ------------------------------------------------------------------------------------------------
// Single-rank fast path: k_sac_reduce + k_sac_apply in ONE launch (a kernel boundary costs ~4.5 us here, the two kernels'
// work ~2 us).  The global gradient norms need every block's partial before any block applies, so the blocks meet at a
// device-scope arrival counter (all n_red <= ~1000 blocks of 256 threads are co-resident on 256 CUs).  The spin is bounded:
// if the barrier is not reached the update is skipped and metrics[0] is set to NaN instead of hanging the GPU.
struct SacFusedArgs {
  SacReduceArgs R;
  SacApplyArgs Ap;
  unsigned int *sync;   // [0] arrivals at the norm barrier, [1] arrivals at the end (both return to 0)
  float *step_count;
};

__global__ void __launch_bounds__(256) k_sac_reduce_apply(SacFusedArgs F) {
  __shared__ float s_scale[3];
  __shared__ int s_ok;
  const SacReduceArgs &A = F.R;
  const SacApplyArgs &Ap = F.Ap;
  const int tid = threadIdx.x;
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + tid;
  const int nblocks = gridDim.x;
  const float count = F.step_count[0] + 1.0f;   // every block reads the old count before it arrives anywhere
  float g = 0.f;
  if (i < A.P) {
    g = slab_sum<16>(A.slab_pi, A.P, A.n_tiles, i);
  } else if (i < A.P + A.Q2) {
    const int j = i - A.P;
    g = slab_sum<16>(A.slab_q, A.Q2, A.n_tiles, j);
  } else if (i == NP - 1) {
    // (all loads of the three sums in flight together: this one thread is the kernel's critical path)
    const float ce = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 0), ac = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 1),
                al = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 2);
    const float invB = 1.0f / (float)A.B;
    g = al * invB;
    A.metrics[0] = 0.5f * ce * (0.5f * invB);
    A.metrics[1] = ac * invB;
    A.metrics[2] = al * invB;
    if (A.metrics_accum) {
      A.metrics_accum[0] += A.metrics[0];
      A.metrics_accum[1] += A.metrics[1];
      A.metrics_accum[2] += A.metrics[2];
      A.metrics_accum[4] += 1.0f;
    }
  }
  if (i < NP) A.grads[i] = g;
  group_sumsq(g, i, A.P, A.Q2, NP, A.ss_part);   // writes this block's three partials (ends in a __syncthreads + store by tid < 3)
  __syncthreads();
  // ---- device-wide meeting point ----
  if (tid == 0) {
    __threadfence();                                                               // partials visible device-wide
    __hip_atomic_fetch_add(&F.sync[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 0;
    for (int spin = 0; spin < (1 << 22); ++spin) {
      if (__hip_atomic_load(&F.sync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)nblocks) {
        ok = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __threadfence();
    s_ok = ok;
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  {
    const int w = tid >> 6, lane = tid & 63;
    if (w < 3) {
      float ss = 0.f;
      for (int p = lane; p < nblocks; p += 64) ss += __hip_atomic_load(&A.ss_part[p * 3 + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ss = wave_sum64(ss);
      if (lane == 0) s_scale[w] = sqrtf(ss) * Ap.grad_scale;
    }
  }
  __syncthreads();
  if (ok && i < NP) {
    const int grp = (i < A.P) ? 0 : (i < A.P + A.Q2 ? 1 : 2);
    const float gnorm = s_scale[grp];
    g = g * Ap.grad_scale;
    if (!(gnorm < Ap.max_norm)) g = (g / gnorm) * Ap.max_norm;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float mu = b1 * Ap.adam_m[i] + 0.1f * g;
    const float nu = b2 * Ap.adam_v[i] + 0.001f * (g * g);
    Ap.adam_m[i] = mu;
    Ap.adam_v[i] = nu;
    const float mu_hat = mu / (1.f - powf(b1, count));
    const float nu_hat = nu / (1.f - powf(b2, count));
    float upd = mu_hat / (sqrtf(nu_hat) + eps);
    const float p = Ap.params[i];
    upd = upd + Ap.wd[grp] * p;
    const float pn = p + (-Ap.lr[grp]) * upd;
    Ap.params[i] = pn;
    if (grp == 1) {
      const int j = i - A.P;
      Ap.target_q[j] = Ap.target_q[j] * Ap.one_minus_tau + pn * Ap.tau;
    } else if (grp == 2) {
      Ap.metrics[3] = expf(pn);
      if (Ap.metrics_accum) Ap.metrics_accum[3] += Ap.metrics[3];
    }
  }
  // ---- last block out advances the optimizer count and re-arms the counters ----
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(&F.sync[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (unsigned)nblocks - 1u) {
      if (ok) F.step_count[0] = count;
      else A.metrics[0] = NAN;
      __hip_atomic_store(&F.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&F.sync[1], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct SacPlan {
  MlpDev pi, q, qt;
  int P, Q, NP, n_tiles, H, LH, n_red;
  size_t lds;
  int ld_x, ld_xu, ld_h, ld_y;
  // workspace offsets (floats)
  long long off_slab_pi, off_slab_q, off_slab_ex, off_ss, off_sync, total;
};

static int same_hidden(const int *dims, int n_layers) {
  if (n_layers < 2) return -1;
  for (int l = 2; l < n_layers; ++l)
    if (dims[l] != dims[1]) return -1;
  return dims[1];
}

static int sac_plan(const mbpo_sac_desc *d, SacPlan *pl, bool need_ptrs) {
  MBPO_REQUIRE(d, MBPO_ERR_ARG, "sac: null descriptor");
  MBPO_REQUIRE(d->x_dim > 0 && d->u_dim > 0 && d->batch_size > 0, MBPO_ERR_ARG, "sac: x_dim/u_dim/batch_size must be positive");
  MBPO_REQUIRE(d->row_len == 2 * d->x_dim + d->u_dim + 3, MBPO_ERR_ARG, "sac: row_len %d != 2x+u+3", d->row_len);
  MBPO_REQUIRE(d->policy_layers >= 2 && d->policy_layers <= MBPO_MAX_LAYERS && d->q_layers >= 2 && d->q_layers <= MBPO_MAX_LAYERS,
               MBPO_ERR_ARG, "sac: networks need at least one hidden layer and at most %d Dense layers", MBPO_MAX_LAYERS);
  MBPO_REQUIRE(d->policy_dims[0] == d->x_dim && d->policy_dims[d->policy_layers] == 2 * d->u_dim, MBPO_ERR_ARG,
               "sac: policy must map [x_dim] -> [2*u_dim]");
  MBPO_REQUIRE(d->q_dims[0] == d->x_dim + d->u_dim && d->q_dims[d->q_layers] == 1, MBPO_ERR_ARG,
               "sac: critic must map [x_dim+u_dim] -> [1]");
  const int Hp = same_hidden(d->policy_dims, d->policy_layers), Hq = same_hidden(d->q_dims, d->q_layers);
  MBPO_REQUIRE(Hp == Hq && (Hp == 64 || Hp == 128), MBPO_ERR_UNSUPPORTED,
               "sac: policy and critic hidden layers must share one width in {64,128} (got %d, %d)", Hp, Hq);
  mbpo_mlp_desc md;
  md.net_stride = 0;
  // policy
  md.params = d->params ? d->params : (const float *)16;  // placeholder for size queries
  md.n_nets = 1;
  md.n_layers = d->policy_layers;
  for (int l = 0; l <= d->policy_layers; ++l) md.dims[l] = d->policy_dims[l];
  md.activation = d->policy_activation;
  int rc = mbpo_make_mlp_dev(&md, &pl->pi, "sac.policy");
  if (rc != MBPO_OK) return rc;
  pl->P = pl->pi.n_params;
  // critics
  md.n_layers = d->q_layers;
  for (int l = 0; l <= d->q_layers; ++l) md.dims[l] = d->q_dims[l];
  md.activation = d->q_activation;
  md.n_nets = 1;
  rc = mbpo_make_mlp_dev(&md, &pl->q, "sac.q");
  if (rc != MBPO_OK) return rc;
  pl->Q = pl->q.n_params;
  pl->q.n_nets = 2;
  pl->q.net_stride = pl->Q;
  pl->q.params = d->params ? d->params + pl->P : nullptr;
  pl->qt = pl->q;
  pl->qt.params = d->target_q;
  pl->NP = pl->P + 2 * pl->Q + 1;
  pl->H = Hp;
  const int lhp = d->policy_layers - 1, lhq = d->q_layers - 1;
  pl->LH = lhp > lhq ? lhp : lhq;
  pl->n_tiles = (d->batch_size + 15) / 16;
  pl->n_red = (pl->NP + 255) / 256;
  auto up4 = [](int v) { return (v + 3) & ~3; };
  pl->ld_x = up4(d->x_dim) + 4;
  pl->ld_xu = up4(d->x_dim + d->u_dim) + 4;
  pl->ld_h = Hp + 4;
  pl->ld_y = up4(2 * d->u_dim) + 4;
  const int U = d->u_dim;
  size_t f = 16ull * up4(d->row_len) + 2ull * 16 * pl->ld_x + 2ull * 16 * pl->ld_xu + 4ull * 16 * pl->ld_h +
             (size_t)pl->LH * 4 * 16 * pl->ld_h + 5ull * 16 * pl->ld_y + 2ull * 16 * pl->ld_xu + 6ull * up4(16 * U) + 64;
  pl->lds = f * sizeof(float);
  pl->off_slab_pi = 0;
  pl->off_slab_q = pl->off_slab_pi + (long long)pl->n_tiles * pl->P;
  pl->off_slab_ex = pl->off_slab_q + (long long)pl->n_tiles * 2 * pl->Q;
  pl->off_ss = pl->off_slab_ex + (long long)pl->n_tiles * 4;
  pl->off_sync = (pl->off_ss + (long long)pl->n_red * 3 + 3) & ~3LL;   // 2 x uint32 grid-barrier counters (zero between launches)
  pl->total = pl->off_sync + 4;
  if (need_ptrs) {
    MBPO_REQUIRE(d->params && d->target_q && d->adam_m && d->adam_v && d->step_count && d->grads && d->workspace && d->metrics,
                 MBPO_ERR_ARG, "sac: null state pointer");
  }
  return MBPO_OK;
}

extern "C" int64_t mbpo_sac_workspace_floats(const mbpo_sac_desc *d) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, false);
  if (rc != MBPO_OK) return rc;
  return pl.total;
}

constexpr int SP64 = 4;   // waves per chain at hidden width 64

// LDS offsets (floats) of the tiles the chains use; must mirror the carve at the top of k_sac_fwd_bwd
static void sac_chain_table(const SacPlan &pl, int D, SacArgs *A) {
  const int ld_x = pl.ld_x, ld_xu = pl.ld_xu, ld_h = pl.ld_h, ld_y = pl.ld_y, LH = pl.LH;
  const int T = 16 * ld_h;
  const int o_row = 0;
  const int o_sn = o_row + 16 * ((D + 3) & ~3), o_sn2 = o_sn + 16 * ld_x, o_qin = o_sn2 + 16 * ld_x, o_qin2 = o_qin + 16 * ld_xu;
  const int o_pp = o_qin2 + 16 * ld_xu, o_st0 = o_pp + 4 * T, o_y = o_st0 + 4 * LH * T, o_dy = o_y + 3 * 16 * ld_y, o_dx = o_dy + 2 * 16 * ld_y;
  const int o_st1 = o_st0 + LH * T, o_st2 = o_st0 + 2 * LH * T, o_st3 = o_st0 + 3 * LH * T;
  const int P = pl.P, Q = pl.Q;   // params = [policy | critic0 | critic1 | log_alpha]; target_q = [critic0 | critic1]
  for (int role = 0; role < 2; ++role)
    for (int ph = 0; ph < 5; ++ph)
      for (int c = 0; c < 4; ++c) {
        SacChainDesc d;
        memset(&d, 0, sizeof(d));
        d.mode = CH_IDLE;
        d.x = d.pp0 = d.pp1 = d.zb = d.hb = d.y = d.dx = -1;
        d.ldx = ld_x;
        const int net = c & 1;
        if (role == 0) {                       // critic role
          if (ph == 0 && c < 3) {              // pi(s') || Q1(s,a) || Q2(s,a)
            d.mode = CH_FWD;
            if (c == 0) {
              d.netid = 0; d.param_off = 0; d.x = o_sn2; d.ldx = ld_x; d.pp0 = o_pp; d.pp1 = o_pp + T; d.y = o_y;
            } else {
              d.netid = 1; d.param_off = P + (c - 1) * Q; d.x = o_qin; d.ldx = ld_xu;
              d.zb = c == 1 ? o_st0 : o_st2; d.hb = c == 1 ? o_st1 : o_st3; d.y = o_y + c * 16 * ld_y;
            }
          } else if (ph == 1 && c < 2) {       // target critics on (s', a')
            d.mode = CH_FWD;
            d.netid = 1; d.base_sel = 1; d.param_off = c * Q; d.x = o_qin2; d.ldx = ld_xu;
            d.pp0 = o_pp + 2 * c * T; d.pp1 = d.pp0 + T; d.y = o_y + (c + 1) * 16 * ld_y;
          } else if (ph == 2) {                // critics backward: dgrad (c < 2) beside wgrad
            d.mode = c < 2 ? CH_DGRAD : CH_WGRAD;
            d.netid = 1; d.param_off = P + net * Q; d.x = o_qin; d.ldx = ld_xu;
            d.pp0 = o_pp + 2 * net * T; d.pp1 = d.pp0 + T; d.zb = net ? o_st2 : o_st0; d.hb = net ? o_st3 : o_st1;
            d.y = o_dy + net * 16 * ld_y;
            d.slab_sel = 2; d.slab_off = net * Q;
          }
        } else {                               // actor + alpha role
          if (ph == 0 && c == 0) {
            d.mode = CH_FWD;
            d.netid = 0; d.param_off = 0; d.x = o_sn; d.ldx = ld_x; d.zb = o_st0; d.hb = o_st1; d.y = o_y;
          } else if (ph == 1 && c < 2) {
            d.mode = CH_FWD;
            d.netid = 1; d.param_off = P + c * Q; d.x = o_qin; d.ldx = ld_xu;
            d.pp0 = o_pp + 2 * c * T; d.pp1 = d.pp0 + T; d.zb = c == 0 ? o_st2 : o_st3; d.y = o_y + (c + 1) * 16 * ld_y;
          } else if (ph == 2 && c < 2) {
            d.mode = CH_DGRAD;
            d.netid = 1; d.param_off = P + net * Q; d.pp0 = o_pp + 2 * net * T; d.pp1 = d.pp0 + T; d.zb = net ? o_st3 : o_st2;
            d.y = o_dy + net * 16 * ld_y; d.dx = o_dx + net * 16 * ld_xu;
          } else if (ph == 3 && c < 2) {
            d.mode = c == 0 ? CH_DGRAD : CH_WGRAD;
            d.netid = 0; d.param_off = 0; d.x = o_sn; d.ldx = ld_x; d.pp0 = o_pp; d.pp1 = o_pp + T; d.zb = o_st0; d.hb = o_st1;
            d.y = o_dy; d.slab_sel = 1; d.slab_off = 0;
          }
        }
        A->tab[role][ph][c] = d;
      }
}

static int sac_grads_impl(const mbpo_sac_desc *d, int phase_mask, void *stream, const mbpo_p2p_desc *xd = nullptr,
                          bool exchange_in_reduce = false) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(d->batch, MBPO_ERR_ARG, "sac_grads: null batch");
  MBPO_REQUIRE((d->norm_mean == nullptr) == (d->norm_std == nullptr), MBPO_ERR_ARG, "sac_grads: norm_mean/norm_std mismatch");
  SacArgs A;
  A.pi = pl.pi; A.q = pl.q; A.qt = pl.qt;
  A.sh_pi = NetShape{pl.pi.dims[0], pl.pi.n_layers, pl.pi.dims[pl.pi.n_layers], pl.pi.act};
  A.sh_q = NetShape{pl.q.dims[0], pl.q.n_layers, pl.q.dims[pl.q.n_layers], pl.q.act};
  sac_chain_table(pl, d->row_len, &A);
  A.X = d->x_dim; A.U = d->u_dim; A.B = d->batch_size; A.D = d->row_len;
  A.batch = d->batch; A.norm_mean = d->norm_mean; A.norm_std = d->norm_std;
  A.log_alpha = d->params + pl.NP - 1;
  A.noise_alpha = d->noise_alpha; A.noise_critic = d->noise_critic; A.noise_actor = d->noise_actor;
  A.seed = d->seed; A.offset = d->offset; A.rng_dev = (const unsigned long long *)d->rng_dev;
  A.stamps = g_sac_stamps;
  P2pDev X;
  A.p2p_epoch = nullptr;
  A.p2p_blocks = (unsigned)pl.n_red;
  if (xd) {
    rc = mbpo_p2p_make_dev(xd, &X);
    if (rc != MBPO_OK) return rc;
    MBPO_REQUIRE(xd->n_max >= pl.NP, MBPO_ERR_ARG, "sac_grads_p2p: exchange regions hold %lld floats, the gradient has %d",
                 (long long)xd->n_max, pl.NP);
    A.p2p_epoch = X.epoch;
  }
  A.discounting = d->discounting; A.reward_scaling = d->reward_scaling; A.target_entropy = d->target_entropy;
  A.neq = d->non_equidistant_time; A.neq_cd = d->continuous_discounting; A.neq_tl = d->min_time_between_switches;
  A.neq_tu = d->max_time_between_switches; A.neq_dt = d->env_dt;
  MBPO_REQUIRE(!A.neq || A.neq_dt > 0.f, MBPO_ERR_ARG, "sac: non_equidistant_time needs env_dt > 0");
  A.slab_pi = d->workspace + pl.off_slab_pi; A.slab_q = d->workspace + pl.off_slab_q; A.slab_ex = d->workspace + pl.off_slab_ex;
  A.ld_x = pl.ld_x; A.ld_xu = pl.ld_xu; A.ld_h = pl.ld_h; A.ld_y = pl.ld_y; A.LH = pl.LH;
  hipStream_t st = (hipStream_t)stream;
  if (phase_mask & 1) {
    if (pl.H == 64) {
      if (net_is_wide(A.sh_pi) || net_is_wide(A.sh_q)) {
        rc = mbpo_ensure_lds<k_sac_fwd_bwd<64, SP64, true>>(pl.lds, "sac_grads");
        if (rc != MBPO_OK) return rc;
        hipLaunchKernelGGL((k_sac_fwd_bwd<64, SP64, true>), dim3(2 * pl.n_tiles), dim3(256 * SP64), pl.lds, st, A);
      } else {
        rc = mbpo_ensure_lds<k_sac_fwd_bwd<64, SP64, false>>(pl.lds, "sac_grads");
        if (rc != MBPO_OK) return rc;
        hipLaunchKernelGGL((k_sac_fwd_bwd<64, SP64, false>), dim3(2 * pl.n_tiles), dim3(256 * SP64), pl.lds, st, A);
      }
    } else {
      rc = mbpo_ensure_lds<k_sac_fwd_bwd<128, 2, false>>(pl.lds, "sac_grads");
      if (rc != MBPO_OK) return rc;
      hipLaunchKernelGGL((k_sac_fwd_bwd<128, 2, false>), dim3(2 * pl.n_tiles), dim3(512), pl.lds, st, A);
    }
  }
  if (!(phase_mask & 2)) {
    MBPO_CHECK_LAUNCH("sac_grads");
    return MBPO_OK;
  }
  SacReduceArgs R;
  R.slab_pi = A.slab_pi; R.slab_q = A.slab_q; R.slab_ex = A.slab_ex;
  R.n_tiles = pl.n_tiles; R.P = pl.P; R.Q2 = 2 * pl.Q; R.B = d->batch_size;
  R.grads = d->grads; R.metrics = d->metrics; R.metrics_accum = d->metrics_accum; R.ss_part = d->workspace + pl.off_ss; R.step_count = d->step_count;
  if (xd && exchange_in_reduce) hipLaunchKernelGGL(k_sac_reduce_exchange, dim3(pl.n_red), dim3(256), 0, st, R, X);
  else if (xd) hipLaunchKernelGGL(k_sac_reduce_push, dim3(pl.n_red), dim3(256), 0, st, R, X);
  else hipLaunchKernelGGL(k_sac_reduce, dim3(pl.n_red), dim3(256), 0, st, R);
  MBPO_CHECK_LAUNCH("sac_grads");
  return MBPO_OK;
}

extern "C" int mbpo_sac_grads_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream) {
  MBPO_REQUIRE(x, MBPO_ERR_ARG, "sac_grads_p2p: null exchange descriptor");
  return sac_grads_impl(d, 3, stream, x);
}

extern "C" int mbpo_sac_grads_exchange_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream) {
  MBPO_REQUIRE(x, MBPO_ERR_ARG, "sac_grads_exchange_p2p: null exchange descriptor");
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  // every workgroup of the reduction waits inside the kernel: they must all be resident at once
  MBPO_REQUIRE(pl.n_red <= 1024, MBPO_ERR_UNSUPPORTED, "sac_grads_exchange_p2p: %d workgroups cannot be assumed co-resident; use "
               "mbpo_sac_grads_p2p + mbpo_sac_gather_p2p", pl.n_red);
  return sac_grads_impl(d, 3, stream, x, true);
}

extern "C" int mbpo_sac_gather_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  P2pDev X;
  rc = mbpo_p2p_make_dev(x, &X);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(x->n_max >= pl.NP, MBPO_ERR_ARG, "sac_gather_p2p: exchange regions too small");
  hipLaunchKernelGGL(k_sac_gather, dim3(pl.n_red), dim3(256), 0, (hipStream_t)stream, X, d->grads, pl.P, 2 * pl.Q, d->workspace + pl.off_ss);
  MBPO_CHECK_LAUNCH("sac_gather_p2p");
  return MBPO_OK;
}

extern "C" int mbpo_sac_grads(const mbpo_sac_desc *d, void *stream) { return sac_grads_impl(d, 3, stream); }

extern "C" int mbpo_sac_grads_phase(const mbpo_sac_desc *d, int32_t phase_mask, void *stream) {
  MBPO_REQUIRE(phase_mask >= 1 && phase_mask <= 3, MBPO_ERR_ARG, "sac_grads_phase: phase_mask must be 1, 2 or 3");
  return sac_grads_impl(d, phase_mask, stream);
}

extern "C" int mbpo_sac_grad_norms(const mbpo_sac_desc *d, void *stream) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  hipLaunchKernelGGL(k_sac_sumsq, dim3(pl.n_red), dim3(256), 0, (hipStream_t)stream, (const float *)d->grads, pl.P, 2 * pl.Q,
                     d->workspace + pl.off_ss);
  MBPO_CHECK_LAUNCH("sac_grad_norms");
  return MBPO_OK;
}

extern "C" int mbpo_sac_apply(const mbpo_sac_desc *d, void *stream) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  SacApplyArgs A;
  A.params = d->params; A.target_q = d->target_q; A.adam_m = d->adam_m; A.adam_v = d->adam_v; A.grads = d->grads;
  A.metrics = d->metrics; A.metrics_accum = d->metrics_accum; A.step_count = d->step_count; A.ss_part = d->workspace + pl.off_ss;
  A.n_parts = pl.n_red; A.P = pl.P; A.Q2 = 2 * pl.Q;
  A.lr[0] = d->lr_policy; A.lr[1] = d->lr_q; A.lr[2] = d->lr_alpha;
  A.wd[0] = d->wd_policy; A.wd[1] = d->wd_q; A.wd[2] = d->wd_alpha;
  A.max_norm = d->max_grad_norm; A.tau = d->tau; A.one_minus_tau = (float)(1.0 - (double)d->tau); A.grad_scale = d->grad_scale;
  hipLaunchKernelGGL(k_sac_apply, dim3(pl.n_red), dim3(256), 0, (hipStream_t)stream, A);
  MBPO_CHECK_LAUNCH("sac_apply");
  return MBPO_OK;
}

extern "C" int mbpo_sac_reduce_apply(const mbpo_sac_desc *d, void *stream) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  SacFusedArgs F;
  F.R.slab_pi = d->workspace + pl.off_slab_pi; F.R.slab_q = d->workspace + pl.off_slab_q; F.R.slab_ex = d->workspace + pl.off_slab_ex;
  F.R.n_tiles = pl.n_tiles; F.R.P = pl.P; F.R.Q2 = 2 * pl.Q; F.R.B = d->batch_size;
  F.R.grads = d->grads; F.R.metrics = d->metrics; F.R.metrics_accum = d->metrics_accum; F.R.ss_part = d->workspace + pl.off_ss;
  F.R.step_count = d->step_count;
  SacApplyArgs &A = F.Ap;
  A.params = d->params; A.target_q = d->target_q; A.adam_m = d->adam_m; A.adam_v = d->adam_v; A.grads = d->grads;
  A.metrics = d->metrics; A.metrics_accum = d->metrics_accum; A.step_count = d->step_count; A.ss_part = d->workspace + pl.off_ss;
  A.n_parts = pl.n_red; A.P = pl.P; A.Q2 = 2 * pl.Q;
  A.lr[0] = d->lr_policy; A.lr[1] = d->lr_q; A.lr[2] = d->lr_alpha;
  A.wd[0] = d->wd_policy; A.wd[1] = d->wd_q; A.wd[2] = d->wd_alpha;
  A.max_norm = d->max_grad_norm; A.tau = d->tau; A.one_minus_tau = (float)(1.0 - (double)d->tau); A.grad_scale = d->grad_scale;
  F.sync = reinterpret_cast<unsigned int *>(d->workspace + pl.off_sync);
  F.step_count = d->step_count;
  MBPO_REQUIRE(pl.n_red <= 2048, MBPO_ERR_UNSUPPORTED, "sac_reduce_apply: %d blocks cannot be assumed co-resident; use grads + apply", pl.n_red);
  hipLaunchKernelGGL(k_sac_reduce_apply, dim3(pl.n_red), dim3(256), 0, (hipStream_t)stream, F);
  MBPO_CHECK_LAUNCH("sac_reduce_apply");
  return MBPO_OK;
}

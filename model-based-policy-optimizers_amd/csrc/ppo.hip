// ppo.hip — P1-P3: one PPO minibatch update (see include/mbpo_hip.h).
//
//   k_ppo_values    V(obs) on all B*T samples + the bootstrap V(next_obs[:, T-1]); also splits the row columns the
//                   GAE scan needs (truncation, termination, scaled reward) into [B,T] arrays.            [fp32 MFMA]
//   mbpo_gae_scan   vs, advantages (scan.hip; batch-major, wavefront-shuffle scan)                         [HBM]
//   k_moments_*     mean / population std of the advantages over the whole minibatch (two passes)         [HBM]
//   k_ppo_fwd_bwd   policy and value forward + hand-written backward of the clipped-surrogate / value / entropy losses;
//                   a workgroup walks many 16-sample tiles and ACCUMULATES its weight gradients into its own slab
//                   (8 waves: policy and value chains, 2 waves each; backward: dgrad + wgrad chains side by side).  [fp32 MFMA]
//   k_ppo_reduce    grads = fixed-order sum of the slabs; loss metrics.                                    [HBM]
//   k_ppo_apply     optax.adamw(lr, wd) (no clipping in this variant, ppo.py:128).                         [HBM]
// Algorithmic work per sample: 3*(2P + 2V) FLOP fwd+bwd (+2V for the value pre-pass), 4*(2x+2u+4) B of row data.
#include "common.hpp"
#include "chain_run.hpp"
#include "ppo_layered.hpp"
#include "ppo_lean.hpp"

#define LOG_SQRT_2PI 0.91893853320467274178f
#define LOG_2 0.69314718055994530942f
#define PPO_MOM_WGS 64

struct PpoArgs {
  MlpDev pi, v;
  NetShape sh_pi, sh_v;
  int X, U, B, T, D;
  const float *data, *norm_mean, *norm_std, *ent_noise;
  unsigned long long seed, offset;
  const unsigned long long *rng_dev;
  float entropy_cost, discounting, reward_scaling, gae_lambda, clip_eps;
  int normalize_advantage;
  // workspace pieces
  float *baseline, *boot, *trunc, *term, *rew, *vs, *adv, *mom, *slabs, *extras;
  int n_slabs;
  int ld_x, ld_h, ld_y, LH;
  int mom_inline;            // (unused)
  float *mom_part;
  int mom_parts;
  int vg_G;                  // k_ppo_values_gae: trajectories per workgroup; its workgroups leave {n, mean, M2} partials in mom_part
  float *step_count_rw;      // optax's count: bumped by block 0 of the FIRST launch of a minibatch_step (k_ppo_values), so that every
                             // later launch of the step — the reduce launch that also applies AdamW, or k_ppo_apply — reads the final value
};

// ------------------------------------------------------------------------------------------------ values pre-pass
template <int H>
__global__ void __launch_bounds__(256) k_ppo_values(PpoArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid_ = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int X = A.X, U = A.U, D = A.D, T = A.T;
  const long long M = (long long)A.B * T, rows = M + A.B;     // samples, then one bootstrap row per trajectory
  float *s_x = smem;                         // [16][ld_x]
  float *s_pp = s_x + 16 * A.ld_x;           // 2 hidden tiles
  float *s_y = s_pp + 2 * 16 * A.ld_h;       // [16][ld_y]
  const long long n_tiles = (rows + 15) >> 4;
  if (blockIdx.x == 0 && tid_ == 0) A.step_count_rw[0] = A.step_count_rw[0] + 1.0f;     // (nothing in this launch reads it)
#pragma nounroll
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tid = opaque(tid_), lane = tid & 63;
    const long long r0 = tile * 16;
    WSet<HT, 4> R;
    chain_fwd_prefetch<HT, 4>(R, A.sh_v, A.v.params, wave, lane);
    for (int idx = tid; idx < 16 * X; idx += 256) {
      const int r = idx & 15, c = idx >> 4;
      const long long i = r0 + r;
      float o = 0.f;
      if (i < M) o = A.data[i * D + c];                                                   // observation
      else if (i < rows) o = A.data[((i - M) * T + (T - 1)) * D + X + U + 2 + c];         // next_observation[-1]  (losses.py:84-85)
      if (A.norm_mean) o = (o - A.norm_mean[c]) / A.norm_std[c];
      s_x[r * A.ld_x + c] = o;
    }
    if (tid < 16) {
      const long long i = r0 + tid;
      if (i < M) {
        const float *row = A.data + i * D;
        const float tr = row[D - 1], disc = row[X + U + 1];
        A.trunc[i] = tr;
        A.term[i] = (1.f - disc) * (1.f - tr);                 // termination = (1 - discount) * (1 - truncation)   (:89)
        A.rew[i] = row[X + U] * A.reward_scaling;              // rewards = data.reward * reward_scaling             (:87)
      }
    }
    __syncthreads();
    chain_fwd_run<HT, 4>(A.sh_v, A.v.params, s_x, A.ld_x, s_pp, s_pp + 16 * A.ld_h, nullptr, nullptr, s_y, A.ld_y, A.ld_h, A.sh_v.L,
                         wave, lane, R);
    if (tid < 16) {
      const long long i = r0 + tid;
      if (i < M) A.baseline[i] = s_y[tid * A.ld_y];
      else if (i < rows) A.boot[i - M] = s_y[tid * A.ld_y];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ moments (two passes)
__device__ __forceinline__ float ppo_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------ values + GAE + moment partials
// One launch for k_ppo_values + mbpo_gae_scan + the advantage moments' first stage: a workgroup owns G whole trajectories (T samples +
// the bootstrap row each), runs the value net on their rows tile by tile (values stay in LDS), then one thread per trajectory walks
// compute_gae's recurrences backwards in the reference's own order (losses.py:150-184, as k_scan_time_major) and the first wave
// leaves {n, mean, M2} of the workgroup's advantages (two passes over LDS: exact).  k_ppo_fwd_bwd combines the partials (Chan et
// al.: M2 = sum M2_i + sum n_i (mean_i - mean)^2, fixed order).  Three launches and two kernel boundaries less per minibatch_step.
template <int H, int NC>     // NC tiles at a time, each on its own chain of 4 waves (a trajectory of T = 40 is 3 tiles: one chain latency, not three)
__global__ void __launch_bounds__(256 * NC) k_ppo_values_gae(PpoArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid_ = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int chain = wave >> 2, sub = wave & 3;
  const int X = A.X, U = A.U, D = A.D, T = A.T, R = T + 1, G = A.vg_G;
  const int per_chain = 16 * A.ld_x + 2 * 16 * A.ld_h + 16 * A.ld_y;
  float *s_x = smem + chain * per_chain;     // [16][ld_x]
  float *s_pp = s_x + 16 * A.ld_x;           // 2 hidden tiles
  float *s_y = s_pp + 2 * 16 * A.ld_h;       // [16][ld_y]
  float *s_val = smem + NC * per_chain;      // [G][R] values (the last of each row: the bootstrap)
  float *s_tr = s_val + ((G * R + 3) & ~3);  // [G][T] truncation | termination | scaled reward | advantages
  float *s_te = s_tr + ((G * T + 3) & ~3), *s_rw = s_te + ((G * T + 3) & ~3), *s_adv = s_rw + ((G * T + 3) & ~3);
  const long long b0 = (long long)blockIdx.x * G;
  const int g_here = (int)((A.B - b0 < G) ? A.B - b0 : G);
  const int rows = g_here * R;
  if (blockIdx.x == 0 && tid_ == 0) A.step_count_rw[0] = A.step_count_rw[0] + 1.0f;     // (nothing in this launch reads it)
#pragma nounroll
  for (int rr = 0; rr < rows; rr += 16 * NC) {
    const int tid = opaque(tid_), lane = tid & 63, ctid = tid & 255;
    const int r0 = rr + 16 * chain;
    const bool live = r0 < rows;             // (wave-uniform: a chain without a tile only keeps the barriers)
    WSet<HT, 4> Rw;
    if (live) {
      chain_fwd_prefetch<HT, 4>(Rw, A.sh_v, A.v.params, sub, lane);
      for (int idx = ctid; idx < 16 * X; idx += 256) {
        const int r = idx & 15, c = idx >> 4;
        const int lrow = r0 + r;
        float o = 0.f;
        if (lrow < rows) {
          const int g = lrow / R, t = lrow - g * R;
          const long long base = ((b0 + g) * T + (t < T ? t : T - 1)) * D;
          o = A.data[base + (t < T ? c : X + U + 2 + c)];                  // observation | next_observation[-1]  (losses.py:84-85)
          if (A.norm_mean) o = (o - A.norm_mean[c]) / A.norm_std[c];
        }
        s_x[r * A.ld_x + c] = o;
      }
      if (ctid < 16) {
        const int lrow = r0 + ctid;
        if (lrow < rows) {
          const int g = lrow / R, t = lrow - g * R;
          if (t < T) {
            const float *row = A.data + ((b0 + g) * T + t) * D;
            const float tr = row[D - 1], disc = row[X + U + 1];
            s_tr[g * T + t] = tr;
            s_te[g * T + t] = (1.f - disc) * (1.f - tr);             // termination = (1 - discount) * (1 - truncation)   (:89)
            s_rw[g * T + t] = row[X + U] * A.reward_scaling;          // rewards = data.reward * reward_scaling             (:87)
          }
        }
      }
    }
    __syncthreads();
    if (live)
      chain_fwd_run<HT, 4>(A.sh_v, A.v.params, s_x, A.ld_x, s_pp, s_pp + 16 * A.ld_h, nullptr, nullptr, s_y, A.ld_y, A.ld_h, A.sh_v.L,
                           sub, lane, Rw);
    else
      chain_idle_run(A.sh_v.L);
    if (live && ctid < 16 && r0 + ctid < rows) s_val[r0 + ctid] = s_y[ctid * A.ld_y];
    __syncthreads();
  }
  // compute_gae, one thread per trajectory, backwards (losses.py:150-184)
  if (tid_ < g_here) {
    const int g = tid_;
    const float boot = s_val[g * R + T];
    float acc = 0.f, v_next = boot, vs_next = boot;
    for (int t = T - 1; t >= 0; --t) {
      const float tr = s_tr[g * T + t], te = s_te[g * T + t], r = s_rw[g * T + t], v = s_val[g * R + t];
      const float m = 1.f - tr;
      const float g1 = A.discounting * (1.f - te);
      const float delta = (r + g1 * v_next - v) * m;        // :157-158
      acc = delta + g1 * m * A.gae_lambda * acc;            // :166
      const float vs = acc + v;                             // :176
      const float adv = (r + g1 * vs_next - v) * m;         // :181-182
      const long long i = (b0 + g) * T + t;
      A.vs[i] = vs;
      A.adv[i] = adv;
      s_adv[g * T + t] = adv;
      v_next = v;
      vs_next = vs;
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int n = g_here * T, lane = tid_ & 63;
    float a = 0.f;
    for (int i = lane; i < n; i += 64) a += s_adv[i];
    a = ppo_wave_sum(a);
    const float mean = __shfl(a, 0, 64) / (float)n;
    float q = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float dd = s_adv[i] - mean;
      q += dd * dd;
    }
    q = ppo_wave_sum(q);
    if (lane == 0) {
      A.mom_part[4 * blockIdx.x + 0] = (float)n;
      A.mom_part[4 * blockIdx.x + 1] = mean;
      A.mom_part[4 * blockIdx.x + 2] = q;
    }
  }
}

// PASS 0: partial[g] = sum x ; PASS 1: partial[g] = sum (x - mean)^2
template <int PASS>
__global__ void __launch_bounds__(256) k_moments_partial(const float *x, long long n, const float *mom, float *partial) {
  __shared__ float s_w[4];
  const float mean = PASS ? mom[0] : 0.f;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = x[i] - mean;
    acc += PASS ? d * d : d;
  }
  acc = ppo_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

template <int PASS>
__global__ void k_moments_final(const float *partial, int n_parts, long long n, float *mom) {
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int g = 0; g < n_parts; ++g) acc += partial[g];
    if (PASS == 0) mom[0] = acc / (float)n;                // mean
    else mom[1] = sqrtf(acc / (float)n);                   // population std (jnp.std)
  }
}

// minibatches up to PPO_MOM_FUSED_MAX elements: both passes in ONE workgroup of 1024 threads (3 kernel boundaries of ~4.5 us less
// per minibatch_step; the data is a few KB).  Fixed strides and a fixed shuffle/LDS tree: deterministic.
#define PPO_MOM_FUSED_MAX 65536
__global__ void __launch_bounds__(1024) k_moments_fused(const float *x, long long n, float *mom) {
  __shared__ float s_w[16];
  __shared__ float s_mean;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float acc = 0.f;
#pragma unroll 4
  for (long long i = tid; i < n; i += 1024) acc += x[i];
  acc = ppo_wave_sum(acc);
  if (lane == 0) s_w[w] = acc;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
    for (int k = 0; k < 16; ++k) a += s_w[k];
    s_mean = a / (float)n;
  }
  __syncthreads();
  const float mean = s_mean;
  acc = 0.f;
#pragma unroll 4
  for (long long i = tid; i < n; i += 1024) {
    const float d = x[i] - mean;
    acc += d * d;
  }
  acc = ppo_wave_sum(acc);
  __syncthreads();
  if (lane == 0) s_w[w] = acc;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
    for (int k = 0; k < 16; ++k) a += s_w[k];
    mom[0] = mean;
    mom[1] = sqrtf(a / (float)n);                          // population std (jnp.std)
  }
}

// ------------------------------------------------------------------------------------------------ loss fwd/bwd
// hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp; as sac.hip and rollout.hip): the loss section runs on a few
// lanes while the other waves of the workgroup wait at its barrier, so its instruction count is tile latency.  libm's
// expf / log1pf / logf / tanhf are 30-60 instructions each.
__device__ __forceinline__ float pp_fexp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
__device__ __forceinline__ float pp_flog(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float pp_fsoftplus(float x) { return fmaxf(x, 0.0f) + pp_flog(1.0f + pp_fexp(-fabsf(x))); }
__device__ __forceinline__ float pp_ftanh(float x) {
  const float e = pp_fexp(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
  return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}
// The advantage moments from the per-workgroup partials {n_i, mean_i, M2_i} of k_ppo_values_gae (Chan et al.: mean = sum n_i mean_i / M,
// M2 = sum M2_i + sum n_i (mean_i - mean)^2), one workgroup, fixed order.  A launch of its own: k_ppo_fwd_bwd<64,2> sits at 128 VGPRs
// with ~50 spilled, and ANY code added to it — this combine inlined, behind a barrier-free LDS hand-off, or behind a noinline call —
// moved the allocator's choices in its tile loop: 6-9 us per launch at every size (rocprofv3, round 3), more than this launch costs.
// (Finishing in the last workgroup of k_ppo_values_gae behind an arrival counter cost that kernel 7 us.)
__global__ void __launch_bounds__(256) k_ppo_moments_combine(const float *part, int n_parts, float M, float *mom) {
  __shared__ float s_cw[4];
  __shared__ float s_cm;
  const int tid = threadIdx.x;
  constexpr int NPT = 4;                      // <= 4 * CUs partials
  float pn[NPT], pm[NPT], pq[NPT];
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int p = tid + k * 256;
    const bool ok = p < n_parts;
    pn[k] = ok ? part[4 * p] : 0.f;
    pm[k] = ok ? part[4 * p + 1] : 0.f;
    pq[k] = ok ? part[4 * p + 2] : 0.f;
  }
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < NPT; ++k) a += pn[k] * pm[k];
  a = ppo_wave_sum(a);
  if ((tid & 63) == 0) s_cw[tid >> 6] = a;
  __syncthreads();
  if (tid == 0) s_cm = (((s_cw[0] + s_cw[1]) + s_cw[2]) + s_cw[3]) / M;
  __syncthreads();
  const float mean = s_cm;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const float dm = pm[k] - mean;
    q += pq[k] + pn[k] * dm * dm;
  }
  q = ppo_wave_sum(q);
  if ((tid & 63) == 0) s_cw[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) {
    mom[0] = mean;
    mom[1] = sqrtf((((s_cw[0] + s_cw[1]) + s_cw[2]) + s_cw[3]) / M);     // population std (jnp.std)
  }
}

// SP = 2 at H = 64 (512 threads, held to 128 VGPRs by the waves-per-SIMD request): TWO workgroups share a CU, so a tile's chain of
// dependent layer steps overlaps with another tile's — the launch is a latency chain per tile (MFMA pipes ~17 % busy with one
// 1024-thread workgroup per CU), not a throughput problem.  The host picks it when there are more tiles than CUs.
template <int H, int SP, bool WIDE>   // 4 chains x SP waves; WIDE: chain_run.hpp fast_shape
__global__ void __launch_bounds__(256 * SP, (H == 64 && SP == 2) ? 4 : 1) k_ppo_fwd_bwd(PpoArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid_ = threadIdx.x, nthreads = 256 * SP;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int chain = wave / SP, sub = wave % SP;
  const int X = A.X, U = A.U, D = A.D;
  const long long M = (long long)A.B * A.T;
  const int ld_x = A.ld_x, ld_h = A.ld_h, ld_y = A.ld_y, LH = A.LH;
  const int TT = 16 * ld_h;
  const int D4 = (D + 3) & ~3;
  float *s_row = smem;                       // [16][D] flat copy of the tile's rows (16*D4 floats reserved)
  float *s_x = s_row + 16 * D4;              // [16][ld_x]   normalised obs
  float *s_store = s_x + 16 * ld_x;          // 4*LH tiles: policy z,h | value z,h
  float *s_pp = s_store + 4 * LH * TT;       // 4 tiles: delta ping-pong (policy, value)
  float *s_y = s_pp + 4 * TT;                // [2][16][ld_y]  logits | value
  float *s_dy = s_y + 2 * 16 * ld_y;         // [2][16][ld_y]
  float *s_scal = s_dy + 2 * 16 * ld_y;      // [4][16]
  float *s_eps = s_scal + 64;                // [16][U]  entropy-sample noise, drawn in the tile-load section
  float *s_lpt = s_eps + ((16 * U + 3) & ~3);   // [16][U] per-dimension log-prob terms  | then [16][U] entropy terms
  float *s_ent = s_lpt + ((16 * U + 3) & ~3);
  float *zp = s_store, *hp = s_store + LH * TT, *zv = s_store + 2 * LH * TT, *hv = s_store + 3 * LH * TT;
  float *y_pi = s_y, *y_v = s_y + 16 * ld_y;
  const int PL = A.pi.n_layers, VL = A.v.n_layers;
  const int Lmax = PL > VL ? PL : VL;
  const float invM = 1.0f / (float)M;
  const float adv_mean = A.normalize_advantage ? A.mom[0] : 0.f;
  const float adv_istd = A.normalize_advantage ? 1.0f / (A.mom[1] + 1e-8f) : 1.f;          // losses.py:101-102
  const RngKey rk_ = rng_resolve(A.seed, A.offset, A.rng_dev);
  const unsigned long long rng_off = rk_.offset, rng_seed = rk_.seed;
  float *slab = A.slabs + (long long)blockIdx.x * (A.pi.n_params + A.v.n_params);
  float *slab_pi = slab, *slab_v = slab + A.pi.n_params;
  float loss_pol = 0.f, loss_v = 0.f, loss_ent = 0.f;     // thread 0..15 partials, reduced at the end
  const long long n_tiles = (M + 15) >> 4;
  bool first = true;
  const int net = chain & 1;   // 0 = policy, 1 = value
  const NetShape sh = net ? A.sh_v : A.sh_pi;
  const float *nparams = net ? A.v.params : A.pi.params;
  const int H1 = 16 * HT;
#pragma nounroll
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, first = false) {
    const int tid = opaque(tid_), lane = tid & 63;
    const long long r0 = tile * 16;
    WSet<HT, SP> R;
    if (chain < 2) chain_fwd_prefetch<HT, SP, WIDE>(R, sh, nparams, sub, lane);
    {
      const long long nvalid = (M - r0 < 16 ? M - r0 : 16) * D;
      for (int idx = tid; idx < 16 * D; idx += nthreads) {   // flat copy, row stride D (no padding, no division)
        s_row[idx] = idx < nvalid ? A.data[r0 * D + idx] : 0.f;
      }
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        const int r = idx & 15, c = idx >> 4;
        float o = (r0 + r < M) ? A.data[(r0 + r) * D + c] : 0.f;
        if (A.norm_mean) o = (o - A.norm_mean[c]) / A.norm_std[c];
        s_x[r * ld_x + c] = o;
      }
      // the entropy sample's noise depends on (seed, offset, element) only: drawn HERE, once, by the last threads of the workgroup
      // while the tile is in flight (Philox + Box-Muller is ~250 instructions; it used to be drawn twice per element inside the
      // loss section, on the 16 lanes every other wave waits for)
      for (int idx = nthreads - 1 - tid; idx < 16 * U; idx += nthreads) {
        const int r = idx / U, d = idx - r * U;
        const long long i = r0 + r;
        float e = 0.f;
        if (i < M) {
          const long long nidx = i * U + d;
          e = A.ent_noise ? A.ent_noise[nidx] : philox_normal(rng_seed, rng_off, MBPO_STREAM_ENTROPY, (unsigned long long)nidx);
        }
        s_eps[idx] = e;
      }
    }
    __syncthreads();
    // ---- forward: policy logits (:80) and value baseline (:82), both stored for the backward
    if (chain < 2)
      chain_fwd_run<HT, SP, WIDE>(sh, nparams, s_x, ld_x, nullptr, nullptr, net ? zv : zp, net ? hv : hp, net ? y_v : y_pi, ld_y, ld_h, Lmax,
                            sub, lane, R);
    else
      chain_idle_run(Lmax);
    if (chain < 2) chain_dgrad_prefetch<HT, SP, WIDE>(R, sh, nparams, sub, lane);
    // ---- per-sample loss terms and output gradients.  Stage 1: one lane per (row, action dim) forms that dimension's log-prob
    //      and entropy terms; stage 2 (after a barrier of its own: the stage-1 lanes may sit in other waves): one lane per row sums them
    //      in dimension order, forms rho and the clipped surrogate's weight and writes the value gradient; stage 3: one lane per
    //      (row, dim) again writes the logits' gradients.  Same arithmetic as the one-lane-per-row loop it replaces.
    for (int idx = tid; idx < 16 * U; idx += nthreads) {
      const int r = idx / U, d = idx - r * U;
      const float *row = s_row + r * D;
      const float loc = y_pi[r * ld_y + d], raw = y_pi[r * ld_y + U + d];
      const float sg = pp_fsoftplus(raw) + 0.001f;
      const float z = row[2 * X + U + 3 + d];                      // raw_action
      const float q = (z - loc) / sg;
      const float lsg = pp_flog(sg);
      s_lpt[idx] = -0.5f * q * q - lsg - LOG_SQRT_2PI - 2.0f * (LOG_2 - z - pp_fsoftplus(-2.0f * z));   // log_prob (:91-92)
      const float zf = loc + sg * s_eps[idx];
      s_ent[idx] = 0.5f + LOG_SQRT_2PI + lsg + 2.0f * (LOG_2 - zf - pp_fsoftplus(-2.0f * zf));          // entropy (:117)
    }
    __syncthreads();
    if (tid < 16) {
      const int r = tid;
      const long long i = r0 + r;
      const bool ok = i < M;
      const float *row = s_row + r * D;
      const float lp_b = row[2 * X + U + 2];                       // behaviour log-prob (policy_extras.log_prob)
      const float adv = ok ? (A.adv[i] - adv_mean) * adv_istd : 0.f;
      const float vs = ok ? A.vs[i] : 0.f;
      float lp_t = 0.f, ent = 0.f;
      for (int d = 0; d < U; ++d) {
        lp_t += s_lpt[r * U + d];
        ent += s_ent[r * U + d];
      }
      const float rho = pp_fexp(lp_t - lp_b);                                                             // :103
      const float lo = 1.f - A.clip_eps, hi = 1.f + A.clip_eps;
      const float s1 = rho * adv, s2 = fminf(fmaxf(rho, lo), hi) * adv;
      // d min(s1,s2)/d rho: inside the clip range s1 == s2 (tie, both branches carry adv); outside only s1 can carry it
      const bool inside = (rho >= lo) && (rho <= hi);
      const float w = inside ? 1.f : (s1 < s2 ? 1.f : 0.f);
      s_scal[48 + r] = ok ? -invM * rho * adv * w : 0.f;            // g_lp = d policy_loss / d lp_t
      const float v = y_v[r * ld_y];
      if (ok) {
        loss_pol += -fminf(s1, s2);
        loss_v += 0.5f * (vs - v) * (vs - v);
        loss_ent += ent;
      }
      s_dy[(16 + r) * ld_y] = ok ? -(vs - v) * invM : 0.f;          // d (0.5*mean((vs-V)^2)) / dV   (:112-114)
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * U; idx += nthreads) {
      const int r = idx / U, d = idx - r * U;
      const bool ok = r0 + r < M;
      const float *row = s_row + r * D;
      const float g_lp = s_scal[48 + r];
      const float g_ent = ok ? -A.entropy_cost * invM : 0.f;        // d entropy_loss / d entropy_i
      const float loc = y_pi[r * ld_y + d], raw = y_pi[r * ld_y + U + d];
      const float sg = pp_fsoftplus(raw) + 0.001f;
      const float z = row[2 * X + U + 3 + d];
      const float q = (z - loc) / sg;
      const float eps = s_eps[idx];
      const float th = pp_ftanh(loc + sg * eps);
      // lp_t: d/dloc = q/sg, d/dsigma = (q*q - 1)/sg ; entropy: d/dloc = -2 tanh(zf), d/dsigma = 1/sg - 2 tanh(zf) eps
      const float g_loc = g_lp * (q / sg) + g_ent * (-2.f * th);
      const float g_sig = g_lp * ((q * q - 1.f) / sg) + g_ent * (1.f / sg - 2.f * th * eps);
      s_dy[r * ld_y + d] = g_loc;
      s_dy[r * ld_y + U + d] = g_sig * fast_sigmoid(raw);
    }
    __syncthreads();
    // ---- backward: chains 0/1 push delta down (policy / value), chains 2/3 accumulate dW/db into this WG's slab
    {
      float *d0 = s_pp + (2 * net) * TT, *d1 = d0 + TT;
      if (chain < 2)
        chain_dgrad_run<HT, SP, WIDE>(sh, nparams, s_dy + net * 16 * ld_y, ld_y, net ? zv : zp, d0, d1, nullptr, ld_x, ld_h, Lmax, sub, lane, R);
      else
        chain_wgrad_run<HT, SP, WIDE>(sh, s_x, ld_x, net ? hv : hp, s_dy + net * 16 * ld_y, ld_y, d0, d1, net ? slab_v : slab_pi, !first, ld_h,
                                Lmax, sub, lane);
    }
  }
  (void)H1;
  const int tid = tid_;
  // ---- loss partials of this workgroup (fixed order)
  if (tid < 16) {
    s_scal[tid] = loss_pol;
    s_scal[16 + tid] = loss_v;
    s_scal[32 + tid] = loss_ent;
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = 0; i < 16; ++i) {
      a += s_scal[i];
      b += s_scal[16 + i];
      c += s_scal[32 + i];
    }
    A.extras[blockIdx.x * 4 + 0] = a;
    A.extras[blockIdx.x * 4 + 1] = b;
    A.extras[blockIdx.x * 4 + 2] = c;
  }
}

struct PpoReduceArgs {
  const float *slabs, *extras;
  int n_slabs, NPV, slab_step;
  int n_extras;              // loss partials to add (= n_slabs on the fused path: one set per workgroup of k_ppo_fwd_bwd)
  long long M;
  float entropy_cost;
  float *grads, *metrics, *metrics_accum;
  const float *step_count;   // already this step's count (k_ppo_values bumped it)
  // fused optimizer step (mbpo_ppo_step: no all-reduce sits between the gradient and AdamW): params != nullptr
  float *params, *adam_m, *adam_v;
  float lr, wd, grad_scale;
};

// stage 1 of the two-stage slab sum (many slabs): groups of 16 slabs, 4-KB contiguous runs (common.hpp slab_group16_sum)
__global__ void __launch_bounds__(256) k_ppo_reduce_groups(float *slabs, int NPV, int n_slabs) { slab_group16_sum(slabs, NPV, n_slabs, NPV); }

__global__ void __launch_bounds__(256) k_ppo_reduce(PpoReduceArgs A) {
  // (A.slab_step = 16 after k_ppo_reduce_groups: the group sums sit in slabs 0, 16, 32, ...)
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const float gsum = slab_sum_wg64(A.slabs, (long long)A.NPV * A.slab_step, (A.n_slabs + A.slab_step - 1) / A.slab_step, i, i < A.NPV);
  if (threadIdx.x < 64 && i < A.NPV) {
    A.grads[i] = gsum;
    if (A.params) {
      // k_ppo_apply's arithmetic on the element this thread has just summed: one launch less per minibatch_step, the same bits
      const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
      const float count = A.step_count[0];
      const float g = gsum * A.grad_scale;
      const float mu = b1 * A.adam_m[i] + 0.1f * g;
      const float nu = b2 * A.adam_v[i] + 0.001f * (g * g);
      A.adam_m[i] = mu;
      A.adam_v[i] = nu;
      const float mu_hat = mu / (1.f - powf(b1, count));
      const float nu_hat = nu / (1.f - powf(b2, count));
      const float p = A.params[i];
      const float upd = mu_hat / (sqrtf(nu_hat) + eps) + A.wd * p;
      A.params[i] = p + (-A.lr) * upd;
    }
  }
  if (blockIdx.x == 0) {
    // The loss partials (three per slab) are summed by the whole workgroup: thread t takes slabs t, t + 256, ..., then a fixed tree
    // over the 256 partial sums in LDS.  ONE thread walking 3 x n_slabs values (its loads in dependent batches of 16) was the
    // launch: 21.6 us at 512 slabs with the gradient sum itself done in 3 (rocprofv3, round 3).
    __shared__ float s_e[3][256];
    const int t = threadIdx.x;
    float e[3] = {0.f, 0.f, 0.f};
    for (int sl = t; sl < A.n_extras; sl += 256) {
#pragma unroll
      for (int k = 0; k < 3; ++k) e[k] += A.extras[(long long)sl * 4 + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) s_e[k][t] = e[k];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (t < w) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s_e[k][t] += s_e[k][t + w];
      }
      __syncthreads();
    }
    if (t == 0) {
      // (what this thread read-modify-writes is requested together)
      float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      if (A.metrics_accum) {
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] = A.metrics_accum[k];
      }
      const float a = s_e[0][0], b = s_e[1][0], c = s_e[2][0];
      const float invM = 1.0f / (float)A.M;
      const float pl = a * invM, vl = b * invM, el = A.entropy_cost * -(c * invM);
      const float m[4] = {pl + vl + el, pl, vl, el};   // m[0]: total_loss
#pragma unroll
      for (int k = 0; k < 4; ++k) A.metrics[k] = m[k];
      if (A.metrics_accum) {
#pragma unroll
        for (int k = 0; k < 4; ++k) A.metrics_accum[k] = acc[k] + m[k];
        A.metrics_accum[4] = acc[4] + 1.0f;
      }
    }
  }
}

struct PpoApplyArgs {
  float *params, *adam_m, *adam_v;
  const float *grads, *step_count;
  int NPV;
  float lr, wd, grad_scale;
};

__global__ void __launch_bounds__(256) k_ppo_apply(PpoApplyArgs A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.NPV) return;
  // [3P optax.adamw] (constants as optax forms them: f32(0.1), f32(0.001) — see sac.hip)
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  const float count = A.step_count[0];
  const float g = A.grads[i] * A.grad_scale;
  const float mu = b1 * A.adam_m[i] + 0.1f * g;
  const float nu = b2 * A.adam_v[i] + 0.001f * (g * g);
  A.adam_m[i] = mu;
  A.adam_v[i] = nu;
  const float mu_hat = mu / (1.f - powf(b1, count));
  const float nu_hat = nu / (1.f - powf(b2, count));
  const float p = A.params[i];
  const float upd = mu_hat / (sqrtf(nu_hat) + eps) + A.wd * p;
  A.params[i] = p + (-A.lr) * upd;
}

// ------------------------------------------------------------------------------------------------ host
struct PpoPlan {
  MlpDev pi, v;
  int P, V, NPV, H, LH, n_slabs;
  int sp2;            // H == 64, one-tile network ends, more tiles than CUs: the two-workgroups-per-CU launch (k_ppo_fwd_bwd<64, 2>)
  long long M;
  int ld_x, ld_h, ld_y;
  size_t lds_values, lds_fb;
  long long off_baseline, off_boot, off_trunc, off_term, off_rew, off_vs, off_adv, off_mom, off_part, off_slabs, off_extras, off_layered, total;
  // values + GAE + moment partials in one launch (k_ppo_values_gae): vg_G trajectories per workgroup, n_vg workgroups; 0 = the three
  // separate launches (layered shapes; trajectories too long for a workgroup's LDS arrays; MBPO_PPO_VALUES_GAE=0)
  int vg_G, n_vg;
  bool vg_lean;              // k_ppo_vg_lean instead of k_ppo_values_gae
  size_t lds_vg;
  long long off_mompart;
  // hidden layers outside the fused kernels' range (one width in {64,128}): values pre-pass and loss forward/backward run layer by
  // layer (ppo_layered.hip) and leave ONE slab; GAE scan, moments, reduction, metrics and AdamW are shared
  bool layered;
  // the benchmark networks (64 x 3, swish, u = 1): k_ppo_lean (ppo_lean.hip) — one workgroup and ONE slab per CU
  bool lean;
};

// Measurement / test hook (not part of include/mbpo_hip.h): 0 = always the generic k_ppo_fwd_bwd, 1 = k_ppo_lean where it applies,
// -1 = the MBPO_PPO_LEAN environment default (on).
static int g_ppo_lean = -1;
static unsigned long long *g_ppo_stamps = nullptr;
// measurement hook: device buffer of >= 64 uint64 that k_ppo_lean fills with s_memtime stamps (workgroup 0, 12 per tile); NULL = off
extern "C" int mbpo_debug_set_ppo_stamps(void *buf) {
  g_ppo_stamps = (unsigned long long *)buf;
  return MBPO_OK;
}
extern "C" int mbpo_debug_set_ppo_lean(int mode) {
  g_ppo_lean = mode;
  return MBPO_OK;
}

static int ppo_same_hidden(const int *dims, int n_layers) {
  if (n_layers < 2) return -1;
  for (int l = 2; l < n_layers; ++l)
    if (dims[l] != dims[1]) return -1;
  return dims[1];
}

static int ppo_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

static int ppo_plan(const mbpo_ppo_desc *d, PpoPlan *pl, bool need_ptrs) {
  MBPO_REQUIRE(d, MBPO_ERR_ARG, "ppo: null descriptor");
  MBPO_REQUIRE(d->x_dim > 0 && d->u_dim > 0 && d->batch_size > 0 && d->unroll_length > 0, MBPO_ERR_ARG, "ppo: bad sizes");
  MBPO_REQUIRE(d->row_len == 2 * d->x_dim + 2 * d->u_dim + 4, MBPO_ERR_ARG, "ppo: row_len %d != 2x+2u+4", d->row_len);
  MBPO_REQUIRE(d->policy_layers >= 2 && d->policy_layers <= MBPO_MAX_LAYERS && d->value_layers >= 2 && d->value_layers <= MBPO_MAX_LAYERS,
               MBPO_ERR_ARG, "ppo: networks need at least one hidden layer");
  MBPO_REQUIRE(d->policy_dims[0] == d->x_dim && d->policy_dims[d->policy_layers] == 2 * d->u_dim, MBPO_ERR_ARG,
               "ppo: policy must map [x_dim] -> [2*u_dim]");
  MBPO_REQUIRE(d->value_dims[0] == d->x_dim && d->value_dims[d->value_layers] == 1, MBPO_ERR_ARG, "ppo: value net must map [x_dim] -> [1]");
  const int Hp = ppo_same_hidden(d->policy_dims, d->policy_layers), Hv = ppo_same_hidden(d->value_dims, d->value_layers);
  static const int layered_env = getenv("MBPO_PPO_LAYERED") ? atoi(getenv("MBPO_PPO_LAYERED")) : 0;    // 1: force the layered path (tests)
  pl->layered = layered_env != 0 || !(Hp == Hv && (Hp == 64 || Hp == 128));
  mbpo_mlp_desc md;
  md.net_stride = 0;
  md.n_nets = 1;
  md.params = d->params ? d->params : (const float *)16;
  md.n_layers = d->policy_layers;
  for (int l = 0; l <= d->policy_layers; ++l) md.dims[l] = d->policy_dims[l];
  md.activation = d->policy_activation;
  int rc = mbpo_make_mlp_dev(&md, &pl->pi, "ppo.policy");
  if (rc != MBPO_OK) return rc;
  pl->P = pl->pi.n_params;
  md.n_layers = d->value_layers;
  for (int l = 0; l <= d->value_layers; ++l) md.dims[l] = d->value_dims[l];
  md.activation = d->value_activation;
  rc = mbpo_make_mlp_dev(&md, &pl->v, "ppo.value");
  if (rc != MBPO_OK) return rc;
  pl->V = pl->v.n_params;
  pl->v.params = d->params ? d->params + pl->P : nullptr;
  pl->NPV = pl->P + pl->V;
  pl->H = Hp;
  const int lhp = d->policy_layers - 1, lhv = d->value_layers - 1;
  pl->LH = lhp > lhv ? lhp : lhv;
  pl->M = (long long)d->batch_size * d->unroll_length;
  auto up4 = [](int v) { return (v + 3) & ~3; };
  pl->ld_x = up4(d->x_dim) + 4;
  pl->ld_h = Hp + 4;
  pl->ld_y = up4(2 * d->u_dim) + 4;
  pl->lds_values = sizeof(float) * (16ull * pl->ld_x + 2ull * 16 * pl->ld_h + 16ull * pl->ld_y);
  pl->lds_fb = sizeof(float) * (16ull * up4(d->row_len) + 16ull * pl->ld_x + (size_t)pl->LH * 4 * 16 * pl->ld_h + 4ull * 16 * pl->ld_h +
                                4ull * 16 * pl->ld_y + 64 + 3ull * ((16 * d->u_dim + 3) & ~3));
  long long tiles = (pl->M + 15) / 16;
  // one gradient slab per RESIDENT workgroup: a 1024-thread (or 512 x 256-VGPR) workgroup fills a CU's registers, so more of them
  // per CU only waited their turn while k_ppo_reduce summed twice the slabs (512 x 17 k floats = 35 MB per minibatch at C3's
  // T = 40: 22 us of a 158 us minibatch_step, rocprofv3 round 3).  The 512-thread / 128-VGPR launch (sp2) holds two per CU.
  {
    const NetShape sp_ = NetShape{d->x_dim, d->policy_layers, 2 * d->u_dim, 0}, sv_ = NetShape{d->x_dim, d->value_layers, 1, 0};
    static const int sp2_env = getenv("MBPO_PPO_SP2") ? atoi(getenv("MBPO_PPO_SP2")) : -1;
    pl->sp2 = (Hp == 64 && !net_is_wide(sp_) && !net_is_wide(sv_) && tiles > ppo_num_cus() && sp2_env != 0) ? 1 : 0;
  }
  long long cap = (pl->sp2 ? 2LL : 1LL) * ppo_num_cus();
  {
    static const int lean_env = getenv("MBPO_PPO_LEAN") ? atoi(getenv("MBPO_PPO_LEAN")) : 1;
    pl->lean = (g_ppo_lean >= 0 ? g_ppo_lean : lean_env) != 0 && !pl->layered &&
               ppo_lean_supports(d->x_dim, d->u_dim, d->policy_dims, d->policy_layers, d->policy_activation, d->value_dims, d->value_layers,
                                 d->value_activation);
    if (pl->lean) cap = ppo_num_cus();
    // the values + GAE launch has its own specialised form: the value network alone decides (64 x 2 or 64 x 3)
    pl->vg_lean = (g_ppo_lean >= 0 ? g_ppo_lean : lean_env) != 0 && !pl->layered && d->u_dim == 1 &&      // (the kernel's row offsets assume u = 1)
                  ppo_vg_lean_supports(d->x_dim, d->value_dims, d->value_layers, d->value_activation);
  }
  pl->n_slabs = (int)(tiles < cap ? tiles : cap);
  if (!pl->layered && (pl->lds_fb > 160 * 1024 || pl->lds_values > 160 * 1024)) pl->layered = true;     // more stored activations than a tile's LDS holds
  if (pl->layered) {
    pl->sp2 = 0;
    pl->n_slabs = 1;
    pl->lean = false;
    pl->vg_lean = false;
  }
  long long o = 0;
  auto take = [&](long long n) { long long at = o; o += (n + 3) & ~3LL; return at; };
  pl->off_baseline = take(pl->M); pl->off_boot = take(d->batch_size); pl->off_trunc = take(pl->M); pl->off_term = take(pl->M);
  pl->off_rew = take(pl->M); pl->off_vs = take(pl->M); pl->off_adv = take(pl->M); pl->off_mom = take(4); pl->off_part = take(PPO_MOM_WGS);
  pl->off_slabs = take((long long)pl->n_slabs * pl->NPV); pl->off_extras = take((long long)pl->n_slabs * 4);
  {
    // G = 1 while that does not oversubscribe the chip (a tile with a few rows of one trajectory wastes MFMA lanes nobody else wants:
    // the launch is a latency chain per workgroup); beyond 4 workgroups per CU, more trajectories per workgroup
    static const int vg_env = getenv("MBPO_PPO_VALUES_GAE") ? atoi(getenv("MBPO_PPO_VALUES_GAE")) : -1;
    // at most 1024 workgroups: k_ppo_moments_combine reads NPT * 256 = 1024 {n, mean, M2} partials (4 x 256 CUs sits exactly at
    // that limit; a device with more CUs must not drop partials silently: ADVICE r3)
    const long long cap_wg = 4LL * ppo_num_cus() < 1024 ? 4LL * ppo_num_cus() : 1024;
    long long G = (d->batch_size + cap_wg - 1) / cap_wg;
    if (G < 1) G = 1;
    const long long R = (long long)d->unroll_length + 1;
    pl->vg_G = (!pl->layered && vg_env != 0 && G * R <= 1024) ? (int)G : 0;
    pl->n_vg = pl->vg_G ? (int)((d->batch_size + pl->vg_G - 1) / pl->vg_G) : 0;
    const long long GT = (long long)pl->vg_G * d->unroll_length;
    pl->lds_vg = pl->lds_values + sizeof(float) * (size_t)(((pl->vg_G * R + 3) & ~3LL) + 4 * ((GT + 3) & ~3LL));
    pl->off_mompart = take(4LL * (pl->n_vg > 0 ? pl->n_vg : 1));
  }
  pl->off_layered = o;
  if (pl->layered) o += ppo_layered_floats(d, pl->pi, pl->v);
  pl->total = o;
  if (need_ptrs)
    MBPO_REQUIRE(d->params && d->adam_m && d->adam_v && d->step_count && d->grads && d->workspace && d->metrics, MBPO_ERR_ARG,
                 "ppo: null state pointer");
  return MBPO_OK;
}

extern "C" int64_t mbpo_ppo_workspace_floats(const mbpo_ppo_desc *d) {
  PpoPlan pl;
  int rc = ppo_plan(d, &pl, false);
  if (rc != MBPO_OK) return rc;
  return pl.total;
}

static int ppo_grads_impl(const mbpo_ppo_desc *d, void *stream, bool fuse_apply) {
  PpoPlan pl;
  int rc = ppo_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(d->data, MBPO_ERR_ARG, "ppo_grads: null data");
  MBPO_REQUIRE((d->norm_mean == nullptr) == (d->norm_std == nullptr), MBPO_ERR_ARG, "ppo_grads: norm_mean/norm_std mismatch");
  float *ws = d->workspace;
  PpoArgs A;
  A.pi = pl.pi; A.v = pl.v;
  A.sh_pi = NetShape{pl.pi.dims[0], pl.pi.n_layers, pl.pi.dims[pl.pi.n_layers], pl.pi.act};
  A.sh_v = NetShape{pl.v.dims[0], pl.v.n_layers, pl.v.dims[pl.v.n_layers], pl.v.act};
  A.X = d->x_dim; A.U = d->u_dim; A.B = d->batch_size; A.T = d->unroll_length; A.D = d->row_len;
  A.data = d->data; A.norm_mean = d->norm_mean; A.norm_std = d->norm_std; A.ent_noise = d->entropy_noise;
  A.step_count_rw = d->step_count;
  A.mom_inline = 0;
  A.mom_part = ws + pl.off_mompart; A.mom_parts = pl.n_vg; A.vg_G = pl.vg_G;
  A.seed = d->seed; A.offset = d->offset; A.rng_dev = (const unsigned long long *)d->rng_dev;
  A.entropy_cost = d->entropy_cost; A.discounting = d->discounting; A.reward_scaling = d->reward_scaling;
  A.gae_lambda = d->gae_lambda; A.clip_eps = d->clipping_epsilon; A.normalize_advantage = d->normalize_advantage;
  A.baseline = ws + pl.off_baseline; A.boot = ws + pl.off_boot; A.trunc = ws + pl.off_trunc; A.term = ws + pl.off_term;
  A.rew = ws + pl.off_rew; A.vs = ws + pl.off_vs; A.adv = ws + pl.off_adv; A.mom = ws + pl.off_mom;
  A.slabs = ws + pl.off_slabs; A.extras = ws + pl.off_extras; A.n_slabs = pl.n_slabs;
  A.ld_x = pl.ld_x; A.ld_h = pl.ld_h; A.ld_y = pl.ld_y; A.LH = pl.LH;
  hipStream_t st = (hipStream_t)stream;
  // 1. values pre-pass
  long long vt = (pl.M + d->batch_size + 15) / 16;
  int vgrid = (int)(vt < 4LL * ppo_num_cus() ? vt : 4LL * ppo_num_cus());
  if (pl.layered) {
    float *values = nullptr;
    rc = ppo_layered_values(d, pl.pi, pl.v, ws + pl.off_layered, A.trunc, A.term, A.rew, &values, st);
    if (rc != MBPO_OK) return rc;
    A.baseline = values;
    A.boot = values + pl.M;
  } else if (pl.vg_G && pl.vg_lean) {
    // 1-3 in one launch on the value network's resident images (ppo_lean.hip k_ppo_vg_lean: the same bits as k_ppo_values_gae)
    PpoVgLeanArgs V;
    V.v_params = d->params + pl.pi.n_params; V.data = d->data; V.norm_mean = d->norm_mean; V.norm_std = d->norm_std;
    V.B = d->batch_size; V.T = d->unroll_length; V.D = A.D; V.G = pl.vg_G; V.n_hid = d->value_layers - 2;
    V.reward_scaling = d->reward_scaling; V.discounting = d->discounting; V.gae_lambda = d->gae_lambda;
    V.vs = A.vs; V.adv = A.adv; V.mom_part = A.mom_part; V.step_count_rw = A.step_count_rw;
    const long long GR = (long long)pl.vg_G * (d->unroll_length + 1), GT = (long long)pl.vg_G * d->unroll_length;
    rc = ppo_vg_lean_launch(V, d->x_dim, pl.n_vg, (size_t)(((GR + 3) & ~3LL) + 4 * ((GT + 3) & ~3LL)), stream);
    if (rc != MBPO_OK) return rc;
    if (d->normalize_advantage)
      hipLaunchKernelGGL(k_ppo_moments_combine, dim3(1), dim3(256), 0, st, (const float *)A.mom_part, pl.n_vg, (float)pl.M, A.mom);
  } else if (pl.vg_G) {
    // 1-3 in one launch: values, GAE, the moments' per-workgroup partials
    const int tiles_wg = (int)(((long long)pl.vg_G * (d->unroll_length + 1) + 15) / 16);
    const int NCv = tiles_wg >= 4 ? 4 : tiles_wg;
    const size_t lds = pl.lds_vg + (size_t)(NCv - 1) * pl.lds_values;
#define LVG(H_, N_)                                                                                  \
  {                                                                                                  \
    rc = mbpo_ensure_lds<k_ppo_values_gae<H_, N_>>(lds, "ppo_grads");                                \
    if (rc != MBPO_OK) return rc;                                                                    \
    hipLaunchKernelGGL((k_ppo_values_gae<H_, N_>), dim3(pl.n_vg), dim3(256 * N_), lds, st, A);        \
  }
    if (pl.H == 64) {
      if (NCv == 1) LVG(64, 1) else if (NCv == 2) LVG(64, 2) else if (NCv == 3) LVG(64, 3) else LVG(64, 4)
    } else {
      if (NCv == 1) LVG(128, 1) else if (NCv == 2) LVG(128, 2) else if (NCv == 3) LVG(128, 3) else LVG(128, 4)
    }
#undef LVG
    if (d->normalize_advantage)
      hipLaunchKernelGGL(k_ppo_moments_combine, dim3(1), dim3(256), 0, st, (const float *)A.mom_part, pl.n_vg, (float)pl.M, A.mom);
  } else if (pl.H == 64) {
    rc = mbpo_ensure_lds<k_ppo_values<64>>(pl.lds_values, "ppo_grads");
    if (rc != MBPO_OK) return rc;
    hipLaunchKernelGGL(k_ppo_values<64>, dim3(vgrid), dim3(256), pl.lds_values, st, A);
  } else {
    rc = mbpo_ensure_lds<k_ppo_values<128>>(pl.lds_values, "ppo_grads");
    if (rc != MBPO_OK) return rc;
    hipLaunchKernelGGL(k_ppo_values<128>, dim3(vgrid), dim3(256), pl.lds_values, st, A);
  }
  // 2. GAE on [B,T] (batch-major: the data's native layout, no transpose)   losses.py:94-99,128-184
  if (!pl.vg_G) {
    rc = mbpo_gae_scan(A.trunc, A.term, A.rew, A.baseline, A.boot, A.vs, A.adv, d->batch_size, d->unroll_length, d->discounting,
                       d->gae_lambda, 0, stream);
    if (rc != MBPO_OK) return rc;
  }
  // 3. advantage moments over the whole minibatch
  if (d->normalize_advantage && !pl.vg_G) {
    float *part = ws + pl.off_part;
    long long blocks = (pl.M + 255) / 256;
    int g = (int)(blocks < PPO_MOM_WGS ? blocks : PPO_MOM_WGS);
    if (pl.M <= PPO_MOM_FUSED_MAX) hipLaunchKernelGGL(k_moments_fused, dim3(1), dim3(1024), 0, st, (const float *)A.adv, pl.M, A.mom);
    else {
    hipLaunchKernelGGL(k_moments_partial<0>, dim3(g), dim3(256), 0, st, (const float *)A.adv, pl.M, (const float *)A.mom, part);
    hipLaunchKernelGGL(k_moments_final<0>, dim3(1), dim3(64), 0, st, (const float *)part, g, pl.M, A.mom);
    hipLaunchKernelGGL(k_moments_partial<1>, dim3(g), dim3(256), 0, st, (const float *)A.adv, pl.M, (const float *)A.mom, part);
    hipLaunchKernelGGL(k_moments_final<1>, dim3(1), dim3(64), 0, st, (const float *)part, g, pl.M, A.mom);
    }
  }
  // 4. loss forward/backward
  float *layered_extras = nullptr;
  int layered_n_extras = 0;
  if (pl.layered) {
    rc = ppo_layered_fwd_bwd(d, pl.pi, pl.v, ws + pl.off_layered, A.vs, A.adv, A.mom, A.slabs, &layered_extras, &layered_n_extras, st);
    if (rc != MBPO_OK) return rc;
  } else if (pl.lean) {
    PpoLeanArgs L;
    L.params = d->params; L.data = d->data; L.norm_mean = d->norm_mean; L.norm_std = d->norm_std;
    L.adv = A.adv; L.vs = A.vs; L.mom = A.mom; L.ent_noise = d->entropy_noise;
    L.rng_dev = A.rng_dev; L.seed = A.seed; L.offset = A.offset;
    L.slabs = A.slabs; L.extras = A.extras; L.M = pl.M;
    L.entropy_cost = d->entropy_cost; L.clip_eps = d->clipping_epsilon; L.normalize_advantage = d->normalize_advantage;
    L.stamps = g_ppo_stamps;
    rc = ppo_lean_launch(L, d->x_dim, d->policy_layers - 2, pl.n_slabs, stream);
    if (rc != MBPO_OK) return rc;
  } else if (pl.H == 64) {
    const bool wide = net_is_wide(A.sh_pi) || net_is_wide(A.sh_v);
    rc = wide ? mbpo_ensure_lds<k_ppo_fwd_bwd<64, 4, true>>(pl.lds_fb, "ppo_grads") : mbpo_ensure_lds<k_ppo_fwd_bwd<64, 4, false>>(pl.lds_fb, "ppo_grads");
    if (rc != MBPO_OK) return rc;
    if (pl.sp2) {
      rc = mbpo_ensure_lds<k_ppo_fwd_bwd<64, 2, false>>(pl.lds_fb, "ppo_grads");
      if (rc != MBPO_OK) return rc;
      hipLaunchKernelGGL((k_ppo_fwd_bwd<64, 2, false>), dim3(pl.n_slabs), dim3(512), pl.lds_fb, st, A);
    } else if (wide) hipLaunchKernelGGL((k_ppo_fwd_bwd<64, 4, true>), dim3(pl.n_slabs), dim3(1024), pl.lds_fb, st, A);
    else hipLaunchKernelGGL((k_ppo_fwd_bwd<64, 4, false>), dim3(pl.n_slabs), dim3(1024), pl.lds_fb, st, A);
  } else {
    rc = mbpo_ensure_lds<k_ppo_fwd_bwd<128, 2, false>>(pl.lds_fb, "ppo_grads");
    if (rc != MBPO_OK) return rc;
    hipLaunchKernelGGL((k_ppo_fwd_bwd<128, 2, false>), dim3(pl.n_slabs), dim3(512), pl.lds_fb, st, A);
  }
  // 5. reduce
  PpoReduceArgs R;
  R.slabs = A.slabs; R.extras = A.extras; R.n_slabs = pl.n_slabs; R.NPV = pl.NPV; R.M = pl.M; R.entropy_cost = d->entropy_cost;
  R.n_extras = pl.n_slabs;
  if (pl.layered) { R.extras = layered_extras; R.n_extras = layered_n_extras; }
  R.grads = d->grads; R.metrics = d->metrics; R.metrics_accum = d->metrics_accum; R.step_count = d->step_count;
  R.params = fuse_apply ? d->params : nullptr; R.adam_m = d->adam_m; R.adam_v = d->adam_v;
  R.lr = d->lr; R.wd = d->wd; R.grad_scale = d->grad_scale;
  // The one-stage sum reads 256-byte pieces 68 KB apart: 22.7 us for the 35 MB of 512 slabs (C3, T = 40), unchanged by 4x the loads
  // in flight or by 1-KB pieces.  With many slabs: first groups of 16 over 4-KB contiguous runs, in place, then the 32 group sums.
  R.slab_step = 1;
  if (pl.n_slabs >= 64) {
    hipLaunchKernelGGL(k_ppo_reduce_groups, dim3((pl.NPV + 1023) / 1024, (pl.n_slabs + 15) / 16), dim3(256), 0, st, A.slabs, pl.NPV,
                       pl.n_slabs);
    R.slab_step = 16;
  }
  hipLaunchKernelGGL(k_ppo_reduce, dim3((pl.NPV + 63) / 64), dim3(256), 0, st, R);
  MBPO_CHECK_LAUNCH("ppo_grads");
  return MBPO_OK;
}

extern "C" int mbpo_ppo_grads(const mbpo_ppo_desc *d, void *stream) { return ppo_grads_impl(d, stream, false); }

// One minibatch_step without a seam for a collective: the reduce launch applies AdamW to the elements it has just summed
// (k_ppo_apply's arithmetic, bit for bit) — one launch less per minibatch_step.
extern "C" int mbpo_ppo_step(const mbpo_ppo_desc *d, void *stream) { return ppo_grads_impl(d, stream, true); }

extern "C" int mbpo_ppo_apply(const mbpo_ppo_desc *d, void *stream) {
  PpoPlan pl;
  int rc = ppo_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  PpoApplyArgs A;
  A.params = d->params; A.adam_m = d->adam_m; A.adam_v = d->adam_v; A.grads = d->grads; A.step_count = d->step_count;
  A.NPV = pl.NPV; A.lr = d->lr; A.wd = d->wd; A.grad_scale = d->grad_scale;
  hipLaunchKernelGGL(k_ppo_apply, dim3((pl.NPV + 255) / 256), dim3(256), 0, (hipStream_t)stream, A);
  MBPO_CHECK_LAUNCH("ppo_apply");
  return MBPO_OK;
}

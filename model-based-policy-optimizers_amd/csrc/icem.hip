// icem.hip — N4: the iCEM trajectory optimizer's device side (trajectory_optimizers/icem_optimizer.py:135-252).
//   k_icem_sample_par  coloured-noise candidates: powerlaw_psd_gaussian (utils/general_utils.py:81-208) restated as a direct inverse
//                  real DFT (horizons are tens of steps: H*K MACs per series, no FFT library), mean + noise*std, clip, previous
//                  elites appended, every candidate replicated over its particles in the rollout kernel's open-loop layout.
//   (rollouts)     mbpo_model_rollout with `actions` (rollout_actions, utils/optimizer_utils.py:11-59) — csrc/rollout.hip
//   k_icem_values  objective: summarize_particles( mean_t reward )                                      (:146-163)
//   k_icem_update  one workgroup: stable rank of the candidates (np.argsort), elite mean / population variance, soft update,
//                  best-so-far, the elite fraction carried to the next iteration                          (:196-232)
// HBM-bound bookkeeping around the rollout; everything stays on the device, the host only sequences the launches.
#include "common.hpp"

#define ICEM_MAX_HU 4096

struct IcemSampleArgs {
  const float *mean, *std, *prev_elites, *u_min, *u_max;
  int S, Kp, H, U, P;
  float exponent;
  unsigned long long seed, offset;
  const unsigned long long *rng_dev;
  float *actions;      // [H][(S+Kp)*P][U]
  float *candidates;   // [S+Kp][H][U]
};

// One thread per (series, step): a block takes 256 / H series, draws their K coefficient pairs into LDS (one Philox + Box-Muller pair
// per thread), then every (series, t) thread forms its inverse-DFT sum over k in order and writes its element and the particles'
// copies.  (Round 3's one-thread-per-series form — same values bit for bit — spent 43 us on 550 series at the reference's test sizes;
// this one 5 us.)
__global__ void __launch_bounds__(256) k_icem_sample_par(IcemSampleArgs A) {
  extern __shared__ float s_tab[];             // cos / sin tables [H][K], scale [K], coefficients [SB][K] x 2
  const int H = A.H, U = A.U, K = H / 2 + 1;
  const int SB = 256 / H;                       // series per block
  float *s_cos = s_tab, *s_sin = s_tab + H * K, *s_scale = s_sin + H * K, *s_r = s_scale + K, *s_i = s_r + SB * K;
  for (int i = threadIdx.x; i < H * K; i += blockDim.x) {
    const int t = i / K, k = i - t * K;
    const float ang = 6.28318530717958647692f * (float)((k * t) % H) / (float)H;
    s_cos[i] = cosf(ang);
    s_sin[i] = sinf(ang);
  }
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float f = fmaxf((float)k / (float)H, 1.0f / (float)H);      // rfftfreq, low-frequency cutoff fmin = 1/samples
    s_scale[k] = powf(f, -0.5f * A.exponent);
  }
  __syncthreads();
  float wsum = 0.f;
  for (int k = 1; k < K; ++k) {
    float w = s_scale[k];
    if (k == K - 1) w *= (1.0f + (float)(H % 2)) * 0.5f;
    wsum += w * w;
  }
  const float sigma = 2.0f * sqrtf(wsum) / (float)H;
  const RngKey rk_ = rng_resolve(A.seed, A.offset, A.rng_dev);
  const unsigned long long off = rk_.offset, rng_seed = rk_.seed;
  const int NC = A.S + A.Kp, N = NC * A.P;
  const int sd0 = blockIdx.x * SB;
  // coefficients of this block's sampled series
  for (int idx = threadIdx.x; idx < SB * K; idx += blockDim.x) {
    const int j = idx / K, k = idx - j * K, sd = sd0 + j;
    if (sd < NC * U && sd / U < A.S) {
      const unsigned long long base = ((unsigned long long)sd * K + k) * 2ull;
      float r = philox_normal(rng_seed, off, MBPO_STREAM_ICEM, base) * s_scale[k];
      float im = philox_normal(rng_seed, off, MBPO_STREAM_ICEM, base + 1ull) * s_scale[k];
      if (!(H % 2) && k == K - 1) {      // even length: the Nyquist coefficient is real (:193-197)
        im = 0.f;
        r *= 1.41421356237309504880f;
      }
      if (k == 0) {                      // the DC coefficient is real (:199-201)
        im = 0.f;
        r *= 1.41421356237309504880f;
      }
      s_r[idx] = r;
      s_i[idx] = im;
    }
  }
  __syncthreads();
  const int j = threadIdx.x / H, t = threadIdx.x - j * H, sd = sd0 + j;
  if (j >= SB || sd >= NC * U) return;
  const int c = sd / U, d = sd - c * U;
  float a;
  if (c >= A.S) {        // previous elites ride along unchanged (:190)
    a = A.prev_elites[((long long)(c - A.S) * H + t) * U + d];
  } else {
    const float *sr = s_r + j * K, *si = s_i + j * K;
    // irfft: y_t = (1/H) [ s_0 + 2 sum_{0<k<H/2} (sr_k cos - si_k sin) + (H even) s_{H/2} cos(pi t) ]
    float y = sr[0];
    const int kmax = (H % 2) ? K : K - 1;
    for (int k = 1; k < kmax; ++k) y += 2.0f * (sr[k] * s_cos[t * K + k] - si[k] * s_sin[t * K + k]);
    if (!(H % 2)) y += sr[K - 1] * s_cos[t * K + K - 1];
    y = y / (float)H / sigma;
    a = A.mean[t * U + d] + y * A.std[t * U + d];                      // :186
    a = fminf(fmaxf(a, A.u_min[d]), A.u_max[d]);                       // :187
  }
  A.candidates[((long long)c * H + t) * U + d] = a;
  for (int p = 0; p < A.P; ++p) A.actions[((long long)t * N + c * A.P + p) * U + d] = a;
}

extern "C" int mbpo_icem_sample(const float *mean, const float *std, const float *prev_elites, const float *u_min, const float *u_max,
                                int32_t n_samples, int32_t n_prev, int32_t horizon, int32_t u_dim, int32_t n_particles, float exponent,
                                uint64_t seed, uint64_t offset, const uint64_t *rng_dev, float *actions, float *candidates, void *stream) {
  MBPO_REQUIRE(mean && std && u_min && u_max && actions && candidates, MBPO_ERR_ARG, "icem_sample: null pointer");
  MBPO_REQUIRE(n_samples > 0 && n_prev >= 0 && u_dim > 0 && n_particles > 0, MBPO_ERR_ARG, "icem_sample: bad sizes");
  MBPO_REQUIRE(horizon >= 2 && horizon <= 128, MBPO_ERR_UNSUPPORTED, "icem_sample: horizon must be in [2, 128]");
  MBPO_REQUIRE(n_prev == 0 || prev_elites, MBPO_ERR_ARG, "icem_sample: prev_elites is NULL");
  IcemSampleArgs A{mean, std, prev_elites, u_min, u_max, n_samples, n_prev, horizon, u_dim, n_particles, exponent, seed, offset, (const unsigned long long *)rng_dev,
                   actions, candidates};
  const int K = horizon / 2 + 1;
  const int work = (n_samples + n_prev) * u_dim;
  const int SB = 256 / horizon;      // series per block of the one-thread-per-(series, step) form (horizon <= 128: SB >= 2)
  const size_t lds = sizeof(float) * (2ull * horizon * K + K + 2ull * SB * K);
  {      // (horizons above ~120 need more than the default 64 KB of dynamic LDS for the cos / sin tables)
    const int rc = mbpo_ensure_lds<k_icem_sample_par>(lds, "icem_sample");
    if (rc != MBPO_OK) return rc;
  }
  hipLaunchKernelGGL(k_icem_sample_par, dim3((work + SB - 1) / SB), dim3(256), lds, (hipStream_t)stream, A);
  MBPO_CHECK_LAUNCH("icem_sample");
  return MBPO_OK;
}

// ---- objective + elite update ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_icem_values(const float *rows, int row_len, int reward_col, int NC, int P, int H, int use_max,
                                                      float *values, const float *particle_cost, float lambda_c, int cost_use_max) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= NC) return;
  const long long N = (long long)NC * P;
  float agg = 0.f;
  for (int p = 0; p < P; ++p) {
    float acc = 0.f;
    for (int t = 0; t < H; ++t) acc += rows[((long long)t * N + (long long)c * P + p) * row_len + reward_col];
    const float m = acc / (float)H;                                   // jnp.mean(transitions.reward, axis=-1)
    agg = (p == 0) ? m : (use_max ? fmaxf(agg, m) : agg + m);
  }
  float value = use_max ? agg : agg / (float)P;                        // summarize_raw_samples: mean (or max under optimism)
  if (particle_cost) {
    // constraint (:161-166): cost = summarize_cost_samples(vmap(cost_fn)(observation, action)) over the particles — mean, or max
    // under pessimism; objective = reward - lambda_constraint * relu(cost).  The per-particle costs come from the user's callable.
    float cagg = 0.f;
    for (int p = 0; p < P; ++p) {
      const float cp = particle_cost[(long long)c * P + p];
      cagg = (p == 0) ? cp : (cost_use_max ? fmaxf(cagg, cp) : cagg + cp);
    }
    const float cost = cost_use_max ? cagg : cagg / (float)P;
    value = value - lambda_c * fmaxf(cost, 0.f);
  }
  values[c] = value;
}

// One wave per candidate: lane p sums particle p's rewards over the horizon (in step order), lane 0 then combines the particles in
// particle order — the arithmetic of the one-thread-per-candidate form (kept below for more than 64 particles), 10 x the loads in flight.
__global__ void __launch_bounds__(256) k_icem_values_wave(const float *rows, int row_len, int reward_col, int NC, int P, int H, int use_max,
                                                           float *values, const float *particle_cost, float lambda_c, int cost_use_max) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= NC) return;
  const long long N = (long long)NC * P;
  float m = 0.f;
  if (lane < P) {
    float acc = 0.f;
    for (int t = 0; t < H; ++t) acc += rows[((long long)t * N + (long long)c * P + lane) * row_len + reward_col];
    m = acc / (float)H;                                               // jnp.mean(transitions.reward, axis=-1)
  }
  float agg = 0.f;
  for (int p = 0; p < P; ++p) {
    const float mp = __shfl(m, p, 64);
    agg = (p == 0) ? mp : (use_max ? fmaxf(agg, mp) : agg + mp);
  }
  if (lane != 0) return;
  float value = use_max ? agg : agg / (float)P;                        // summarize_raw_samples: mean (or max under optimism)
  if (particle_cost) {
    float cagg = 0.f;
    for (int p = 0; p < P; ++p) {
      const float cp = particle_cost[(long long)c * P + p];
      cagg = (p == 0) ? cp : (cost_use_max ? fmaxf(cagg, cp) : cagg + cp);
    }
    const float cost = cost_use_max ? cagg : cagg / (float)P;
    value = value - lambda_c * fmaxf(cost, 0.f);
  }
  values[c] = value;
}

struct IcemUpdateArgs {
  const float *values, *candidates;
  int NC, H, U, n_elites, n_prev;
  float alpha;
  float *mean, *std, *best_value, *best_sequence, *prev_elites;
  int *rank;   // workspace [NC]
};

__global__ void __launch_bounds__(1024) k_icem_update(IcemUpdateArgs A) {
  const int tid = threadIdx.x, NC = A.NC, HU = A.H * A.U;
  // stable ascending rank = position in np.argsort(values)
  for (int c = tid; c < NC; c += 1024) {
    const float v = A.values[c];
    int r = 0;
    for (int j = 0; j < NC; ++j) {
      const float w = A.values[j];
      r += (w < v || (w == v && j < c)) ? 1 : 0;
    }
    A.rank[c] = r;
  }
  __threadfence();      // (agent scope: the ranks are re-read from global memory by other waves behind the barrier)
  __syncthreads();
  const int first = NC - A.n_elites;      // elites = sorted positions [first, NC)
  // the best elite (rank NC-1) and the best-so-far sequence (:212-221)
  __shared__ int s_best;
  for (int c = tid; c < NC; c += 1024)
    if (A.rank[c] == NC - 1) s_best = c;
  __syncthreads();
  const int cb = s_best;
  const float best_elite = A.values[cb];
  const bool take = A.best_value[0] <= best_elite;
  for (int i = tid; i < HU; i += 1024) {
    // elite mean / population variance of element i over the elites, in sorted order
    float m = 0.f;
    for (int c = 0; c < NC; ++c)
      if (A.rank[c] >= first) m += A.candidates[(long long)c * HU + i];
    m /= (float)A.n_elites;
    float v = 0.f;
    for (int c = 0; c < NC; ++c)
      if (A.rank[c] >= first) {
        const float dlt = A.candidates[(long long)c * HU + i] - m;
        v += dlt * dlt;
      }
    v /= (float)A.n_elites;
    const float sd = A.std[i];
    const float nm = A.mean[i] * A.alpha + (1.f - A.alpha) * m;          // :205
    const float nv = sd * sd * A.alpha + (1.f - A.alpha) * v;            // :206
    A.mean[i] = nm;
    A.std[i] = sqrtf(nv);                                                // :209
    if (take) A.best_sequence[i] = A.candidates[(long long)cb * HU + i];
  }
  // elites[-n_prev:] in sorted order -> next iteration's prev_elites (:227)
  for (int c = tid; c < NC; c += 1024) {
    const int pos = A.rank[c] - (NC - A.n_prev);
    if (pos >= 0)
      for (int i = 0; i < HU; ++i) A.prev_elites[(long long)pos * HU + i] = A.candidates[(long long)c * HU + i];
  }
  __syncthreads();
  if (tid == 0 && take) A.best_value[0] = best_elite;
}

// The same update with everything small in LDS (values, ranks, the elites' indices in candidate order, the elites' sequences):
// ranks from LDS-resident values, the elite list built once, the elites' H x U sequences fetched by all threads at once, and the
// per-element sums then walk LDS in the SAME order as k_icem_update (ascending candidate index) — identical results; the one-workgroup
// form above spent its time in 2 x NC dependent global round trips per element on H x U threads (144 us at the reference's test sizes).
// LDS floats: 2 NC + n_elites + n_elites * H * U.
__global__ void __launch_bounds__(1024) k_icem_update_lds(IcemUpdateArgs A) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, NC = A.NC, HU = A.H * A.U, NE = A.n_elites;
  float *s_val = sm;
  int *s_rank = reinterpret_cast<int *>(sm + NC);
  int *s_el = s_rank + NC;
  float *s_seq = reinterpret_cast<float *>(s_el + NE);      // [NE][HU]
  __shared__ int s_best;
  for (int c = tid; c < NC; c += 1024) s_val[c] = A.values[c];
  __syncthreads();
  for (int c = tid; c < NC; c += 1024) {
    const float v = s_val[c];
    int r = 0;
    for (int j = 0; j < NC; ++j) {
      const float w = s_val[j];
      r += (w < v || (w == v && j < c)) ? 1 : 0;
    }
    s_rank[c] = r;
    A.rank[c] = r;
    if (r == NC - 1) s_best = c;
  }
  __syncthreads();
  const int first = NC - NE;      // elites = sorted positions [first, NC)
  for (int c = tid; c < NC; c += 1024) {
    if (s_rank[c] >= first) {
      int pos = 0;
      for (int j = 0; j < c; ++j) pos += (s_rank[j] >= first) ? 1 : 0;
      s_el[pos] = c;              // the elites in ascending candidate order: the order k_icem_update adds them in
    }
  }
  __syncthreads();
  for (int idx = tid; idx < NE * HU; idx += 1024) {
    const int k = idx / HU, i = idx - k * HU;
    s_seq[idx] = A.candidates[(long long)s_el[k] * HU + i];
  }
  const int cb = s_best;
  const float best_elite = s_val[cb];
  const bool take = A.best_value[0] <= best_elite;
  __syncthreads();
  for (int i = tid; i < HU; i += 1024) {
    float m = 0.f;
    for (int k = 0; k < NE; ++k) m += s_seq[k * HU + i];
    m /= (float)NE;
    float v = 0.f;
    for (int k = 0; k < NE; ++k) {
      const float dlt = s_seq[k * HU + i] - m;
      v += dlt * dlt;
    }
    v /= (float)NE;
    const float sd = A.std[i];
    const float nm = A.mean[i] * A.alpha + (1.f - A.alpha) * m;          // :205
    const float nv = sd * sd * A.alpha + (1.f - A.alpha) * v;            // :206
    A.mean[i] = nm;
    A.std[i] = sqrtf(nv);                                                // :209
    if (take) A.best_sequence[i] = A.candidates[(long long)cb * HU + i];
  }
  // elites[-n_prev:] in sorted order -> next iteration's prev_elites (:227)
  for (int idx = tid; idx < NE * HU; idx += 1024) {
    const int k = idx / HU, i = idx - k * HU;
    const int pos = s_rank[s_el[k]] - (NC - A.n_prev);
    if (pos >= 0) A.prev_elites[(long long)pos * HU + i] = s_seq[idx];
  }
  __syncthreads();
  if (tid == 0 && take) A.best_value[0] = best_elite;
}

extern "C" int mbpo_icem_update_constrained(const float *rows, int32_t row_len, int32_t reward_col, int32_t n_candidates, int32_t n_particles,
                                            int32_t horizon, int32_t u_dim, const float *candidates, int32_t n_elites, int32_t n_prev,
                                            float alpha, int32_t use_max, const float *particle_cost, float lambda_constraint,
                                            int32_t cost_use_max, float *mean, float *std, float *best_value, float *best_sequence,
                                            float *prev_elites, float *values, int32_t *workspace, void *stream) {
  MBPO_REQUIRE(rows && candidates && mean && std && best_value && best_sequence && values && workspace, MBPO_ERR_ARG, "icem_update: null pointer");
  MBPO_REQUIRE(n_candidates > 0 && n_particles > 0 && horizon > 0 && u_dim > 0, MBPO_ERR_ARG, "icem_update: bad sizes");
  MBPO_REQUIRE(n_elites > 0 && n_elites <= n_candidates && n_prev >= 0 && n_prev <= n_elites, MBPO_ERR_ARG, "icem_update: bad elite counts");
  MBPO_REQUIRE(n_prev == 0 || prev_elites, MBPO_ERR_ARG, "icem_update: prev_elites is NULL");
  MBPO_REQUIRE(reward_col >= 0 && reward_col < row_len, MBPO_ERR_ARG, "icem_update: bad reward column");
  hipStream_t st = (hipStream_t)stream;
  if (n_particles <= 64)
    hipLaunchKernelGGL(k_icem_values_wave, dim3((n_candidates + 3) / 4), dim3(256), 0, st, rows, row_len, reward_col, n_candidates, n_particles,
                       horizon, use_max, values, particle_cost, lambda_constraint, cost_use_max);
  else
    hipLaunchKernelGGL(k_icem_values, dim3((n_candidates + 255) / 256), dim3(256), 0, st, rows, row_len, reward_col, n_candidates, n_particles,
                       horizon, use_max, values, particle_cost, lambda_constraint, cost_use_max);
  IcemUpdateArgs A{values, candidates, n_candidates, horizon, u_dim, n_elites, n_prev, alpha, mean, std, best_value, best_sequence,
                   prev_elites, workspace};
  const size_t lds = (2ull * n_candidates + n_elites + (size_t)n_elites * horizon * u_dim) * sizeof(float);
  if (lds <= 60 * 1024) hipLaunchKernelGGL(k_icem_update_lds, dim3(1), dim3(1024), lds, st, A);
  else hipLaunchKernelGGL(k_icem_update, dim3(1), dim3(1024), 0, st, A);
  MBPO_CHECK_LAUNCH("icem_update");
  return MBPO_OK;
}

extern "C" int mbpo_icem_update(const float *rows, int32_t row_len, int32_t reward_col, int32_t n_candidates, int32_t n_particles,
                                int32_t horizon, int32_t u_dim, const float *candidates, int32_t n_elites, int32_t n_prev, float alpha,
                                int32_t use_max, float *mean, float *std, float *best_value, float *best_sequence, float *prev_elites,
                                float *values, int32_t *workspace, void *stream) {
  return mbpo_icem_update_constrained(rows, row_len, reward_col, n_candidates, n_particles, horizon, u_dim, candidates, n_elites, n_prev, alpha,
                                      use_max, nullptr, 0.f, 0, mean, std, best_value, best_sequence, prev_elites, values, workspace, stream);
}

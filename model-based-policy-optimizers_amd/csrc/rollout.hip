// rollout.hip — R2 ensemble MLP forward and R1-R8 fused model rollout (see include/mbpo_hip.h).
//
// Decomposition (MI355X): one workgroup owns a tile of 16 envs for ALL steps of the rollout and ALL ensemble
// members, because every step ends in a reduction over members (mean / member pick) that feeds the next step's
// input.  Each wave walks a whole (tile, network) chain (wave_mlp.hpp): wave 0 the policy, then waves 0..E-1 one
// ensemble member each, with two workgroup barriers per step instead of one per layer.  Activations live in LDS,
// weights stream from L2 (flat params are ~0.2 MB, shared by every workgroup), the transition row is assembled in
// LDS and written to HBM as one contiguous block per step.
// Per (env, step) HBM traffic is one row write (row_len*4 B) — the kernel is MFMA/latency bound by design.
#include "common.hpp"
#include "chain_run.hpp"
#include "ens_lean.hpp"
#include "rollout_shared.hpp"
#include "rollout_lean.hpp"
#include <stdlib.h>
#include <string.h>

// ------------------------------------------------------------------------------------------------
// R2: y[e][row][:] = MLP_e(x[row])  — replaces vmap(Dynamics.next_state) (base_dynamics.py:15-20).
// ------------------------------------------------------------------------------------------------
struct EnsFwdArgs {
  MlpDev mlp;
  const float *x;
  float *y;
  long long n_rows;
  int shared_input;
  int ld_x, ld_h, ld_y, n_chains;
};

// workgroup = n_chains waves; wave w walks member (round*n_chains + w)'s chain on the tile
template <int H>
__global__ void __launch_bounds__(512) k_ensemble_forward(EnsFwdArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const MlpDev &m = A.mlp;
  const int E = m.n_nets, din = m.dims[0], dout = m.dims[m.n_layers];
  const int n_in_nets = A.shared_input ? 1 : E;
  const int T = 16 * A.ld_h;
  float *s_x = smem;                                // [n_in_nets][16][ld_x]
  float *s_pp = s_x + n_in_nets * 16 * A.ld_x;      // [n_chains][2] hidden tiles
  float *s_y = s_pp + A.n_chains * 2 * T;           // [E][16][ld_y]
  const long long n_tiles = (A.n_rows + 15) >> 4;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long row0 = tile * 16;
    for (int idx = tid; idx < n_in_nets * 16 * din; idx += nthreads) {
      int e = idx / (16 * din), rem = idx - e * 16 * din;
      int r = rem / din, c = rem - r * din;
      long long row = row0 + r;
      float v = 0.f;
      if (row < A.n_rows) v = A.x[((long long)e * A.n_rows + row) * din + c];
      s_x[(e * 16 + r) * A.ld_x + c] = v;
    }
    __syncthreads();
    for (int e0 = 0; e0 < E; e0 += A.n_chains) {
      const int e = e0 + wave;
      if (e < E)
        wave_mlp_fwd<HT>(m, m.params + (long long)e * m.net_stride, s_x + (A.shared_input ? 0 : e * 16 * A.ld_x), A.ld_x,
                         s_pp + wave * 2 * T, s_pp + wave * 2 * T + T, nullptr, nullptr, A.ld_h, s_y + e * 16 * A.ld_y, A.ld_y, lane);
    }
    __syncthreads();
    for (int idx = tid; idx < E * 16 * dout; idx += nthreads) {
      int e = idx / (16 * dout), rem = idx - e * 16 * dout;
      int r = rem / dout, c = rem - r * dout;
      long long row = row0 + r;
      if (row < A.n_rows) A.y[((long long)e * A.n_rows + row) * dout + c] = s_y[(e * 16 + r) * A.ld_y + c];
    }
    __syncthreads();
  }
}

static int hidden_width(const MlpDev &m) {
  // all hidden layers must share one width H in {64,128,256}; a 1-layer MLP (no hidden) uses the 64 build.
  if (m.n_layers == 1) return 64;
  int H = m.dims[1];
  for (int l = 2; l < m.n_layers; ++l)
    if (m.dims[l] != H) return -1;
  return H;
}

static int up4(int v) { return (v + 3) & ~3; }

// how many chains (waves) a workgroup runs side by side: one per member, capped at 8 waves (512 threads: two waves per
// SIMD keep 256 VGPRs each) and by the LDS that the per-chain ping-pong tiles need next to `fixed_floats` of other
// buffers; larger ensembles run in rounds of n_chains members
static int pick_chains(int n_members, size_t fixed_floats, int ld_h) {
  int c = n_members < 1 ? 1 : (n_members > 8 ? 8 : n_members);
  while (c >= 1 && (fixed_floats + 2ull * c * 16 * ld_h) * sizeof(float) > 160 * 1024) --c;
  return c;
}

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// Measurement / test hook (not part of include/mbpo_hip.h): 0 = always the generic k_ensemble_forward, 1 = k_ens_fwd_lean where it applies,
// -1 = the MBPO_ENS_LEAN environment default (on).
static int g_ens_lean = -1;
extern "C" int mbpo_debug_set_ens_lean(int mode) {
  g_ens_lean = mode;
  return MBPO_OK;
}

extern "C" int mbpo_ensemble_mlp_forward(const mbpo_mlp_desc *mlp, const float *x, int32_t shared_input, float *y,
                                         int64_t n_rows, void *stream) {
  MBPO_REQUIRE(mlp && x && y, MBPO_ERR_ARG, "ensemble_mlp_forward: null pointer");
  MBPO_REQUIRE(n_rows >= 0, MBPO_ERR_ARG, "ensemble_mlp_forward: n_rows < 0");
  EnsFwdArgs A;
  int rc = mbpo_make_mlp_dev(mlp, &A.mlp, "ensemble_mlp_forward");
  if (rc != MBPO_OK) return rc;
  if (n_rows == 0) return MBPO_OK;
  {
    // 64-wide member networks with 4 or 5 inputs: the throughput kernel (ens_lean.hip) — weights resident per workgroup, two tiles in
    // flight per workgroup, two workgroups per CU.  MBPO_ENS_LEAN=0 / mbpo_debug_set_ens_lean(0) keeps the generic kernel.
    static const int lean_env = getenv("MBPO_ENS_LEAN") ? atoi(getenv("MBPO_ENS_LEAN")) : 1;
    if ((g_ens_lean >= 0 ? g_ens_lean : lean_env) != 0 && ens_lean_supports(A.mlp.dims, A.mlp.n_layers, A.mlp.act)) {
      EnsLeanArgs L;
      L.params = mlp->params; L.net_stride = mlp->n_nets > 1 ? mlp->net_stride : A.mlp.n_params;
      L.x = x; L.y = y; L.n_rows = n_rows; L.E = mlp->n_nets; L.N = A.mlp.dims[A.mlp.n_layers]; L.shared_input = shared_input ? 1 : 0;
      const long long pairs = (((n_rows + 15) >> 4) + 1) >> 1;
      long long wpm = (2LL * num_cus()) / L.E;
      if (wpm < 1) wpm = 1;
      if (wpm > pairs) wpm = pairs;
      L.wgs_per_member = (int)wpm;
      L.n_hid = A.mlp.n_layers - 2;
      rc = ens_lean_launch(L, A.mlp.dims[0], num_cus(), stream);
      if (rc != MBPO_OK) return rc;
      MBPO_CHECK_LAUNCH("ensemble_mlp_forward");
      return MBPO_OK;
    }
  }
  const int H = hidden_width(A.mlp);
  MBPO_REQUIRE(H == 64 || H == 128 || H == 256, MBPO_ERR_UNSUPPORTED,
               "ensemble_mlp_forward: hidden layers must share one width in {64,128,256}");
  A.x = x;
  A.y = y;
  A.n_rows = n_rows;
  A.shared_input = shared_input ? 1 : 0;
  A.ld_x = up4(A.mlp.dims[0]) + 4;
  A.ld_h = H + 4;
  A.ld_y = up4(A.mlp.dims[A.mlp.n_layers]) + 4;
  const int E = A.mlp.n_nets;
  const size_t fixed_f = (size_t)(shared_input ? 1 : E) * 16 * A.ld_x + (size_t)E * 16 * A.ld_y;
  A.n_chains = pick_chains(E, fixed_f, A.ld_h);
  MBPO_REQUIRE(A.n_chains >= 1, MBPO_ERR_UNSUPPORTED, "ensemble_mlp_forward: shapes do not fit 160 KiB of LDS");
  size_t lds = sizeof(float) * (fixed_f + 2ull * A.n_chains * 16 * A.ld_h);
  long long n_tiles = (n_rows + 15) >> 4;
  int grid = (int)(n_tiles < 8LL * num_cus() ? n_tiles : 8LL * num_cus());
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_ENS(HH)                                                          \
  {                                                                             \
    rc = mbpo_ensure_lds<k_ensemble_forward<HH>>(lds, "ensemble_mlp_forward");          \
    if (rc != MBPO_OK) return rc;                                               \
    hipLaunchKernelGGL(k_ensemble_forward<HH>, dim3(grid), dim3(A.n_chains * 64), lds, st, A); \
  }
  if (H == 64) LAUNCH_ENS(64) else if (H == 128) LAUNCH_ENS(128) else LAUNCH_ENS(256)
#undef LAUNCH_ENS
  MBPO_CHECK_LAUNCH("ensemble_mlp_forward");
  return MBPO_OK;
}

// ------------------------------------------------------------------------------------------------
// R1-R8 fused rollout.
// ------------------------------------------------------------------------------------------------
template <int H>
__global__ void __launch_bounds__(512) k_model_rollout(RolloutArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int X = A.x_dim, U = A.u_dim, D = A.row_len;
  const int E = (A.system_kind == MBPO_SYS_ENSEMBLE) ? A.dyn.n_nets : 0;
  const long long N = A.n_envs;
  const int AR = A.action_repeat;

  // ---- LDS carve (floats) ----
  float *s_obs = smem;                           // [16][ld_x]  current raw obs
  float *s_first = s_obs + 16 * A.ld_x;          // [16][ld_x]
  float *s_pin = s_first + 16 * A.ld_x;          // [16][ld_x]  normalised obs (policy input)
  float *s_xu = s_pin + 16 * A.ld_x;             // [16][ld_xu] dynamics input [x,u]
  const int T = 16 * A.ld_h;
  float *s_pp = s_xu + 16 * A.ld_xu;             // [n_chains][2] hidden tiles (ping-pong per chain)
  float *s_hA = s_pp;                            // also scratch between chains
  float *s_y = s_pp + A.n_chains * 2 * T;        // [n_out][16][ld_y]
  float *s_row = s_y + A.n_out * 16 * A.ld_y;    // [16][D4]
  float *s_steps = s_row + 16 * ((D + 3) & ~3);  // [16]
  float *s_done = s_steps + 16;                  // [16]
  float *s_rew = s_done + 16;                    // [16]

  const RngKey rk_ = rng_resolve(A.seed, A.offset, A.rng_dev);
  const unsigned long long rng_off = rk_.offset, rng_seed = rk_.seed;
  const long long n_tiles = (N + 15) >> 4;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long env0 = tile * 16;
    // ---- load env state ----
    for (int idx = tid; idx < 16 * X; idx += nthreads) {
      int r = idx / X, c = idx - r * X;
      long long env = env0 + r;
      float o = 0.f, f = 0.f;
      if (env < N) {
        o = A.obs[env * X + c];
        f = A.first_obs[env * X + c];
      }
      s_obs[r * A.ld_x + c] = o;
      s_first[r * A.ld_x + c] = f;
    }
    if (tid < 16) {
      long long env = env0 + tid;
      s_steps[tid] = env < N ? A.steps[env] : 0.f;
      s_done[tid] = env < N ? A.done[env] : 0.f;
    }
    __syncthreads();

    for (int s = 0; s < A.n_steps; ++s) {
      if (A.actions) {
        // open-loop actions (rollout_actions, optimizer_utils.py:26-38): no policy
        for (int idx = tid; idx < 16 * U; idx += nthreads) {
          int r = idx / U, d = idx - r * U;
          long long env = env0 + r;
          float a = env < N ? A.actions[((long long)s * N + env) * U + d] : 0.f;
          s_xu[r * A.ld_xu + X + d] = a;
          s_row[r * D + X + d] = a;
        }
      } else {
      // ---- policy input: running_statistics.normalize = (obs - mean) / std ----
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        int r = idx / X, c = idx - r * X;
        float o = s_obs[r * A.ld_x + c];
        s_pin[r * A.ld_x + c] = A.norm_mean ? (o - A.norm_mean[c]) / A.norm_std[c] : o;
      }
      __syncthreads();
      // ---- policy MLP -> logits in s_y[0]: waves 0..3 share the chain (column slices), one barrier per layer ----
      {
        FwdChain fc{&A.policy, A.policy.params, s_pin, A.ld_x, s_pp, s_pp + T, nullptr, nullptr, s_y};
        for (int l = 0; l < A.policy.n_layers; ++l) {
          if (wave < 4) group_fwd_step<HT, 4>(fc, l, A.ld_h, A.ld_y, wave, lane);
          __syncthreads();
        }
      }
      // ---- NormalTanh sample (parametric_distribution.py:97-124) + AutoReset pre-step (training.py:119-124) ----
      for (int idx = tid; idx < 16 * U; idx += nthreads) {
        int r = idx / U, d = idx - r * U;
        long long env = env0 + r;
        float loc = s_y[r * A.ld_y + d], raw = s_y[r * A.ld_y + U + d];
        float sigma = softplus_f(raw) + 0.001f;
        float eps = 0.f;
        if (!A.deterministic && env < N) {
          long long nidx = ((long long)s * N + env) * U + d;
          eps = A.policy_noise ? A.policy_noise[nidx]
                               : philox_normal(rng_seed, rng_off, MBPO_STREAM_POLICY_NOISE, (unsigned long long)nidx);
        }
        float z = loc + sigma * eps;
        float a = tanhf(z);
        if (A.action_clip > 0.f) a = fminf(fmaxf(a, -A.action_clip), A.action_clip);
        s_xu[r * A.ld_xu + X + d] = a;
        s_row[r * D + X + d] = a;
        if (A.ppo_extras) {
          // log N(z; loc, sigma) - log|d tanh/dz|, per action dim; summed below
          float lp = -0.5f * eps * eps - logf(sigma) - 0.91893853320467274178f;
          float ldj = 2.0f * (0.69314718055994530942f - z - softplus_f(-2.0f * z));
          s_row[r * D + 2 * X + U + 2 + 1 + d] = z;               // raw_action
          s_hA[r * U + d] = lp - ldj;                             // scratch: per-dim log-prob
        }
      }
      }
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        int r = idx / X, c = idx - r * X;
        float o = s_obs[r * A.ld_x + c];
        s_xu[r * A.ld_xu + c] = o;
        s_row[r * D + c] = o;  // Transition.observation = env_state.obs (acting.py:47)
      }
      if (tid < 16) {
        // AutoReset: steps <- 0 where previously done; done <- 0
        if (s_done[tid] != 0.f) s_steps[tid] = 0.f;
        s_done[tid] = 0.f;
        s_rew[tid] = 0.f;
      }
      __syncthreads();
      if (A.ppo_extras && tid < 16) {
        float lp = 0.f;
        for (int d = 0; d < U; ++d) lp += s_hA[tid * U + d];
        s_row[tid * D + 2 * X + U + 2] = lp;  // log_prob (summed over action dims)
      }
      if (A.ppo_extras) __syncthreads();

      // ---- EpisodeWrapper inner scan over action_repeat (training.py:91-97) ----
      for (int ar = 0; ar < AR; ++ar) {
        if (A.system_kind == MBPO_SYS_ENSEMBLE) {
          // one wave per ensemble member chain (no barrier inside a chain), n_chains members side by side
          for (int e0 = 0; e0 < E; e0 += A.n_chains) {
            const int e = e0 + wave;
            if (wave < A.n_chains && e < E)
              wave_mlp_fwd<HT>(A.dyn, A.dyn.params + (long long)e * A.dyn.net_stride, s_xu, A.ld_xu, s_pp + wave * 2 * T,
                               s_pp + wave * 2 * T + T, nullptr, nullptr, A.ld_h, s_y + e * 16 * A.ld_y, A.ld_y, lane);
          }
          __syncthreads();
        }
        // reward uses the pre-step x and the action; next state from the system
        if (tid < 16) {
          const int r = tid;
          const float *xr = s_xu + r * A.ld_xu;
          float rew;
          if (A.reward_kind == MBPO_REWARD_PENDULUM) {
            rew = pendulum_reward(xr, xr[X], A.reward_params);
          } else {
            const float *tp = A.reward_params, *qp = tp + X, *rp = qp + X;
            float cx = 0.f, cu = 0.f;
            for (int c = 0; c < X; ++c) { float dd = xr[c] - tp[c]; cx += qp[c] * (dd * dd); }
            for (int d = 0; d < U; ++d) { float uu = xr[X + d]; cu += rp[d] * (uu * uu); }
            rew = -cx - cu;
          }
          s_rew[r] += rew;
        }
        __syncthreads();
        if (A.system_kind == MBPO_SYS_PENDULUM) {
          if (tid < 16) {
            float xn[3];
            pendulum_step(s_xu + tid * A.ld_xu, s_xu[tid * A.ld_xu + X], A.sys_params, xn);
            s_xu[tid * A.ld_xu + 0] = xn[0];
            s_xu[tid * A.ld_xu + 1] = xn[1];
            s_xu[tid * A.ld_xu + 2] = xn[2];
          }
        } else {
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            int r = idx / X, c = idx - r * X;
            long long env = env0 + r;
            float base = A.ens_predict_delta ? s_xu[r * A.ld_xu + c] : 0.f;
            float v;
            if (A.ens_mode == MBPO_ENS_MEAN) {
              float acc = 0.f;
              for (int e = 0; e < E; ++e) acc += s_y[(e * 16 + r) * A.ld_y + c];
              v = base + acc / (float)E;
            } else {
              int mem = 0;
              long long eidx = ((long long)s * AR + ar) * N + env;
              if (env < N) {
                if (A.ens_mode == MBPO_ENS_TSINF) mem = (int)(env % E);
                else mem = A.member_idx ? A.member_idx[eidx]
                                        : philox_randint(rng_seed, rng_off, MBPO_STREAM_MEMBER, (unsigned long long)eidx, 0, E);
              }
              float mu = s_y[(mem * 16 + r) * A.ld_y + c];
              v = base + mu;
              if (A.ens_sample_noise && env < N) {
                float sg = softplus_f(s_y[(mem * 16 + r) * A.ld_y + X + c]) + A.ens_min_std;
                long long nidx = eidx * X + c;
                float eps = A.model_noise ? A.model_noise[nidx]
                                          : philox_normal(rng_seed, rng_off, MBPO_STREAM_MODEL_NOISE, (unsigned long long)nidx);
                v += sg * eps;
              }
            }
            s_hA[r * X + c] = v;  // scratch (s_y must stay intact until every thread has read it)
          }
          __syncthreads();
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            int r = idx / X, c = idx - r * X;
            s_xu[r * A.ld_xu + c] = s_hA[r * X + c];
          }
        }
        __syncthreads();
      }

      // ---- EpisodeWrapper / AutoReset post-step (training.py:98-107, 126-137) + Transition (acting.py:46-55) ----
      if (tid < 16) {
        const int r = tid;
        float st = s_steps[r] + (float)AR;
        float sys_done = 0.f;  // SystemState.done default (base_systems.py:25)
        float dn = (st >= (float)A.episode_length) ? 1.f : sys_done;
        float trunc = (st >= (float)A.episode_length) ? (1.f - sys_done) : 0.f;
        s_steps[r] = st;
        s_done[r] = dn;
        s_row[r * D + X + U] = s_rew[r];
        s_row[r * D + X + U + 1] = 1.f - dn;
        s_row[r * D + D - 1] = trunc;
      }
      __syncthreads();
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        int r = idx / X, c = idx - r * X;
        float v = (s_done[r] != 0.f) ? s_first[r * A.ld_x + c] : s_xu[r * A.ld_xu + c];
        s_obs[r * A.ld_x + c] = v;
        s_row[r * D + X + U + 2 + c] = v;  // next_observation = nstate.obs (post auto-reset)
      }
      __syncthreads();
      // ---- write the tile's 16 rows ----
      for (int idx = tid; idx < 16 * D; idx += nthreads) {
        int r = idx / D, c = idx - r * D;
        long long env = env0 + r;
        if (env < N) {
          long long row = A.env_major ? (env * A.n_steps + s) : ((long long)s * N + env);
          A.transitions[row * D + c] = s_row[idx];
        }
      }
      __syncthreads();
    }

    // ---- write back env state ----
    for (int idx = tid; idx < 16 * X; idx += nthreads) {
      int r = idx / X, c = idx - r * X;
      long long env = env0 + r;
      if (env < N) A.obs[env * X + c] = s_obs[r * A.ld_x + c];
    }
    if (tid < 16) {
      long long env = env0 + tid;
      if (env < N) {
        A.steps[env] = s_steps[tid];
        A.done[env] = s_done[tid];
      }
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------
// H == 64: the rollout on phase runners (chain_run.hpp).  12 waves per 16-env tile:
//   policy phase   waves 0..3 walk the policy chain in lockstep (SP = 4), one barrier per layer;
//   model phase    waves 2e, 2e+1 walk member e's chain (SP = 2), up to 6 members side by side, one barrier per layer —
//                  5 members x 64 MFMAs per layer is 2560 MFMA cycles per SIMD and layer: this phase is fp32-MFMA-throughput
//                  bound on the CU, everything else is arranged to stay out of its way;
//   next-layer weights are requested one step ahead (first layers before the preceding bookkeeping section);
//   bookkeeping sections index with r = idx & 15 (no integer division), use the hardware transcendentals for the NormalTanh
//   sample, and the transition rows are double-buffered so a step's write-out overlaps the next step's first section.
// Barriers per env step (action_repeat 1): policy layers + model layers + 4.
// ------------------------------------------------------------------------------------------------
struct RolloutArgs64 {
  RolloutArgs a;
  NetShape sh_pi, sh_dyn;
  unsigned long long *stamps;   // measurement hook (mbpo_debug_set_rollout_stamps): s_memtime at the section boundaries of tile 0, step 1
};

static unsigned long long *g_ro_stamps = nullptr;
static int g_ro_lean = -1;
// Diagnostic switch (not part of include/mbpo_hip.h): 0 the generic 64-wide rollout kernel, 1 the specialised one, 2 / 3 the specialised one
// with two tiles in flight per workgroup forced on / off, -1 = default.
extern "C" int mbpo_debug_set_rollout_lean(int mode) {
  g_ro_lean = mode;
  return MBPO_OK;
}
extern "C" int mbpo_debug_set_rollout_stamps(void *buf) {
  g_ro_stamps = (unsigned long long *)buf;
  return MBPO_OK;
}
#define RO_STAMP(i)                                                                  \
  if (AA.stamps && blockIdx.x == 0 && s == 1 && tid_ == 0) {                           \
    unsigned long long t_;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
    AA.stamps[i] = t_;                                                               \
  }

// 12 waves (3 per SIMD -> 170 VGPRs each): with 16 waves the 128-register cap spilled 60 VGPRs into the step loop and the PMC
// counters showed 63 MB of scratch writes per launch against 1 MB of transition rows (profiles/r01_pmc_traffic.json).
// (Dealing the members 2,2,2,2,4 waves to level the MFMA load per SIMD was tried: the second runner instantiation brought
// 26 spills back and the kernel got slower, 82 -> 92 us.)
#define RO64_WAVES 12
// The thread id, re-derived where it is needed: wave id (an SGPR since the top of the kernel) * 64 + the lane's position from
// v_mbcnt.  Holding threadIdx.x itself across the step loop made hipcc park it in scratch (one of 168 VGPRs too many): 8 bytes
// per lane stored at the top of every launch — 1.6 MB of the 2.59 MB the PMC counters saw this kernel write for 0.98 MB of rows
// (profiles/r02_pmc_traffic.json).  volatile: not hoisted out of the loops, so nothing has to stay live for it.
__device__ __forceinline__ int ro_tid_now(int wave) {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return (wave << 6) | l;
}

template <bool WIDE>
__global__ void __launch_bounds__(64 * RO64_WAVES) k_model_rollout64(RolloutArgs64 AA) {
  extern __shared__ __align__(16) float smem[];
  const RolloutArgs &A = AA.a;
  constexpr int HT = 4;
  const int tid_ = threadIdx.x, nthreads = 64 * RO64_WAVES;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int X = A.x_dim, U = A.u_dim, D = A.row_len;
  const int E = (A.system_kind == MBPO_SYS_ENSEMBLE) ? A.dyn.n_nets : 0;
  const long long N = A.n_envs;
  const int AR = A.action_repeat;
  const int ld_x = A.ld_x, ld_xu = A.ld_xu, ld_h = A.ld_h, ld_y = A.ld_y;
  const int T = 16 * ld_h;
  const int PL = A.actions ? 0 : AA.sh_pi.L, DL = AA.sh_dyn.L;

  // ---- LDS carve (floats) ----
  float *s_obs = smem;                           // [16][ld_x]  current raw obs
  float *s_first = s_obs + 16 * ld_x;            // [16][ld_x]
  float *s_pin = s_first + 16 * ld_x;            // [16][ld_x]  normalised obs (policy input)
  float *s_xu = s_pin + 16 * ld_x;               // [16][ld_xu] dynamics input [x,u]
  float *s_pp = s_xu + 16 * ld_xu;               // [n_chains][2] hidden tiles (ping-pong per chain)
  float *s_y = s_pp + A.n_chains * 2 * T;        // [n_out][16][ld_y]
  const int D4 = (D + 3) & ~3;
  float *s_rows = s_y + A.n_out * 16 * ld_y;     // [2][16][D]  transition rows, double-buffered over steps
  float *s_steps = s_rows + 2 * 16 * D4;         // [2][16]     double-buffered over steps
  float *s_done = s_steps + 32;                  // [16]
  float *s_rew = s_done + 16;                    // [16]
  float *s_scr = s_rew + 16;                     // [16 * max(X, U)] scratch: next state / per-dim log-prob
  const int SC = X > U ? X : U;
  (void)SC;
  // reward parameters staged once: section C read them from global memory inside a runtime-trip loop (one exposed scalar
  // load per term: ~1.5 k of that section's 3.6 k cycles)
  float *s_rp = s_scr + ((16 * (X + U) + 16 + 3) & ~3);   // [2X+U] quadratic (target, q, r) or [3] pendulum
  {
    const int n_rp = (A.reward_kind == MBPO_REWARD_PENDULUM) ? 3 : 2 * X + U;
    for (int idx = tid_; idx < n_rp; idx += nthreads) s_rp[idx] = A.reward_params[idx];
  }

  const RngKey rk_ = rng_resolve(A.seed, A.offset, A.rng_dev);
  const unsigned long long rng_off = rk_.offset, rng_seed = rk_.seed;
  const long long n_tiles = (N + 15) >> 4;
  const int mchain = wave >> 1, msub = wave & 1;   // member-phase role of this wave
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long env0 = tile * 16;
    {
      const int tid = ro_tid_now(wave);
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        const int r = idx & 15, c = idx >> 4;
        const long long env = env0 + r;
        float o = 0.f, f = 0.f;
        if (env < N) {
          o = A.obs[env * X + c];
          f = A.first_obs[env * X + c];
        }
        s_obs[r * ld_x + c] = o;
        s_first[r * ld_x + c] = f;
      }
      if (tid < 16) {
        const long long env = env0 + tid;
        s_steps[tid] = env < N ? A.steps[env] : 0.f;
        s_done[tid] = env < N ? A.done[env] : 0.f;
      }
    }
    __syncthreads();

#pragma nounroll
    for (int s = 0; s < A.n_steps; ++s) {
      const int tid = ro_tid_now(wave), lane = tid & 63;
      float *s_row = s_rows + (s & 1) * 16 * D4;
      float *steps_cur = s_steps + (s & 1) * 16, *steps_nxt = s_steps + ((s + 1) & 1) * 16;
      WSet<HT, 4> Rp;
      WSet<HT, 2> Rm;
      RO_STAMP(0);
      // ---- section A: policy input, obs into the dynamics input and the row; open-loop actions ----
      if (!A.actions && wave < 4) chain_fwd_prefetch<HT, 4, WIDE>(Rp, AA.sh_pi, A.policy.params, wave, lane);
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        const int r = idx & 15, c = idx >> 4;
        const float o = s_obs[r * ld_x + c];
        s_pin[r * ld_x + c] = A.norm_mean ? (o - A.norm_mean[c]) / A.norm_std[c] : o;   // running_statistics.normalize
        s_xu[r * ld_xu + c] = o;
        s_row[r * D + c] = o;                                                            // Transition.observation (acting.py:47)
      }
      if (A.actions) {
        // open-loop actions (rollout_actions, optimizer_utils.py:26-38): no policy
        for (int idx = tid; idx < 16 * U; idx += nthreads) {
          const int r = idx & 15, d = idx >> 4;
          const long long env = env0 + r;
          const float a = env < N ? A.actions[((long long)s * N + env) * U + d] : 0.f;
          s_xu[r * ld_xu + X + d] = a;
          s_row[r * D + X + d] = a;
        }
      }
      __syncthreads();
      RO_STAMP(1);
      // ---- policy chain -> logits in s_y[0] ----
      if (!A.actions) {
        if (wave < 4) chain_fwd_run<HT, 4, WIDE>(AA.sh_pi, A.policy.params, s_pin, ld_x, s_pp, s_pp + T, nullptr, nullptr, s_y, ld_y, ld_h, PL, wave, lane, Rp);
        else chain_idle_run(PL);
      }
      RO_STAMP(2);
      const bool mactive = (E > 0) && (mchain < A.n_chains) && (mchain < E);
      if (mactive) chain_fwd_prefetch<HT, 2, WIDE>(Rm, AA.sh_dyn, A.dyn.params + (long long)mchain * A.dyn.net_stride, msub, lane);
      // ---- section B: NormalTanh sample (parametric_distribution.py:97-124) + AutoReset pre-step (training.py:119-124) ----
      if (!A.actions) {
        for (int idx = tid; idx < 16 * U; idx += nthreads) {
          const int r = idx & 15, d = idx >> 4;
          const long long env = env0 + r;
          const float loc = s_y[r * ld_y + d], raw = s_y[r * ld_y + U + d];
          const float sigma = ro_fsoftplus(raw) + 0.001f;
          float eps = 0.f;
          if (!A.deterministic && env < N) {
            const long long nidx = ((long long)s * N + env) * U + d;
            eps = A.policy_noise ? A.policy_noise[nidx]
                                 : philox_normal(rng_seed, rng_off, MBPO_STREAM_POLICY_NOISE, (unsigned long long)nidx);
          }
          const float z = loc + sigma * eps;
          float a = ro_ftanh(z);
          if (A.action_clip > 0.f) a = fminf(fmaxf(a, -A.action_clip), A.action_clip);
          s_xu[r * ld_xu + X + d] = a;
          s_row[r * D + X + d] = a;
          if (A.ppo_extras) {
            // log N(z; loc, sigma) - log|d tanh/dz|, per action dim; summed in section D
            const float lp = -0.5f * eps * eps - ro_flog(sigma) - 0.91893853320467274178f;
            const float ldj = 2.0f * (0.69314718055994530942f - z - ro_fsoftplus(-2.0f * z));
            s_row[r * D + 2 * X + U + 2 + 1 + d] = z;               // raw_action
            s_scr[16 * X + r * U + d] = lp - ldj;                   // per-dim log-prob (behind the next-state scratch)
          }
        }
      }
      if (tid < 16) {
        // AutoReset: steps <- 0 where previously done; done <- 0
        if (s_done[tid] != 0.f) steps_cur[tid] = 0.f;
        s_done[tid] = 0.f;
        s_rew[tid] = 0.f;
      }
      __syncthreads();
      RO_STAMP(3);

      // ---- EpisodeWrapper inner scan over action_repeat (training.py:91-97) ----
#pragma nounroll
      for (int ar = 0; ar < AR; ++ar) {
        if (E > 0) {
#pragma nounroll
          for (int e0 = 0; e0 < E; e0 += A.n_chains) {
            const int e = e0 + mchain;
            const bool act = (mchain < A.n_chains) && (e < E);
            if (act && (e0 > 0 || ar > 0))
              chain_fwd_prefetch<HT, 2, WIDE>(Rm, AA.sh_dyn, A.dyn.params + (long long)e * A.dyn.net_stride, msub, lane);
            if (act)
              chain_fwd_run<HT, 2, WIDE>(AA.sh_dyn, A.dyn.params + (long long)e * A.dyn.net_stride, s_xu, ld_xu, s_pp + mchain * 2 * T,
                                   s_pp + mchain * 2 * T + T, nullptr, nullptr, s_y + e * 16 * ld_y, ld_y, ld_h, DL, msub, lane, Rm);
            else
              chain_idle_run(DL);
          }
        }
        RO_STAMP(4);
        // ---- section C: reward on the pre-step (x, u); next state into the scratch ----
        if (tid < 16) {
          const int r = tid;
          const float *xr = s_xu + r * ld_xu;
          float rew;
          if (A.reward_kind == MBPO_REWARD_PENDULUM) {
            rew = pendulum_reward(xr, xr[X], s_rp);
          } else {
            const float *tp = s_rp, *qp = tp + X, *rp = qp + X;
            float cx = 0.f, cu = 0.f;
            for (int c = 0; c < X; ++c) { float dd = xr[c] - tp[c]; cx += qp[c] * (dd * dd); }
            for (int d = 0; d < U; ++d) { float uu = xr[X + d]; cu += rp[d] * (uu * uu); }
            rew = -cx - cu;
          }
          s_rew[r] += rew;
          if (A.system_kind == MBPO_SYS_PENDULUM) {
            float xn[3];
            pendulum_step(xr, xr[X], A.sys_params, xn);
            s_scr[r * X + 0] = xn[0];
            s_scr[r * X + 1] = xn[1];
            s_scr[r * X + 2] = xn[2];
          }
        }
        if (A.system_kind != MBPO_SYS_PENDULUM) {
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx & 15, c = idx >> 4;
            const long long env = env0 + r;
            const float base = A.ens_predict_delta ? s_xu[r * ld_xu + c] : 0.f;
            float v;
            if (A.ens_mode == MBPO_ENS_MEAN) {
              float acc = 0.f;
              for (int e = 0; e < E; ++e) acc += s_y[(e * 16 + r) * ld_y + c];
              v = base + acc / (float)E;
            } else {
              int mem = 0;
              const long long eidx = ((long long)s * AR + ar) * N + env;
              if (env < N) {
                if (A.ens_mode == MBPO_ENS_TSINF) mem = (int)(env % E);
                else mem = A.member_idx ? A.member_idx[eidx]
                                        : philox_randint(rng_seed, rng_off, MBPO_STREAM_MEMBER, (unsigned long long)eidx, 0, E);
              }
              const float mu = s_y[(mem * 16 + r) * ld_y + c];
              v = base + mu;
              if (A.ens_sample_noise && env < N) {
                const float sg = softplus_f(s_y[(mem * 16 + r) * ld_y + X + c]) + A.ens_min_std;
                const long long nidx = eidx * X + c;
                const float eps = A.model_noise ? A.model_noise[nidx]
                                                : philox_normal(rng_seed, rng_off, MBPO_STREAM_MODEL_NOISE, (unsigned long long)nidx);
                v += sg * eps;
              }
            }
            s_scr[r * X + c] = v;
          }
        }
        __syncthreads();
        if (ar + 1 < AR) {   // the next inner step starts from the new state
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx & 15, c = idx >> 4;
            s_xu[r * ld_xu + c] = s_scr[r * X + c];
          }
          __syncthreads();
        }
      }

      RO_STAMP(5);
      // ---- section D: EpisodeWrapper / AutoReset post-step (training.py:98-107, 126-137) + Transition (acting.py:46-55) ----
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        const int r = idx & 15, c = idx >> 4;
        const float st = steps_cur[r] + (float)AR;
        const bool dn = st >= (float)A.episode_length;
        const float v = dn ? s_first[r * ld_x + c] : s_scr[r * X + c];
        s_obs[r * ld_x + c] = v;
        s_row[r * D + X + U + 2 + c] = v;  // next_observation = nstate.obs (post auto-reset)
      }
      if (tid < 16) {
        const int r = tid;
        const float st = steps_cur[r] + (float)AR;
        const float sys_done = 0.f;  // SystemState.done default (base_systems.py:25)
        const float dn = (st >= (float)A.episode_length) ? 1.f : sys_done;
        const float trunc = (st >= (float)A.episode_length) ? (1.f - sys_done) : 0.f;
        steps_nxt[r] = st;
        s_done[r] = dn;
        s_row[r * D + X + U] = s_rew[r];
        s_row[r * D + X + U + 1] = 1.f - dn;
        s_row[r * D + D - 1] = trunc;
        if (A.ppo_extras) {
          float lp = 0.f;
          for (int d = 0; d < U; ++d) lp += s_scr[16 * X + r * U + d];
          s_row[r * D + 2 * X + U + 2] = lp;  // log_prob (summed over action dims)
        }
      }
      __syncthreads();
      RO_STAMP(6);
      // ---- write the tile's 16 rows (the next step's section A works on the other row buffer) ----
      if (A.env_major) {
        for (int r = wave; r < 16; r += RO64_WAVES) {
          const long long env = env0 + r;
          if (env < N)
            for (int c = lane; c < D; c += 64) A.transitions[(env * A.n_steps + s) * D + c] = s_row[r * D + c];
        }
      } else {
        const long long nvalid = (N - env0 < 16 ? N - env0 : 16) * D;
        float *dst = A.transitions + ((long long)s * N + env0) * D;
        for (int idx = tid; idx < nvalid; idx += nthreads) dst[idx] = s_row[idx];
      }
      RO_STAMP(7);
    }
    __syncthreads();
    // ---- write back env state ----
    {
      const int tid = ro_tid_now(wave);
      const float *steps_fin = s_steps + (A.n_steps & 1) * 16;
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        const int r = idx & 15, c = idx >> 4;
        const long long env = env0 + r;
        if (env < N) A.obs[env * X + c] = s_obs[r * ld_x + c];
      }
      if (tid < 16) {
        const long long env = env0 + tid;
        if (env < N) {
          A.steps[env] = steps_fin[tid];
          A.done[env] = s_done[tid];
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Open-loop rollout of the analytic Pendulum system (rollout_actions, utils/optimizer_utils.py:26-38 — the iCEM planner's candidate
// evaluation, icem_optimizer.py:146-160): no network, so a 16-env tile on a 768-thread workgroup has 16 busy lanes per step (the
// tile kernels spent 130 us on 5500 envs x 20 steps).  Here one THREAD owns an env for all steps: the sections of k_model_rollout64 —
// reward on the pre-step (x, u), the step, EpisodeWrapper / AutoReset bookkeeping, the Transition row — with the same device functions
// in the same order: the same rows bit for bit (tests/test_gpu_icem.py).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_openloop_pendulum(RolloutArgs A) {
  const long long env = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long N = A.n_envs;
  if (env >= N) return;
  constexpr int X = 3, U = 1;
  const int D = A.row_len, AR = A.action_repeat;
  float o[3], f[3];
#pragma unroll
  for (int c = 0; c < X; ++c) {
    o[c] = A.obs[env * X + c];
    f[c] = A.first_obs[env * X + c];
  }
  float steps = A.steps[env], done = A.done[env];
  const float rp[3] = {A.reward_params[0], A.reward_params[1], A.reward_params[2]};
  for (int s = 0; s < A.n_steps; ++s) {
    float *row = A.transitions + (A.env_major ? (env * A.n_steps + s) : ((long long)s * N + env)) * D;
    const float a = A.actions[((long long)s * N + env) * U];
#pragma unroll
    for (int c = 0; c < X; ++c) row[c] = o[c];
    row[X] = a;
    if (done != 0.f) steps = 0.f;      // AutoReset pre-step (training.py:119-124)
    done = 0.f;
    float rew = 0.f;
    float xu[4] = {o[0], o[1], o[2], a};
    for (int ar = 0; ar < AR; ++ar) {   // EpisodeWrapper inner scan over action_repeat (training.py:91-97)
      rew += (A.reward_kind == MBPO_REWARD_PENDULUM) ? pendulum_reward(xu, a, rp) : 0.f;
      float xn[3];
      pendulum_step(xu, a, A.sys_params, xn);
      xu[0] = xn[0]; xu[1] = xn[1]; xu[2] = xn[2];
    }
    const float st = steps + (float)AR;
    const bool dn = st >= (float)A.episode_length;
#pragma unroll
    for (int c = 0; c < X; ++c) {
      const float v = dn ? f[c] : xu[c];
      o[c] = v;
      row[X + U + 2 + c] = v;          // next_observation = nstate.obs (post auto-reset)
    }
    row[X + U] = rew;
    row[X + U + 1] = 1.f - (dn ? 1.f : 0.f);
    row[D - 1] = dn ? 1.f : 0.f;       // truncation
    steps = st;
    done = dn ? 1.f : 0.f;
  }
#pragma unroll
  for (int c = 0; c < X; ++c) A.obs[env * X + c] = o[c];
  A.steps[env] = steps;
  A.done[env] = done;
}

extern "C" int mbpo_model_rollout(const mbpo_rollout_desc *d, void *stream) {
  MBPO_REQUIRE(d, MBPO_ERR_ARG, "model_rollout: null descriptor");
  MBPO_REQUIRE(d->x_dim > 0 && d->u_dim > 0, MBPO_ERR_ARG, "model_rollout: x_dim/u_dim must be positive");
  MBPO_REQUIRE(d->n_envs >= 0 && d->n_steps >= 0, MBPO_ERR_ARG, "model_rollout: negative n_envs/n_steps");
  MBPO_REQUIRE(d->episode_length > 0 && d->action_repeat > 0, MBPO_ERR_ARG,
               "model_rollout: episode_length and action_repeat must be positive");
  const bool empty = (d->n_envs == 0 || d->n_steps == 0);  // empty tensors carry NULL data pointers
  MBPO_REQUIRE(empty || (d->obs && d->first_obs && d->steps && d->done && d->transitions), MBPO_ERR_ARG,
               "model_rollout: null state/output pointer");
  MBPO_REQUIRE((d->norm_mean == nullptr) == (d->norm_std == nullptr), MBPO_ERR_ARG,
               "model_rollout: norm_mean and norm_std must both be set or both NULL");
  MBPO_REQUIRE(d->reward_params, MBPO_ERR_ARG, "model_rollout: reward_params is NULL");
  const int X = d->x_dim, U = d->u_dim;
  const int want_row = 2 * X + U + 3 + (d->ppo_extras ? 1 + U : 0);
  MBPO_REQUIRE(d->row_len == want_row, MBPO_ERR_ARG, "model_rollout: row_len %d != expected %d", d->row_len, want_row);
  MBPO_REQUIRE(d->reward_kind == MBPO_REWARD_PENDULUM || d->reward_kind == MBPO_REWARD_QUADRATIC, MBPO_ERR_ARG,
               "model_rollout: unknown reward_kind %d", d->reward_kind);
  MBPO_REQUIRE(d->reward_kind != MBPO_REWARD_PENDULUM || (X == 3 && U == 1), MBPO_ERR_ARG,
               "model_rollout: pendulum reward needs x_dim=3,u_dim=1");
  RolloutArgs A;
  int rc;
  int H = 64;
  const bool has_policy = (d->actions == nullptr);
  MBPO_REQUIRE(has_policy || !d->ppo_extras, MBPO_ERR_ARG, "model_rollout: ppo_extras needs a policy (actions must be NULL)");
  if (has_policy) {
    rc = mbpo_make_mlp_dev(&d->policy, &A.policy, "model_rollout.policy");
    if (rc != MBPO_OK) return rc;
    MBPO_REQUIRE(A.policy.n_nets == 1, MBPO_ERR_ARG, "model_rollout: policy.n_nets must be 1");
    MBPO_REQUIRE(A.policy.dims[0] == X && A.policy.dims[A.policy.n_layers] == 2 * U, MBPO_ERR_ARG,
                 "model_rollout: policy must map [x_dim] -> [2*u_dim]");
    H = hidden_width(A.policy);
  }
  int E = 0;
  int dyn_out = 0;
  if (d->system_kind == MBPO_SYS_ENSEMBLE) {
    rc = mbpo_make_mlp_dev(&d->dynamics, &A.dyn, "model_rollout.dynamics");
    if (rc != MBPO_OK) return rc;
    E = A.dyn.n_nets;
    dyn_out = A.dyn.dims[A.dyn.n_layers];
    MBPO_REQUIRE(A.dyn.dims[0] == X + U, MBPO_ERR_ARG, "model_rollout: dynamics input must be x_dim+u_dim");
    MBPO_REQUIRE(dyn_out == 2 * X || (dyn_out == X && !(d->ens_sample_noise && d->ens_mode != MBPO_ENS_MEAN)),
                 MBPO_ERR_ARG, "model_rollout: dynamics output must be 2*x_dim (mean, raw std) or x_dim (mean only, no sampling)");
    MBPO_REQUIRE(d->ens_mode >= 0 && d->ens_mode <= 2, MBPO_ERR_ARG, "model_rollout: unknown ens_mode");
    int Hd = hidden_width(A.dyn);
    if (!has_policy || A.policy.n_layers == 1) H = Hd;
    MBPO_REQUIRE(Hd == H || A.dyn.n_layers == 1, MBPO_ERR_UNSUPPORTED,
                 "model_rollout: policy and dynamics hidden widths must match (got %d vs %d)", H, Hd);
  } else if (d->system_kind == MBPO_SYS_PENDULUM) {
    MBPO_REQUIRE(X == 3 && U == 1, MBPO_ERR_ARG, "model_rollout: pendulum system needs x_dim=3,u_dim=1");
    MBPO_REQUIRE(d->sys_params, MBPO_ERR_ARG, "model_rollout: sys_params is NULL");
    memset(&A.dyn, 0, sizeof(A.dyn));  // unused
  } else {
    MBPO_REQUIRE(false, MBPO_ERR_ARG, "model_rollout: unknown system_kind %d", d->system_kind);
  }
  MBPO_REQUIRE(H == 64 || H == 128 || H == 256, MBPO_ERR_UNSUPPORTED,
               "model_rollout: hidden layers must share one width in {64,128,256}");
  if (d->n_envs == 0 || d->n_steps == 0) return MBPO_OK;

  A.x_dim = X; A.u_dim = U; A.n_envs = d->n_envs; A.n_steps = d->n_steps;
  A.episode_length = d->episode_length; A.action_repeat = d->action_repeat;
  A.system_kind = d->system_kind; A.ens_mode = d->ens_mode; A.ens_predict_delta = d->ens_predict_delta;
  A.ens_sample_noise = d->ens_sample_noise; A.ens_min_std = d->ens_min_std;
  A.reward_kind = d->reward_kind; A.reward_params = d->reward_params; A.sys_params = d->sys_params;
  A.norm_mean = d->norm_mean; A.norm_std = d->norm_std;
  A.deterministic = d->deterministic; A.ppo_extras = d->ppo_extras; A.env_major = d->env_major;
  A.actions = d->actions;
  A.action_clip = d->action_clip;
  if (!has_policy) memset(&A.policy, 0, sizeof(A.policy));
  A.policy_noise = d->policy_noise; A.model_noise = d->model_noise; A.member_idx = d->member_idx;
  A.seed = d->seed; A.offset = d->offset; A.rng_dev = (const unsigned long long *)d->rng_dev;
  A.obs = d->obs; A.first_obs = d->first_obs; A.steps = d->steps; A.done = d->done;
  A.transitions = d->transitions; A.row_len = d->row_len;
  A.n_out = E > 1 ? E : 1;
  A.ld_x = up4(X) + 4;
  A.ld_xu = up4(X + U) + 4;
  A.ld_h = H + 4;
  int ymax = 2 * U > dyn_out ? 2 * U : dyn_out;
  A.ld_y = up4(ymax) + 4;
  const int scr = 16 * (X + U) + 16;
  const size_t fixed_f = 3ull * 16 * A.ld_x + 16ull * A.ld_xu + (size_t)A.n_out * 16 * A.ld_y + 2ull * 16 * up4(d->row_len) + 80 +
                         (size_t)up4(scr) + (size_t)up4(2 * X + U + 4);   // + staged reward parameters (k_model_rollout64)
  A.n_chains = pick_chains(E, fixed_f, A.ld_h);
  MBPO_REQUIRE(A.n_chains >= 1, MBPO_ERR_UNSUPPORTED, "model_rollout: shapes do not fit 160 KiB of LDS");
  size_t lds = (fixed_f + 2ull * A.n_chains * 16 * A.ld_h) * sizeof(float);
  const int n_waves = A.n_chains > 4 ? A.n_chains : 4;  // the policy chain is shared by waves 0..3
  long long n_tiles = (d->n_envs + 15) >> 4;
  int grid = (int)(n_tiles < 4LL * num_cus() ? n_tiles : 4LL * num_cus());
  hipStream_t st = (hipStream_t)stream;
  if (!has_policy && d->system_kind == MBPO_SYS_PENDULUM && d->reward_kind == MBPO_REWARD_PENDULUM && !d->ppo_extras &&
      !(g_ro_lean == 0)) {      // (mbpo_debug_set_rollout_lean(0): the tile kernel, for the A/B test)
    hipLaunchKernelGGL(k_openloop_pendulum, dim3((unsigned)((d->n_envs + 255) / 256)), dim3(256), 0, st, A);
    MBPO_CHECK_LAUNCH("model_rollout");
    return MBPO_OK;
  }
#define LAUNCH_RO(HH)                                                        \
  {                                                                          \
    rc = mbpo_ensure_lds<k_model_rollout<HH>>(lds, "model_rollout");                 \
    if (rc != MBPO_OK) return rc;                                            \
    hipLaunchKernelGGL(k_model_rollout<HH>, dim3(grid), dim3(n_waves * 64), lds, st, A); \
  }
  if (H == 64) {
    // the kernel specialised for the benchmark networks (rollout_lean.hip): MBPO_ROLLOUT_LEAN=0 / mbpo_debug_set_rollout_lean(0) keep the
    // generic one
    static const int env_lean = getenv("MBPO_ROLLOUT_LEAN") ? atoi(getenv("MBPO_ROLLOUT_LEAN")) : 1;
    const int lean_mode = g_ro_lean >= 0 ? g_ro_lean : env_lean;      // 0 generic, 1 lean, 2 lean + two tiles in flight forced, 3 lean without
    if (lean_mode && rollout_lean_supports(A, has_policy, E)) {
      RoLeanArgs L;
      L.a = A;
      L.E = E;
      L.n_dyn_out = dyn_out;
      L.stamps = g_ro_stamps;
      // two tiles in flight per workgroup once every CU has at least two (MBPO_ROLLOUT_PIPE=0/1 forces it off / on)
      static const int env_pipe = getenv("MBPO_ROLLOUT_PIPE") ? atoi(getenv("MBPO_ROLLOUT_PIPE")) : -1;
      const int pipe_force = lean_mode == 2 ? 1 : (lean_mode == 3 ? 0 : env_pipe);
      const bool pipe = E > 0 && (pipe_force >= 0 ? pipe_force != 0 : n_tiles >= 2LL * num_cus());
      const long long units = pipe ? (n_tiles + 1) / 2 : n_tiles;
      const int lgrid = (int)(units < (long long)num_cus() ? units : (long long)num_cus());
      rc = rollout_lean_launch(L, lgrid, pipe, stream);
      if (rc != MBPO_OK) return rc;
      MBPO_CHECK_LAUNCH("model_rollout");
      return MBPO_OK;
    }
    RolloutArgs64 AA;
    AA.a = A;
    AA.stamps = g_ro_stamps;
    if (has_policy) AA.sh_pi = NetShape{A.policy.dims[0], A.policy.n_layers, A.policy.dims[A.policy.n_layers], A.policy.act};
    else AA.sh_pi = NetShape{X, 0, 2 * U, 0};
    if (E > 0) AA.sh_dyn = NetShape{A.dyn.dims[0], A.dyn.n_layers, A.dyn.dims[A.dyn.n_layers], A.dyn.act};
    else AA.sh_dyn = NetShape{X + U, 0, X, 0};
    if (AA.a.n_chains > RO64_WAVES / 2) AA.a.n_chains = RO64_WAVES / 2;   // member chains of 2 waves side by side
    if (AA.a.n_chains < 1) AA.a.n_chains = 1;
    lds = (fixed_f + 2ull * (AA.a.n_chains > 1 ? AA.a.n_chains : 1) * 16 * A.ld_h) * sizeof(float);
    const bool wide = net_is_wide(AA.sh_pi) || net_is_wide(AA.sh_dyn);
    rc = wide ? mbpo_ensure_lds<k_model_rollout64<true>>(lds, "model_rollout") : mbpo_ensure_lds<k_model_rollout64<false>>(lds, "model_rollout");
    if (rc != MBPO_OK) return rc;
    if (wide) hipLaunchKernelGGL(k_model_rollout64<true>, dim3(grid), dim3(64 * RO64_WAVES), lds, st, AA);
    else hipLaunchKernelGGL(k_model_rollout64<false>, dim3(grid), dim3(64 * RO64_WAVES), lds, st, AA);
  } else if (H == 128) LAUNCH_RO(128) else LAUNCH_RO(256)
#undef LAUNCH_RO
  MBPO_CHECK_LAUNCH("model_rollout");
  return MBPO_OK;
}

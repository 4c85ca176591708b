// rollout_lean.hip — k_rollout_lean<X>: the fused model rollout (R1-R8 of SURVEY §8a: get_experience / generate_unroll, sac/acting.py:
// 25-79 over the BraxWrapper step, brax_utils/training.py:85-137) specialised for the benchmark networks: policy X -> 64 -> 64 -> 64 -> 2
// (or the reference experiments' 64 x 2: one hidden-to-hidden layer less, a run-time count per network),
// members (X + 1) -> 64 -> 64 -> 64 -> (X | 2X), swish, u = 1, X = 2 .. 4, at most five members (or the analytic Pendulum system), action_repeat 1.
//
// The generic k_model_rollout64 (rollout.hip) walks the same 16-env tile per workgroup on the shared runners: every layer of every chain
// re-requests its weights each env step, activations go to LDS as 4 x b32 per lane, four bookkeeping sections of all 768 threads sit
// between the chains — 34.5 k cycles per env step (6 k policy, 19 k members, 9 k bookkeeping; profiles/r03_phase_stamps.txt), 17-19 % of
// the fp32-MFMA roof.  Here (round 4, the blocks of lean_blocks.hpp):
//  * 12 waves = 6 chains x 2 waves; chain 0 is the policy, chains 1..E the members; a wave owns 32 hidden columns (two 16-column MFMA
//    blocks that share the activation reads and interleave their accumulation chains);
//  * every wave keeps ITS network's images — input layer, two hidden layers, output layer — in registers for all steps of all tiles the
//    workgroup walks (both roles run the same code on role-dependent scalars, so the register file holds one network, not two);
//  * the output-layer wave of the policy samples the action from its registers; one wave holds the tile's env state (obs element, first
//    obs, steps, done per lane) in registers across the steps and runs reward + member aggregation + episode / auto-reset bookkeeping +
//    the next step's policy input as ONE section; the policy noise of a step is drawn by an idle member wave during the policy's first
//    layer, and the previous step's rows are written out by the idle member waves at the same time;
//  * 9 workgroup barriers per env step instead of 12.
// Every dot product is the generic kernel's MFMA sequence (operands swapped; the input layer's k groups as wset_fwd_hidden<IN> forms
// them), every elementwise expression is the generic section's: the rows, obs, steps and done are BIT-IDENTICAL to k_model_rollout64's
// (tests/test_gpu_rollout.py::test_rollout_lean_equals_generic_kernel).
#include "common.hpp"
#include "chain_run.hpp"
#include "lean_blocks.hpp"
#include "rollout_shared.hpp"
#include "rollout_lean.hpp"

namespace {
constexpr int RL_WAVES = 12, RL_THREADS = 64 * RL_WAVES;
constexpr int RL_LDY = 20;                          // row stride of a member's output tile (<= 16 outputs)
constexpr int RL_MAX_E = 5;
// LDS carve (floats)
constexpr int R_PIN = 0;                            // [16][8]  normalised obs (policy input)
constexpr int R_XU = 128;                           // [16][8]  [obs | action] (member input, reward)
constexpr int R_EPS = 256;                          // [16] policy noise of the step
constexpr int R_LP = 272;                           // [16] ppo_extras: log-prob of the sampled action
constexpr int R_RP = 288;                           // [16] reward parameters
constexpr int R_NORM = 304;                         // [2][4] normaliser mean, std
constexpr int R_Y = 312;                            // [5][16][RL_LDY] member outputs
constexpr int R_ROWS = R_Y + RL_MAX_E * 16 * RL_LDY;   // [2][16][<= 16] transition rows, double-buffered over steps
constexpr int R_BIAS = R_ROWS + 2 * 16 * 16;        // [6 chains][3 x 64 + 16] the chains' bias vectors (read back per use: 28 registers fewer)
constexpr int RL_NB = 3 * LH + 16;
constexpr int R_TILES = R_BIAS + 6 * RL_NB;         // [6 chains][2] hidden tiles
constexpr int R_SLOT1 = R_TILES + 12 * LT;          // PIPE: the second tile's [pin | xu | eps | lp | rows]: 128 + 128 + 16 + 16 + 256
constexpr int S_PIN = 0, S_XU = 128, S_EPS = 256, S_LP = 272, S_ROW = 288, RL_SLOT_F = 544;
constexpr int R_SLOT0 = R_SLOT1 + RL_SLOT_F;
constexpr size_t RL_LDS_BYTES = (size_t)(R_SLOT0 + RL_SLOT_F) * sizeof(float);
constexpr int SW = 3;                               // the wave that holds the tile's env state
constexpr int NW = 4;                               // the wave that draws the policy noise

// the policy noise of (step s, env env0 + lane) for lanes 0..15
__device__ __forceinline__ float rl_noise(const RolloutArgs &A, int s, long long env0, int lane, unsigned long long rng_seed,
                                          unsigned long long rng_off) {
  const long long env = env0 + lane;
  float eps = 0.f;
  if (!A.deterministic && env < A.n_envs) {
    const long long nidx = ((long long)s * A.n_envs + env);      // u = 1
    eps = A.policy_noise ? A.policy_noise[nidx] : philox_normal(rng_seed, rng_off, MBPO_STREAM_POLICY_NOISE, (unsigned long long)nidx);
  }
  return eps;
}

// NormalTanh sample (parametric_distribution.py:97-124) from the policy output wave's registers, lanes 0..15 (row = lane)
template <int X>
__device__ __forceinline__ void rl_sample(const RolloutArgs &A, const f32x4 acc, const float (&bo)[4], float eps, float *s_xu, float *s_row,
                                          float *s_lp, int D, int r) {
  constexpr int U = 1;
  const float loc = acc[0] + bo[0], raw = acc[1] + bo[1];
  const float sigma = ro_fsoftplus(raw) + 0.001f;
  const float z = loc + sigma * eps;
  float a = ro_ftanh(z);
  if (A.action_clip > 0.f) a = fminf(fmaxf(a, -A.action_clip), A.action_clip);
  s_xu[r * LDX + X] = a;
  s_row[r * D + X] = a;
  if (A.ppo_extras) {
    // log N(z; loc, sigma) - log|d tanh/dz|
    const float lp = -0.5f * eps * eps - ro_flog(sigma) - 0.91893853320467274178f;
    const float ldj = 2.0f * (0.69314718055994530942f - z - ro_fsoftplus(-2.0f * z));
    s_row[r * D + 2 * X + U + 2 + 1] = z;               // raw_action
    s_lp[r] = 0.f + (lp - ldj);                          // (the generic kernel sums the action dims from 0)
  }
}

// One section on the wave that holds a tile's env state (lane (r, c) = (lane & 15, lane >> 4): element c of env r's observation; every
// lane of a row its steps / done): AutoReset pre-step (training.py:119-124), reward on the pre-step (x, u), next state, EpisodeWrapper /
// AutoReset post-step (training.py:98-107, 126-137), Transition (acting.py:46-55), and the next step's inputs (n_row: the row buffer of
// step s + 1, or NULL after the last step).
template <int X, bool PEND>
__device__ __forceinline__ void rl_fused(const RolloutArgs &A, int E, int s, long long env0, int lane, float *smem, float *s_pin, float *s_xu,
                                         const float *s_lp, float *s_row, float *n_row, float &o, float fo, float &steps, float &done,
                                         unsigned long long rng_seed, unsigned long long rng_off) {
  constexpr int U = 1;
  const int sr = lane & 15, sc = lane >> 4, D = A.row_len;
  const long long N = A.n_envs;
  const bool s_ok = sc < X;
  if (done != 0.f) steps = 0.f;
  done = 0.f;
  const float *xr = s_xu + sr * LDX;
  const float *s_rp = smem + R_RP;
  float rew;
  if (PEND && A.reward_kind == MBPO_REWARD_PENDULUM) {
    rew = pendulum_reward(xr, xr[X], s_rp);
  } else {
    const float *tp = s_rp, *qp = tp + X, *rp = qp + X;
    float cx = 0.f, cu = 0.f;
#pragma unroll
    for (int c = 0; c < X; ++c) { float dd = xr[c] - tp[c]; cx += qp[c] * (dd * dd); }
#pragma unroll
    for (int d = 0; d < U; ++d) { float uu = xr[X + d]; cu += rp[d] * (uu * uu); }
    rew = -cx - cu;
  }
  rew = 0.f + rew;                                   // (s_rew starts the step at zero in the generic kernel)
  float v = 0.f;
  if (PEND && A.system_kind == MBPO_SYS_PENDULUM) {
    float xn[3];
    pendulum_step(xr, xr[X], A.sys_params, xn);
    v = sc == 0 ? xn[0] : (sc == 1 ? xn[1] : xn[2]);
  } else if (s_ok) {
    const int r = sr, c = sc;
    const long long env = env0 + r;
    const float *s_y = smem + R_Y;
    const float base = A.ens_predict_delta ? xr[c] : 0.f;
    if (A.ens_mode == MBPO_ENS_MEAN) {
      float acc = 0.f;
      for (int e = 0; e < E; ++e) acc += s_y[(e * 16 + r) * RL_LDY + c];
      v = base + acc / (float)E;
    } else {
      int mem = 0;
      const long long eidx = (long long)s * N + env;
      if (env < N) {
        if (A.ens_mode == MBPO_ENS_TSINF) mem = (int)(env % E);
        else mem = A.member_idx ? A.member_idx[eidx] : philox_randint(rng_seed, rng_off, MBPO_STREAM_MEMBER, (unsigned long long)eidx, 0, E);
      }
      const float mu = s_y[(mem * 16 + r) * RL_LDY + c];
      v = base + mu;
      if (A.ens_sample_noise && env < N) {
        const float sg = softplus_f(s_y[(mem * 16 + r) * RL_LDY + X + c]) + A.ens_min_std;
        const long long nidx = eidx * X + c;
        const float eps = A.model_noise ? A.model_noise[nidx] : philox_normal(rng_seed, rng_off, MBPO_STREAM_MODEL_NOISE, (unsigned long long)nidx);
        v += sg * eps;
      }
    }
  }
  const float st = steps + 1.0f;
  const bool dnb = st >= (float)A.episode_length;
  const float v2 = dnb ? fo : v;
  if (s_ok) s_row[sr * D + X + U + 2 + sc] = v2;      // next_observation = nstate.obs (post auto-reset)
  if (lane < 16) {
    const float dn = dnb ? 1.f : 0.f;                // SystemState.done defaults to 0 (base_systems.py:25)
    s_row[sr * D + X + U] = rew;
    s_row[sr * D + X + U + 1] = 1.f - dn;
    s_row[sr * D + D - 1] = dnb ? 1.f : 0.f;         // truncation
    if (A.ppo_extras) s_row[sr * D + 2 * X + U + 2] = s_lp[sr];
  }
  steps = st;
  done = dnb ? 1.f : 0.f;
  o = v2;
  if (n_row && s_ok) {                               // section A of the next step
    s_pin[sr * LDX + sc] = A.norm_mean ? (o - smem[R_NORM + sc]) / smem[R_NORM + 4 + sc] : o;      // running_statistics.normalize
    s_xu[sr * LDX + sc] = o;
    n_row[sr * D + sc] = o;                          // Transition.observation (acting.py:47)
  }
}
}  // namespace

#define RL_STAMP(i)                                                                  \
  if (AA.stamps && blockIdx.x == 0 && s == 1 && tid == 0) {                          \
    unsigned long long t_;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
    AA.stamps[i] = t_;                                                               \
  }

// PEND: the analytic Pendulum step / reward are compiled in (their atan2f / sinf / cosf / fmodf expansions need ~40 more registers than
// the resident weights leave; the ensemble + quadratic-reward instantiation — the benchmark's — carries neither)
// PIPE: two tiles in flight per workgroup — the policy phase of one beside the member phase of the other (see the pipelined loop below)
template <int X, bool PEND, bool PIPE>
__global__ void __launch_bounds__(RL_THREADS) k_rollout_lean(const RoLeanArgs AA) {
  extern __shared__ __align__(16) float smem[];
  const RolloutArgs &A = AA.a;
  constexpr int K = X + 1, U = 1;
  const int tid = threadIdx.x, lane_ = tid & 63, lane = lane_;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain = wave >> 1, sub2 = wave & 1, c0 = 32 * sub2;
  const int E = AA.E, D = A.row_len, D4 = (D + 3) & ~3;
  const long long N = A.n_envs;
  const bool pol = chain == 0;
  const bool member = !pol && chain <= E;
  const bool out_wave = (pol || member) && sub2 == 0;
  // ---- this wave's network: the policy, or member chain - 1 (idle chains read the policy's words and compute nothing) ----
  const float *const net_p = member ? A.dyn.params + (long long)(chain - 1) * A.dyn.net_stride : A.policy.params;
  const int Kr = member ? K : X, Nr = member ? AA.n_dyn_out : 2;
  const int kc = (Kr + 3) >> 2;
  const int nh = (member ? A.dyn.n_layers : A.policy.n_layers) - 2;      // 64 x 64 layers of this wave's network: 2 (64 x 3 nets) or 1 (64 x 2)
  const int W1 = Kr * LH + LH, OUT = W1 + nh * HID;
  const int i16 = lane & 15, g = lane >> 4;
  // The small vectors that go to LDS are requested FIRST: vector-memory results return in order, so a store of something requested
  // behind the 84 image requests waits for all of them (the bias table written at the end of the request list held every wave at the
  // top of the kernel until its whole network had arrived).
  float bv0 = 0.f, bv1 = 0.f, bv2 = 0.f;
  if (lane < 32) {
    bv0 = net_p[Kr * LH + c0 + lane];
    bv1 = net_p[W1 + LH * LH + c0 + lane];
    bv2 = net_p[W1 + (nh - 1) * HID + LH * LH + c0 + lane];
  } else if (sub2 == 0 && lane < 48) {
    const int q = lane - 32;
    bv0 = (q < Nr) ? net_p[OUT + LH * Nr + q] : 0.f;
  }
  Img0 I0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int k = g * kc + s;
      const bool ok = (s < kc) && (k < Kr);
      const float v = net_p[(ok ? k : 0) * LH + c0 + 16 * t + i16];
      I0.w[t][s] = ok ? v : 0.f;
    }
  }
  float w1a[16], w1b[16], w2a[16], w2b[16];
  img_w_request(w1a, net_p + W1, c0, lane);
  img_w_request(w1b, net_p + W1, c0 + 16, lane);
  img_w_request(w2a, net_p + W1 + (nh - 1) * HID, c0, lane);      // (a network with one 64 x 64 layer: requested again, never used)
  img_w_request(w2b, net_p + W1 + (nh - 1) * HID, c0 + 16, lane);
  float wo[16];
  {
    const float *p = net_p + OUT + (16 * g) * Nr + (i16 < Nr ? i16 : 0);      // matrix row i = output column i
#pragma unroll
    for (int s = 0; s < 16; ++s) wo[s] = out_wave ? p[s * Nr] : 0.f;
  }
  // the chain's bias vectors -> LDS
  float *const bias = smem + R_BIAS + chain * RL_NB;
  if (lane < 32) {
    bias[c0 + lane] = bv0;
    bias[LH + c0 + lane] = bv1;
    bias[2 * LH + c0 + lane] = bv2;
  } else if (sub2 == 0 && lane < 48) {
    bias[3 * LH + (lane - 32)] = bv0;
  }
  float *const tiles = smem + R_TILES + chain * 2 * LT;
  {
    const int n_rp = (A.reward_kind == MBPO_REWARD_PENDULUM) ? 3 : 2 * X + U;
    if (tid < n_rp) smem[R_RP + tid] = A.reward_params[tid];
  }
  const RngKey rk_ = rng_resolve(A.seed, A.offset, A.rng_dev);
  const unsigned long long rng_off = rk_.offset, rng_seed = rk_.seed;
  // the state wave: lane (r, c) = (lane & 15, lane >> 4) holds element c of env r's observation; every lane of a row its steps / done
  const int sr = lane & 15, sc = lane >> 4;
  const bool s_ok = sc < X;
  if (tid < X) {
    smem[R_NORM + tid] = A.norm_mean ? A.norm_mean[tid] : 0.f;
    smem[R_NORM + 4 + tid] = A.norm_mean ? A.norm_std[tid] : 1.f;
  }
  __syncthreads();

  const long long n_tiles = (N + 15) >> 4;
  if constexpr (PIPE) {
    // ---- two tiles in flight: half-step h runs the policy phase of tile slot h & 1 (its step h >> 1) on chain 0 BESIDE the member
    //      phase of the other slot (its step (h - 1) >> 1) on chains 1..E, over the same four barriers; then the member tile's state
    //      wave runs its bookkeeping section and writes the finished rows out itself.  2 S + 1 half-steps of max(policy, members)
    //      for two tiles x S steps, instead of 2 S steps of policy + members; 5 barriers per tile-step instead of 9.
    constexpr int SWB = 5;                            // slot 1's state wave (slot 0's: SW)
    const bool is_sw = wave == SW || wave == SWB;
    const int my_slot = wave == SWB ? 1 : 0;
    const long long n_pairs = (n_tiles + 1) >> 1;
    const int S = A.n_steps;
#pragma nounroll
    for (long long pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
      const bool validB = 2 * pair + 1 < n_tiles;
      const long long my_env0 = (2 * pair + my_slot) * 16;
      const bool my_valid = my_slot == 0 || validB;
      float o = 0.f, fo = 0.f, steps = 0.f, done = 0.f;
      float *const my_base = smem + (my_slot ? R_SLOT1 : R_SLOT0);
      if (is_sw && my_valid) {
        const long long env = my_env0 + sr;
        if (env < N) {
          if (s_ok) {
            o = A.obs[env * X + sc];
            fo = A.first_obs[env * X + sc];
          }
          steps = A.steps[env];
          done = A.done[env];
        }
        if (s_ok) {      // section A of step 0
          my_base[S_PIN + sr * LDX + sc] = A.norm_mean ? (o - smem[R_NORM + sc]) / smem[R_NORM + 4 + sc] : o;
          my_base[S_XU + sr * LDX + sc] = o;
          my_base[S_ROW + sr * D + sc] = o;
        }
      }
      if (wave == 1 && lane < 16 && S > 0) smem[R_SLOT0 + S_EPS + lane] = rl_noise(A, 0, 2 * pair * 16, lane, rng_seed, rng_off);
      __syncthreads();
#pragma nounroll
      for (int h = 0; h <= 2 * S && S > 0; ++h) {
        const int lane = opaque(lane_);
        const int i16 = lane & 15, g = lane >> 4;
        const int ps = h & 1, sp = h >> 1, ms = ps ^ 1, sm = (h - 1) >> 1;
        const bool pv = h < 2 * S && (ps == 0 || validB), mv = h >= 1 && (ms == 0 || validB);
        const bool mine = pol ? pv : (member && mv);
        float *const base = smem + ((pol ? ps : ms) ? R_SLOT1 : R_SLOT0);
        if (mine) in_fwd2(I0, bias, base + (pol ? S_PIN : S_XU), kc, Kr, tiles, c0, lane);
        __syncthreads();
        if (mine) hid_fwd2(w1a, w1b, bias + LH, tiles, tiles + LT, c0, lane);
        __syncthreads();
        if (mine && nh == 2) hid_fwd2(w2a, w2b, bias + 2 * LH, tiles + LT, tiles, c0, lane);
        __syncthreads();
        if (mine && sub2 == 0) {
          const f32x4 acc = out_fwd(wo, nh == 2 ? tiles : tiles + LT, lane);
          float bo[4];
          load_vec_lds<4>(bias + 3 * LH + 4 * g, bo);
          if (!pol) {
            const float yv[4] = {acc[0] + bo[0], acc[1] + bo[1], acc[2] + bo[2], acc[3] + bo[3]};
            store_vec_lds<4>(smem + R_Y + ((chain - 1) * 16 + i16) * RL_LDY + 4 * g, yv);
          } else if (lane < 16) {
            rl_sample<X>(A, acc, bo, base[S_EPS + lane], base + S_XU, base + S_ROW, base + S_LP, D, lane);
          }
        } else if (wave == 1 && lane < 16) {
          // the noise of the NEXT half-step's policy tile (this wave idles through the output layer)
          const int nxt = h + 1, nslot = nxt & 1;
          if (nxt < 2 * S && (nslot == 0 || validB))
            smem[(nslot ? R_SLOT1 : R_SLOT0) + S_EPS + lane] = rl_noise(A, nxt >> 1, (2 * pair + nslot) * 16, lane, rng_seed, rng_off);
        }
        __syncthreads();
        if (is_sw && my_slot == ms && mv) {
          float *const row = my_base + S_ROW;
          rl_fused<X, PEND>(A, E, sm, my_env0, lane, smem, my_base + S_PIN, my_base + S_XU, my_base + S_LP, row, nullptr, o, fo, steps, done,
                            rng_seed, rng_off);
          // the finished rows of step sm leave from here (this wave's own LDS writes and reads stay in order); then the next step's inputs
          if (A.env_major) {
            for (int r = 0; r < 16; ++r) {
              const long long env = my_env0 + r;
              if (env < N && lane < D) A.transitions[(env * S + sm) * D + lane] = row[r * D + lane];
            }
          } else {
            const int nvalid = (int)(N - my_env0 < 16 ? N - my_env0 : 16) * D;
            float *dst = A.transitions + ((long long)sm * N + my_env0) * D;
            for (int idx = lane; idx < nvalid; idx += 64) dst[idx] = row[idx];
          }
          if (sm + 1 < S && s_ok) {
            my_base[S_PIN + sr * LDX + sc] = A.norm_mean ? (o - smem[R_NORM + sc]) / smem[R_NORM + 4 + sc] : o;
            my_base[S_XU + sr * LDX + sc] = o;
            row[sr * D + sc] = o;
          }
        }
        __syncthreads();
      }
      if (is_sw && my_valid) {
        const long long env = my_env0 + sr;
        if (env < N) {
          if (s_ok) A.obs[env * X + sc] = o;
          if (lane < 16) {
            A.steps[env] = steps;
            A.done[env] = done;
          }
        }
      }
      __syncthreads();
    }
    return;
  }
#pragma nounroll
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long env0 = tile * 16;
    float o = 0.f, fo = 0.f, steps = 0.f, done = 0.f;
    if (wave == SW) {
      const long long env = env0 + sr;
      if (env < N) {
        if (s_ok) {
          o = A.obs[env * X + sc];
          fo = A.first_obs[env * X + sc];
        }
        steps = A.steps[env];
        done = A.done[env];
      }
      if (s_ok) {      // section A of step 0
        smem[R_PIN + sr * LDX + sc] = A.norm_mean ? (o - smem[R_NORM + sc]) / smem[R_NORM + 4 + sc] : o;
        smem[R_XU + sr * LDX + sc] = o;
        smem[R_ROWS + sr * D + sc] = o;
      }
    }
    __syncthreads();

#pragma nounroll
    for (int s = 0; s < A.n_steps; ++s) {
      float *const s_row = smem + R_ROWS + (s & 1) * 16 * D4;
      // ---- phase 0: the policy chain (| noise of this step, rows of the previous step on idle waves); phase 1: the members side by
      //      side.  One loop body for both: every block has ONE call site, a wave runs it on its own network's registers ----
#pragma nounroll
      for (int ph = 0; ph < (E > 0 ? 2 : 1); ++ph) {
        const int lane = opaque(lane_);      // per-lane addresses are re-derived per phase: hoisted out of the loops they fill the register file
        const int i16 = lane & 15, g = lane >> 4;
        const bool mine = ph == 0 ? pol : member;
        RL_STAMP(4 * ph);
        if (mine) {
          in_fwd2(I0, bias, smem + (ph == 0 ? R_PIN : R_XU), kc, Kr, tiles, c0, lane);
        } else if (ph == 0) {
          if (wave == NW && lane < 16) smem[R_EPS + lane] = rl_noise(A, s, env0, lane, rng_seed, rng_off);
          if (wave > NW && s > 0) {
            const float *rb = smem + R_ROWS + ((s - 1) & 1) * 16 * D4;
            if (A.env_major) {
              for (int r = wave - (NW + 1); r < 16; r += RL_WAVES - (NW + 1)) {
                const long long env = env0 + r;
                if (env < N)
                  for (int c = lane; c < D; c += 64) A.transitions[(env * A.n_steps + (s - 1)) * D + c] = rb[r * D + c];
              }
            } else {
              const int nvalid = (int)(N - env0 < 16 ? N - env0 : 16) * D;
              float *dst = A.transitions + ((long long)(s - 1) * N + env0) * D;
              for (int idx = (wave - (NW + 1)) * 64 + lane; idx < nvalid; idx += RL_THREADS - 64 * (NW + 1)) dst[idx] = rb[idx];
            }
          }
        }
        __syncthreads();
        RL_STAMP(4 * ph + 1);
        if (mine) hid_fwd2(w1a, w1b, bias + LH, tiles, tiles + LT, c0, lane);
        __syncthreads();
        RL_STAMP(4 * ph + 2);
        if (mine && nh == 2) hid_fwd2(w2a, w2b, bias + 2 * LH, tiles + LT, tiles, c0, lane);
        __syncthreads();
        RL_STAMP(4 * ph + 3);
        if (mine && sub2 == 0) {
          const f32x4 acc = out_fwd(wo, nh == 2 ? tiles : tiles + LT, lane);
          float bo[4];
          load_vec_lds<4>(bias + 3 * LH + 4 * g, bo);
          if (ph == 1) {
            const float yv[4] = {acc[0] + bo[0], acc[1] + bo[1], acc[2] + bo[2], acc[3] + bo[3]};
            store_vec_lds<4>(smem + R_Y + ((chain - 1) * 16 + i16) * RL_LDY + 4 * g, yv);      // (columns >= Nr: never read)
          } else if (lane < 16) {
            rl_sample<X>(A, acc, bo, smem[R_EPS + lane], smem + R_XU, s_row, smem + R_LP, D, lane);
          }
        }
        __syncthreads();
      }
      RL_STAMP(8);
      // ---- one section on the state wave: AutoReset pre-step (training.py:119-124), reward on the pre-step (x, u), next state,
      //      EpisodeWrapper / AutoReset post-step (training.py:98-107, 126-137), Transition (acting.py:46-55), next step's inputs ----
      if (wave == SW)
        rl_fused<X, PEND>(A, E, s, env0, opaque(lane_), smem, smem + R_PIN, smem + R_XU, smem + R_LP, s_row,
                          s + 1 < A.n_steps ? smem + R_ROWS + ((s + 1) & 1) * 16 * D4 : nullptr, o, fo, steps, done, rng_seed, rng_off);
      __syncthreads();
      RL_STAMP(9);
    }
    // ---- the last step's rows, the env state ----
    if (A.n_steps > 0) {
      const int ls = A.n_steps - 1;
      const float *rb = smem + R_ROWS + (ls & 1) * 16 * D4;
      if (A.env_major) {
        for (int r = wave; r < 16; r += RL_WAVES) {
          const long long env = env0 + r;
          if (env < N)
            for (int c = lane; c < D; c += 64) A.transitions[(env * A.n_steps + ls) * D + c] = rb[r * D + c];
        }
      } else {
        const int nvalid = (int)(N - env0 < 16 ? N - env0 : 16) * D;
        float *dst = A.transitions + ((long long)ls * N + env0) * D;
        for (int idx = tid; idx < nvalid; idx += RL_THREADS) dst[idx] = rb[idx];
      }
    }
    if (wave == SW) {
      const long long env = env0 + sr;
      if (env < N) {
        if (s_ok) A.obs[env * X + sc] = o;
        if (lane < 16) {
          A.steps[env] = steps;
          A.done[env] = done;
        }
      }
    }
    __syncthreads();
  }
}

bool rollout_lean_supports(const RolloutArgs &A, bool has_policy, int E) {
  if (!has_policy || A.actions) return false;
  const int X = A.x_dim;
  if (A.u_dim != 1 || X < 2 || X > 4 || A.action_repeat != 1) return false;      // (the state wave holds 16 envs x <= 4 obs elements)
  auto net_ok = [](const MlpDev &m, int k_in) {
    if ((m.n_layers != 4 && m.n_layers != 3) || m.act != MBPO_ACT_SWISH || m.dims[0] != k_in) return false;
    for (int l = 1; l < m.n_layers; ++l)
      if (m.dims[l] != LH) return false;
    return true;
  };
  if (!net_ok(A.policy, X) || A.policy.dims[A.policy.n_layers] != 2) return false;
  if (A.system_kind == MBPO_SYS_ENSEMBLE) {
    if (E < 1 || E > RL_MAX_E || !net_ok(A.dyn, X + 1)) return false;
    if (A.dyn.dims[A.dyn.n_layers] != X && A.dyn.dims[A.dyn.n_layers] != 2 * X) return false;
  } else if (A.system_kind != MBPO_SYS_PENDULUM || X != 3) {
    return false;
  }
  return A.row_len <= 16;
}

int rollout_lean_launch(const RoLeanArgs &A, int grid, bool pipe, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const bool pend = A.a.system_kind == MBPO_SYS_PENDULUM || A.a.reward_kind == MBPO_REWARD_PENDULUM;
#define RL_LAUNCH(X_, P_, PP_)                                                                              \
  {                                                                                                         \
    rc = mbpo_ensure_lds<k_rollout_lean<X_, P_, PP_>>(RL_LDS_BYTES, "rollout_lean");                        \
    if (rc != MBPO_OK) return rc;                                                                           \
    hipLaunchKernelGGL((k_rollout_lean<X_, P_, PP_>), dim3(grid), dim3(RL_THREADS), RL_LDS_BYTES, st, A);   \
  }
  if (pipe) {
    if (A.a.x_dim == 3) {
      if (pend) RL_LAUNCH(3, true, true) else RL_LAUNCH(3, false, true)
    } else if (A.a.x_dim == 2) {
      RL_LAUNCH(2, false, true)
    } else {
      RL_LAUNCH(4, false, true)
    }
  } else {
    if (A.a.x_dim == 3) {
      if (pend) RL_LAUNCH(3, true, false) else RL_LAUNCH(3, false, false)
    } else if (A.a.x_dim == 2) {
      RL_LAUNCH(2, false, false)
    } else {
      RL_LAUNCH(4, false, false)
    }
  }
#undef RL_LAUNCH
  return MBPO_OK;
}

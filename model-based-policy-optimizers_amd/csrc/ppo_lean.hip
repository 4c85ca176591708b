// ppo_lean.hip — k_ppo_lean<X>: PPOLoss.loss forward/backward (ppo/losses.py:56-126) for a minibatch of B x T rows, specialised for the
// benchmark networks: policy X -> 64 -> 64 -> 64 -> 2, value X -> 64 -> 64 -> 64 -> 1, swish, u = 1 (BASELINE config 3).
//
// The generic k_ppo_fwd_bwd (ppo.hip) is a latency chain per 16-row tile on the shared runners: 128 VGPRs with 21-64 spilled, weight
// images re-requested for every layer of every tile, weight gradients accumulated into the slab by global read-modify-write per tile,
// two 512-thread workgroups per CU to overlap two tiles' chains — 66 us per launch at C3 (1280 tiles), 13.6 % of the fp32-MFMA roof.
// Here (round 4, the blocks of csrc/lean_blocks.hpp as in k_sac_lean):
//  * ONE 8-wave workgroup per CU with 256 VGPRs per wave walks its tiles; chain c = the network (0 policy, 1 value), 4 waves each.
//  * Every weight image a wave needs — thin column, two hidden forward images, output image, two input-gradient images — is loaded ONCE
//    per launch and stays in registers for all tiles (the generic kernel issues ~100 requests per wave and tile).
//  * Weight gradients are summed over the workgroup's tiles IN REGISTERS (each tile's tile-sum formed in fresh accumulators and
//    added: the order of the generic kernel's read-modify-write, without the traffic) and stored once at the end: no per-tile slab
//    traffic, and one slab per CU instead of two (the two-stage slab sum reads half the bytes).
//  * Stored derivative instead of pre-activation (swish'(z) written by the forward pass; the delta tiles overwrite them in place);
//    all input gradients first, then the weight gradients of both hidden layers and both thin layers with no barrier between them.
//  * The next tile's rows are requested one tile ahead; its entropy noise is drawn by an idle wave during the output-layer step.
// Measured and NOT kept (round 4, scripts/ppo_lean_dev.py stamps; 15 k cycles per tile, 4.5 k of them the weight gradients): forming a
// tile's weight gradients inside the next tile's thin / tile-load steps (double-buffered tile sets) — those steps have no idle
// waves, the work just moves (15.1 -> 16.0 k); giving the six waves that idle during the output-layer step the previous tile's
// hidden-layer gradients — the output wave they share a SIMD with slows down by what the phase at the end gains (15.4 k); the thin
// layers' gradients as 4-MFMA tiles instead of FMA loops (-0.65 k) with the input-gradient images re-requested per tile to pay for
// their 16 accumulator registers (+1.3 k: the requests are not back in two steps); the advantage-moment combine (k_ppo_moments_combine, a
// one-workgroup launch in front of this one) done by every workgroup in its own prologue, before the weights are requested: 74.6 ->
// 74.1 us per minibatch step at T = 40 — the launch it removes costs ~3 us in the graph, the extra round trip and four barriers at the
// top of this kernel 1.6 us, and the x = 4 instantiation went from 11 to 25 spilled VGPRs.
#include "common.hpp"
#include "chain_run.hpp"
#include "lean_blocks.hpp"
#include "ppo_lean.hpp"

namespace {

#define LOG_SQRT_2PI 0.91893853320467274178f
#define LOG_2 0.69314718055994530942f

__device__ __forceinline__ float pl_fexp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
__device__ __forceinline__ float pl_flog(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float pl_fsoftplus(float x) { return fmaxf(x, 0.0f) + pl_flog(1.0f + pl_fexp(-fabsf(x))); }
__device__ __forceinline__ float pl_ftanh(float x) {
  const float e = pl_fexp(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
  return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}

template <int X, int NH>      // NH: 64 x 64 layers per network — 2 (64 x 3 networks) or 1 (64 x 2: tests/test_ppo.py, the reference's experiments)
struct PNet {
  static constexpr int D = 2 * X + 6;                                     // obs, action, reward, discount, next_obs, log_prob, raw_action, truncation
  static constexpr int W1 = X * LH + LH, OUT = W1 + NH * HID;             // offsets inside a net: layer 1, output layer
  static constexpr int P = OUT + LH * 2 + 2, V = OUT + LH + 1;
};

// LDS carve (floats)
constexpr int P_X = 0;                 // [2][16][8]  normalised observations, double-buffered on the tile parity
constexpr int P_AUX = 256;             // [2][16][4]  raw_action, behaviour log-prob, advantage, value target
constexpr int P_EPS = 384;             // [2][16]     entropy-sample noise
constexpr int P_DY = 416;              // [2][16][4]  output gradients: policy (d/dloc, d/draw), value (d/dV)
constexpr int P_LOSS = 544;            // [3][16]     loss partials at the end
constexpr int P_TILES = 592;           // 12 tiles: net c at + 6 c: d0 d1 d2 (swish'(z), then the deltas in place) | h0 h1 h2
constexpr size_t PPO_LEAN_LDS_BYTES = (size_t)(P_TILES + 12 * LT) * sizeof(float);

// weight gradient of a hidden layer for this wave's 16 columns, tile sum in fresh accumulators (hid_wgrad of lean_blocks.hpp without
// the store): acc[a] lane (j, g) reg i = dW[4 j + a][c0 + 4 g + i], accb lane (0, g) reg i = db[c0 + 4 g + i]
__device__ __forceinline__ void hid_wgrad_regs(const float *hin, const float *delta, int c0, int lane, f32x4 (&acc)[4], f32x4 &accb) {
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb = (f32x4){0.f, 0.f, 0.f, 0.f};
  float hv[4][4], dv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = 4 * g + s;
    load_vec_lds<4>(hin + row * LDH + 4 * r, hv[s]);
    dv[s] = delta[row * LDH + c0 + r];
  }
  const float one0 = (r == 0) ? 1.f : 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = MFMA(dv[s], hv[s][a], acc[a]);
    accb = MFMA(dv[s], one0, accb);
  }
}

}  // namespace

#define PPO_STAMP(i)                                                                    \
  if (A.stamps && blockIdx.x == 0 && tid == 0 && stamp_base + (i) < 64) {               \
    unsigned long long t_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    A.stamps[stamp_base + (i)] = t_;                                                    \
  }

template <int X, int NH>
__global__ void __launch_bounds__(512) k_ppo_lean(const PpoLeanArgs A) {
  int stamp_base = 0;
  extern __shared__ __align__(16) float smem[];
  using N = PNet<X, NH>;
  constexpr int D = N::D;
  constexpr int HL = NH;        // index of the last hidden layer: its derivative / delta tile PT(HL), its activation tile PT(3 + HL)
  static_assert(16 * D + 32 <= 512 && X <= LDX, "tile rows + advantage / target lanes must fit the workgroup");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wave >> 2, sub = wave & 3, c0 = sub * 16;        // chain c = network: 0 policy, 1 value
  PPO_STAMP(0);
  const long long M = A.M;
  const long long n_tiles = (M + 15) >> 4;
  const float invM = 1.0f / (float)M;
  const float *const net_p = A.params + (c ? N::P : 0);
  float *const tiles = smem + P_TILES + c * 6 * LT;
#define PT(n) (tiles + (n) * LT)      /* 0..2: stored derivatives / deltas of layers 0..2, 3..5: activations h0..h2 */

  // ---- per-thread role in the tile load: one dword of the tile's 16 x D block, or an advantage / value-target element ----
  const bool has_elem = tid < 16 * D;
  const int r_t = tid / D, cc_t = tid - r_t * D;
  const bool is_obs = has_elem && cc_t < X;
  const bool is_adv = tid >= 16 * D && tid < 16 * D + 16, is_vs = tid >= 16 * D + 16 && tid < 16 * D + 32;
  float mu_t = 0.f, sd_t = 1.f;
  if (is_obs && A.norm_mean) {
    mu_t = A.norm_mean[cc_t];
    sd_t = A.norm_std[cc_t];
  }
  const float adv_mean = A.normalize_advantage ? A.mom[0] : 0.f;
  const float adv_istd = A.normalize_advantage ? 1.0f / (A.mom[1] + 1e-8f) : 1.f;          // losses.py:101-102
  const RngKey rk = rng_resolve(A.seed, A.offset, A.rng_dev);
  auto tile_request = [&](long long tile) __attribute__((always_inline)) -> float {
    const long long r0 = tile * 16;
    float v = 0.f;
    if (tile < n_tiles) {
      if (has_elem) {
        const long long nvalid = (M - r0 < 16 ? M - r0 : 16) * D;
        if (tid < nvalid) v = A.data[r0 * D + tid];
      } else if (is_adv || is_vs) {
        const long long i = r0 + (tid & 15);
        if (i < M) v = is_adv ? A.adv[i] : A.vs[i];
      }
    }
    return v;
  };
  auto draw_noise = [&](long long tile, int par) __attribute__((always_inline)) {
    if (lane < 16 && tile < n_tiles) {
      const long long i = tile * 16 + lane;
      float e = 0.f;
      if (i < M) e = A.ent_noise ? A.ent_noise[i] : philox_normal(rk.seed, rk.offset, MBPO_STREAM_ENTROPY, (unsigned long long)i);
      smem[P_EPS + 16 * par + lane] = e;
    }
  };
  float v_next = tile_request(blockIdx.x);      // FIRST in the vector-memory queue (results return in order), before ~70 weight requests
  if (wave == 7) draw_noise(blockIdx.x, 0);     // (and the first tile's noise before this wave's requests: the first barrier waits for the last wave)

  // ---- every weight image this wave will use, once per launch ----
  float tw[X + 1];
  ImgF I1, I2;
  float wo[16];
  float bo0 = 0.f, bo1 = 0.f;
  float two[2] = {0.f, 0.f};
  float G2[16], G1[16];
  thin_col_request<X>(tw, net_p, lane);
  img_fwd_request(I1, net_p + N::W1, c0, lane);
  if constexpr (NH == 2) img_fwd_request(I2, net_p + N::W1 + HID, c0, lane);
  const bool out_wave = sub == c;                                 // wave 0 (policy) and wave 5 (value): different SIMDs
  if (out_wave) {
    if (c == 0) {
      img_out_request<2>(wo, net_p + N::OUT, lane);
      bo0 = net_p[N::OUT + LH * 2];
      bo1 = net_p[N::OUT + LH * 2 + 1];
    } else {
      img_out_request<1>(wo, net_p + N::OUT, lane);
      bo0 = net_p[N::OUT + LH];
    }
  }
  if (c == 0) {
    two[0] = net_p[N::OUT + lane * 2];
    two[1] = net_p[N::OUT + lane * 2 + 1];
  } else {
    two[0] = net_p[N::OUT + lane];
  }
  if constexpr (NH == 2) img_dgrad_request(G2, net_p + N::W1 + HID, c0, lane);
  img_dgrad_request(G1, net_p + N::W1, c0, lane);

  // ---- running sums over this workgroup's tiles (registers) ----
  f32x4 S2[4], S2b, S1[4], S1b;
#pragma unroll
  for (int a = 0; a < 4; ++a) S2[a] = S1[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  S2b = S1b = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s_first0 = 0.f, s_first1 = 0.f;         // layer 0: jobs `sub` and `sub + 4` of X rows + the bias (column = lane)
  float s_last = 0.f, s_lastb = 0.f;            // output layer: dWout[lane][sub] (sub < outputs), db[lane] (wave sub 0)
  float loss_a = 0.f, loss_b = 0.f;             // out waves, lanes 0..15: policy: surrogate, entropy; value: squared error
  bool first = true;

  int par = 0;
#pragma nounroll
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
    const long long r0 = tile * 16;
    float *const s_x = smem + P_X + 128 * par, *const s_aux = smem + P_AUX + 64 * par, *const s_dy = smem + P_DY + 64 * c;
    // ---- the tile's rows (requested one tile ago) to LDS; the next tile's are requested now ----
    {
      const float v = v_next;
      if (is_obs) s_x[r_t * LDX + cc_t] = A.norm_mean ? (v - mu_t) / sd_t : v;
      else if (has_elem && cc_t == 2 * X + 4) s_aux[r_t * 4 + 0] = v;              // raw_action
      else if (has_elem && cc_t == 2 * X + 3) s_aux[r_t * 4 + 1] = v;              // behaviour log-prob (policy_extras.log_prob)
      else if (is_adv) s_aux[(tid & 15) * 4 + 2] = v;
      else if (is_vs) s_aux[(tid & 15) * 4 + 3] = v;
      v_next = tile_request(tile + gridDim.x);
    }
    PPO_STAMP(1);
    __syncthreads();
    PPO_STAMP(2);
    // ---- forward: policy logits (:80) and value baseline (:82); swish'(z) stored for the backward pass ----
    thin_first<X, true, false>(tw, s_x, PT(3), PT(0), nullptr, sub, lane);
    __syncthreads();
    PPO_STAMP(3);
    hid_fwd<true>(I1, PT(3), PT(4), PT(1), c0, lane);
    __syncthreads();
    PPO_STAMP(4);
    if constexpr (NH == 2) {
      hid_fwd<true>(I2, PT(4), PT(5), PT(2), c0, lane);
      __syncthreads();
    }
    PPO_STAMP(5);
    // ---- output layers and the loss terms on the waves that hold them (:84-126) ----
    if (out_wave) {
      const f32x4 y = out_fwd(wo, PT(3 + HL), lane);
      const int r = lane & 15;
      const bool ok = r0 + r < M;
      if (c == 0) {
        const float loc = y[0] + bo0, raw = y[1] + bo1;
        const float z = s_aux[r * 4 + 0], lp_b = s_aux[r * 4 + 1];
        const float eps = smem[P_EPS + 16 * par + r];
        const float adv = ok ? (s_aux[r * 4 + 2] - adv_mean) * adv_istd : 0.f;
        const float sg = pl_fsoftplus(raw) + 0.001f;
        const float q = (z - loc) / sg;
        const float lsg = pl_flog(sg);
        const float lpt = -0.5f * q * q - lsg - LOG_SQRT_2PI - 2.0f * (LOG_2 - z - pl_fsoftplus(-2.0f * z));      // log_prob (:91-92)
        const float zf = loc + sg * eps;
        const float ent_d = 0.5f + LOG_SQRT_2PI + lsg + 2.0f * (LOG_2 - zf - pl_fsoftplus(-2.0f * zf));          // entropy (:117)
        float lp_t = 0.f, ent = 0.f;
        lp_t += lpt;
        ent += ent_d;
        const float rho = pl_fexp(lp_t - lp_b);                                                                   // :103
        const float lo = 1.f - A.clip_eps, hi = 1.f + A.clip_eps;
        const float s1 = rho * adv, s2 = fminf(fmaxf(rho, lo), hi) * adv;
        const bool inside = (rho >= lo) && (rho <= hi);
        const float w = inside ? 1.f : (s1 < s2 ? 1.f : 0.f);
        const float g_lp = ok ? -invM * rho * adv * w : 0.f;
        const float g_ent = ok ? -A.entropy_cost * invM : 0.f;
        const float th = pl_ftanh(loc + sg * eps);
        const float g_loc = g_lp * (q / sg) + g_ent * (-2.f * th);
        const float g_sig = g_lp * ((q * q - 1.f) / sg) + g_ent * (1.f / sg - 2.f * th * eps);
        if (lane < 16) {
          if (ok) {
            loss_a += -fminf(s1, s2);
            loss_b += ent;
          }
          s_dy[r * 4 + 0] = g_loc;
          s_dy[r * 4 + 1] = g_sig * fast_sigmoid(raw);
        }
      } else {
        const float v = y[0] + bo0;
        const float vs = ok ? s_aux[r * 4 + 3] : 0.f;
        if (lane < 16) {
          if (ok) loss_a += 0.5f * (vs - v) * (vs - v);
          s_dy[r * 4 + 0] = ok ? -(vs - v) * invM : 0.f;            // d (0.5 * mean((vs - V)^2)) / dV   (:112-114)
        }
      }
    } else if (wave == 7) {
      draw_noise(tile + gridDim.x, par ^ 1);        // the next tile's entropy noise, on a wave that has nothing to do in this step
    }
    PPO_STAMP(6);
    __syncthreads();
    PPO_STAMP(7);
    // ---- backward, input gradients first: delta_2, delta_1, delta_0 overwrite the stored derivatives in place ----
    {
      float zq[4];
      thin_z_preload(zq, PT(HL), sub, lane);
      if (c == 0) thin_dgrad_last<2>(two, s_dy, zq, PT(HL), sub, lane);
      else thin_dgrad_last<1>(*reinterpret_cast<float(*)[1]>(&two[0]), s_dy, zq, PT(HL), sub, lane);
    }
    __syncthreads();
    PPO_STAMP(8);
    if constexpr (NH == 2) {
      hid_dgrad(G2, PT(2), PT(1), PT(1), c0, lane);
      __syncthreads();
    }
    PPO_STAMP(9);
    hid_dgrad(G1, PT(1), PT(0), PT(0), c0, lane);
    __syncthreads();
    PPO_STAMP(10);
    // ---- weight gradients of the four layers: nothing depends on them inside the tile, no barrier between them ----
    {
      f32x4 acc[4], accb;
      if constexpr (NH == 2) {
        hid_wgrad_regs(PT(4), PT(2), c0, lane, acc, accb);          // dW2 = h1^T delta_2
#pragma unroll
        for (int a = 0; a < 4; ++a) S2[a] = first ? acc[a] : S2[a] + acc[a];
        S2b = first ? accb : S2b + accb;
      }
      hid_wgrad_regs(PT(3), PT(1), c0, lane, acc, accb);          // dW1 = h0^T delta_1
#pragma unroll
      for (int a = 0; a < 4; ++a) S1[a] = first ? acc[a] : S1[a] + acc[a];
      S1b = first ? accb : S1b + accb;
    }
    {
      // layer 0: dW0[k][col] = sum_r x[r][k] delta_0[r][col] for k = sub (and k = sub + 4: a row or, at k = X, the bias)
      float dv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) dv[r] = PT(0)[r * LDH + lane];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int k = sub + 4 * jj;
        if (k <= X) {
          float acc = 0.f;
          if (k < X) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = fmaf(s_x[r * LDX + k], dv[r], acc);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc += dv[r];
          }
          if (jj == 0) s_first0 = first ? acc : s_first0 + acc;
          else s_first1 = first ? acc : s_first1 + acc;
        }
      }
      // output layer: dWout[col][o] = sum_r h2[r][col] dY[r][o] on wave sub = o; db[o] on wave sub 0, lane o
      const int NO = c == 0 ? 2 : 1;
      if (sub < NO) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = fmaf(PT(3 + HL)[r * LDH + lane], s_dy[r * 4 + sub], acc);
        s_last = first ? acc : s_last + acc;
      }
      if (sub == 0 && lane < NO) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += s_dy[r * 4 + lane];
        s_lastb = first ? acc : s_lastb + acc;
      }
    }
    PPO_STAMP(11);
    stamp_base += 12;
    first = false;
    // (no barrier here: the next tile's rows go to the other parity of s_x / s_aux, and nobody overwrites this tile's tiles before
    //  the barrier behind that store, which every wave reaches only after its weight gradients)
  }

  // ---- this workgroup's slab [policy | value] and loss partials ----
  float *const slab = A.slabs + (long long)blockIdx.x * (N::P + N::V) + (c ? N::P : 0);
  {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float o2[4] = {S2[a][0], S2[a][1], S2[a][2], S2[a][3]}, o1[4] = {S1[a][0], S1[a][1], S1[a][2], S1[a][3]};
      if constexpr (NH == 2) store_vec_global<4>(slab + N::W1 + HID + (4 * r + a) * LH + c0 + 4 * g, o2);
      store_vec_global<4>(slab + N::W1 + (4 * r + a) * LH + c0 + 4 * g, o1);
    }
    if (r == 0) {
      const float o2[4] = {S2b[0], S2b[1], S2b[2], S2b[3]}, o1[4] = {S1b[0], S1b[1], S1b[2], S1b[3]};
      if constexpr (NH == 2) store_vec_global<4>(slab + N::W1 + HID + LH * LH + c0 + 4 * g, o2);
      store_vec_global<4>(slab + N::W1 + LH * LH + c0 + 4 * g, o1);
    }
    if (sub <= X) slab[sub * LH + lane] = s_first0;                    // rows 0..3 (or the bias when X < 4 and sub == X)
    if (sub + 4 <= X) slab[(sub + 4) * LH + lane] = s_first1;
    const int NO = c == 0 ? 2 : 1;
    if (sub < NO) slab[N::OUT + lane * NO + sub] = s_last;
    if (sub == 0 && lane < NO) slab[N::OUT + LH * NO + lane] = s_lastb;
  }
  __syncthreads();
  if (out_wave && lane < 16) {
    if (c == 0) {
      smem[P_LOSS + lane] = loss_a;
      smem[P_LOSS + 32 + lane] = loss_b;
    } else {
      smem[P_LOSS + 16 + lane] = loss_a;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f, e = 0.f;
    for (int i = 0; i < 16; ++i) {
      a += smem[P_LOSS + i];
      b += smem[P_LOSS + 16 + i];
      e += smem[P_LOSS + 32 + i];
    }
    A.extras[blockIdx.x * 4 + 0] = a;
    A.extras[blockIdx.x * 4 + 1] = b;
    A.extras[blockIdx.x * 4 + 2] = e;
  }
}

// ------------------------------------------------------------------------------------------------
// k_ppo_vg_lean<X>: the values pre-pass (ppo/losses.py:84-86), compute_gae (:128-184) and the advantage moments' per-workgroup partials
// in one launch — ppo.hip's k_ppo_values_gae with the value network's images resident in registers: a workgroup owns G whole
// trajectories (T samples + the bootstrap row each), 2 chains x 4 waves walk their row tiles two at a time, one thread per trajectory
// then walks compute_gae backwards in the reference's order, the first wave leaves {n, mean, M2}.  The input layer is formed with the
// generic runner's k groups (wset_fwd_hidden<IN>), so the values — and with them vs, adv and the partials — are the generic launch's bits.
// (Six chains of two waves — all of C3's six tiles per workgroup in one round instead of three — was measured: 77.5 us per minibatch step
// against 71.6; twelve waves requesting the images make the prologue's burst 2.4 x as long, more than the two rounds saved.)
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int VG_X = 0;                 // [2 chains][16][8] input tiles
constexpr int VG_TILES = 256;           // [2 chains][2] hidden tiles
constexpr int VG_ARR = VG_TILES + 4 * LT;     // s_val [G R] | s_tr | s_te | s_rw | s_adv [G T] each (rounded up to 4)
}  // namespace

template <int X>
__global__ void __launch_bounds__(512) k_ppo_vg_lean(const PpoVgLeanArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int U = 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wave >> 2, sub = wave & 3, c0 = sub * 16;
  const int T = A.T, R = T + 1, G = A.G, D = A.D;
  const int i16 = lane & 15, g4 = lane >> 4;
  float *s_val = smem + VG_ARR;
  float *s_tr = s_val + ((G * R + 3) & ~3);
  const int GT4 = (G * T + 3) & ~3;
  float *s_te = s_tr + GT4, *s_rw = s_te + GT4, *s_adv = s_rw + GT4;
  const long long b0 = (long long)blockIdx.x * G;
  const int g_here = (int)((A.B - b0 < G) ? A.B - b0 : G);
  const int rows = g_here * R;
  if (blockIdx.x == 0 && tid == 0) A.step_count_rw[0] = A.step_count_rw[0] + 1.0f;     // (nothing in this launch reads it)
  // ---- the value network's images, once ----
  const float *const net_p = A.v_params;
  constexpr int W1 = X * LH + LH;
  const int nh = A.n_hid, OUT = W1 + nh * HID;
  constexpr int kc = (X + 3) >> 2;      // 1 for x in {3, 4}
  float w0[2], b0v[4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int k = g4 * kc + s;
    const bool ok = (s < kc) && (k < X);
    const float v = net_p[(ok ? k : 0) * LH + c0 + i16];
    w0[s] = ok ? v : 0.f;
  }
  {
    const f4u t = *reinterpret_cast<const f4u *>(net_p + X * LH + c0 + 4 * g4);
    b0v[0] = t[0]; b0v[1] = t[1]; b0v[2] = t[2]; b0v[3] = t[3];
  }
  ImgF I1, I2;
  img_fwd_request(I1, net_p + W1, c0, lane);
  img_fwd_request(I2, net_p + W1 + (nh - 1) * HID, c0, lane);      // (one 64 x 64 layer: requested again, never used)
  float wo[16];
  float bo = 0.f;
  const bool out_wave = sub == c;                                      // waves 0 and 5: different SIMDs
  if (out_wave) {
    img_out_request<1>(wo, net_p + OUT, lane);
    bo = net_p[OUT + LH];
  }
  float *const s_x = smem + VG_X + c * 128;
  float *const tiles = smem + VG_TILES + c * 2 * LT;
  const int ct = tid & 255;
  float nm = 0.f, ns = 1.f;
  const int col_t = ct % X, row_t = ct / X;
  if (ct < 16 * X && A.norm_mean) {
    nm = A.norm_mean[col_t];
    ns = A.norm_std[col_t];
  }
#pragma nounroll
  for (int rr = 0; rr < rows; rr += 32) {
    const int r0 = rr + 16 * c;
    const bool live = r0 < rows;             // (wave-uniform: a chain without a tile only keeps the barriers)
    if (live) {
      if (ct < 16 * X) {
        const int lrow = r0 + row_t;
        float o = 0.f;
        if (lrow < rows) {
          const int g = lrow / R, t = lrow - g * R;
          const long long base = ((b0 + g) * T + (t < T ? t : T - 1)) * D;
          o = A.data[base + (t < T ? col_t : X + U + 2 + col_t)];                  // observation | next_observation[-1]  (losses.py:84-85)
          if (A.norm_mean) o = (o - nm) / ns;
        }
        s_x[row_t * LDX + col_t] = o;
      } else if (ct >= 192 && ct < 208) {
        const int lrow = r0 + (ct - 192);
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
    if (live) {
      // input layer on the generic runner's k groups: lane group g takes inputs g * kc .. (one MFMA, the second multiplies zeros)
      float xs[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int k = g4 * kc + s;
        const bool ok = (s < kc) && (k < X);
        const float v = s_x[i16 * LDX + (ok ? k : 0)];
        xs[s] = ok ? v : 0.f;
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = MFMA(w0[0], xs[0], acc);
      acc = MFMA(w0[1], xs[1], acc);
      float zv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) zv[q] = acc[q] + b0v[q];
      act_apply_vec<4>(zv, MBPO_ACT_SWISH);
      store_vec_lds<4>(tiles + i16 * LDH + c0 + 4 * g4, zv);
    }
    __syncthreads();
    if (live) hid_fwd<false>(I1, tiles, tiles + LT, nullptr, c0, lane);
    __syncthreads();
    if (nh == 2) {
      if (live) hid_fwd<false>(I2, tiles + LT, tiles, nullptr, c0, lane);
      __syncthreads();
    }
    if (live && out_wave) {
      const f32x4 y = out_fwd(wo, nh == 2 ? tiles : tiles + LT, lane);
      if (lane < 16 && r0 + lane < rows) s_val[r0 + lane] = y[0] + bo;
    }
    // (no barrier: the next round's input tile and the first hidden tile were last read two barriers ago)
  }
  __syncthreads();
  // compute_gae, one thread per trajectory, backwards (losses.py:150-184)
  if (tid < g_here) {
    const int g = tid;
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
    const int n = g_here * T;
    float a = 0.f;
    for (int i = lane; i < n; i += 64) a += s_adv[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    const float mean = __shfl(a, 0, 64) / (float)n;
    float q = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float dd = s_adv[i] - mean;
      q += dd * dd;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_down(q, o, 64);
    if (lane == 0) {
      A.mom_part[4 * blockIdx.x + 0] = (float)n;
      A.mom_part[4 * blockIdx.x + 1] = mean;
      A.mom_part[4 * blockIdx.x + 2] = q;
    }
  }
}

int ppo_vg_lean_launch(const PpoVgLeanArgs &A, int x_dim, int n_wgs, size_t arr_floats, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(VG_ARR + arr_floats) * sizeof(float);
  int rc;
#define VG_X_(X_)                                                                       \
  if (x_dim == X_) {                                                                    \
    rc = mbpo_ensure_lds<k_ppo_vg_lean<X_>>(lds, "ppo_vg_lean");                        \
    if (rc != MBPO_OK) return rc;                                                       \
    hipLaunchKernelGGL(k_ppo_vg_lean<X_>, dim3(n_wgs), dim3(512), lds, st, A);          \
    return MBPO_OK;                                                                     \
  }
  VG_X_(2) VG_X_(3) VG_X_(4) VG_X_(5) VG_X_(6)
#undef VG_X_
  {
    mbpo_set_error("ppo_vg_lean: x_dim %d has no instantiation", x_dim);
    return MBPO_ERR_UNSUPPORTED;
  }
  return MBPO_OK;
}

bool ppo_vg_lean_supports(int x_dim, const int *value_dims, int value_layers, int value_act) {
  if (x_dim < 2 || x_dim > 6 || (value_layers != 4 && value_layers != 3) || value_act != MBPO_ACT_SWISH) return false;
  for (int l = 1; l < value_layers; ++l)
    if (value_dims[l] != LH) return false;
  return value_dims[0] == x_dim && value_dims[value_layers] == 1;
}

bool ppo_lean_supports(int x_dim, int u_dim, const int *policy_dims, int policy_layers, int policy_act, const int *value_dims, int value_layers,
                       int value_act) {
  if (u_dim != 1 || x_dim < 2 || x_dim > 6) return false;
  if ((policy_layers != 4 && policy_layers != 3) || value_layers != policy_layers || policy_act != MBPO_ACT_SWISH || value_act != MBPO_ACT_SWISH) return false;
  for (int l = 1; l < policy_layers; ++l)
    if (policy_dims[l] != LH || value_dims[l] != LH) return false;
  return policy_dims[0] == x_dim && policy_dims[policy_layers] == 2 && value_dims[0] == x_dim && value_dims[value_layers] == 1;
}

int ppo_lean_launch(const PpoLeanArgs &A, int x_dim, int n_hid, int n_wgs, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc;
#define PL_X_(X_)                                                                                        \
  if (x_dim == X_ && n_hid == 2) {                                                                       \
    rc = mbpo_ensure_lds<k_ppo_lean<X_, 2>>(PPO_LEAN_LDS_BYTES, "ppo_lean");                             \
    if (rc != MBPO_OK) return rc;                                                                        \
    hipLaunchKernelGGL((k_ppo_lean<X_, 2>), dim3(n_wgs), dim3(512), PPO_LEAN_LDS_BYTES, st, A);          \
    return MBPO_OK;                                                                                      \
  }                                                                                                      \
  if (x_dim == X_ && n_hid == 1) {                                                                       \
    rc = mbpo_ensure_lds<k_ppo_lean<X_, 1>>(PPO_LEAN_LDS_BYTES, "ppo_lean");                             \
    if (rc != MBPO_OK) return rc;                                                                        \
    hipLaunchKernelGGL((k_ppo_lean<X_, 1>), dim3(n_wgs), dim3(512), PPO_LEAN_LDS_BYTES, st, A);          \
    return MBPO_OK;                                                                                      \
  }
  PL_X_(2) PL_X_(3) PL_X_(4) PL_X_(5) PL_X_(6)
#undef PL_X_
  {
    mbpo_set_error("ppo_lean: x_dim %d has no instantiation", x_dim);
    return MBPO_ERR_UNSUPPORTED;
  }
  return MBPO_OK;
}

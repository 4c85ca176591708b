// lean_blocks.hpp — building blocks of the kernels specialised for the benchmark networks (64-wide hidden layers, swish, network
// inputs <= 8 wide, outputs <= 2): csrc/sac_lean.hip (k_sac_lean, the SAC forward/backward launch) and csrc/ppo_lean.hip (k_ppo_lean,
// the PPO loss forward/backward launch).  Every dot product is formed by the same v_mfma_f32_16x16x4_f32 sequence over the same k
// groups in the same order as the generic runners of chain_run.hpp — but with the MFMA operands SWAPPED (A = weights, B =
// activations: D^T comes out, a*b = b*a), so a lane holds FOUR CONSECUTIVE COLUMNS of ONE row: every activation / delta store is
// one ds_write_b128, every stored-derivative read one ds_read_b128, every weight-gradient row piece 16 bytes wide.  (The same swap
// inside chain_run.hpp was measured in round 4 and reverted: bit-identical, but the generic kernels sit at their register limits
// and the four bias registers per n-tile spilled — rollout 78 -> 91 us, PPO minibatch 103 -> 132 us.)
#pragma once
#include "common.hpp"
#include "chain_run.hpp"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int LH = 64;            // hidden width
constexpr int LDH = 68;           // row stride of a hidden tile (floats): rows 16-byte aligned, bank offset 4 per row
constexpr int LT = 16 * LDH;      // one hidden tile
constexpr int LDX = 8;            // row stride of an input tile ([obs | action], x + u <= 8)
constexpr int HID = LH * LH + LH; // one hidden layer's parameters (W + b)

// ---- register images ----------------------------------------------------------------------------------------------
// forward, hidden layer (W [64][64] then bias [64]), this wave's 16 columns c0..c0+15:
//   w[s] = W[16 g + s][c0 + i]   (A operand: matrix row i = column c0 + i, k = g -> input 16 g + s)
//   b[i'] = bias[c0 + 4 g + i']  (the lane's results are y[row j][c0 + 4 g + i'], j = lane & 15)
struct ImgF {
  float w[16];
  float b[4];
};
__device__ __forceinline__ void img_fwd_request(ImgF &I, const float *__restrict__ W, int c0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const float *p = W + (16 * g) * LH + c0 + i;
#pragma unroll
  for (int s = 0; s < 16; ++s) I.w[s] = p[s * LH];
  const f4u t = *reinterpret_cast<const f4u *>(W + LH * LH + c0 + 4 * g);
  I.b[0] = t[0]; I.b[1] = t[1]; I.b[2] = t[2]; I.b[3] = t[3];
}
// forward, output layer (W [64][N], N <= 2): matrix row i stands for column i & 3, so every lane group ends up with the outputs of
// its row: w[s] = W[16 g + s][(i & 3) < N ? i & 3 : 0]
template <int N>
__device__ __forceinline__ void img_out_request(float (&w)[16], const float *__restrict__ W, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int col = ((i & 3) < N) ? (i & 3) : 0;
  const float *p = W + (16 * g) * N + col;
#pragma unroll
  for (int s = 0; s < 16; ++s) w[s] = p[s * N];
}
// input-gradient of a hidden layer: w[n] = W[k0 + i][16 g + n]  (A operand: matrix row i = input k0 + i, k = g -> output 16 g + n)
__device__ __forceinline__ void img_dgrad_request(float (&w)[16], const float *__restrict__ W, int k0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const float *p = W + (k0 + i) * LH + 16 * g;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4u t = *reinterpret_cast<const f4u *>(p + 4 * q);
    w[4 * q] = t[0]; w[4 * q + 1] = t[1]; w[4 * q + 2] = t[2]; w[4 * q + 3] = t[3];
  }
}
// thin first layer: this lane's column of W0 [K][64] and its bias
template <int K>
__device__ __forceinline__ void thin_col_request(float (&tw)[K + 1], const float *__restrict__ W0, int lane) {
#pragma unroll
  for (int k = 0; k < K; ++k) tw[k] = W0[k * LH + lane];
  tw[K] = W0[K * LH + lane];
}

// the B operand of a hidden layer: 16 consecutive activations of row j from k group g (four ds_read_b128)
__device__ __forceinline__ void read_row16(float (&av)[16], const float *tile, int lane) {
  const int j = lane & 15, g = lane >> 4;
  const float *xr = tile + j * LDH + 16 * g;
#pragma unroll
  for (int q = 0; q < 4; ++q) load_vec_lds<4>(xr + 4 * q, *reinterpret_cast<float(*)[4]>(&av[4 * q]));
}

// ---- layer steps --------------------------------------------------------------------------------------------------
// Requests that ride in a layer step's MFMA shadow: a wave issues in order and every MFMA of a chain waits ~40 cycles for its
// predecessor, so vector-memory requests placed BETWEEN the MFMAs cost nothing, while the same requests in front of the step cost
// their issue time on every wave of the workgroup at once (the CU's address path takes one wave-instruction per ~6 cycles: 8 waves x
// 40 requests in one burst were 2.5 k cycles of a 25 k-cycle kernel).  `pf(s)` issues the s-th piece of the riding request; a
// schedule fence that only ALU and LDS instructions may cross pins "MFMA s, then request s" (hipcc otherwise gathers the requests in
// front of the first MFMA; sched_group_barrier pipelines were ignored here).
struct NoSt {
  __device__ __forceinline__ void operator()(int) const {}
};
struct NoPf {
  static constexpr bool active = false;
  __device__ __forceinline__ void operator()(int) const {}
};
template <class F>
struct Pf {
  static constexpr bool active = true;
  F f;
  __device__ __forceinline__ void operator()(int s) const { f(s); }
};
template <class F>
__device__ __forceinline__ Pf<F> make_pf(F f) {
  return Pf<F>{f};
}
#define PIN_ORDER() __builtin_amdgcn_sched_barrier(0x0086)      /* VALU, SALU and DS may cross; MFMA and VMEM may not */

// piece s (0..15) of a forward hidden image request: one weight each, the bias vector with the last
__device__ __forceinline__ void img_fwd_request_piece(ImgF &I, const float *__restrict__ W, int c0, int lane, int s) {
  const int i = lane & 15, g = lane >> 4;
  I.w[s] = W[(16 * g + s) * LH + c0 + i];
  if (s == 15) {
    const f4u t = *reinterpret_cast<const f4u *>(W + LH * LH + c0 + 4 * g);
    I.b[0] = t[0]; I.b[1] = t[1]; I.b[2] = t[2]; I.b[3] = t[3];
  }
}

// h = swish(z) (in place) and d = swish'(z) from ONE sigmoid: the expressions of act_apply_vec / act_grad_mul_vec (common.hpp) on the
// same z, so h and, later, delta * d are the bits the generic kernel forms — but the backward pass, whose epilogues sit on the
// critical path, multiplies by a stored factor instead of evaluating 2 transcendentals per element again.
__device__ __forceinline__ void swish_and_grad4(float (&z)[4], float (&d)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float sg = fast_sigmoid(z[i]);
    d[i] = sg * (1.0f + z[i] * (1.0f - sg));
    z[i] = z[i] * sg;
  }
}

// hidden layer forward: h_out[row j][c0 + 4 g ..] = swish(x W + b); STORE_Z: swish'(pre-activation) to z_out for the backward pass
template <bool STORE_Z, class PF = NoPf>
__device__ __forceinline__ void hid_fwd(const ImgF &I, const float *xin, float *h_out, float *z_out, int c0, int lane, PF pf = PF()) {
  const int j = lane & 15, g = lane >> 4;
  float av[16];
  read_row16(av, xin, lane);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    acc = MFMA(I.w[s], av[s], acc);
    if (PF::active) {
      pf(s);
      PIN_ORDER();
    }
  }
  float zv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) zv[i] = acc[i] + I.b[i];
  const int o = j * LDH + c0 + 4 * g;
  if (STORE_Z) {
    float dv[4];
    swish_and_grad4(zv, dv);
    store_vec_lds<4>(z_out + o, dv);
  } else {
    act_apply_vec<4>(zv, MBPO_ACT_SWISH);
  }
  store_vec_lds<4>(h_out + o, zv);
}
// the same with the tangent tile riding along (chain_run.hpp JVP): t_out = swish'(z) * (t_in W)
__device__ __forceinline__ void hid_fwd_jvp(const ImgF &I, const float *xin, const float *tin, float *h_out, float *t_out, int c0, int lane) {
  const int j = lane & 15, g = lane >> 4;
  float av[16], tv_in[16];
  read_row16(av, xin, lane);
  read_row16(tv_in, tin, lane);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, tacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    acc = MFMA(I.w[s], av[s], acc);
    tacc = MFMA(I.w[s], tv_in[s], tacc);
  }
  float zv[4], tv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    zv[i] = acc[i] + I.b[i];
    tv[i] = tacc[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float sg = fast_sigmoid(zv[i]);
    tv[i] *= sg * (1.0f + zv[i] * (1.0f - sg));
    zv[i] = zv[i] * sg;
  }
  const int o = j * LDH + c0 + 4 * g;
  store_vec_lds<4>(t_out + o, tv);
  store_vec_lds<4>(h_out + o, zv);
}
// output layer forward: every lane ends up with y[row j][col i] in acc[i] (bias not added)
__device__ __forceinline__ f32x4 out_fwd(const float (&w)[16], const float *xin, int lane) {
  float av[16];
  read_row16(av, xin, lane);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = MFMA(w[s], av[s], acc);
  return acc;
}
// value and tangent of an output layer side by side (two independent accumulation chains: 32-cycle issue instead of 40)
__device__ __forceinline__ void out_fwd2(const float (&w)[16], const float *xin, const float *tin, int lane, f32x4 &y, f32x4 &ty) {
  float av[16], tv[16];
  read_row16(av, xin, lane);
  read_row16(tv, tin, lane);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, tacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    acc = MFMA(w[s], av[s], acc);
    tacc = MFMA(w[s], tv[s], tacc);
  }
  y = acc;
  ty = tacc;
}
// hidden layer input-gradient: d_out[row j][k0 + 4 g ..] = (delta W^T) * swish'(z_prev); `zprev` holds swish'(z_prev) itself
template <class ST = NoSt>
__device__ __forceinline__ void hid_dgrad(const float (&w)[16], const float *din, const float *zprev, float *d_out, int k0, int lane, ST st = ST()) {
  const int j = lane & 15, g = lane >> 4;
  float av[16];
  st(0);
  read_row16(av, din, lane);
  const int o = j * LDH + k0 + 4 * g;
  float zv[4];
  load_vec_lds<4>(zprev + o, zv);
  st(1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < 16; ++n) acc = MFMA(w[n], av[n], acc);
  float ov[4];
  st(2);
#pragma unroll
  for (int i = 0; i < 4; ++i) ov[i] = acc[i] * zv[i];      // zv = swish'(z_prev), stored by the forward pass
  store_vec_lds<4>(d_out + o, ov);
  st(3);
}
// hidden layer weight gradient (chain_run.hpp wgrad_tile_fast<4, 1, true>, operands swapped): columns c0..c0+15 of dW [64][64] and
// of db.  acc[a] lane (j, g) reg i = dW[4 j + a][c0 + 4 g + i]; the bias tile's B operand is the indicator of column 0.
// Results go to an LDS stage, not to the slab: a 1-KB global store costs the issuing wave ~200 cycles (stores are issue-bound,
// ~20 B/clk per CU), and the wave that forms the gradients must reach the layer's barrier; copy_stage_out moves them later.
template <class ST = NoSt>
__device__ __forceinline__ void hid_wgrad(const float *hin, const float *delta, float *gst, int c0, int lane, ST st = ST()) {
  const int r = lane & 15, g = lane >> 4;
  st(0);
  f32x4 acc[4], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float hv[4][4], dv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = 4 * g + s;
    load_vec_lds<4>(hin + row * LDH + 4 * r, hv[s]);
    dv[s] = delta[row * LDH + c0 + r];
  }
  const float one0 = (r == 0) ? 1.f : 0.f;
  st(1);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = MFMA(dv[s], hv[s][a], acc[a]);
    accb = MFMA(dv[s], one0, accb);
  }
  st(2);
  // to the LDS stage: row k = 4 r + a, 16-byte chunk (c0 / 4 + g) rotated by r — eight lanes of a ds_write_b128 group then hit eight
  // different bank groups (rows are 256 bytes apart: unrotated they would all hit one)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float ov[4] = {acc[a][0], acc[a][1], acc[a][2], acc[a][3]};
    store_vec_lds<4>(gst + (4 * r + a) * LH + 4 * (((c0 >> 2) + g + r) & 15), ov);
  }
  if (r == 0) {
    const float ov[4] = {accb[0], accb[1], accb[2], accb[3]};
    store_vec_lds<4>(gst + LH * LH + c0 + 4 * g, ov);
  }
  st(3);
}

// LDS stage -> slab: participant `t` of `n_part` threads moves 16-byte chunks t, t + n_part, ... of one hidden layer's block
// (1024 rotated chunks of dW, 16 of db) with ds_read_b128 + global_store_dwordx4.
__device__ __forceinline__ void copy_stage_out(const float *gst, float *__restrict__ gW, int t, int n_part, int q_begin = 0, int q_end = 1040) {
  for (int q = q_begin + t; q < q_end; q += n_part) {
    float v[4];
    load_vec_lds<4>(gst + 4 * q, v);
    int dst = 4 * q;
    if (q < 1024) {
      const int k = q >> 4, cc = q & 15;
      dst = k * LH + 4 * ((cc - (k >> 2)) & 15);
    }
    store_vec_global<4>(gW + dst, v);
  }
}

// Sum over the 16 lanes of a row, result on the row's lane 0: v_l += v_{l+8}, += v_{l+4}, += v_{l+2}, += v_{l+1} on DPP row shifts.
// For values that are zero outside lanes 0..15 this is wave_sum64's tree (sac_shared.hpp: shuffles by 32, 16, 8, 4, 2, 1 — the
// first two steps add zeros) with the same pairs in the same order — the same bits — in 4 VALU instructions instead of 6
// ds_bpermute round trips on the one wave every other wave waits for.
template <int CTRL>
__device__ __forceinline__ float dpp_shl(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_shl<0x108>(v);
  v += dpp_shl<0x104>(v);
  v += dpp_shl<0x102>(v);
  v += dpp_shl<0x101>(v);
  return v;
}

// ---- thin layers (chain_run.hpp "thin layers by VALU", shapes as constants) -----------------------------------------
// layer 0 of a forward chain: rows 4 sub .. 4 sub + 3, column = lane
template <int K, bool STORE_Z, bool TANGENT>
__device__ __forceinline__ void thin_first(const float (&tw)[K + 1], const float *x, float *h0, float *z0, float *t0, int sub, int lane) {
  float xv[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    load_vec_lds<4>(x + (4 * sub + i) * LDX, *reinterpret_cast<float(*)[4]>(&xv[i][0]));
    if (K > 4) load_vec_lds<4>(x + (4 * sub + i) * LDX + 4, *reinterpret_cast<float(*)[4]>(&xv[i][4]));
  }
  float zv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float z = tw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) z = fmaf(xv[i][k], tw[k], z);
    zv[i] = z;
  }
  if (TANGENT) {      // d h0 / d x[K - 1] = swish'(z0) * W0[K - 1][col]
    float tv[4] = {tw[K - 1], tw[K - 1], tw[K - 1], tw[K - 1]};
    act_grad_mul_vec<4>(tv, zv, MBPO_ACT_SWISH);
#pragma unroll
    for (int i = 0; i < 4; ++i) t0[(4 * sub + i) * LDH + lane] = tv[i];
  }
  if (STORE_Z) {      // swish'(z0) for the backward pass
    float dv[4];
    swish_and_grad4(zv, dv);
#pragma unroll
    for (int i = 0; i < 4; ++i) z0[(4 * sub + i) * LDH + lane] = dv[i];
  } else {
    act_apply_vec<4>(zv, MBPO_ACT_SWISH);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) h0[(4 * sub + i) * LDH + lane] = zv[i];
}
// delta_2 = (dY Wout^T) * swish'(z_2), rows 4 sub .. 4 sub + 3, column = lane; two[o] = Wout[lane][o]; zv = this lane's z_2 of those
// rows, read before the section that produces dY
template <int N>
__device__ __forceinline__ void thin_dgrad_last(const float (&two)[N], const float *dY, const float (&zv)[4], float *d_out, int sub, int lane) {
  float dv[4][N], sv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int o = 0; o < N; ++o) dv[i][o] = dY[(4 * sub + i) * 4 + o];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < N; ++o) s = fmaf(dv[i][o], two[o], s);
    sv[i] = s * zv[i];      // zv = swish'(z_2), stored by the forward pass
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) d_out[(4 * sub + i) * LDH + lane] = sv[i];
}
__device__ __forceinline__ void thin_z_preload(float (&zv)[4], const float *z2, int sub, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) zv[i] = z2[(4 * sub + i) * LDH + lane];
}
// output layer's weight gradient: dWout[c][o] = sum_r h2[r][c] dY[r][o] (wave `sub` takes o = sub), db[o] (wave 0); hc = this lane's
// column of h2, read before the section that produces dY
template <int N>
__device__ __forceinline__ void thin_wgrad_last(const float (&hc)[16], const float *dY, float *__restrict__ gW, int sub, int lane) {
  if (sub < N) {
    float dv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[r] = dY[r * 4 + sub];
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = fmaf(hc[r], dv[r], acc);
    gW[lane * N + sub] = acc;
  }
  if (sub == 0 && lane < N) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += dY[r * 4 + lane];
    gW[LH * N + lane] = acc;
  }
}
__device__ __forceinline__ void thin_col_preload(float (&hc)[16], const float *h2, int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) hc[r] = h2[r * LDH + lane];
}
// layer 0's weight gradient: wave w8 takes input row k = w8 (< K), wave K the bias; column = lane
template <int K>
__device__ __forceinline__ void thin_wgrad_first_(const float *x, const float *d0, float *__restrict__ gW, int w8, int lane) {
  if (w8 > K) return;
  float dv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dv[r] = d0[r * LDH + lane];
  if (w8 < K) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = fmaf(x[r * LDX + w8], dv[r], acc);
    gW[w8 * LH + lane] = acc;
  } else {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += dv[r];
    gW[K * LH + lane] = acc;
  }
}

// ---- two 16-column blocks per wave (chains of TWO waves: csrc/rollout_lean.hip, k_ppo_vg_lean) -----------------------------------
// input layer of this wave's two column blocks: w[t][s] = W0[g * kc + s][c0 + 16 t + i] (zero outside the network's K inputs)
struct Img0 {
  float w[2][2];
};

// layer 0: the generic runner's k groups (wset_fwd_hidden<IN>: lane group g takes inputs g * kc .. g * kc + kc - 1; its MFMAs beyond kc
// multiply zeros and are skipped here)
__device__ __forceinline__ void in_fwd2(const Img0 &I, const float *bias, const float *x, int kc, int K, float *h_out, int c0, int lane) {
  const int j = lane & 15, g = lane >> 4;
  float xs[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int k = g * kc + s;
    const bool ok = (s < kc) && (k < K);
    const float v = x[j * LDX + (ok ? k : 0)];
    xs[s] = ok ? v : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = MFMA(I.w[t][0], xs[0], acc);
    acc = MFMA(I.w[t][1], xs[1], acc);
    float zv[4], bv[4];
    load_vec_lds<4>(bias + c0 + 16 * t + 4 * g, bv);
#pragma unroll
    for (int q = 0; q < 4; ++q) zv[q] = acc[q] + bv[q];
    act_apply_vec<4>(zv, MBPO_ACT_SWISH);
    store_vec_lds<4>(h_out + j * LDH + c0 + 16 * t + 4 * g, zv);
  }
}

// hidden layer, two column blocks: one set of activation reads, two interleaved accumulation chains
__device__ __forceinline__ void hid_fwd2(const float (&wa)[16], const float (&wb)[16], const float *bias, const float *xin, float *h_out, int c0,
                                         int lane) {
  const int j = lane & 15, g = lane >> 4;
  float av[16];
  read_row16(av, xin, lane);
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    a = MFMA(wa[s], av[s], a);
    b = MFMA(wb[s], av[s], b);
  }
  float za[4], zb[4], ba[4], bb[4];
  load_vec_lds<4>(bias + c0 + 4 * g, ba);
  load_vec_lds<4>(bias + c0 + 16 + 4 * g, bb);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    za[q] = a[q] + ba[q];
    zb[q] = b[q] + bb[q];
  }
  act_apply_vec<4>(za, MBPO_ACT_SWISH);
  act_apply_vec<4>(zb, MBPO_ACT_SWISH);
  store_vec_lds<4>(h_out + j * LDH + c0 + 4 * g, za);
  store_vec_lds<4>(h_out + j * LDH + c0 + 16 + 4 * g, zb);
}
// forward image of a hidden layer's 16 columns c0..c0+15 without the bias: w[s] = W[16 g + s][c0 + i]
__device__ __forceinline__ void img_w_request(float (&w)[16], const float *__restrict__ W, int c0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const float *p = W + (16 * g) * LH + c0 + i;
#pragma unroll
  for (int s = 0; s < 16; ++s) w[s] = p[s * LH];
}

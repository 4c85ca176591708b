// mlp_tile.hpp — fp32-MFMA dense layers on a 16-row tile held in LDS.
//
// One workgroup owns a tile of 16 rows (envs / minibatch samples).  A Dense layer
// y[16][N] = x[16][K] @ W[K][N] + b is cut into (net, 16-column n-tile) "items"; wave w takes items
// w, w+NW, ... and runs them in groups of up to 8 independent accumulators so the 40-cycle dependent
// latency of v_mfma_f32_16x16x4_f32 (32-cycle issue) is covered by independent chains.
//
// v_mfma_f32_16x16x4_f32 operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l&15][k = l>>4];  B: lane l holds B[k = l>>4][j = l&15];
//   C/D: lane l, reg i holds D[row = 4*(l>>4) + i][col = l&15].
// The k index an MFMA step sums over is arbitrary as long as A and B agree, so lane group g = l>>4
// takes the CONTIGUOUS k range [g*kc, (g+1)*kc), kc = ceil(K/4): A comes from kc consecutive LDS
// floats of the lane's row and k >= K is masked to zero on both operands.
//
// LDS activation tiles are row-major [16][ld] with ld odd (H+1): the A read of lane (row r, group g)
// hits bank (r*ld + g*kc + s) % 32 = (r + const) % 32 — conflict-free for ds_read_b32.
#pragma once
#include "common.hpp"

// Source of the B operand (weights): global memory (L2-resident flat params).
struct DenseIO {
  const float *in;   // LDS, net 0
  int in_net_stride; // floats between nets (0 = all nets read the same input tile)
  int ld_in;
  float *out;        // LDS, net 0
  int out_net_stride;
  int ld_out;
};

template <int NE, int KC_STATIC>
__device__ __forceinline__ void dense_group(const MlpDev &m, int l, int K, int N, int NT, const DenseIO &io,
                                            bool apply_act, int q0, int qstep, int n_items, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const int kc = KC_STATIC > 0 ? KC_STATIC : ((K + 3) >> 2);
  const float *ap[NE];
  const float *wp[NE];
  int col[NE], net[NE];
  bool valid[NE];
  f32x4 acc[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    int q = q0 + j * qstep;
    valid[j] = q < n_items;
    int qq = valid[j] ? q : q0;
    net[j] = qq / NT;
    int nt = qq - net[j] * NT;
    col[j] = nt * 16 + r;
    ap[j] = io.in + net[j] * io.in_net_stride + r * io.ld_in + g * kc;
    // clamp the column so masked lanes still form a legal address
    int ccol = col[j] < N ? col[j] : 0;
    wp[j] = m.params + (long long)net[j] * m.net_stride + m.w_off[l] + (long long)(g * kc) * N + ccol;
    acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (KC_STATIC > 0) {
#pragma unroll
    for (int s = 0; s < (KC_STATIC > 0 ? KC_STATIC : 1); ++s) {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        float a = ap[j][s];
        float b = col[j] < N ? wp[j][(long long)s * N] : 0.f;
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      }
    }
  } else {
    for (int s = 0; s < kc; ++s) {
      const bool kin = (g * kc + s) < K;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        float a = kin ? ap[j][s] : 0.f;
        float b = (kin && col[j] < N) ? wp[j][(long long)s * N] : 0.f;
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    if (valid[j] && col[j] < N) {
      const float bias = m.params[(long long)net[j] * m.net_stride + m.b_off[l] + col[j]];
      float *o = io.out + net[j] * io.out_net_stride + (4 * g) * io.ld_out + col[j];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[j][i] + bias;
        if (apply_act) v = act_apply(v, m.act);
        o[i * io.ld_out] = v;
      }
    }
  }
}

// One Dense layer of `n_nets` nets on the tile.  All waves of the workgroup call this; the caller
// places __syncthreads() between layers.  H_STATIC: compile-time hidden width (K == H_STATIC fast path).
template <int H_STATIC>
__device__ __forceinline__ void dense_layer(const MlpDev &m, int l, int n_nets, const DenseIO &io, bool apply_act,
                                            int wave, int n_waves, int lane) {
  const int K = m.dims[l], N = m.dims[l + 1];
  const int NT = (N + 15) >> 4;
  const int n_items = n_nets * NT;
  int q = wave;
  const bool fastk = (K == H_STATIC);
  // groups of up to 8 independent accumulators per wave
  while (q < n_items) {
    int left = (n_items - q + n_waves - 1) / n_waves;  // items this wave still owns
    if (fastk) {
      if (left >= 8) { dense_group<8, H_STATIC / 4>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += 8 * n_waves; }
      else if (left >= 5) { dense_group<5, H_STATIC / 4>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += 5 * n_waves; }
      else if (left >= 3) { dense_group<3, H_STATIC / 4>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += 3 * n_waves; }
      else if (left >= 2) { dense_group<2, H_STATIC / 4>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += 2 * n_waves; }
      else { dense_group<1, H_STATIC / 4>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += n_waves; }
    } else {
      if (left >= 8) { dense_group<8, 0>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += 8 * n_waves; }
      else if (left >= 5) { dense_group<5, 0>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += 5 * n_waves; }
      else if (left >= 3) { dense_group<3, 0>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += 3 * n_waves; }
      else if (left >= 2) { dense_group<2, 0>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += 2 * n_waves; }
      else { dense_group<1, 0>(m, l, K, N, NT, io, apply_act, q, n_waves, n_items, lane); q += n_waves; }
    }
  }
}

// Full MLP forward for `n_nets` nets on a 16-row tile.
//   x_in  : LDS [16][ld_x] input (dims[0] valid columns), shared by all nets if x_net_stride == 0
//   hA,hB : LDS ping-pong hidden buffers, [n_nets][16][ld_h]   (ld_h = H_STATIC+1)
//   y_out : LDS [n_nets][16][ld_y] output layer (no activation)
// Ends with a __syncthreads(): y_out is readable by every thread on return.
template <int H_STATIC>
__device__ __forceinline__ void mlp_forward_tile(const MlpDev &m, int n_nets, const float *x_in, int x_net_stride,
                                                 int ld_x, float *hA, float *hB, int ld_h, float *y_out, int ld_y,
                                                 int wave, int n_waves, int lane) {
  const int h_net_stride = 16 * ld_h;
  const float *cur = x_in;
  int cur_stride = x_net_stride, cur_ld = ld_x;
  float *nxt = hA;
  for (int l = 0; l < m.n_layers; ++l) {
    const bool last = (l == m.n_layers - 1);
    DenseIO io;
    io.in = cur;
    io.in_net_stride = cur_stride;
    io.ld_in = cur_ld;
    io.out = last ? y_out : nxt;
    io.out_net_stride = last ? 16 * ld_y : h_net_stride;
    io.ld_out = last ? ld_y : ld_h;
    dense_layer<H_STATIC>(m, l, n_nets, io, !last, wave, n_waves, lane);
    __syncthreads();
    cur = nxt;
    cur_stride = h_net_stride;
    cur_ld = ld_h;
    nxt = (nxt == hA) ? hB : hA;
  }
}

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
  float *zout;       // optional LDS copy of the PRE-activation (same strides as out); needed by backward
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
      const int ooff = net[j] * io.out_net_stride + (4 * g) * io.ld_out + col[j];
      float *o = io.out + ooff;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[j][i] + bias;
        if (io.zout) io.zout[ooff + i * io.ld_out] = v;
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
    io.zout = nullptr;
    dense_layer<H_STATIC>(m, l, n_nets, io, !last, wave, n_waves, lane);
    __syncthreads();
    cur = nxt;
    cur_stride = h_net_stride;
    cur_ld = ld_h;
    nxt = (nxt == hA) ? hB : hA;
  }
}

// =================================================================================================
// Backward building blocks (SAC/PPO/BPTT updates).  Same 16-row tile, same MFMA operand maps.
// =================================================================================================

// Forward that keeps what backward needs: for every hidden layer l the pre-activation z_l in zbuf[l][net][16][ld_h]
// and the activation h_l in hbuf[l][net][16][ld_h] — or, with h_pingpong (input-gradient-only backward, which never
// reads h), in hbuf[l & 1][...].  Output layer -> y_out [net][16][ld_y].
template <int H_STATIC>
__device__ __forceinline__ void mlp_forward_tile_store(const MlpDev &m, int n_nets, const float *x_in, int x_net_stride,
                                                       int ld_x, float *zbuf, float *hbuf, bool h_pingpong, int ld_h,
                                                       float *y_out, int ld_y, int wave, int n_waves, int lane) {
  const int h_net_stride = 16 * ld_h;
  const int layer_stride = n_nets * h_net_stride;
  const float *cur = x_in;
  int cur_stride = x_net_stride, cur_ld = ld_x;
  for (int l = 0; l < m.n_layers; ++l) {
    const bool last = (l == m.n_layers - 1);
    float *hl = hbuf + (h_pingpong ? (l & 1) : l) * layer_stride;
    DenseIO io;
    io.in = cur;
    io.in_net_stride = cur_stride;
    io.ld_in = cur_ld;
    io.out = last ? y_out : hl;
    io.out_net_stride = last ? 16 * ld_y : h_net_stride;
    io.ld_out = last ? ld_y : ld_h;
    io.zout = last ? nullptr : zbuf + l * layer_stride;
    dense_layer<H_STATIC>(m, l, n_nets, io, !last, wave, n_waves, lane);
    __syncthreads();
    cur = hl;
    cur_stride = h_net_stride;
    cur_ld = ld_h;
  }
}

// dW_l[K][N] (+)= h_prev^T[K][16] . delta[16][N]  for n_nets nets, written to a global gradient slab laid out like the
// flat params (slab + net*net_stride + w_off[l]).  Items = (net, k-tile, n-tile); the MFMA "k" runs over the 16 rows.
__device__ __forceinline__ void dense_wgrad(const MlpDev &m, int l, int n_nets, const float *h_prev, int hp_net_stride,
                                            int ld_hp, const float *delta, int d_net_stride, int ld_d, float *slab,
                                            int wave, int n_waves, int lane) {
  const int K = m.dims[l], N = m.dims[l + 1];
  const int KT = (K + 15) >> 4, NT = (N + 15) >> 4;
  const int n_items = n_nets * KT * NT;
  const int r = lane & 15, g = lane >> 4;
  for (int q = wave; q < n_items; q += n_waves) {
    const int net = q / (KT * NT);
    const int rem = q - net * KT * NT;
    const int kt = rem / NT, nt = rem - kt * NT;
    const int kcol = kt * 16 + r, ncol = nt * 16 + r;
    const float *hp = h_prev + net * hp_net_stride;
    const float *dp = delta + net * d_net_stride;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = 4 * g + s;
      float a = kcol < K ? hp[row * ld_hp + kcol] : 0.f;   // A[i = kcol][k = row]
      float b = ncol < N ? dp[row * ld_d + ncol] : 0.f;    // B[k = row][j = ncol]
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    if (ncol < N) {
      float *o = slab + (long long)net * m.net_stride + m.w_off[l];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kt * 16 + 4 * g + i;   // D[row = 4g+i][col = r]  ->  dW[k][ncol]
        if (k < K) o[(long long)k * N + ncol] = acc[i];
      }
    }
  }
}

// db_l[n] = sum over the 16 rows of delta[row][n]
__device__ __forceinline__ void dense_bgrad(const MlpDev &m, int l, int n_nets, const float *delta, int d_net_stride,
                                            int ld_d, float *slab, int tid, int n_threads) {
  const int N = m.dims[l + 1];
  for (int idx = tid; idx < n_nets * N; idx += n_threads) {
    const int net = idx / N, n = idx - net * N;
    const float *dp = delta + net * d_net_stride + n;
    float acc = 0.f;
#pragma unroll
    for (int row = 0; row < 16; ++row) acc += dp[row * ld_d];
    slab[(long long)net * m.net_stride + m.b_off[l] + n] = acc;
  }
}

// delta_prev[16][K] = (delta[16][N] . W_l^T[N][K]) * act'(z_prev)   (z_prev == nullptr: no activation factor, e.g. the
// network input).  Items = (net, k-tile); the MFMA sums over n.  W is read "transposed": lane (j, g) reads the contiguous
// run W[k = kt*16+j][g*nc .. g*nc+nc).
template <int NC_STATIC>
__device__ __forceinline__ void dense_dgrad(const MlpDev &m, int l, int n_nets, const float *delta, int d_net_stride,
                                            int ld_d, const float *z_prev, int z_net_stride, int ld_z, float *delta_prev,
                                            int dp_net_stride, int ld_dp, int wave, int n_waves, int lane) {
  const int K = m.dims[l], N = m.dims[l + 1];
  const int KT = (K + 15) >> 4;
  const int n_items = n_nets * KT;
  const int r = lane & 15, g = lane >> 4;
  const int nc = NC_STATIC > 0 ? NC_STATIC : ((N + 3) >> 2);
  for (int q = wave; q < n_items; q += n_waves) {
    const int net = q / KT, kt = q - net * KT;
    const int kcol = kt * 16 + r;
    const float *dp = delta + net * d_net_stride + r * ld_d + g * nc;                               // A[i = row r][k = n]
    const float *wp = m.params + (long long)net * m.net_stride + m.w_off[l] + (long long)(kcol < K ? kcol : 0) * N + g * nc;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (NC_STATIC > 0) {
#pragma unroll
      for (int s = 0; s < (NC_STATIC > 0 ? NC_STATIC : 1); ++s) {
        float a = dp[s];
        float b = kcol < K ? wp[s] : 0.f;                                                           // B[k = n][j = kcol]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
    } else {
      for (int s = 0; s < nc; ++s) {
        const bool nin = (g * nc + s) < N;
        float a = nin ? dp[s] : 0.f;
        float b = (nin && kcol < K) ? wp[s] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
    }
    if (kcol < K) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * g + i;
        float v = acc[i];
        if (z_prev) v *= act_grad(z_prev[net * z_net_stride + row * ld_z + kcol], m.act);
        delta_prev[net * dp_net_stride + row * ld_dp + kcol] = v;
      }
    }
  }
}

// Full backward of n_nets MLPs on the tile.
//   dY      : LDS [n_nets][16][ld_y]  gradient wrt the output layer's output
//   zbuf/hbuf: from mlp_forward_tile_store
//   dA, dB  : LDS ping-pong delta buffers [n_nets][16][ld_h]
//   slab    : global gradient slab (flat-param layout) or nullptr (no weight gradients, e.g. critic inside the actor loss)
//   dX      : LDS [n_nets][16][ld_x] gradient wrt the network input, or nullptr
// Ends with __syncthreads().
template <int H_STATIC>
__device__ __forceinline__ void mlp_backward_tile(const MlpDev &m, int n_nets, const float *x_in, int x_net_stride, int ld_x,
                                                  const float *zbuf, const float *hbuf, int ld_h, const float *dY, int ld_y,
                                                  float *dA, float *dB, float *slab, float *dX, int wave, int n_waves,
                                                  int lane, int tid, int n_threads) {
  const int h_net_stride = 16 * ld_h;
  const int layer_stride = n_nets * h_net_stride;
  const float *dcur = dY;
  int dcur_stride = 16 * ld_y, dcur_ld = ld_y;
  float *dnext = dA;
  for (int l = m.n_layers - 1; l >= 0; --l) {
    const float *hp = (l == 0) ? x_in : hbuf + (l - 1) * layer_stride;
    const int hp_stride = (l == 0) ? x_net_stride : h_net_stride;
    const int hp_ld = (l == 0) ? ld_x : ld_h;
    if (slab) {
      dense_wgrad(m, l, n_nets, hp, hp_stride, hp_ld, dcur, dcur_stride, dcur_ld, slab, wave, n_waves, lane);
      dense_bgrad(m, l, n_nets, dcur, dcur_stride, dcur_ld, slab, tid, n_threads);
    }
    if (l > 0) {
      const float *zp = zbuf + (l - 1) * layer_stride;
      // z_prev shares delta_prev's geometry: [net][16][ld_h]
      if (m.dims[l + 1] == H_STATIC)
        dense_dgrad<H_STATIC / 4>(m, l, n_nets, dcur, dcur_stride, dcur_ld, zp, h_net_stride, ld_h, dnext, h_net_stride, ld_h, wave, n_waves, lane);
      else
        dense_dgrad<0>(m, l, n_nets, dcur, dcur_stride, dcur_ld, zp, h_net_stride, ld_h, dnext, h_net_stride, ld_h, wave, n_waves, lane);
      __syncthreads();
      dcur = dnext;
      dcur_stride = h_net_stride;
      dcur_ld = ld_h;
      dnext = (dnext == dA) ? dB : dA;
    } else if (dX) {
      dense_dgrad<0>(m, 0, n_nets, dcur, dcur_stride, dcur_ld, nullptr, 0, 0, dX, 16 * ld_x, ld_x, wave, n_waves, lane);
      __syncthreads();
    } else {
      __syncthreads();
    }
  }
}

// wave_mlp.hpp — fp32-MFMA MLP layers where ONE WAVE owns a whole (16-row tile, network) chain.
//
// Why: the nets are tiny (64..128 wide) and the batch tile is 16 rows, so a Dense layer is only 64..256
// v_mfma_f32_16x16x4_f32.  Splitting a layer's columns over waves costs a workgroup barrier per layer and leaves each
// wave with scalar (4-byte) operand loads; at one wave per SIMD that instruction/latency overhead was ~10x the MFMA
// time (profiles/r01_baseline_*).  Here a wave computes ALL n-tiles of a layer and walks the whole chain of layers by
// itself: no barrier inside a chain (LDS is in-order per wave), and the operand maps below make every access a vector.
//
// v_mfma_f32_16x16x4_f32 operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l&15][k = l>>4];   B: lane l holds B[k = l>>4][j = l&15];
//   C/D: lane l, reg i holds D[row = 4*(l>>4) + i][col = l&15].
// Freedom used:
//   * the k index is arbitrary as long as A and B agree -> lane group g = l>>4 takes the contiguous range
//     [g*kc, (g+1)*kc): the A operand of 4 consecutive k-steps is ONE ds_read_b128 of the lane's LDS row;
//   * which matrix column "col j of n-tile t" denotes is arbitrary -> n = NT*j + t: the B operands of a k-step's NT
//     n-tiles are NT CONSECUTIVE floats of a weight row (one global_load_dwordx4 for NT=4), and the NT results a lane
//     holds for a row are consecutive columns (one ds_write_b128).  Same trick on the row index of weight gradients.
// LDS tiles are row-major [16][ld], ld % 4 == 0, true column order (the permutations never leave the registers).
#pragma once
#include "common.hpp"

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // global memory, dword-aligned (flat params)
typedef float f4a __attribute__((ext_vector_type(4)));              // LDS, 16-byte aligned

#define WAVE_FENCE() __builtin_amdgcn_wave_barrier()

template <int NV>
__device__ __forceinline__ void load_vec_global(const float *p, float (&v)[NV]) {
  if constexpr (NV % 4 == 0) {
#pragma unroll
    for (int c = 0; c < NV / 4; ++c) {
      f4u t = *reinterpret_cast<const f4u *>(p + 4 * c);
      v[4 * c + 0] = t[0]; v[4 * c + 1] = t[1]; v[4 * c + 2] = t[2]; v[4 * c + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c] = p[c];
  }
}

template <int NV>
__device__ __forceinline__ void load_vec_lds(const float *p, float (&v)[NV]) {
  if constexpr (NV % 4 == 0) {
#pragma unroll
    for (int c = 0; c < NV / 4; ++c) {
      f4a t = *reinterpret_cast<const f4a *>(p + 4 * c);
      v[4 * c + 0] = t[0]; v[4 * c + 1] = t[1]; v[4 * c + 2] = t[2]; v[4 * c + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c] = p[c];
  }
}

template <int NV>
__device__ __forceinline__ void store_vec_lds(float *p, const float (&v)[NV]) {
  if constexpr (NV % 4 == 0) {
#pragma unroll
    for (int c = 0; c < NV / 4; ++c) {
      f4a t = {v[4 * c + 0], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
      *reinterpret_cast<f4a *>(p + 4 * c) = t;
    }
  } else {
#pragma unroll
    for (int c = 0; c < NV; ++c) p[c] = v[c];
  }
}

template <int NV>
__device__ __forceinline__ void store_vec_global(float *p, const float (&v)[NV]) {
  if constexpr (NV % 4 == 0) {
#pragma unroll
    for (int c = 0; c < NV / 4; ++c) {
      f4u t = {v[4 * c + 0], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
      *reinterpret_cast<f4u *>(p + 4 * c) = t;
    }
  } else {
#pragma unroll
    for (int c = 0; c < NV; ++c) p[c] = v[c];
  }
}

// ------------------------------------------------------------------------------------------------ forward
// y[16][N] = x[16][K] @ W[K][N] + b ; z_out (optional) gets the pre-activation, h_out gets act(y) (act < 0: linear).
// NT = number of 16-column n-tiles this wave carries (N <= 16*NT).
// Every shape class has a fully unrolled, branch-free operand path: a runtime-trip-count loop (or a masked load, which
// hipcc turns into a branch) makes the compiler wait vmcnt(0) after each single load — one L2 round trip per k-step.
//   KS > 0  : K == 4*KS known at compile time (layers fed by a hidden layer): B fetched block-wise into two register
//             banks, every load of a bank in flight before its MFMAs.
//   KS == 0 : K <= 32 (the network input layer): 8 unrolled k-steps, out-of-range steps get a = 0 and a clamped B row.
//   otherwise a generic runtime loop (not used by the shapes this library dispatches).
// Columns beyond N are handled by clamping the B column and dropping the result.
// Slicing: W / bias / h_out / z_out may point at a column offset of the full layer; `ldw` is the row stride of W (the
// full layer width) and N the number of columns this call produces — that is how several waves share one layer.
template <int NT, int KS = 0>
__device__ __forceinline__ void wave_dense_fwd(const float *x, int ldx, int K, const float *__restrict__ W, int ldw,
                                               const float *__restrict__ bias, int N, float *h_out, float *z_out, int ldo,
                                               int act, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const int kc = (K + 3) >> 2;
  const bool full_n = (N == 16 * NT);
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *xr = x + r * ldx + g * kc;
  // column base of this lane's NT outputs; clamped so that partial tiles still form legal addresses
  int ncol[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) ncol[t] = (NT * r + t < N) ? NT * r + t : 0;
  if (KS > 0 && K == 4 * KS) {
    constexpr int KSS = KS > 0 ? KS : 8;
    constexpr int KB = 8;  // k-steps per register bank
    constexpr int NB = KSS / KB;
    const float *wr = W + (g * KSS) * ldw;
    float bv[2][KB][NT];
    auto load_bank = [&](int blk, float (&bank)[KB][NT]) {
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const float *row = wr + (blk * KB + u) * ldw;
        if (full_n) {
          load_vec_global<NT>(row + NT * r, bank[u]);
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t) bank[u][t] = row[ncol[t]];
        }
      }
    };
    load_bank(0, bv[0]);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      if (blk + 1 < NB) load_bank(blk + 1, bv[(blk + 1) & 1]);
#pragma unroll
      for (int q = 0; q < KB / 4; ++q) {
        float av[4];
        load_vec_lds<4>(xr + blk * KB + 4 * q, av);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[blk & 1][4 * q + u][t], acc[t], 0, 0, 0);
      }
    }
  } else if (kc <= 8) {
    float av[8], bv[8][NT];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = g * kc + s;
      const bool ok = (s < kc) && (k < K);
      const int kk = ok ? k : 0;
      av[s] = x[r * ldx + kk];
      av[s] = ok ? av[s] : 0.f;
      const float *row = W + kk * ldw;
      if (full_n) {
        load_vec_global<NT>(row + NT * r, bv[s]);
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[s][t] = row[ncol[t]];
      }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][t], acc[t], 0, 0, 0);
  } else {
    for (int s = 0; s < kc; ++s) {
      const int k = g * kc + s;
      const bool kin = k < K;
      const int kk = kin ? k : 0;
      float a = x[r * ldx + kk];
      a = kin ? a : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, W[kk * ldw + ncol[t]], acc[t], 0, 0, 0);
    }
  }
  // epilogue: lane holds y[row = 4g+i][n = NT*r + t]
  if (full_n) {
    float bb[NT];
    load_vec_global<NT>(bias + NT * r, bb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float zv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) zv[t] = acc[t][i] + bb[t];
      const int o = (4 * g + i) * ldo + NT * r;
      if (z_out) store_vec_lds<NT>(z_out + o, zv);
      if (act >= 0) act_apply_vec<NT>(zv, act);
      store_vec_lds<NT>(h_out + o, zv);
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = NT * r + t;
      const float bb = bias[ncol[t]];
      if (n < N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float zv = acc[t][i] + bb;
          const int o = (4 * g + i) * ldo + n;
          if (z_out) z_out[o] = zv;
          h_out[o] = act >= 0 ? act_apply(zv, act) : zv;
        }
      }
    }
  }
}

// Whole-MLP forward by one wave.  HT = hidden width / 16 (4 for 64, 8 for 128, 16 for 256).
//   x      : LDS [16][ldx]
//   pp0/pp1: LDS ping-pong hidden tiles [16][ldh] (used when zbuf == nullptr)
//   zbuf/hbuf: optional stores for backward: layer l's z / h at zbuf + l*tile, hbuf + l*tile (tile = 16*ldh);
//              hbuf may be nullptr (then h ping-pongs through pp0/pp1: input-gradient-only backward never reads h)
//   y      : LDS [16][ldy] output layer (linear)
template <int HT>
__device__ __forceinline__ void wave_mlp_fwd(const MlpDev &m, const float *params, const float *x, int ldx, float *pp0,
                                             float *pp1, float *zbuf, float *hbuf, int ldh, float *y, int ldy, int lane) {
  const int tile = 16 * ldh;
  const float *cur = x;
  int ldc = ldx;
  for (int l = 0; l < m.n_layers; ++l) {
    const bool last = (l == m.n_layers - 1);
    const int K = m.dims[l], N = m.dims[l + 1];
    const float *W = params + m.w_off[l], *b = params + m.b_off[l];
    if (last) {
      // output layer: K is the hidden width (static path) unless the MLP has no hidden layer at all
      if (K == 16 * HT) {
        if (N <= 16) wave_dense_fwd<1, 4 * HT>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
        else if (N <= 32) wave_dense_fwd<2, 4 * HT>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
        else if (N <= 64) wave_dense_fwd<4, 4 * HT>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
        else wave_dense_fwd<8, 4 * HT>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
      } else {
        if (N <= 16) wave_dense_fwd<1>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
        else if (N <= 32) wave_dense_fwd<2>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
        else if (N <= 64) wave_dense_fwd<4>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
        else wave_dense_fwd<8>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
      }
    } else {
      float *ho = hbuf ? hbuf + l * tile : ((l & 1) ? pp1 : pp0);
      float *zo = zbuf ? zbuf + l * tile : nullptr;
      if (K == 16 * HT) wave_dense_fwd<HT, 4 * HT>(cur, ldc, K, W, N, b, N, ho, zo, ldh, m.act, lane);
      else wave_dense_fwd<HT>(cur, ldc, K, W, N, b, N, ho, zo, ldh, m.act, lane);
      cur = ho;
      ldc = ldh;
    }
    WAVE_FENCE();
  }
}

// ------------------------------------------------------------------------------------------------ backward pieces
// dW[K][N] = x^T[K][16] . delta[16][N], written to `gW` (global, row-major [K][N]); the MFMA k runs over the 16 rows.
// Tiles (kt, nt) in chunks of KTC x NTC accumulators; row map k = KT*rho + kt, column map n = NT*j + t.
// Slicing: x / delta / gW may point at a column offset; `ldw` is the row stride of gW (full layer width), K and N the
// extents of the block this call produces.
// accumulate: gW += (the slab belongs to this workgroup alone, so a plain read-modify-write is race-free).
template <int KT, int NT>
__device__ __forceinline__ void wave_dense_wgrad(const float *x, int ldx, int K, const float *delta, int ldd, int N,
                                                 float *__restrict__ gW, int ldw, int lane, bool accumulate = false) {
  const int r = lane & 15, g = lane >> 4;
  constexpr int KTC = KT > 4 ? 4 : KT, NTC = NT > 4 ? 4 : NT;
  const bool full = (K == 16 * KT) && (N == 16 * NT);
  for (int kt0 = 0; kt0 < KT; kt0 += KTC) {
    for (int nt0 = 0; nt0 < NT; nt0 += NTC) {
      f32x4 acc[KTC][NTC];
#pragma unroll
      for (int a = 0; a < KTC; ++a)
#pragma unroll
        for (int b = 0; b < NTC; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = 4 * g + s;
        float av[KTC], bv[NTC];
        if (full) {
          load_vec_lds<KTC>(x + row * ldx + KT * r + kt0, av);         // A[i -> k = KT*r + kt][k_mfma = row]
          load_vec_lds<NTC>(delta + row * ldd + NT * r + nt0, bv);     // B[k_mfma = row][j -> n = NT*r + t]
        } else {
#pragma unroll
          for (int a = 0; a < KTC; ++a) {
            const int k = KT * r + kt0 + a;
            av[a] = k < K ? x[row * ldx + k] : 0.f;
          }
#pragma unroll
          for (int b = 0; b < NTC; ++b) {
            const int n = NT * r + nt0 + b;
            bv[b] = n < N ? delta[row * ldd + n] : 0.f;
          }
        }
#pragma unroll
        for (int a = 0; a < KTC; ++a)
#pragma unroll
          for (int b = 0; b < NTC; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
      // D[rho = 4g+i][col j] of tile (kt, nt)  ->  dW[k = KT*rho + kt][n = NT*j + t]
#pragma unroll
      for (int a = 0; a < KTC; ++a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = KT * (4 * g + i) + kt0 + a;
          if (full) {
            float ov[NTC];
#pragma unroll
            for (int b = 0; b < NTC; ++b) ov[b] = acc[a][b][i];
            float *dst = gW + k * ldw + NT * r + nt0;
            if (accumulate) {
              float old[NTC];
              load_vec_global<NTC>(dst, old);
#pragma unroll
              for (int b = 0; b < NTC; ++b) ov[b] += old[b];
            }
            store_vec_global<NTC>(dst, ov);
          } else if (k < K) {
#pragma unroll
            for (int b = 0; b < NTC; ++b) {
              const int n = NT * r + nt0 + b;
              if (n < N) gW[k * ldw + n] = accumulate ? gW[k * ldw + n] + acc[a][b][i] : acc[a][b][i];
            }
          }
        }
      }
    }
  }
}

// db[n] = sum_rows delta[row][n]
__device__ __forceinline__ void wave_dense_bgrad(const float *delta, int ldd, int N, float *__restrict__ gb, int lane,
                                                 bool accumulate = false) {
  for (int n = lane; n < N; n += 64) {
    float acc = 0.f;
#pragma unroll
    for (int row = 0; row < 16; ++row) acc += delta[row * ldd + n];
    gb[n] = accumulate ? gb[n] + acc : acc;
  }
}

// dx[16][K] = (delta[16][N] . W^T) * act'(z_prev)    (z_prev == nullptr: no activation factor)
// The MFMA sums over n: lane group g takes n in [g*nc, (g+1)*nc); output column map k = KT*j + kt (K <= 16*KT).
//   NS > 0 : N == 4*NS known at compile time (delta of a hidden layer): W fetched block-wise into two register banks;
//   NS == 0: N <= 32 (delta of the output layer): 8 unrolled n-steps with clamped addresses.
// Slicing: W may point at a ROW offset (W + k0*ldw) with z_prev / dx at the matching column offset k0; K is then the
// number of output columns of this call.  `ldw` = row stride of W.
template <int KT, int NS = 0>
__device__ __forceinline__ void wave_dense_dgrad(const float *delta, int ldd, int N, const float *__restrict__ W, int ldw,
                                                 int K, const float *z_prev, int ldz, int act, float *dx, int ldx, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const int nc = (N + 3) >> 2;
  const bool full_k = (K == 16 * KT);
  f32x4 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // W rows of this lane's KT outputs (B[k_mfma = n][j -> k = KT*r + t] = W[k][n]); clamped for partial tiles
  const float *wrow[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) wrow[t] = W + ((KT * r + t < K) ? KT * r + t : 0) * ldw;
  if (NS > 0 && N == 4 * NS) {
    constexpr int NSS = NS > 0 ? NS : 8;
    constexpr int NBK = 8;  // n-steps per bank
    constexpr int NB = NSS / NBK;
    const float *dr = delta + r * ldd + g * NSS;  // A[i = row r][k_mfma = n]
    float bv[2][KT][NBK];
#pragma unroll
    for (int t = 0; t < KT; ++t) load_vec_global<NBK>(wrow[t] + g * NSS, bv[0][t]);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      if (blk + 1 < NB) {
#pragma unroll
        for (int t = 0; t < KT; ++t) load_vec_global<NBK>(wrow[t] + g * NSS + (blk + 1) * NBK, bv[(blk + 1) & 1][t]);
      }
#pragma unroll
      for (int q = 0; q < NBK / 4; ++q) {
        float av[4];
        load_vec_lds<4>(dr + blk * NBK + 4 * q, av);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < KT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[blk & 1][t][4 * q + u], acc[t], 0, 0, 0);
      }
    }
  } else if (nc <= 8) {
    float av[8], bv[8][KT];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int n = g * nc + s;
      const bool ok = (s < nc) && (n < N);
      const int nn = ok ? n : 0;
      av[s] = delta[r * ldd + nn];
      av[s] = ok ? av[s] : 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) bv[s][t] = wrow[t][nn];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][t], acc[t], 0, 0, 0);
  } else {
    for (int s = 0; s < nc; ++s) {
      const int n = g * nc + s;
      const bool nin = n < N;
      const int nn = nin ? n : 0;
      float a = delta[r * ldd + nn];
      a = nin ? a : 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wrow[t][nn], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * g + i;
    if (full_k) {
      float ov[KT];
#pragma unroll
      for (int t = 0; t < KT; ++t) ov[t] = acc[t][i];
      if (z_prev) {
        float zv[KT];
        load_vec_lds<KT>(z_prev + row * ldz + KT * r, zv);
        act_grad_mul_vec<KT>(ov, zv, act);
      }
      store_vec_lds<KT>(dx + row * ldx + KT * r, ov);
    } else {
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int k = KT * r + t;
        if (k < K) {
          float v = acc[t][i];
          if (z_prev) v *= act_grad(z_prev[row * ldz + k], act);
          dx[row * ldx + k] = v;
        }
      }
    }
  }
}

// One backward layer step, split in the two halves that two waves run side by side:
//   dgrad half: delta_{l-1} = dgrad(l)(delta_l) * act'(z_{l-1})     (l > 0), or dX = dgrad(0)(delta_0) when wanted
//   wgrad half: dW_l, db_l from (h_{l-1} | x, delta_l)
// HT = hidden width / 16.
template <int HT>
__device__ __forceinline__ void wave_bwd_dgrad_step(const MlpDev &m, const float *params, int l, const float *delta, int ldd,
                                                    const float *zbuf, int ldh, float *dprev, float *dX, int ldx_in, int lane) {
  const int K = m.dims[l], N = m.dims[l + 1];
  const float *W = params + m.w_off[l];
  if (l > 0) {
    const float *zp = zbuf + (l - 1) * 16 * ldh;
    if (N == 16 * HT) wave_dense_dgrad<HT, 4 * HT>(delta, ldd, N, W, N, K, zp, ldh, m.act, dprev, ldh, lane);
    else wave_dense_dgrad<HT>(delta, ldd, N, W, N, K, zp, ldh, m.act, dprev, ldh, lane);
  } else if (dX) {
    // network input: K = dims[0] is small (<= 16*IT columns); N is the hidden width (static) when a hidden layer exists
    if (N == 16 * HT) {
      if (K <= 16) wave_dense_dgrad<1, 4 * HT>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
      else if (K <= 32) wave_dense_dgrad<2, 4 * HT>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
      else wave_dense_dgrad<4, 4 * HT>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
    } else {
      if (K <= 16) wave_dense_dgrad<1>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
      else if (K <= 32) wave_dense_dgrad<2>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
      else wave_dense_dgrad<4>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
    }
  }
  WAVE_FENCE();
}

template <int HT>
__device__ __forceinline__ void wave_bwd_wgrad_step(const MlpDev &m, int l, const float *x_in, int ldx_in, const float *hbuf,
                                                    int ldh, const float *delta, int ldd, float *slab, int lane) {
  const int K = m.dims[l], N = m.dims[l + 1];
  const float *hp = (l == 0) ? x_in : hbuf + (l - 1) * 16 * ldh;
  const int ldp = (l == 0) ? ldx_in : ldh;
  float *gW = slab + m.w_off[l], *gb = slab + m.b_off[l];
  const bool last = (l == m.n_layers - 1);
  if (l == 0) {
    // K small (network input), N = hidden width
    if (K <= 16) wave_dense_wgrad<1, HT>(hp, ldp, K, delta, ldd, N, gW, N, lane);
    else if (K <= 32) wave_dense_wgrad<2, HT>(hp, ldp, K, delta, ldd, N, gW, N, lane);
    else wave_dense_wgrad<4, HT>(hp, ldp, K, delta, ldd, N, gW, N, lane);
  } else if (last) {
    if (N <= 16) wave_dense_wgrad<HT, 1>(hp, ldp, K, delta, ldd, N, gW, N, lane);
    else if (N <= 32) wave_dense_wgrad<HT, 2>(hp, ldp, K, delta, ldd, N, gW, N, lane);
    else wave_dense_wgrad<HT, 4>(hp, ldp, K, delta, ldd, N, gW, N, lane);
  } else {
    wave_dense_wgrad<HT, HT>(hp, ldp, K, delta, ldd, N, gW, N, lane);
  }
  wave_dense_bgrad(delta, ldd, N, gb, lane);
}

// Input-gradient-only backward of a whole MLP by one wave (the critics inside the actor loss): no weight gradients.
//   dY [16][ldy] -> dX [16][ldx_in];  zbuf from the forward;  d0/d1 LDS ping-pong delta tiles [16][ldh].
template <int HT>
__device__ __forceinline__ void wave_mlp_bwd_input(const MlpDev &m, const float *params, const float *zbuf, int ldh,
                                                   const float *dY, int ldy, float *d0, float *d1, float *dX, int ldx_in,
                                                   int lane) {
  const float *dcur = dY;
  int ldc = ldy;
  for (int l = m.n_layers - 1; l >= 0; --l) {
    float *dn = (l & 1) ? d1 : d0;
    wave_bwd_dgrad_step<HT>(m, params, l, dcur, ldc, zbuf, ldh, dn, dX, ldx_in, lane);
    dcur = dn;
    ldc = ldh;
  }
}

// =================================================================================================
// Lockstep groups: SP waves share one chain, each producing a slice of H/SP columns of every hidden layer.  The caller
// walks the layers and places ONE workgroup barrier per layer; several chains (nets) advance side by side, so the
// barrier count is the depth of one net, not the sum over nets.  HT = H/16, CT = HT/SP n-tiles per wave.
// =================================================================================================

// forward layer l of one chain; `cur` is the layer input, ho/zo the layer's hidden output tiles (zo optional), y the
// output-layer tile.  Wave `sub` of the chain's SP waves.
template <int HT, int SP>
__device__ __forceinline__ void group_fwd_layer(const MlpDev &m, const float *params, int l, const float *cur, int ldc,
                                                float *ho, float *zo, int ldh, float *y, int ldy, int sub, int lane) {
  constexpr int CT = HT / SP;
  const int K = m.dims[l], N = m.dims[l + 1];
  const float *W = params + m.w_off[l], *b = params + m.b_off[l];
  if (l < m.n_layers - 1) {
    const int c0 = sub * 16 * CT;
    if (K == 16 * HT) wave_dense_fwd<CT, 4 * HT>(cur, ldc, K, W + c0, N, b + c0, 16 * CT, ho + c0, zo ? zo + c0 : nullptr, ldh, m.act, lane);
    else wave_dense_fwd<CT>(cur, ldc, K, W + c0, N, b + c0, 16 * CT, ho + c0, zo ? zo + c0 : nullptr, ldh, m.act, lane);
  } else if (sub == 0) {
    if (K == 16 * HT) {
      if (N <= 16) wave_dense_fwd<1, 4 * HT>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
      else if (N <= 32) wave_dense_fwd<2, 4 * HT>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
      else if (N <= 64) wave_dense_fwd<4, 4 * HT>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
      else wave_dense_fwd<8, 4 * HT>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
    } else {
      if (N <= 16) wave_dense_fwd<1>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
      else if (N <= 32) wave_dense_fwd<2>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
      else if (N <= 64) wave_dense_fwd<4>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
      else wave_dense_fwd<8>(cur, ldc, K, W, N, b, N, y, nullptr, ldy, -1, lane);
    }
  }
}

// Bookkeeping of a forward chain walked layer by layer (which tile is the current input / output).
struct FwdChain {
  const MlpDev *m;
  const float *params;
  const float *x;   // network input tile
  int ldx;
  float *pp0, *pp1; // ping-pong hidden tiles (used when hbuf == nullptr)
  float *zbuf, *hbuf;  // optional per-layer stores (layer l at + l*tile)
  float *y;         // output tile
};

template <int HT, int SP>
__device__ __forceinline__ void group_fwd_step(const FwdChain &c, int l, int ldh, int ldy, int sub, int lane) {
  if (l >= c.m->n_layers) return;
  const int tile = 16 * ldh;
  const float *cur;
  int ldc;
  if (l == 0) {
    cur = c.x;
    ldc = c.ldx;
  } else {
    cur = c.hbuf ? c.hbuf + (l - 1) * tile : (((l - 1) & 1) ? c.pp1 : c.pp0);
    ldc = ldh;
  }
  float *ho = c.hbuf ? c.hbuf + l * tile : ((l & 1) ? c.pp1 : c.pp0);
  float *zo = c.zbuf ? c.zbuf + l * tile : nullptr;
  group_fwd_layer<HT, SP>(*c.m, c.params, l, cur, ldc, ho, zo, ldh, c.y, ldy, sub, lane);
}

// backward layer l, dgrad half of one chain: delta_{l-1}[:, slice] (l > 0) or dX (l == 0, wave 0 only).
template <int HT, int SP>
__device__ __forceinline__ void group_bwd_dgrad_layer(const MlpDev &m, const float *params, int l, const float *delta, int ldd,
                                                      const float *zbuf, int ldh, float *dprev, float *dX, int ldx_in,
                                                      int sub, int lane) {
  constexpr int CT = HT / SP;
  const int K = m.dims[l], N = m.dims[l + 1];
  const float *W = params + m.w_off[l];
  if (l > 0) {
    const int k0 = sub * 16 * CT;  // K == 16*HT: this wave's slice of the previous hidden layer
    const float *zp = zbuf + (l - 1) * 16 * ldh + k0;
    if (N == 16 * HT) wave_dense_dgrad<CT, 4 * HT>(delta, ldd, N, W + k0 * N, N, 16 * CT, zp, ldh, m.act, dprev + k0, ldh, lane);
    else wave_dense_dgrad<CT>(delta, ldd, N, W + k0 * N, N, 16 * CT, zp, ldh, m.act, dprev + k0, ldh, lane);
  } else if (dX && sub == 0) {
    if (N == 16 * HT) {
      if (K <= 16) wave_dense_dgrad<1, 4 * HT>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
      else if (K <= 32) wave_dense_dgrad<2, 4 * HT>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
      else wave_dense_dgrad<4, 4 * HT>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
    } else {
      if (K <= 16) wave_dense_dgrad<1>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
      else if (K <= 32) wave_dense_dgrad<2>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
      else wave_dense_dgrad<4>(delta, ldd, N, W, N, K, nullptr, 0, 0, dX, ldx_in, lane);
    }
  }
}

// backward layer l, wgrad half of one chain: dW_l[:, slice], db_l[slice] from (h_{l-1} | x, delta_l).
template <int HT, int SP>
__device__ __forceinline__ void group_bwd_wgrad_layer(const MlpDev &m, int l, const float *x_in, int ldx_in, const float *hbuf,
                                                      int ldh, const float *delta, int ldd, float *slab, int sub, int lane,
                                                      bool accumulate = false) {
  constexpr int CT = HT / SP;
  const int K = m.dims[l], N = m.dims[l + 1];
  const float *hp = (l == 0) ? x_in : hbuf + (l - 1) * 16 * ldh;
  const int ldp = (l == 0) ? ldx_in : ldh;
  float *gW = slab + m.w_off[l], *gb = slab + m.b_off[l];
  if (l == m.n_layers - 1) {
    // output layer: N small.  Split the K rows of dW over the SP waves instead (K == 16*HT).
    const int k0 = sub * 16 * CT;
    if (N <= 16) wave_dense_wgrad<CT, 1>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, lane, accumulate);
    else if (N <= 32) wave_dense_wgrad<CT, 2>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, lane, accumulate);
    else wave_dense_wgrad<CT, 4>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, lane, accumulate);
    if (sub == 0) wave_dense_bgrad(delta, ldd, N, gb, lane, accumulate);
  } else {
    const int c0 = sub * 16 * CT;  // N == 16*HT: column slice
    if (l == 0) {
      if (K <= 16) wave_dense_wgrad<1, CT>(hp, ldp, K, delta + c0, ldd, 16 * CT, gW + c0, N, lane, accumulate);
      else if (K <= 32) wave_dense_wgrad<2, CT>(hp, ldp, K, delta + c0, ldd, 16 * CT, gW + c0, N, lane, accumulate);
      else wave_dense_wgrad<4, CT>(hp, ldp, K, delta + c0, ldd, 16 * CT, gW + c0, N, lane, accumulate);
    } else {
      wave_dense_wgrad<HT, CT>(hp, ldp, K, delta + c0, ldd, 16 * CT, gW + c0, N, lane, accumulate);
    }
    wave_dense_bgrad(delta + c0, ldd, 16 * CT, gb + c0, lane, accumulate);
  }
}

// =================================================================================================
// Generic (any shape) compact routines: runtime loops, scalar operand loads.  They serve the shapes that have no
// specialised fast path (network inputs wider than 32, outputs wider than the wave's slice, ...) at a fraction of the
// code size of one unrolled variant per shape: correctness everywhere, speed where the shapes are the common ones.
// =================================================================================================
// y[16][c_lo..c_hi) = x[16][K] W[K][ldw] + b, standard column map (col j of an n-tile = column c_lo + 16*nt + j)
__device__ __forceinline__ void gen_dense_fwd(const float *x, int ldx, int K, const float *__restrict__ W, int ldw,
                                              const float *__restrict__ bias, int c_lo, int c_hi, float *h_out, float *z_out, int ldo,
                                              int act, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const int kc = (K + 3) >> 2;
#pragma nounroll
  for (int n0 = c_lo; n0 < c_hi; n0 += 16) {
    const int n = n0 + r;
    const bool nin = n < c_hi;
    const int nn = nin ? n : c_lo;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma nounroll
    for (int s = 0; s < kc; ++s) {
      const int k = g * kc + s;
      const bool kin = k < K;
      const int kk = kin ? k : 0;
      float a = x[r * ldx + kk];
      a = kin ? a : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, W[kk * ldw + nn], acc, 0, 0, 0);
    }
    const float bb = bias[nn];
    if (nin) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float zv = acc[i] + bb;
        const int o = (4 * g + i) * ldo + n;
        if (z_out) z_out[o] = zv;
        h_out[o] = act >= 0 ? act_apply(zv, act) : zv;
      }
    }
  }
}

// dx[16][k_lo..k_hi) = (delta[16][N] . W[k][0..N)^T) * act'(z_prev)   (z_prev == nullptr: no activation factor)
__device__ __forceinline__ void gen_dense_dgrad(const float *delta, int ldd, int N, const float *__restrict__ W, int ldw, int k_lo,
                                                int k_hi, const float *z_prev, int ldz, int act, float *dx, int ldx, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const int nc = (N + 3) >> 2;
#pragma nounroll
  for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
    const int k = k0 + r;
    const bool kin = k < k_hi;
    const float *wrow = W + (kin ? k : k_lo) * ldw;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma nounroll
    for (int s = 0; s < nc; ++s) {
      const int n = g * nc + s;
      const bool nin = n < N;
      const int nn = nin ? n : 0;
      float a = delta[r * ldd + nn];
      a = nin ? a : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wrow[nn], acc, 0, 0, 0);
    }
    if (kin) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * g + i;
        float v = acc[i];
        if (z_prev) v *= act_grad(z_prev[row * ldz + k], act);
        dx[row * ldx + k] = v;
      }
    }
  }
}

// gW[k][n] (+)= sum_rows x[row][k] delta[row][n]  for k in [k_lo,k_hi), n in [n_lo,n_hi); gW row stride ldw
__device__ __forceinline__ void gen_dense_wgrad(const float *x, int ldx, int k_lo, int k_hi, const float *delta, int ldd, int n_lo,
                                                int n_hi, float *__restrict__ gW, int ldw, int lane, bool accumulate) {
  const int r = lane & 15, g = lane >> 4;
#pragma nounroll
  for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
#pragma nounroll
    for (int n0 = n_lo; n0 < n_hi; n0 += 16) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int ka = k0 + r, nb = n0 + r;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = 4 * g + s;
        const float a = ka < k_hi ? x[row * ldx + ka] : 0.f;
        const float b = nb < n_hi ? delta[row * ldd + nb] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + 4 * g + i;
        if (k < k_hi && nb < n_hi) gW[k * ldw + nb] = accumulate ? gW[k * ldw + nb] + acc[i] : acc[i];
      }
    }
  }
}

// phase kinds of a wave inside a kernel built on chain_run.hpp
enum { CH_IDLE = 0, CH_FWD = 1, CH_DGRAD = 2, CH_WGRAD = 3 };

// layered.hpp — the layer-by-layer path for network shapes the fused wave-chain kernels are not built for (hidden widths above 128,
// or more stored activations than a 16-row tile's 160 KiB of LDS holds).  The reference accepts any tuple of hidden sizes
// (sac/sac.py:84-88, ppo/ppo.py:60-63; experiments/train_inverted_pendulum/exp_ppo.py uses a 256x5 critic): here a Dense layer is
// then ONE fp32-MFMA GEMM launch over the whole minibatch (activations in HBM/L2 between layers) instead of a step of a
// register/LDS-resident chain.  Slower per update than the fused kernels (launch-bound: ~45 launches), any shape.
#pragma once
#include "common.hpp"

// C(m, n) = sum_k A(m, k) * B(k, n), fp32 v_mfma_f32_16x16x4_f32, 64 x 64 output tile per workgroup, K through LDS in steps of 16.
// Operands by strides (elements): A(m,k) = A[z*zA + m*sam + k*sak], B(k,n) = B[z*zB + k*sbk + n*sbn], C(m,n) = C[z*zC + m*ldc + n];
// z = batch index (the two critics).  Out-of-range rows / columns / k read as zero and are not written.
struct GemmArgs {
  const float *A, *B;
  float *C, *C2;                     // MODE 0: C = pre-activation (may be null), C2 = activation (may be null)
  const float *bias;                 // MODE 0: [N] added before the activation (may be null)
  const float *Zprev;                // MODE 1: C = acc * act'(Zprev(m, n)) (null: C = acc)
  long long sam, sak, zA, sbk, sbn, zB, ldc, zC, zBias, ldz, zZ;
  int M, N, K, nz;
  int act;                           // MBPO_ACT_* or -1 (identity)
  int ones_row;                      // A's row M-1 reads as 1.0 for every k: the bias gradient as one more row of a weight gradient
  int n_split, k_chunk;              // split of the k range over blockIdx.z (weight gradients: k = minibatch rows); partial s of batch z
  long long zSplit;                  //   is written at C + (z * n_split + s) * zSplit ... only when n_split > 1 (then zC is ignored)
};
// MODE 0: forward (bias, activation), 1: input gradient (times act'), 2: plain
int layered_gemm(int mode, const GemmArgs &G, hipStream_t st);
// out[z][i] = sum_s part[(z * n_split + s) * stride + i], s in order (deterministic), i < n
int layered_split_sum(const float *part, long long stride, int n_split, int nz, float *out, long long out_stride, long long n, hipStream_t st);

// An MLP's layers run as GEMMs.  acts: per hidden layer l = 1..L-1 the buffers Z[l], H[l] ([nz][rows][dims[l]], nz-major).
struct LayeredNet {
  const float *params;               // net z at params + z * net_stride
  long long net_stride;
  int nz, L, act;
  int dims[MBPO_MAX_LAYERS + 1], w_off[MBPO_MAX_LAYERS], b_off[MBPO_MAX_LAYERS];
};
LayeredNet layered_net(const MlpDev &m, const float *params, long long net_stride, int nz);
// forward of `rows` inputs x ([rows][dims[0]], shared by the nz nets unless zx != 0).  Z / H: arrays indexed by layer (entries 1..L-1;
// Z entries may be null = not stored); y [nz][rows][dims[L]].
int layered_forward(const LayeredNet &n, const float *x, long long zx, int rows, float *const *Z, float *const *H, float *y, hipStream_t st);
// backward from dy [nz][rows][dims[L]]: weight gradients (with the bias gradient as the last row) to dw + z * dw_stride + w_off[l] when
// dw != null; dx [nz][rows][dims[0]] when dx != null.  x, Z, H as given to layered_forward (H[l] = input of layer l for l >= 1).
// tmp0 / tmp1: [nz][rows][max hidden] each; part: split-k partials (layered_part_floats).
int layered_backward(const LayeredNet &n, const float *x, long long zx, int rows, float *const *Z, float *const *H, const float *dy,
                     float *dw, long long dw_stride, float *dx, float *tmp0, float *tmp1, float *part, hipStream_t st);
long long layered_part_floats(const LayeredNet &n, int rows);
// Several passes advanced level by level, the problems of a level in ONE launch (round 4: the path is launch-bound).  Passes of one call
// must not share output / scratch buffers (Z, H, y; tmp0, tmp1, part): they run side by side.
struct LayeredFwd {
  LayeredNet net;
  const float *x; long long zx; int rows;
  float *const *Z; float *const *H; float *y;
};
struct LayeredBwd {
  LayeredNet net;
  const float *x; long long zx; int rows;
  float *const *Z; float *const *H; const float *dy;
  float *dw; long long dw_stride; float *dx; float *tmp0, *tmp1, *part;
};
int layered_forward_multi(const LayeredFwd *passes, int n_pass, hipStream_t st);
int layered_backward_multi(const LayeredBwd *passes, int n_pass, hipStream_t st);
int layered_max_hidden(const LayeredNet &n);

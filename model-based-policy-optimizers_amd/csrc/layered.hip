// layered.hip — Dense layers as whole-minibatch fp32-MFMA GEMM launches (see layered.hpp).
#include "layered.hpp"
#include <stdlib.h>

namespace {
constexpr int BK = 64;
// LDS layouts, chosen by which index is contiguous in global memory so that both the global loads and the LDS stores of a tile are
// unit-stride across lanes:  k contiguous -> [row][k] with row stride LDK (68 % 32 == 4: the 16 rows x 4 k a wave's operand read
// touches fall in 32 distinct banks, twice);  row/column contiguous -> [k][row] with stride tile + 16 (% 32 == 16: same property).
constexpr int LDK = BK + 4;

// One launch = one Dense layer over the whole minibatch.  Output tile per workgroup 32T x 32T (2 x 2 waves, T x T MFMA blocks each):
// T = 1 unless the problem is huge (layered_gemm) — fp32 MFMA is 256 FLOP/cycle/CU, so a 256 x 256 x 256 layer on sixteen 64 x 64
// tiles is 8 k cycles on 16 CUs; on sixty-four 32 x 32 tiles 2 k cycles on 64.  K walks through LDS in chunks of 64;
// the NEXT chunk's global loads are issued (into registers) before the current chunk's MFMAs.
template <int T>
struct GemmTile {
  static constexpr int BM = 32 * T, LDT = BM + 16, PER_T = BM * BK / 256;
  static constexpr int TILE_F = (BK * LDT > BM * LDK) ? BK * LDT : BM * LDK;
};

// One output tile of one problem.  MODE is the epilogue (a compile-time constant in k_layered_gemm<MODE, T>, a uniform run-time value in
// k_layered_group<T>); (bx, by, bz) = the tile's position in that problem's grid.
template <int T>
__device__ __forceinline__ void gemm_tile(const GemmArgs &G, const int MODE, float *__restrict__ sA, float *__restrict__ sB, const int bx,
                                          const int by, const int bz) {
  constexpr int BM = GemmTile<T>::BM, LDT = GemmTile<T>::LDT, PER_T = GemmTile<T>::PER_T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int zb = bz / G.n_split, sp = bz - zb * G.n_split;
  const int m0 = by * BM, n0 = bx * BM;
  const float *A = G.A + (long long)zb * G.zA, *B = G.B + (long long)zb * G.zB;
  const int k_begin = sp * G.k_chunk;
  const int k_end = (k_begin + G.k_chunk < G.K) ? k_begin + G.k_chunk : G.K;
  const bool a_kfast = G.sak == 1, b_kfast = G.sbk == 1 && G.sbn != 1;
  // Element e = tid + 256 i of an operand tile: (fast, slow) index = (e % 64, e / 64) when k is the contiguous one, (e % BM, e / BM)
  // otherwise — 256 is a multiple of both, so the fast index is the same for every i and the slow one advances by a constant: the
  // global offset, the LDS slot and the bounds of element i are all affine in i (a handful of integer operations per load; forming
  // every address from scratch cost more VALU time per chunk than its MFMAs).
  constexpr int SLOW_K = 256 / BK, SLOW_R = 256 / BM;     // slow-index step per i
  //   A
  const int a_r0 = a_kfast ? tid / BK : tid % BM, a_k0 = a_kfast ? tid % BK : tid / BM;   // tile row / k of element i = 0
  const int a_ri = a_kfast ? SLOW_K : 0, a_ki = a_kfast ? 0 : SLOW_R;                     // rows / k advanced per i
  const int a_rrem = G.M - (m0 + a_r0);                                                   // valid while i * a_ri < a_rrem
  const long long a_gstep = (long long)a_ri * G.sam + (long long)a_ki * G.sak;
  const float *a_ptr = A + (long long)(m0 + a_r0) * G.sam + (long long)a_k0 * G.sak;       // + k0 * sak per chunk
  const int a_ls0 = a_kfast ? a_r0 * LDK + a_k0 : a_k0 * LDT + a_r0, a_lstep = a_kfast ? SLOW_K * LDK : SLOW_R * LDT;
  const int a_one = G.ones_row ? (G.M - 1) - (m0 + a_r0) : -1;                             // element i is the all-ones row iff i * a_ri == a_one
  //   B
  const int b_c0 = b_kfast ? tid / BK : tid % BM, b_k0 = b_kfast ? tid % BK : tid / BM;
  const int b_ci = b_kfast ? SLOW_K : 0, b_ki = b_kfast ? 0 : SLOW_R;
  const int b_crem = G.N - (n0 + b_c0);
  const long long b_gstep = (long long)b_ci * G.sbn + (long long)b_ki * G.sbk;
  const float *b_ptr = B + (long long)(n0 + b_c0) * G.sbn + (long long)b_k0 * G.sbk;
  const int b_ls0 = b_kfast ? b_c0 * LDK + b_k0 : b_k0 * LDT + b_c0, b_lstep = b_kfast ? SLOW_K * LDK : SLOW_R * LDT;
  float ra[PER_T], rb[PER_T];
  auto fetch = [&](int k0) {
    const int krem = k_end - k0;                       // valid while k index < krem
    const float *pa = a_ptr + (long long)k0 * G.sak, *pb = b_ptr + (long long)k0 * G.sbk;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const bool oka = (i * a_ri < a_rrem) && (a_k0 + i * a_ki < krem);
      float v = 0.f;
      if (oka) v = (a_one >= 0 && i * a_ri == a_one) ? 1.0f : pa[0];
      ra[i] = v;
      const bool okb = (i * b_ci < b_crem) && (b_k0 + i * b_ki < krem);
      rb[i] = okb ? pb[0] : 0.f;
      pa += a_gstep;
      pb += b_gstep;
    }
  };
  f32x4 acc[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm = (wave >> 1) * 16 * T, wn = (wave & 1) * 16 * T;
  const int lr = lane & 15, lg = lane >> 4;
  // operand read addresses: element (row, k) of A at row * a_rs + k * a_ks, element (k, col) of B at col * b_cs + k * b_ks
  const int a_rs = a_kfast ? LDK : 1, a_ks = a_kfast ? 1 : LDT, b_cs = b_kfast ? LDK : 1, b_ks = b_kfast ? 1 : LDT;
  const int a_off = (wm + lr) * a_rs + lg * a_ks, b_off = (wn + lr) * b_cs + lg * b_ks;
  fetch(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      sA[a_ls0 + i * a_lstep] = ra[i];
      sB[b_ls0 + i * b_lstep] = rb[i];
    }
    __syncthreads();
    if (k0 + BK < k_end) fetch(k0 + BK);        // in flight during this chunk's MFMAs
#pragma unroll 4
    for (int ks = 0; ks < BK / 4; ++ks) {
      float a[T], b[T];
#pragma unroll
      for (int i = 0; i < T; ++i) {
        a[i] = sA[a_off + 16 * i * a_rs + 4 * ks * a_ks];
        b[i] = sB[b_off + 16 * i * b_cs + 4 * ks * b_ks];
      }
#pragma unroll
      for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // lane holds D[row = 4 * (lane >> 4) + r][col = lane & 15] of each 16 x 16 block
  float *C = G.C ? G.C + (G.n_split > 1 ? (long long)bz * G.zSplit : (long long)zb * G.zC) : nullptr;
  float *C2 = G.C2 ? G.C2 + (long long)zb * G.zC : nullptr;
  const float *bias = (MODE == 0 && G.bias) ? G.bias + (long long)zb * G.zBias : nullptr;
  const float *Zp = (MODE == 1 && G.Zprev) ? G.Zprev + (long long)zb * G.zZ : nullptr;
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const int gn = n0 + wn + 16 * j + lr;
      if (gn >= G.N) continue;
      const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm + 16 * i + 4 * lg + r;
        if (gm >= G.M) continue;
        float v = acc[i][j][r];
        if (MODE == 0) {
          v += bv;
          if (C) C[(long long)gm * G.ldc + gn] = v;
          if (C2) C2[(long long)gm * G.ldc + gn] = G.act >= 0 ? act_apply(v, G.act) : v;
        } else if (MODE == 1) {
          if (Zp) v *= act_grad(Zp[(long long)gm * G.ldz + gn], G.act);
          C[(long long)gm * G.ldc + gn] = v;
        } else {
          C[(long long)gm * G.ldc + gn] = v;
        }
      }
    }
}

template <int MODE, int T>
__global__ void __launch_bounds__(256) k_layered_gemm(GemmArgs G) {
  __shared__ float sA[GemmTile<T>::TILE_F], sB[GemmTile<T>::TILE_F];
  gemm_tile<T>(G, MODE, sA, sB, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent problems of one dependency level as ONE launch (round 4): the layered step is launch-bound below ~256-wide
// layers (one Dense layer of one pass per launch: 46 launches per SAC update), and the passes of a level — policy(obs) / policy(next
// obs); the critics on (s, a), (s, a~pi), the target critics on (s', a'); a layer's weight gradient and the input gradient below it —
// depend on the previous level only.  A workgroup finds its problem from the tile prefix sums (uniform scalar work) and runs that
// problem's tile; the epilogue kind is a run-time value here.
constexpr int GROUP_MAX = 6;
struct GemmGroupArgs {
  int n;
  int start[GROUP_MAX + 1];          // first workgroup of problem p; start[n] = grid size
  int mode[GROUP_MAX];
  GemmArgs g[GROUP_MAX];
};

template <int T>
__global__ void __launch_bounds__(256) k_layered_group(const GemmGroupArgs GG) {
  __shared__ float sA[GemmTile<T>::TILE_F], sB[GemmTile<T>::TILE_F];
  constexpr int BM = GemmTile<T>::BM;
  int p = 0;
#pragma unroll
  for (int q = 1; q < GROUP_MAX; ++q)
    if (q < GG.n && (int)blockIdx.x >= GG.start[q]) p = q;
  p = __builtin_amdgcn_readfirstlane(p);
  const GemmArgs &G = GG.g[p];
  const int local = (int)blockIdx.x - GG.start[p];
  const int tx = (G.N + BM - 1) / BM, ty = (G.M + BM - 1) / BM;
  const int bx = local % tx, r = local / tx;
  gemm_tile<T>(G, GG.mode[p], sA, sB, bx, r % ty, r / ty);
}

__global__ void __launch_bounds__(256) k_layered_split_sum(const float *part, long long stride, int n_split, float *out, long long out_stride,
                                                           long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int z = blockIdx.y;
  out[(long long)z * out_stride + i] = slab_sum<8>(part + (long long)z * n_split * stride, stride, n_split, i);
}
}  // namespace

static int gemm_prepare(int mode, const GemmArgs &G, GemmArgs *H, long long *wg64) {
  MBPO_REQUIRE(G.A && G.B && (G.C || (mode == 0 && G.C2)), MBPO_ERR_ARG, "layered_gemm: null operand");
  MBPO_REQUIRE(G.M > 0 && G.N > 0 && G.K > 0 && G.nz >= 1 && G.n_split >= 1, MBPO_ERR_ARG, "layered_gemm: bad shape %d x %d x %d", G.M, G.N, G.K);
  MBPO_REQUIRE(G.n_split == 1 || (G.k_chunk > 0 && G.k_chunk % BK == 0), MBPO_ERR_ARG, "layered_gemm: k_chunk must be a multiple of %d", BK);
  *H = G;
  if (H->n_split == 1) H->k_chunk = H->K;
  *wg64 = (long long)((G.N + 63) / 64) * ((G.M + 63) / 64) * G.nz * G.n_split;
  return MBPO_OK;
}

// 32 x 32 tiles until the 64 x 64 grid alone would hold every CU many times over: measured on PPO's C3 minibatch with a 256x5 value
// net (M = 20 992 rows: 328 x 4 tiles of 64) 1404 us per minibatch step with 64-tiles, 1298 with 32-tiles; SAC 256x3 at B = 4096
// 883 -> 829 us (scripts/layered_ppo_timing.py, layered_timing.py; MBPO_LAYERED_T2_MIN overrides)
static int tile_factor(long long wg64) {
  static const long long t2_min = getenv("MBPO_LAYERED_T2_MIN") ? atoll(getenv("MBPO_LAYERED_T2_MIN")) : 4096;
  return wg64 >= t2_min ? 2 : 1;
}

int layered_gemm(int mode, const GemmArgs &G, hipStream_t st) {
  GemmArgs H;
  long long wg64;
  int rc = gemm_prepare(mode, G, &H, &wg64);
  if (rc != MBPO_OK) return rc;
  const int T = tile_factor(wg64);
  const int bm = 32 * T;
  const dim3 grid((unsigned)((G.N + bm - 1) / bm), (unsigned)((G.M + bm - 1) / bm), (unsigned)(G.nz * G.n_split));
#define LG(MODE_, T_) hipLaunchKernelGGL((k_layered_gemm<MODE_, T_>), grid, dim3(256), 0, st, H)
  if (T == 2) {
    if (mode == 0) LG(0, 2); else if (mode == 1) LG(1, 2); else LG(2, 2);
  } else {
    if (mode == 0) LG(0, 1); else if (mode == 1) LG(1, 1); else LG(2, 1);
  }
#undef LG
  MBPO_CHECK_LAUNCH("layered_gemm");
  return MBPO_OK;
}

// The problems of one dependency level, issued as one launch (k_layered_group); one problem alone takes the plain kernel.
// MBPO_LAYERED_GROUP=0: every problem its own launch (the round-3 behaviour, for A/B timing).
namespace {
struct GemmBatch {
  GemmGroupArgs a;
  long long wg64;
  GemmBatch() : wg64(0) { a.n = 0; }
  int add(int mode, const GemmArgs &G) {
    MBPO_REQUIRE(a.n < GROUP_MAX, MBPO_ERR_ARG, "layered: more than %d problems in one level", GROUP_MAX);
    long long w;
    int rc = gemm_prepare(mode, G, &a.g[a.n], &w);
    if (rc != MBPO_OK) return rc;
    a.mode[a.n++] = mode;
    wg64 += w;
    return MBPO_OK;
  }
  int flush(hipStream_t st) {
    static const bool grouped = !(getenv("MBPO_LAYERED_GROUP") && atoi(getenv("MBPO_LAYERED_GROUP")) == 0);
    const int n = a.n;
    a.n = 0;
    const long long w = wg64;
    wg64 = 0;
    if (n == 0) return MBPO_OK;
    if (n == 1 || !grouped) {
      for (int p = 0; p < n; ++p) {
        int rc = layered_gemm(a.mode[p], a.g[p], st);
        if (rc != MBPO_OK) return rc;
      }
      return MBPO_OK;
    }
    const int T = tile_factor(w), bm = 32 * T;
    long long total = 0;
    for (int p = 0; p < n; ++p) {
      const GemmArgs &G = a.g[p];
      a.start[p] = (int)total;
      total += (long long)((G.N + bm - 1) / bm) * ((G.M + bm - 1) / bm) * G.nz * G.n_split;
    }
    MBPO_REQUIRE(total < (1LL << 31), MBPO_ERR_ARG, "layered: level too large (%lld tiles)", total);
    for (int p = n; p <= GROUP_MAX; ++p) a.start[p] = (int)total;
    a.n = n;
    if (T == 2) hipLaunchKernelGGL((k_layered_group<2>), dim3((unsigned)total), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_layered_group<1>), dim3((unsigned)total), dim3(256), 0, st, a);
    a.n = 0;
    MBPO_CHECK_LAUNCH("layered_group");
    return MBPO_OK;
  }
};
}  // namespace

int layered_split_sum(const float *part, long long stride, int n_split, int nz, float *out, long long out_stride, long long n, hipStream_t st) {
  hipLaunchKernelGGL(k_layered_split_sum, dim3((unsigned)((n + 255) / 256), (unsigned)nz), dim3(256), 0, st, part, stride, n_split, out,
                     out_stride, n);
  MBPO_CHECK_LAUNCH("layered_split_sum");
  return MBPO_OK;
}

LayeredNet layered_net(const MlpDev &m, const float *params, long long net_stride, int nz) {
  LayeredNet n;
  n.params = params; n.net_stride = net_stride; n.nz = nz; n.L = m.n_layers; n.act = m.act;
  for (int l = 0; l <= m.n_layers; ++l) n.dims[l] = m.dims[l];
  for (int l = 0; l < m.n_layers; ++l) { n.w_off[l] = m.w_off[l]; n.b_off[l] = m.b_off[l]; }
  return n;
}

int layered_max_hidden(const LayeredNet &n) {
  int h = 1;
  for (int l = 0; l <= n.L; ++l) h = n.dims[l] > h ? n.dims[l] : h;
  return h;
}

// weight gradients reduce over the minibatch rows: beyond 1024 rows the range is split over workgroups (a 256 x 256 weight is only
// 16 output tiles) and the partials are added in a fixed order
static void split_of(int rows, int *n_split, int *k_chunk) {
  int s = (rows + 1023) / 1024;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  int c = (rows + s - 1) / s;
  c = (c + BK - 1) / BK * BK;
  *n_split = (rows + c - 1) / c;
  *k_chunk = c;
}

long long layered_part_floats(const LayeredNet &n, int rows) {
  int s, c;
  split_of(rows, &s, &c);
  if (s == 1) return 0;
  long long mx = 0;
  for (int l = 0; l < n.L; ++l) {
    const long long v = (long long)(n.dims[l] + 1) * n.dims[l + 1];
    mx = v > mx ? v : mx;
  }
  return mx * s * n.nz;
}

static GemmArgs forward_args(const LayeredFwd &p, int l, const float *in, long long zin) {
  const LayeredNet &n = p.net;
  const int K = n.dims[l], N = n.dims[l + 1];
  const bool last = l == n.L - 1;
  GemmArgs G = {};
  G.A = in; G.sam = K; G.sak = 1; G.zA = zin;
  G.B = n.params + n.w_off[l]; G.sbk = N; G.sbn = 1; G.zB = n.net_stride;
  G.bias = n.params + n.b_off[l]; G.zBias = n.net_stride;
  G.M = p.rows; G.N = N; G.K = K; G.nz = n.nz; G.n_split = 1;
  G.ldc = N; G.zC = (long long)p.rows * N;
  if (last) {
    G.C = p.y; G.C2 = nullptr; G.act = -1;
  } else {
    G.C = p.Z ? p.Z[l + 1] : nullptr; G.C2 = p.H[l + 1]; G.act = n.act;
  }
  return G;
}

int layered_forward_multi(const LayeredFwd *passes, int n_pass, hipStream_t st) {
  MBPO_REQUIRE(n_pass >= 1 && n_pass <= GROUP_MAX, MBPO_ERR_ARG, "layered_forward_multi: 1..%d passes", GROUP_MAX);
  int maxL = 0;
  for (int i = 0; i < n_pass; ++i) maxL = passes[i].net.L > maxL ? passes[i].net.L : maxL;
  GemmBatch batch;
  for (int l = 0; l < maxL; ++l) {
    for (int i = 0; i < n_pass; ++i) {
      const LayeredFwd &p = passes[i];
      if (l >= p.net.L) continue;
      const float *in = l == 0 ? p.x : p.H[l];
      const long long zin = l == 0 ? p.zx : (long long)p.rows * p.net.dims[l];
      int rc = batch.add(0, forward_args(p, l, in, zin));
      if (rc != MBPO_OK) return rc;
    }
    int rc = batch.flush(st);
    if (rc != MBPO_OK) return rc;
  }
  return MBPO_OK;
}

int layered_forward(const LayeredNet &n, const float *x, long long zx, int rows, float *const *Z, float *const *H, float *y, hipStream_t st) {
  const LayeredFwd p = {n, x, zx, rows, Z, H, y};
  return layered_forward_multi(&p, 1, st);
}

// Level s of the backward passes = layer l = L-1-s of each: its weight gradient and the input gradient below it both read dz_l only.
int layered_backward_multi(const LayeredBwd *passes, int n_pass, hipStream_t st) {
  MBPO_REQUIRE(n_pass >= 1 && 2 * n_pass <= GROUP_MAX, MBPO_ERR_ARG, "layered_backward_multi: 1..%d passes", GROUP_MAX / 2);
  int maxL = 0;
  const float *dz[GROUP_MAX];
  int flip[GROUP_MAX];
  for (int i = 0; i < n_pass; ++i) {
    maxL = passes[i].net.L > maxL ? passes[i].net.L : maxL;
    dz[i] = passes[i].dy;
    flip[i] = 0;
  }
  GemmBatch batch;
  for (int s = 0; s < maxL; ++s) {
    bool any_split = false;
    for (int i = 0; i < n_pass; ++i) {
      const LayeredBwd &p = passes[i];
      const LayeredNet &n = p.net;
      const int l = n.L - 1 - s;
      if (l < 0) continue;
      const int K = n.dims[l], N = n.dims[l + 1], rows = p.rows;
      int n_split, k_chunk;
      split_of(rows, &n_split, &k_chunk);
      if (p.dw) {
        const float *in = l == 0 ? p.x : p.H[l];
        const long long zin = l == 0 ? p.zx : (long long)rows * K;
        // [dW; db](k, n) = sum_rows [in, 1](row, k) * dz(row, n): the GEMM's m = k (K + 1 rows, the last all ones), its k = minibatch rows
        GemmArgs G = {};
        G.A = in; G.sam = 1; G.sak = K; G.zA = zin; G.ones_row = 1;
        G.B = dz[i]; G.sbk = N; G.sbn = 1; G.zB = (long long)rows * N;
        G.M = K + 1; G.N = N; G.K = rows; G.nz = n.nz;
        G.ldc = N;
        if (n_split == 1) {
          G.n_split = 1; G.C = p.dw + n.w_off[l]; G.zC = p.dw_stride;
        } else {
          G.n_split = n_split; G.k_chunk = k_chunk; G.C = p.part; G.zSplit = (long long)(K + 1) * N;
          any_split = true;
        }
        int rc = batch.add(2, G);
        if (rc != MBPO_OK) return rc;
      }
      if (l == 0 && !p.dx) continue;
      // d in(row, k) = sum_n dz(row, n) * W(k, n), times act'(Z[l]) for a hidden layer's input
      GemmArgs G = {};
      G.A = dz[i]; G.sam = N; G.sak = 1; G.zA = (long long)rows * N;
      G.B = n.params + n.w_off[l]; G.sbk = 1; G.sbn = N; G.zB = n.net_stride;
      G.M = rows; G.N = K; G.K = N; G.nz = n.nz; G.n_split = 1;
      G.ldc = K; G.zC = (long long)rows * K;
      float *const out = flip[i] ? p.tmp1 : p.tmp0;
      if (l == 0) {
        G.C = p.dx; G.Zprev = nullptr; G.act = -1;
      } else {
        G.C = out; G.Zprev = p.Z[l]; G.ldz = K; G.zZ = (long long)rows * K; G.act = n.act;
      }
      int rc = batch.add(1, G);
      if (rc != MBPO_OK) return rc;
      if (l > 0) { dz[i] = out; flip[i] ^= 1; }
    }
    int rc = batch.flush(st);
    if (rc != MBPO_OK) return rc;
    if (any_split) {
      // beyond 1024 rows a weight gradient's row range is split over workgroups: add this level's partials (fixed order) before the
      // next level reuses the partial buffers
      for (int i = 0; i < n_pass; ++i) {
        const LayeredBwd &p = passes[i];
        const int l = p.net.L - 1 - s;
        if (l < 0 || !p.dw) continue;
        int n_split, k_chunk;
        split_of(p.rows, &n_split, &k_chunk);
        if (n_split == 1) continue;
        const long long sz = (long long)(p.net.dims[l] + 1) * p.net.dims[l + 1];
        rc = layered_split_sum(p.part, sz, n_split, p.net.nz, p.dw + p.net.w_off[l], p.dw_stride, sz, st);
        if (rc != MBPO_OK) return rc;
      }
    }
  }
  return MBPO_OK;
}

int layered_backward(const LayeredNet &n, const float *x, long long zx, int rows, float *const *Z, float *const *H, const float *dy,
                     float *dw, long long dw_stride, float *dx, float *tmp0, float *tmp1, float *part, hipStream_t st) {
  const LayeredBwd p = {n, x, zx, rows, Z, H, dy, dw, dw_stride, dx, tmp0, tmp1, part};
  return layered_backward_multi(&p, 1, st);
}

// ------------------------------------------------------------------------------------------------ C-ABI: any-shape MLP forward / VJP
// The network halves of an autograd graph for shapes the fused wave-chain kernels are not built for (mbpo_mlp_vjp: hidden width 64
// only): BPTT through actor (128, 128) / critic (256, 256) networks (bptt_optimizer.py:183-186 accepts any feature tuple).
struct MlpLayeredPlan {
  MlpDev m;
  LayeredNet net;
  long long off_z[MBPO_MAX_LAYERS + 1], off_h[MBPO_MAX_LAYERS + 1], off_tmp0, off_tmp1, off_part, off_y, total;
};

static int mlp_layered_plan(const mbpo_mlp_desc *mlp, int64_t n, MlpLayeredPlan *pl) {
  MBPO_REQUIRE(mlp, MBPO_ERR_ARG, "mlp_layered: null descriptor");
  MBPO_REQUIRE(n >= 1 && n < (1LL << 24), MBPO_ERR_ARG, "mlp_layered: n must be in [1, 2^24)");
  mbpo_mlp_desc d = *mlp;
  if (!d.params) d.params = (const float *)16;      // size queries
  int rc = mbpo_make_mlp_dev(&d, &pl->m, "mlp_layered");
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(pl->m.n_layers >= 2, MBPO_ERR_UNSUPPORTED, "mlp_layered: the network needs at least one hidden layer");
  pl->net = layered_net(pl->m, mlp->params, mlp->n_nets > 1 ? mlp->net_stride : pl->m.n_params, mlp->n_nets);
  const long long nz = mlp->n_nets;
  long long off = 0;
  for (int l = 1; l < pl->m.n_layers; ++l) {
    pl->off_z[l] = off; off += nz * n * pl->m.dims[l];
    pl->off_h[l] = off; off += nz * n * pl->m.dims[l];
  }
  const long long mh = layered_max_hidden(pl->net);
  pl->off_tmp0 = off; off += nz * n * mh;
  pl->off_tmp1 = off; off += nz * n * mh;
  pl->off_y = off; off += nz * n * pl->m.dims[pl->m.n_layers];
  pl->off_part = off; off += layered_part_floats(pl->net, (int)n);
  pl->total = off;
  return MBPO_OK;
}

extern "C" int64_t mbpo_mlp_layered_workspace_floats(const mbpo_mlp_desc *mlp, int64_t n) {
  MlpLayeredPlan pl;
  int rc = mlp_layered_plan(mlp, n, &pl);
  if (rc != MBPO_OK) return rc;
  return pl.total;
}

extern "C" int mbpo_mlp_layered_vjp(const mbpo_mlp_desc *mlp, const float *x, int64_t n, const float *dy, float *y, float *dx, float *dw,
                                    float *workspace, void *stream) {
  MlpLayeredPlan pl;
  int rc = mlp_layered_plan(mlp, n, &pl);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(mlp->params && x && workspace, MBPO_ERR_ARG, "mlp_layered_vjp: null params / x / workspace");
  MBPO_REQUIRE(y || dy, MBPO_ERR_ARG, "mlp_layered_vjp: nothing to compute (neither y nor dy given)");
  MBPO_REQUIRE(!dy || dx || dw, MBPO_ERR_ARG, "mlp_layered_vjp: dy given but neither dx nor dw requested");
  MBPO_REQUIRE(mlp->n_nets == 1 || mlp->net_stride == pl.m.n_params || !dw, MBPO_ERR_UNSUPPORTED,
               "mlp_layered_vjp: dw is laid out [net][params]: net_stride must equal the parameters per net (%d)", pl.m.n_params);
  hipStream_t st = (hipStream_t)stream;
  float *Z[MBPO_MAX_LAYERS + 1], *H[MBPO_MAX_LAYERS + 1];
  for (int l = 0; l <= MBPO_MAX_LAYERS; ++l) Z[l] = H[l] = nullptr;
  for (int l = 1; l < pl.m.n_layers; ++l) {
    Z[l] = dy ? workspace + pl.off_z[l] : nullptr;      // pre-activations are only needed by the backward pass
    H[l] = workspace + pl.off_h[l];
  }
  float *yy = y ? y : workspace + pl.off_y;
  rc = layered_forward(pl.net, x, 0, (int)n, Z, H, yy, st);
  if (rc != MBPO_OK) return rc;
  if (dy) {
    rc = layered_backward(pl.net, x, 0, (int)n, Z, H, dy, dw, pl.m.n_params, dx, workspace + pl.off_tmp0, workspace + pl.off_tmp1,
                          workspace + pl.off_part, st);
    if (rc != MBPO_OK) return rc;
  }
  MBPO_CHECK_LAUNCH("mlp_layered_vjp");
  return MBPO_OK;
}

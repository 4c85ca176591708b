// mlp_vjp.hip — vector-Jacobian product of one or two 64-wide MLPs on a batch, for BPTT through a USER-DEFINED System
// (bptt_optimizer.py:327-378 with a `System.step` that exists only as the user's differentiable torch code).
//
// On that path the horizon loop lives on the host (the user's step runs between the kernels), so the networks cannot be walked
// inside k_bptt_actor; what remains device work per horizon step is
//   actor        logits = MLP(normalise(stop_gradient(x_t)))            -> d loss / d actor parameters     (weights only)
//   twin target V(normalise(x'_t))                                      -> d min(v1, v2) / d x'_t            (input only)
// i.e. "given dL/dy, give me dL/dW and / or dL/dx".  One launch: per 16-row tile the forward is RECOMPUTED with z and h kept in
// LDS (as k_critic_fwd_bwd and k_bptt_actor do: a stash through HBM would cost more than the extra forward), then a dgrad chain
// and a wgrad chain per net run side by side on the phase runners (chain_run.hpp); weight gradients accumulate in one slab per
// workgroup, summed by a second launch in a fixed order (bit-reproducible).  fp32 MFMA throughout.
#include "common.hpp"
#include "chain_run.hpp"

struct VjpArgs {
  MlpDev net;
  NetShape sh;
  int X, N_out, n_nets, want_dw;
  const float *x, *mean, *std, *dy;
  float *y, *dx, *slabs;
  long long n;
  int ld_x, ld_h, ld_y, LH;
};

// 4 chain slots x SP waves.  Slots 0/1: net 0/1 forward, then dgrad (input gradient to LDS when dx is wanted);
// slots 2/3: net 0/1 wgrad beside the dgrad (idle when no weight gradient is wanted or the net does not exist).
template <int H, int SP, bool WIDE>
__global__ void __launch_bounds__(256 * SP) k_mlp_vjp(VjpArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid_ = threadIdx.x, nthreads = 256 * SP;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int chain = wave / SP, sub = wave % SP;
  const int X = A.X, NO = A.N_out, ld_x = A.ld_x, ld_h = A.ld_h, ld_y = A.ld_y, LH = A.LH;
  const int T = 16 * ld_h;
  float *s_x = smem;                          // [16][ld_x]     (normalised) inputs
  float *s_dx = s_x + 16 * ld_x;              // [2][16][ld_x]  input gradients
  float *s_y = s_dx + 2 * 16 * ld_x;          // [2][16][ld_y]  outputs
  float *s_dy = s_y + 2 * 16 * ld_y;          // [2][16][ld_y]  output gradients
  float *s_st = s_dy + 2 * 16 * ld_y;         // 4*LH tiles: z1 h1 z2 h2
  float *s_pp = s_st + 4 * LH * T;            // 4 delta tiles
  const int net = chain & 1;
  const bool have = net < A.n_nets;
  float *zb = s_st + (2 * net) * LH * T, *hb = zb + LH * T;
  const float *params = A.net.params + (long long)net * A.net.net_stride;
  const int CL = A.net.n_layers;
  float *slab = A.slabs ? A.slabs + (long long)blockIdx.x * A.n_nets * A.net.n_params + (long long)net * A.net.n_params : nullptr;
  bool first = true;
  const long long n_tiles = (A.n + 15) >> 4;
#pragma nounroll
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, first = false) {
    const int tid = opaque(tid_), lane = tid & 63;
    const long long j0 = tile * 16;
    WSet<HT, SP> R;
    if (chain < 2 && have) chain_fwd_prefetch<HT, SP, WIDE>(R, A.sh, params, sub, lane);
    for (int idx = tid; idx < 16 * X; idx += nthreads) {
      const int r = idx & 15, c = idx >> 4;
      const long long j = j0 + r;
      float o = 0.f;
      if (j < A.n) {
        o = A.x[j * X + c];
        if (A.mean) o = (o - A.mean[c]) / A.std[c];
      }
      s_x[r * ld_x + c] = o;
    }
    for (int idx = tid; idx < A.n_nets * 16 * NO; idx += nthreads) {
      const int k = idx / (16 * NO), rem = idx - k * 16 * NO, r = rem & 15, c = rem >> 4;
      const long long j = j0 + r;
      s_dy[(k * 16 + r) * ld_y + c] = (j < A.n) ? A.dy[((long long)k * A.n + j) * NO + c] : 0.f;
    }
    __syncthreads();
    if (chain < 2 && have)
      chain_fwd_run<HT, SP, WIDE>(A.sh, params, s_x, ld_x, nullptr, nullptr, zb, hb, s_y + net * 16 * ld_y, ld_y, ld_h, CL, sub, lane, R);
    else chain_idle_run(CL);
    if (chain < 2 && have) chain_dgrad_prefetch<HT, SP, WIDE>(R, A.sh, params, sub, lane);
    if (A.y) {          // the recomputed outputs, for callers that want them from this launch (visible: the runner ended in a barrier)
      for (int idx = tid; idx < A.n_nets * 16 * NO; idx += nthreads) {
        const int k = idx / (16 * NO), rem = idx - k * 16 * NO, r = rem & 15, c = rem >> 4;
        const long long j = j0 + r;
        if (j < A.n) A.y[((long long)k * A.n + j) * NO + c] = s_y[(k * 16 + r) * ld_y + c];
      }
    }
    {
      float *d0 = s_pp + (2 * net) * T, *d1 = d0 + T;
      if (chain < 2 && have)
        chain_dgrad_run<HT, SP, WIDE>(A.sh, params, s_dy + net * 16 * ld_y, ld_y, zb, d0, d1, A.dx ? s_dx + net * 16 * ld_x : nullptr, ld_x,
                                      ld_h, CL, sub, lane, R);
      else if (chain >= 2 && have && A.want_dw)
        chain_wgrad_run<HT, SP, WIDE>(A.sh, s_x, ld_x, hb, s_dy + net * 16 * ld_y, ld_y, d0, d1, slab, !first, ld_h, CL, sub, lane);
      else chain_idle_run(CL);
    }
    if (A.dx) {         // d/d(raw input) = d/d(normalised input) / std
      for (int idx = tid; idx < A.n_nets * 16 * X; idx += nthreads) {
        const int k = idx / (16 * X), rem = idx - k * 16 * X, r = rem & 15, c = rem >> 4;
        const long long j = j0 + r;
        if (j < A.n) {
          float g = s_dx[(k * 16 + r) * ld_x + c];
          if (A.std) g /= A.std[c];
          A.dx[((long long)k * A.n + j) * X + c] = g;
        }
      }
      __syncthreads();   // s_dx is rewritten by the next tile's dgrad only after several barriers, s_x at once: keep the order explicit
    }
  }
}

__global__ void __launch_bounds__(256) k_vjp_reduce(const float *slabs, int n_slabs, int NW, float *dw) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const float g = slab_sum_wg64(slabs, NW, n_slabs, i, i < NW);
  if (threadIdx.x < 64 && i < NW) dw[i] = g;
}

static int vjp_num_cus() {
  static int n = 0;
  if (!n) {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

struct VjpPlan {
  MlpDev net;
  int n_slabs, ld_x, ld_h, ld_y, LH;
  size_t lds;
  long long total;
};

static int vjp_plan(const mbpo_mlp_desc *mlp, long long n, VjpPlan *pl, bool need_ptrs) {
  MBPO_REQUIRE(mlp, MBPO_ERR_ARG, "mlp_vjp: null mlp descriptor");
  MBPO_REQUIRE(n > 0, MBPO_ERR_ARG, "mlp_vjp: n must be positive");
  mbpo_mlp_desc md = *mlp;
  if (!need_ptrs) md.params = (const float *)16;
  int rc = mbpo_make_mlp_dev(&md, &pl->net, "mlp_vjp");
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(md.n_nets >= 1 && md.n_nets <= 2, MBPO_ERR_UNSUPPORTED, "mlp_vjp: 1 or 2 nets per launch (got %d)", md.n_nets);
  MBPO_REQUIRE(md.n_layers >= 2, MBPO_ERR_UNSUPPORTED, "mlp_vjp: need at least one hidden layer");
  for (int l = 1; l < md.n_layers; ++l)
    MBPO_REQUIRE(md.dims[l] == 64, MBPO_ERR_UNSUPPORTED, "mlp_vjp: hidden layers must all be 64 wide (dims[%d] = %d); zero-pad narrower nets",
                 l, md.dims[l]);
  MBPO_REQUIRE(md.dims[0] <= 32 && md.dims[md.n_layers] <= 32, MBPO_ERR_UNSUPPORTED,
               "mlp_vjp: input width %d / output width %d beyond the register-image shapes (<= 32)", md.dims[0], md.dims[md.n_layers]);
  pl->LH = md.n_layers - 1;
  pl->ld_x = ((md.dims[0] + 3) & ~3) + 4;
  pl->ld_h = 68;
  pl->ld_y = ((md.dims[md.n_layers] + 3) & ~3) + 4;
  pl->lds = sizeof(float) * (3ull * 16 * pl->ld_x + 4ull * 16 * pl->ld_y + (size_t)(4 * pl->LH + 4) * 16 * pl->ld_h);
  const long long tiles = (n + 15) / 16, cap = 1LL * vjp_num_cus();   // one 1024-thread workgroup fills a CU: one slab per CU
  pl->n_slabs = (int)(tiles < cap ? tiles : cap);
  pl->total = (long long)pl->n_slabs * md.n_nets * pl->net.n_params;
  return MBPO_OK;
}

extern "C" int64_t mbpo_mlp_vjp_workspace_floats(const mbpo_mlp_desc *mlp, int64_t n) {
  VjpPlan pl;
  int rc = vjp_plan(mlp, n, &pl, false);
  if (rc != MBPO_OK) return rc;
  return pl.total;
}

extern "C" int mbpo_mlp_vjp(const mbpo_mlp_desc *mlp, const float *x, int64_t n, const float *norm_mean, const float *norm_std,
                            const float *dy, float *y, float *dx, float *dw, float *workspace, void *stream) {
  VjpPlan pl;
  int rc = vjp_plan(mlp, n, &pl, true);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(x && dy, MBPO_ERR_ARG, "mlp_vjp: null x / dy");
  MBPO_REQUIRE((norm_mean == nullptr) == (norm_std == nullptr), MBPO_ERR_ARG, "mlp_vjp: norm_mean / norm_std mismatch");
  MBPO_REQUIRE(dx || dw, MBPO_ERR_ARG, "mlp_vjp: neither dx nor dw requested");
  MBPO_REQUIRE(!dw || workspace, MBPO_ERR_ARG, "mlp_vjp: dw needs a workspace of mbpo_mlp_vjp_workspace_floats() floats");
  MBPO_REQUIRE(mlp->n_nets == 1 || mlp->net_stride == pl.net.n_params || !dw, MBPO_ERR_UNSUPPORTED,
               "mlp_vjp: dw is laid out [net][params]: net_stride must equal the parameters per net (%d)", pl.net.n_params);
  VjpArgs A;
  A.net = pl.net;
  A.sh = NetShape{mlp->dims[0], mlp->n_layers, mlp->dims[mlp->n_layers], mlp->activation};
  A.X = mlp->dims[0]; A.N_out = mlp->dims[mlp->n_layers]; A.n_nets = mlp->n_nets; A.want_dw = dw ? 1 : 0;
  A.x = x; A.mean = norm_mean; A.std = norm_std; A.dy = dy; A.y = y; A.dx = dx; A.slabs = dw ? workspace : nullptr;
  A.n = n; A.ld_x = pl.ld_x; A.ld_h = pl.ld_h; A.ld_y = pl.ld_y; A.LH = pl.LH;
  const bool wide = net_is_wide(A.sh);
  rc = wide ? mbpo_ensure_lds<k_mlp_vjp<64, 4, true>>(pl.lds, "mlp_vjp") : mbpo_ensure_lds<k_mlp_vjp<64, 4, false>>(pl.lds, "mlp_vjp");
  if (rc != MBPO_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (wide) hipLaunchKernelGGL((k_mlp_vjp<64, 4, true>), dim3(pl.n_slabs), dim3(1024), pl.lds, st, A);
  else hipLaunchKernelGGL((k_mlp_vjp<64, 4, false>), dim3(pl.n_slabs), dim3(1024), pl.lds, st, A);
  if (dw) {
    const int NW = mlp->n_nets * pl.net.n_params;
    hipLaunchKernelGGL(k_vjp_reduce, dim3((NW + 63) / 64), dim3(256), 0, st, (const float *)workspace, pl.n_slabs, NW, dw);
  }
  MBPO_CHECK_LAUNCH("mlp_vjp");
  return MBPO_OK;
}

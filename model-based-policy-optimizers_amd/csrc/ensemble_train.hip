// ensemble_train.hip — N3: the model-learning step that feeds the rollout path — Gaussian negative
// log-likelihood gradients of every ensemble member on its own (bootstrapped) minibatch of true transitions.
// Not in the reference (its model would come from the external `bsm` package, setup.py:22): semantics are this build's,
// chosen to be the exact inverse of what the rollout kernel consumes (EnsembleDynamics.next_state):
//     (mu, raw) = MLP_e([x, u]);  mean = mu (+ x if predict_delta);  sigma = softplus(raw) + min_std
//     loss_e = mean_b sum_d [ 0.5 ((x'_d - mean_d) / sigma_d)^2 + log sigma_d ]          (+ const)
// One workgroup = (member, 16-row tile): forward chain with stored activations, elementwise output gradient, then a dgrad
// and a wgrad chain side by side (chain_run.hpp); a workgroup walks tiles and accumulates into its slab; fixed-order reduce.
// fp32 MFMA; algorithmic work per (member, sample): 3 * 2M FLOP, HBM 4*(2x+u) B gathered.
#include "common.hpp"
#include "chain_run.hpp"

struct EnsTrainArgs {
  NetShape sh;
  const float *params;
  long long net_stride;
  int n_params, E, X, U, D, noff;
  const float *rows;
  const int *idx;
  long long batch;
  int predict_delta;
  float min_std;
  float *slabs, *extras;
  int n_slots, ld_xu, ld_h, ld_y, LH;
};

template <int SP, bool WIDE>
__global__ void __launch_bounds__(128 * SP) k_ens_nll_fwd_bwd(EnsTrainArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = 4;
  const int tid_ = threadIdx.x, nthreads = 128 * SP;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int chain = wave / SP, sub = wave % SP;    // chain 0: forward, then dgrad; chain 1: wgrad
  const int e = blockIdx.x / A.n_slots, slot = blockIdx.x - e * A.n_slots;
  const int X = A.X, U = A.U, ld_xu = A.ld_xu, ld_h = A.ld_h, ld_y = A.ld_y, LH = A.LH;
  const int T = 16 * ld_h;
  float *s_xu = smem;                       // [16][ld_xu]  [x, u]
  float *s_t = s_xu + 16 * ld_xu;           // [16][ld_y]   regression target (first X columns)
  float *s_y = s_t + 16 * ld_y;             // [16][ld_y]   (mu, raw)
  float *s_dy = s_y + 16 * ld_y;            // [16][ld_y]
  float *s_st = s_dy + 16 * ld_y;           // 2*LH tiles: z, h
  float *s_pp = s_st + 2 * LH * T;          // 2 delta tiles
  float *s_ls = s_pp + 2 * T;               // [16] loss partials
  const float *params = A.params + (long long)e * A.net_stride;
  const int L = A.sh.L;
  const float invB = 1.0f / (float)A.batch;
  float *slab = A.slabs + ((long long)e * A.n_slots + slot) * A.n_params;
  const int *idx = A.idx + (long long)e * A.batch;
  float loss = 0.f;
  bool first = true;
  const long long n_tiles = (A.batch + 15) >> 4;
#pragma nounroll
  for (long long tile = slot; tile < n_tiles; tile += A.n_slots, first = false) {
    const int tid = opaque(tid_), lane = tid & 63;
    const long long j0 = tile * 16;
    WSet<HT, SP> R;
    if (chain == 0) chain_fwd_prefetch<HT, SP, WIDE>(R, A.sh, params, sub, lane);
    for (int i2 = tid; i2 < 16 * (X + U); i2 += nthreads) {
      const int r = i2 & 15, c = i2 >> 4;
      const long long j = j0 + r;
      s_xu[r * ld_xu + c] = (j < A.batch) ? A.rows[(long long)idx[j] * A.D + c] : 0.f;
    }
    for (int i2 = tid; i2 < 16 * X; i2 += nthreads) {
      const int r = i2 & 15, c = i2 >> 4;
      const long long j = j0 + r;
      float t = 0.f;
      if (j < A.batch) {
        const float *row = A.rows + (long long)idx[j] * A.D;
        t = row[A.noff + c] - (A.predict_delta ? row[c] : 0.f);
      }
      s_t[r * ld_y + c] = t;
    }
    __syncthreads();
    if (chain == 0) chain_fwd_run<HT, SP, WIDE>(A.sh, params, s_xu, ld_xu, nullptr, nullptr, s_st, s_st + LH * T, s_y, ld_y, ld_h, L, sub, lane, R);
    else chain_idle_run(L);
    if (chain == 0) chain_dgrad_prefetch<HT, SP, WIDE>(R, A.sh, params, sub, lane);
    // d loss / d(mu, raw) per element, loss partial per row
    for (int i2 = tid; i2 < 16 * X; i2 += nthreads) {
      const int r = i2 & 15, c = i2 >> 4;
      const bool ok = j0 + r < A.batch;
      const float mu = s_y[r * ld_y + c], raw = s_y[r * ld_y + X + c];
      const float sg = softplus_f(raw) + A.min_std;
      const float q = (s_t[r * ld_y + c] - mu) / sg;
      s_dy[r * ld_y + c] = ok ? -(q / sg) * invB : 0.f;
      s_dy[r * ld_y + X + c] = ok ? ((1.f - q * q) / sg) * sigmoid_f(raw) * invB : 0.f;
      s_t[r * ld_y + X + c] = ok ? 0.5f * q * q + logf(sg) : 0.f;      // per-element loss, summed below
    }
    __syncthreads();
    if (tid < 16) {
      float a = 0.f;
      for (int c = 0; c < X; ++c) a += s_t[tid * ld_y + X + c];
      s_ls[tid] = a;
    }
    if (chain == 0) chain_dgrad_run<HT, SP, WIDE>(A.sh, params, s_dy, ld_y, s_st, s_pp, s_pp + T, nullptr, ld_xu, ld_h, L, sub, lane, R);
    else chain_wgrad_run<HT, SP, WIDE>(A.sh, s_xu, ld_xu, s_st + LH * T, s_dy, ld_y, s_pp, s_pp + T, slab, !first, ld_h, L, sub, lane);
    if (tid == 0)
      for (int i = 0; i < 16; ++i) loss += s_ls[i];
    __syncthreads();
  }
  if (tid_ == 0) A.extras[(long long)e * A.n_slots + slot] = loss;
}

__global__ void __launch_bounds__(256) k_ens_reduce(const float *slabs, const float *extras, int n_slots, int n_params, long long batch,
                                                     float *grads, float *metrics) {
  const int e = blockIdx.y;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const float gsum = slab_sum_wg64(slabs + (long long)e * n_slots * n_params, n_params, n_slots, i, i < n_params);
  if (threadIdx.x < 64 && i < n_params) grads[(long long)e * n_params + i] = gsum;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float a = slab_sum<16>(extras + (long long)e * n_slots, 1, n_slots, 0);
    metrics[e] = a / (float)batch;
  }
}

static int ens_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

struct EnsPlan {
  MlpDev dyn;
  int n_slots, ld_xu, ld_h, ld_y, LH;
  size_t lds;
  long long total;
};

static int ens_plan(const mbpo_ens_train_desc *d, EnsPlan *pl, bool need_ptrs) {
  MBPO_REQUIRE(d, MBPO_ERR_ARG, "ens_nll: null descriptor");
  MBPO_REQUIRE(d->x_dim > 0 && d->u_dim > 0 && d->batch > 0, MBPO_ERR_ARG, "ens_nll: x_dim/u_dim/batch must be positive");
  mbpo_mlp_desc md = d->dynamics;
  if (!md.params) md.params = (const float *)16;
  int rc = mbpo_make_mlp_dev(&md, &pl->dyn, "ens_nll.dynamics");
  if (rc != MBPO_OK) return rc;
  const int X = d->x_dim, U = d->u_dim, L = pl->dyn.n_layers;
  MBPO_REQUIRE(pl->dyn.dims[0] == X + U && pl->dyn.dims[L] == 2 * X, MBPO_ERR_ARG, "ens_nll: dynamics must map [x+u] -> [2x] (mean, raw std)");
  MBPO_REQUIRE(L >= 2, MBPO_ERR_ARG, "ens_nll: the member networks need at least one hidden layer");
  for (int l = 1; l < L; ++l)
    MBPO_REQUIRE(pl->dyn.dims[l] == 64, MBPO_ERR_UNSUPPORTED, "ens_nll: hidden layers must all be 64 wide (got %d)", pl->dyn.dims[l]);
  MBPO_REQUIRE(d->row_len >= d->next_obs_off + X && d->next_obs_off >= X + U, MBPO_ERR_ARG, "ens_nll: bad row_len / next_obs_off");
  auto up4 = [](int v) { return (v + 3) & ~3; };
  pl->LH = L - 1;
  pl->ld_xu = up4(X + U) + 4;
  pl->ld_h = 68;
  pl->ld_y = up4(2 * X) + 4;
  pl->lds = sizeof(float) * (16ull * pl->ld_xu + 3ull * 16 * pl->ld_y + (size_t)(2 * pl->LH + 2) * 16 * pl->ld_h + 16);
  MBPO_REQUIRE(pl->lds <= 160 * 1024, MBPO_ERR_UNSUPPORTED, "ens_nll: shapes do not fit 160 KiB of LDS");
  const long long tiles = (d->batch + 15) / 16;
  const int E = pl->dyn.n_nets;
  long long cap = (2LL * ens_num_cus() + E - 1) / E;
  if (cap < 1) cap = 1;
  pl->n_slots = (int)(tiles < cap ? tiles : cap);
  pl->total = (long long)E * pl->n_slots * pl->dyn.n_params + (((long long)E * pl->n_slots + 3) & ~3LL);
  if (need_ptrs)
    MBPO_REQUIRE(d->dynamics.params && d->rows && d->idx && d->grads && d->metrics && d->workspace, MBPO_ERR_ARG, "ens_nll: null pointer");
  return MBPO_OK;
}

extern "C" int64_t mbpo_ens_nll_workspace_floats(const mbpo_ens_train_desc *d) {
  EnsPlan pl;
  int rc = ens_plan(d, &pl, false);
  if (rc != MBPO_OK) return rc;
  return pl.total;
}

extern "C" int mbpo_ens_nll_grads(const mbpo_ens_train_desc *d, void *stream) {
  EnsPlan pl;
  int rc = ens_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  EnsTrainArgs A;
  const int L = pl.dyn.n_layers, E = pl.dyn.n_nets;
  A.sh = NetShape{pl.dyn.dims[0], L, pl.dyn.dims[L], pl.dyn.act};
  A.params = d->dynamics.params; A.net_stride = pl.dyn.net_stride; A.n_params = pl.dyn.n_params; A.E = E;
  A.X = d->x_dim; A.U = d->u_dim; A.D = d->row_len; A.noff = d->next_obs_off;
  A.rows = d->rows; A.idx = d->idx; A.batch = d->batch; A.predict_delta = d->predict_delta; A.min_std = d->min_std;
  A.slabs = d->workspace; A.extras = d->workspace + (long long)E * pl.n_slots * pl.dyn.n_params;
  A.n_slots = pl.n_slots; A.ld_xu = pl.ld_xu; A.ld_h = pl.ld_h; A.ld_y = pl.ld_y; A.LH = pl.LH;
  const bool wide = net_is_wide(A.sh);
  rc = wide ? mbpo_ensure_lds<k_ens_nll_fwd_bwd<4, true>>(pl.lds, "ens_nll_grads") : mbpo_ensure_lds<k_ens_nll_fwd_bwd<4, false>>(pl.lds, "ens_nll_grads");
  if (rc != MBPO_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (wide) hipLaunchKernelGGL((k_ens_nll_fwd_bwd<4, true>), dim3(E * pl.n_slots), dim3(512), pl.lds, st, A);
  else hipLaunchKernelGGL((k_ens_nll_fwd_bwd<4, false>), dim3(E * pl.n_slots), dim3(512), pl.lds, st, A);
  hipLaunchKernelGGL(k_ens_reduce, dim3((pl.dyn.n_params + 63) / 64, E), dim3(256), 0, st, (const float *)A.slabs, (const float *)A.extras,
                     pl.n_slots, pl.dyn.n_params, (long long)d->batch, d->grads, d->metrics);
  MBPO_CHECK_LAUNCH("ens_nll_grads");
  return MBPO_OK;
}

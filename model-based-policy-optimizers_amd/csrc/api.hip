// api.hip — library-level entry points and host-side argument validation shared by all kernels.
#include "common.hpp"
#include <string.h>

static thread_local char g_err[512] = "";

void mbpo_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *mbpo_last_error(void) { return g_err; }

extern "C" int mbpo_version(void) { return 100; /* 0.1.0 */ }

int mbpo_make_mlp_dev(const mbpo_mlp_desc *d, MlpDev *out, const char *name) {
  MBPO_REQUIRE(d != nullptr, MBPO_ERR_ARG, "%s: null mlp descriptor", name);
  MBPO_REQUIRE(d->params != nullptr, MBPO_ERR_ARG, "%s: null params", name);
  MBPO_REQUIRE(d->n_layers >= 1 && d->n_layers <= MBPO_MAX_LAYERS, MBPO_ERR_ARG, "%s: n_layers=%d out of [1,%d]", name,
               d->n_layers, MBPO_MAX_LAYERS);
  MBPO_REQUIRE(d->n_nets >= 1, MBPO_ERR_ARG, "%s: n_nets=%d < 1", name, d->n_nets);
  MBPO_REQUIRE(d->activation >= 0 && d->activation <= 2, MBPO_ERR_ARG, "%s: unknown activation %d", name, d->activation);
  int off = 0;
  for (int l = 0; l <= d->n_layers; ++l) {
    MBPO_REQUIRE(d->dims[l] >= 1 && d->dims[l] <= 4096, MBPO_ERR_ARG, "%s: dims[%d]=%d out of range", name, l, d->dims[l]);
    out->dims[l] = d->dims[l];
  }
  for (int l = 0; l < d->n_layers; ++l) {
    out->w_off[l] = off;
    off += d->dims[l] * d->dims[l + 1];
    out->b_off[l] = off;
    off += d->dims[l + 1];
  }
  MBPO_REQUIRE(d->n_nets == 1 || d->net_stride >= off, MBPO_ERR_ARG, "%s: net_stride=%lld < params per net %d", name,
               (long long)d->net_stride, off);
  out->params = d->params;
  out->net_stride = d->net_stride;
  out->n_nets = d->n_nets;
  out->n_layers = d->n_layers;
  out->act = d->activation;
  out->n_params = off;
  return MBPO_OK;
}

// ---------------------------------------------------------------------------------------------- device RNG control words
__global__ void k_rng_advance(unsigned long long *rng, unsigned long long inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += inc;
}

extern "C" int mbpo_rng_advance(uint64_t *rng_dev, uint64_t inc, void *stream) {
  MBPO_REQUIRE(rng_dev, MBPO_ERR_ARG, "rng_advance: null rng_dev");
  hipLaunchKernelGGL(k_rng_advance, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long *)rng_dev, (unsigned long long)inc);
  MBPO_CHECK_LAUNCH("rng_advance");
  return MBPO_OK;
}

// ---------------------------------------------------------------------------------------------- standard-normal fill
// out[i] = philox_normal(seed + rng_dev[0], offset + rng_dev[1], stream, elem_base + i): the draws a fused kernel makes in
// registers, as a tensor — for host-side loops that walk a horizon step by step with a user's code between the kernels (BPTT
// through a user-defined System) and must consume the SAME numbers as the fused kernel (k_bptt_actor: element (traj*H + t)*u + d).
__global__ void __launch_bounds__(256) k_philox_normal_fill(unsigned long long seed, unsigned long long offset, const unsigned long long *rng_dev,
                                                            unsigned int stream, unsigned long long elem_base, long long n, float *out) {
  const RngKey k = rng_resolve(seed, offset, rng_dev);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = philox_normal(k.seed, k.offset, stream, elem_base + (unsigned long long)i);
}

extern "C" int mbpo_philox_normal_fill(uint64_t seed, uint64_t offset, const uint64_t *rng_dev, uint32_t stream, uint64_t elem_base,
                                       int64_t n, float *out, void *stream_) {
  MBPO_REQUIRE(out && n > 0, MBPO_ERR_ARG, "philox_normal_fill: null out / n <= 0");
  MBPO_REQUIRE(stream >= 1 && stream <= 10, MBPO_ERR_ARG, "philox_normal_fill: unknown stream id %u", stream);
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_philox_normal_fill, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream_,
                     (unsigned long long)seed, (unsigned long long)offset, (const unsigned long long *)rng_dev, stream,
                     (unsigned long long)elem_base, (long long)n, out);
  MBPO_CHECK_LAUNCH("philox_normal_fill");
  return MBPO_OK;
}

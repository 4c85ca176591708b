// common.hpp — shared device/host helpers for libmbpo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mbpo_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// ---------------------------------------------------------------- error plumbing (host)
void mbpo_set_error(const char *fmt, ...);

#define MBPO_REQUIRE(cond, code, ...)        \
  do {                                       \
    if (!(cond)) {                           \
      mbpo_set_error(__VA_ARGS__);           \
      return (code);                         \
    }                                        \
  } while (0)

#define MBPO_CHECK_LAUNCH(what)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      mbpo_set_error("%s: %s", what, hipGetErrorString(e__));                \
      return MBPO_ERR_LAUNCH;                                                \
    }                                                                        \
  } while (0)

// Raise a kernel's dynamic-LDS limit when needed.  The attribute is sticky per kernel, so it is set only when a launch
// needs more than any earlier one did: steady-state launches (and hipGraph capture) issue no attribute call at all.
// The kernel is a template ARGUMENT, so every kernel instantiation owns its own `granted`.
template <auto Kern>
int mbpo_ensure_lds(size_t bytes, const char *what) {
  static size_t granted = 48 * 1024;
  if (bytes > 160 * 1024) {
    mbpo_set_error("%s: needs %zu B of LDS per workgroup (> 160 KiB); reduce ensemble size, hidden width or depth", what, bytes);
    return MBPO_ERR_UNSUPPORTED;
  }
  if (bytes > granted) {
    hipError_t e = hipFuncSetAttribute((const void *)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      mbpo_set_error("%s: hipFuncSetAttribute(%zu): %s", what, bytes, hipGetErrorString(e));
      return MBPO_ERR_LAUNCH;
    }
    granted = bytes;
  }
  return MBPO_OK;
}

// ---------------------------------------------------------------- device-side MLP description
struct MlpDev {
  const float *params;
  long long net_stride;
  int n_nets;
  int n_layers;
  int dims[MBPO_MAX_LAYERS + 1];
  int w_off[MBPO_MAX_LAYERS];
  int b_off[MBPO_MAX_LAYERS];
  int act;
  int n_params;  // floats per net
};

// host: validate + fill offsets. Returns MBPO_OK or error.
int mbpo_make_mlp_dev(const mbpo_mlp_desc *d, MlpDev *out, const char *name);

// ---------------------------------------------------------------- math
// sigmoid on the hardware transcendental path: v_exp_f32 (2^x) + v_rcp_f32, both ~1 ulp.  One wave evaluates 16
// activations per layer, so the libm expf + IEEE-division expansion (~50 VALU instructions per element) cost as much
// as the layer's MFMAs; this form is ~5 instructions.  |relative error| <~ 1e-6 for |v| <= 20 (tests state tolerances).
__device__ __forceinline__ float fast_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
}

__device__ __forceinline__ float act_apply(float v, int act) {
  // swish(x) = x * sigmoid(x)  (flax.linen.swish); relu; tanh
  if (act == MBPO_ACT_SWISH) return v * fast_sigmoid(v);
  if (act == MBPO_ACT_RELU) return fmaxf(v, 0.0f);
  return tanhf(v);
}

// d act(v) / dv
__device__ __forceinline__ float act_grad(float v, int act) {
  if (act == MBPO_ACT_SWISH) {
    float sg = fast_sigmoid(v);
    return sg * (1.0f + v * (1.0f - sg));
  }
  if (act == MBPO_ACT_RELU) return v > 0.0f ? 1.0f : 0.0f;
  float t = tanhf(v);
  return 1.0f - t * t;
}

// element-wise activation over a small register array with the (wave-uniform) switch hoisted out of the loop
template <int NV>
__device__ __forceinline__ void act_apply_vec(float (&v)[NV], int act) {
  if (act == MBPO_ACT_SWISH) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = v[i] * fast_sigmoid(v[i]);
  } else if (act == MBPO_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = fmaxf(v[i], 0.0f);
  } else if (act == MBPO_ACT_TANH) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = tanhf(v[i]);
  }
}

// v[i] *= act'(z[i])
template <int NV>
__device__ __forceinline__ void act_grad_mul_vec(float (&v)[NV], const float (&z)[NV], int act) {
  if (act == MBPO_ACT_SWISH) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float sg = fast_sigmoid(z[i]);
      v[i] *= sg * (1.0f + z[i] * (1.0f - sg));
    }
  } else if (act == MBPO_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = z[i] > 0.0f ? v[i] : 0.0f;
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float t = tanhf(z[i]);
      v[i] *= 1.0f - t * t;
    }
  }
}

// jax.nn.softplus(x) = logaddexp(x, 0) = max(x,0) + log1p(exp(-|x|))
__device__ __forceinline__ float softplus_f(float x) {
  return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------- Philox4x32-10 (counter-based RNG)
// Matches oracle/philox.py bit for bit (integer arithmetic only).
struct Philox4 {
  uint32_t v[4];
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// RNG streams (counter word 2): which consumer draws
#define MBPO_STREAM_POLICY_NOISE 1u
#define MBPO_STREAM_MODEL_NOISE 2u
#define MBPO_STREAM_MEMBER 3u
#define MBPO_STREAM_REPLAY 4u
#define MBPO_STREAM_SAC_ALPHA 5u
#define MBPO_STREAM_SAC_CRITIC 6u
#define MBPO_STREAM_SAC_ACTOR 7u
#define MBPO_STREAM_PERM 8u
#define MBPO_STREAM_ENTROPY 9u
#define MBPO_STREAM_ICEM 10u

// Device-resident RNG control (optional on every Philox-drawing entry point): rng_dev = uint64[2] {seed word, step counter}.
// The effective key is (seed + rng_dev[0], offset + rng_dev[1]): a captured hipGraph whose host-side seed/offset are baked
// constants draws exactly what the eager path draws, because everything that changes between steps lives in these two
// device words (mbpo_rng_advance bumps the counter).  Integer arithmetic only: no 2^24 float-counter saturation.
struct RngKey {
  unsigned long long seed, offset;
};
__device__ __forceinline__ RngKey rng_resolve(unsigned long long seed, unsigned long long offset, const unsigned long long *rng_dev) {
  RngKey k;
  k.seed = seed + (rng_dev ? rng_dev[0] : 0ull);
  k.offset = offset + (rng_dev ? rng_dev[1] : 0ull);
  return k;
}

// standard normal for element `idx` of stream `stream` at call counter `offset` under `seed`
// (Box-Muller on two of the four Philox words; one Philox call per element keeps draws order-independent).
// (in two halves so that a kernel can place the integer rounds and the transcendental part in different barrier intervals)
__device__ __forceinline__ Philox4 philox_normal_bits(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx) {
  return philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), stream ^ (uint32_t)(offset >> 32) * 0x9E3779B9u,
                       (uint32_t)offset, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ float philox_normal_from_bits(const Philox4 &p) {
  float u1 = ((float)(p.v[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
  float u2 = (float)(p.v[1] >> 8) * (1.0f / 16777216.0f);           // [0,1)
  float rad = sqrtf(-2.0f * logf(u1));
  return rad * cosf(6.28318530717958647692f * u2);
}
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx) {
  return philox_normal_from_bits(philox_normal_bits(seed, offset, stream, idx));
}

// uniform integer in [lo, hi) for element idx: lo + mulhi(u32, span)   (span = hi-lo > 0)
__device__ __forceinline__ int32_t philox_randint(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx,
                                                  int32_t lo, int32_t hi) {
  Philox4 p = philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), stream ^ (uint32_t)(offset >> 32) * 0x9E3779B9u,
                            (uint32_t)offset, (uint32_t)seed, (uint32_t)(seed >> 32));
  uint32_t span = (uint32_t)(hi - lo);
  return lo + (int32_t)(((uint64_t)p.v[0] * span) >> 32);
}

// sum_t slab[t*stride + i] in t order with the loads of 8 terms in flight at a time (a plain `g += slab[t]` loop with a runtime
// trip count issues load, wait, add, load, ...: one dependent HBM/L2 round trip per term — 53 us for PPO's 160 slabs).  Same
// additions in the same order as the plain loop: bit-identical.
template <int NB = 8>
__device__ __forceinline__ float slab_sum(const float *slab, long long stride, int n_tiles, long long i) {
  float g = 0.f;
  int t = 0;
  for (; t + NB <= n_tiles; t += NB) {
    float v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = slab[(long long)(t + k) * stride + i];
#pragma unroll
    for (int k = 0; k < NB; ++k) g += v[k];
  }
  for (; t < n_tiles; ++t) g += slab[(long long)t * stride + i];
  return g;
}


// Workgroup form for MANY slabs (n up to the CU count) and few outputs: a 256-thread workgroup produces 64 outputs, wave q sums
// the q-th quarter of the terms (coalesced: 64 consecutive outputs per wave), the four partials are then added in q order.
// 4x the workgroups of one-output-per-thread (a 10 k-parameter gradient is 40 workgroups otherwise: 40 of 256 CUs busy) and a
// quarter of the dependent batches per thread.  Fixed order: deterministic.  Returns the sum on wave 0 (other waves: 0).
// i = this thread's output index (blockIdx.x * 64 + (threadIdx.x & 63)), valid = i in range.
__device__ __forceinline__ float slab_sum_wg64(const float *slab, long long stride, int n, long long i, bool valid) {
  __shared__ float s_q[4][64];
  const int p = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int chunk = (n + 3) >> 2;
  const int t0 = q * chunk, t1 = (t0 + chunk < n) ? t0 + chunk : n;
  s_q[q][p] = (valid && t1 > t0) ? slab_sum<16>(slab + (long long)t0 * stride, stride, t1 - t0, i) : 0.f;
  __syncthreads();
  return q == 0 ? ((s_q[0][p] + s_q[1][p]) + s_q[2][p]) + s_q[3][p] : 0.f;
}

// First stage of a two-stage sum of MANY long slabs: workgroup (g, c) adds slabs 16 g .. 16 g + 15 over the column block
// [1024 c, 1024 c + 1024) — sixteen 4-KB runs, each contiguous — and leaves the result IN slab 16 g (it alone reads those columns
// of those slabs).  The second stage is the ordinary strided sum over slabs 0, 16, 32, ... (slab_sum_wg64 with stride 16 * stride).
// Fixed order ((s0 + .. + s15) + (s16 + ..) + ..): deterministic.  256 threads; thread t owns columns c0 + t + 256 k, k < 4.
__device__ __forceinline__ void slab_group16_sum(float *slab, long long stride, int n, long long n_cols) {
  const int g = blockIdx.y;
  const long long c0 = (long long)blockIdx.x * 1024 + threadIdx.x;
  const int t0 = g * 16, t1 = (t0 + 16 < n) ? t0 + 16 : n;
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ok[k] = c0 + 256 * k < n_cols;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int t = t0;
  for (; t + 8 <= t1; t += 8) {
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k] = ok[k] ? slab[(long long)(t + u) * stride + c0 + 256 * k] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += v[u][k];
  }
  for (; t < t1; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += ok[k] ? slab[(long long)t * stride + c0 + 256 * k] : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (ok[k]) slab[(long long)t0 * stride + c0 + 256 * k] = acc[k];
}


// replay.hip — R9 UniformSamplingQueue (insert / gather / sample) and R8 running_statistics (see mbpo_hip.h).
//
// HBM-bound byte movers: rows are moved as dwords with consecutive lanes on consecutive floats of the
// flattened [n_rows*row_len] range, so every wave-instruction touches 256 contiguous bytes on the
// contiguous side of the copy (the gather side is row-granular: row_len*4 B segments).
// All position arithmetic is int32/int64 and identical to the reference's (bit-exact requirement).
#include "common.hpp"
#include <stdlib.h>

// state = {insert_position, sample_position, head, total_inserted}
struct ReplayPos {
  int pos, roll, head;
};

__device__ __forceinline__ ReplayPos replay_plan(const int *state, long long max_size, long long n) {
  // insert_internal [3P brax]: roll = min(0, len(data) - position - len(update)); data = roll(data, roll);
  // position += roll; dynamic_update_slice at position.
  ReplayPos p;
  long long pos = state[0];
  long long roll = max_size - pos - n;
  if (roll > 0) roll = 0;
  p.roll = (int)roll;
  p.pos = (int)(pos + roll);
  // jnp.roll(data, roll<0): new[i] = old[(i - roll) % max]  =>  head' = (head - roll) % max
  p.head = (int)(((long long)state[2] - roll) % max_size);
  return p;
}

// The rows of one insert are consecutive LOGICAL rows, i.e. consecutive physical rows up to (at most) one wrap of the ring: the
// insert is two flat copies.  No per-element division (the first version paid two 64-bit divisions per float), 16 bytes per lane
// when both sides are 16-byte aligned (row_len a multiple of 4, or a start that happens to be).
__global__ void __launch_bounds__(256) k_replay_insert(float *data, long long max_size, int D, const int *state,
                                                        const float *rows, long long n_rows) {
  const ReplayPos p = replay_plan(state, max_size, n_rows);
  const long long s0 = ((long long)p.pos + p.head) % max_size;          // physical row of the first inserted row
  const long long n1 = (max_size - s0) < n_rows ? (max_size - s0) : n_rows;   // rows before the wrap
  const long long e1 = n1 * D, total = n_rows * D;
  float *dst1 = data + s0 * D;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long long)gridDim.x * blockDim.x;
  const bool vec = ((e1 | total) & 3) == 0 && ((((unsigned long long)dst1) | ((unsigned long long)data) | ((unsigned long long)rows)) & 15ull) == 0;
  if (vec) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(rows);
    f32x4 *d1 = reinterpret_cast<f32x4 *>(dst1), *d2 = reinterpret_cast<f32x4 *>(data);
    const long long q1 = e1 >> 2, qt = total >> 2;
    for (long long i = gtid; i < qt; i += gsz) {
      const f32x4 v = src[i];
      if (i < q1) d1[i] = v;
      else d2[i - q1] = v;
    }
  } else {
    for (long long i = gtid; i < total; i += gsz) {
      const float v = rows[i];
      if (i < e1) dst1[i] = v;
      else data[i - e1] = v;
    }
  }
}

__global__ void k_replay_advance(int *state, long long max_size, long long n_rows) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const ReplayPos p = replay_plan(state, max_size, n_rows);
    long long newpos = ((long long)p.pos + n_rows) % (max_size + 1);  // position = (position + len(update)) % (len(data)+1)
    long long sp = (long long)state[1] + p.roll;                        // sample_position = max(0, sample_position + roll)
    state[0] = (int)newpos;
    state[1] = (int)(sp > 0 ? sp : 0);
    state[2] = p.head;
    state[3] = (int)((unsigned)state[3] + (unsigned)n_rows);
  }
}

extern "C" int mbpo_replay_insert(float *data, int64_t max_size, int32_t row_len, int32_t *state, const float *rows,
                                  int64_t n_rows, void *stream) {
  MBPO_REQUIRE(data && state, MBPO_ERR_ARG, "replay_insert: null pointer");
  MBPO_REQUIRE(max_size > 0 && max_size < (1LL << 31) - 1 && row_len > 0, MBPO_ERR_ARG, "replay_insert: bad max_size/row_len");
  MBPO_REQUIRE(n_rows >= 0 && n_rows <= max_size, MBPO_ERR_ARG,
               "replay_insert: %lld rows do not fit a buffer of %lld rows", (long long)n_rows, (long long)max_size);
  if (n_rows == 0) return MBPO_OK;
  MBPO_REQUIRE(rows, MBPO_ERR_ARG, "replay_insert: null rows");
  long long total = n_rows * row_len;
  int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_replay_insert, dim3(grid), dim3(256), 0, st, data, (long long)max_size, row_len, state, rows,
                     (long long)n_rows);
  hipLaunchKernelGGL(k_replay_advance, dim3(1), dim3(64), 0, st, state, (long long)max_size, (long long)n_rows);
  MBPO_CHECK_LAUNCH("replay_insert");
  return MBPO_OK;
}

// SAMPLE: false -> idx given (gather); true -> idx from Philox
// A workgroup takes 256 rows at a time: every thread resolves ONE row's physical position (one Philox block per row — the first
// version drew it again for every one of the row's D floats), then the 256 rows are copied cooperatively, 16 bytes per lane
// when D is a multiple of 4 (rows of 2x+u+3 floats with x=4,u=1 are 48 bytes: three lanes per row, consecutive lanes write
// consecutive 16-byte pieces of `out`).  The gather side reads whole rows, i.e. every 64-byte sector it touches is used at 75 %
// or more; the scatter into random rows of a table much larger than L2 is what bounds this kernel below the streaming rate.
static int gather_rows_per_block(int row_len) {
  const int pieces = (row_len & 3) == 0 ? row_len >> 2 : row_len;     // 16-byte (or 4-byte) pieces per row
  int rb = (2048 + pieces - 1) / pieces;
  return rb < 1 ? 1 : (rb > 256 ? 256 : rb);
}

template <bool SAMPLE>
__global__ void __launch_bounds__(256) k_replay_gather(const float *data, long long max_size, int D, const int *state,
                                                        const int *idx, unsigned long long seed, unsigned long long offset,
                                                        const unsigned long long *rng_dev, long long n, int *idx_out, float *out, int RB) {
  // RB <= 256 rows per workgroup iteration (host: ~2 k 16-byte pieces per iteration, so long rows — PPO gathers whole trajectories of
  // T*D floats — spread over many workgroups instead of 256 rows x 1920 B on each of 64)
  __shared__ long long s_phys[256];
  if (SAMPLE) {
    const RngKey rk = rng_resolve(seed, offset, rng_dev);
    seed = rk.seed;
    offset = rk.offset;
  }
  const int head = state[2];
  const int lo = state[1], hi = state[0];
  const int tid = threadIdx.x;
  const bool vec = (D & 3) == 0 && ((((unsigned long long)data) | ((unsigned long long)out)) & 15ull) == 0;
  for (long long row0 = (long long)blockIdx.x * RB; row0 < n; row0 += (long long)gridDim.x * RB) {
    const long long j = row0 + tid;
    if (tid < RB && j < n) {
      long long li;
      if (SAMPLE) {
        // jax.random.randint(sample_key, (n,), minval=sample_position, maxval=insert_position) — stream restated with Philox
        li = philox_randint(seed, offset, MBPO_STREAM_REPLAY, (unsigned long long)j, lo, hi);
        if (idx_out) idx_out[j] = (int)li;
      } else {
        li = idx[j];
      }
      // jnp.take(mode='wrap'): python-style modulo
      long long w = li % max_size;
      if (w < 0) w += max_size;
      s_phys[tid] = (w + head) % max_size;
    }
    __syncthreads();
    const int rows_here = (int)((n - row0) < RB ? (n - row0) : RB);
    if (vec) {
      const int D4 = D >> 2, total = rows_here * D4;
      const f32x4 *src = reinterpret_cast<const f32x4 *>(data);
      f32x4 *dst = reinterpret_cast<f32x4 *>(out) + row0 * D4;
      for (int e = tid; e < total; e += 256) {
        const int r = e / D4, c = e - r * D4;
        dst[e] = src[s_phys[r] * D4 + c];
      }
    } else {
      const int total = rows_here * D;
      float *dst = out + row0 * D;
      for (int e = tid; e < total; e += 256) {
        const int r = e / D, c = e - r * D;
        dst[e] = data[s_phys[r] * D + c];
      }
    }
    __syncthreads();
  }
}

extern "C" int mbpo_replay_gather(const float *data, int64_t max_size, int32_t row_len, const int32_t *state,
                                  const int32_t *idx, int64_t n, float *out, void *stream) {
  MBPO_REQUIRE(data && state, MBPO_ERR_ARG, "replay_gather: null pointer");
  MBPO_REQUIRE(max_size > 0 && row_len > 0 && n >= 0, MBPO_ERR_ARG, "replay_gather: bad sizes");
  if (n == 0) return MBPO_OK;
  MBPO_REQUIRE(idx && out, MBPO_ERR_ARG, "replay_gather: null idx/out");
  const int RB = gather_rows_per_block(row_len);
  int grid = (int)((n + RB - 1) / RB < 4096 ? (n + RB - 1) / RB : 4096);
  hipLaunchKernelGGL(k_replay_gather<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, data, (long long)max_size, row_len,
                     state, idx, 0ull, 0ull, (const unsigned long long *)nullptr, (long long)n, (int *)nullptr, out, RB);
  MBPO_CHECK_LAUNCH("replay_gather");
  return MBPO_OK;
}

extern "C" int mbpo_replay_sample(const float *data, int64_t max_size, int32_t row_len, const int32_t *state, uint64_t seed,
                                  uint64_t offset, const uint64_t *rng_dev, int64_t n, int32_t *idx_out, float *out,
                                  void *stream) {
  MBPO_REQUIRE(data && state, MBPO_ERR_ARG, "replay_sample: null pointer");
  MBPO_REQUIRE(max_size > 0 && row_len > 0 && n >= 0, MBPO_ERR_ARG, "replay_sample: bad sizes");
  if (n == 0) return MBPO_OK;
  MBPO_REQUIRE(out, MBPO_ERR_ARG, "replay_sample: null out");
  const int RB = gather_rows_per_block(row_len);
  int grid = (int)((n + RB - 1) / RB < 4096 ? (n + RB - 1) / RB : 4096);
  hipLaunchKernelGGL(k_replay_gather<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, data, (long long)max_size, row_len,
                     state, (const int *)nullptr, (unsigned long long)seed, (unsigned long long)offset, (const unsigned long long *)rng_dev, (long long)n, idx_out, out, RB);
  MBPO_CHECK_LAUNCH("replay_sample");
  return MBPO_OK;
}

// ------------------------------------------------------------------------------------------------
// running_statistics.update, split into reduce (per-rank sums) and apply.
// ------------------------------------------------------------------------------------------------
#define STATS_WGS 512

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Two passes, exactly as acme's update (so n=1 / constant columns give summed_variance == 0 exactly):
//   pass 0: partial[g][X] = sum over the workgroup's rows of d = obs - mean_old
//   pass 1: partial[g][X] = sum of d * (d - upd), upd = sum_d / (count + n)  (sum_d, n read from `sums`,
//           which a multi-GPU host has all-reduced in between — the reference's psum under pmap_axis_name)
// Fixed summation order everywhere (deterministic).
template <int PASS>
__global__ void __launch_bounds__(256) k_stats_partial(const float *rows, long long n_rows, int D, int col_off, int X,
                                                        const float *stats, const float *sums, float *partial) {
  const int tid = threadIdx.x;
  const float *mean = stats + 1;
  const int rows_per_pass = 256 / X;
  const int c = tid % X, r0 = tid / X;
  float acc = 0.f;
  if (r0 < rows_per_pass) {
    const float m = mean[c];
    float upd = 0.f;
    if (PASS == 1) upd = sums[1 + c] / (stats[0] + sums[0]);
    for (long long r = (long long)blockIdx.x * rows_per_pass + r0; r < n_rows; r += (long long)gridDim.x * rows_per_pass) {
      float d = rows[r * D + col_off + c] - m;
      acc += (PASS == 0) ? d : d * (d - upd);
    }
  }
  __shared__ float s_all[256];
  s_all[tid] = acc;
  __syncthreads();
  if (tid < X) {
    float a = 0.f;
    for (int t = tid; t < rows_per_pass * X; t += X) a += s_all[t];
    partial[(long long)blockIdx.x * X + tid] = a;
  }
}

// column c's partials are summed by 256/X threads (slice sl takes partials sl, sl + nsl, ...; 8 loads in flight), the slices then
// in slice order: a fixed order for given (n_parts, X)
// (workgroup function: 256 threads; returns column tid's total on threads tid < X; s_sl: 256 floats of LDS; ends behind a barrier)
__device__ __forceinline__ float stats_column_sums(const float *partial, int n_parts, int X, float *s_sl) {
  const int tid = threadIdx.x;
  const int nsl = 256 / X, c = tid % X, sl = tid / X;
  float acc = 0.f;
  if (sl < nsl) {
    int g = sl;
    for (; g + 7 * nsl < n_parts; g += 8 * nsl) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = partial[(long long)(g + k * nsl) * X + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; g < n_parts; g += nsl) acc += partial[(long long)g * X + c];
  }
  s_sl[tid] = acc;
  __syncthreads();
  float a = 0.f;
  if (tid < X)
    for (int k = 0; k < nsl; ++k) a += s_sl[k * X + tid];
  __syncthreads();
  return a;
}

template <int PASS>
__global__ void __launch_bounds__(256) k_stats_sums(const float *partial, int n_parts, int X, long long n_rows, float *sums) {
  __shared__ float s_sl[256];
  const int tid = threadIdx.x;
  const float a = stats_column_sums(partial, n_parts, X, s_sl);
  if (tid < X) sums[1 + PASS * X + tid] = a;
  if (PASS == 0 && tid == 0) sums[0] = (float)n_rows;
}

extern "C" int64_t mbpo_running_stats_workspace_floats(int32_t x_dim) {
  if (x_dim <= 0 || x_dim > 128) return MBPO_ERR_ARG;
  return 2 * (int64_t)STATS_WGS * x_dim;      // one block of workgroup partials per pass (mbpo_running_stats_update keeps both)
}

extern "C" int mbpo_running_stats_reduce(const float *rows, int64_t n_rows, int32_t row_len, int32_t col_off, int32_t x_dim,
                                         const float *stats, float *sums, float *workspace, int32_t pass, void *stream) {
  MBPO_REQUIRE(stats && sums && workspace, MBPO_ERR_ARG, "running_stats_reduce: null pointer");
  MBPO_REQUIRE(x_dim > 0 && x_dim <= 128 && row_len > 0 && col_off >= 0 && col_off + x_dim <= row_len, MBPO_ERR_ARG,
               "running_stats_reduce: bad column range (x_dim must be <= 128)");
  MBPO_REQUIRE(n_rows >= 0 && (n_rows == 0 || rows), MBPO_ERR_ARG, "running_stats_reduce: bad rows");
  MBPO_REQUIRE(pass == 0 || pass == 1, MBPO_ERR_ARG, "running_stats_reduce: pass must be 0 or 1");
  hipStream_t st = (hipStream_t)stream;
  int rows_per_pass = 256 / x_dim;
  long long want = (n_rows + rows_per_pass - 1) / rows_per_pass;
  int grid = (int)(want < 1 ? 1 : (want < STATS_WGS ? want : STATS_WGS));
  if (pass == 0) {
    hipLaunchKernelGGL(k_stats_partial<0>, dim3(grid), dim3(256), 0, st, rows, (long long)n_rows, row_len, col_off, x_dim, stats,
                       (const float *)sums, workspace);
    hipLaunchKernelGGL(k_stats_sums<0>, dim3(1), dim3(256), 0, st, workspace, grid, x_dim, (long long)n_rows, sums);
  } else {
    hipLaunchKernelGGL(k_stats_partial<1>, dim3(grid), dim3(256), 0, st, rows, (long long)n_rows, row_len, col_off, x_dim, stats,
                       (const float *)sums, workspace);
    hipLaunchKernelGGL(k_stats_sums<1>, dim3(1), dim3(256), 0, st, workspace, grid, x_dim, (long long)n_rows, sums);
  }
  MBPO_CHECK_LAUNCH("running_stats_reduce");
  return MBPO_OK;
}

__global__ void k_stats_apply(float *stats, const float *sums, int X, float std_min, float std_max) {
  const int c = threadIdx.x;
  // [3P] running_statistics.update: count = state.count + step_increment; mean += sum(diff_to_old_mean)/count;
  // summed_variance += sum(diff_to_old * diff_to_new); std = clip(sqrt(max(sv,0)/count), 1e-6, 1e6)
  const float count = stats[0] + sums[0];
  if (c < X && count > 0.f) {
    float *mean = stats + 1, *sv = stats + 1 + X, *sd = stats + 1 + 2 * X;
    mean[c] = mean[c] + sums[1 + c] / count;
    float nsv = sv[c] + sums[1 + X + c];
    sv[c] = nsv;
    float s = sqrtf(fmaxf(nsv, 0.f) / count);
    sd[c] = fminf(fmaxf(s, std_min), std_max);
  }
  __syncthreads();
  if (c == 0) stats[0] = count;
}

extern "C" int mbpo_running_stats_apply(float *stats, const float *sums, int32_t x_dim, float std_min, float std_max, void *stream) {
  MBPO_REQUIRE(stats && sums, MBPO_ERR_ARG, "running_stats_apply: null pointer");
  MBPO_REQUIRE(x_dim > 0 && x_dim <= 128, MBPO_ERR_ARG, "running_stats_apply: x_dim out of range");
  hipLaunchKernelGGL(k_stats_apply, dim3(1), dim3(128), 0, (hipStream_t)stream, stats, sums, x_dim, std_min, std_max);
  MBPO_CHECK_LAUNCH("running_stats_apply");
  return MBPO_OK;
}

// ---- the whole update in THREE launches instead of five, for a single rank (no all-reduce between the passes): every workgroup of
// the second pass adds the first pass's partials up itself (the same fixed order in each: identical totals), and the one workgroup
// that sums the second pass's partials also applies the update.  Bit-identical to reduce(0) -> reduce(1) -> apply.
__global__ void __launch_bounds__(256) k_stats_pass1_fused(const float *rows, long long n_rows, int D, int col_off, int X,
                                                           const float *stats, const float *partial0, int n_parts0, float *sums,
                                                           float *partial1) {
  __shared__ float s_all[256];
  __shared__ float s_sum0[128];
  const int tid = threadIdx.x;
  const float tot = stats_column_sums(partial0, n_parts0, X, s_all);
  if (tid < X) {
    s_sum0[tid] = tot;
    if (blockIdx.x == 0) sums[1 + tid] = tot;
  }
  if (blockIdx.x == 0 && tid == 0) sums[0] = (float)n_rows;
  __syncthreads();
  const float *mean = stats + 1;
  const int rows_per_pass = 256 / X;
  const int c = tid % X, r0 = tid / X;
  float acc = 0.f;
  if (r0 < rows_per_pass) {
    const float m = mean[c];
    const float upd = s_sum0[c] / (stats[0] + (float)n_rows);
    for (long long r = (long long)blockIdx.x * rows_per_pass + r0; r < n_rows; r += (long long)gridDim.x * rows_per_pass) {
      float d = rows[r * D + col_off + c] - m;
      acc += d * (d - upd);
    }
  }
  s_all[tid] = acc;
  __syncthreads();
  if (tid < X) {
    float a = 0.f;
    for (int t = tid; t < rows_per_pass * X; t += X) a += s_all[t];
    partial1[(long long)blockIdx.x * X + tid] = a;
  }
}

__global__ void __launch_bounds__(256) k_stats_sums1_apply(const float *partial1, int n_parts, int X, float *sums, float *stats,
                                                           float std_min, float std_max) {
  __shared__ float s_sl[256];
  const int c = threadIdx.x;
  const float a = stats_column_sums(partial1, n_parts, X, s_sl);
  const float count = stats[0] + sums[0];
  if (c < X) {
    sums[1 + X + c] = a;
    if (count > 0.f) {
      float *mean = stats + 1, *sv = stats + 1 + X, *sd = stats + 1 + 2 * X;
      mean[c] = mean[c] + sums[1 + c] / count;
      float nsv = sv[c] + a;
      sv[c] = nsv;
      float s = sqrtf(fmaxf(nsv, 0.f) / count);
      sd[c] = fminf(fmaxf(s, std_min), std_max);
    }
  }
  __syncthreads();
  if (c == 0) stats[0] = count;
}

extern "C" int mbpo_running_stats_update(const float *rows, int64_t n_rows, int32_t row_len, int32_t col_off, int32_t x_dim,
                                         float *stats, float *sums, float *workspace, float std_min, float std_max, void *stream) {
  MBPO_REQUIRE(stats && sums && workspace, MBPO_ERR_ARG, "running_stats_update: null pointer");
  MBPO_REQUIRE(x_dim > 0 && x_dim <= 128 && row_len > 0 && col_off >= 0 && col_off + x_dim <= row_len, MBPO_ERR_ARG,
               "running_stats_update: bad column range (x_dim must be <= 128)");
  MBPO_REQUIRE(n_rows >= 0 && (n_rows == 0 || rows), MBPO_ERR_ARG, "running_stats_update: bad rows");
  hipStream_t st = (hipStream_t)stream;
  int rows_per_pass = 256 / x_dim;
  long long want = (n_rows + rows_per_pass - 1) / rows_per_pass;
  int grid = (int)(want < 1 ? 1 : (want < STATS_WGS ? want : STATS_WGS));
  float *partial1 = workspace + (long long)STATS_WGS * x_dim;
  hipLaunchKernelGGL(k_stats_partial<0>, dim3(grid), dim3(256), 0, st, rows, (long long)n_rows, row_len, col_off, x_dim,
                     (const float *)stats, (const float *)sums, workspace);
  hipLaunchKernelGGL(k_stats_pass1_fused, dim3(grid), dim3(256), 0, st, rows, (long long)n_rows, row_len, col_off, x_dim,
                     (const float *)stats, (const float *)workspace, grid, sums, partial1);
  hipLaunchKernelGGL(k_stats_sums1_apply, dim3(1), dim3(256), 0, st, (const float *)partial1, grid, x_dim, sums, stats, std_min, std_max);
  MBPO_CHECK_LAUNCH("running_stats_update");
  return MBPO_OK;
}

// ------------------------------------------------------------------------------------------------
// Random permutation of [0, n): the shared shuffle of PPO.sgd_step (ppo/ppo.py:166-171, jr.permutation with ONE key for
// every leaf == one permutation of whole trajectories).  JAX's stream is not reproducible; here
//   key_i = Philox(seed, offset [+ rng_dev], stream PERM, i).word0,   perm = stable argsort(key)
// computed as a rank count: perm[#{j : (key_j, j) < (key_i, i)}] = i.  O(n^2) compares, all of them from LDS tiles: n = B*M is
// 4 k-16 k trajectories (C3: 16384 -> 2.7e8 compares, a few tens of microseconds once per num_minibatches updates); bit-exact
// against numpy's stable argsort of the same keys (oracle/philox.py:philox_permutation).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_perm_keys(unsigned long long seed, unsigned long long offset, const unsigned long long *rng_dev,
                                                    long long n, unsigned int *keys) {
  const RngKey rk = rng_resolve(seed, offset, rng_dev);
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Philox4 p = philox4x32_10((uint32_t)i, (uint32_t)((unsigned long long)i >> 32), MBPO_STREAM_PERM ^ (uint32_t)(rk.offset >> 32) * 0x9E3779B9u,
                            (uint32_t)rk.offset, (uint32_t)rk.seed, (uint32_t)(rk.seed >> 32));
  keys[i] = p.v[0];
}

__global__ void __launch_bounds__(256) k_perm_rank(const unsigned int *keys, long long n, int *perm) {
  __shared__ unsigned int s_k[1024];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned int ki = i < n ? keys[i] : 0u;
  int rank = 0;
  for (long long j0 = 0; j0 < n; j0 += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) s_k[t] = (j0 + t < n) ? keys[j0 + t] : 0xFFFFFFFFu;
    __syncthreads();
    const int m = (int)((n - j0) < 1024 ? (n - j0) : 1024);
    for (int t = 0; t < m; ++t) {
      const unsigned int kj = s_k[t];
      rank += (kj < ki || (kj == ki && j0 + t < i)) ? 1 : 0;
    }
  }
  if (i < n) perm[rank] = (int)i;
}

// n <= 16384 (C3's B*M, the reference test's 4096): ONE workgroup draws the keys straight into LDS as 64-bit composites
// (key << 32 | index) — all distinct, so ANY correct sort of them is numpy's stable argsort of the keys — and runs a bitonic sort
// there: log2(np)(log2(np)+1)/2 compare-exchange sweeps of np/2 pairs, no global traffic but the n indices written at the end.
// The rank count above is 2.7e8 compares from 64 workgroups at n = 16384: 498 us per update epoch, 10 % of C3's training step
// (rocprofv3, round 3).  Padding to the next power of two with all-ones composites, which sort behind every real element.
__global__ void __launch_bounds__(1024) k_perm_sort_lds(unsigned long long seed, unsigned long long offset, const unsigned long long *rng_dev,
                                                         int n, int np, int *perm, unsigned int *flag) {
  extern __shared__ __align__(16) unsigned long long s_c[];
  if (flag) {                      // the fallback behind k_perm_bucket_sort: only if a bucket overflowed
    if (flag[0] == 0u) return;
    __syncthreads();
    if (threadIdx.x == 0) flag[0] = 0u;
  }
  const RngKey rk = rng_resolve(seed, offset, rng_dev);
  const int tid = threadIdx.x;
  for (int i = tid; i < np; i += 1024) {
    unsigned long long c = ~0ull;
    if (i < n) {
      Philox4 p = philox4x32_10((uint32_t)i, 0u, MBPO_STREAM_PERM ^ (uint32_t)(rk.offset >> 32) * 0x9E3779B9u, (uint32_t)rk.offset,
                                (uint32_t)rk.seed, (uint32_t)(rk.seed >> 32));
      c = ((unsigned long long)p.v[0] << 32) | (unsigned int)i;
    }
    s_c[i] = c;
  }
  for (int k = 2; k <= np; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = tid; t < (np >> 1); t += 1024) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // bit j clear
        const int hi = lo | j;
        const unsigned long long a = s_c[lo], b = s_c[hi];
        const bool up = (lo & k) == 0;                             // this pair's run is sorted ascending
        if ((a > b) == up) {
          s_c[lo] = b;
          s_c[hi] = a;
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) perm[i] = (int)(unsigned int)(s_c[i] & 0xFFFFFFFFull);
}

// 1024 < n <= 16384: the same permutation from 2^lb workgroups, each owning the keys whose top lb bits equal its index.  Every workgroup
// draws ALL n keys itself (Philox is integer arithmetic: n/256 draws per thread, nothing read from memory), counts the keys of
// smaller buckets (= where its run starts in the output), collects its own bucket's composites in LDS (~n / 2^lb = 64 of them; LDS
// append order does not matter: they are sorted next), bitonic-sorts them and writes its run.  A bucket that overflows the LDS list
// (never, for uniform keys: mean 64, room for 1024) raises flag[0]; k_perm_sort_lds — launched right behind with flag != nullptr —
// then redoes the whole permutation the old way and clears the flag, otherwise it returns at once.  The flag word's initial value
// does not matter: a stale 1 only costs one redundant full sort.  152 -> ~10 us at n = 16384 (C3's shared permutation, 8 per training step).
#define PERM_BUCKET_CAP 1024
__global__ void __launch_bounds__(256) k_perm_bucket_sort(unsigned long long seed, unsigned long long offset, const unsigned long long *rng_dev,
                                                          int n, int lb, int *perm, unsigned int *flag) {
  __shared__ unsigned long long s_c[PERM_BUCKET_CAP];
  __shared__ int s_cnt, s_lt[4];
  const RngKey rk = rng_resolve(seed, offset, rng_dev);
  const int tid = threadIdx.x;
  const unsigned int b = blockIdx.x;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int lt = 0;
  for (int i = tid; i < n; i += 256) {
    Philox4 p = philox4x32_10((uint32_t)i, 0u, MBPO_STREAM_PERM ^ (uint32_t)(rk.offset >> 32) * 0x9E3779B9u, (uint32_t)rk.offset,
                              (uint32_t)rk.seed, (uint32_t)(rk.seed >> 32));
    const unsigned int key = p.v[0], kb = key >> (32 - lb);
    lt += kb < b ? 1 : 0;
    if (kb == b) {
      const int at = atomicAdd(&s_cnt, 1);
      if (at < PERM_BUCKET_CAP) s_c[at] = ((unsigned long long)key << 32) | (unsigned int)i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lt += __shfl_down(lt, o, 64);
  if ((tid & 63) == 0) s_lt[tid >> 6] = lt;
  __syncthreads();
  const int cnt = s_cnt;
  if (cnt > PERM_BUCKET_CAP) {
    if (tid == 0) flag[0] = 1u;
    return;
  }
  const int base = s_lt[0] + s_lt[1] + s_lt[2] + s_lt[3];
  int np = 2;
  while (np < cnt) np <<= 1;
  for (int i = cnt + tid; i < np; i += 256) s_c[i] = ~0ull;
  for (int k = 2; k <= np; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = tid; t < (np >> 1); t += 256) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const unsigned long long a = s_c[lo], c = s_c[hi];
        const bool up = (lo & k) == 0;
        if ((a > c) == up) {
          s_c[lo] = c;
          s_c[hi] = a;
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < cnt; i += 256) perm[base + i] = (int)(unsigned int)(s_c[i] & 0xFFFFFFFFull);
}

extern "C" int mbpo_philox_permutation(uint64_t seed, uint64_t offset, const uint64_t *rng_dev, int64_t n, int32_t *perm,
                                       uint32_t *workspace, void *stream) {
  MBPO_REQUIRE(n >= 0 && n <= (1 << 20), MBPO_ERR_ARG, "philox_permutation: n=%lld outside [0, 2^20] (the rank count is O(n^2))", (long long)n);
  if (n == 0) return MBPO_OK;
  MBPO_REQUIRE(perm && workspace, MBPO_ERR_ARG, "philox_permutation: null perm/workspace");
  hipStream_t st = (hipStream_t)stream;
  if (n <= 16384) {
    int np = 2;
    while (np < n) np <<= 1;
    const size_t lds = (size_t)np * sizeof(unsigned long long);
    int rc = mbpo_ensure_lds<k_perm_sort_lds>(lds, "philox_permutation");
    if (rc != MBPO_OK) return rc;
    static const int bucket_env = getenv("MBPO_PERM_BUCKETS") ? atoi(getenv("MBPO_PERM_BUCKETS")) : 1;
    unsigned int *flag = nullptr;
    if (n > 1024 && bucket_env != 0) {
      int lb = 1;
      while ((n >> lb) > 64 && lb < 8) ++lb;          // ~64 keys per bucket, at most 256 buckets
      flag = workspace;
      hipLaunchKernelGGL(k_perm_bucket_sort, dim3(1u << lb), dim3(256), 0, st, (unsigned long long)seed, (unsigned long long)offset,
                         (const unsigned long long *)rng_dev, (int)n, lb, perm, flag);
    }
    hipLaunchKernelGGL(k_perm_sort_lds, dim3(1), dim3(1024), lds, st, (unsigned long long)seed, (unsigned long long)offset,
                       (const unsigned long long *)rng_dev, (int)n, np, perm, flag);
    MBPO_CHECK_LAUNCH("philox_permutation");
    return MBPO_OK;
  }
  const int grid = (int)((n + 255) / 256);
  hipLaunchKernelGGL(k_perm_keys, dim3(grid), dim3(256), 0, st, (unsigned long long)seed, (unsigned long long)offset,
                     (const unsigned long long *)rng_dev, (long long)n, workspace);
  hipLaunchKernelGGL(k_perm_rank, dim3(grid), dim3(256), 0, st, (const unsigned int *)workspace, (long long)n, perm);
  MBPO_CHECK_LAUNCH("philox_permutation");
  return MBPO_OK;
}

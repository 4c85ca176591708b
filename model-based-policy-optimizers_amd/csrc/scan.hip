// scan.hip — P4 GAE (ppo/losses.py:128-184) and B2 lambda-return (utils/optimizer_utils.py:119-152).
//
// Both are reverse first-order linear recurrences  A_t = d_t + c_t * A_{t+1}.  HBM-bound: GAE moves
// 4 reads + 2 writes = 24 B per (t,b) element (+4/T for the bootstrap), lambda-return 2 reads + 1 write = 12 B.
//
// batch-major [B,T] (PPO's native layout): the time axis sits on the LANES of a wavefront.  Each element
// is the affine map f_t(a) = d_t + c_t*a; a Hillis-Steele suffix scan over lanes with __shfl_down composes
// (c1,d1) o (c2,d2) = (c1*c2, d1 + c1*d2) in log2(Tp) steps.  Rows shorter than 64 are packed 64/Tp per
// wave (segmented scan, Tp = next power of two >= T) so loads stay coalesced; rows longer than 64 are
// walked in 64-step chunks from the end with a scalar carry.
// time-major [T,B] (the reference's layout after its transpose): lane <-> b, sequential over t, coalesced.
#include "common.hpp"

#define MODE_GAE 0
#define MODE_LAMBDA 1

struct ScanArgs {
  const float *trunc, *term, *rew, *val, *boot;  // GAE inputs; lambda: rew, val = next_values
  const float *disc;                             // GAE: optional per-element discount (non-equidistant time, losses_new.py:105-226)
  float *out0, *out1;                            // GAE: vs, adv;  lambda: returns
  long long B;
  int T;
  float gamma, lam;
};

template <int MODE>
__global__ void __launch_bounds__(256) k_scan_time_major(ScanArgs A) {
  const long long B = A.B;
  const int T = A.T;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (long long)gridDim.x * blockDim.x) {
    if (MODE == MODE_GAE) {
      const float boot = A.boot[b];
      float acc = 0.f;        // acc = zeros_like(bootstrap_value)   (losses.py:161)
      float v_next = boot;    // values_t_plus_1[-1] = bootstrap      (:155-156)
      float vs_next = boot;   // vs_t_plus_1[-1] = bootstrap          (:179-180)
      for (int t = T - 1; t >= 0; --t) {
        const long long i = (long long)t * B + b;
        const float tr = A.trunc[i], te = A.term[i], r = A.rew[i], v = A.val[i];
        const float m = 1.f - tr;
        const float g1 = (A.disc ? A.disc[i] : A.gamma) * (1.f - te);
        const float delta = (r + g1 * v_next - v) * m;        // :157-158
        acc = delta + g1 * m * A.lam * acc;                   // :166
        const float vs = acc + v;                             // :176
        A.out0[i] = vs;
        A.out1[i] = (r + g1 * vs_next - v) * m;               // :181-182
        v_next = v;
        vs_next = vs;
      }
    } else {
      float agg = A.val[(long long)(T - 1) * B + b];          // start = next_values[-1] (optimizer_utils.py:131)
      const float gl = A.gamma * A.lam;
      for (int t = T - 1; t >= 0; --t) {
        const long long i = (long long)t * B + b;
        const float inp = A.rew[i] + A.gamma * A.val[i] * (1.f - A.lam);   // :128
        agg = inp + gl * agg;                                              // :130
        A.out0[i] = agg;
      }
    }
  }
}

// TP: segment width on the lanes (power of two, <= 64)
template <int MODE, int TP>
__global__ void __launch_bounds__(256) k_scan_batch_major(ScanArgs A) {
  constexpr int RPW = 64 / TP;  // rows per wave
  const int lane = threadIdx.x & 63;
  const int seg = lane / TP, tl = lane % TP;
  const long long wave_global = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long n_groups = (A.B + RPW - 1) / RPW;
  const int T = A.T;
  const int n_chunks = (T + TP - 1) / TP;  // > 1 only when TP == 64
  for (long long grp = wave_global; grp < n_groups; grp += n_waves) {
    const long long b = grp * RPW + seg;
    const bool row_ok = b < A.B;
    float carry_a, carry_vs = 0.f;
    float boot = 0.f;
    if (MODE == MODE_GAE) {
      boot = row_ok ? A.boot[b] : 0.f;
      carry_a = 0.f;
      carry_vs = boot;
    } else {
      carry_a = row_ok ? A.val[b * T + (T - 1)] : 0.f;
    }
    for (int ch = n_chunks - 1; ch >= 0; --ch) {
      const int t = ch * TP + tl;
      const bool ok = row_ok && t < T;
      const long long i = b * T + t;
      float c = 1.f, d = 0.f, v = 0.f, r = 0.f, m = 0.f, g1 = 0.f;
      if (ok) {
        if (MODE == MODE_GAE) {
          const float tr = A.trunc[i], te = A.term[i];
          r = A.rew[i];
          v = A.val[i];
          const float v_next = (t == T - 1) ? boot : A.val[i + 1];
          m = 1.f - tr;
          g1 = (A.disc ? A.disc[i] : A.gamma) * (1.f - te);
          d = (r + g1 * v_next - v) * m;
          c = g1 * m * A.lam;
        } else {
          d = A.rew[i] + A.gamma * A.val[i] * (1.f - A.lam);
          c = A.gamma * A.lam;
        }
      }
      // lanes past the end of the row are the identity map (c=1,d=0): they pass the carry through
      // inclusive suffix scan within the segment
#pragma unroll
      for (int off = 1; off < TP; off <<= 1) {
        float c2 = __shfl_down(c, off, TP);
        float d2 = __shfl_down(d, off, TP);
        if (tl + off < TP) {
          d = d + c * d2;
          c = c * c2;
        }
      }
      const float a = d + c * carry_a;  // A_t
      if (MODE == MODE_GAE) {
        const float vs = a + v;
        float vs_next = __shfl_down(vs, 1, TP);
        // the last valid lane of the chunk takes the carry (bootstrap, or the first vs of the later chunk)
        const bool last_in_chunk = (tl == TP - 1) || (t == T - 1);
        if (last_in_chunk) vs_next = carry_vs;
        if (ok) {
          A.out0[i] = vs;
          A.out1[i] = (r + g1 * vs_next - v) * m;
        }
        carry_vs = __shfl(vs, seg * TP, 64);
      } else {
        if (ok) A.out0[i] = a;
      }
      carry_a = __shfl(a, seg * TP, 64);
    }
  }
}

// batch-major, T a multiple of 4 (and T <= 256): every lane owns FOUR consecutive time steps of a row (one 16-byte load per input
// array instead of four 4-byte ones), TL = T/4 lanes make a row and floor(64/TL) rows share a wave — at T = 40 that is 6 rows on 60
// lanes where the one-step-per-lane kernel above puts 1 row on 40.  In-lane the four affine maps are composed sequentially, across
// the lanes of a row with a segmented Hillis-Steele suffix scan (ceil(log2 TL) shuffle rounds), then the lane walks its four steps.
template <int MODE>
__global__ void __launch_bounds__(256) k_scan_batch_major_v4(ScanArgs A, int TL, int RPW) {
  const int lane = threadIdx.x & 63;
  const int seg = lane / TL, tl = lane - seg * TL;
  const long long wave_global = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long n_groups = (A.B + RPW - 1) / RPW;
  const int T = A.T;
  for (long long grp = wave_global; grp < n_groups; grp += n_waves) {
    const long long b = grp * RPW + seg;
    const bool ok = seg < RPW && b < A.B;
    const long long i = ok ? b * T + 4 * tl : 0;
    f32x4 r4 = {0.f, 0.f, 0.f, 0.f}, v4 = r4, m4 = r4, g4 = r4;
    float c[4] = {1.f, 1.f, 1.f, 1.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
    float carry = 0.f, boot = 0.f;
    if (ok) {
      r4 = *reinterpret_cast<const f32x4 *>(A.rew + i);
      v4 = *reinterpret_cast<const f32x4 *>(A.val + i);
      if (MODE == MODE_GAE) {
        const f32x4 tr = *reinterpret_cast<const f32x4 *>(A.trunc + i), te = *reinterpret_cast<const f32x4 *>(A.term + i);
        f32x4 ds = {A.gamma, A.gamma, A.gamma, A.gamma};
        if (A.disc) ds = *reinterpret_cast<const f32x4 *>(A.disc + i);
        boot = A.boot[b];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          m4[k] = 1.f - tr[k];
          g4[k] = ds[k] * (1.f - te[k]);
        }
      } else {
        carry = A.val[b * T + (T - 1)];
      }
    }
    // the value one step past this lane's four: the next lane's first, or the bootstrap at the end of the row
    float v_after = __shfl_down(v4[0], 1, 64);
    if (tl == TL - 1) v_after = boot;
    if (ok) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (MODE == MODE_GAE) {
          const float v_next = k < 3 ? v4[k + 1] : v_after;
          d[k] = (r4[k] + g4[k] * v_next - v4[k]) * m4[k];        // losses.py:157-158
          c[k] = g4[k] * m4[k] * A.lam;                            // :166
        } else {
          d[k] = r4[k] + A.gamma * v4[k] * (1.f - A.lam);          // optimizer_utils.py:128
          c[k] = A.gamma * A.lam;
        }
      }
    }
    // the lane's four steps as one map  a_first = D + C * a_after
    float C = c[0] * c[1] * c[2] * c[3];
    float D = d[0] + c[0] * (d[1] + c[1] * (d[2] + c[2] * d[3]));
    for (int off = 1; off < TL; off <<= 1) {      // inclusive suffix scan over the lanes of one row
      const float C2 = __shfl_down(C, off, 64), D2 = __shfl_down(D, off, 64);
      if (tl + off < TL) {
        D = D + C * D2;
        C = C * C2;
      }
    }
    // what enters this lane from the right: the next lane's inclusive result applied to the row's carry
    const float Cn = __shfl_down(C, 1, 64), Dn = __shfl_down(D, 1, 64);
    float a_in = (tl == TL - 1) ? carry : Dn + Cn * carry;
    float a[4];
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      a[k] = d[k] + c[k] * a_in;
      a_in = a[k];
    }
    if (MODE == MODE_GAE) {
      f32x4 vs, adv;
#pragma unroll
      for (int k = 0; k < 4; ++k) vs[k] = a[k] + v4[k];                     // :176
      float vs_after = __shfl_down(vs[0], 1, 64);
      if (tl == TL - 1) vs_after = boot;                                    // :179-180
#pragma unroll
      for (int k = 0; k < 4; ++k) adv[k] = (r4[k] + g4[k] * (k < 3 ? vs[k + 1] : vs_after) - v4[k]) * m4[k];   // :181-182
      if (ok) {
        *reinterpret_cast<f32x4 *>(A.out0 + i) = vs;
        *reinterpret_cast<f32x4 *>(A.out1 + i) = adv;
      }
    } else if (ok) {
      f32x4 o = {a[0], a[1], a[2], a[3]};
      *reinterpret_cast<f32x4 *>(A.out0 + i) = o;
    }
  }
}

template <int MODE>
static int launch_scan(const ScanArgs &A, int time_major, hipStream_t st, const char *what) {
  if (A.B == 0 || A.T == 0) return MBPO_OK;
  if (time_major) {
    long long blocks = (A.B + 255) / 256;
    int grid = (int)(blocks < 2048 ? blocks : 2048);
    hipLaunchKernelGGL(k_scan_time_major<MODE>, dim3(grid), dim3(256), 0, st, A);
  } else {
    // 16-byte form: T a multiple of 4 (rows then start on 16-byte boundaries whenever the arrays do), at most 64 lanes per row
    const unsigned long long al = (unsigned long long)A.rew | (unsigned long long)A.val | (unsigned long long)A.out0 |
                                  (unsigned long long)A.trunc | (unsigned long long)A.term | (unsigned long long)A.out1 | (unsigned long long)A.disc;
    if ((A.T & 3) == 0 && A.T >= 8 && A.T <= 256 && (al & 15ull) == 0) {
      const int TL = A.T / 4, RPW = 64 / TL;
      const long long groups = (A.B + RPW - 1) / RPW;
      const long long blocks = (groups + 3) / 4;
      const int grid = (int)(blocks < 4096 ? blocks : 4096);
      hipLaunchKernelGGL(k_scan_batch_major_v4<MODE>, dim3(grid), dim3(256), 0, st, A, TL, RPW);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        mbpo_set_error("%s: %s", what, hipGetErrorString(e));
        return MBPO_ERR_LAUNCH;
      }
      return MBPO_OK;
    }
    int tp = 1;
    while (tp < A.T && tp < 64) tp <<= 1;
    long long groups = (A.B + (64 / tp) - 1) / (64 / tp);
    long long blocks = (groups + 3) / 4;
    int grid = (int)(blocks < 2048 ? blocks : 2048);
#define CASE_TP(V) case V: hipLaunchKernelGGL((k_scan_batch_major<MODE, V>), dim3(grid), dim3(256), 0, st, A); break;
    switch (tp) {
      CASE_TP(1) CASE_TP(2) CASE_TP(4) CASE_TP(8) CASE_TP(16) CASE_TP(32) CASE_TP(64)
    }
#undef CASE_TP
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mbpo_set_error("%s: %s", what, hipGetErrorString(e));
    return MBPO_ERR_LAUNCH;
  }
  return MBPO_OK;
}

extern "C" int mbpo_gae_scan(const float *truncation, const float *termination, const float *rewards, const float *values,
                             const float *bootstrap, float *vs, float *advantages, int64_t B, int32_t T, float gamma,
                             float lam, int32_t time_major, void *stream) {
  MBPO_REQUIRE(B >= 0 && T >= 0, MBPO_ERR_ARG, "gae_scan: negative size");
  if (B == 0 || T == 0) return MBPO_OK;
  MBPO_REQUIRE(truncation && termination && rewards && values && bootstrap && vs && advantages, MBPO_ERR_ARG,
               "gae_scan: null pointer");
  ScanArgs A{truncation, termination, rewards, values, bootstrap, nullptr, vs, advantages, B, T, gamma, lam};
  return launch_scan<MODE_GAE>(A, time_major, (hipStream_t)stream, "gae_scan");
}

extern "C" int mbpo_gae_scan_discounts(const float *truncation, const float *termination, const float *rewards, const float *values,
                                       const float *bootstrap, const float *discounts, float *vs, float *advantages, int64_t B,
                                       int32_t T, float lam, int32_t time_major, void *stream) {
  MBPO_REQUIRE(B >= 0 && T >= 0, MBPO_ERR_ARG, "gae_scan_discounts: negative size");
  if (B == 0 || T == 0) return MBPO_OK;
  MBPO_REQUIRE(truncation && termination && rewards && values && bootstrap && discounts && vs && advantages, MBPO_ERR_ARG,
               "gae_scan_discounts: null pointer");
  ScanArgs A{truncation, termination, rewards, values, bootstrap, discounts, vs, advantages, B, T, 0.f, lam};
  return launch_scan<MODE_GAE>(A, time_major, (hipStream_t)stream, "gae_scan_discounts");
}

extern "C" int mbpo_lambda_return_scan(const float *rewards, const float *next_values, float *returns, int64_t B, int32_t T,
                                       float gamma, float lam, int32_t time_major, void *stream) {
  MBPO_REQUIRE(B >= 0 && T >= 0, MBPO_ERR_ARG, "lambda_return_scan: negative size");
  if (B == 0 || T == 0) return MBPO_OK;
  MBPO_REQUIRE(rewards && next_values && returns, MBPO_ERR_ARG, "lambda_return_scan: null pointer");
  ScanArgs A{nullptr, nullptr, rewards, next_values, nullptr, nullptr, returns, nullptr, B, T, gamma, lam};
  return launch_scan<MODE_LAMBDA>(A, time_major, (hipStream_t)stream, "lambda_return_scan");
}

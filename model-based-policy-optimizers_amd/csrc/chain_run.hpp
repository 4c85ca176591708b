// chain_run.hpp — phase-level runners: one wave group walks a whole MLP chain (forward, input-gradient or weight-gradient)
// with its barriers inside, from shapes that live in SGPRs.
//
// Why (measured on MI355X, scripts/sac_phase_stamps.py + scripts/probes/icache_probe.hip):
//  * a kernel assembled from per-phase inlined copies of the layer routines was 270 KB of straight-line code against a
//    64 KB instruction cache: every step streamed cold code (~1.1 B/cycle);
//  * a step-level interpreter that looked its shapes up in an LDS table was compact but every shape became a VGPR
//    ("divergent") value: hipcc wrapped each weight load in its own exec-mask block and the step got slower;
//  * a lone wave issues ~1 instruction per 4-5 cycles, so a layer step costs what its instruction count costs:
//    32 MFMAs are 1024 cycles, everything else has to stay in the low hundreds of instructions.
// So: shapes are kernel-argument scalars (NetShape), the wave's role comes from readfirstlane, each runner is ONE loop whose
// body holds each layer routine once, weights are requested one layer ahead (WSet, below), and a kernel calls each
// runner from a single call site.  Waves of different chains run different runners between the same barriers: every
// runner executes exactly `n_steps` workgroup barriers.
#pragma once
#include "wave_mlp.hpp"

struct NetShape {
  int K_in;    // network input width
  int L;       // number of Dense layers (>= 2); hidden layers are all 16*HT wide
  int N_out;   // network output width
  int act;
};

// Re-materialise the lane id inside a loop body: hipcc otherwise hoists every per-lane address of every phase out of the
// phase loop (LICM), keeps them all live (256 VGPRs, ~90 spills) and the hot loop pays for the spill traffic.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// ------------------------------------------------------------------------------------------------ weight prefetch
// The next layer's weights are requested one step ahead into the OTHER of two register sets; the sets swap roles by
// unrolling the hidden-layer loop by two.  (A struct copy `cur = next` at the end of a step reads registers whose loads are
// still in flight, i.e. waits for them on the spot — measured: no overlap at all.  Inline-asm loads with a manual
// s_waitcnt are not an option either: hipcc copies/spills asm outputs right after the asm statement, before the data lands.)
// Plain loads keep hipcc's own counted vmcnt(N) waits exact.
typedef float f2v __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4v __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ const float *gaddr(const float *base, unsigned voff) {
  return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + voff);
}
__device__ __forceinline__ void gload1(float &d, const float *base, unsigned voff) { d = *gaddr(base, voff); }
__device__ __forceinline__ void gload2(float *d, const float *base, unsigned voff) {
  f2v t = *reinterpret_cast<const f2v *>(gaddr(base, voff));
  d[0] = t[0]; d[1] = t[1];
}
__device__ __forceinline__ void gload4(float *d, const float *base, unsigned voff) {
  f4v t = *reinterpret_cast<const f4v *>(gaddr(base, voff));
  d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3];
}
template <int NV>
__device__ __forceinline__ void gloadv(float *d, const float *base, unsigned voff) {
  if constexpr (NV == 1) gload1(d[0], base, voff);
  else if constexpr (NV == 2) gload2(d, base, voff);
  else {
    static_assert(NV % 4 == 0, "gloadv width");
#pragma unroll
    for (int c = 0; c < NV / 4; ++c) gload4(d + 4 * c, base, voff + 16u * c);
  }
}

// One lane's share of one layer (NW weights + NB biases) and what it holds.
template <int HT, int SP>
struct WSet {
  static constexpr int KS = 4 * HT, CT = HT / SP, NW = KS * CT;
  float w[NW];
  float b[CT];
};

// (kept as the place where a set's requests are expected to have landed; the compiler places the counted wait itself)
template <int HT, int SP>
__device__ __forceinline__ void wset_wait(WSet<HT, SP> &S) {}

// forward, hidden -> hidden slice: w[s*CT + t] = W[g*KS + s][c0 + CT*r + t], b[t] = bias[c0 + CT*r + t]
template <int HT, int SP>
__device__ __forceinline__ void wset_load_fwd_full(WSet<HT, SP> &S, const float *W, int sub, int lane) {
  constexpr int H = 16 * HT, KS = 4 * HT, CT = HT / SP;
  const int r = lane & 15, g = lane >> 4;
  const unsigned v0 = (unsigned)(((g * KS) * H + sub * 16 * CT + CT * r) * 4);
#pragma unroll
  for (int s = 0; s < KS; ++s) gloadv<CT>(&S.w[s * CT], W, v0 + (unsigned)(s * H * 4));
  gloadv<CT>(S.b, W, (unsigned)((H * H + sub * 16 * CT + CT * r) * 4));
}

// forward, network input layer (K <= 32): w[s*CT + t] = W[clamp(g*kc + s)][c0 + CT*r + t], s < 8
template <int HT, int SP>
__device__ __forceinline__ void wset_load_fwd_in(WSet<HT, SP> &S, const float *W, int K, int sub, int lane) {
  constexpr int H = 16 * HT, CT = HT / SP;
  const int r = lane & 15, g = lane >> 4;
  const int kc = (K + 3) >> 2;
  const unsigned vc = (unsigned)((sub * 16 * CT + CT * r) * 4);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = g * kc + s;
    const int kk = ((s < kc) && (k < K)) ? k : 0;
    gloadv<CT>(&S.w[s * CT], W, vc + (unsigned)(kk * H * 4));
  }
  gloadv<CT>(S.b, W, vc + (unsigned)(K * H * 4));
}

// forward, output layer (K == H, N <= 16*NTL <= 16*CT), wave 0 of the group: w[s*NTL + t] = W[g*KS + s][col(t)]
template <int HT, int SP, int NTL>
__device__ __forceinline__ void wset_load_fwd_out(WSet<HT, SP> &S, const float *W, int N, int lane) {
  constexpr int H = 16 * HT, KS = 4 * HT;
  const int r = lane & 15, g = lane >> 4;
  unsigned vcol[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) vcol[t] = (unsigned)(((NTL * r + t < N) ? NTL * r + t : 0) * 4);
  const unsigned vrow = (unsigned)(g * KS * N * 4);
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < NTL; ++t) gload1(S.w[s * NTL + t], W, vrow + (unsigned)(s * N * 4) + vcol[t]);
#pragma unroll
  for (int t = 0; t < NTL; ++t) gload1(S.b[t], W, (unsigned)(H * N * 4) + vcol[t]);
}

// forward, output layer wider than one n-tile (16 < N <= H): every wave of the group takes its hidden-layer column slice,
// w[s*CT + t] = W[g*KS + s][c0 + CT*r + t] (row stride N; columns >= N read column 0 and are never stored)
template <int HT, int SP>
__device__ __forceinline__ void wset_load_fwd_outw(WSet<HT, SP> &S, const float *W, int N, int sub, int lane) {
  constexpr int H = 16 * HT, KS = 4 * HT, CT = HT / SP;
  const int r = lane & 15, g = lane >> 4;
  unsigned vcol[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int n = sub * 16 * CT + CT * r + t;
    vcol[t] = (unsigned)((n < N ? n : 0) * 4);
  }
  const unsigned vrow = (unsigned)(g * KS * N * 4);
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < CT; ++t) gload1(S.w[s * CT + t], W, vrow + (unsigned)(s * N * 4) + vcol[t]);
#pragma unroll
  for (int t = 0; t < CT; ++t) gload1(S.b[t], W, (unsigned)(H * N * 4) + vcol[t]);
}

// dgrad of a hidden layer (W [H][H]): w[t*NS + n] = W[k0 + CT*r + t][g*NS + n]
template <int HT, int SP>
__device__ __forceinline__ void wset_load_dg_full(WSet<HT, SP> &S, const float *W, int sub, int lane) {
  constexpr int H = 16 * HT, NS = 4 * HT, CT = HT / SP;
  const int r = lane & 15, g = lane >> 4;
  const unsigned v0 = (unsigned)(((sub * 16 * CT + CT * r) * H + g * NS) * 4);
#pragma unroll
  for (int t = 0; t < CT; ++t) gloadv<NS>(&S.w[t * NS], W, v0 + (unsigned)(t * H * 4));
}

// dgrad of the output layer (W [H][N], N <= 4*NSTEP): w[s*CT + t] = W[k0 + CT*r + t][clamp(g*nc + s)], s < NSTEP
template <int HT, int SP, int NSTEP = 8>
__device__ __forceinline__ void wset_load_dg_out(WSet<HT, SP> &S, const float *W, int N, int sub, int lane) {
  constexpr int CT = HT / SP;
  static_assert(NSTEP * CT <= WSet<HT, SP>::NW, "output-layer dgrad image does not fit the register set");
  const int r = lane & 15, g = lane >> 4;
  const int nc = (N + 3) >> 2;
  const unsigned vrow = (unsigned)((sub * 16 * CT + CT * r) * N * 4);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int n = g * nc + s;
    const int nn = ((s < nc) && (n < N)) ? n : 0;
#pragma unroll
    for (int t = 0; t < CT; ++t) gload1(S.w[s * CT + t], W, vrow + (unsigned)((t * N + nn) * 4));
  }
}

// dgrad of layer 0 towards the network input (W [K][H], K <= 32): wave `sub` takes input rows 16*sub .. 16*sub+15,
// w[n] = W[clamp(16*sub + r)][g*NS + n]
template <int HT, int SP>
__device__ __forceinline__ void wset_load_dg_in(WSet<HT, SP> &S, const float *W, int K, int sub, int lane) {
  constexpr int H = 16 * HT, NS = 4 * HT;
  const int r = lane & 15, g = lane >> 4;
  const int k = 16 * sub + r;
  gloadv<NS>(S.w, W, (unsigned)((((k < K) ? k : 0) * H + g * NS) * 4));
}

// ---- computes on a register image ----------------------------------------------------------------------------
// hidden slice from a full (K == H) or input (K <= 32) image
// JVP (forward-mode tangent riding along, SAC actor loss): a SECOND 16-row tile goes through the same weights in the same wave —
// the tangent d(.)/d(input k_tan) of every row.  th_in/th_out: tangent of the previous / this layer's activations (same tile
// layout); the lane that holds z[row][col] also holds the tangent pre-activation of that (row, col), so
// h' = act'(z) * z' is formed in registers, with no exchange between waves.  Layer 0 (IN): the tangent input is the one-hot e_{k_tan}
// for every row.  th_out == nullptr: no tangent (the only form the other kernels instantiate).
template <int HT, int SP, bool IN, bool JVP = false>
__device__ __forceinline__ void wset_fwd_hidden(const WSet<HT, SP> &S, const float *x, int ldx, int K, float *h_out, float *z_out,
                                                int ldo, int act, int lane, const float *th_in = nullptr, float *th_out = nullptr,
                                                int k_tan = 0) {
  constexpr int KS = 4 * HT, CT = HT / SP;
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc[CT];
  f32x4 tacc[JVP ? CT : 1];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (JVP) {
#pragma unroll
    for (int t = 0; t < CT; ++t) tacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (!IN) {
    const float *xr = x + r * ldx + g * KS;
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) {
      float av[4];
      load_vec_lds<4>(xr + 4 * q, av);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], S.w[(4 * q + u) * CT + t], acc[t], 0, 0, 0);
    }
    if constexpr (JVP) {
      if (th_out) {
        const float *tr = th_in + r * ldx + g * KS;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          float av[4];
          load_vec_lds<4>(tr + 4 * q, av);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < CT; ++t) tacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], S.w[(4 * q + u) * CT + t], tacc[t], 0, 0, 0);
        }
      }
    }
  } else {
    const int kc = (K + 3) >> 2;
    float av[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = g * kc + s;
      const bool ok = (s < kc) && (k < K);
      av[s] = x[r * ldx + (ok ? k : 0)];
      av[s] = ok ? av[s] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], S.w[s * CT + t], acc[t], 0, 0, 0);
    if constexpr (JVP) {
      if (th_out) {
        // tangent input e_{k_tan} in every row: z'_0[row][n] = W0[k_tan][n]
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const float one = ((s < kc) && (g * kc + s == k_tan)) ? 1.f : 0.f;
#pragma unroll
          for (int t = 0; t < CT; ++t) tacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(one, S.w[s * CT + t], tacc[t], 0, 0, 0);
        }
      }
    }
  }
  // epilogue: the (uniform) activation / z-store decisions are taken once, not per row, and the 4*CT values of a lane go
  // through the activation together (independent transcendental ops pipeline instead of serialising per row)
  float zv[4 * CT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < CT; ++t) zv[i * CT + t] = acc[t][i] + S.b[t];
  const int o0 = (4 * g) * ldo + CT * r;
  if (z_out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) store_vec_lds<CT>(z_out + o0 + i * ldo, *reinterpret_cast<float(*)[CT]>(&zv[i * CT]));
  }
  if constexpr (JVP) {
    if (th_out) {
      float tv[4 * CT];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) tv[i * CT + t] = tacc[t][i];
      if (act == MBPO_ACT_SWISH) {
        // one sigmoid per element serves both: h = z * sg, h' = sg * (1 + z * (1 - sg)) * z'  (the same expressions as
        // act_apply / act_grad, evaluated once)
#pragma unroll
        for (int i = 0; i < 4 * CT; ++i) {
          const float sg = fast_sigmoid(zv[i]);
          tv[i] *= sg * (1.0f + zv[i] * (1.0f - sg));
          zv[i] = zv[i] * sg;
        }
      } else {
        act_grad_mul_vec<4 * CT>(tv, zv, act);      // h' = act'(z) * z'
        act_apply_vec<4 * CT>(zv, act);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) store_vec_lds<CT>(th_out + o0 + i * ldo, *reinterpret_cast<float(*)[CT]>(&tv[i * CT]));
#pragma unroll
      for (int i = 0; i < 4; ++i) store_vec_lds<CT>(h_out + o0 + i * ldo, *reinterpret_cast<float(*)[CT]>(&zv[i * CT]));
      return;
    }
  }
  act_apply_vec<4 * CT>(zv, act);
#pragma unroll
  for (int i = 0; i < 4; ++i) store_vec_lds<CT>(h_out + o0 + i * ldo, *reinterpret_cast<float(*)[CT]>(&zv[i * CT]));
}

template <int HT, int SP, int NTL, bool JVP = false>
__device__ __forceinline__ void wset_fwd_out(const WSet<HT, SP> &S, const float *x, int ldx, int N, float *y, int ldy, int lane,
                                             const float *th_in = nullptr, float *ty = nullptr, int ldty = 0) {
  constexpr int KS = 4 * HT;
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *xr = x + r * ldx + g * KS;
#pragma unroll
  for (int q = 0; q < KS / 4; ++q) {
    float av[4];
    load_vec_lds<4>(xr + 4 * q, av);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < NTL; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], S.w[(4 * q + u) * NTL + t], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const int n = NTL * r + t;
    if (n < N) {
#pragma unroll
      for (int i = 0; i < 4; ++i) y[(4 * g + i) * ldy + n] = acc[t][i] + S.b[t];
    }
  }
  if constexpr (JVP) {
    if (ty) {      // tangent of the output: y' = h' W (no bias)
      f32x4 tacc[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t) tacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float *tr = th_in + r * ldx + g * KS;
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) {
        float av[4];
        load_vec_lds<4>(tr + 4 * q, av);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < NTL; ++t) tacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], S.w[(4 * q + u) * NTL + t], tacc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int n = NTL * r + t;
        if (n < N) {
#pragma unroll
          for (int i = 0; i < 4; ++i) ty[(4 * g + i) * ldty + n] = tacc[t][i];
        }
      }
    }
  }
}

template <int HT, int SP>
__device__ __forceinline__ void wset_fwd_outw(const WSet<HT, SP> &S, const float *x, int ldx, int N, float *y, int ldy, int sub, int lane) {
  constexpr int KS = 4 * HT, CT = HT / SP;
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *xr = x + r * ldx + g * KS;
#pragma unroll
  for (int q = 0; q < KS / 4; ++q) {
    float av[4];
    load_vec_lds<4>(xr + 4 * q, av);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], S.w[(4 * q + u) * CT + t], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int n = sub * 16 * CT + CT * r + t;
    if (n < N) {
#pragma unroll
      for (int i = 0; i < 4; ++i) y[(4 * g + i) * ldy + n] = acc[t][i] + S.b[t];
    }
  }
}

// delta_{l-1}[:, k0 .. k0+16*CT) from a hidden (FULL) or output-layer (OUT: N <= 4*NSTEP) dgrad image
template <int HT, int SP, bool OUT, int NSTEP = 8>
__device__ __forceinline__ void wset_dgrad(const WSet<HT, SP> &S, const float *delta, int ldd, int N, const float *zp, float *dx, int ldh,
                                           int act, int lane) {
  constexpr int NS = 4 * HT, CT = HT / SP;
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (!OUT) {
    const float *dr = delta + r * ldd + g * NS;
#pragma unroll
    for (int q = 0; q < NS / 4; ++q) {
      float av[4];
      load_vec_lds<4>(dr + 4 * q, av);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], S.w[t * NS + 4 * q + u], acc[t], 0, 0, 0);
    }
  } else {
    const int nc = (N + 3) >> 2;
    float av[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int n = g * nc + s;
      const bool ok = (s < nc) && (n < N);
      av[s] = delta[r * ldd + (ok ? n : 0)];
      av[s] = ok ? av[s] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], S.w[s * CT + t], acc[t], 0, 0, 0);
  }
  float ov[4 * CT], zv[4 * CT];
  const int o0 = (4 * g) * ldh + CT * r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    load_vec_lds<CT>(zp + o0 + i * ldh, *reinterpret_cast<float(*)[CT]>(&zv[i * CT]));
#pragma unroll
    for (int t = 0; t < CT; ++t) ov[i * CT + t] = acc[t][i];
  }
  act_grad_mul_vec<4 * CT>(ov, zv, act);
#pragma unroll
  for (int i = 0; i < 4; ++i) store_vec_lds<CT>(dx + o0 + i * ldh, *reinterpret_cast<float(*)[CT]>(&ov[i * CT]));
}

template <int HT, int SP>
__device__ __forceinline__ void wset_dgrad_in(const WSet<HT, SP> &S, const float *delta, int ldd, int K, float *dX, int ld_dx, int sub,
                                              int lane) {
  constexpr int NS = 4 * HT;
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *dr = delta + r * ldd + g * NS;
#pragma unroll
  for (int q = 0; q < NS / 4; ++q) {
    float av[4];
    load_vec_lds<4>(dr + 4 * q, av);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], S.w[4 * q + u], acc, 0, 0, 0);
  }
  const int k = 16 * sub + r;
  if (k < K) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dX[(4 * g + i) * ld_dx + k] = acc[i];
  }
}

// ------------------------------------------------------------------------------------------------ fast-path predicates
// WIDE (compile-time) selects which shapes take the register-image path:
//   !WIDE: networks whose input and output fit one 16-column tile where it matters (K_in <= 32 forward, <= 16 for an input
//          gradient; N_out <= 16 on wave 0, or <= 32 when a wave owns 2 column tiles).  The lean code every x<=16 kernel runs.
//    WIDE: additionally the output layer on all waves' hidden-layer column slices (N_out <= H), its dgrad through a 16-step
//          image (N_out <= 64), the input gradient in two row tiles (K_in <= 32), wider weight-gradient tiles.  More code and
//          registers in every kernel that instantiates it, so the host picks the WIDE kernel only for such shapes.
template <int HT, int SP, bool WIDE = false>
__device__ __forceinline__ bool fast_shape(const NetShape &sh) {
  if constexpr (WIDE) return (4 * HT * (HT / SP) <= 64) && sh.K_in <= 32 && sh.N_out <= 16 * HT && sh.N_out <= 64 && sh.L >= 2;
  else return (4 * HT * (HT / SP) <= 64) && sh.K_in <= 32 && sh.N_out <= 16 * (HT / SP) && sh.N_out <= 32 && sh.L >= 2;
}

// host side: does a network need the WIDE kernel variants (anything beyond one 16-column tile at its input or output)?
static inline bool net_is_wide(const NetShape &sh) { return sh.L > 0 && (sh.K_in > 16 || sh.N_out > 16); }

// First-layer request of a forward / dgrad phase (issued by the kernel before the preceding elementwise section).
template <int HT, int SP, bool WIDE = false>
__device__ __forceinline__ void chain_fwd_prefetch(WSet<HT, SP> &A, const NetShape sh, const float *params, int sub, int lane) {
  if (fast_shape<HT, SP, WIDE>(sh)) wset_load_fwd_in<HT, SP>(A, params, sh.K_in, sub, lane);
}
template <int HT, int SP, bool WIDE = false>
__device__ __forceinline__ void chain_dgrad_prefetch(WSet<HT, SP> &A, const NetShape sh, const float *params, int sub, int lane) {
  constexpr int H = 16 * HT;
  if (fast_shape<HT, SP, WIDE>(sh)) {
    const float *Wl = params + (sh.K_in * H + H) + (sh.L - 2) * (H * H + H);
    if (!WIDE || sh.N_out <= 32) wset_load_dg_out<HT, SP, 8>(A, Wl, sh.N_out, sub, lane);
    else if constexpr (WIDE) wset_load_dg_out<HT, SP, 16>(A, Wl, sh.N_out, sub, lane);
  }
}

// request for layer ln (>= 1) of a forward chain: a hidden image, or the output image for the wave that computes it
template <int HT, int SP, bool WIDE = false>
__device__ __forceinline__ void fwd_request(WSet<HT, SP> &S, const float *W1, int ln, int L, int N_out, int sub, int lane) {
  constexpr int H = 16 * HT;
  const float *Wn = W1 + (ln - 1) * (H * H + H);
  if (ln < L - 1) wset_load_fwd_full<HT, SP>(S, Wn, sub, lane);
  else if (ln == L - 1) {
    if (!WIDE || N_out <= 16) {
      if (sub == 0) wset_load_fwd_out<HT, SP, 1>(S, Wn, N_out, lane);   // one n-tile image on wave 0
    } else if (sub * 16 * (HT / SP) < N_out) {
      if constexpr (WIDE) wset_load_fwd_outw<HT, SP>(S, Wn, N_out, sub, lane);
    }
  }
}

// request for layer ln's dgrad image (ln <= L-2): hidden layers, or the input layer when the input gradient is wanted
template <int HT, int SP>
__device__ __forceinline__ void dgrad_request(WSet<HT, SP> &S, const float *params, const float *W1, int ln, int K_in, bool want_dx,
                                              int sub, int lane) {
  constexpr int H = 16 * HT;
  if (ln >= 1) wset_load_dg_full<HT, SP>(S, W1 + (ln - 1) * (H * H + H), sub, lane);
  else if (ln == 0 && want_dx && 16 * sub < K_in) wset_load_dg_in<HT, SP>(S, params, K_in, sub, lane);
}

// ------------------------------------------------------------------------------------------------ thin layers by VALU
// A layer with K_in <= 8 inputs or N_out <= 4 outputs is 1/16 .. 1/8 of a hidden layer's arithmetic but cost a whole layer step
// (request, LDS round trip, MFMA latency, epilogue, barrier: ~2 k cycles).  In `thin` mode (H == 64, SP == 4) the chain's four
// waves do such a layer with plain FMAs — wave `sub` rows 4*sub .. 4*sub+3, lane = hidden column — in front of / behind the
// runner, which then walks the H x H layers only.  Operands come from LDS by broadcast (x, dy) or coalesced (z, h, delta) reads.
#define THIN_KMAX 8
#define THIN_NMAX 4

// forward: layer 1's image into A, this lane's column of layer 0 (K_in weights + bias) into tw
template <int HT, int SP, bool WIDE = false>
__device__ __forceinline__ void chain_fwd_prefetch_thin(WSet<HT, SP> &A, const NetShape sh, const float *params, int sub, int lane,
                                                        float (&tw)[THIN_KMAX + 1]) {
  constexpr int H = 16 * HT;
  fwd_request<HT, SP, WIDE>(A, params + sh.K_in * H + H, 1, sh.L, sh.N_out, sub, lane);
#pragma unroll
  for (int k = 0; k < THIN_KMAX; ++k) {
    const float w = params[(k < sh.K_in ? k : 0) * H + lane];
    tw[k] = k < sh.K_in ? w : 0.f;
  }
  tw[THIN_KMAX] = params[sh.K_in * H + lane];
}
// dgrad: layer L-2's image into A, this lane's row of the output layer (N_out weights) into tw
template <int HT, int SP>
__device__ __forceinline__ void chain_dgrad_prefetch_thin(WSet<HT, SP> &A, const NetShape sh, const float *params, int sub, int lane,
                                                          float (&tw)[THIN_KMAX + 1]) {
  constexpr int H = 16 * HT;
  const float *W1 = params + sh.K_in * H + H;
  wset_load_dg_full<HT, SP>(A, W1 + (sh.L - 3) * (H * H + H), sub, lane);      // layer L-2 (>= 1)
  const float *Wo = W1 + (sh.L - 2) * (H * H + H);                             // [H][N_out]
#pragma unroll
  for (int o = 0; o < THIN_NMAX; ++o) {
    const float w = Wo[lane * sh.N_out + (o < sh.N_out ? o : 0)];
    tw[o] = o < sh.N_out ? w : 0.f;
  }
}

// layer 0 of a forward chain: h0 = act(x W0 + b0) for rows 4*sub .. 4*sub+3, column = lane; z0 and the tangent
// d h0 / d x[k_tan] = act'(z0) W0[k_tan][col] are stored when asked for (tile layout [16][ldh]).
// All LDS operands are requested up front (two 16-byte reads per row: the rows of an input tile are >= 8 floats apart and
// 16-byte aligned); columns >= K_in are replaced by 0 (they are not initialised), the matching weights are loaded as 0.
__device__ __forceinline__ void thin_fwd_first(const float (&tw)[THIN_KMAX + 1], int K_in, int act, const float *x, int ldx, float *h0,
                                               float *z0, float *t0, int k_tan, int ldh, int sub, int lane) {
  float xv[4][THIN_KMAX];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    load_vec_lds<4>(x + (4 * sub + i) * ldx, *reinterpret_cast<float(*)[4]>(&xv[i][0]));
    load_vec_lds<4>(x + (4 * sub + i) * ldx + 4, *reinterpret_cast<float(*)[4]>(&xv[i][4]));
  }
  float wt = 0.f;
#pragma unroll
  for (int k = 0; k < THIN_KMAX; ++k) wt = (k == k_tan) ? tw[k] : wt;
  float zv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float z = tw[THIN_KMAX];
#pragma unroll
    for (int k = 0; k < THIN_KMAX; ++k) z = fmaf(k < K_in ? xv[i][k] : 0.f, tw[k], z);
    zv[i] = z;
  }
  if (z0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) z0[(4 * sub + i) * ldh + lane] = zv[i];
  }
  if (t0) {
    float tv[4] = {wt, wt, wt, wt};
    act_grad_mul_vec<4>(tv, zv, act);
#pragma unroll
    for (int i = 0; i < 4; ++i) t0[(4 * sub + i) * ldh + lane] = tv[i];
  }
  act_apply_vec<4>(zv, act);
#pragma unroll
  for (int i = 0; i < 4; ++i) h0[(4 * sub + i) * ldh + lane] = zv[i];
}

// delta_{L-2} = (dY Wout^T) * act'(z_{L-2}) for rows 4*sub .. 4*sub+3, column = lane (one 16-byte read of dY per row)
__device__ __forceinline__ void thin_dgrad_out(const float (&tw)[THIN_KMAX + 1], int N_out, int act, const float *dY, int ldy,
                                               const float *z, float *d_out, int ldh, int sub, int lane) {
  float dv[4][THIN_NMAX], zv[4], sv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    load_vec_lds<4>(dY + (4 * sub + i) * ldy, dv[i]);
    zv[i] = z[(4 * sub + i) * ldh + lane];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < THIN_NMAX; ++o) s = fmaf(o < N_out ? dv[i][o] : 0.f, tw[o], s);
    sv[i] = s;
  }
  act_grad_mul_vec<4>(sv, zv, act);
#pragma unroll
  for (int i = 0; i < 4; ++i) d_out[(4 * sub + i) * ldh + lane] = sv[i];
}

// output layer's weight gradient dW[c][o] = sum_r h[r][c] dY[r][o] (wave `sub` takes o = sub), db[o] = sum_r dY[r][o] (wave 0)
template <int H>
__device__ __forceinline__ void thin_wgrad_out(int N_out, const float *h, int ldh, const float *dY, int ldy, float *__restrict__ gW,
                                               int sub, int lane) {
  if (sub < N_out) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = fmaf(h[r * ldh + lane], dY[r * ldy + sub], acc);
    gW[lane * N_out + sub] = acc;
  }
  if (sub == 0 && lane < N_out) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += dY[r * ldy + lane];
    gW[H * N_out + lane] = acc;
  }
}

// layer 0's weight gradient dW0[k][c] = sum_r x[r][k] delta0[r][c] and db0[c] = sum_r delta0[r][c], column c = lane: wave `w8`
// of the eight waves of a dgrad / wgrad pair takes input row k = w8 (< K_in <= 8); the bias goes to wave K_in, or, when all eight
// carry a row, to wave 7 as well
template <int H>
__device__ __forceinline__ void thin_wgrad_first(int K_in, const float *x, int ldx, const float *d0, int ldh, float *__restrict__ gW,
                                                 int w8, int lane) {
  float dv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dv[r] = d0[r * ldh + lane];
  if (w8 < K_in) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = fmaf(x[r * ldx + w8], dv[r], acc);
    gW[w8 * H + lane] = acc;
  }
  if (w8 == (K_in < 8 ? K_in : 7)) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += dv[r];
    gW[K_in * H + lane] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ forward runner
// x: network input tile [16][ldx].  Hidden outputs go to hbuf + l*T (when hbuf) or ping-pong pp0/pp1; pre-activations to
// zbuf + l*T (when zbuf); the output layer to y [16][ldy].  A must hold layer 0's request (chain_fwd_prefetch).
// Executes exactly n_steps workgroup barriers.
// JVP: tp0 / tp1 (ping-pong hidden tiles for the tangent), ty [16][ldty] (tangent of the output) and k_tan (the input the tangent is
// taken with respect to); tp0 == nullptr: no tangent.  Register-image path with a one-tile output only.
template <int HT, int SP, bool WIDE = false, bool JVP = false, bool THIN = false>
__device__ __forceinline__ void chain_fwd_run(const NetShape sh, const float *__restrict__ params, const float *x, int ldx, float *pp0,
                                              float *pp1, float *zbuf, float *hbuf, float *y, int ldy, int ldh, int n_steps, int sub,
                                              int lane_, WSet<HT, SP> &A, unsigned long long *dbg = nullptr, float *tp0 = nullptr,
                                              float *tp1 = nullptr, float *ty = nullptr, int ldty = 0, int k_tan = 0) {
  constexpr bool thin = THIN;
  // thin (register-image path, L >= 3): the caller has computed layer 0 itself (thin_fwd_first: a K_in <= 8 layer is a handful
  // of VALU FMAs per output, not worth a layer step with its barrier) and A holds layer 1's image (chain_fwd_prefetch_thin);
  // the runner starts at layer 1 and executes one barrier fewer.
  constexpr int H = 16 * HT, CT = HT / SP;
  const int T = 16 * ldh, c0 = sub * 16 * CT;
  const int L = sh.L;
#define DBG_STAMP(i)                                                                 \
  if (dbg && lane_ == 0) {                                                           \
    unsigned long long t_;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
    dbg[i] = t_;                                                                     \
  }
  // (macros, not [&] lambdas: a by-reference closure kept in memory pins every captured variable — and the register sets
  //  passed through it — to scratch)
#define hout(l) ((hbuf ? hbuf + (l) * T : (((l) & 1) ? pp1 : pp0)) + c0)
#define zout(l) (zbuf ? zbuf + (l) * T + c0 : nullptr)
#define hin(l) ((const float *)(hbuf ? hbuf + ((l) - 1) * T : ((((l) - 1) & 1) ? pp1 : pp0)))   /* l >= 1 */
#define tout(l) (tp0 ? (((l) & 1) ? tp1 : tp0) + c0 : nullptr)
#define tin(l) ((const float *)((((l) - 1) & 1) ? tp1 : tp0))                                     /* l >= 1 */
  if (fast_shape<HT, SP, WIDE>(sh)) {
    if constexpr (4 * HT * (HT / SP) <= 64) {
      WSet<HT, SP> B;
      const float *W1 = params + sh.K_in * H + H;   // layer 1
#define request(S, ln, lane) fwd_request<HT, SP, WIDE>(S, W1, ln, L, sh.N_out, sub, lane)
      // ---- layer 0 (input image in A) ----
      if constexpr (!thin) {
        const int lane = opaque(lane_);
        wset_wait(A);
        request(B, 1, lane);
        wset_fwd_hidden<HT, SP, true, JVP>(A, x, ldx, sh.K_in, hout(0), zout(0), ldh, sh.act, lane, nullptr, tout(0), k_tan);
        __syncthreads();
      } else {
        wset_wait(A);
        B = A;      // layer 1's image was requested into A; the loop below expects it in B (a few register moves)
      }
      // ---- hidden layers 1 .. L-2, two per trip: B then A ----
      int l = 1;
#pragma nounroll
      for (; l + 1 <= L - 2; l += 2) {
        const int lane = opaque(lane_);
        DBG_STAMP(0);
        wset_wait(B);
        request(A, l + 1, lane);
        DBG_STAMP(1);
        wset_fwd_hidden<HT, SP, false, JVP>(B, hin(l), ldh, H, hout(l), zout(l), ldh, sh.act, lane, tin(l), tout(l));
        DBG_STAMP(2);
        __syncthreads();
        DBG_STAMP(3);
        wset_wait(A);
        request(B, l + 2, lane);
        DBG_STAMP(4);
        wset_fwd_hidden<HT, SP, false, JVP>(A, hin(l + 1), ldh, H, hout(l + 1), zout(l + 1), ldh, sh.act, lane, tin(l + 1), tout(l + 1));
        DBG_STAMP(5);
        __syncthreads();
        DBG_STAMP(6);
      }
      bool in_a = false;
      if (l <= L - 2) {   // one hidden layer left: it is in B, the output image goes to A
        const int lane = opaque(lane_);
        wset_wait(B);
        request(A, l + 1, lane);
        wset_fwd_hidden<HT, SP, false, JVP>(B, hin(l), ldh, H, hout(l), zout(l), ldh, sh.act, lane, tin(l), tout(l));
        __syncthreads();
        ++l;
        in_a = true;
      }
      // ---- output layer L-1 (image in B, or in A after an odd number of hidden layers; no register copies) ----
      if (!WIDE || sh.N_out <= 16) {
        if (sub == 0) {
          const int lane = opaque(lane_);
          if (sh.N_out <= 16) {
            if (in_a) wset_fwd_out<HT, SP, 1, JVP>(A, hin(L - 1), ldh, sh.N_out, y, ldy, lane, tp0 ? tin(L - 1) : nullptr, ty, ldty);
            else wset_fwd_out<HT, SP, 1, JVP>(B, hin(L - 1), ldh, sh.N_out, y, ldy, lane, tp0 ? tin(L - 1) : nullptr, ty, ldty);
          } else {
            if constexpr (!WIDE && CT >= 2) gen_dense_fwd(hin(L - 1), ldh, H, W1 + (L - 2) * (H * H + H), sh.N_out,
                                                          W1 + (L - 2) * (H * H + H) + H * sh.N_out, 0, sh.N_out, y, nullptr, ldy, -1, lane);
          }
        }
      } else if (c0 < sh.N_out) {
        if constexpr (WIDE) {
          const int lane = opaque(lane_);
          if (in_a) wset_fwd_outw<HT, SP>(A, hin(L - 1), ldh, sh.N_out, y, ldy, sub, lane);
          else wset_fwd_outw<HT, SP>(B, hin(L - 1), ldh, sh.N_out, y, ldy, sub, lane);
        }
      }
      __syncthreads();
    }
  } else {
    // shapes without register images: self-loading routines (banked for wide hidden layers, generic otherwise)
    const float *W = params;
    const float *cur = x;
    int ldc = ldx;
#pragma nounroll
    for (int l = 0; l < L; ++l) {
      const int lane = opaque(lane_);
      const int K = (l == 0) ? sh.K_in : H;
      const bool last = (l == L - 1);
      const int N = last ? sh.N_out : H;
      if (!last) {
        float *ho = hout(l), *zo = zout(l);
        if (K == H) wave_dense_fwd<CT, 4 * HT>(cur, ldc, K, W + c0, H, W + K * H + c0, 16 * CT, ho, zo, ldh, sh.act, lane);
        else if (K <= 32) wave_dense_fwd<CT>(cur, ldc, K, W + c0, H, W + K * H + c0, 16 * CT, ho, zo, ldh, sh.act, lane);
        else gen_dense_fwd(cur, ldc, K, W, H, W + K * H, c0, c0 + 16 * CT, ho - c0, zo ? zo - c0 : nullptr, ldh, sh.act, lane);
        cur = ho - c0;
        ldc = ldh;
      } else if (sub == 0) {
        if (N <= 16) wave_dense_fwd<1, 4 * HT>(cur, ldc, K, W, N, W + K * N, N, y, nullptr, ldy, -1, lane);
        else gen_dense_fwd(cur, ldc, K, W, N, W + K * N, 0, N, y, nullptr, ldy, -1, lane);
      }
      __syncthreads();
      W += K * N + N;
    }
  }
#pragma nounroll
  for (int l = L - (thin ? 1 : 0); l < n_steps; ++l) __syncthreads();
#undef hout
#undef zout
#undef hin
#undef tout
#undef tin
#undef request
#undef DBG_STAMP
}

// ------------------------------------------------------------------------------------------------ dgrad runner
// dY [16][ldy] is the delta of the output layer; deltas of hidden layers ping-pong through d0/d1 (layer l's dgrad writes
// delta_{l-1} to (l&1 ? d1 : d0)); zbuf from the forward (layer l at + l*T); dX (optional) receives the input gradient.
// A must hold layer L-1's request (chain_dgrad_prefetch).  Executes exactly n_steps workgroup barriers.
template <int HT, int SP, bool WIDE = false, bool THIN = false>
__device__ __forceinline__ void chain_dgrad_run(const NetShape sh, const float *__restrict__ params, const float *dY, int ldy,
                                                const float *zbuf, float *d0, float *d1, float *dX, int ld_dx, int ldh, int n_steps,
                                                int sub, int lane_, WSet<HT, SP> &A) {
  constexpr bool thin = THIN;
  // thin (register-image path, L >= 3, no dX): the caller has formed delta_{L-2} itself (thin_dgrad_out: an N_out <= 4 output
  // layer is a few FMAs per element) and A holds layer L-2's image (chain_dgrad_prefetch_thin); layers L-2 .. 1 run here:
  // L - 2 barriers.
  constexpr int H = 16 * HT, CT = HT / SP;
  const int T = 16 * ldh, k0 = sub * 16 * CT;
  const int L = sh.L;
#define din(l) ((const float *)((((l) + 1) & 1) ? d1 : d0))   /* delta_l for l < L-1 */
#define dout(l) ((((l) & 1) ? d1 : d0) + k0)                 /* delta_{l-1}, this wave's slice */
#define zprev(l) (zbuf + ((l) - 1) * T + k0)
  const float *W1 = params + sh.K_in * H + H;   // layer 1
  const bool fast = fast_shape<HT, SP, WIDE>(sh) && (WIDE || dX == nullptr || sh.K_in <= 16);
  if (fast) {
    if constexpr (4 * HT * (HT / SP) <= 64) {
      WSet<HT, SP> B;
#define request(S, ln, lane) dgrad_request<HT, SP>(S, params, W1, ln, sh.K_in, dX != nullptr, sub, lane)
      // ---- output layer L-1 (image in A) ----
      if constexpr (thin) {
        wset_wait(A);
        B = A;
      } else {
        const int lane = opaque(lane_);
        wset_wait(A);
        request(B, L - 2, lane);
        if (!WIDE || sh.N_out <= 32) wset_dgrad<HT, SP, true, 8>(A, dY, ldy, sh.N_out, zprev(L - 1), dout(L - 1), ldh, sh.act, lane);
        else if constexpr (WIDE) wset_dgrad<HT, SP, true, 16>(A, dY, ldy, sh.N_out, zprev(L - 1), dout(L - 1), ldh, sh.act, lane);
        __syncthreads();
      }
      // ---- hidden layers L-2 .. 1, two per trip: B then A ----
      int l = L - 2;
#pragma nounroll
      for (; l - 1 >= 1; l -= 2) {
        const int lane = opaque(lane_);
        wset_wait(B);
        request(A, l - 1, lane);
        wset_dgrad<HT, SP, false>(B, din(l), ldh, H, zprev(l), dout(l), ldh, sh.act, lane);
        __syncthreads();
        wset_wait(A);
        request(B, l - 2, lane);
        wset_dgrad<HT, SP, false>(A, din(l - 1), ldh, H, zprev(l - 1), dout(l - 1), ldh, sh.act, lane);
        __syncthreads();
      }
      bool in_a = false;
      if (l >= 1) {   // one hidden layer left: it is in B, the input image (if any) goes to A
        const int lane = opaque(lane_);
        wset_wait(B);
        request(A, l - 1, lane);
        wset_dgrad<HT, SP, false>(B, din(l), ldh, H, zprev(l), dout(l), ldh, sh.act, lane);
        __syncthreads();
        --l;
        in_a = true;
      }
      // ---- layer 0: input gradient (image in B, or in A after an odd number of hidden layers) ----
      if (dX && 16 * sub < sh.K_in) {
        const int lane = opaque(lane_);
        if (in_a) wset_dgrad_in<HT, SP>(A, din(0), ldh, sh.K_in, dX, ld_dx, sub, lane);
        else wset_dgrad_in<HT, SP>(B, din(0), ldh, sh.K_in, dX, ld_dx, sub, lane);
      }
      if (!thin) __syncthreads();
    }
  } else {
    const float *W = W1 + (L - 2) * (H * H + H);   // layer L-1
    const float *delta = dY;
    int ldd = ldy;
#pragma nounroll
    for (int l = L - 1; l >= 0; --l) {
      const int lane = opaque(lane_);
      const int N = (l == L - 1) ? sh.N_out : H;
      if (l > 0) {
        if (N == H) wave_dense_dgrad<CT, 4 * HT>(delta, ldd, N, W + k0 * N, N, 16 * CT, zprev(l), ldh, sh.act, dout(l), ldh, lane);
        else if (N <= 32) wave_dense_dgrad<CT>(delta, ldd, N, W + k0 * N, N, 16 * CT, zprev(l), ldh, sh.act, dout(l), ldh, lane);
        else gen_dense_dgrad(delta, ldd, N, W, N, k0, k0 + 16 * CT, zbuf + (l - 1) * T, ldh, sh.act, dout(l) - k0, ldh, lane);
      } else if (dX && sub == 0) {
        if (sh.K_in <= 16) wave_dense_dgrad<1, 4 * HT>(delta, ldd, H, W, H, sh.K_in, nullptr, 0, 0, dX, ld_dx, lane);
        else gen_dense_dgrad(delta, ldd, H, W, H, 0, sh.K_in, nullptr, 0, 0, dX, ld_dx, lane);
      }
      __syncthreads();
      delta = dout(l) - k0;
      ldd = ldh;
      W -= (l - 1 == 0) ? (sh.K_in * H + H) : (H * H + H);
    }
  }
#pragma nounroll
  for (int l = L - (thin ? 2 : 0); l < n_steps; ++l) __syncthreads();
#undef din
#undef dout
#undef zprev
#undef request
}

// ------------------------------------------------------------------------------------------------ fast weight gradients
// dW block [16*KT rows][16*NT cols] = A^T . delta over the 16 batch rows, plus (BIAS) db = sum_rows delta through one more
// MFMA tile whose A operand is the indicator of tile-row 0 (no LDS reduction, no extra reads).
//   A image: a_src[row*lda + KT*r + a], a < KT (k = KT*rho + a: a lane's KT results for a tile row are consecutive k)
//   B image: delta[row*ldd + NT*r + b], b < NT (n = NT*j + b)
// Bounds: k_valid / n_valid columns are real (the rest reads as zero and is not stored).  gW points at the block's first
// element, row stride ldw.  With `accumulate` the old values are requested before the MFMAs and added at the end.
template <int KT, int NT, bool BIAS>
__device__ __forceinline__ void wgrad_tile_fast(const float *a_src, int lda, int k_valid, const float *delta, int ldd, int n_valid,
                                                float *__restrict__ gW, int ldw, float *__restrict__ gb, int lane, bool accumulate) {
  const int r = lane & 15, g = lane >> 4;
  const bool full = (k_valid == 16 * KT) && (n_valid == 16 * NT);
  f32x4 acc[KT][NT];
  f32x4 accb[NT];
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < NT; ++b) accb[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float old[KT][4][NT];
  float oldb[NT];
  if (accumulate) {
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = KT * (4 * g + i) + a;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int n = NT * r + b;
          old[a][i][b] = (k < k_valid && n < n_valid) ? gW[k * ldw + n] : 0.f;
        }
      }
    if (BIAS) {
#pragma unroll
      for (int b = 0; b < NT; ++b) oldb[b] = (NT * r + b < n_valid) ? gb[NT * r + b] : 0.f;
    }
  }
  const float one0 = (r == 0) ? 1.f : 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = 4 * g + s;
    float av[KT], bv[NT];
    if (full) {
      load_vec_lds<KT>(a_src + row * lda + KT * r, av);
      load_vec_lds<NT>(delta + row * ldd + NT * r, bv);
    } else {
#pragma unroll
      for (int a = 0; a < KT; ++a) {
        const int k = KT * r + a;
        av[a] = a_src[row * lda + (k < k_valid ? k : 0)];
        av[a] = k < k_valid ? av[a] : 0.f;
      }
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int n = NT * r + b;
        bv[b] = delta[row * ldd + (n < n_valid ? n : 0)];
        bv[b] = n < n_valid ? bv[b] : 0.f;
      }
    }
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    if (BIAS) {
#pragma unroll
      for (int b = 0; b < NT; ++b) accb[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(one0, bv[b], accb[b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = KT * (4 * g + i) + a;
      float ov[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) ov[b] = acc[a][b][i] + (accumulate ? old[a][i][b] : 0.f);
      if (full) {
        store_vec_global<NT>(gW + k * ldw + NT * r, ov);
      } else if (k < k_valid) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
          if (NT * r + b < n_valid) gW[k * ldw + NT * r + b] = ov[b];
      }
    }
  if (BIAS && g == 0) {
#pragma unroll
    for (int b = 0; b < NT; ++b)
      if (NT * r + b < n_valid) gb[NT * r + b] = accb[b][0] + (accumulate ? oldb[b] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------ wgrad runner
// Walks L-1..0 beside a dgrad runner that shares dY/d0/d1: dW_l, db_l from (h_{l-1} | x, delta_l) into `slab` (flat layout
// of one net).  Executes n_steps barriers.
template <int HT, int SP, bool WIDE = false, bool THIN = false>
__device__ __forceinline__ void chain_wgrad_run(const NetShape sh, const float *x, int ldx, const float *hbuf, const float *dY, int ldy,
                                                const float *d0, const float *d1, float *__restrict__ slab, bool accumulate, int ldh,
                                                int n_steps, int sub, int lane_, unsigned long long *dbg = nullptr) {
  constexpr bool thin = THIN;
  // thin: the output layer's and layer 0's weight gradients are formed by the caller (thin_wgrad_out / thin_wgrad_first);
  // layers L-2 .. 1 run here: L - 2 barriers.
  constexpr int H = 16 * HT, CT = HT / SP;
  const int T = 16 * ldh;
  const int L = sh.L;
#define DBG_STAMP(i)                                                                 \
  if (dbg && lane_ == 0) {                                                           \
    unsigned long long t_;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
    dbg[i] = t_;                                                                     \
  }
  const int l_hi = thin ? L - 2 : L - 1, l_lo = thin ? 1 : 0;
  float *gW = slab + (sh.K_in * H + H) + (l_hi - 1) * (H * H + H);   // layer l_hi
#pragma nounroll
  for (int l = l_hi; l >= l_lo; --l) {
    const int lane = opaque(lane_);
    const bool out_layer = (l == L - 1);
    const int K = (l == 0) ? sh.K_in : H, N = out_layer ? sh.N_out : H;
    const float *delta = out_layer ? dY : (((l + 1) & 1) ? d1 : d0);
    const int ldd = out_layer ? ldy : ldh;
    const float *hp = (l == 0) ? x : hbuf + (l - 1) * T;
    const int ldp = (l == 0) ? ldx : ldh;
    float *gb = gW + K * N;
    if (l == L - 2) { DBG_STAMP(0); }
    if (out_layer) {
      const int k0 = sub * 16 * CT;  // split the H rows of dW over the SP waves; wave 0 also forms db
      if (N <= 16) {
        if (sub == 0) wgrad_tile_fast<CT, 1, true>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, gb, lane, accumulate);
        else wgrad_tile_fast<CT, 1, false>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, gb, lane, accumulate);
      } else if (WIDE && N <= 32) {
        if (sub == 0) wgrad_tile_fast<CT, 2, true>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, gb, lane, accumulate);
        else wgrad_tile_fast<CT, 2, false>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, gb, lane, accumulate);
      } else if (WIDE && N <= 64) {
        if (sub == 0) wgrad_tile_fast<CT, 4, true>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, gb, lane, accumulate);
        else wgrad_tile_fast<CT, 4, false>(hp + k0, ldp, 16 * CT, delta, ldd, N, gW + k0 * N, N, gb, lane, accumulate);
      } else {
        gen_dense_wgrad(hp, ldp, k0, k0 + 16 * CT, delta, ldd, 0, N, gW, N, lane, accumulate);
        if (sub == 0) wave_dense_bgrad(delta, ldd, N, gb, lane, accumulate);
      }
    } else {
      const int c0 = sub * 16 * CT;
      if (l > 0) {
        wgrad_tile_fast<HT, CT, true>(hp, ldp, H, delta + c0, ldd, 16 * CT, gW + c0, N, gb + c0, lane, accumulate);
      } else if (K <= 16) {
        wgrad_tile_fast<1, CT, true>(hp, ldp, K, delta + c0, ldd, 16 * CT, gW + c0, N, gb + c0, lane, accumulate);
      } else if (WIDE && K <= 32) {
        wgrad_tile_fast<2, CT, true>(hp, ldp, K, delta + c0, ldd, 16 * CT, gW + c0, N, gb + c0, lane, accumulate);
      } else {
        gen_dense_wgrad(hp, ldp, 0, K, delta, ldd, c0, c0 + 16 * CT, gW, N, lane, accumulate);
        wave_dense_bgrad(delta + c0, ldd, 16 * CT, gb + c0, lane, accumulate);
      }
      if (l == L - 2) { DBG_STAMP(1); }
    }
    if (l == L - 2) { DBG_STAMP(2); }
    __syncthreads();
    if (l == L - 2) { DBG_STAMP(3); }
    gW -= (l - 1 == 0) ? (sh.K_in * H + H) : (H * H + H);
  }
#pragma nounroll
  for (int l = l_hi - l_lo + 1; l < n_steps; ++l) __syncthreads();
}

#undef DBG_STAMP
__device__ __forceinline__ void chain_idle_run(int n_steps) {
#pragma nounroll
  for (int l = 0; l < n_steps; ++l) __syncthreads();
}

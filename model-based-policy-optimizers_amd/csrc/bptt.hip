// bptt.hip — B1-B5: BPTT actor gradient through the learned model (see include/mbpo_hip.h).
//
// One workgroup (8 waves) owns a tile of 16 trajectories for the whole horizon.
//   FORWARD  t = 0..H-1 : policy (4 waves share the chain) -> sample -> model (ensemble members, 2 waves per chain, 4 chains
//                         per round; or the analytic pendulum) -> reward -> target critics on the next state.  Checkpoints
//                         x_t, a_t, eps_t, r_t, V_t, argmin go to HBM (the same workgroup reads them back: workgroup-scope
//                         visibility after __syncthreads), transitions go to the output.
//   LAMBDA              : R_t = r~_t + g(1-l)V_t + g*l*R_{t+1} per trajectory; dL/dR_t has a closed form (same for every row).
//   BACKWARD t = H-1..0 : recompute policy (z,h kept) and target critics (z kept) in lockstep; critic input-gradient;
//                         per ensemble round: recompute members (z kept) + input-gradient; reward gradient; action /
//                         log-prob terms; policy backward = two delta chains (total, log-prob-only) + one wgrad chain that
//                         ACCUMULATES into this workgroup's slab.  dL/dx_t is carried in LDS.
// Recompute instead of stashing: the stash would be ~180 KB per (tile, step) (E=10), i.e. GBs per train step through HBM;
// recomputation costs one extra forward (4x instead of 3x forward FLOPs) and keeps everything in LDS.
// Algorithmic FLOP per (trajectory, step): 3*(2P) + 2*(2*E*M) + 2*(2*2V)  (SURVEY §8d).
#include "common.hpp"
#include "chain_run.hpp"
#include <stdlib.h>
#include <string.h>

#define LOG_SQRT_2PI_B 0.91893853320467274178f

struct BpttArgs {
  MlpDev pi, cr, dyn;
  int X, U, H;
  long long n;
  int system_kind, predict_delta, reward_kind;
  const float *reward_params, *sys_params, *s_mean, *s_std, *r_ms;
  const float *init_states, *act_noise;
  unsigned long long seed, offset;
  const unsigned long long *rng_dev;
  float c0, discount, lambda_, ent_coef;
  float *transitions, *lambda_values;
  float *w_xs, *w_as, *w_eps, *w_rs, *w_vs, *w_km;
  float *w_z;                    // member pre-activations of the forward sweep, [tile][t][member][hidden layer][16][64], or NULL (recompute)
  float *slabs, *extras;
  int ld_x, ld_xu, ld_h, ld_y, ld_ye, LH, EC;
  NetShape sh_pi, sh_cr, sh_dyn;
  unsigned long long *stamps;   // measurement hook (mbpo_debug_set_bptt_stamps): [16] cycles per op kind + [16] op counts, block 0; or NULL
};

static unsigned long long *g_bptt_stamps = nullptr;
// Measurement hook (not part of include/mbpo_hip.h): device buffer of 32 uint64; block 0 adds the s_memtime cycles of every op
// (run + elementwise section + barrier) to [elem code] and counts it in [16 + elem code].  NULL switches it off.
extern "C" int mbpo_debug_set_bptt_stamps(void *buf) {
  g_bptt_stamps = (unsigned long long *)buf;
  return MBPO_OK;
}

// analytic pendulum step + vector-Jacobian product (dynamics/pendulum_dynamics.py:29-63)
__device__ __forceinline__ void pend_fwd(const float *x, float u, const float *sp, float *xn) {
  const float ms = sp[0], mt = sp[1], dt = sp[2], g = sp[3], mm = sp[4], l = sp[5];
  const float th = atan2f(x[1], x[0]);
  const float uc = fminf(fmaxf(u, -1.f), 1.f) * mt;
  const float thdd = (3.f * g) / (2.f * l) * sinf(th) + 3.f / (mm * (l * l)) * uc;
  float nv = x[2] + thdd * dt;
  nv = fminf(fmaxf(nv, -ms), ms);
  const float nth = th + nv * dt;
  xn[0] = cosf(nth);
  xn[1] = sinf(nth);
  xn[2] = nv;
}
// gx[3] = dL/dx'  ->  dL/dx (3) and dL/du
__device__ __forceinline__ void pend_vjp(const float *x, float u, const float *sp, const float *gx, float *dx, float *du) {
  const float ms = sp[0], mt = sp[1], dt = sp[2], g = sp[3], mm = sp[4], l = sp[5];
  const float r2 = x[0] * x[0] + x[1] * x[1];
  const float th = atan2f(x[1], x[0]);
  const float K = (3.f * g) / (2.f * l), c = 3.f / (mm * (l * l));
  const bool uin = (u > -1.f) && (u < 1.f);
  const float uc = fminf(fmaxf(u, -1.f), 1.f) * mt;
  const float thdd = K * sinf(th) + c * uc;
  const float nv_raw = x[2] + thdd * dt;
  const bool vin = (nv_raw > -ms) && (nv_raw < ms);
  const float nv = fminf(fmaxf(nv_raw, -ms), ms);
  const float nth = th + nv * dt;
  // x' = (cos nth, sin nth, nv)
  const float g_nth = -sinf(nth) * gx[0] + cosf(nth) * gx[1];
  const float g_nv = gx[2] + g_nth * dt;
  const float g_raw = vin ? g_nv : 0.f;
  const float g_th = g_nth + g_raw * dt * K * cosf(th);
  dx[0] = g_th * (-x[1] / r2);
  dx[1] = g_th * (x[0] / r2);
  dx[2] = g_raw;
  *du = uin ? g_raw * dt * c * mt : 0.f;
}

__device__ __forceinline__ float reward_fwd(const BpttArgs &A, const float *xu) {
  const int X = A.X, U = A.U;
  if (A.reward_kind == MBPO_REWARD_PENDULUM) {
    const float *rp = A.reward_params;
    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
    const float theta = atan2f(xu[1], xu[0]);
    float mpy = fmodf(theta - rp[2] + PI_F, TWO_PI_F);
    if (mpy < 0.f) mpy += TWO_PI_F;
    const float d = mpy - PI_F;
    return -(rp[0] * (d * d) + 0.1f * (xu[2] * xu[2])) - rp[1] * (xu[X] * xu[X]);
  }
  const float *tp = A.reward_params, *qp = tp + X, *rp = qp + X;
  float cx = 0.f, cu = 0.f;
  for (int c = 0; c < X; ++c) { float dd = xu[c] - tp[c]; cx += qp[c] * (dd * dd); }
  for (int d = 0; d < U; ++d) cu += rp[d] * (xu[X + d] * xu[X + d]);
  return -cx - cu;
}
// one component (c < X: state, c >= X: action) of g * d reward / d(x,u): lets the (row, column) threads of a section each take one
__device__ __forceinline__ float reward_grad_comp(const BpttArgs &A, const float *xu, float g, int c) {
  const int X = A.X;
  if (A.reward_kind == MBPO_REWARD_PENDULUM) {
    const float *rp = A.reward_params;
    if (c == 2) return g * (-0.2f * xu[2]);
    if (c >= X) return g * (-2.f * rp[1] * xu[X]);
    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
    const float theta = atan2f(xu[1], xu[0]);
    float mpy = fmodf(theta - rp[2] + PI_F, TWO_PI_F);
    if (mpy < 0.f) mpy += TWO_PI_F;
    const float d = mpy - PI_F;
    const float r2 = xu[0] * xu[0] + xu[1] * xu[1];
    const float g_th = g * (-2.f * rp[0] * d);
    return c == 0 ? g_th * (-xu[1] / r2) : g_th * (xu[0] / r2);
  }
  const float *tp = A.reward_params, *qp = tp + X, *rp = qp + X;
  return c < X ? g * (-2.f * qp[c] * (xu[c] - tp[c])) : g * (-2.f * rp[c - X] * xu[c]);
}

// hardware transcendentals for the elementwise sections (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp, as in sac.hip / ppo.hip): at
// u = 6 a section is 96 elements on two of the eight waves — its libm calls (expf, log1pf, logf x3, tanhf: 30-60 instructions
// each) were the section's whole time while the other waves waited at its barrier.
__device__ __forceinline__ float bp_fexp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
__device__ __forceinline__ float bp_flog(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float bp_fsoftplus(float x) { return fmaxf(x, 0.0f) + bp_flog(1.0f + bp_fexp(-fabsf(x))); }
__device__ __forceinline__ float bp_ftanh(float x) {
  const float e = bp_fexp(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
  return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}

template <int H, bool WIDE>
__global__ void __launch_bounds__(512) k_bptt_actor(BpttArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid_ = threadIdx.x, nthreads = blockDim.x;
  const int tid = tid_, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int X = A.X, U = A.U, HZ = A.H;
  const int ld_x = A.ld_x, ld_xu = A.ld_xu, ld_h = A.ld_h, ld_y = A.ld_y, ld_ye = A.ld_ye, LH = A.LH, EC = A.EC;
  const int T = 16 * ld_h;
  const int E = (A.system_kind == MBPO_SYS_ENSEMBLE) ? A.dyn.n_nets : 0;
  const int DT = 2 * X + U + 2;   // transition row (no extras)
  // ---- LDS carve ----
  float *s_x = smem;                          // [16][ld_x]   x_t
  float *s_on = s_x + 16 * ld_x;              // [16][ld_x]   normalised x_t (policy input)
  float *s_nn = s_on + 16 * ld_x;             // [16][ld_x]   normalised x_{t+1} (critic input)
  float *s_xn = s_nn + 16 * ld_x;             // [16][ld_x]   x_{t+1}
  float *s_gx = s_xn + 16 * ld_x;             // [16][ld_x]   dL/dx_{t+1} carry
  float *s_don = s_gx + 16 * ld_x;            // [16][ld_x]   policy input gradient (log-prob chain)
  float *s_dxc = s_don + 16 * ld_x;           // [2][16][ld_x] critic input gradients
  float *s_xu = s_dxc + 2 * 16 * ld_x;        // [16][ld_xu]  [x_t, a_t]
  float *s_dxu = s_xu + 16 * ld_xu;           // [16][ld_xu]  dL/d[x_t, a_t] (model + reward)
  float *s_dxe = s_dxu + 16 * ld_xu;          // [EC][16][ld_xu] per-chain model input gradients
  float *s_y = s_dxe + EC * 16 * ld_xu;       // [16][ld_y]   policy logits
  float *s_dyt = s_y + 16 * ld_y;             // [16][ld_y]   dL/dlogits, total
  float *s_dyl = s_dyt + 16 * ld_y;           // [16][ld_y]   dL/dlogits, log-prob path only
  float *s_yv = s_dyl + 16 * ld_y;            // [2][16][4]   critic outputs
  float *s_dyv = s_yv + 2 * 16 * 4;           // [2][16][4]
  float *s_ye = s_dyv + 2 * 16 * 4;           // [EC][16][ld_ye] ensemble outputs (one round)
  float *s_dye = s_ye + EC * 16 * ld_ye;      // [EC][16][ld_ye]
  const int U4 = (16 * U + 3) & ~3;
  float *s_a = s_dye + EC * 16 * ld_ye;       // [16][U]
  float *s_eps = s_a + U4;
  float *s_scal = s_eps + U4;                 // [8][16] per-row scalars
  float *s_gR = s_scal + 128;                 // [HZ+1]  dL/dR_t (row independent)
  float *s_pi = s_gR + ((HZ + 4) & ~3);       // 2*LH tiles: policy z, h
  float *s_B = s_pi + 2 * LH * T;             // shared region: critics (2*LH z + 4 pp) | ensemble round EC*(LH+2) | policy deltas
  float *zp = s_pi, *hp = s_pi + LH * T;

  const float gam = A.discount, lam = A.lambda_;
  const float r_mean = A.r_ms[0], r_std = A.r_ms[1];
  const float invNH = 1.0f / ((float)A.n * (float)HZ);
  const float w_lp = -A.ent_coef * invNH;      // dL/d log_prob_t   (entropy_loss = -mean_t lp, weight ent_coef)
  const RngKey rk_ = rng_resolve(A.seed, A.offset, A.rng_dev);
  const unsigned long long rng_off = rk_.offset, rng_seed = rk_.seed;
  const int PL = A.pi.n_layers, CL = A.cr.n_layers, DL = A.dyn.n_layers;
  const float *cr1 = A.cr.params, *cr2 = A.cr.params + A.cr.net_stride;
  const int chain2 = wave >> 1, sub2 = wave & 1;   // SP = 2 grouping

  // dL/dR_t: R_t enters the loss with -g^t/(nH) and R_{t-1} with g*l  (closed form, same for every trajectory)
  if (tid == 0) {
    float gr = 0.f, gp = 1.f;
    for (int t = 0; t < HZ; ++t) {
      gr = -gp * invNH + gam * lam * gr;
      s_gR[t] = gr;
      gp *= gam;
    }
    s_gR[HZ] = gam * lam * gr;   // R_H = V_{H-1}
  }

  float *slab = A.slabs + (long long)blockIdx.x * A.pi.n_params;
  float loss_ret = 0.f, loss_lp = 0.f;   // loss_ret: threads 0..15; loss_lp: the (row, action-dim) threads of the logits section
  bool first_tile = true;
  const long long n_tiles = (A.n + 15) >> 4;
  const bool ens = A.system_kind == MBPO_SYS_ENSEMBLE;
  const int NR = ens ? (E + EC - 1) / EC : 0;   // ensemble rounds of EC member chains
  const int Rm = ens ? NR : 1;                  // model ops of a forward step (pendulum: one elementwise op)
  const int nF = Rm + 3, nB = 5 + (ens ? 2 * NR : 1);
  // LDS offsets (floats from smem) of everything a chain can touch: the chain descriptor below is a handful of scalars
  const int o_on = (int)(s_on - smem), o_nn = (int)(s_nn - smem), o_xu = (int)(s_xu - smem), o_y = (int)(s_y - smem);
  const int o_dyt = (int)(s_dyt - smem), o_dyl = (int)(s_dyl - smem), o_yv = (int)(s_yv - smem), o_dyv = (int)(s_dyv - smem);
  const int o_ye = (int)(s_ye - smem), o_dye = (int)(s_dye - smem), o_don = (int)(s_don - smem), o_dxc = (int)(s_dxc - smem);
  const int o_dxe = (int)(s_dxe - smem), o_zp = (int)(zp - smem), o_hp = (int)(hp - smem), o_B = (int)(s_B - smem);
#define P(off) ((off) < 0 ? (float *)nullptr : smem + (off))

  // One step of the horizon = a short PROGRAM of ops; an op = (chain run, elementwise section, barrier).  Forward and backward
  // steps walk the same loop body, so every runner (forward / dgrad / wgrad) is instantiated at exactly ONE call site — the
  // per-site copies of the first version were 270 KB of code for a 64 KB instruction cache and spilled 250 VGPRs.
  // (Requesting the NEXT op's first layer before the current op's elementwise section was tried: the loop-carried chain
  //  descriptor cost 110 more SGPR spills and the kernel got 0-6% slower; the request stays right in front of its run.)
  //   forward  t: [pi fwd | sample a_t] [member round r fwd | mean += ..]* (or [pendulum step]) [- | reward, normalise x'] [V1,V2 fwd | store, advance]
  //   backward t: [- | reload step t] [pi, V1, V2 recompute | dL/dV] [V dgrad | dL/dx' += ..]
  //               ([member round r recompute | dL/dy_e] [member dgrad | dL/d(x,a) += ..])* (or [- | pendulum vjp])
  //               [- | reward vjp, dL/dlogits] [pi dgrad x2, wgrad | dL/dx_t]
  enum { R_NONE, R_PI_FWD, R_ENS_FWD, R_CR_FWD, R_RECOMP, R_CR_DG, R_ENS_REFWD, R_ENS_DG, R_PI_BWD };
  enum { E_SAMPLE, E_ACCUM, E_PEND, E_NN, E_STORE, E_RELOAD, E_DV, E_GXACC, E_DYE, E_DXUACC, E_PENDVJP, E_LOGITS, E_GXFINAL };

  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, first_tile = false) {
    const long long row0 = tile * 16;
    for (int idx = tid; idx < 16 * X; idx += nthreads) {
      const int r = idx / X, c = idx - r * X;
      const long long i = row0 + r;
      const float v = i < A.n ? A.init_states[i * X + c] : 0.f;
      s_x[r * ld_x + c] = v;
      s_on[r * ld_x + c] = (v - A.s_mean[c]) / A.s_std[c];
      if (i < A.n) A.w_xs[(i * (HZ + 1)) * X + c] = v;
    }
    __syncthreads();
#pragma nounroll
    for (int step = 0; step < 2 * HZ; ++step) {
      const bool bwd = step >= HZ;
      const int t = bwd ? 2 * HZ - 1 - step : step;
      const int nops = bwd ? nB : nF;
#pragma nounroll
      for (int op = 0; op < nops; ++op) {
        // per-lane values are re-derived every op: hoisted out of the loops they fill the register file and spill
        const int tid = opaque(tid_), lane = tid & 63;
        unsigned long long t_op0 = 0;
        if (A.stamps && blockIdx.x == 0 && tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_op0)::"memory");
        // ---- decode the op (wave-uniform scalars) ----
        int run = R_NONE, elem, rr = 0;
        if (!bwd) {
          if (op == 0) { run = R_PI_FWD; elem = E_SAMPLE; }
          else if (op <= Rm) { run = ens ? R_ENS_FWD : R_NONE; elem = ens ? E_ACCUM : E_PEND; rr = op - 1; }
          else if (op == Rm + 1) { elem = E_NN; }
          else { run = R_CR_FWD; elem = E_STORE; }
        } else {
          const int om = op - 3;   // position inside the model block
          if (op == 0) { elem = E_RELOAD; }
          else if (op == 1) { run = R_RECOMP; elem = E_DV; }
          else if (op == 2) { run = R_CR_DG; elem = E_GXACC; }
          else if (op < nB - 2) {
            if (ens) { rr = om >> 1; run = (om & 1) ? R_ENS_DG : (A.w_z ? R_NONE : R_ENS_REFWD); elem = (om & 1) ? E_DXUACC : E_DYE; }
            else { elem = E_PENDVJP; }
          } else if (op == nB - 2) { elem = E_LOGITS; }
          else { run = R_PI_BWD; elem = E_GXFINAL; }
        }
        // ---- this wave's chain in the op ----
        int mode = CH_IDLE, len = 0, cldx = ld_x, cldy = ld_y, cld_dx = ld_x;
        int cx = -1, cpp0 = -1, cpp1 = -1, czb = -1, chb = -1, cy = -1, cdx = -1;
        NetShape sh = A.sh_pi;
        const float *cparams = A.pi.params;
        if (run == R_PI_FWD) {
          len = PL;
          if (chain2 == 0) { mode = CH_FWD; cx = o_on; cpp0 = o_B; cpp1 = o_B + T; cy = o_y; }
        } else if (run == R_ENS_FWD || run == R_ENS_REFWD || run == R_ENS_DG) {
          len = DL;
          const int e = rr * EC + chain2;
          if (chain2 < EC && e < E) {
            sh = A.sh_dyn;
            cparams = A.dyn.params + (long long)e * A.dyn.net_stride;
            cldy = ld_ye;
            if (run == R_ENS_FWD && !A.w_z) {
              mode = CH_FWD; cx = o_xu; cldx = ld_xu; cpp0 = o_B + chain2 * 2 * T; cpp1 = cpp0 + T; cy = o_ye + chain2 * 16 * ld_ye;
            } else {
              const int ze = o_B + chain2 * (LH + 2) * T;
              czb = ze; cpp0 = ze + LH * T; cpp1 = cpp0 + T;
              if (run != R_ENS_DG) { mode = CH_FWD; cx = o_xu; cldx = ld_xu; cy = o_ye + chain2 * 16 * ld_ye; }
              else { mode = CH_DGRAD; cy = o_dye + chain2 * 16 * ld_ye; cdx = o_dxe + chain2 * 16 * ld_xu; cld_dx = ld_xu; }
            }
          }
        } else if (run == R_CR_FWD) {
          len = CL;
          if (chain2 < 2) {
            mode = CH_FWD; sh = A.sh_cr; cparams = chain2 ? cr2 : cr1;
            cx = o_nn; cpp0 = o_B + 2 * chain2 * T; cpp1 = cpp0 + T; cy = o_yv + chain2 * 16 * 4; cldy = 4;
          }
        } else if (run == R_RECOMP) {          // critics region: z | z | 4 ping-pong tiles
          len = PL > CL ? PL : CL;
          if (chain2 == 0) { mode = CH_FWD; cx = o_on; czb = o_zp; chb = o_hp; cy = o_y; }
          else if (chain2 < 3) {
            const int net = chain2 - 1;
            mode = CH_FWD; sh = A.sh_cr; cparams = net ? cr2 : cr1;
            cx = o_nn; cpp0 = o_B + 2 * LH * T + 2 * net * T; cpp1 = cpp0 + T; czb = o_B + net * LH * T; cy = o_yv + net * 16 * 4; cldy = 4;
          }
        } else if (run == R_CR_DG) {
          len = CL;
          if (chain2 == 1 || chain2 == 2) {
            const int net = chain2 - 1;
            mode = CH_DGRAD; sh = A.sh_cr; cparams = net ? cr2 : cr1;
            cy = o_dyv + net * 16 * 4; cldy = 4; czb = o_B + net * LH * T; cpp0 = o_B + 2 * LH * T + 2 * net * T; cpp1 = cpp0 + T;
            cdx = o_dxc + net * 16 * ld_x;
          }
        } else if (run == R_PI_BWD) {          // chain 0: delta(total), chain 1: delta(log-prob path) + input gradient, chain 2: wgrad
          len = PL;
          if (chain2 < 2) {
            mode = CH_DGRAD; cy = chain2 ? o_dyl : o_dyt; czb = o_zp; cpp0 = o_B + 2 * chain2 * T; cpp1 = cpp0 + T;
            cdx = chain2 ? o_don : -1;
          } else if (chain2 == 2) {
            mode = CH_WGRAD; cx = o_on; chb = o_hp; cy = o_dyt; cpp0 = o_B; cpp1 = o_B + T;
          }
        }
        // ---- the run: one call site per runner ----
        if (run != R_NONE) {
          if (mode == CH_FWD) {
            WSet<HT, 2> R2;
            chain_fwd_prefetch<HT, 2, WIDE>(R2, sh, cparams, sub2, lane);
            chain_fwd_run<HT, 2, WIDE>(sh, cparams, P(cx), cldx, P(cpp0), P(cpp1), P(czb), P(chb), P(cy), cldy, ld_h, len, sub2, lane, R2);
          } else if (mode == CH_DGRAD) {
            WSet<HT, 2> R2;
            chain_dgrad_prefetch<HT, 2, WIDE>(R2, sh, cparams, sub2, lane);
            chain_dgrad_run<HT, 2, WIDE>(sh, cparams, P(cy), cldy, P(czb), P(cpp0), P(cpp1), P(cdx), cld_dx, ld_h, len, sub2, lane, R2);
          } else if (mode == CH_WGRAD) {
            chain_wgrad_run<HT, 2, WIDE>(sh, P(cx), cldx, P(chb), P(cy), cldy, P(cpp0), P(cpp1), slab, !(first_tile && t == HZ - 1), ld_h, len,
                                         sub2, lane);
          } else {
            chain_idle_run(len);
          }
        }
        // ---- the elementwise section ----
        if (elem == E_SAMPLE) {
          for (int idx = tid; idx < 16 * U; idx += nthreads) {
            const int r = idx / U, d = idx - r * U;
            const long long i = row0 + r;
            const float mu = s_y[r * ld_y + d];
            const float sg = fminf(fmaxf(bp_fsoftplus(s_y[r * ld_y + U + d] + A.c0), 1e-6f), 1e2f);
            float eps = 0.f;
            if (i < A.n) {
              const long long nidx = (i * HZ + t) * U + d;
              eps = A.act_noise ? A.act_noise[nidx] : philox_normal(rng_seed, rng_off, MBPO_STREAM_POLICY_NOISE, (unsigned long long)nidx);
              A.w_eps[nidx] = eps;
            }
            const float a = fminf(fmaxf(bp_ftanh(mu + eps * sg), -0.999f), 0.999f);   // squash_action (:313-317)
            s_xu[r * ld_xu + X + d] = a;
            if (i < A.n) A.w_as[(i * HZ + t) * U + d] = a;
          }
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx / X, c = idx - r * X;
            const float xv = s_x[r * ld_x + c];
            s_xu[r * ld_xu + c] = xv;
            if (ens) s_xn[r * ld_x + c] = A.predict_delta ? xv : 0.f;   // the member means are added round by round
          }
        } else if (elem == E_ACCUM) {
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx / X, c = idx - r * X;
            float acc = 0.f;
            for (int cc = 0; cc < EC && rr * EC + cc < E; ++cc) acc += s_ye[(cc * 16 + r) * ld_ye + c];
            s_xn[r * ld_x + c] += acc / (float)E;
          }
          if (A.w_z) {
            // the round's pre-activation tiles -> global memory: the backward sweep reloads them instead of running the members'
            // forward pass a second time (round 4: that recompute was 22 % of a horizon step at E = 10)
            // (tile m = cc * nz + l of the round is 4 KB at round base + 4 KB * m: members are consecutive in the store)
            const int nz = DL - 1, cnt = (E - rr * EC) < EC ? (E - rr * EC) : EC, total = cnt * nz * 256;
            float *const zg = A.w_z + ((((tile * HZ + t) * E + rr * EC) * nz) << 10);
            for (int idx = tid; idx < total; idx += nthreads) {
              const int m = idx >> 8, q = idx & 255;
              const int cc = (m >= nz) + (m >= 2 * nz) + (m >= 3 * nz), l = m - cc * nz;
              *reinterpret_cast<f32x4 *>(zg + 4 * idx) =
                  *reinterpret_cast<const f32x4 *>(smem + o_B + cc * (LH + 2) * T + l * T + (q >> 4) * ld_h + 4 * (q & 15));
            }
          }
        } else if (elem == E_PEND) {
          if (tid < 16) {
            float xn[3];
            pend_fwd(s_xu + tid * ld_xu, s_xu[tid * ld_xu + X], A.sys_params, xn);
            s_xn[tid * ld_x + 0] = xn[0]; s_xn[tid * ld_x + 1] = xn[1]; s_xn[tid * ld_x + 2] = xn[2];
          }
        } else if (elem == E_NN) {
          if (tid < 16) {
            const long long i = row0 + tid;
            const float rew = reward_fwd(A, s_xu + tid * ld_xu);
            if (i < A.n) A.w_rs[i * HZ + t] = rew;
            s_scal[tid] = rew;
          }
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx / X, c = idx - r * X;
            s_nn[r * ld_x + c] = (s_xn[r * ld_x + c] - A.s_mean[c]) / A.s_std[c];
          }
        } else if (elem == E_STORE) {
          if (tid < 16) {
            const long long i = row0 + tid;
            const float v1 = s_yv[tid * 4], v2 = s_yv[(16 + tid) * 4];
            if (i < A.n) {
              A.w_vs[i * HZ + t] = fminf(v1, v2);
              A.w_km[i * HZ + t] = v1 < v2 ? 0.f : (v2 < v1 ? 1.f : 2.f);
            }
          }
          for (int idx = tid; idx < 16 * DT; idx += nthreads) {   // transition row
            const int r = idx / DT, c = idx - r * DT;
            const long long i = row0 + r;
            if (i < A.n) {
              float v;
              if (c < X + U) v = s_xu[r * ld_xu + c];
              else if (c == X + U) v = s_scal[r];
              else if (c == X + U + 1) v = 1.0f;                     // discount = ones (optimizer_utils.py:114)
              else v = s_xn[r * ld_x + (c - X - U - 2)];
              A.transitions[(i * HZ + t) * DT + c] = v;
            }
          }
          for (int idx = tid; idx < 16 * X; idx += nthreads) {    // advance: x_{t+1} becomes x_t (+ its normalised copy)
            const int r = idx / X, c = idx - r * X;
            const long long i = row0 + r;
            const float v = s_xn[r * ld_x + c];
            s_x[r * ld_x + c] = v;
            s_on[r * ld_x + c] = (v - A.s_mean[c]) / A.s_std[c];
            if (i < A.n) A.w_xs[(i * (HZ + 1) + t + 1) * X + c] = v;
          }
        } else if (elem == E_RELOAD) {
          if (t == HZ - 1) {
            // (the forward sweep's stores of this wave — checkpoints, member pre-activations — have left it before anything of the
            //  backward sweep is requested; the op's closing barrier then orders them against every wave's reloads)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ===== lambda returns (once per tile, between the sweeps): R_t = r~_t + g(1-l)V_t + g*l*R_{t+1} =====
            if (tid < 16) {
              const long long i = row0 + tid;
              if (i < A.n) {
                float agg = A.w_vs[i * HZ + HZ - 1];
                for (int tt = HZ - 1; tt >= 0; --tt) {
                  const float rn = (A.w_rs[i * HZ + tt] - r_mean) / r_std;
                  agg = rn + gam * A.w_vs[i * HZ + tt] * (1.f - lam) + gam * lam * agg;
                  A.lambda_values[i * HZ + tt] = agg;
                }
                float gp = 1.f, acc = 0.f;
                for (int tt = 0; tt < HZ; ++tt) {
                  acc += A.lambda_values[i * HZ + tt] * gp;
                  gp *= gam;
                }
                loss_ret += acc;
              }
            }
            for (int idx = tid; idx < 16 * ld_x; idx += nthreads) s_gx[idx] = 0.f;
          }
          // reload the step: x_t, x_{t+1}, a_t, eps_t
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx / X, c = idx - r * X;
            const long long i = row0 + r;
            const float xt = i < A.n ? A.w_xs[(i * (HZ + 1) + t) * X + c] : 0.f;
            const float xn = i < A.n ? A.w_xs[(i * (HZ + 1) + t + 1) * X + c] : 0.f;
            s_x[r * ld_x + c] = xt;
            s_xu[r * ld_xu + c] = xt;
            s_on[r * ld_x + c] = (xt - A.s_mean[c]) / A.s_std[c];
            s_nn[r * ld_x + c] = (xn - A.s_mean[c]) / A.s_std[c];
          }
          for (int idx = tid; idx < 16 * U; idx += nthreads) {
            const int r = idx / U, d = idx - r * U;
            const long long i = row0 + r;
            const float a = i < A.n ? A.w_as[(i * HZ + t) * U + d] : 0.f;
            s_a[idx] = a;
            s_eps[idx] = i < A.n ? A.w_eps[(i * HZ + t) * U + d] : 0.f;
            s_xu[r * ld_xu + X + d] = a;
          }
        } else if (elem == E_DV) {
          if (tid < 16) {   // dL/dV_t on the arg-min target critic
            const long long i = row0 + tid;
            float dV = s_gR[t] * gam * (1.f - lam);
            if (t == HZ - 1) dV += s_gR[HZ];
            const float km = i < A.n ? A.w_km[i * HZ + t] : 0.f;
            if (i >= A.n) dV = 0.f;
            s_dyv[tid * 4] = km == 0.f ? dV : (km == 2.f ? 0.5f * dV : 0.f);
            s_dyv[(16 + tid) * 4] = km == 1.f ? dV : (km == 2.f ? 0.5f * dV : 0.f);
          }
        } else if (elem == E_GXACC) {
          // dL/dx_{t+1} += critic path (d nn / d x' = 1/std); dL/d[x_t, a_t] starts from it (delta model) or from zero
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx / X, c = idx - r * X;
            const float gxn = s_gx[r * ld_x + c] + (s_dxc[r * ld_x + c] + s_dxc[(16 + r) * ld_x + c]) / A.s_std[c];
            s_gx[r * ld_x + c] = gxn;
            s_dxu[r * ld_xu + c] = (ens && A.predict_delta) ? gxn : 0.f;
          }
          for (int idx = tid; idx < 16 * U; idx += nthreads) {
            const int r = idx / U, d = idx - r * U;
            s_dxu[r * ld_xu + X + d] = 0.f;
          }
        } else if (elem == E_DYE) {
          if (A.w_z) {
            // straight into LDS, one 256-byte tile row per wave instruction (global_load_lds_dword: wave-uniform LDS base + 4 * lane,
            // so a padded row is exactly one instruction); no data registers, every row of the round in flight at once; the op's
            // closing __syncthreads() waits for them (vmcnt(0))
            const int nz = DL - 1, cnt = (E - rr * EC) < EC ? (E - rr * EC) : EC, rows = cnt * nz * 16;
            const float *const zg = A.w_z + ((((tile * HZ + t) * E + rr * EC) * nz) << 10);
            for (int rix = wave; rix < rows; rix += (nthreads >> 6)) {
              const int m = rix >> 4, r = rix & 15;
              const int cc = (m >= nz) + (m >= 2 * nz) + (m >= 3 * nz), l = m - cc * nz;
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) float *)(zg + rix * 64 + lane),
                                               (__attribute__((address_space(3))) float *)(smem + o_B + cc * (LH + 2) * T + l * T + r * ld_h), 4, 0, 0);
            }
          }
          const int dout = A.dyn.dims[DL];
          for (int idx = tid; idx < 16 * dout; idx += nthreads) {
            const int r = idx / dout, c = idx - r * dout;
            const float v = c < X ? s_gx[r * ld_x + c] / (float)E : 0.f;   // x' = base + mean_e mu_e: the same for every member
            for (int cc = 0; cc < EC; ++cc) s_dye[(cc * 16 + r) * ld_ye + c] = v;
          }
        } else if (elem == E_DXUACC) {
          for (int idx = tid; idx < 16 * (X + U); idx += nthreads) {
            const int r = idx / (X + U), c = idx - r * (X + U);
            float acc = 0.f;
            for (int cc = 0; cc < EC && rr * EC + cc < E; ++cc) acc += s_dxe[(cc * 16 + r) * ld_xu + c];
            s_dxu[r * ld_xu + c] += acc;
          }
        } else if (elem == E_PENDVJP) {
          if (tid < 16) {
            float dx[3], du;
            pend_vjp(s_xu + tid * ld_xu, s_xu[tid * ld_xu + X], A.sys_params, s_gx + tid * ld_x, dx, &du);
            s_dxu[tid * ld_xu + 0] = dx[0]; s_dxu[tid * ld_xu + 1] = dx[1]; s_dxu[tid * ld_xu + 2] = dx[2];
            s_dxu[tid * ld_xu + X] = du;
          }
        } else if (elem == E_LOGITS) {
          // reward gradient, action / log-prob terms -> dL/dlogits (total, log-prob path); one (row, column) per thread
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx / X, c = idx - r * X;
            const bool ok = row0 + r < A.n;
            s_dxu[r * ld_xu + c] += reward_grad_comp(A, s_xu + r * ld_xu, ok ? s_gR[t] / r_std : 0.f, c);   // dL/dr_t = dL/dR_t / r_std
          }
          for (int idx = tid; idx < 16 * U; idx += nthreads) {
            const int r = idx / U, d = idx - r * U;
            const bool ok = row0 + r < A.n;
            const float mu = s_y[r * ld_y + d], sraw = s_y[r * ld_y + U + d] + A.c0;
            const float sp = bp_fsoftplus(sraw);
            const bool sin_ = (sp > 1e-6f) && (sp < 1e2f);
            const float sg = fminf(fmaxf(sp, 1e-6f), 1e2f);
            const float dsig = sin_ ? fast_sigmoid(sraw) : 0.f;
            const float a = s_a[r * U + d], eps = s_eps[r * U + d];
            const float th = bp_ftanh(mu + eps * sg);
            const float dadw = (th > -0.999f && th < 0.999f) ? (1.f - th * th) : 0.f;
            const float om = 1.f - a * a;
            const float u = 0.5f * bp_flog((1.f + a) * __builtin_amdgcn_rcpf(1.f - a));            // atanh(a)   (:111-120)
            const float q = (u - mu) / sg;
            const float lp = -0.5f * q * q - bp_flog(sg) - LOG_SQRT_2PI_B - bp_flog(om);
            const float dlp_da = (-q / sg) / om + 2.f * a / om;
            const float Ga = s_dxu[r * ld_xu + X + d] + reward_grad_comp(A, s_xu + r * ld_xu, ok ? s_gR[t] / r_std : 0.f, X + d) + w_lp * dlp_da;
            const float l_mu = w_lp * (q / sg), l_sr = w_lp * ((q * q - 1.f) / sg) * dsig;
            const float a_mu = Ga * dadw, a_sr = Ga * dadw * eps * dsig;
            s_dyl[r * ld_y + d] = ok ? l_mu : 0.f;
            s_dyl[r * ld_y + U + d] = ok ? l_sr : 0.f;
            s_dyt[r * ld_y + d] = ok ? l_mu + a_mu : 0.f;
            s_dyt[r * ld_y + U + d] = ok ? l_sr + a_sr : 0.f;
            if (ok) loss_lp += lp;
          }
        } else {   // E_GXFINAL: dL/dx_t = model/reward x-part + policy-input path (through the state normaliser)
          for (int idx = tid; idx < 16 * X; idx += nthreads) {
            const int r = idx / X, c = idx - r * X;
            s_gx[r * ld_x + c] = s_dxu[r * ld_xu + c] + s_don[r * ld_x + c] / A.s_std[c];
          }
        }
        __syncthreads();
        if (A.stamps && blockIdx.x == 0 && tid == 0) {
          unsigned long long t_op1;
          asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_op1)::"memory");
          A.stamps[elem] += t_op1 - t_op0;
          A.stamps[16 + elem] += 1ull;
        }
      }
    }
  }
#undef P
  {
    float lpw = loss_lp;   // fixed shuffle tree per wave, waves in order: deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lpw += __shfl_down(lpw, o, 64);
    if (tid < 16) s_scal[tid] = loss_ret;
    if (lane == 0) s_scal[16 + wave] = lpw;
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < 16; ++i) a += s_scal[i];
    for (int w = 0; w < (int)(nthreads >> 6); ++w) b += s_scal[16 + w];
    A.extras[blockIdx.x * 2 + 0] = a;
    A.extras[blockIdx.x * 2 + 1] = b;
  }
}

struct BpttReduceArgs {
  const float *slabs, *extras;
  int n_slabs, P, H;
  long long n;
  float ent_coef;
  float *grads, *metrics;
};

__global__ void __launch_bounds__(256) k_bptt_reduce(BpttReduceArgs A) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const float gsum = slab_sum_wg64(A.slabs, A.P, A.n_slabs, i, i < A.P);
  if (threadIdx.x < 64 && i < A.P) A.grads[i] = gsum;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float a = slab_sum<16>(A.extras, 2, A.n_slabs, 0), b = slab_sum<16>(A.extras, 2, A.n_slabs, 1);
    const float invNH = 1.0f / ((float)A.n * (float)A.H);
    const float ent = -b * invNH;                 // entropy_loss = -mean log_prob
    A.metrics[0] = -a * invNH + A.ent_coef * ent; // actor_loss (:352)
    A.metrics[1] = ent;
  }
}

// ------------------------------------------------------------------------------------------------ host
struct BpttPlan {
  MlpDev pi, cr, dyn;
  int P, C, H, LH, EC, n_slabs;
  int ld_x, ld_xu, ld_h, ld_y, ld_ye;
  size_t lds;
  long long o_xs, o_as, o_eps, o_rs, o_vs, o_km, o_z, o_slabs, o_extras, total;
};

static int bptt_hidden(const int *dims, int n_layers) {
  if (n_layers < 2) return -1;
  for (int l = 2; l < n_layers; ++l)
    if (dims[l] != dims[1]) return -1;
  return dims[1];
}

static int bptt_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

static int bptt_plan(const mbpo_bptt_desc *d, BpttPlan *pl, bool need_ptrs) {
  MBPO_REQUIRE(d, MBPO_ERR_ARG, "bptt: null descriptor");
  MBPO_REQUIRE(d->x_dim > 0 && d->u_dim > 0 && d->horizon > 0 && d->horizon <= 1024 && d->n > 0, MBPO_ERR_ARG, "bptt: bad sizes");
  MBPO_REQUIRE(d->actor_layers >= 2 && d->actor_layers <= MBPO_MAX_LAYERS && d->critic_layers >= 2 && d->critic_layers <= MBPO_MAX_LAYERS,
               MBPO_ERR_ARG, "bptt: networks need at least one hidden layer");
  MBPO_REQUIRE(d->actor_dims[0] == d->x_dim && d->actor_dims[d->actor_layers] == 2 * d->u_dim, MBPO_ERR_ARG, "bptt: actor must map [x] -> [2u]");
  MBPO_REQUIRE(d->critic_dims[0] == d->x_dim && d->critic_dims[d->critic_layers] == 1, MBPO_ERR_ARG, "bptt: critic must map [x] -> [1]");
  const int Ha = bptt_hidden(d->actor_dims, d->actor_layers), Hc = bptt_hidden(d->critic_dims, d->critic_layers);
  MBPO_REQUIRE(Ha == Hc && Ha == 64, MBPO_ERR_UNSUPPORTED, "bptt: actor/critic hidden layers must all be 64 wide (got %d, %d)", Ha, Hc);
  mbpo_mlp_desc md;
  md.net_stride = 0; md.n_nets = 1;
  md.params = d->actor_params ? d->actor_params : (const float *)16;
  md.n_layers = d->actor_layers;
  for (int l = 0; l <= d->actor_layers; ++l) md.dims[l] = d->actor_dims[l];
  md.activation = d->actor_activation;
  int rc = mbpo_make_mlp_dev(&md, &pl->pi, "bptt.actor");
  if (rc != MBPO_OK) return rc;
  pl->P = pl->pi.n_params;
  md.params = d->target_critic_params ? d->target_critic_params : (const float *)16;
  md.n_layers = d->critic_layers;
  for (int l = 0; l <= d->critic_layers; ++l) md.dims[l] = d->critic_dims[l];
  md.activation = d->critic_activation;
  rc = mbpo_make_mlp_dev(&md, &pl->cr, "bptt.critic");
  if (rc != MBPO_OK) return rc;
  pl->C = pl->cr.n_params;
  pl->cr.n_nets = 2;
  pl->cr.net_stride = pl->C;
  int lh = d->actor_layers - 1;
  if (d->critic_layers - 1 > lh) lh = d->critic_layers - 1;
  int dyn_out = 0;
  int E = 0;
  if (d->system_kind == MBPO_SYS_ENSEMBLE) {
    rc = mbpo_make_mlp_dev(&d->dynamics, &pl->dyn, "bptt.dynamics");
    if (rc != MBPO_OK) return rc;
    E = pl->dyn.n_nets;
    dyn_out = pl->dyn.dims[pl->dyn.n_layers];
    MBPO_REQUIRE(pl->dyn.dims[0] == d->x_dim + d->u_dim && dyn_out >= d->x_dim, MBPO_ERR_ARG, "bptt: dynamics must map [x+u] -> [>= x]");
    MBPO_REQUIRE(pl->dyn.n_layers >= 2 && bptt_hidden(pl->dyn.dims, pl->dyn.n_layers) == 64, MBPO_ERR_UNSUPPORTED,
                 "bptt: dynamics hidden layers must all be 64 wide");
    if (pl->dyn.n_layers - 1 > lh) lh = pl->dyn.n_layers - 1;
  } else if (d->system_kind == MBPO_SYS_PENDULUM) {
    MBPO_REQUIRE(d->x_dim == 3 && d->u_dim == 1 && d->sys_params, MBPO_ERR_ARG, "bptt: pendulum system needs x=3,u=1,sys_params");
    memset(&pl->dyn, 0, sizeof(pl->dyn));
    pl->dyn.n_layers = 1;
  } else {
    MBPO_REQUIRE(false, MBPO_ERR_ARG, "bptt: unknown system_kind");
  }
  MBPO_REQUIRE(d->reward_kind == MBPO_REWARD_QUADRATIC || (d->reward_kind == MBPO_REWARD_PENDULUM && d->x_dim == 3 && d->u_dim == 1),
               MBPO_ERR_ARG, "bptt: bad reward_kind");
  pl->H = 64;
  pl->LH = lh;
  auto up4 = [](int v) { return (v + 3) & ~3; };
  pl->ld_x = up4(d->x_dim) + 4;
  pl->ld_xu = up4(d->x_dim + d->u_dim) + 4;
  pl->ld_h = 64 + 4;
  pl->ld_y = up4(2 * d->u_dim) + 4;
  pl->ld_ye = up4(dyn_out > 0 ? dyn_out : 1) + 4;
  const int T = 16 * pl->ld_h;
  // ensemble chains per round: as many as fit next to everything else in 160 KiB of LDS (at most 4 = 8 waves / 2)
  pl->EC = 0;
  for (int ec = E > 0 ? (E < 4 ? E : 4) : 1; ec >= 1; --ec) {
    int regionB = 2 * pl->LH + 4;                                     // critics: 2 z stacks + 4 ping-pong tiles
    if (ec * (pl->LH + 2) > regionB) regionB = ec * (pl->LH + 2);     // one ensemble round: per chain z stack + 2 tiles
    size_t f = 8ull * 16 * pl->ld_x + 2ull * 16 * pl->ld_xu + (size_t)ec * 16 * pl->ld_xu + 3ull * 16 * pl->ld_y + 4ull * 16 * 4 +
               2ull * ec * 16 * pl->ld_ye + 2ull * up4(16 * d->u_dim) + 128 + up4(d->horizon + 4) + (size_t)(2 * pl->LH + regionB) * T;
    if (f * sizeof(float) <= 160 * 1024) {
      pl->EC = ec;
      pl->lds = f * sizeof(float);
      break;
    }
  }
  MBPO_REQUIRE(pl->EC >= 1, MBPO_ERR_UNSUPPORTED, "bptt: shapes do not fit 160 KiB of LDS");
  long long tiles = (d->n + 15) / 16;
  long long cap = bptt_num_cus();
  pl->n_slabs = (int)(tiles < cap ? tiles : cap);
  long long o = 0;
  auto take = [&](long long n) { long long at = o; o += (n + 3) & ~3LL; return at; };
  pl->o_xs = take(d->n * (d->horizon + 1) * d->x_dim);
  pl->o_as = take(d->n * d->horizon * d->u_dim);
  pl->o_eps = take(d->n * d->horizon * d->u_dim);
  pl->o_rs = take(d->n * d->horizon);
  pl->o_vs = take(d->n * d->horizon);
  pl->o_km = take(d->n * d->horizon);
  // member pre-activations kept from the forward sweep (the backward sweep then skips the members' recompute): 4 KB per (trajectory
  // tile, step, member, hidden layer) — 1 GB at BASELINE config 5 (n = 4096, H = 32, E = 10); beyond MBPO_BPTT_ZSTORE_MAX_MB (default
  // 16384) the kernel recomputes instead
  pl->o_z = -1;
  if (E > 0) {
    static const long long max_mb = getenv("MBPO_BPTT_ZSTORE_MAX_MB") ? atoll(getenv("MBPO_BPTT_ZSTORE_MAX_MB")) : 16384;
    const long long zf = tiles * d->horizon * E * (pl->dyn.n_layers - 1) * 1024;
    if (zf * 4 <= max_mb * (1LL << 20)) pl->o_z = take(zf);
  }
  pl->o_slabs = take((long long)pl->n_slabs * pl->P);
  pl->o_extras = take((long long)pl->n_slabs * 2);
  pl->total = o;
  if (need_ptrs)
    MBPO_REQUIRE(d->actor_params && d->target_critic_params && d->reward_params && d->state_mean && d->state_std && d->reward_mean_std &&
                     d->init_states && d->transitions && d->lambda_values && d->grads && d->metrics && d->workspace,
                 MBPO_ERR_ARG, "bptt: null pointer");
  return MBPO_OK;
}

extern "C" int64_t mbpo_bptt_workspace_floats(const mbpo_bptt_desc *d) {
  BpttPlan pl;
  int rc = bptt_plan(d, &pl, false);
  if (rc != MBPO_OK) return rc;
  return pl.total;
}

extern "C" int mbpo_bptt_actor_grads(const mbpo_bptt_desc *d, void *stream) {
  BpttPlan pl;
  int rc = bptt_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  BpttArgs A;
  A.pi = pl.pi; A.cr = pl.cr; A.dyn = pl.dyn;
  A.X = d->x_dim; A.U = d->u_dim; A.H = d->horizon; A.n = d->n;
  A.system_kind = d->system_kind; A.predict_delta = d->ens_predict_delta; A.reward_kind = d->reward_kind;
  A.reward_params = d->reward_params; A.sys_params = d->sys_params; A.s_mean = d->state_mean; A.s_std = d->state_std;
  A.r_ms = d->reward_mean_std; A.init_states = d->init_states; A.act_noise = d->act_noise;
  A.seed = d->seed; A.offset = d->offset; A.rng_dev = (const unsigned long long *)d->rng_dev;
  const double s0 = (double)d->init_stddev;
  A.c0 = (float)(s0 < 20.0 ? log(exp(s0) - 1.0) : s0);            // inv_softplus (:107-108)
  A.discount = d->discount; A.lambda_ = d->lambda_; A.ent_coef = d->ent_coef;
  A.transitions = d->transitions; A.lambda_values = d->lambda_values;
  float *ws = d->workspace;
  A.w_xs = ws + pl.o_xs; A.w_as = ws + pl.o_as; A.w_eps = ws + pl.o_eps; A.w_rs = ws + pl.o_rs; A.w_vs = ws + pl.o_vs;
  A.w_km = ws + pl.o_km; A.w_z = pl.o_z >= 0 ? ws + pl.o_z : nullptr; A.slabs = ws + pl.o_slabs; A.extras = ws + pl.o_extras;
  A.ld_x = pl.ld_x; A.ld_xu = pl.ld_xu; A.ld_h = pl.ld_h; A.ld_y = pl.ld_y; A.ld_ye = pl.ld_ye; A.LH = pl.LH; A.EC = pl.EC;
  A.stamps = g_bptt_stamps;
  A.sh_pi = NetShape{A.pi.dims[0], A.pi.n_layers, A.pi.dims[A.pi.n_layers], A.pi.act};
  A.sh_cr = NetShape{A.cr.dims[0], A.cr.n_layers, A.cr.dims[A.cr.n_layers], A.cr.act};
  if (A.system_kind == MBPO_SYS_ENSEMBLE) A.sh_dyn = NetShape{A.dyn.dims[0], A.dyn.n_layers, A.dyn.dims[A.dyn.n_layers], A.dyn.act};
  else A.sh_dyn = NetShape{A.X + A.U, 0, A.X, 0};
  const bool wide = net_is_wide(A.sh_pi) || net_is_wide(A.sh_cr) || net_is_wide(A.sh_dyn);
  rc = wide ? mbpo_ensure_lds<k_bptt_actor<64, true>>(pl.lds, "bptt_actor_grads") : mbpo_ensure_lds<k_bptt_actor<64, false>>(pl.lds, "bptt_actor_grads");
  if (rc != MBPO_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (wide) hipLaunchKernelGGL((k_bptt_actor<64, true>), dim3(pl.n_slabs), dim3(512), pl.lds, st, A);
  else hipLaunchKernelGGL((k_bptt_actor<64, false>), dim3(pl.n_slabs), dim3(512), pl.lds, st, A);
  BpttReduceArgs R;
  R.slabs = A.slabs; R.extras = A.extras; R.n_slabs = pl.n_slabs; R.P = pl.P; R.H = d->horizon; R.n = d->n; R.ent_coef = d->ent_coef;
  R.grads = d->grads; R.metrics = d->metrics;
  hipLaunchKernelGGL(k_bptt_reduce, dim3((pl.P + 63) / 64), dim3(256), 0, st, R);
  MBPO_CHECK_LAUNCH("bptt_actor_grads");
  return MBPO_OK;
}

// ------------------------------------------------------------------------------------------------
// B4: twin-V critic regression on gathered transitions (bptt_optimizer.py:385-419)
// ------------------------------------------------------------------------------------------------
struct CriticArgs {
  MlpDev cr;
  NetShape sh;
  int X, D;
  const float *transitions, *lambda_values, *s_mean, *s_std;
  const int *idx;
  long long batch;
  float *slabs, *extras;
  int ld_x, ld_h, LH;
};

// 4 chains x SP waves on phase runners (chain_run.hpp): forward = critic_1 | critic_2 (z and h kept); backward = a dgrad and a
// wgrad chain per net side by side; a workgroup walks tiles and accumulates into its slab.
template <int H, int SP, bool WIDE>
__global__ void __launch_bounds__(256 * SP) k_critic_fwd_bwd(CriticArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid_ = threadIdx.x, nthreads = 256 * SP;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int chain = wave / SP, sub = wave % SP;
  const int X = A.X, ld_x = A.ld_x, ld_h = A.ld_h, LH = A.LH;
  const int T = 16 * ld_h;
  float *s_x = smem;                    // [16][ld_x] normalised obs
  float *s_yv = s_x + 16 * ld_x;        // [2][16][4]
  float *s_dyv = s_yv + 128;            // [2][16][4]
  float *s_tg = s_dyv + 128;            // [16] targets
  float *s_ls = s_tg + 16;              // [32] loss partials
  float *s_st = s_ls + 32;              // 4*LH tiles: z1 h1 z2 h2
  float *s_pp = s_st + 4 * LH * T;      // 4 delta tiles
  const int net = chain & 1;
  float *zb = s_st + (2 * net) * LH * T, *hb = zb + LH * T;
  const float *params = A.cr.params + (long long)net * A.cr.net_stride;
  const int CL = A.cr.n_layers;
  const float invB = 1.0f / (float)A.batch;
  float *slab = A.slabs + (long long)blockIdx.x * 2 * A.cr.n_params + (long long)net * A.cr.n_params;
  float loss = 0.f;
  bool first = true;
  const long long n_tiles = (A.batch + 15) >> 4;
#pragma nounroll
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, first = false) {
    const int tid = opaque(tid_), lane = tid & 63;
    const long long j0 = tile * 16;
    WSet<HT, SP> R;
    if (chain < 2) chain_fwd_prefetch<HT, SP, WIDE>(R, A.sh, params, sub, lane);
    for (int idx = tid; idx < 16 * X; idx += nthreads) {
      const int r = idx & 15, c = idx >> 4;
      const long long j = j0 + r;
      float o = 0.f;
      if (j < A.batch) o = (A.transitions[(long long)A.idx[j] * A.D + c] - A.s_mean[c]) / A.s_std[c];   // traj.observation normalised (:399)
      s_x[r * ld_x + c] = o;
    }
    if (tid < 16) s_tg[tid] = (j0 + tid < A.batch) ? A.lambda_values[A.idx[j0 + tid]] : 0.f;
    __syncthreads();
    if (chain < 2) chain_fwd_run<HT, SP, WIDE>(A.sh, params, s_x, ld_x, nullptr, nullptr, zb, hb, s_yv + net * 64, 4, ld_h, CL, sub, lane, R);
    else chain_idle_run(CL);
    if (chain < 2) chain_dgrad_prefetch<HT, SP, WIDE>(R, A.sh, params, sub, lane);
    if (tid < 32) {
      const int k = tid >> 4, r = tid & 15;
      const bool ok = j0 + r < A.batch;
      const float e = ok ? s_yv[(k * 16 + r) * 4] - s_tg[r] : 0.f;
      s_ls[tid] = 0.5f * e * e;                         // optax.l2_loss
      s_dyv[(k * 16 + r) * 4] = 0.5f * e * invB;        // d [0.5 * mean_j l2] / dv
    }
    __syncthreads();
    if (tid == 0)
      for (int i = 0; i < 32; ++i) loss += s_ls[i];
    {
      float *d0 = s_pp + (2 * net) * T, *d1 = d0 + T;
      if (chain < 2) chain_dgrad_run<HT, SP, WIDE>(A.sh, params, s_dyv + net * 64, 4, zb, d0, d1, nullptr, ld_x, ld_h, CL, sub, lane, R);
      else chain_wgrad_run<HT, SP, WIDE>(A.sh, s_x, ld_x, hb, s_dyv + net * 64, 4, d0, d1, slab, !first, ld_h, CL, sub, lane);
    }
  }
  if (tid_ == 0) A.extras[blockIdx.x] = loss;
}

__global__ void __launch_bounds__(256) k_critic_reduce(const float *slabs, const float *extras, int n_slabs, int C2, long long batch,
                                                        float *grads, float *metrics) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const float gsum = slab_sum_wg64(slabs, C2, n_slabs, i, i < C2);
  if (threadIdx.x < 64 && i < C2) grads[i] = gsum;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float a = slab_sum<16>(extras, 1, n_slabs, 0);
    metrics[0] = 0.5f * a / (float)batch;     // 0.5 * (mean l2(v1) + mean l2(v2))
  }
}

static int critic_plan(int x_dim, int critic_layers, const int *critic_dims, long long batch, MlpDev *cr, int *n_slabs, size_t *lds,
                       int *ld_x, int *ld_h, int *LH, long long *total) {
  MBPO_REQUIRE(x_dim > 0 && batch > 0 && critic_dims, MBPO_ERR_ARG, "critic: bad sizes");
  MBPO_REQUIRE(critic_layers >= 2 && critic_layers <= MBPO_MAX_LAYERS, MBPO_ERR_ARG, "critic: need at least one hidden layer");
  MBPO_REQUIRE(critic_dims[0] == x_dim && critic_dims[critic_layers] == 1, MBPO_ERR_ARG, "critic must map [x] -> [1]");
  MBPO_REQUIRE(bptt_hidden(critic_dims, critic_layers) == 64, MBPO_ERR_UNSUPPORTED, "critic: hidden layers must all be 64 wide");
  mbpo_mlp_desc md;
  md.net_stride = 0; md.n_nets = 1; md.params = (const float *)16; md.n_layers = critic_layers; md.activation = 0;
  for (int l = 0; l <= critic_layers; ++l) md.dims[l] = critic_dims[l];
  int rc = mbpo_make_mlp_dev(&md, cr, "critic");
  if (rc != MBPO_OK) return rc;
  *LH = critic_layers - 1;
  *ld_x = ((x_dim + 3) & ~3) + 4;
  *ld_h = 68;
  *lds = sizeof(float) * (16ull * *ld_x + 128 + 128 + 16 + 32 + (size_t)(4 * *LH + 4) * 16 * *ld_h);
  long long tiles = (batch + 15) / 16, cap = 1LL * bptt_num_cus();   // one 1024-thread workgroup fills a CU: one slab per CU (see ppo.hip)
  *n_slabs = (int)(tiles < cap ? tiles : cap);
  *total = (long long)*n_slabs * 2 * cr->n_params + ((*n_slabs + 3) & ~3);
  return MBPO_OK;
}

extern "C" int64_t mbpo_critic_workspace_floats(int32_t x_dim, int32_t critic_layers, const int32_t *critic_dims, int64_t batch) {
  MlpDev cr; int ns, ldx, ldh, lh; size_t lds; long long total;
  int rc = critic_plan(x_dim, critic_layers, critic_dims, batch, &cr, &ns, &lds, &ldx, &ldh, &lh, &total);
  if (rc != MBPO_OK) return rc;
  return total;
}

extern "C" int mbpo_critic_grads(const float *critic_params, int32_t x_dim, int32_t critic_layers, const int32_t *critic_dims,
                                 int32_t activation, const float *transitions, int32_t row_len, const float *lambda_values,
                                 const int32_t *idx, int64_t batch, const float *state_mean, const float *state_std, float *grads,
                                 float *metrics, float *workspace, void *stream) {
  CriticArgs A;
  int ns; size_t lds; long long total;
  int rc = critic_plan(x_dim, critic_layers, critic_dims, batch, &A.cr, &ns, &lds, &A.ld_x, &A.ld_h, &A.LH, &total);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(critic_params && transitions && lambda_values && idx && state_mean && state_std && grads && metrics && workspace,
               MBPO_ERR_ARG, "critic_grads: null pointer");
  MBPO_REQUIRE(activation >= 0 && activation <= 2 && row_len >= x_dim, MBPO_ERR_ARG, "critic_grads: bad activation/row_len");
  A.cr.params = critic_params; A.cr.act = activation; A.cr.n_nets = 2; A.cr.net_stride = A.cr.n_params;
  A.X = x_dim; A.D = row_len; A.transitions = transitions; A.lambda_values = lambda_values; A.s_mean = state_mean; A.s_std = state_std;
  A.idx = idx; A.batch = batch; A.slabs = workspace; A.extras = workspace + (long long)ns * 2 * A.cr.n_params;
  A.sh = NetShape{x_dim, critic_layers, 1, activation};
  const bool wide = net_is_wide(A.sh);
  rc = wide ? mbpo_ensure_lds<k_critic_fwd_bwd<64, 4, true>>(lds, "critic_grads") : mbpo_ensure_lds<k_critic_fwd_bwd<64, 4, false>>(lds, "critic_grads");
  if (rc != MBPO_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (wide) hipLaunchKernelGGL((k_critic_fwd_bwd<64, 4, true>), dim3(ns), dim3(1024), lds, st, A);
  else hipLaunchKernelGGL((k_critic_fwd_bwd<64, 4, false>), dim3(ns), dim3(1024), lds, st, A);
  const int C2 = 2 * A.cr.n_params;
  hipLaunchKernelGGL(k_critic_reduce, dim3((C2 + 63) / 64), dim3(256), 0, st, (const float *)A.slabs, (const float *)A.extras, ns, C2,
                     (long long)batch, grads, metrics);
  MBPO_CHECK_LAUNCH("critic_grads");
  return MBPO_OK;
}

// sac.hip — S3-S8: SAC sgd_step as three kernels (see include/mbpo_hip.h for the flat state layout).
//
//   k_sac_fwd_bwd   2 workgroups per 16-sample tile: a CRITIC role (pi(s') fwd, target-Q fwd, Q fwd+bwd) and an
//                   ACTOR role (pi(s) fwd+bwd, alpha loss, Q fwd + input-gradient bwd).  All three losses are taken
//                   at the OLD parameters (sac/sac.py:234-258), so the roles are independent.  fp32 MFMA; activations,
//                   pre-activations and deltas live in LDS; each workgroup writes its weight gradients to a private
//                   SLAB laid out like the flat params (no atomics: the cross-tile sum is a fixed-order reduce ->
//                   bitwise reproducible).
//   k_sac_reduce    grads[i] = sum over tiles of slab[t][i]; loss metrics; per-group sum-of-squares partials.
//   k_sac_apply     clip_by_global_norm + AdamW per optimizer group + Polyak on the critics.
//   k_sac_reduce_apply  (the default path, 2 launches per sgd_step instead of 3) both of the above in ONE launch: the optimizer step
//                   is applied UNCLIPPED, the previous state goes to an undo log, and the clip decision — which needs the global
//                   norm, i.e. every block's partial — is resolved by the NEXT consumer of the parameters: the prologue of the
//                   next k_sac_fwd_bwd, or k_sac_finalize.  clip_by_global_norm(max_grad_norm = 1e5, the reference default,
//                   sac/sac.py:96) practically never triggers; when it does, the consumer recomputes the clipped step from the
//                   undo log (same formulas, same order: bit-identical to the three-launch path).  A device-wide meeting point
//                   inside one launch was measured SLOWER than the kernel boundary it removes (round 1: 48 vs 40 us).
//
// Algorithmic work per sample per sgd_step: 2*(5P + 12Q) FLOP — latency-bound at B=256, hence the
// few fat launches and the slab scheme instead of a tree of small kernels.
#include "common.hpp"
#include "chain_run.hpp"
#include "p2p.hpp"
#include "sac_layered.hpp"
#include <string.h>
#include <stdlib.h>
int mbpo_p2p_make_dev(const mbpo_p2p_desc *d, P2pDev *P);

#include "sac_shared.hpp"
#include "sac_lean.hpp"

// What one wave group walks in one phase, filled in by the host (sac_chain_table): the kernel fetches its entry with one
// scalar load instead of ~200 scalar instructions of role/phase/chain case analysis per phase.
struct SacChainDesc {
  int mode;        // CH_*
  int netid;       // 0 = policy shape, 1 = critic shape
  int base_sel;    // parameters relative to: 0 = params, 1 = target_q
  int param_off;   // floats
  int x, ldx;      // LDS offsets (floats from the dynamic-LDS base), -1 = none
  int pp0, pp1, zb, hb, y, dx;
  int slab_sel;    // 0 none, 1 = policy slab, 2 = critic slab
  int slab_off;    // floats inside the tile's slab
  int tp0, tp1;    // CH_FWD with a tangent (jvp): the tangent's ping-pong hidden tiles (its output goes to `dx`), -1 = none
};


struct SacArgs {
  MlpDev pi, q, qt;
  NetShape sh_pi, sh_q;
  SacChainDesc tab[3][5][4];   // [table role][phase (index nph = idle)][chain]; table role 2 exists only with `split`
  int X, U, B, D;
  const float *batch, *norm_mean, *norm_std, *log_alpha;
  const float *noise_alpha, *noise_critic, *noise_actor;
  unsigned long long seed, offset;
  const unsigned long long *rng_dev;
  float discounting, reward_scaling, target_entropy;
  int neq;                      // non_equidistant_time (losses.py:90-98)
  float neq_cd, neq_tl, neq_tu, neq_dt;
  float *slab_pi, *slab_q, *slab_ex;
  int ld_x, ld_xu, ld_h, ld_y, LH;
  unsigned int *p2p_epoch;      // multi-GPU peer exchange: [0] += 1, [1] += p2p_blocks at the start of every step (or NULL)
  unsigned int p2p_blocks;
  unsigned long long *stamps;   // measurement hook (mbpo_debug_set_stamps): [2 roles][16] s_memtime values of tile 0, or NULL
  SacOptArgs opt;               // clip check of the previous speculative step
  float *step_count_rw;         // optimizer count: bumped by block 0 of every fwd/bwd launch
  int split;                    // 1: THREE workgroups per tile — critic 0, actor(+alpha), critic 1 (see k_sac_fwd_bwd)
  int thin;                     // 1: layers with <= 8 inputs / <= 4 outputs by VALU around the runners (chain_run.hpp 'thin layers')
  int jvp;                      // 1 (u_dim == 1): the actor role gets dQ/da in FORWARD mode — the tangent rides along the critics' forward
                                // pass (chain_run.hpp JVP) and the critics' input-gradient phase does not exist: 12 dependent layer
                                // steps instead of 16 on the role that bounds the kernel
};

// Timeline stamps for DESIGN.md's phase breakdown: one s_memtime per phase boundary, written by thread 0 of tile 0.
#define SAC_STAMP(i)                                                                   \
  if (A.stamps && tile == 0 && tid == 0) {                                             \
    unsigned long long t_;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
    A.stamps[role * 16 + (i)] = t_;                                                    \
  }

static unsigned long long *g_sac_stamps = nullptr;
// Measurement / test hook (not part of include/mbpo_hip.h): 0 = always the generic k_sac_fwd_bwd, 1 = the specialised k_sac_lean
// where it applies, -1 = the MBPO_SAC_LEAN environment default (on).  tests/test_gpu_sac_lean.py flips it inside one process.
static int g_sac_lean = -1;
extern "C" int mbpo_debug_set_sac_lean(int mode) {
  g_sac_lean = mode;
  return MBPO_OK;
}
// Measurement hook (not part of include/mbpo_hip.h): device buffer of >= 32 uint64 that k_sac_fwd_bwd fills with
// s_memtime stamps at its phase boundaries (tile 0, both roles); NULL switches the stamps off.
extern "C" int mbpo_debug_set_stamps(void *buf) {
  g_sac_stamps = (unsigned long long *)buf;
  return MBPO_OK;
}


// Workgroup = 8 waves = one 16-sample tile in one ROLE.  Wave w = (chain c = w/2, sub = w%2): each network chain is
// shared by 2 waves (column slices of H/2), chains advance side by side in lockstep, one workgroup barrier per layer
// (wave_mlp.hpp "Lockstep groups").  The per-sgd_step critical path is ~16 layer-steps of ~32 MFMAs per wave.
// (16 waves x 4 per chain halves the MFMAs per step again but caps a wave at 128 VGPRs: hipcc spilled ~850 registers.)
//   CRITIC role                                     ACTOR(+alpha) role
//   P1  c0: pi(s') fwd  c1: Q1(s,a) fwd+store       P1  c0: pi(s) fwd+store
//                       c2: Q2(s,a) fwd+store
//   -- sample a', log pi'                           -- sample alpha/actor actions
//   P2  c0: Qtgt1(s',a')  c1: Qtgt2(s',a')          P2  c0: Q1(s,a~) fwd (z kept)  c1: Q2(s,a~)
//   -- target y, errors, dL/dq                      -- min_q, dL/dq
//   P3  per layer: c0/c1: Q1/Q2 dgrad               P3  per layer: c0/c1: Q1/Q2 input-gradient
//                  c2/c3: Q1/Q2 wgrad               -- dL/dlogits
//                                                   P4  per layer: c0: pi dgrad   c1: pi wgrad
// SP = waves per chain, NCH = chain slots: NCH x SP waves; WIDE: chain_run.hpp fast_shape.  NCH = 2 is for the `split` launch, where
// no role carries more than two chains at a time: at hidden width 128 that makes room for 4 waves per chain (8 waves, 2 per SIMD,
// 256 VGPRs each) and with them for the register-image weight prefetch (a lane's share of a 128-wide layer is 64 weights + 2
// biases per image, two images) that 16 waves x 128 VGPRs cannot hold.
template <int H, int SP, bool WIDE, int NCH = 4, bool THIN = false>
__global__ void __launch_bounds__(64 * SP * NCH) k_sac_fwd_bwd(SacArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int HT = H / 16;
  const int tid_ = threadIdx.x, nthreads = 64 * SP * NCH;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int chain = wave / SP, sub = wave % SP;
  // Two workgroups per tile (critic role, actor role), or — `split`, used at hidden width 128 where a phase is bound by the
  // fp32-MFMA rate of its CU — three: the critic role once per critic.  Each critic workgroup runs pi(s') and BOTH target critics
  // (the target needs their minimum) but the forward, dgrad and wgrad of ITS critic only: no phase carries more than two chains, the
  // critic backward 2 x 256 MFMAs per layer on a CU instead of 4 x 256.  kq = the critic this workgroup owns (-1: both).
  int role, tile, trole, kq = -1;
  if (A.split) {
    trole = blockIdx.x % 3;
    tile = blockIdx.x / 3;
    role = trole == 1 ? 1 : 0;
    kq = trole == 1 ? -1 : (trole >> 1);
  } else {
    role = blockIdx.x & 1;   // 0 = critic, 1 = actor(+alpha)
    tile = blockIdx.x >> 1;
    trole = role;
  }
  const int X = A.X, U = A.U, D = A.D, B = A.B;
  const int row0 = tile * 16;
  const int ld_x = A.ld_x, ld_xu = A.ld_xu, ld_h = A.ld_h, ld_y = A.ld_y, LH = A.LH;
  const int T = 16 * ld_h;  // one hidden tile

  // ---- LDS carve (every region a multiple of 4 floats: rows stay 16-byte aligned) ----
  float *s_row = smem;                      // [16][D] (flat copy of the tile's transitions)
  const int D4 = D;
  float *s_sn = s_row + 16 * ((D + 3) & ~3);  // [16][ld_x]   normalised obs
  float *s_sn2 = s_sn + 16 * ld_x;          // [16][ld_x]   normalised next obs
  float *s_qin = s_sn2 + 16 * ld_x;         // [16][ld_xu]  [sn, a]
  float *s_qin2 = s_qin + 16 * ld_xu;       // [16][ld_xu]  [s'n, a']
  float *s_pp = s_qin2 + 16 * ld_xu;        // 4 tiles: ping-pong hidden / delta tiles
  float *s_store = s_pp + 4 * T;            // 4*LH tiles: stored z / h
  float *s_y = s_store + 4 * LH * T;        // [3][16][ld_y]  network outputs
  float *s_dy = s_y + 3 * 16 * ld_y;        // [2][16][ld_y]  output gradients
  float *s_dx = s_dy + 2 * 16 * ld_y;       // [2][16][ld_xu] critic input gradients (actor role)
  const int U4 = (16 * U + 3) & ~3;
  float *s_eps = s_dx + 2 * 16 * ld_xu;     // [16][U] ...
  float *s_a = s_eps + U4;
  float *s_sig = s_a + U4;
  float *s_raw = s_sig + U4;
  float *s_lp = s_raw + U4;
  float *s_lpa = s_lp + U4;
  float *s_scal = s_lpa + U4;               // [4][16]
  float *s_gn = s_scal + 64;                // [4]: group norms of the previous speculative optimizer step

  {
    const int tid = tid_;
    SAC_STAMP(0);
  }
  // Clip check of the previous speculative optimizer step (k_sac_reduce_apply).  Nothing here may WAIT for a load (a branch on these
  // words at this point put a whole cold-miss latency in front of the tile loads: +2 us per launch), nothing may add work to a wave
  // the other waves wait for, and nothing may live in registers across the phase loop (this kernel is short of scalar registers;
  // every such value costs the lone waves spill code in every phase — measured 0.2-0.4 us each).  So: ONE scalar load of eight
  // words here (sequence numbers + the two slots of the quick verdict); after the first barrier they are folded into one flag —
  // "some workgroup of the reduce launch found its sum of squares at or above limit / n_workgroups, or not finite, and raised the
  // slot's word to +inf with a plain store" (group_sumsq): practically never at the reference's max_grad_norm = 1e5.  Only then,
  // behind the phases, are the canonical norms formed (fixed order: the clip decision and the clipped step are bit-identical to
  // mbpo_sac_apply's).
  // Warm the scalar cache with the whole kernel-argument block.  The phase loop fetches its operands lazily — the 64-byte chain
  // descriptor of each phase, network shapes, slab pointers — and every first touch of a kernarg line was a miss to memory
  // (the block is fresh for every launch): ~2 k cycles of "chain set-up" per phase, on every wave.  The LAST wave, which has no
  // tile element to load, reads one dword of every 64-byte line here (scalar loads: they fill the scalar cache the whole CU
  // shares) and pays the one cold-miss latency while the other waves wait for the tile anyway.
  if (wave == NCH * SP - 1) {
    const int *ka = reinterpret_cast<const int *>(&A.pi);
    int acc = 0;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(SacArgs) / 64); ++i) acc += ka[i * 16];
    asm volatile("" ::"s"(acc));
  }
  const uint4 qw0 = *reinterpret_cast<const uint4 *>(A.opt.seq), qw1 = *reinterpret_cast<const uint4 *>(A.opt.seq + 4);
  const float count_in = A.step_count_rw[0];
  const unsigned int ep0_in = A.p2p_epoch ? A.p2p_epoch[0] : 0u, ep1_in = A.p2p_epoch ? A.p2p_epoch[1] : 0u;
  bool maybe_clip = false;
  // requested now, consumed after the first layer phase: the two scalar loads overlap the tile load instead of preceding it
  const float log_alpha_top = A.log_alpha[0];   // requested now, consumed after the first layer phase
  const float invB = 1.0f / (float)B;
  const float *pi_p = A.pi.params, *q1_p = A.q.params, *q2_p = A.q.params + A.q.net_stride;
  const float *t1_p = A.qt.params, *t2_p = A.qt.params + A.qt.net_stride;
  const int QL = A.q.n_layers, PL = A.pi.n_layers;
  const int Lmax = QL > PL ? QL : PL;
  float *y_pi = s_y, *y_q1 = s_y + 16 * ld_y, *y_q2 = s_y + 2 * 16 * ld_y;
  // stored activations.  critic role: Q1 (z,h), Q2 (z,h);  actor role: policy (z,h), Q1 z, Q2 z
  const int o_sn = (int)(s_sn - smem), o_sn2 = (int)(s_sn2 - smem), o_qin = (int)(s_qin - smem), o_qin2 = (int)(s_qin2 - smem);
  const int o_pp = (int)(s_pp - smem), o_y = (int)(s_y - smem), o_dy = (int)(s_dy - smem), o_dx = (int)(s_dx - smem);
  const int o_st0 = (int)(s_store - smem), o_st1 = o_st0 + LH * T, o_st2 = o_st0 + 2 * LH * T, o_st3 = o_st0 + 3 * LH * T;

  // Phases (one barrier per layer step inside a phase, one after each elementwise section E):
  //   critic role:  E load | F0: c0 pi(s') fwd, c1/c2 Q1/Q2(s,a) fwd | E sample a' | F1: c0/c1 Qtgt1/2(s',a') fwd | E targets
  //                 | B2: c0/c1 Q1/Q2 dgrad, c2/c3 Q1/Q2 wgrad | E loss partials
  //   actor role:   E load | F0: c0 pi(s) fwd | E sample a | F1: c0/c1 Q1/Q2(s,a) fwd | E dL/dq | B2: c0/c1 Q input-grad
  //                 | E dL/dlogits | B3: c0 pi dgrad, c1 pi wgrad | E loss partials
  // What this wave walks in the current phase (wave-uniform scalars; LDS operands as offsets from smem, -1 = none).
  WSet<HT, SP> R;
  constexpr int H_ = H;
  constexpr bool THIN_OK = THIN;     // thin layers by VALU (chain_run.hpp): a kernel variant of its own, chosen by the host
  static_assert(!THIN || (H_ == 64 && SP == 4 && !WIDE && NCH == 2), "thin layers: H = 64, four waves per chain, two chain slots");
  float tw[THIN_KMAX + 1];                 // this lane's share of the next phase's thin layer (a column of W0 + bias, or a row of Wout)
#pragma unroll
  for (int k = 0; k <= THIN_KMAX; ++k) tw[k] = 0.f;
  int mode = CH_IDLE, netid = 0;
  const float *cparams = pi_p;
  float *cslab = nullptr;
  int cx = -1, cldx = ld_x, cpp0 = -1, cpp1 = -1, czb = -1, chb = -1, cy = -1, cdx = -1, ctp0 = -1, ctp1 = -1;
#define P(off) ((off) < 0 ? (float *)nullptr : smem + (off))
  const int nph = (role == 0 || A.jvp) ? 3 : 4;
  // At most two passes over the phases: the second one only after a clip fix-up of the previous optimizer step (rare), whose code
  // sits behind the phase loop so that it costs the loop neither registers nor instruction-cache lines.
  // The clip decision must not touch the phase loop's control flow: a `break` on it, or a loop bound that depends on it, made hipcc
  // duplicate the runners' code (+2.8 us per launch).  So the phases always run to the end; the norms (formed on the side by an
  // idle wave during phase 0) are looked at AFTER the loop, and in the rare case of a clip the optimizer step is fixed up and ALL
  // phases run a second time — whatever the first pass wrote (slabs, loss partials) is overwritten.
  // (A loop around the phase loop for the second pass cost the first pass 2 us — hipcc compiles a phase loop nested in another
  // loop worse than one that stands alone — so the phases are a lambda expanded twice: the second copy is cold code behind the first.)
  auto run_phases = [&](const float log_alpha_v) __attribute__((always_inline)) {
#pragma nounroll
  for (int ph = -1; ph < nph; ++ph) {
    const int tid = opaque(tid_), lane = tid & 63;   // keeps per-lane addresses of all phases from being hoisted and spilled
    if (ph >= 0) {
      int len = (role == 0) ? (ph == 0 ? Lmax : QL) : ((ph == 0 || ph == nph - 1) ? PL : QL);
      const NetShape sh = netid == 0 ? A.sh_pi : A.sh_q;
      constexpr bool thin = THIN_OK;
      if constexpr (THIN_OK) {
        {
          // thin layers by VALU (chain_run.hpp): the chain's four waves form layer 0 of a forward chain, or delta_{L-2} and the
          // output layer's weight gradient of a backward pair, from what the preceding section left in LDS; then the runners
          // walk the H x H layers only (forward: L - 1 barriers instead of L, backward: L - 2)
          if (mode == CH_FWD)
            thin_fwd_first(tw, sh.K_in, sh.act, P(cx), cldx, chb >= 0 ? P(chb) : P(cpp0), P(czb), P(ctp0), X, ld_h, sub, lane);
          else if (mode == CH_DGRAD)
            thin_dgrad_out(tw, sh.N_out, sh.act, P(cy), ld_y, P(czb) + (sh.L - 2) * T, ((sh.L - 1) & 1) ? P(cpp1) : P(cpp0), ld_h, sub, lane);
          else if (mode == CH_WGRAD)
            thin_wgrad_out<H>(sh.N_out, P(chb) + (sh.L - 2) * T, ld_h, P(cy), ld_y,
                              cslab + (sh.K_in * H + H) + (sh.L - 2) * (H * H + H), sub, lane);
          __syncthreads();
          len -= (ph == nph - 1) ? 2 : 1;
        }
      }
      if (mode == CH_FWD)
        chain_fwd_run<HT, SP, WIDE, !WIDE, THIN_OK>(sh, cparams, P(cx), cldx, P(cpp0), P(cpp1), P(czb), P(chb), P(cy), ld_y, ld_h, len, sub, lane, R,
                                           (A.stamps && tile == 0 && role == 1 && ph == 0 && wave == 0) ? A.stamps + 40 : nullptr,
                                           P(ctp0), P(ctp1), ctp0 >= 0 ? P(cdx) : nullptr, ld_xu, X);
      else if (mode == CH_DGRAD)
        chain_dgrad_run<HT, SP, WIDE, THIN_OK>(sh, cparams, P(cy), ld_y, P(czb), P(cpp0), P(cpp1), P(cdx), ld_xu, ld_h, len, sub, lane, R);
      else if (mode == CH_WGRAD) {
        chain_wgrad_run<HT, SP, WIDE, THIN_OK>(sh, P(cx), cldx, P(chb), P(cy), ld_y, P(cpp0), P(cpp1), cslab, false, ld_h, len, sub, lane,
                                (A.stamps && tile == 0 && role == 0 && sub == 0 && chain == (A.split ? 1 : 2)) ? A.stamps + 48 : nullptr);
      } else
        chain_idle_run(len);
      if constexpr (THIN_OK) {
        // layer 0's weight gradient: delta_0 is complete (the runners' last barrier); the dgrad and the wgrad chain of the pair
        // share input tile, delta tiles and slab, so all eight waves take one input row k (or the bias) each
        if (ph == nph - 1 && (mode == CH_WGRAD || mode == CH_DGRAD))
          thin_wgrad_first<H>(sh.K_in, P(cx), cldx, P(cpp1), ld_h, cslab, chain * SP + sub, lane);
      }
    }
    SAC_STAMP(2 * ph + 3);
    // ---- the chain this wave walks in the NEXT phase; its first layer's weights are requested now ----
    {
      const int nx = ph + 1;
      const SacChainDesc &cd = A.tab[trole][nx][chain];
      mode = cd.mode;
      netid = cd.netid;
      cparams = (cd.base_sel ? A.qt.params : A.pi.params) + cd.param_off;
      cx = cd.x; cldx = cd.ldx; cpp0 = cd.pp0; cpp1 = cd.pp1; czb = cd.zb; chb = cd.hb; cy = cd.y; cdx = cd.dx;
      ctp0 = cd.tp0; ctp1 = cd.tp1;
      cslab = cd.slab_sel == 1 ? A.slab_pi + (long long)tile * A.pi.n_params + cd.slab_off
                               : (cd.slab_sel == 2 ? A.slab_q + (long long)tile * (2 * A.q.n_params) + cd.slab_off : nullptr);
      const NetShape shn = netid == 0 ? A.sh_pi : A.sh_q;
      if constexpr (THIN_OK) {
        if (mode == CH_FWD) chain_fwd_prefetch_thin<HT, SP, WIDE>(R, shn, cparams, sub, lane, tw);
        else if (mode == CH_DGRAD) chain_dgrad_prefetch_thin<HT, SP>(R, shn, cparams, sub, lane, tw);
      } else {
        if (mode == CH_FWD) chain_fwd_prefetch<HT, SP, WIDE>(R, shn, cparams, sub, lane);
        else if (mode == CH_DGRAD) chain_dgrad_prefetch<HT, SP, WIDE>(R, shn, cparams, sub, lane);
      }
    }
    if (A.stamps && tile == 0 && tid == 0 && ph == 1) {
      unsigned long long t_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
      A.stamps[role * 16 + 14] = t_;
    }
    // ---- elementwise section after phase ph ----
    if (ph == -1) {
      // load the tile's transitions; normalise observations (q and policy both preprocess obs: sac/networks.py:76-78,96-98).
      // Every global load of this section is independent: one latency, one barrier.
      const long long nvalid = (long long)(B - row0 < 16 ? B - row0 : 16) * D;
      for (int idx = tid; idx < 16 * D; idx += nthreads) s_row[idx] = idx < nvalid ? A.batch[(long long)row0 * D + idx] : 0.f;
      for (int idx = tid; idx < 16 * X; idx += nthreads) {
        const int r = idx & 15, cc = idx >> 4;
        float o = 0.f, o2 = 0.f;
        if (row0 + r < B) {
          o = A.batch[(long long)(row0 + r) * D + cc];
          o2 = A.batch[(long long)(row0 + r) * D + X + U + 2 + cc];
        }
        if (A.norm_mean) {
          const float mu = A.norm_mean[cc], sd = A.norm_std[cc];
          o = (o - mu) / sd;
          o2 = (o2 - mu) / sd;
        }
        s_sn[r * ld_x + cc] = o;
        s_sn2[r * ld_x + cc] = o2;
        s_qin[r * ld_xu + cc] = o;
        s_qin2[r * ld_xu + cc] = o2;
      }
      if (role == 0) {
        for (int idx = tid; idx < 16 * U; idx += nthreads) {
          const int r = idx & 15, d = idx >> 4;
          s_qin[r * ld_xu + X + d] = (row0 + r < B) ? A.batch[(long long)(row0 + r) * D + X + d] : 0.f;  // transitions.action
        }
      }
      // The noise of the sampling sections depends on (seed, offset, element) only: it is drawn HERE, by waves that have no tile
      // element to load, while the tile is in flight — Philox + Box-Muller (logf, sqrtf, cosf: ~250 instructions) used to sit in
      // the sampling sections, on the lone wave every other wave waits for.  critic role: next-action noise -> s_eps;
      // actor role: actor-loss noise -> s_eps, alpha-loss noise -> s_lpa (replaced by the log-prob when it is used).
      {
        const int half = nthreads / 2;
        const RngKey rkq = rng_resolve(A.seed, A.offset, A.rng_dev);
        if (tid >= half) {
          const bool second = tid >= half + half / 2;
          if (!second || role == 1) {
            for (int i2 = tid - half - (second ? half / 2 : 0); i2 < 16 * U; i2 += half / 2) {
              const int r = i2 & 15, d = i2 >> 4, idx = r * U + d;
              const long long nidx = (long long)(row0 + r) * U + d;
              float e = 0.f;
              if (row0 + r < B) {
                const float *given = second ? A.noise_alpha : (role == 0 ? A.noise_critic : A.noise_actor);
                const unsigned int stream = second ? MBPO_STREAM_SAC_ALPHA : (role == 0 ? MBPO_STREAM_SAC_CRITIC : MBPO_STREAM_SAC_ACTOR);
                e = given ? given[nidx] : philox_normal(rkq.seed, rkq.offset, stream, (unsigned long long)nidx);
              }
              (second ? s_lpa : s_eps)[idx] = e;
            }
          }
        }
      }
    }
    const float alpha = expf(log_alpha_v);
    if (ph == -1) {
    } else if (role == 0) {
      // ============================== CRITIC (sac/losses.py:74-110) ==============================
      if (ph == 0) {
        // next_action ~ policy(next_observation); next_log_prob (:80-87)
        for (int i2 = tid; i2 < 16 * U; i2 += nthreads) {
          const int r = i2 & 15, d = i2 >> 4, idx = r * U + d;
          const float eps = s_eps[idx];      // drawn in the tile section
          ActSample sm = normal_tanh_sample(y_pi[r * ld_y + d], y_pi[r * ld_y + U + d], eps);
          s_qin2[r * ld_xu + X + d] = sm.a;  // postprocess(next_action)
          s_lp[idx] = sm.lp;
        }
        if (tid < 32) {  // keep q_old_action (:78-79) out of the way of the target critics' outputs
          const int k = tid >> 4, r = tid & 15;
          if (kq < 0 || k == kq) s_scal[32 + tid] = (k == 0 ? y_q1 : y_q2)[r * ld_y];
        }
      } else if (ph == 1) {
        float e2 = 0.f;
        if (tid < 32) {
          const int k = tid >> 4, r = tid & 15;
          const bool ok = row0 + r < B;
          float nlp = 0.f;
          for (int d = 0; d < U; ++d) nlp += s_lp[r * U + d];
          const float nq = fminf(y_q1[r * ld_y], y_q2[r * ld_y]);
          const float next_v = nq - alpha * nlp;                                                   // :89
          const float rew = s_row[r * D4 + X + U], disc = s_row[r * D4 + X + U + 1];
          float gamma = A.discounting;
          if (A.neq) {                                                                             // :90-96
            const float pseudo = s_row[r * D4 + X + U - 1];                                        // transitions.action[..., -1]
            float tfa = (A.neq_tu - A.neq_tl) / 2.0f * pseudo + (A.neq_tu + A.neq_tl) / 2.0f;
            tfa = floor_divide_f(tfa, A.neq_dt) * A.neq_dt;
            gamma = expf(-A.neq_cd * tfa);
          }
          const float target = rew * A.reward_scaling + disc * gamma * next_v;                     // :101-103
          const float trunc = s_row[r * D4 + D - 1];
          const float err = (ok && (kq < 0 || k == kq)) ? (s_scal[32 + tid] - target) * (1.f - trunc) : 0.f;   // q_error :104-108 (own critic)
          e2 = err * err;
          // loss = 0.5*mean(err^2) over [B,2]  ->  dL/dq = err*(1-trunc)/(2B)
          s_dy[(k * 16 + r) * ld_y] = err * (1.f - trunc) * (0.5f * invB);
        }
        // the tile's loss partial, formed here from the producers' registers (a shuffle tree over the first wave: fixed order)
        // instead of by one thread re-reading 32 LDS words at the very end of the launch
        if (wave == 0) {
          const float acc = wave_sum64(e2);
          if (tid == 0) A.slab_ex[tile * 4 + (kq == 1 ? 3 : 0)] = acc;     // (split: the two critics' partials in slots 0 and 3, added by the reduce)
        }
      }
    } else {
      // ============================== ACTOR + ALPHA (sac/losses.py:61-72, 112-125) ==============================
      if (ph == 0) {
        // two independent samples per element: the first half of the workgroup draws the actor-loss sample, the second half
        // the alpha-loss sample (different waves: the two instruction streams run side by side)
        const int half = nthreads / 2;
        const bool second = tid >= half;
        for (int i2 = second ? tid - half : tid; i2 < 16 * U; i2 += half) {
          const int r = i2 & 15, d = i2 >> 4, idx = r * U + d;
          const float loc = y_pi[r * ld_y + d], raw = y_pi[r * ld_y + U + d];
          if (second) {
            const float e_al = s_lpa[idx];     // drawn in the tile section
            ActSample sal = normal_tanh_sample(loc, raw, e_al);   // alpha loss sample (:66-68)
            s_lpa[idx] = sal.lp;
          } else {
            const float e_ac = s_eps[idx];     // drawn in the tile section
            ActSample sac = normal_tanh_sample(loc, raw, e_ac);   // actor loss sample (:117-119)
            s_lp[idx] = sac.lp;
            s_eps[idx] = e_ac;
            s_a[idx] = sac.a;
            s_sig[idx] = sac.sigma;
            s_raw[idx] = raw;
            s_qin[r * ld_xu + X + d] = sac.a;  // postprocess(action) (:120)
          }
        }
      } else if (ph == 1) {
        float l_al = 0.f, l_ac = 0.f;
        if (tid < 16) {
          const int r = tid;
          const bool ok = row0 + r < B;
          float lp_al = 0.f, lp_ac = 0.f;
          for (int d = 0; d < U; ++d) {
            lp_al += s_lpa[r * U + d];
            lp_ac += s_lp[r * U + d];
          }
          // alpha_loss = alpha * stop_gradient(-log_prob - target_entropy); d/dlog_alpha = the same value   (:70-72)
          l_al = ok ? alpha * (-lp_al - A.target_entropy) : 0.f;
          const float q0 = y_q1[r * ld_y], q1 = y_q2[r * ld_y];
          const float mq = fminf(q0, q1);
          l_ac = ok ? (alpha * lp_ac - mq) : 0.f;   // actor_loss = alpha*log_prob - min_q   (:123-124)
          // d(mean(-min_q))/dq_k: -1/B on the arg-min critic (ties split evenly, as jnp.min's gradient does)
          float g0 = 0.f, g1 = 0.f;
          if (ok) {
            if (q0 < q1) g0 = -invB;
            else if (q1 < q0) g1 = -invB;
            else g0 = g1 = -0.5f * invB;
          }
          if (A.jvp) {
            // forward mode (u_dim == 1): the critics' forward pass carried d q_k / d a along (s_dx[k][r][0]); the actor loss's
            // gradient with respect to the action and, from it, the logits' gradients follow here — no critic backward phase
            const float dLda = g0 * s_dx[r * ld_xu] + g1 * s_dx[(16 + r) * ld_xu];
            const float a = s_a[r], sg = s_sig[r], eps = s_eps[r], raw = s_raw[r];
            const float gz = dLda * (1.f - a * a) + alpha * invB * 2.f * a;
            const float gsig = gz * eps - alpha * invB / sg;
            s_dy[r * ld_y] = ok ? gz : 0.f;                              // d/dloc
            s_dy[r * ld_y + 1] = ok ? gsig * fast_sigmoid(raw) : 0.f;    // d/draw = d/dsigma * softplus'(raw)
          } else {
            s_dy[r * ld_y] = g0;
            s_dy[(16 + r) * ld_y] = g1;
          }
        }
        if (wave == 0) {     // the tile's loss partials from the producers' registers (shuffle tree: fixed order)
          const float al = wave_sum64(l_al), ac = wave_sum64(l_ac);
          if (tid == 0) {
            A.slab_ex[tile * 4 + 1] = ac;
            A.slab_ex[tile * 4 + 2] = al;
          }
        }
      } else if (ph == 2 && !A.jvp) {
        for (int i2 = tid; i2 < 16 * U; i2 += nthreads) {
          const int r = i2 & 15, d = i2 >> 4, idx = r * U + d;
          const bool ok = row0 + r < B;
          const float a = s_a[idx], sg = s_sig[idx], eps = s_eps[idx], raw = s_raw[idx];
          const float dLda = s_dx[r * ld_xu + X + d] + s_dx[(16 + r) * ld_xu + X + d];
          // z = loc + sigma*eps;  log_prob = const - log(sigma) - log(1 - tanh(z)^2)  =>  dlp/dz = 2a, dlp/dsigma = -1/sigma
          const float gz = dLda * (1.f - a * a) + alpha * invB * 2.f * a;
          const float gsig = gz * eps - alpha * invB / sg;
          s_dy[r * ld_y + d] = ok ? gz : 0.f;                              // d/dloc
          s_dy[r * ld_y + U + d] = ok ? gsig * fast_sigmoid(raw) : 0.f;    // d/draw = d/dsigma * softplus'(raw)
        }
      }
    }
    __syncthreads();
    if (ph == -1) {
      // the words requested at the top have arrived with the tile: use them up (idempotent: a second pass repeats the same stores)
      {
        const unsigned int seq_issued = qw0.x, seq_resolved = qw0.y;
        const bool odd = (seq_issued & 1u) == 0u;        // the step being checked is number seq_issued - 1: its slot is the OTHER one
        const float q0 = __uint_as_float(odd ? qw1.y : qw0.z), q1 = __uint_as_float(odd ? qw1.z : qw0.w), q2 = __uint_as_float(odd ? qw1.w : qw1.x);
        const float lim = (A.opt.max_norm / A.opt.grad_scale) * (A.opt.max_norm / A.opt.grad_scale) * 0.9998f;
        maybe_clip = seq_issued != seq_resolved && !(q0 < lim && q1 < lim && q2 < lim);
      }
      if (blockIdx.x == 0 && tid == 0) {
        // this step's slot: published for the reduce launch, emptied for its atomics
        const unsigned int slot = qw0.x & 1u;
        A.opt.slot_word[0] = slot;
        float *q = reinterpret_cast<float *>(A.opt.seq) + 2 + 3 * slot;
        q[0] = 0.f; q[1] = 0.f; q[2] = 0.f;
        // optax's count: one per sgd_step.  Nothing in this kernel uses it; the reduce / apply launches of this step read the final
        // value (kernel boundary), the clip fix-up of the NEXT launch reads the copy k_sac_reduce_apply keeps.  (A load -> add ->
        // store chain at the very top made block 0 wait a cold-miss latency before its first tile load.)
        A.step_count_rw[0] = count_in + 1.0f;
        if (A.p2p_epoch) {   // multi-GPU peer exchange: the epoch is stable while the reduce / gather launches of this step read it
          A.p2p_epoch[0] = ep0_in + 1u;
          A.p2p_epoch[1] = ep1_in + A.p2p_blocks;
        }
      }
    }

    SAC_STAMP(2 * ph + 4);
  }
  };
  run_phases(log_alpha_top);
  if (!maybe_clip) return;        // the common case: this launch is done
  // RARE from here on.  Canonical norms (fixed order), the exact decision, and if a group really clips: fix its step up from the undo
  // log and run ALL phases again — what the first pass wrote (slabs, loss partials) is overwritten.
  sac_group_norms(A.opt, s_gn, tid_);
  __syncthreads();
  if (s_gn[0] < A.opt.max_norm && s_gn[1] < A.opt.max_norm && s_gn[2] < A.opt.max_norm) return;
  if (blockIdx.x == 0 && tid_ == 0) A.opt.seq[SAC_CTL_CLIP_EVENTS] += 1u;     // (one writer per launch; read by the host between epochs)
  sac_clip_fixup(A.opt, s_gn, opaque(tid_), nthreads);
  __threadfence();
  __syncthreads();
  run_phases(A.log_alpha[0]);
}

// ------------------------------------------------------------------------------------------------
struct SacReduceArgs {
  const float *slab_pi, *slab_q, *slab_ex;
  int n_tiles, P, Q2, B;
  float *grads, *metrics, *metrics_accum, *ss_part, *step_count;
  // sequence words {issued, resolved} of the speculative two-launch steps (SacOptArgs::seq).  The reduce launch of EVERY step
  // flavour records that whatever was pending before this step's fwd/bwd launch has been resolved by it (its prologue ran the
  // check and the fix-up) — this launch overwrites ss_part, the operands of that check, so a later reader must not repeat it on
  // the new step's partials (mbpo_sac_step -> mbpo_sac_grads + mbpo_sac_apply -> mbpo_sac_finalize).  Written by the
  // launch AFTER the fwd/bwd launch, never from inside it: late-starting workgroups read these words at their top.
  unsigned int *seq;
};

// sum-of-squares partials per workgroup and optimizer group (0 policy, 1 critics, 2 alpha), fixed order
__device__ __forceinline__ void group_sumsq(float g, int i, int P, int Q2, int NP, float *ss_part, float *quick = nullptr,
                                            float blk_lim = 0.f) {
  __shared__ float s_ss[3][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float gg = (i < NP) ? g * g : 0.f;
  const int grp = (i < P) ? 0 : (i < P + Q2 ? 1 : 2);
  float v0 = wave_sum64(grp == 0 ? gg : 0.f), v1 = wave_sum64(grp == 1 ? gg : 0.f), v2 = wave_sum64(grp == 2 ? gg : 0.f);
  if (lane == 0) {
    s_ss[0][wave] = v0;
    s_ss[1][wave] = v1;
    s_ss[2][wave] = v2;
  }
  __syncthreads();
  if (tid < 3) {
    const float part = s_ss[tid][0] + s_ss[tid][1] + s_ss[tid][2] + s_ss[tid][3];
    ss_part[blockIdx.x * 3 + tid] = part;       // the canonical partials: summed in a FIXED order by whoever needs the norm
  }
  // The quick clip test, without atomics: the groups share max_grad_norm, so if EVERY workgroup's sum over the three groups stays
  // below limit / n_workgroups, every group's total is below the limit.  A workgroup that is not (or holds a NaN) raises the
  // slot's word with a plain store (several may: same value); the reader's "word < limit" then fails and the canonical norms
  // decide.  (Round 2 first added the per-group sums up with float atomics: 309 same-line atomics per launch, and a launch does
  // not end before its last atomic has been performed — 0.5-0.9 us per update.)
  if (quick && tid == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) tot += s_ss[k][0] + s_ss[k][1] + s_ss[k][2] + s_ss[k][3];
    // +inf, not a large finite number: with max_grad_norm so large that the limits overflow to +inf (clipping "disabled") a
    // non-finite gradient must still fail the reader's `word < limit` (inf < inf is false), as the three-launch path's
    // canonical check does
    if (!(tot < blk_lim)) quick[0] = __builtin_inff();
  }
}

__global__ void __launch_bounds__(256) k_sac_reduce(SacReduceArgs A) {
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned int seq_issued = (i == NP - 1) ? A.seq[0] : 0u;      // requested beside the slab sums
  float g = 0.f;
  if (i < A.P) {
    g = slab_sum<16>(A.slab_pi, A.P, A.n_tiles, i);
  } else if (i < A.P + A.Q2) {
    const int j = i - A.P;
    g = slab_sum<16>(A.slab_q, A.Q2, A.n_tiles, j);
  } else if (i == NP - 1) {
    // (all loads of the three sums in flight together: this one thread is the kernel's critical path)
    const float ce = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 0) + slab_sum<16>(A.slab_ex, 4, A.n_tiles, 3), ac = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 1),
                al = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 2);
    const float invB = 1.0f / (float)A.B;
    g = al * invB;                          // d alpha_loss / d log_alpha
    const float m0 = 0.5f * ce * (0.5f * invB), m1 = ac * invB, m2 = al * invB;   // critic_loss = 0.5 * mean over [B,2]
    A.metrics[0] = m0;
    A.metrics[1] = m1;
    A.seq[1] = seq_issued;       // nothing is pending any more: this step is not speculative
    A.metrics[2] = m2;
    if (A.metrics_accum) {       // (no load of what was just stored: each such round trip is part of the launch's tail)
      const float a0 = A.metrics_accum[0], a1 = A.metrics_accum[1], a2 = A.metrics_accum[2], a4 = A.metrics_accum[4];
      A.metrics_accum[0] = a0 + m0;
      A.metrics_accum[1] = a1 + m1;
      A.metrics_accum[2] = a2 + m2;
      A.metrics_accum[4] = a4 + 1.0f;
    }
  }
  if (i < NP) A.grads[i] = g;
  group_sumsq(g, i, A.P, A.Q2, NP, A.ss_part);
}

// multi-GPU: the reduced local gradient goes straight into every rank's exchange region (p2p.hpp) instead of a collective
__global__ void __launch_bounds__(256) k_sac_reduce_push(SacReduceArgs A, P2pDev X) {
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned epoch = X.epoch[0];
  const unsigned int seq_issued = (i == NP - 1) ? A.seq[0] : 0u;      // requested beside the slab sums
  float g = 0.f;
  if (i < A.P) {
    g = slab_sum<16>(A.slab_pi, A.P, A.n_tiles, i);
  } else if (i < A.P + A.Q2) {
    const int j = i - A.P;
    g = slab_sum<16>(A.slab_q, A.Q2, A.n_tiles, j);
  } else if (i == NP - 1) {
    // (all loads of the three sums in flight together: this one thread is the kernel's critical path)
    const float ce = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 0) + slab_sum<16>(A.slab_ex, 4, A.n_tiles, 3), ac = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 1),
                al = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 2);
    const float invB = 1.0f / (float)A.B;
    g = al * invB;
    const float m0 = 0.5f * ce * (0.5f * invB), m1 = ac * invB, m2 = al * invB;
    A.metrics[0] = m0;
    A.metrics[1] = m1;
    A.seq[1] = seq_issued;       // nothing is pending any more: this step is not speculative
    A.metrics[2] = m2;
    if (A.metrics_accum) {       // (no load of what was just stored: each such round trip is part of the launch's tail)
      const float a0 = A.metrics_accum[0], a1 = A.metrics_accum[1], a2 = A.metrics_accum[2], a4 = A.metrics_accum[4];
      A.metrics_accum[0] = a0 + m0;
      A.metrics_accum[1] = a1 + m1;
      A.metrics_accum[2] = a2 + m2;
      A.metrics_accum[4] = a4 + 1.0f;
    }
  }
  p2p_push(X, epoch, i, NP, g);
}

// multi-GPU: wait for every rank's gradient, add the world slots in rank order, form the clip-norm partials
__global__ void __launch_bounds__(256) k_sac_gather(P2pDev X, float *grads, int P, int Q2, float *ss_part) {
  const int NP = P + Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned epoch = X.epoch[0];
  const bool ok = p2p_wait(X, X.epoch[1]);
  float g = 0.f;
  if (i < NP) {
    g = ok ? p2p_sum(X, epoch, i) : NAN;
    grads[i] = g;
  }
  group_sumsq(g, i, P, Q2, NP, ss_part);
}

// multi-GPU, one launch: local slab reduction -> stores into every rank's region -> wait for every rank's arrivals -> sum of the
// world slots in rank order -> clip-norm partials.  All workgroups of the grid (NP/256 ~ 100) are co-resident, so waiting inside
// the producing kernel for the peers' same kernel cannot deadlock; compared with k_sac_reduce_push + k_sac_gather this removes a
// kernel boundary from every sgd_step and leaves the xGMI store latency as the only cost of the exchange.
__global__ void __launch_bounds__(256) k_sac_reduce_exchange(SacReduceArgs A, P2pDev X) {
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned epoch = X.epoch[0], want = X.epoch[1];
  const unsigned int seq_issued = (i == NP - 1) ? A.seq[0] : 0u;      // requested beside the slab sums
  float g = 0.f;
  if (i < A.P) {
    g = slab_sum<16>(A.slab_pi, A.P, A.n_tiles, i);
  } else if (i < A.P + A.Q2) {
    const int j = i - A.P;
    g = slab_sum<16>(A.slab_q, A.Q2, A.n_tiles, j);
  } else if (i == NP - 1) {
    const float ce = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 0) + slab_sum<16>(A.slab_ex, 4, A.n_tiles, 3), ac = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 1),
                al = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 2);
    const float invB = 1.0f / (float)A.B;
    g = al * invB;
    const float m0 = 0.5f * ce * (0.5f * invB), m1 = ac * invB, m2 = al * invB;
    A.metrics[0] = m0;
    A.metrics[1] = m1;
    A.seq[1] = seq_issued;       // nothing is pending any more: this step is not speculative
    A.metrics[2] = m2;
    if (A.metrics_accum) {       // (no load of what was just stored: each such round trip is part of the launch's tail)
      const float a0 = A.metrics_accum[0], a1 = A.metrics_accum[1], a2 = A.metrics_accum[2], a4 = A.metrics_accum[4];
      A.metrics_accum[0] = a0 + m0;
      A.metrics_accum[1] = a1 + m1;
      A.metrics_accum[2] = a2 + m2;
      A.metrics_accum[4] = a4 + 1.0f;
    }
  }
  p2p_push(X, epoch, i, NP, g);
  const bool ok = p2p_wait(X, want);
  float gs = 0.f;
  if (i < NP) {
    gs = ok ? p2p_sum(X, epoch, i) : NAN;
    A.grads[i] = gs;
  }
  group_sumsq(gs, i, A.P, A.Q2, NP, A.ss_part);
}

__global__ void __launch_bounds__(256) k_sac_sumsq(const float *grads, int P, int Q2, float *ss_part) {
  const int NP = P + Q2 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float g = i < NP ? grads[i] : 0.f;
  group_sumsq(g, i, P, Q2, NP, ss_part);
}

__global__ void __launch_bounds__(256) k_sac_apply(SacOptArgs A) {
  __shared__ float s_scale[3];
  __shared__ float s_corr[2];
  const int tid = threadIdx.x;
  // the element's own operands are requested first: their latency overlaps the norm reduction below
  const int NP = A.P + A.Q2 + 1;
  const int i = blockIdx.x * 256 + tid;
  const bool in = i < NP;
  const float g_in = in ? A.grads[i] : 0.f, m_in = in ? A.adam_m[i] : 0.f, v_in = in ? A.adam_v[i] : 0.f, p_in = in ? A.params[i] : 0.f;
  const float count_in = A.step_count[0];
  const bool crit = in && i >= A.P && i < A.P + A.Q2;
  const float tq_in = crit ? A.target_q[i - A.P] : 0.f;
  sac_group_norms(A, s_scale, tid);          // waves 0..2: fixed shuffle tree -> deterministic
  if (tid == 192) {
    // the Adam bias corrections are the same for every element: the fourth wave forms them (two powf, ~300 instructions)
    // beside the three norm reductions instead of every wave after the barrier
    s_corr[0] = 1.f - powf(0.9f, count_in);
    s_corr[1] = 1.f - powf(0.999f, count_in);
  }
  __syncthreads();
  if (!in) return;
  const int grp = (i < A.P) ? 0 : (i < A.P + A.Q2 ? 1 : 2);
  const float gnorm = s_scale[grp];
  float g = g_in * A.grad_scale;
  // [3P optax.clip_by_global_norm] g_norm = sqrt(sum g^2); g <- g if g_norm < max_norm else (g / g_norm) * max_norm
  if (!(gnorm < A.max_norm)) g = (g / gnorm) * A.max_norm;
  const AdamOut o = sac_adam(p_in, m_in, v_in, g, s_corr[0], s_corr[1], A.lr[grp], A.wd[grp]);   // count_in is already this step's count
  A.adam_m[i] = o.m;
  A.adam_v[i] = o.v;
  A.params[i] = o.p;
  if (grp == 1) {
    A.target_q[i - A.P] = tq_in * A.one_minus_tau + o.p * A.tau;           // sac.py:260-261 ((1 - tau) formed in double on the host)
  } else if (grp == 2) {
    A.metrics[3] = expf(o.p);                                      // 'alpha': exp(alpha_params) (sac.py:267)
    if (A.metrics_accum) A.metrics_accum[3] += A.metrics[3];
    // (this is the one thread of the launch that owns log_alpha) a step in which any group was clipped counts as a clip event
    if (!(s_scale[0] < A.max_norm) || !(s_scale[1] < A.max_norm) || !(s_scale[2] < A.max_norm)) A.seq[SAC_CTL_CLIP_EVENTS] += 1u;
  }
}

// ------------------------------------------------------------------------------------------------ host side

// ------------------------------------------------------------------------------------------------
// k_sac_reduce_apply: k_sac_reduce (+ the peer exchange when EXCHANGE) and the optimizer step in ONE launch, without a device-wide
// meeting point: the step is applied UNCLIPPED and the previous (params, m, v, target) go to the undo log; whoever reads the
// parameters next (k_sac_fwd_bwd's prologue, k_sac_finalize) sums the clip-norm partials this kernel leaves and, if a group's
// norm reached max_norm, recomputes that group's step from the undo log (sac_clip_fixup).  step_count was bumped by this step's
// fwd/bwd launch: every block reads the same final count.
template <bool EXCHANGE>
__global__ void __launch_bounds__(256) k_sac_reduce_apply(SacReduceArgs A, SacOptArgs O, P2pDev X) {
  __shared__ float s_corr[2];
  const int NP = A.P + A.Q2 + 1;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  const bool in = i < NP;
  // the element's own optimizer operands are requested first: their latency overlaps the slab sums
  const float m_in = in ? O.adam_m[i] : 0.f, v_in = in ? O.adam_v[i] : 0.f, p_in = in ? O.params[i] : 0.f;
  const bool crit = in && i >= A.P && i < A.P + A.Q2;
  const float tq_in = crit ? O.target_q[i - A.P] : 0.f;
  const float count = O.step_count[0];
  // the launch's last element (log_alpha) also keeps the running metric sums and the sequence word: everything it will
  // read-modify-write is requested HERE, beside its slab sums — one after the other behind them (a load of what it had just
  // stored among them) these round trips were the tail of the launch: 0.45 us per update
  const bool last = (i == NP - 1);
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, acc4 = 0.f;
  unsigned int seq0 = 0u;
  if (last) {
    seq0 = O.seq[0];
    if (A.metrics_accum) {
      acc0 = A.metrics_accum[0];
      acc1 = A.metrics_accum[1];
      acc2 = A.metrics_accum[2];
      acc3 = A.metrics_accum[3];
      acc4 = A.metrics_accum[4];
    }
  }
  unsigned epoch = 0, want = 0;
  if (EXCHANGE) {
    epoch = X.epoch[0];
    want = X.epoch[1];
  }
  if (tid == 0) {
    s_corr[0] = 1.f - powf(0.9f, count);
    s_corr[1] = 1.f - powf(0.999f, count);
  }
  float g = 0.f;
  if (i < A.P) {
    g = slab_sum<16>(A.slab_pi, A.P, A.n_tiles, i);
  } else if (i < A.P + A.Q2) {
    const int j = i - A.P;
    g = slab_sum<16>(A.slab_q, A.Q2, A.n_tiles, j);
  } else if (i == NP - 1) {
    const float ce = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 0) + slab_sum<16>(A.slab_ex, 4, A.n_tiles, 3), ac = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 1),
                al = slab_sum<16>(A.slab_ex, 4, A.n_tiles, 2);
    const float invB = 1.0f / (float)A.B;
    g = al * invB;
    const float m0 = 0.5f * ce * (0.5f * invB), m1 = ac * invB, m2 = al * invB;
    A.metrics[0] = m0;
    A.metrics[1] = m1;
    A.metrics[2] = m2;
    if (A.metrics_accum) {
      A.metrics_accum[0] = acc0 + m0;
      A.metrics_accum[1] = acc1 + m1;
      A.metrics_accum[2] = acc2 + m2;
      A.metrics_accum[4] = acc4 + 1.0f;
    }
    O.undo_count[0] = count;
    O.undo_count[1] = acc3;          // metrics_accum[3] before this step adds its 'alpha' (0 without running sums)
    O.seq[1] = seq0;                 // everything before this step was resolved by this step's fwd/bwd launch ...
    O.seq[0] = seq0 + 1u;            // ... and this one speculative step is pending now
  }
  if (EXCHANGE) {
    p2p_push(X, epoch, i, NP, g);
    const bool ok = p2p_wait(X, want);
    g = in ? (ok ? p2p_sum(X, epoch, i) : NAN) : 0.f;
  }
  if (in) A.grads[i] = g;
  // (slot_word was published by this step's fwd/bwd launch: stable during this launch)
  group_sumsq(g, i, A.P, A.Q2, NP, A.ss_part, reinterpret_cast<float *>(O.seq) + 2 + 3 * (O.slot_word[0] & 1u),
              (O.max_norm / O.grad_scale) * (O.max_norm / O.grad_scale) * 0.9998f / (float)O.n_parts);   // ends in a __syncthreads
  if (!in) return;
  const int grp = (i < A.P) ? 0 : (i < A.P + A.Q2 ? 1 : 2);
  float *u_p = O.undo, *u_m = O.undo + NP, *u_v = O.undo + 2 * NP, *u_tq = O.undo + 3 * NP;
  u_p[i] = p_in;
  u_m[i] = m_in;
  u_v[i] = v_in;
  const AdamOut o = sac_adam(p_in, m_in, v_in, g * O.grad_scale, s_corr[0], s_corr[1], O.lr[grp], O.wd[grp]);
  O.params[i] = o.p;
  O.adam_m[i] = o.m;
  O.adam_v[i] = o.v;
  if (grp == 1) {
    u_tq[i - A.P] = tq_in;
    O.target_q[i - A.P] = tq_in * O.one_minus_tau + o.p * O.tau;       // sac.py:260-261
  } else if (grp == 2) {
    O.undo_count[2] = s_corr[0];
    O.undo_count[3] = s_corr[1];
    const float al = expf(o.p);                                        // 'alpha' (sac.py:267); repaired by the fix-up if the group clips
    O.metrics[3] = al;
    if (O.metrics_accum) O.metrics_accum[3] = acc3 + al;
  }
}

// Resolves the clip check of the last speculative step when no further fwd/bwd launch will (end of training_step's sgd scan,
// a single sgd_step of the eager API): one workgroup.  Idempotent.
__global__ void __launch_bounds__(1024) k_sac_finalize(SacOptArgs O, unsigned long long *rng_dev, unsigned long long rng_inc) {
  __shared__ float s_gn[4];
  const int tid = threadIdx.x;
  // (mbpo_sac_finalize_advance) the end of a training step: the device RNG's step counter moves on in this launch as well
  if (rng_dev && tid == 0) rng_dev[1] += rng_inc;
  if (O.seq[0] == O.seq[1]) return;
  sac_group_norms(O, s_gn, tid);
  __syncthreads();
  const bool clip = !(s_gn[0] < O.max_norm) || !(s_gn[1] < O.max_norm) || !(s_gn[2] < O.max_norm);
  if (clip) {
    sac_clip_fixup(O, s_gn, tid, 1024);
    if (tid == 0) O.seq[SAC_CTL_CLIP_EVENTS] += 1u;
    __threadfence();
  }
  __syncthreads();
  if (tid == 0) O.seq[1] = O.seq[0];
}

struct SacPlan {
  MlpDev pi, q, qt;
  int P, Q, NP, n_tiles, H, LH, n_red;
  size_t lds;
  int ld_x, ld_xu, ld_h, ld_y;
  // workspace offsets (floats)
  long long off_slab_pi, off_slab_q, off_slab_ex, off_ss, off_seq, off_undo, off_layered, total;
  // hidden layers outside the fused kernel's range (one width in {64,128} that fits a tile's LDS): the forward/backward half runs
  // layer by layer (sac_layered.hip) and leaves ONE tile of slabs; everything behind it is shared
  bool layered;
};

static int same_hidden(const int *dims, int n_layers) {
  if (n_layers < 2) return -1;
  for (int l = 2; l < n_layers; ++l)
    if (dims[l] != dims[1]) return -1;
  return dims[1];
}

static int sac_plan(const mbpo_sac_desc *d, SacPlan *pl, bool need_ptrs) {
  MBPO_REQUIRE(d, MBPO_ERR_ARG, "sac: null descriptor");
  MBPO_REQUIRE(d->x_dim > 0 && d->u_dim > 0 && d->batch_size > 0, MBPO_ERR_ARG, "sac: x_dim/u_dim/batch_size must be positive");
  MBPO_REQUIRE(d->row_len == 2 * d->x_dim + d->u_dim + 3, MBPO_ERR_ARG, "sac: row_len %d != 2x+u+3", d->row_len);
  MBPO_REQUIRE(d->policy_layers >= 2 && d->policy_layers <= MBPO_MAX_LAYERS && d->q_layers >= 2 && d->q_layers <= MBPO_MAX_LAYERS,
               MBPO_ERR_ARG, "sac: networks need at least one hidden layer and at most %d Dense layers", MBPO_MAX_LAYERS);
  MBPO_REQUIRE(d->policy_dims[0] == d->x_dim && d->policy_dims[d->policy_layers] == 2 * d->u_dim, MBPO_ERR_ARG,
               "sac: policy must map [x_dim] -> [2*u_dim]");
  MBPO_REQUIRE(d->q_dims[0] == d->x_dim + d->u_dim && d->q_dims[d->q_layers] == 1, MBPO_ERR_ARG,
               "sac: critic must map [x_dim+u_dim] -> [1]");
  const int Hp = same_hidden(d->policy_dims, d->policy_layers), Hq = same_hidden(d->q_dims, d->q_layers);
  static const int layered_env = getenv("MBPO_SAC_LAYERED") ? atoi(getenv("MBPO_SAC_LAYERED")) : 0;    // 1: force the layered path (tests)
  pl->layered = layered_env != 0 || !(Hp == Hq && (Hp == 64 || Hp == 128));
  mbpo_mlp_desc md;
  md.net_stride = 0;
  // policy
  md.params = d->params ? d->params : (const float *)16;  // placeholder for size queries
  md.n_nets = 1;
  md.n_layers = d->policy_layers;
  for (int l = 0; l <= d->policy_layers; ++l) md.dims[l] = d->policy_dims[l];
  md.activation = d->policy_activation;
  int rc = mbpo_make_mlp_dev(&md, &pl->pi, "sac.policy");
  if (rc != MBPO_OK) return rc;
  pl->P = pl->pi.n_params;
  // critics
  md.n_layers = d->q_layers;
  for (int l = 0; l <= d->q_layers; ++l) md.dims[l] = d->q_dims[l];
  md.activation = d->q_activation;
  md.n_nets = 1;
  rc = mbpo_make_mlp_dev(&md, &pl->q, "sac.q");
  if (rc != MBPO_OK) return rc;
  pl->Q = pl->q.n_params;
  pl->q.n_nets = 2;
  pl->q.net_stride = pl->Q;
  pl->q.params = d->params ? d->params + pl->P : nullptr;
  pl->qt = pl->q;
  pl->qt.params = d->target_q;
  pl->NP = pl->P + 2 * pl->Q + 1;
  pl->H = Hp;
  const int lhp = d->policy_layers - 1, lhq = d->q_layers - 1;
  pl->LH = lhp > lhq ? lhp : lhq;
  pl->n_tiles = (d->batch_size + 15) / 16;
  pl->n_red = (pl->NP + 255) / 256;
  pl->off_layered = 0;
  auto up4 = [](int v) { return (v + 3) & ~3; };
  pl->ld_x = up4(d->x_dim) + 4;
  pl->ld_xu = up4(d->x_dim + d->u_dim) + 4;
  pl->ld_h = Hp + 4;
  pl->ld_y = up4(2 * d->u_dim) + 4;
  const int U = d->u_dim;
  size_t f = 16ull * up4(d->row_len) + 2ull * 16 * pl->ld_x + 2ull * 16 * pl->ld_xu + 4ull * 16 * pl->ld_h +
             (size_t)pl->LH * 4 * 16 * pl->ld_h + 5ull * 16 * pl->ld_y + 2ull * 16 * pl->ld_xu + 6ull * up4(16 * U) + 64 + 4;
  pl->lds = f * sizeof(float);
  if (!pl->layered && pl->lds > 160 * 1024) pl->layered = true;      // more stored activations than a tile's LDS holds
  if (pl->layered) pl->n_tiles = 1;
  pl->off_slab_pi = 0;
  pl->off_slab_q = pl->off_slab_pi + (long long)pl->n_tiles * pl->P;
  pl->off_slab_ex = pl->off_slab_q + (long long)pl->n_tiles * 2 * pl->Q;
  pl->off_ss = pl->off_slab_ex + (long long)pl->n_tiles * 4;
  pl->off_seq = (pl->off_ss + (long long)pl->n_red * 3 + 3) & ~3LL;   // 8 words {seq issued, seq resolved, 2 x 3 quick sums} (16-byte aligned),
                                                                       // then slot word + undo count, accum, 2 bias corrections (zero at start)
  pl->off_undo = pl->off_seq + 16;                                     // undo log of the speculative optimizer step
  pl->total = pl->off_undo + 3LL * pl->NP + 2LL * pl->Q;
  if (pl->layered) {
    pl->off_layered = (pl->total + 3) & ~3LL;
    pl->total = pl->off_layered + sac_layered_floats(d, pl->pi, pl->q);
  }
  if (need_ptrs) {
    MBPO_REQUIRE(d->params && d->target_q && d->adam_m && d->adam_v && d->step_count && d->grads && d->workspace && d->metrics,
                 MBPO_ERR_ARG, "sac: null state pointer");
  }
  return MBPO_OK;
}

extern "C" int64_t mbpo_sac_workspace_floats(const mbpo_sac_desc *d) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, false);
  if (rc != MBPO_OK) return rc;
  return pl.total;
}

extern "C" int64_t mbpo_sac_control_offset(const mbpo_sac_desc *d) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, false);
  if (rc != MBPO_OK) return rc;
  return pl.off_seq;
}

constexpr int SP64 = 4;   // waves per chain at hidden width 64

// LDS offsets (floats) of the tiles the chains use; must mirror the carve at the top of k_sac_fwd_bwd
static void sac_chain_table(const SacPlan &pl, int D, SacArgs *A, bool split, bool jvp) {
  const int ld_x = pl.ld_x, ld_xu = pl.ld_xu, ld_h = pl.ld_h, ld_y = pl.ld_y, LH = pl.LH;
  const int T = 16 * ld_h;
  const int o_row = 0;
  const int o_sn = o_row + 16 * ((D + 3) & ~3), o_sn2 = o_sn + 16 * ld_x, o_qin = o_sn2 + 16 * ld_x, o_qin2 = o_qin + 16 * ld_xu;
  const int o_pp = o_qin2 + 16 * ld_xu, o_st0 = o_pp + 4 * T, o_y = o_st0 + 4 * LH * T, o_dy = o_y + 3 * 16 * ld_y, o_dx = o_dy + 2 * 16 * ld_y;
  const int o_st1 = o_st0 + LH * T, o_st2 = o_st0 + 2 * LH * T, o_st3 = o_st0 + 3 * LH * T;
  const int P = pl.P, Q = pl.Q;   // params = [policy | critic0 | critic1 | log_alpha]; target_q = [critic0 | critic1]
  for (int role = 0; role < 3; ++role)
    for (int ph = 0; ph < 5; ++ph)
      for (int c = 0; c < 4; ++c) {
        SacChainDesc d;
        memset(&d, 0, sizeof(d));
        d.mode = CH_IDLE;
        d.x = d.pp0 = d.pp1 = d.zb = d.hb = d.y = d.dx = d.tp0 = d.tp1 = -1;
        d.ldx = ld_x;
        const int net = c & 1;
        if (split && role != 1) {              // critic role of ONE critic kq: never more than two chains
          const int kq = role >> 1;
          if (ph == 0 && c < 2) {              // pi(s') || Qk(s,a)
            d.mode = CH_FWD;
            if (c == 0) {
              d.netid = 0; d.param_off = 0; d.x = o_sn2; d.ldx = ld_x; d.pp0 = o_pp; d.pp1 = o_pp + T; d.y = o_y;
            } else {
              d.netid = 1; d.param_off = P + kq * Q; d.x = o_qin; d.ldx = ld_xu;
              d.zb = kq == 0 ? o_st0 : o_st2; d.hb = kq == 0 ? o_st1 : o_st3; d.y = o_y + (kq + 1) * 16 * ld_y;
            }
          } else if (ph == 1 && c < 2) {       // BOTH target critics on (s', a')
            d.mode = CH_FWD;
            d.netid = 1; d.base_sel = 1; d.param_off = c * Q; d.x = o_qin2; d.ldx = ld_xu;
            d.pp0 = o_pp + 2 * c * T; d.pp1 = d.pp0 + T; d.y = o_y + (c + 1) * 16 * ld_y;
          } else if (ph == 2 && c < 2) {       // critic kq backward: dgrad beside wgrad
            d.mode = c == 0 ? CH_DGRAD : CH_WGRAD;
            d.netid = 1; d.param_off = P + kq * Q; d.x = o_qin; d.ldx = ld_xu;
            d.pp0 = o_pp + 2 * kq * T; d.pp1 = d.pp0 + T; d.zb = kq ? o_st2 : o_st0; d.hb = kq ? o_st3 : o_st1;
            d.y = o_dy + kq * 16 * ld_y;
            d.slab_sel = 2; d.slab_off = kq * Q;
          }
        } else if (role == 2) {
          // (table role 2 is not used without `split`)
        } else if (role == 0) {                // critic role
          if (ph == 0 && c < 3) {              // pi(s') || Q1(s,a) || Q2(s,a)
            d.mode = CH_FWD;
            if (c == 0) {
              d.netid = 0; d.param_off = 0; d.x = o_sn2; d.ldx = ld_x; d.pp0 = o_pp; d.pp1 = o_pp + T; d.y = o_y;
            } else {
              d.netid = 1; d.param_off = P + (c - 1) * Q; d.x = o_qin; d.ldx = ld_xu;
              d.zb = c == 1 ? o_st0 : o_st2; d.hb = c == 1 ? o_st1 : o_st3; d.y = o_y + c * 16 * ld_y;
            }
          } else if (ph == 1 && c < 2) {       // target critics on (s', a')
            d.mode = CH_FWD;
            d.netid = 1; d.base_sel = 1; d.param_off = c * Q; d.x = o_qin2; d.ldx = ld_xu;
            d.pp0 = o_pp + 2 * c * T; d.pp1 = d.pp0 + T; d.y = o_y + (c + 1) * 16 * ld_y;
          } else if (ph == 2) {                // critics backward: dgrad (c < 2) beside wgrad
            d.mode = c < 2 ? CH_DGRAD : CH_WGRAD;
            d.netid = 1; d.param_off = P + net * Q; d.x = o_qin; d.ldx = ld_xu;
            d.pp0 = o_pp + 2 * net * T; d.pp1 = d.pp0 + T; d.zb = net ? o_st2 : o_st0; d.hb = net ? o_st3 : o_st1;
            d.y = o_dy + net * 16 * ld_y;
            d.slab_sel = 2; d.slab_off = net * Q;
          }
        } else {                               // actor + alpha role
          if (ph == 0 && c == 0) {
            d.mode = CH_FWD;
            d.netid = 0; d.param_off = 0; d.x = o_sn; d.ldx = ld_x; d.zb = o_st0; d.hb = o_st1; d.y = o_y;
          } else if (ph == 1 && c < 2) {
            d.mode = CH_FWD;
            d.netid = 1; d.param_off = P + c * Q; d.x = o_qin; d.ldx = ld_xu;
            d.pp0 = o_pp + 2 * c * T; d.pp1 = d.pp0 + T; d.zb = c == 0 ? o_st2 : o_st3; d.y = o_y + (c + 1) * 16 * ld_y;
            if (jvp) {     // no z store (no backward through the critics): its tiles carry the tangent, whose output goes to s_dx[c]
              d.zb = -1;
              d.tp0 = c == 0 ? o_st2 : o_st3; d.tp1 = d.tp0 + T; d.dx = o_dx + c * 16 * ld_xu;
            }
          } else if (jvp && ph == 2 && c < 2) {      // the policy's backward pass follows at once
            d.mode = c == 0 ? CH_DGRAD : CH_WGRAD;
            d.netid = 0; d.param_off = 0; d.x = o_sn; d.ldx = ld_x; d.pp0 = o_pp; d.pp1 = o_pp + T; d.zb = o_st0; d.hb = o_st1;
            d.y = o_dy; d.slab_sel = 1; d.slab_off = 0;
          } else if (jvp) {
            // (idle)
          } else if (ph == 2 && c < 2) {
            d.mode = CH_DGRAD;
            d.netid = 1; d.param_off = P + net * Q; d.pp0 = o_pp + 2 * net * T; d.pp1 = d.pp0 + T; d.zb = net ? o_st3 : o_st2;
            d.y = o_dy + net * 16 * ld_y; d.dx = o_dx + net * 16 * ld_xu;
          } else if (ph == 3 && c < 2) {
            d.mode = c == 0 ? CH_DGRAD : CH_WGRAD;
            d.netid = 0; d.param_off = 0; d.x = o_sn; d.ldx = ld_x; d.pp0 = o_pp; d.pp1 = o_pp + T; d.zb = o_st0; d.hb = o_st1;
            d.y = o_dy; d.slab_sel = 1; d.slab_off = 0;
          }
        }
        A->tab[role][ph][c] = d;
      }
}

static void sac_fill_opt(const mbpo_sac_desc *d, const SacPlan &pl, SacOptArgs *A) {
  A->params = d->params; A->target_q = d->target_q; A->adam_m = d->adam_m; A->adam_v = d->adam_v; A->grads = d->grads;
  A->metrics = d->metrics; A->metrics_accum = d->metrics_accum; A->step_count = d->step_count; A->ss_part = d->workspace + pl.off_ss;
  A->undo = d->workspace + pl.off_undo;
  A->seq = reinterpret_cast<unsigned int *>(d->workspace + pl.off_seq);
  A->slot_word = reinterpret_cast<unsigned int *>(d->workspace + pl.off_seq + 8);
  A->undo_count = d->workspace + pl.off_seq + 9;
  A->n_parts = pl.n_red; A->P = pl.P; A->Q2 = 2 * pl.Q;
  A->lr[0] = d->lr_policy; A->lr[1] = d->lr_q; A->lr[2] = d->lr_alpha;
  A->wd[0] = d->wd_policy; A->wd[1] = d->wd_q; A->wd[2] = d->wd_alpha;
  A->max_norm = d->max_grad_norm; A->tau = d->tau; A->one_minus_tau = (float)(1.0 - (double)d->tau); A->grad_scale = d->grad_scale;
}

// phase_mask: bit0 = k_sac_fwd_bwd, bit1 = the slab reduction.  apply_in_reduce: the reduction launch is k_sac_reduce_apply.
static int sac_grads_impl(const mbpo_sac_desc *d, int phase_mask, void *stream, const mbpo_p2p_desc *xd = nullptr,
                          bool exchange_in_reduce = false, bool apply_in_reduce = false) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(d->batch, MBPO_ERR_ARG, "sac_grads: null batch");
  MBPO_REQUIRE((d->norm_mean == nullptr) == (d->norm_std == nullptr), MBPO_ERR_ARG, "sac_grads: norm_mean/norm_std mismatch");
  SacArgs A;
  A.pi = pl.pi; A.q = pl.q; A.qt = pl.qt;
  A.sh_pi = NetShape{pl.pi.dims[0], pl.pi.n_layers, pl.pi.dims[pl.pi.n_layers], pl.pi.act};
  A.sh_q = NetShape{pl.q.dims[0], pl.q.n_layers, pl.q.dims[pl.q.n_layers], pl.q.act};
  // three workgroups per tile (one critic each): at hidden width 128 a phase is MFMA-bound on its CU, at 64 the two critic
  // workgroups finish before the actor role does and 8-wave workgroups have cheaper barriers (26.9 -> 24.0 us with the forward-mode
  // actor role).  The WIDE kernels keep two workgroups per tile.  MBPO_SAC_SPLIT=0/1 overrides (measurement).
  static const int split_env = getenv("MBPO_SAC_SPLIT") ? atoi(getenv("MBPO_SAC_SPLIT")) : -1;
  const bool wide_any = net_is_wide(A.sh_pi) || net_is_wide(A.sh_q);
  const bool split = split_env >= 0 ? split_env != 0 : !wide_any;
  A.split = split ? 1 : 0;
  const int wg_per_tile = split ? 3 : 2;
  // forward-mode dQ/da in the actor role: one action dimension (one tangent), register-image kernels with one-tile network ends
  static const int jvp_env = getenv("MBPO_SAC_JVP") ? atoi(getenv("MBPO_SAC_JVP")) : -1;
  const bool reg_path = (pl.H == 64 || split) && !(net_is_wide(A.sh_pi) || net_is_wide(A.sh_q)) && pl.LH >= 2;
  const bool jvp = d->u_dim == 1 && reg_path && (jvp_env < 0 || jvp_env != 0);
  A.jvp = jvp ? 1 : 0;
  // thin layers: the three-workgroup launch at H = 64 with forward-mode actor (no input-gradient chains), small input / output
  // layers and at least one H x H layer in each network
  {
    const char *e = getenv("MBPO_SAC_THIN");
    const bool want = e ? atoi(e) != 0 : true;
    A.thin = (want && split && jvp && pl.H == 64 && A.sh_pi.K_in <= THIN_KMAX && A.sh_q.K_in <= THIN_KMAX && A.sh_pi.N_out <= THIN_NMAX &&
              A.sh_q.N_out <= THIN_NMAX && A.sh_pi.L >= 3 && A.sh_q.L >= 3) ? 1 : 0;
  }
  if (!pl.layered) sac_chain_table(pl, d->row_len, &A, split, jvp);
  A.X = d->x_dim; A.U = d->u_dim; A.B = d->batch_size; A.D = d->row_len;
  A.batch = d->batch; A.norm_mean = d->norm_mean; A.norm_std = d->norm_std;
  A.log_alpha = d->params + pl.NP - 1;
  A.noise_alpha = d->noise_alpha; A.noise_critic = d->noise_critic; A.noise_actor = d->noise_actor;
  A.seed = d->seed; A.offset = d->offset; A.rng_dev = (const unsigned long long *)d->rng_dev;
  sac_fill_opt(d, pl, &A.opt);
  A.step_count_rw = d->step_count;
  A.stamps = g_sac_stamps;
  P2pDev X;
  memset(&X, 0, sizeof(X));
  A.p2p_epoch = nullptr;
  A.p2p_blocks = (unsigned)pl.n_red;
  if (xd) {
    rc = mbpo_p2p_make_dev(xd, &X);
    if (rc != MBPO_OK) return rc;
    MBPO_REQUIRE(xd->n_max >= pl.NP, MBPO_ERR_ARG, "sac_grads_p2p: exchange regions hold %lld floats, the gradient has %d",
                 (long long)xd->n_max, pl.NP);
    A.p2p_epoch = X.epoch;
  }
  A.discounting = d->discounting; A.reward_scaling = d->reward_scaling; A.target_entropy = d->target_entropy;
  A.neq = d->non_equidistant_time; A.neq_cd = d->continuous_discounting; A.neq_tl = d->min_time_between_switches;
  A.neq_tu = d->max_time_between_switches; A.neq_dt = d->env_dt;
  MBPO_REQUIRE(!A.neq || A.neq_dt > 0.f, MBPO_ERR_ARG, "sac: non_equidistant_time needs env_dt > 0");
  A.slab_pi = d->workspace + pl.off_slab_pi; A.slab_q = d->workspace + pl.off_slab_q; A.slab_ex = d->workspace + pl.off_slab_ex;
  A.ld_x = pl.ld_x; A.ld_xu = pl.ld_xu; A.ld_h = pl.ld_h; A.ld_y = pl.ld_y; A.LH = pl.LH;
  hipStream_t st = (hipStream_t)stream;
  // MBPO_SAC_LEAN=0 keeps the generic kernel on the benchmark networks too (A/B runs, the bit-identity test)
  static const int lean_env = getenv("MBPO_SAC_LEAN") ? atoi(getenv("MBPO_SAC_LEAN")) : 1;
  const bool lean = (g_sac_lean >= 0 ? g_sac_lean : lean_env) != 0 && !pl.layered && A.thin &&
                    sac_lean_supports(d->x_dim, d->u_dim, d->policy_dims, d->policy_layers, d->policy_activation, d->q_dims, d->q_layers,
                                      d->q_activation);
  if (pl.layered) {
    if (phase_mask & 1) {
      const SacLayeredBegin bg = {d->step_count, A.opt.seq, A.opt.slot_word, A.p2p_epoch, A.p2p_blocks};
      rc = sac_layered_fwd_bwd(d, pl.pi, pl.q, pl.qt, d->workspace + pl.off_layered, A.slab_pi, A.slab_q, A.slab_ex, bg, st);
      if (rc != MBPO_OK) return rc;
    }
  } else if ((phase_mask & 1) && lean) {
    // the benchmark networks: every shape a compile-time constant (sac_lean.hip), same slabs bit for bit
    SacLeanArgs L;
    L.params = d->params; L.target_q = d->target_q; L.batch = d->batch; L.norm_mean = d->norm_mean; L.norm_std = d->norm_std;
    L.noise_alpha = d->noise_alpha; L.noise_critic = d->noise_critic; L.noise_actor = d->noise_actor;
    L.rng_dev = A.rng_dev; L.seed = A.seed; L.offset = A.offset;
    L.slab_pi = A.slab_pi; L.slab_q = A.slab_q; L.slab_ex = A.slab_ex;
    L.step_count_rw = d->step_count; L.p2p_epoch = A.p2p_epoch; L.p2p_blocks = A.p2p_blocks; L.B = d->batch_size;
    L.discounting = A.discounting; L.reward_scaling = A.reward_scaling; L.target_entropy = A.target_entropy;
    L.neq = A.neq; L.neq_cd = A.neq_cd; L.neq_tl = A.neq_tl; L.neq_tu = A.neq_tu; L.neq_dt = A.neq_dt;
    L.stamps = g_sac_stamps;
    L.pad0 = 0;
    L.opt = A.opt;
    rc = sac_lean_launch(L, d->x_dim, pl.n_tiles, stream);
    if (rc != MBPO_OK) return rc;
  } else if (phase_mask & 1) {
    if (pl.H == 64) {
      if (net_is_wide(A.sh_pi) || net_is_wide(A.sh_q)) {
        rc = mbpo_ensure_lds<k_sac_fwd_bwd<64, SP64, true>>(pl.lds, "sac_grads");
        if (rc != MBPO_OK) return rc;
        hipLaunchKernelGGL((k_sac_fwd_bwd<64, SP64, true>), dim3(wg_per_tile * pl.n_tiles), dim3(256 * SP64), pl.lds, st, A);
      } else if (split) {
        if (A.thin) {
          rc = mbpo_ensure_lds<k_sac_fwd_bwd<64, SP64, false, 2, true>>(pl.lds, "sac_grads");
          if (rc != MBPO_OK) return rc;
          hipLaunchKernelGGL((k_sac_fwd_bwd<64, SP64, false, 2, true>), dim3(wg_per_tile * pl.n_tiles), dim3(128 * SP64), pl.lds, st, A);
        } else {
          rc = mbpo_ensure_lds<k_sac_fwd_bwd<64, SP64, false, 2>>(pl.lds, "sac_grads");
          if (rc != MBPO_OK) return rc;
          hipLaunchKernelGGL((k_sac_fwd_bwd<64, SP64, false, 2>), dim3(wg_per_tile * pl.n_tiles), dim3(128 * SP64), pl.lds, st, A);
        }
      } else {
        rc = mbpo_ensure_lds<k_sac_fwd_bwd<64, SP64, false>>(pl.lds, "sac_grads");
        if (rc != MBPO_OK) return rc;
        hipLaunchKernelGGL((k_sac_fwd_bwd<64, SP64, false>), dim3(wg_per_tile * pl.n_tiles), dim3(256 * SP64), pl.lds, st, A);
      }
    } else if (split && !(net_is_wide(A.sh_pi) || net_is_wide(A.sh_q))) {
      rc = mbpo_ensure_lds<k_sac_fwd_bwd<128, 4, false, 2>>(pl.lds, "sac_grads");
      if (rc != MBPO_OK) return rc;
      hipLaunchKernelGGL((k_sac_fwd_bwd<128, 4, false, 2>), dim3(wg_per_tile * pl.n_tiles), dim3(512), pl.lds, st, A);
    } else {
      rc = mbpo_ensure_lds<k_sac_fwd_bwd<128, 2, false>>(pl.lds, "sac_grads");
      if (rc != MBPO_OK) return rc;
      hipLaunchKernelGGL((k_sac_fwd_bwd<128, 2, false>), dim3(wg_per_tile * pl.n_tiles), dim3(512), pl.lds, st, A);
    }
  }
  if (!(phase_mask & 2)) {
    MBPO_CHECK_LAUNCH("sac_grads");
    return MBPO_OK;
  }
  SacReduceArgs R;
  R.slab_pi = A.slab_pi; R.slab_q = A.slab_q; R.slab_ex = A.slab_ex;
  R.n_tiles = pl.n_tiles; R.P = pl.P; R.Q2 = 2 * pl.Q; R.B = d->batch_size;
  R.grads = d->grads; R.metrics = d->metrics; R.metrics_accum = d->metrics_accum; R.ss_part = d->workspace + pl.off_ss; R.step_count = d->step_count;
  R.seq = A.opt.seq;
  if (apply_in_reduce && pl.layered) {
    // no speculative step on the layered path (two launches out of ~50 is nothing to win): the reduction, then the clipped
    // optimizer step — what mbpo_sac_step is defined to equal bit for bit
    if (xd) hipLaunchKernelGGL(k_sac_reduce_exchange, dim3(pl.n_red), dim3(256), 0, st, R, X);
    else hipLaunchKernelGGL(k_sac_reduce, dim3(pl.n_red), dim3(256), 0, st, R);
    hipLaunchKernelGGL(k_sac_apply, dim3(pl.n_red), dim3(256), 0, st, A.opt);
  } else if (apply_in_reduce) {
    if (xd) hipLaunchKernelGGL(k_sac_reduce_apply<true>, dim3(pl.n_red), dim3(256), 0, st, R, A.opt, X);
    else hipLaunchKernelGGL(k_sac_reduce_apply<false>, dim3(pl.n_red), dim3(256), 0, st, R, A.opt, X);
  } else if (xd && exchange_in_reduce) hipLaunchKernelGGL(k_sac_reduce_exchange, dim3(pl.n_red), dim3(256), 0, st, R, X);
  else if (xd) hipLaunchKernelGGL(k_sac_reduce_push, dim3(pl.n_red), dim3(256), 0, st, R, X);
  else hipLaunchKernelGGL(k_sac_reduce, dim3(pl.n_red), dim3(256), 0, st, R);
  MBPO_CHECK_LAUNCH("sac_grads");
  return MBPO_OK;
}

extern "C" int mbpo_sac_grads_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream) {
  MBPO_REQUIRE(x, MBPO_ERR_ARG, "sac_grads_p2p: null exchange descriptor");
  return sac_grads_impl(d, 3, stream, x);
}

extern "C" int mbpo_sac_grads_exchange_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream) {
  MBPO_REQUIRE(x, MBPO_ERR_ARG, "sac_grads_exchange_p2p: null exchange descriptor");
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  // every workgroup of the reduction waits inside the kernel: they must all be resident at once
  MBPO_REQUIRE(pl.n_red <= 1024, MBPO_ERR_UNSUPPORTED, "sac_grads_exchange_p2p: %d workgroups cannot be assumed co-resident; use "
               "mbpo_sac_grads_p2p + mbpo_sac_gather_p2p", pl.n_red);
  return sac_grads_impl(d, 3, stream, x, true);
}

extern "C" int mbpo_sac_gather_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  P2pDev X;
  rc = mbpo_p2p_make_dev(x, &X);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(x->n_max >= pl.NP, MBPO_ERR_ARG, "sac_gather_p2p: exchange regions too small");
  hipLaunchKernelGGL(k_sac_gather, dim3(pl.n_red), dim3(256), 0, (hipStream_t)stream, X, d->grads, pl.P, 2 * pl.Q, d->workspace + pl.off_ss);
  MBPO_CHECK_LAUNCH("sac_gather_p2p");
  return MBPO_OK;
}

extern "C" int mbpo_sac_grads(const mbpo_sac_desc *d, void *stream) { return sac_grads_impl(d, 3, stream); }

extern "C" int mbpo_sac_grads_phase(const mbpo_sac_desc *d, int32_t phase_mask, void *stream) {
  MBPO_REQUIRE(phase_mask >= 1 && phase_mask <= 3, MBPO_ERR_ARG, "sac_grads_phase: phase_mask must be 1, 2 or 3");
  return sac_grads_impl(d, phase_mask, stream);
}

extern "C" int mbpo_sac_grad_norms(const mbpo_sac_desc *d, void *stream) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  hipLaunchKernelGGL(k_sac_sumsq, dim3(pl.n_red), dim3(256), 0, (hipStream_t)stream, (const float *)d->grads, pl.P, 2 * pl.Q,
                     d->workspace + pl.off_ss);
  MBPO_CHECK_LAUNCH("sac_grad_norms");
  return MBPO_OK;
}

extern "C" int mbpo_sac_apply(const mbpo_sac_desc *d, void *stream) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  SacOptArgs A;
  sac_fill_opt(d, pl, &A);
  hipLaunchKernelGGL(k_sac_apply, dim3(pl.n_red), dim3(256), 0, (hipStream_t)stream, A);
  MBPO_CHECK_LAUNCH("sac_apply");
  return MBPO_OK;
}

// One sgd_step in TWO launches: k_sac_fwd_bwd, then k_sac_reduce_apply (slab reduction + [peer exchange +] unclipped optimizer
// step + undo log).  The clip check is resolved by the next mbpo_sac_step / mbpo_sac_grads launch or by mbpo_sac_finalize.
extern "C" int mbpo_sac_step(const mbpo_sac_desc *d, void *stream) { return sac_grads_impl(d, 3, stream, nullptr, false, true); }

extern "C" int mbpo_sac_step_p2p(const mbpo_sac_desc *d, const mbpo_p2p_desc *x, void *stream) {
  MBPO_REQUIRE(x, MBPO_ERR_ARG, "sac_step_p2p: null exchange descriptor");
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  // every workgroup of the reduction waits inside the kernel for the peers' same kernel: they must all be resident at once
  MBPO_REQUIRE(pl.n_red <= 1024, MBPO_ERR_UNSUPPORTED, "sac_step_p2p: %d workgroups cannot be assumed co-resident; use "
               "mbpo_sac_grads_p2p + mbpo_sac_gather_p2p + mbpo_sac_apply", pl.n_red);
  return sac_grads_impl(d, 3, stream, x, true, true);
}

extern "C" int mbpo_sac_finalize(const mbpo_sac_desc *d, void *stream) {
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  SacOptArgs A;
  sac_fill_opt(d, pl, &A);
  hipLaunchKernelGGL(k_sac_finalize, dim3(1), dim3(1024), 0, (hipStream_t)stream, A, (unsigned long long *)nullptr, 0ull);
  MBPO_CHECK_LAUNCH("sac_finalize");
  return MBPO_OK;
}

extern "C" int mbpo_sac_finalize_advance(const mbpo_sac_desc *d, uint64_t *rng_dev, uint64_t inc, void *stream) {
  MBPO_REQUIRE(rng_dev, MBPO_ERR_ARG, "sac_finalize_advance: null rng_dev");
  SacPlan pl;
  int rc = sac_plan(d, &pl, true);
  if (rc != MBPO_OK) return rc;
  SacOptArgs A;
  sac_fill_opt(d, pl, &A);
  hipLaunchKernelGGL(k_sac_finalize, dim3(1), dim3(1024), 0, (hipStream_t)stream, A, (unsigned long long *)rng_dev, (unsigned long long)inc);
  MBPO_CHECK_LAUNCH("sac_finalize_advance");
  return MBPO_OK;
}

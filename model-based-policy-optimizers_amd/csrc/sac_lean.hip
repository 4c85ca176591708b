// sac_lean.hip — k_sac_lean<X>: the SAC forward/backward launch (S3-S6 of sgd_step, sac/sac.py:227-281, sac/losses.py:61-125)
// specialised for the benchmark networks: policy X -> 64 -> 64 -> 64 -> 2, critics X+1 -> 64 -> 64 -> 64 -> 1, swish, u = 1, X = 2 .. 6.
//
// Why a second kernel (round 4): the generic k_sac_fwd_bwd (sac.hip) serves every shape from kernel-argument tables — chain
// descriptors, NetShape scalars, one runner per chain kind — and sits at 104 SGPRs with 219 spilled to VGPR lanes, 3.5 KB of
// kernel arguments that need a warm-up wave, ~1.4 k cycles of chain set-up in front of every phase (profiles/r03_phase_stamps.txt).
// Here every shape, LDS offset, role and chain layout is a compile-time constant: the role paths are straight-line code, the
// kernel arguments are ~50 dwords, nothing is decoded at run time.
//
// What is kept bit for bit: every number.  Each dot product is formed by the same v_mfma_f32_16x16x4_f32 sequence over the same
// k groups in the same order as the generic kernel's (chain_run.hpp), the thin layers by the same FMA chains, the elementwise
// sections by the same expressions — so the per-tile gradient slabs, and with them everything behind this launch, are identical
// to the generic kernel's (tests/test_gpu_sac_lean.py compares the slabs with torch.equal).
//
// What is different:
//  * MFMA operands SWAPPED (A = weights, B = activations): D^T comes out, so a lane holds FOUR CONSECUTIVE COLUMNS of ONE row
//    instead of one column of four rows — every activation / delta store is one ds_write_b128 (was 4 x b32), every pre-activation
//    read of the backward one ds_read_b128, every weight-gradient store one global_store_dwordx4 (was 4 dwords: 16 + 1 store
//    instructions per wave and layer became 4 + 1).  a*b = b*a and the k order is unchanged, so the sums are the same bits.
//  * The weights of a whole PHASE (thin layer, two hidden layers, output layer) are requested at once, one phase ahead, into
//    registers that are simply named per layer: no register-set swapping, no loop unrolled by two, no descriptor fetch.
//  * The wave that forms a network's output layer keeps it in registers and runs the elementwise section itself (NormalTanh
//    sample / value hand-off) — the output image maps matrix row i to column i & 3, so all four lane groups hold (loc, raw) of
//    their row and the actor role's two samples run side by side in one pass.  One barrier and one LDS round trip less per section.
//  * Tile load: one coalesced dword per thread (the tile's 16 rows are contiguous), classified by column.
//
// Roles as in the generic kernel's `split` launch: 3 workgroups per 16-sample tile — critic 0, actor(+alpha), critic 1; 8 waves =
// 2 chains x 4 waves; forward-mode dQ/da in the actor role (the tangent rides along the critics' forward pass).
//
// Measured and NOT kept (round 4, scripts/lean_dev.py, same box): the hidden-layer images staged through LDS instead of requested by
// every chain wave as strided dwords — (a) by the two idle auxiliary waves with global_load_lds_dwordx4 (33 x 1 KB per network, raw
// barriers around the copies in flight): 13.6 us per launch against 11.3 — the LDS-DMA copies of 66 KB take > 4 k cycles on a CU and
// the barrier behind them waits; (b) by all eleven waves, six coalesced 16-byte requests per thread, stored to LDS behind the first
// barrier: 12.0 us — the thin-layer step then ends 9 k cycles after the kernel's start.  Caveat for both: the critics' blocks start at
// odd float offsets of the flat layout (P = 8770, Q = 8769 at x = 4), so half of the 16-byte requests were 4- or 8-byte aligned; a
// dword-granular copy (4 x the instructions) was not tried.  The kernel costs the same back to back on L2-warm parameters as behind
// the optimizer launch (11.3 us both), so the prologue is not a cold-miss problem either; the per-wave strided requests stay.
// What did pay (same session): WHERE the requests are issued.  A strided 64-lane dword request costs the issuing wave ~45 cycles
// wherever it sits (F1's two images moved from the hidden steps' MFMA shadows into the thin step: that step +1.6 k cycles, the hidden
// steps -1.0 k), and a wave's first burst delays the tile load of every wave behind the same address path.  The second hidden
// layer's images are now requested in the thin-layer step instead of the prologue, and the actor role's idle chain issues all of its
// requests behind the first barrier: 11.26 -> 10.8 us per launch; moving the output-layer images as well cost 0.3 us (the output
// wave is the critical one).  A transposed copy of the hidden matrices (4 x fewer, 16-byte requests per lane) was timed with the request
// pattern alone before building it: 12.7 us — a lane's 64 contiguous bytes sit 256 bytes from its neighbour's, so one request touches
// 64 separate 64-byte segments where the strided dword request touches four.  Image-major copies were timed the same way: 16 fully
// contiguous dword requests per image 10.8 us (no change), four contiguous 1-KB requests per image 11.2 us (worse).  The request
// pattern is not the lever; a reordered copy of the weights is not worth keeping.
#include "common.hpp"
#include "chain_run.hpp"
#include "lean_blocks.hpp"
#include "sac_shared.hpp"
#include "sac_lean.hpp"

namespace {

template <int X>
struct Net {
  static constexpr int KP = X, KQ = X + 1, D = 2 * X + 4;
  static constexpr int P_W1 = KP * LH + LH, P_OUT = P_W1 + 2 * HID, P = P_OUT + LH * 2 + 2;
  static constexpr int Q_W1 = KQ * LH + LH, Q_OUT = Q_W1 + 2 * HID, Q = Q_OUT + LH + 1;
};

// LDS carve (floats)
constexpr int O_QIN = 0;               // [16][8]  [sn | a]   (critic: the transition's action; actor: the sampled action)
constexpr int O_QIN2 = 128;            // [16][8]  [s'n | a'] (critic role)
constexpr int O_AUX = 256;             // [16][4]  reward, discount, truncation, transitions.action[..., -1]
constexpr int O_EPS = 320;             // [16] noise of the role's first sample
constexpr int O_EPS2 = 336;            // [16] actor role: noise of the alpha-loss sample, then that sample's log-prob
constexpr int O_LP = 352;              // [16]
constexpr int O_A = 368, O_SIG = 384, O_RAW = 400;
constexpr int O_Q = 416;               // [2][16] network values: critic role target critics, actor role Q1/Q2
constexpr int O_DQ = 448;              // [2][16] actor role: dQ/da
constexpr int O_QOLD = 480;            // [16] critic role: Q_k(s, a)
constexpr int O_DY = 496;              // [16][4] output-layer gradient of the network being differentiated
constexpr int O_FLAG = 560;            // [4] word 0: the aux wave's quick clip verdict for the end of the pass
constexpr int O_TILES = 564;
constexpr int N_TILES = 14;
constexpr int N_AUX = 3;                             // waves 8..10: no chain; noise, loss section, clip words (wave 8), gradient copy-out (all)
constexpr int LEAN_THREADS = 64 * (8 + N_AUX);
constexpr int O_STAGE = O_TILES + N_TILES * LT;      // [2][GST] weight gradients of the two hidden layers on their way to the slab
constexpr int GST = LH * LH + LH;                    // one hidden layer's block: dW [64][64] (16-byte chunks rotated per row) then db [64]
constexpr size_t LEAN_LDS_BYTES = (size_t)(O_STAGE + 2 * GST) * sizeof(float);

#define LEAN_STAMP(i)                                                                   \
  if (STAMPS) {                                                                         \
    if (A.stamps && tile == 0 && tid == 0) {                                            \
      unsigned long long t_;                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
      A.stamps[(trole == 1 ? 16 : 0) + (i)] = t_;                                       \
    }                                                                                   \
  }

}  // namespace

// per-wave fine stamps (diagnostic instantiation only): wave `w`'s first lane writes slot 32 + i
#define FINE_STAMP(w, i)                                                                \
  if (STAMPS) {                                                                         \
    if (A.stamps && tile == 0 && trole == 0 && wave == (w) && lane == 0) {              \
      unsigned long long t_;                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
      A.stamps[32 + (i)] = t_;                                                          \
    }                                                                                   \
  }

template <int X, bool STAMPS>
__global__ void __launch_bounds__(LEAN_THREADS) k_sac_lean(const SacLeanArgs A) {
  extern __shared__ __align__(16) float smem[];
  using N = Net<X>;
  constexpr int D = N::D, KP = N::KP, KQ = N::KQ;
  static_assert(X >= 1 && X + 1 <= LDX, "x + u must fit an 8-column input tile");
  const int tid_ = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int c = wave >> 2, sub = wave & 3, c0 = sub * 16;
  const int trole = blockIdx.x % 3, tile = blockIdx.x / 3;      // 0 = critic 0, 1 = actor + alpha, 2 = critic 1
  // The chain waves' arguments: ONE scalar load of the first 16 dwords, pinned in SGPRs here.  (Left to itself hipcc fetched the
  // argument block piecemeal next to each use, every piece with a full wait, and the clip words behind a pointer in between: six
  // dependent scalar round trips — 5.4 k cycles — before the tile load was issued.)
  const float *a_params = A.params, *a_target_q = A.target_q, *a_batch = A.batch, *a_norm_mean = A.norm_mean, *a_norm_std = A.norm_std;
  float *a_slab_pi = A.slab_pi, *a_slab_q = A.slab_q;
  int B = A.B;
  asm volatile("" : "+s"(a_params), "+s"(a_target_q), "+s"(a_batch), "+s"(a_norm_mean), "+s"(a_norm_std), "+s"(a_slab_pi), "+s"(a_slab_q), "+s"(B));
  const int row0 = tile * 16;
  {
    const int tid = tid_;
    LEAN_STAMP(0);
  }
  const float invB = 1.0f / (float)B;

  float *const s_qin = smem + O_QIN, *const s_qin2 = smem + O_QIN2, *const s_aux = smem + O_AUX;
  float *const tiles = smem + O_TILES;
#define TILE(n) (tiles + (n) * LT)

  // The clip check of the previous speculative optimizer step (sac.hip k_sac_fwd_bwd: same protocol, same words) is the aux wave's:
  // it reads the sequence numbers and the quick verdict, leaves "maybe clipped" in LDS for the end of the pass, and block 0's aux
  // wave publishes this step's slot, the optimizer count and the exchange epoch.  First pass only.
  auto aux_clip_words = [&]() __attribute__((always_inline)) {
    const uint4 qw0 = *reinterpret_cast<const uint4 *>(A.opt.seq), qw1 = *reinterpret_cast<const uint4 *>(A.opt.seq + 4);
    const unsigned int seq_issued = qw0.x, seq_resolved = qw0.y;
    const bool odd = (seq_issued & 1u) == 0u;
    const float q0 = __uint_as_float(odd ? qw1.y : qw0.z), q1 = __uint_as_float(odd ? qw1.z : qw0.w), q2 = __uint_as_float(odd ? qw1.w : qw1.x);
    const float lim = (A.opt.max_norm / A.opt.grad_scale) * (A.opt.max_norm / A.opt.grad_scale) * 0.9998f;
    const bool maybe = seq_issued != seq_resolved && !(q0 < lim && q1 < lim && q2 < lim);
    if (tid_ == 512) smem[O_FLAG] = maybe ? 1.f : 0.f;
    if (blockIdx.x == 0 && tid_ == 512) {
      const float count_in = A.step_count_rw[0];
      const unsigned int slot = qw0.x & 1u;
      A.opt.slot_word[0] = slot;
      float *q = reinterpret_cast<float *>(A.opt.seq) + 2 + 3 * slot;
      q[0] = 0.f; q[1] = 0.f; q[2] = 0.f;
      A.step_count_rw[0] = count_in + 1.0f;
      if (A.p2p_epoch) {
        const unsigned int ep0_in = A.p2p_epoch[0], ep1_in = A.p2p_epoch[1];
        A.p2p_epoch[0] = ep0_in + 1u;
        A.p2p_epoch[1] = ep1_in + A.p2p_blocks;
      }
    }
  };

  auto run = [&](const bool second_pass) __attribute__((always_inline)) {
    const int tid = opaque(tid_), lane = tid & 63;
    const float *const pi_p = a_params;
    // ---- the tile's transitions: ONE coalesced dword per thread (the 16 rows are contiguous), requested before anything else —
    // vector-memory results return in order, so whatever is requested in front of it is waited for with it.  Column cc of row r goes
    // to: obs -> s_qin, action -> s_qin[X] (critic role), reward / discount / truncation -> s_aux, next obs -> s_qin2 (critic role).
    const bool has_elem = tid < 16 * D;
    const int r_t = tid / D, cc_t = tid - r_t * D;
    const bool is_obs = cc_t < X, is_obs2 = cc_t >= X + 3 && cc_t < 2 * X + 3;
    const int oc_t = is_obs ? cc_t : (is_obs2 ? cc_t - (X + 3) : 0);
    float v_t = 0.f, mu_t = 0.f, sd_t = 1.f;
    if (has_elem && (trole != 1 || is_obs)) {
      const int nvalid = (B - row0 < 16 ? B - row0 : 16) * D;
      if (tid < nvalid) v_t = a_batch[(long long)row0 * D + tid];
      if (a_norm_mean && (is_obs || is_obs2)) {
        mu_t = a_norm_mean[oc_t];
        sd_t = a_norm_std[oc_t];
      }
    }
    auto tile_to_lds = [&]() __attribute__((always_inline)) {
      if (has_elem) {
        float v = v_t;
        if (a_norm_mean && (is_obs || is_obs2)) v = (v - mu_t) / sd_t;
        int dst = -1;
        if (is_obs) dst = O_QIN + r_t * LDX + cc_t;
        else if (trole != 1) {
          if (is_obs2) dst = O_QIN2 + r_t * LDX + oc_t;
          else if (cc_t == X) dst = O_QIN + r_t * LDX + X;                  // transitions.action
          else if (cc_t == X + 1) dst = O_AUX + r_t * 4 + 0;                // reward
          else if (cc_t == X + 2) dst = O_AUX + r_t * 4 + 1;                // discount
          else dst = O_AUX + r_t * 4 + 2;                                   // truncation
        }
        if (dst >= 0) smem[dst] = v;
      }
    };
    if (wave >= 8) {
      const int ax = wave - 8;
      // =========================================== AUX WAVES ===========================================
      // The ninth wave walks no chain.  It owns what would otherwise sit on a chain wave's critical path: the sampling noise (Philox
      // rounds in the first hidden-layer interval, Box-Muller in the second: a chain wave that drew it behind its 46 weight requests
      // held the first barrier for 5.5 k cycles) and the loss section (its operands that exist early are in registers before the
      // output layers finish).  It executes the chain waves' 13 barriers.
      const bool critic = trole != 1;
      const int kq = trole >> 1;
      const RngKey rk = rng_resolve(A.seed, A.offset, A.rng_dev);
      const bool second = lane >= 16;                          // actor role: lanes 16..31 draw the alpha-loss noise
      const int r = lane & 15;
      const bool draws = ax == 0 && lane < (critic ? 16 : 32);
      const float *const given = critic ? A.noise_critic : (second ? A.noise_alpha : A.noise_actor);
      const unsigned int stream = critic ? MBPO_STREAM_SAC_CRITIC : (second ? MBPO_STREAM_SAC_ALPHA : MBPO_STREAM_SAC_ACTOR);
      const long long nidx = row0 + r;
      float e_given = 0.f;
      if (draws && given && row0 + r < B) e_given = given[nidx];
      __syncthreads();      // 1: tile in LDS
      if (ax == 0 && !second_pass) aux_clip_words();      // (a second pass must not bump the counters again)
      __syncthreads();      // 2: thin layer 0
      Philox4 bits;
      bits.v[0] = bits.v[1] = bits.v[2] = bits.v[3] = 0u;
      if (draws && !given) bits = philox_normal_bits(rk.seed, rk.offset, stream, (unsigned long long)nidx);
      __syncthreads();      // 3: hidden layer 1
      if (draws) {
        float e = 0.f;
        if (row0 + r < B) e = given ? e_given : philox_normal_from_bits(bits);
        smem[(second ? O_EPS2 : O_EPS) + r] = e;
      }
      __syncthreads();      // 4: hidden layer 2 (the noise is in LDS for the sampling section)
      __syncthreads();      // 5: output layer + sample
      __syncthreads();      // 6: thin layer (F1)
      __syncthreads();      // 7
      // (second pass: the fix-up has just rewritten log_alpha; an agent-scope load does not come from this CU's caches)
      const float log_alpha_v = second_pass ? __hip_atomic_load(a_params + N::P + 2 * N::Q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : a_params[N::P + 2 * N::Q];
      const float alpha = expf(log_alpha_v);
      const bool ok = row0 + r < B;
      if (critic) {
        // ---- targets, errors, dL/dq (:88-110) ----
        float nlp = 0.f;
        nlp += smem[O_LP + r];
        const float rew = s_aux[r * 4 + 0], disc = s_aux[r * 4 + 1], trunc = s_aux[r * 4 + 2], qold = smem[O_QOLD + r];
        float gamma = A.discounting;
        if (A.neq) {                                                                             // :90-96
          const float pseudo = s_qin[r * LDX + X];                                               // transitions.action[..., -1]
          float tfa = (A.neq_tu - A.neq_tl) / 2.0f * pseudo + (A.neq_tu + A.neq_tl) / 2.0f;
          tfa = floor_divide_f(tfa, A.neq_dt) * A.neq_dt;
          gamma = expf(-A.neq_cd * tfa);
        }
        __syncthreads();    // 8
        __syncthreads();    // 9: the target critics' outputs
        float e2 = 0.f;
        if (ax == 0 && lane < 16) {
          const float nq = fminf(smem[O_Q + r], smem[O_Q + 16 + r]);
          const float next_v = nq - alpha * nlp;                                                   // :89
          const float target = rew * A.reward_scaling + disc * gamma * next_v;                     // :101-103
          const float err = ok ? (qold - target) * (1.f - trunc) : 0.f;                            // :104-108
          e2 = err * err;
          smem[O_DY + r * 4] = err * (1.f - trunc) * (0.5f * invB);
        }
        const float acc = row_sum16(e2);
        if (ax == 0 && lane == 0) A.slab_ex[tile * 4 + (kq == 1 ? 3 : 0)] = acc;
      } else {
        float lp_al = 0.f, lp_ac = 0.f;
        lp_al += smem[O_EPS2 + r];
        lp_ac += smem[O_LP + r];
        const float a = smem[O_A + r], sg = smem[O_SIG + r], eps = smem[O_EPS + r], raw = smem[O_RAW + r];
        __syncthreads();    // 8
        __syncthreads();    // 9: the critics' values and tangents
        float l_al = 0.f, l_ac = 0.f;
        if (ax == 0 && lane < 16) {
          l_al = ok ? alpha * (-lp_al - A.target_entropy) : 0.f;            // alpha_loss (:70-72)
          const float q0 = smem[O_Q + r], q1 = smem[O_Q + 16 + r];
          const float mq = fminf(q0, q1);
          l_ac = ok ? (alpha * lp_ac - mq) : 0.f;                           // actor_loss (:123-124)
          float g0 = 0.f, g1 = 0.f;
          if (ok) {
            if (q0 < q1) g0 = -invB;
            else if (q1 < q0) g1 = -invB;
            else g0 = g1 = -0.5f * invB;
          }
          const float dLda = g0 * smem[O_DQ + r] + g1 * smem[O_DQ + 16 + r];
          const float gz = dLda * (1.f - a * a) + alpha * invB * 2.f * a;
          const float gsig = gz * eps - alpha * invB / sg;
          smem[O_DY + r * 4] = ok ? gz : 0.f;                               // d/dloc
          smem[O_DY + r * 4 + 1] = ok ? gsig * fast_sigmoid(raw) : 0.f;     // d/draw
        }
        const float al = row_sum16(l_al), ac = row_sum16(l_ac);
        if (ax == 0 && lane == 0) {
          A.slab_ex[tile * 4 + 1] = ac;
          A.slab_ex[tile * 4 + 2] = al;
        }
      }
      __syncthreads();      // 10: dY
      __syncthreads();      // 11
      __syncthreads();      // 12: dW2 is in stage 0 — moved to the slab while the chain waves walk layer 1
      float *const aslab = critic ? a_slab_q + (long long)tile * (2 * N::Q) + kq * N::Q : a_slab_pi + (long long)tile * N::P;
      const int w1 = critic ? N::Q_W1 : N::P_W1, kk = critic ? KQ : KP;
      FINE_STAMP(8, 16);
      copy_stage_out(smem + O_STAGE, aslab + w1 + HID, ax * 64 + lane, N_AUX * 64);
      FINE_STAMP(8, 17);
      __syncthreads();      // 13: dW1 is in stage 1 — moved by this wave and the chain waves that have no row of layer 0's gradient
      FINE_STAMP(8, 18);
      copy_stage_out(smem + O_STAGE + GST, aslab + w1, (7 - kk + ax) * 64 + lane, (7 - kk + N_AUX) * 64);
      FINE_STAMP(8, 19);
    } else if (trole != 1) {
      // =========================================== CRITIC kq (sac/losses.py:74-110) ===========================================
      const int kq = trole >> 1;
      const float *const qk_p = a_params + N::P + kq * N::Q;
      const float *const qt_p = a_target_q + c * N::Q;                  // the target critic chain c walks in F1
      // ---- requests of phase F0: chain 0 = pi(s'), chain 1 = Q_k(s, a) ----
      float tw[KQ + 1];
      ImgF I1, I2;
      float wo[16];
      float bo0 = 0.f, bo1 = 0.f;
      if (c == 0) {
        thin_col_request<KP>(*reinterpret_cast<float(*)[KP + 1]>(&tw[0]), pi_p, lane);
        img_fwd_request(I1, pi_p + N::P_W1, c0, lane);
        if (sub == 0) {
          img_out_request<2>(wo, pi_p + N::P_OUT, lane);
          bo0 = pi_p[N::P_OUT + LH * 2];
          bo1 = pi_p[N::P_OUT + LH * 2 + 1];
        }
      } else {
        thin_col_request<KQ>(tw, qk_p, lane);
        img_fwd_request(I1, qk_p + N::Q_W1, c0, lane);
        if (sub == 1) {
          img_out_request<1>(wo, qk_p + N::Q_OUT, lane);
          bo0 = qk_p[N::Q_OUT + LH];
        }
      }
      tile_to_lds();
      __syncthreads();
      LEAN_STAMP(1);
      // ---- F0 thin layer; the target critics' thin columns are requested here (6 requests) ----
      float tw1[KQ + 1];
      ImgF J1, J2;
      if (c == 0) thin_first<KP, false, false>(*reinterpret_cast<float(*)[KP + 1]>(&tw[0]), s_qin2, TILE(0), nullptr, nullptr, sub, lane);
      else thin_first<KQ, true, false>(tw, s_qin, TILE(7), TILE(4), nullptr, sub, lane);
      thin_col_request<KQ>(tw1, qt_p, lane);
      // (the second hidden layer's images are requested here, not at the top: eight waves x 17 requests less in the prologue's burst)
      img_fwd_request(I2, (c == 0 ? pi_p + N::P_W1 : qk_p + N::Q_W1) + HID, c0, lane);
      __syncthreads();
      LEAN_STAMP(2);
      // ---- F0 hidden layers; F1's two images are requested in their MFMA shadows ----
      {
        auto pf = make_pf([&](int s) __attribute__((always_inline)) { img_fwd_request_piece(J1, qt_p + N::Q_W1, c0, lane, s); });
        if (c == 0) hid_fwd<false>(I1, TILE(0), TILE(1), nullptr, c0, lane, pf);
        else hid_fwd<true>(I1, TILE(7), TILE(8), TILE(5), c0, lane, pf);
      }
      __syncthreads();
      LEAN_STAMP(3);
      {
        auto pf = make_pf([&](int s) __attribute__((always_inline)) { img_fwd_request_piece(J2, qt_p + N::Q_W1 + HID, c0, lane, s); });
        if (c == 0) hid_fwd<false>(I2, TILE(1), TILE(0), nullptr, c0, lane, pf);
        else hid_fwd<true>(I2, TILE(8), TILE(9), TILE(6), c0, lane, pf);
      }
      __syncthreads();
      LEAN_STAMP(4);
      // ---- output layers + the sampling section on the wave that holds the policy's output; the idle waves request what comes
      // later: F1's output images (waves 1 and 6) and the backward phase's images (chain 0) ----
      float wo1[16];
      float bt = 0.f;
      float two[1];
      float G2[16], G1[16];
      if (c == 0 && sub == 0) {
        const f32x4 y = out_fwd(wo, TILE(0), lane);
        const float loc = y[0] + bo0, raw = y[1] + bo1;
        const int j = lane & 15;
        const ActSample sm = normal_tanh_sample(loc, raw, smem[O_EPS + j]);      // next_action, next_log_prob (:80-87)
        if (lane < 16) {
          s_qin2[j * LDX + X] = sm.a;
          smem[O_LP + j] = sm.lp;
        }
      } else if (c == 1 && sub == 1) {
        const f32x4 y = out_fwd(wo, TILE(9), lane);
        if (lane < 16) smem[O_QOLD + lane] = y[0] + bo0;                         // q_old_action (:78-79), this workgroup's critic
      }
      if (sub == c + 1) {
        img_out_request<1>(wo1, qt_p + N::Q_OUT, lane);
        bt = qt_p[N::Q_OUT + LH];
      }
      if (c == 0) {
        two[0] = qk_p[N::Q_OUT + lane];
        img_dgrad_request(G2, qk_p + N::Q_W1 + HID, c0, lane);
        img_dgrad_request(G1, qk_p + N::Q_W1, c0, lane);
      }
      __syncthreads();
      LEAN_STAMP(5);
      // ---- F1: target critics ----
      thin_first<KQ, false, false>(tw1, s_qin2, TILE(2 * c), nullptr, nullptr, sub, lane);
      __syncthreads();
      LEAN_STAMP(6);
      hid_fwd<false>(J1, TILE(2 * c), TILE(2 * c + 1), nullptr, c0, lane);
      __syncthreads();
      LEAN_STAMP(7);
      hid_fwd<false>(J2, TILE(2 * c + 1), TILE(2 * c), nullptr, c0, lane);
      __syncthreads();
      LEAN_STAMP(8);
      // output layers (waves 1 and 6); the backward phase's LDS operands that exist already are read meanwhile
      float zq[4], hc[16];
      if (sub == c + 1) {
        const f32x4 y = out_fwd(wo1, TILE(2 * c), lane);
        if (lane < 16) smem[O_Q + 16 * c + lane] = y[0] + bt;
      }
      if (c == 0) thin_z_preload(zq, TILE(6), sub, lane);
      else if (sub == 0) thin_col_preload(hc, TILE(9), lane);
      __syncthreads();
      LEAN_STAMP(9);
      // (the loss section runs on the aux wave)
      __syncthreads();
      LEAN_STAMP(10);
      // ---- backward of Q_k: chain 0 input gradients, chain 1 weight gradients ----
      float *const slab = a_slab_q + (long long)tile * (2 * N::Q) + kq * N::Q;
      if (c == 0) thin_dgrad_last<1>(two, smem + O_DY, zq, TILE(0), sub, lane);
      else if (sub == 0) thin_wgrad_last<1>(hc, smem + O_DY, slab + N::Q_OUT, sub, lane);
      __syncthreads();
      LEAN_STAMP(11);
      {
        auto st0 = [&](int i) __attribute__((always_inline)) { FINE_STAMP(0, i); };
        auto st4 = [&](int i) __attribute__((always_inline)) { FINE_STAMP(4, 8 + i); };
        if (c == 0) hid_dgrad(G2, TILE(0), TILE(5), TILE(1), c0, lane, st0);                       // delta_1
        else hid_wgrad(TILE(8), TILE(0), smem + O_STAGE, c0, lane, st4);                           // dW2 = h1^T delta_2 -> stage 0
      }
      FINE_STAMP(0, 4);
      FINE_STAMP(4, 12);
      __syncthreads();
      FINE_STAMP(0, 5);
      FINE_STAMP(4, 13);
      LEAN_STAMP(12);
      if (c == 0) hid_dgrad(G1, TILE(1), TILE(4), TILE(2), c0, lane);                              // delta_0
      else hid_wgrad(TILE(7), TILE(1), smem + O_STAGE + GST, c0, lane);                            // dW1 = h0^T delta_1 -> stage 1
      FINE_STAMP(0, 20);
      FINE_STAMP(4, 21);
      __syncthreads();
      FINE_STAMP(0, 22);
      LEAN_STAMP(13);
      // layer 0's weight gradient on waves 0..K; the other chain waves help the aux wave move stage 1 to the slab
      if (wave <= KQ) thin_wgrad_first_<KQ>(s_qin, TILE(2), slab, wave, lane);
      else copy_stage_out(smem + O_STAGE + GST, slab + N::Q_W1, (wave - (KQ + 1)) * 64 + lane, (7 - KQ + N_AUX) * 64);
      LEAN_STAMP(14);
    } else {
      // =========================================== ACTOR + ALPHA (sac/losses.py:61-72, 112-125) ===========================================
      const float *const q_p = a_params + N::P + c * N::Q;              // the critic chain c walks in F1
      // ---- requests: chain 0 = pi(s) for F0; chain 1 has no F0 work and requests its F1 images (Q2) at once ----
      float tw[KP + 1];
      ImgF I1, I2;
      float wo[16];
      float bo0 = 0.f, bo1 = 0.f;
      float tw1[KQ + 1];
      ImgF J1, J2;
      float wo1[16];
      float bq = 0.f;
      if (c == 0) {
        thin_col_request<KP>(tw, pi_p, lane);
        img_fwd_request(I1, pi_p + N::P_W1, c0, lane);
        if (sub == 0) {
          img_out_request<2>(wo, pi_p + N::P_OUT, lane);
          bo0 = pi_p[N::P_OUT + LH * 2];
          bo1 = pi_p[N::P_OUT + LH * 2 + 1];
        }
      }      // (chain 1 walks nothing before F1: its requests wait behind the first barrier, out of the launch's first burst)
      tile_to_lds();
      __syncthreads();
      LEAN_STAMP(1);
      // (the second hidden layer's images are requested here, not at the top: eight waves x 17 requests less in the prologue's burst)
      if (c == 0) {
        thin_first<KP, true, false>(tw, s_qin, TILE(7), TILE(4), nullptr, sub, lane);
        thin_col_request<KQ>(tw1, q_p, lane);
        img_fwd_request(I2, pi_p + N::P_W1 + HID, c0, lane);
      } else {
        thin_col_request<KQ>(tw1, q_p, lane);
        img_fwd_request(J1, q_p + N::Q_W1, c0, lane);
        img_fwd_request(J2, q_p + N::Q_W1 + HID, c0, lane);
        if (sub == 2) {
          img_out_request<1>(wo1, q_p + N::Q_OUT, lane);
          bq = q_p[N::Q_OUT + LH];
        }
      }
      __syncthreads();
      LEAN_STAMP(2);
      if (c == 0) {
        auto pf = make_pf([&](int s) __attribute__((always_inline)) { img_fwd_request_piece(J1, q_p + N::Q_W1, c0, lane, s); });
        hid_fwd<true>(I1, TILE(7), TILE(8), TILE(5), c0, lane, pf);
      }
      __syncthreads();
      LEAN_STAMP(3);
      if (c == 0) {
        auto pf = make_pf([&](int s) __attribute__((always_inline)) { img_fwd_request_piece(J2, q_p + N::Q_W1 + HID, c0, lane, s); });
        hid_fwd<true>(I2, TILE(8), TILE(9), TILE(6), c0, lane, pf);
      }
      __syncthreads();
      LEAN_STAMP(4);
      float two[2];
      float G2[16], G1[16];
      if (wave == 0) {
        // policy output; lane group 0 draws the actor-loss sample (:117-120), lane group 1 the alpha-loss sample (:66-68)
        const f32x4 y = out_fwd(wo, TILE(9), lane);
        const float loc = y[0] + bo0, raw = y[1] + bo1;
        const int j = lane & 15, g = lane >> 4;
        const ActSample sm = normal_tanh_sample(loc, raw, smem[(g == 1 ? O_EPS2 : O_EPS) + j]);
        if (g == 0) {
          smem[O_LP + j] = sm.lp;
          smem[O_A + j] = sm.a;
          smem[O_SIG + j] = sm.sigma;
          smem[O_RAW + j] = raw;
          s_qin[j * LDX + X] = sm.a;      // postprocess(action)
        } else if (g == 1) {
          smem[O_EPS2 + j] = sm.lp;
        }
      } else if (wave == 1) {
        img_out_request<1>(wo1, q_p + N::Q_OUT, lane);          // chain 0's F1 output image (Q1)
        bq = q_p[N::Q_OUT + LH];
      }
      if (c == 0) {
        // the policy's backward images
        two[0] = pi_p[N::P_OUT + lane * 2];
        two[1] = pi_p[N::P_OUT + lane * 2 + 1];
        img_dgrad_request(G2, pi_p + N::P_W1 + HID, c0, lane);
        img_dgrad_request(G1, pi_p + N::P_W1, c0, lane);
      }
      __syncthreads();
      LEAN_STAMP(5);
      // ---- F1: value tiles ping-pong through (0,1) / (10,11), tangent tiles through (2,3) / (12,13) ----
      const int tv0 = c == 0 ? 0 : 10, tt0 = c == 0 ? 2 : 12;
      thin_first<KQ, false, true>(tw1, s_qin, TILE(tv0), nullptr, TILE(tt0), sub, lane);
      __syncthreads();
      LEAN_STAMP(6);
      hid_fwd_jvp(J1, TILE(tv0), TILE(tt0), TILE(tv0 + 1), TILE(tt0 + 1), c0, lane);
      __syncthreads();
      LEAN_STAMP(7);
      hid_fwd_jvp(J2, TILE(tv0 + 1), TILE(tt0 + 1), TILE(tv0), TILE(tt0), c0, lane);
      __syncthreads();
      LEAN_STAMP(8);
      float zq[4], hc[16];
      if (sub == c + 1) {
        f32x4 y, ty;
        out_fwd2(wo1, TILE(tv0), TILE(tt0), lane, y, ty);
        if (lane < 16) {
          smem[O_Q + 16 * c + lane] = y[0] + bq;
          smem[O_DQ + 16 * c + lane] = ty[0];
        }
      }
      if (c == 0) thin_z_preload(zq, TILE(6), sub, lane);
      else if (sub < 2 && sub != 2) thin_col_preload(hc, TILE(9), lane);
      __syncthreads();
      LEAN_STAMP(9);
      // (the loss section runs on the aux wave)
      __syncthreads();
      LEAN_STAMP(10);
      // ---- backward of the policy ----
      float *const slab = a_slab_pi + (long long)tile * N::P;
      if (c == 0) thin_dgrad_last<2>(two, smem + O_DY, zq, TILE(0), sub, lane);
      else if (sub < 2) thin_wgrad_last<2>(hc, smem + O_DY, slab + N::P_OUT, sub, lane);
      __syncthreads();
      LEAN_STAMP(11);
      if (c == 0) hid_dgrad(G2, TILE(0), TILE(5), TILE(1), c0, lane);
      else hid_wgrad(TILE(8), TILE(0), smem + O_STAGE, c0, lane);
      __syncthreads();
      LEAN_STAMP(12);
      if (c == 0) hid_dgrad(G1, TILE(1), TILE(4), TILE(2), c0, lane);
      else hid_wgrad(TILE(7), TILE(1), smem + O_STAGE + GST, c0, lane);
      __syncthreads();
      LEAN_STAMP(13);
      if (wave <= KP) thin_wgrad_first_<KP>(s_qin, TILE(2), slab, wave, lane);
      else copy_stage_out(smem + O_STAGE + GST, slab + N::P_W1, (wave - (KP + 1)) * 64 + lane, (7 - KP + N_AUX) * 64);
      LEAN_STAMP(14);
    }
  };

  run(false);
  // the aux wave left the quick verdict in LDS before the second barrier; every wave has passed the thirteenth
  if (smem[O_FLAG] == 0.f) return;
  // RARE: the previous optimizer step may have needed clipping.  Canonical norms, the exact decision, and if a group really clips:
  // fix its step up from the undo log and run the whole pass again on the repaired parameters (sac.hip, same order of events).
  float *const s_gn = smem + O_DQ;
  __syncthreads();
  sac_group_norms(A.opt, s_gn, tid_);
  __syncthreads();
  if (s_gn[0] < A.opt.max_norm && s_gn[1] < A.opt.max_norm && s_gn[2] < A.opt.max_norm) return;
  if (blockIdx.x == 0 && tid_ == 0) A.opt.seq[SAC_CTL_CLIP_EVENTS] += 1u;
  sac_clip_fixup(A.opt, s_gn, opaque(tid_), LEAN_THREADS);
  __threadfence();
  __builtin_amdgcn_s_dcache_inv();      // the output layers' biases are scalar loads: the scalar cache may hold the pre-fix-up lines
  __syncthreads();
  run(true);
}

namespace {

template <int X>
int launch_x(const SacLeanArgs &A, int n_tiles, hipStream_t st) {
  int rc;
  if (A.stamps) {
    rc = mbpo_ensure_lds<k_sac_lean<X, true>>(LEAN_LDS_BYTES, "sac_lean");
    if (rc != MBPO_OK) return rc;
    hipLaunchKernelGGL((k_sac_lean<X, true>), dim3(3 * n_tiles), dim3(LEAN_THREADS), LEAN_LDS_BYTES, st, A);
  } else {
    rc = mbpo_ensure_lds<k_sac_lean<X, false>>(LEAN_LDS_BYTES, "sac_lean");
    if (rc != MBPO_OK) return rc;
    hipLaunchKernelGGL((k_sac_lean<X, false>), dim3(3 * n_tiles), dim3(LEAN_THREADS), LEAN_LDS_BYTES, st, A);
  }
  return MBPO_OK;
}

}  // namespace

bool sac_lean_supports(int x_dim, int u_dim, const int *policy_dims, int policy_layers, int policy_act, const int *q_dims, int q_layers, int q_act) {
  if (u_dim != 1 || x_dim < 2 || x_dim > 6) return false;      // (layer 0's weight gradient takes x + 2 of the eight chain waves)
  if (policy_layers != 4 || q_layers != 4 || policy_act != MBPO_ACT_SWISH || q_act != MBPO_ACT_SWISH) return false;
  for (int l = 1; l <= 3; ++l)
    if (policy_dims[l] != LH || q_dims[l] != LH) return false;
  return policy_dims[0] == x_dim && policy_dims[4] == 2 && q_dims[0] == x_dim + 1 && q_dims[4] == 1;
}

int sac_lean_launch(const SacLeanArgs &A, int x_dim, int n_tiles, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (x_dim == 3) return launch_x<3>(A, n_tiles, st);
  if (x_dim == 4) return launch_x<4>(A, n_tiles, st);
  // other observation widths: the plain instantiation only (no in-kernel timeline)
#define LEAN_X(X_)                                                                                                    \
  if (x_dim == X_) {                                                                                                  \
    int rc = mbpo_ensure_lds<k_sac_lean<X_, false>>(LEAN_LDS_BYTES, "sac_lean");                                      \
    if (rc != MBPO_OK) return rc;                                                                                     \
    hipLaunchKernelGGL((k_sac_lean<X_, false>), dim3(3 * n_tiles), dim3(LEAN_THREADS), LEAN_LDS_BYTES, st, A);        \
    return MBPO_OK;                                                                                                   \
  }
  LEAN_X(2) LEAN_X(5) LEAN_X(6)
#undef LEAN_X
  mbpo_set_error("sac_lean: x_dim %d has no instantiation", x_dim);
  return MBPO_ERR_UNSUPPORTED;
}

// sac_layered.hpp — SAC's forward/backward for shapes outside the fused kernel's range (sac_layered.hip), called by sac.hip.
#pragma once
#include "common.hpp"

// what block 0 of k_sac_fwd_bwd does at the top of every step, done by the layered path's first launch
struct SacLayeredBegin {
  float *step_count;            // optax's count: += 1
  unsigned int *seq;            // control words (SacOptArgs::seq): this step's quick-verdict slot is emptied
  unsigned int *slot_word;
  unsigned int *p2p_epoch;      // multi-GPU peer exchange epoch words, or null
  unsigned int p2p_blocks;
};
// floats of workspace the layered path needs behind the fused path's regions
long long sac_layered_floats(const mbpo_sac_desc *d, const MlpDev &pi, const MlpDev &q);
// leaves slab_pi [P], slab_q [2Q] (gradient of the minibatch, divided by B) and slab_ex [4] (loss sums): one "tile" for the reduction
int sac_layered_fwd_bwd(const mbpo_sac_desc *d, const MlpDev &pi, const MlpDev &q, const MlpDev &qt, float *ws, float *slab_pi,
                        float *slab_q, float *slab_ex, const SacLayeredBegin &bg, hipStream_t st);

// sac_lean.hpp — host interface of the SAC forward/backward kernel specialised for the benchmark networks (sac_lean.hip).
#pragma once
#include "sac_shared.hpp"

// Kernel arguments of k_sac_lean: ~50 dwords (the generic kernel's SacArgs is 3.5 KB of tables and needs a warm-up wave).
struct SacLeanArgs {
  // ---- what the chain waves need: the first 16 dwords, fetched with one scalar load at the top of the kernel ----
  const float *params;          // [policy | critic 0 | critic 1 | log_alpha]
  const float *target_q;        // [critic 0 | critic 1]
  const float *batch, *norm_mean, *norm_std;
  float *slab_pi, *slab_q;
  int B;
  int pad0;
  // ---- the aux wave's (noise, loss section, clip check of the previous speculative optimizer step, counters) ----
  float *slab_ex;
  const float *noise_alpha, *noise_critic, *noise_actor;
  const unsigned long long *rng_dev;
  unsigned long long seed, offset;
  float *step_count_rw;
  unsigned int *p2p_epoch;
  unsigned int p2p_blocks;
  int neq;
  float discounting, reward_scaling, target_entropy;
  float neq_cd, neq_tl, neq_tu, neq_dt;
  unsigned long long *stamps;   // measurement hook (mbpo_debug_set_stamps): selects the stamping instantiation, or NULL
  SacOptArgs opt;
};

// does the specialised kernel cover these networks?  (policy x -> 64^3 -> 2, critics x+1 -> 64^3 -> 1, swish, u = 1, x in {3, 4})
bool sac_lean_supports(int x_dim, int u_dim, const int *policy_dims, int policy_layers, int policy_act, const int *q_dims, int q_layers,
                       int q_act);
// 3 workgroups per 16-sample tile; writes the same per-tile slabs as k_sac_fwd_bwd (sac.hip), bit for bit
int sac_lean_launch(const SacLeanArgs &A, int x_dim, int n_tiles, void *stream);

// ppo_lean.hpp — host interface of the PPO loss forward/backward kernel specialised for the benchmark networks (ppo_lean.hip).
#pragma once
#include "common.hpp"

struct PpoLeanArgs {
  const float *params;                  // [policy | value]
  const float *data;                    // [M][2x + 2u + 4] rows of the shuffled minibatch (ppo.py:142-156)
  const float *norm_mean, *norm_std;
  const float *adv, *vs, *mom;          // per-row GAE advantages / value targets (compute_gae), mom = {mean, std} of the advantages
  const float *ent_noise;               // given entropy-sample noise, or NULL (Philox)
  const unsigned long long *rng_dev;
  unsigned long long seed, offset;
  float *slabs, *extras;                // per workgroup: gradient slab [policy | value], 4 loss partials
  long long M;
  float entropy_cost, clip_eps;
  int normalize_advantage;
  unsigned long long *stamps;           // measurement hook (mbpo_debug_set_ppo_stamps): s_memtime per section of workgroup 0's first tiles, or NULL
};

// policy x -> 64^3 -> 2, value x -> 64^3 -> 1 (or both 64^2), swish, u = 1, x = 2 .. 6
bool ppo_lean_supports(int x_dim, int u_dim, const int *policy_dims, int policy_layers, int policy_act, const int *value_dims, int value_layers,
                       int value_act);
// one workgroup per CU (n_wgs <= tiles), each walks tiles blockIdx.x, + n_wgs, ... and leaves ONE slab
int ppo_lean_launch(const PpoLeanArgs &A, int x_dim, int n_hid, int n_wgs, void *stream);      // n_hid: 2 (64 x 3) or 1 (64 x 2, both networks)

// ---- values + GAE + advantage-moment partials in one launch (ppo.hip k_ppo_values_gae) for the same value network ----
struct PpoVgLeanArgs {
  const float *v_params;                // the value network
  const float *data;                    // [B][T][D] rows of the shuffled minibatch
  const float *norm_mean, *norm_std;
  int B, T, D, G;                       // G whole trajectories per workgroup (G * (T + 1) <= 1024)
  int n_hid;                            // 64 x 64 layers of the value network: 2 (three hidden layers) or 1 (two)
  float reward_scaling, discounting, gae_lambda;
  float *vs, *adv;                      // [B][T]
  float *mom_part;                      // [workgroups][4]: {n, mean, M2, -}
  float *step_count_rw;
};
int ppo_vg_lean_launch(const PpoVgLeanArgs &A, int x_dim, int n_wgs, size_t lds_extra_floats, void *stream);
// the value network alone: x -> 64 -> 64 [-> 64] -> 1, swish, x = 2 .. 6
bool ppo_vg_lean_supports(int x_dim, const int *value_dims, int value_layers, int value_act);

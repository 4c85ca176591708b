// ppo_layered.hpp — PPO's minibatch forward/backward for shapes outside the fused kernels' range (ppo_layered.hip), called by ppo.hip.
#pragma once
#include "common.hpp"

struct PpoLayeredBufs;
// floats of workspace the layered path needs behind the fused path's regions
long long ppo_layered_floats(const mbpo_ppo_desc *d, const MlpDev &pi, const MlpDev &v);
// step 1: normalised observations of the M = B*T samples and the B bootstrap rows, the GAE inputs (trunc, term, rew: [M]), the value
// net on all M + B rows with stored activations -> values [M + B] (baseline = values, bootstrap = values + M); bumps step_count
int ppo_layered_values(const mbpo_ppo_desc *d, const MlpDev &pi, const MlpDev &v, float *ws, float *trunc, float *term, float *rew,
                       float **values_out, hipStream_t st);
// step 2 (after the GAE scan and the advantage moments): policy forward, loss terms and output gradients (ppo/losses.py:91-126),
// both backward passes.  Leaves slab [P + V] (the minibatch's gradient, divided by M) and *n_extras_out loss partials {policy, value,
// entropy, -} at *extras_out (one per 256 rows; the reduction launch adds them in order).
int ppo_layered_fwd_bwd(const mbpo_ppo_desc *d, const MlpDev &pi, const MlpDev &v, float *ws, const float *vs, const float *adv,
                        const float *mom, float *slab, float **extras_out, int *n_extras_out, hipStream_t st);

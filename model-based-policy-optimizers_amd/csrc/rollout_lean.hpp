// rollout_lean.hpp — host interface of the rollout kernel specialised for the benchmark networks (rollout_lean.hip).
#pragma once
#include "common.hpp"
#include "rollout_shared.hpp"

struct RoLeanArgs {
  RolloutArgs a;
  int E;                        // member networks (0: the analytic Pendulum system)
  int n_dyn_out;                // outputs of a member network (x_dim or 2 * x_dim)
  unsigned long long *stamps;   // measurement hook (mbpo_debug_set_rollout_stamps): s_memtime at the phase boundaries of workgroup 0, step 1
};

// policy x -> 64 -> 64 -> 64 -> 2 (swish), u = 1, x = 2 .. 4; Pendulum system, or <= 5 members (x + 1) -> 64 -> 64 -> 64 -> (x | 2x)
// (swish); action_repeat 1; closed loop (no open-loop actions)
bool rollout_lean_supports(const RolloutArgs &A, bool has_policy, int E);
// pipe: two tiles in flight per workgroup (members only; grid = workgroups, each walking PAIRS of tiles)
int rollout_lean_launch(const RoLeanArgs &A, int grid, bool pipe, void *stream);

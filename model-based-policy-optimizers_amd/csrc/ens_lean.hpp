// ens_lean.hpp — host interface of the ensemble forward kernel specialised for 64-wide member networks (ens_lean.hip).
#pragma once
#include "common.hpp"

struct EnsLeanArgs {
  const float *params;          // member e at params + e * net_stride
  long long net_stride;
  const float *x;               // [n_rows][K] shared by the members, or [E][n_rows][K]
  float *y;                     // [E][n_rows][N]
  long long n_rows;
  int E, N, shared_input, wgs_per_member;
  int n_hid;                    // 64 x 64 layers: 2 (three hidden layers) or 1 (two)
};

// K = 3 .. 7 inputs, two or three 64-wide hidden layers, N <= 16 outputs, swish
bool ens_lean_supports(const int *dims, int n_layers, int act);
int ens_lean_launch(const EnsLeanArgs &A, int K, int n_cus, void *stream);

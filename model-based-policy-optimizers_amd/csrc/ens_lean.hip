// ens_lean.hip — k_ens_fwd_lean<K>: mbpo_ensemble_mlp_forward (R2 of SURVEY §8a: the vmapped Dynamics.next_state of a learned ensemble,
// base_dynamics.py:15-20) for member networks K -> 64 -> 64 -> 64 -> N (K = x + u in 3 .. 7, N <= 16, swish), in the THROUGHPUT regime.
//
// The generic k_ensemble_forward (rollout.hip) gives every (16-row tile, member) chain to ONE wave that re-requests each layer's 16 KB
// of weights for every tile and walks 64 MFMAs per layer alone: 20 % of the fp32-MFMA roof at N = 32768 rows.
// Here (round 4, the blocks of lean_blocks.hpp):
//  * a workgroup is bound to ONE member: its thin column, two hidden images and output image are loaded once and stay in registers
//    for all the tiles the workgroup walks;
//  * 8 waves = 2 chains x 4 waves: two tiles of that member in flight per workgroup, and two workgroups per CU (<= 128 VGPRs), so four
//    independent layer steps share a CU's matrix pipes and one tile's LDS round trips and barriers hide behind the others' MFMAs;
//  * three rotating hidden tiles per chain: four barriers per pair of tiles and none behind the output layer, whose results go from
//    registers straight to global memory as 16-byte stores; the next pair's input rows are requested one iteration ahead.
#include "common.hpp"
#include "chain_run.hpp"
#include "lean_blocks.hpp"
#include "ens_lean.hpp"

namespace {
constexpr int E_X = 0;                 // [2 chains][2 parities][16][8] input tiles
constexpr int E_TILES = 512;           // [2 chains][3] hidden tiles
constexpr size_t ENS_LEAN_LDS_BYTES = (size_t)(E_TILES + 6 * LT) * sizeof(float);
}  // namespace

template <int K>
__global__ void __launch_bounds__(512, 4) k_ens_fwd_lean(const EnsLeanArgs A) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wave >> 2, sub = wave & 3, c0 = sub * 16;
  const int e = blockIdx.x % A.E, jw = blockIdx.x / A.E;              // member, index among that member's workgroups
  const int N = A.N;
  const long long n_rows = A.n_rows, n_tiles = (n_rows + 15) >> 4;
  const float *const net_p = A.params + (long long)e * A.net_stride;
  constexpr int W1 = K * LH + LH;
  const int nh = A.n_hid, OUT = W1 + nh * HID;
  float *const tiles = smem + E_TILES + c * 3 * LT;

  // ---- this chain's rows of the first pair (requested before the weights: results return in order) ----
  const int ct = tid & 255;                                            // thread within the chain's four waves
  const bool has_elem = ct < 16 * K;
  const float *const xin = A.x + (A.shared_input ? 0 : (long long)e * n_rows * K);
  auto tile_request = [&](long long tile) __attribute__((always_inline)) -> float {
    float v = 0.f;
    if (has_elem && tile < n_tiles) {
      const long long r0 = tile * 16;
      const long long nvalid = (n_rows - r0 < 16 ? n_rows - r0 : 16) * K;
      if (ct < nvalid) v = xin[r0 * K + ct];
    }
    return v;
  };
  const long long stride = 2LL * A.wgs_per_member;
  long long tile = 2LL * jw + c;
  float v_next = tile_request(tile);

  // ---- the member's weights, once per launch ----
  float tw[K + 1];
  ImgF I1, I2;
  float wo[16], bo[4];
  thin_col_request<K>(tw, net_p, lane);
  img_fwd_request(I1, net_p + W1, c0, lane);
  img_fwd_request(I2, net_p + W1 + (nh - 1) * HID, c0, lane);      // (one 64 x 64 layer: requested again, never used)
  const bool out_wave = sub == c;                                      // waves 0 and 5: different SIMDs
  if (out_wave) {
    const int i = lane & 15, g = lane >> 4;
    const float *p = net_p + OUT + (16 * g) * N + (i < N ? i : 0);      // matrix row i = output column i
#pragma unroll
    for (int s = 0; s < 16; ++s) wo[s] = p[s * N];
#pragma unroll
    for (int q = 0; q < 4; ++q) bo[q] = net_p[OUT + LH * N + (4 * g + q < N ? 4 * g + q : 0)];
  }

  int par = 0;
#pragma nounroll
  for (; tile - c < n_tiles; tile += stride, par ^= 1) {               // (both chains leave together: the pair's first tile decides)
    float *const s_x = smem + E_X + (c * 2 + par) * 128;
    if (has_elem) s_x[(ct / K) * LDX + (ct % K)] = v_next;
    v_next = tile_request(tile + stride);
    __syncthreads();
    thin_first<K, false, false>(tw, s_x, tiles, nullptr, nullptr, sub, lane);
    __syncthreads();
    hid_fwd<false>(I1, tiles, tiles + LT, nullptr, c0, lane);
    __syncthreads();
    if (nh == 2) {
      hid_fwd<false>(I2, tiles + LT, tiles + 2 * LT, nullptr, c0, lane);
      __syncthreads();
    }
    if (out_wave && tile < n_tiles) {
      const f32x4 acc = out_fwd(wo, tiles + nh * LT, lane);
      const int j = lane & 15, g = lane >> 4;
      const long long row = tile * 16 + j;
      if (row < n_rows && 4 * g < N) {
        float *dst = A.y + ((long long)e * n_rows + row) * N + 4 * g;
        if (4 * g + 4 <= N) {
          const float o[4] = {acc[0] + bo[0], acc[1] + bo[1], acc[2] + bo[2], acc[3] + bo[3]};
          store_vec_global<4>(dst, o);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (4 * g + q < N) dst[q] = acc[q] + bo[q];
        }
      }
    }
    // (no barrier: the next pair's thin layer writes tile 0, last read two barriers ago; its input rows go to the other parity)
  }
}

bool ens_lean_supports(const int *dims, int n_layers, int act) {
  if ((n_layers != 4 && n_layers != 3) || act != MBPO_ACT_SWISH) return false;
  if (dims[0] < 3 || dims[0] > 7) return false;
  for (int l = 1; l < n_layers; ++l)
    if (dims[l] != LH) return false;
  return dims[n_layers] >= 1 && dims[n_layers] <= 16;
}

int ens_lean_launch(const EnsLeanArgs &A, int K, int n_cus, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const int grid = A.E * A.wgs_per_member;
  int rc;
#define EL_K_(K_)                                                                                     \
  if (K == K_) {                                                                                      \
    rc = mbpo_ensure_lds<k_ens_fwd_lean<K_>>(ENS_LEAN_LDS_BYTES, "ens_lean");                         \
    if (rc != MBPO_OK) return rc;                                                                     \
    hipLaunchKernelGGL(k_ens_fwd_lean<K_>, dim3(grid), dim3(512), ENS_LEAN_LDS_BYTES, st, A);         \
    (void)n_cus;                                                                                      \
    return MBPO_OK;                                                                                   \
  }
  EL_K_(3) EL_K_(4) EL_K_(5) EL_K_(6) EL_K_(7)
#undef EL_K_
  {
    mbpo_set_error("ens_lean: %d inputs have no instantiation", K);
    return MBPO_ERR_UNSUPPORTED;
  }
  (void)n_cus;
  return MBPO_OK;
}

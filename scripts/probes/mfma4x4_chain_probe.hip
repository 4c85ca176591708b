// Probe for DESIGN §7's plan: ONE wave walks a 64-wide MLP on a 4-ROW tile with v_mfma_f32_4x4x1 (16 blocks): block b covers
// columns 4b..4b+3, the same 4 rows in every block, so one instruction is a rank-1 update of the 4 x 64 tile.
//   A operand: lane l -> h[row l%4][k]      (wave-private LDS tile, 16 ds_read_b128 per layer, no barrier)
//   B operand: lane l -> W[k][l]            (64 coalesced row loads per layer)
//   D        : lane l, VGPR v -> out[row v][col l]
// Prints cycles per layer (s_memtime) for a lone wave and for 4 / 8 / 16 waves per CU, and checks the result against the CPU.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma4x4_chain_probe.hip -o scripts/probes/mfma4x4_chain_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define H 64
#define LD (H + 4)

__device__ __forceinline__ float swish_fast(float z) {
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z);
  return z * __builtin_amdgcn_rcpf(1.0f + e);
}

// x [tiles][4][H], W [L][H][H] (flax layout [in][out]), b [L][H], y [tiles][4][H]
__global__ void __launch_bounds__(512) k_chain(const float *x, const float *W, const float *b, float *y, int L, unsigned long long *cyc) {
  extern __shared__ __align__(16) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * (blockDim.x >> 6) + wave;
  float *h = smem + wave * 4 * LD;          // this wave's private [4][LD] tile
  for (int r = 0; r < 4; ++r) h[r * LD + lane] = x[((long long)tile * 4 + r) * H + lane];
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  const int row = lane & 3;
  // the weights of layer l+1 are requested (all 64 rows + bias) before layer l is computed: register double buffer, as in the
  // library's runners (left to itself hipcc sinks each load to its MFMA and waits for it there: 7.7 k cycles per layer)
  // two register sets swapped by unrolling the layer loop by two (no copies of in-flight registers); four independent
  // accumulators (k mod 4) so that consecutive MFMAs do not wait for each other
  float wa[H], wb[H], ba, bb;
#define REQUEST(wset, bset, ll)                                                    \
  {                                                                                \
    const int l_ = (ll) < L ? (ll) : L - 1;                                        \
    const float *W_ = W + (long long)l_ * H * H;                                   \
    _Pragma("unroll") for (int k = 0; k < H; ++k) wset[k] = W_[k * H + lane];      \
    bset = b[l_ * H + lane];                                                       \
  }
#define LAYER(wset, bset)                                                                                                 \
  {                                                                                                                       \
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;                                                           \
    f32x4 av[H / 4];   /* the whole A operand first (hipcc otherwise issues each ds_read right before its MFMAs and waits) */ \
    _Pragma("unroll") for (int q = 0; q < H / 4; ++q) av[q] = *reinterpret_cast<const f32x4 *>(h + row * LD + 4 * q);     \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    _Pragma("unroll") for (int q = 0; q < H / 4; ++q) {                                                                   \
      const f32x4 a = av[q];                                                                                              \
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], wset[4 * q + 0], a0, 0, 0, 0);                                        \
      a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], wset[4 * q + 1], a1, 0, 0, 0);                                        \
      a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], wset[4 * q + 2], a2, 0, 0, 0);                                        \
      a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], wset[4 * q + 3], a3, 0, 0, 0);                                        \
    }                                                                                                                     \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) h[v * LD + lane] = swish_fast(((a0[v] + a1[v]) + (a2[v] + a3[v])) + bset); \
  }
  REQUEST(wa, ba, 0);
#pragma nounroll
  for (int l = 0; l < L; l += 2) {
    REQUEST(wb, bb, l + 1);
    LAYER(wa, ba);
    REQUEST(wa, ba, l + 2);
    LAYER(wb, bb);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  for (int r = 0; r < 4; ++r) y[((long long)tile * 4 + r) * H + lane] = h[r * LD + lane];
  if (lane == 0) cyc[tile] = t1 - t0;
}

int main() {
  const int L = 8;
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int waves = cfg == 0 ? 1 : (cfg == 1 ? 4 : 8);          // waves per workgroup (<= 8: 256 VGPRs per wave available)
    const int blocks = cfg == 3 ? 512 : 256;                      // 512 workgroups of 8 waves = 16 waves per CU
    const int tiles = blocks * waves;
    std::vector<float> hx((size_t)tiles * 4 * H), hW((size_t)L * H * H), hb((size_t)L * H), hy(hx.size());
    srand(1);
    for (auto &v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto &v : hW) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.2f;
    for (auto &v : hb) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.1f;
    float *x, *W, *b, *y;
    unsigned long long *cyc;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&b, hb.size() * 4); hipMalloc(&y, hy.size() * 4);
    hipMalloc(&cyc, tiles * 8);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)waves * 4 * LD * 4;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64 * waves), lds, 0, x, W, b, y, L, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64 * waves), lds, 0, x, W, b, y, L, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(tiles);
    hipMemcpy(hc.data(), cyc, tiles * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
    // CPU check of tile 0 and the last tile
    double maxerr = 0;
    for (int t : {0, tiles - 1}) {
      float cur[4][H], nxt[4][H];
      for (int r = 0; r < 4; ++r) for (int c = 0; c < H; ++c) cur[r][c] = hx[((size_t)t * 4 + r) * H + c];
      for (int l = 0; l < L; ++l) {
        for (int r = 0; r < 4; ++r) for (int n = 0; n < H; ++n) {
          float z = hb[l * H + n];
          for (int k = 0; k < H; ++k) z = fmaf(cur[r][k], hW[((size_t)l * H + k) * H + n], z);
          nxt[r][n] = z / (1.0f + expf(-z));
        }
        for (int r = 0; r < 4; ++r) for (int c = 0; c < H; ++c) cur[r][c] = nxt[r][c];
      }
      for (int r = 0; r < 4; ++r) for (int c = 0; c < H; ++c) maxerr = fmax(maxerr, fabs(cur[r][c] - hy[((size_t)t * 4 + r) * H + c]));
    }
    double avg = 0;
    for (auto c : hc) avg += (double)c;
    avg /= tiles;
    const double flop = (double)tiles * 4 * L * 2.0 * H * H;
    printf("%2d waves/workgroup (%d workgroups, %5d tiles of 4 rows): %7.0f cycles per layer per wave, kernel %.1f us, %.2f TFLOP/s, max |err| %.2e\n",
           waves, blocks, tiles, avg / L, ms * 1e3 / 20, flop / (ms * 1e-3 / 20) / 1e12, maxerr);
    hipFree(x); hipFree(W); hipFree(b); hipFree(y); hipFree(cyc);
  }
  return 0;
}

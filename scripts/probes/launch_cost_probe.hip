// Fixed cost of a dependent kernel launch as a function of the launch shape (workgroups, threads, dynamic LDS, kernarg bytes):
// an (almost) empty kernel launched back to back on one stream, time per launch from HIP events.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/launch_cost_probe.hip -o /tmp/launch_cost_probe && /tmp/launch_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[880]; };   // ~3.5 KB of kernel arguments, like SacArgs
extern __shared__ float smem[];
__global__ void __launch_bounds__(512) k_small(float *out) { if (out && threadIdx.x == 9999) out[0] = smem[0]; }
__global__ void __launch_bounds__(512) k_big(Big b, float *out) { if (out && threadIdx.x == 9999) out[0] = smem[0] + b.v[threadIdx.x & 511]; }
__global__ void __launch_bounds__(512) k_big_touch(Big b, float *out) {
  // one barrier and one dependent global load behind the kernel arguments
  float v = out[b.v[0] & 3];
  smem[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 9999) out[0] = smem[1];
}
template <class F> static float time_it(F f, int n = 400) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / n * 1e3f;
}
int main() {
  float *d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  Big b{}; 
  hipFuncSetAttribute((const void *)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void *)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void *)k_big_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int wgs : {1, 48, 256}) for (int thr : {64, 512}) for (int lds : {0, 100 * 1024}) {
    float t0 = time_it([&] { hipLaunchKernelGGL(k_small, dim3(wgs), dim3(thr), lds, 0, d); });
    float t1 = time_it([&] { hipLaunchKernelGGL(k_big, dim3(wgs), dim3(thr), lds, 0, b, d); });
    float t2 = time_it([&] { hipLaunchKernelGGL(k_big_touch, dim3(wgs), dim3(thr), lds, 0, b, d); });
    printf("wgs %3d threads %3d lds %6d B: empty %.2f us | 3.5 KB kernarg %.2f us | + load + barrier %.2f us\n", wgs, thr, lds, t0, t1, t2);
  }
  // the same through a graph of 64 launches (no host launch cost)
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(k_big_touch, dim3(48), dim3(512), 100 * 1024, st, b, d);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  float tg = time_it([&] { hipGraphLaunch(ge, st); }, 50);
  printf("graph of 64 x (48 wgs, 512 threads, 100 KB LDS, kernarg + load + barrier): %.2f us per launch\n", tg / 64);
  return 0;
}

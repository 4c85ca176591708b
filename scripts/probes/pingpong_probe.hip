// Can the launch gap between DEPENDENT kernels be hidden by launching them on two streams and ordering them with device flags?
//
// The SAC update is A (fwd/bwd, ~15 us of dependent work + ~4.5 us of prologue that does not depend on the parameters) followed by
// B (slab reduction + optimizer step, ~0.5 us of work), 64 times per training step, each launch a ~2.9 us boundary in a hipGraph
// (launch_cost_probe.hip).  Here: the same shape with dummy work.  "serial" = A0 B0 A1 B1 ... on one stream (what the trainer does).
// "flags" = A0 A1 ... on stream 1 and B0 B1 ... on stream 2, no cross-stream edges; A_g spins until B_{g-1} has published, B_g until
// A_g has — sc1 (agent-scope) flag stores / loads, bounded spins (a timeout sets a sticky error word and every later wait falls
// through: nothing can hang).  A's prologue (work that needs nothing from B) runs BEFORE its wait.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/pingpong_probe.hip -o /tmp/pingpong_probe && /tmp/pingpong_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define SPIN_MAX 4000000
__constant__ int c_fence = 1;   // 1: release fence (buffer_wbl2) before the flag; 0: vmcnt(0) only (data stored sc1)

__device__ __forceinline__ void busy_cycles(long long cyc) {
  const long long t0 = __builtin_readcyclecounter();       // s_memtime: shader clock
  while (__builtin_readcyclecounter() - t0 < cyc) __builtin_amdgcn_s_sleep(2);
}

__device__ __forceinline__ bool wait_flag(const unsigned *flag, unsigned want, unsigned *err) {
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  for (int i = 0; i < SPIN_MAX; ++i) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// flags[0] = number of A workgroups finished (monotonic), flags[16] = number of B workgroups finished, flags[32] = error
__global__ void __launch_bounds__(512) k_a(unsigned *flags, int g, int n_a, int n_b, long long pro_cyc, long long main_cyc, int use_flags, float *data) {
  extern __shared__ float smem[];
  busy_cycles(pro_cyc);                                     // prologue: independent of B
  if (use_flags && g > 0) {
    if (threadIdx.x == 0) wait_flag(flags + 16, (unsigned)(g * n_b), flags + 32);
    __syncthreads();
  }
  smem[threadIdx.x] = data[(blockIdx.x * 512 + threadIdx.x) & 4095];   // "read the parameters"
  __syncthreads();
  busy_cycles(main_cyc);
  __hip_atomic_store(&data[4096 + ((blockIdx.x * 512 + threadIdx.x) & 4095)], smem[(threadIdx.x + 1) & 511], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // "write the slab"
  if (use_flags) {
    if (c_fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(256) k_b(unsigned *flags, int g, int n_a, int n_b, long long work_cyc, int use_flags, float *data) {
  if (use_flags) {
    if (threadIdx.x == 0) wait_flag(flags, (unsigned)((g + 1) * n_a), flags + 32);
    __syncthreads();
  }
  const float v = data[4096 + ((blockIdx.x * 256 + threadIdx.x) & 4095)];
  busy_cycles(work_cyc);
  __hip_atomic_store(&data[(blockIdx.x * 256 + threadIdx.x) & 4095], v * 0.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (use_flags) {
    if (c_fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static float run_graph(hipGraphExec_t ge, hipStream_t st, unsigned *flags, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipMemsetAsync(flags, 0, 256, st);
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main(int argc, char **argv) {
  const int G = 64, NA = 48, NB = argc > 2 ? atoi(argv[2]) : 103;
  const int fence = argc > 1 ? atoi(argv[1]) : 1;
  hipMemcpyToSymbol(HIP_SYMBOL(c_fence), &fence, sizeof(int));
  printf("fence=%d NB=%d\n", fence, NB);
  const double ghz = 2.1;                 // nominal shader clock for the dummy-work durations (only their ratio to the gaps matters)
  unsigned *flags; float *data;
  hipMalloc(&flags, 256); hipMalloc(&data, 8192 * sizeof(float));
  hipMemset(flags, 0, 256); hipMemset(data, 0, 8192 * sizeof(float));
  hipFuncSetAttribute((const void *)k_a, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  const long long pro = (long long)(4.5e3 * ghz), mainc = (long long)(15.0e3 * ghz), bw = (long long)(0.5e3 * ghz);
  // ---- serial: one stream, kernel boundaries order everything
  hipGraph_t g0; hipGraphExec_t ge0;
  hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal);
  for (int g = 0; g < G; ++g) {
    hipLaunchKernelGGL(k_a, dim3(NA), dim3(512), 70 * 1024, s1, flags, g, NA, NB, pro, mainc, 0, data);
    hipLaunchKernelGGL(k_b, dim3(NB), dim3(256), 0, s1, flags, g, NA, NB, bw, 0, data);
  }
  hipStreamEndCapture(s1, &g0); hipGraphInstantiate(&ge0, g0, nullptr, nullptr, 0);
  // ---- flags: two chains, no cross edges
  hipGraph_t g1; hipGraphExec_t ge1;
  hipEvent_t fork, join; hipEventCreate(&fork); hipEventCreate(&join);
  hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal);
  hipEventRecord(fork, s1);
  hipStreamWaitEvent(s2, fork, 0);
  for (int g = 0; g < G; ++g) hipLaunchKernelGGL(k_a, dim3(NA), dim3(512), 70 * 1024, s1, flags, g, NA, NB, pro, mainc, 1, data);
  for (int g = 0; g < G; ++g) hipLaunchKernelGGL(k_b, dim3(NB), dim3(256), 0, s2, flags, g, NA, NB, bw, 1, data);
  hipEventRecord(join, s2);
  hipStreamWaitEvent(s1, join, 0);
  hipStreamEndCapture(s1, &g1);
  hipError_t e = hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0);
  if (e != hipSuccess) { printf("instantiate failed: %s\n", hipGetErrorString(e)); return 1; }
  const float t_serial = run_graph(ge0, s1, flags, 5);
  const float t_flags = run_graph(ge1, s1, flags, 5);
  unsigned h[64]; hipMemcpy(h, flags, 256, hipMemcpyDeviceToHost);
  // ---- the same two chains launched directly (no graph): what the runtime's stream-to-queue mapping gives
  float t_direct = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipMemsetAsync(flags, 0, 256, s1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s1);
    hipEventRecord(fork, s1);
    hipStreamWaitEvent(s2, fork, 0);
    for (int g = 0; g < G; ++g) {
      hipLaunchKernelGGL(k_a, dim3(NA), dim3(512), 70 * 1024, s1, flags, g, NA, NB, pro, mainc, 1, data);
      hipLaunchKernelGGL(k_b, dim3(NB), dim3(256), 0, s2, flags, g, NA, NB, bw, 1, data);
    }
    hipEventRecord(join, s2);
    hipStreamWaitEvent(s1, join, 0);
    hipEventRecord(e1, s1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms * 1e3f < t_direct) t_direct = ms * 1e3f;
  }
  unsigned h2[64]; hipMemcpy(h2, flags, 256, hipMemcpyDeviceToHost);
  // ---- three streams: even A's on s1, odd A's on s3 (so A_{g+1}'s launch gap and prologue overlap A_g's main part), B's on s2
  hipStream_t s3; hipStreamCreate(&s3);
  hipEvent_t join3; hipEventCreate(&join3);
  float t_three = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipMemsetAsync(flags, 0, 256, s1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s1);
    hipEventRecord(fork, s1);
    hipStreamWaitEvent(s2, fork, 0);
    hipStreamWaitEvent(s3, fork, 0);
    for (int g = 0; g < G; ++g) {
      hipLaunchKernelGGL(k_a, dim3(NA), dim3(512), 70 * 1024, (g & 1) ? s3 : s1, flags, g, NA, NB, pro, mainc, 1, data);
      hipLaunchKernelGGL(k_b, dim3(NB), dim3(256), 0, s2, flags, g, NA, NB, bw, 1, data);
    }
    hipEventRecord(join, s2);
    hipEventRecord(join3, s3);
    hipStreamWaitEvent(s1, join, 0);
    hipStreamWaitEvent(s1, join3, 0);
    hipEventRecord(e1, s1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms * 1e3f < t_three) t_three = ms * 1e3f;
  }
  unsigned h3[64]; hipMemcpy(h3, flags, 256, hipMemcpyDeviceToHost);
  printf("dummy work per update: prologue 4.5 us + main 15 us (A, %d WGs) + 0.5 us (B, %d WGs)  [at %.1f GHz]\n", NA, NB, ghz);
  printf("serial (one stream, 128 boundaries): %.2f us per update\n", t_serial / G);
  printf("flags  (two chains, device flags)  : %.2f us per update   [A done %u/%d, B done %u/%d, timeout flag %u]\n", t_flags / G, h[0], G * NA,
         h[16], G * NB, h[32]);
  printf("direct (two streams, no graph)     : %.2f us per update   [A done %u, B done %u, timeout flag %u]\n", t_direct / G, h2[0], h2[16], h2[32]);
  printf("direct (three streams, A alternates): %.2f us per update   [A done %u, B done %u, timeout flag %u]\n", t_three / G, h3[0], h3[16], h3[32]);
  return 0;
}

// Does the instruction cache survive a kernel boundary?  A ~32 KB straight-line body timed with s_memtime:
// launch A (cold), launch A again (warm if the I$ survives), and inside one launch the body run twice (loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY() asm volatile(".rept 4000\n\tv_add_f32_e64 %0, %0, 1.0\n\t.endr" : "+v"(x))
__global__ void k_body(unsigned long long *out, float *sink, int reps) {
  float x = threadIdx.x;
  for (int r = 0; r < reps; ++r) {
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    BODY();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x * 8 + r] = t1 - t0;
  }
  sink[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void k_other(float *p) { p[threadIdx.x] += 1.f; }
int main() {
  unsigned long long *out, h[64];
  float *sink;
  hipMalloc(&out, 64 * 8);
  hipMalloc(&sink, 4096 * 4);
  for (int trial = 0; trial < 3; ++trial) {
    hipMemset(out, 0, 64 * 8);
    hipLaunchKernelGGL(k_body, dim3(1), dim3(64), 0, 0, out, sink, 3);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
    printf("trial %d: one launch, body x3: %llu %llu %llu cycles\n", trial, h[0], h[1], h[2]);
    hipLaunchKernelGGL(k_other, dim3(1), dim3(64), 0, 0, sink);
  }
  // same with 8 workgroups (different CUs / XCDs)
  hipMemset(out, 0, 64 * 8);
  hipLaunchKernelGGL(k_body, dim3(8), dim3(64), 0, 0, out, sink, 2);
  hipDeviceSynchronize();
  hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
  for (int b = 0; b < 8; ++b) printf("wg %d: %llu %llu\n", b, h[b * 8], h[b * 8 + 1]);
  return 0;
}

// p2p.hip — exchange regions (hipIpc) and the generic one-shot all-reduce (see p2p.hpp and include/mbpo_hip.h).
#include "common.hpp"
#include "p2p.hpp"
#include <string.h>

static int p2p_dev_from_desc(const mbpo_p2p_desc *d, P2pDev *P) {
  MBPO_REQUIRE(d, MBPO_ERR_ARG, "p2p: null descriptor");
  MBPO_REQUIRE(d->world >= 1 && d->world <= MBPO_P2P_MAX_RANKS && d->rank >= 0 && d->rank < d->world, MBPO_ERR_ARG,
               "p2p: bad world/rank %d/%d", d->world, d->rank);
  MBPO_REQUIRE(d->n_max > 0, MBPO_ERR_ARG, "p2p: n_max must be positive");
  P->world = d->world; P->rank = d->rank; P->n_max = d->n_max;
  for (int r = 0; r < d->world; ++r) {
    MBPO_REQUIRE(d->regions[r], MBPO_ERR_ARG, "p2p: region %d is not mapped", r);
    unsigned int *base = reinterpret_cast<unsigned int *>(d->regions[r]);
    P->flags[r] = base;
    P->slots[r] = reinterpret_cast<float *>(base + p2p_header_words());
  }
  unsigned int *own = reinterpret_cast<unsigned int *>(d->regions[d->rank]);
  P->epoch = own + p2p_flags_words();
  P->status = own + p2p_flags_words() + P2P_FLAG_STRIDE;
  return MBPO_OK;
}
int mbpo_p2p_make_dev(const mbpo_p2p_desc *d, P2pDev *P) { return p2p_dev_from_desc(d, P); }

extern "C" int64_t mbpo_p2p_region_bytes(int32_t world, int64_t n_max) {
  if (world < 1 || world > MBPO_P2P_MAX_RANKS || n_max <= 0) return MBPO_ERR_ARG;
  return p2p_region_bytes(world, n_max);
}

extern "C" int mbpo_p2p_alloc(int64_t bytes, void **ptr, void *handle64) {
  MBPO_REQUIRE(bytes > 0 && ptr && handle64, MBPO_ERR_ARG, "p2p_alloc: bad arguments");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  void *p = nullptr;
  // fine-grained: remote stores and the owner's loads bypass the non-coherent L2 paths
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&p, (size_t)bytes);
  }
  MBPO_REQUIRE(e == hipSuccess, MBPO_ERR_LAUNCH, "p2p_alloc: %s", hipGetErrorString(e));
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle64), p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    MBPO_REQUIRE(false, MBPO_ERR_LAUNCH, "p2p_alloc: %s", hipGetErrorString(e));
  }
  *ptr = p;
  return MBPO_OK;
}

extern "C" int mbpo_p2p_open(const void *handle64, int32_t peer_device, void **ptr) {
  MBPO_REQUIRE(handle64 && ptr, MBPO_ERR_ARG, "p2p_open: bad arguments");
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  MBPO_REQUIRE(e == hipSuccess, MBPO_ERR_LAUNCH, "p2p_open: %s", hipGetErrorString(e));
  if (peer_device >= 0 && peer_device != dev) {
    int can = 0;
    e = hipDeviceCanAccessPeer(&can, dev, peer_device);
    MBPO_REQUIRE(e == hipSuccess && can, MBPO_ERR_UNSUPPORTED, "p2p_open: device %d cannot access device %d", dev, peer_device);
    e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) {
      (void)hipGetLastError();
      e = hipSuccess;
    }
    MBPO_REQUIRE(e == hipSuccess, MBPO_ERR_LAUNCH, "p2p_open: enable peer access: %s", hipGetErrorString(e));
  }
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void *p = nullptr;
  e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  MBPO_REQUIRE(e == hipSuccess, MBPO_ERR_LAUNCH, "p2p_open: %s", hipGetErrorString(e));
  // touch the mapping through the runtime's copy path first: a mapping this device cannot reach then comes back as an error
  // code here (and the caller keeps the library collective) rather than as a memory fault inside the exchange kernels
  unsigned int word = 0;
  e = hipMemcpy(&word, p, sizeof(word), hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipIpcCloseMemHandle(p);
    MBPO_REQUIRE(false, MBPO_ERR_UNSUPPORTED, "p2p_open: mapped peer region is not readable from device %d: %s", dev, hipGetErrorString(e));
  }
  *ptr = p;
  return MBPO_OK;
}

extern "C" int mbpo_p2p_close(void *ptr) {
  hipError_t e = hipIpcCloseMemHandle(ptr);
  MBPO_REQUIRE(e == hipSuccess, MBPO_ERR_LAUNCH, "p2p_close: %s", hipGetErrorString(e));
  return MBPO_OK;
}

extern "C" int mbpo_p2p_free(void *ptr) {
  hipError_t e = hipFree(ptr);
  MBPO_REQUIRE(e == hipSuccess, MBPO_ERR_LAUNCH, "p2p_free: %s", hipGetErrorString(e));
  return MBPO_OK;
}

// ---- generic all-reduce: begin (epoch += 1) -> push -> gather ------------------------------------------------------
__global__ void k_p2p_begin(unsigned int *epoch, unsigned blocks) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    epoch[0] = epoch[0] + 1u;
    epoch[1] = epoch[1] + blocks;
  }
}

__global__ void __launch_bounds__(256) k_p2p_push(P2pDev P, const float *src, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned e = P.epoch[0];
  p2p_push(P, e, i, n, i < n ? src[i] : 0.f);
}

__global__ void __launch_bounds__(256) k_p2p_gather(P2pDev P, float *dst, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned e = P.epoch[0];
  const bool ok = p2p_wait(P, P.epoch[1]);
  if (i < n) dst[i] = ok ? p2p_sum(P, e, i) : NAN;
}

extern "C" int mbpo_p2p_all_reduce_sum(const mbpo_p2p_desc *d, float *buf, int64_t n, void *stream) {
  P2pDev P;
  int rc = p2p_dev_from_desc(d, &P);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(buf && n > 0 && n <= d->n_max, MBPO_ERR_ARG, "p2p_all_reduce_sum: n must be in (0, n_max]");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((n + 255) / 256);
  hipLaunchKernelGGL(k_p2p_begin, dim3(1), dim3(64), 0, st, P.epoch, (unsigned)blocks);
  hipLaunchKernelGGL(k_p2p_push, dim3(blocks), dim3(256), 0, st, P, (const float *)buf, (long long)n);
  hipLaunchKernelGGL(k_p2p_gather, dim3(blocks), dim3(256), 0, st, P, buf, (long long)n);
  MBPO_CHECK_LAUNCH("p2p_all_reduce_sum");
  return MBPO_OK;
}

extern "C" int mbpo_p2p_status(const mbpo_p2p_desc *d, int32_t *status_out) {
  P2pDev P;
  int rc = p2p_dev_from_desc(d, &P);
  if (rc != MBPO_OK) return rc;
  MBPO_REQUIRE(status_out, MBPO_ERR_ARG, "p2p_status: null output");
  unsigned int v = 0;
  hipError_t e = hipMemcpy(&v, P.status, sizeof(v), hipMemcpyDeviceToHost);
  MBPO_REQUIRE(e == hipSuccess, MBPO_ERR_LAUNCH, "p2p_status: %s", hipGetErrorString(e));
  *status_out = (int32_t)v;
  return MBPO_OK;
}

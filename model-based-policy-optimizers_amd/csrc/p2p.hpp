// p2p.hpp — one-shot all-reduce of a small vector over xGMI peer memory (device side).
//
// The SAC gradient is 26 309 floats (103 KiB): an all-reduce of that size is pure latency, and a library collective costs a
// kernel launch of its own plus a multi-step ring.  Here every rank owns an exchange region that all peers have mapped
// (hipIpc): a producer kernel writes its vector straight into slot[rank] of EVERY peer's region (stores over xGMI, no reads),
// publishes with a system-scope fence + one arrival count per block, and the consumer kernel on each rank waits for the
// arrivals of all ranks and adds the world slots in rank order — every rank forms bit-identical sums.
// Slots are double-buffered on the epoch's parity: a rank can only run two epochs ahead of a peer after that peer has pushed
// the epoch in between, which it does after it finished reading the older slot.
// Waiting is bounded (P2P_SPIN_MAX polls): on timeout the consumer reports failure instead of hanging the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef MBPO_P2P_MAX_RANKS
#define MBPO_P2P_MAX_RANKS 16
#endif
#define P2P_FLAG_STRIDE 16          // uint32 words between two flags (64 B apart)
#define P2P_SPIN_MAX (1 << 24)   // ~20-30 s of polling: longer than any start-up skew between ranks, still finite

struct P2pDev {
  int world, rank;
  long long n_max;                              // floats per slot
  unsigned int *flags[MBPO_P2P_MAX_RANKS];      // region r: arrival counters, one per source rank (monotonic)
  float *slots[MBPO_P2P_MAX_RANKS];             // region r: [2][world][n_max]
  unsigned int *epoch;                          // local: [0] number of exchanges started so far, [1] producer blocks expected so far
  unsigned int *status;                         // local: set to 1 by a consumer that timed out
};

// region layout (floats / words from the region base)
__host__ __device__ inline long long p2p_flags_words() { return (long long)MBPO_P2P_MAX_RANKS * P2P_FLAG_STRIDE; }
__host__ __device__ inline long long p2p_header_words() { return p2p_flags_words() + 2 * P2P_FLAG_STRIDE; }   // + epoch, status
__host__ inline long long p2p_region_bytes(int world, long long n_max) { return 4 * (p2p_header_words() + 2LL * world * n_max); }

// producer, called by every thread of a block with its element (i < n valid): store to all peers, then publish.
__device__ __forceinline__ void p2p_push(const P2pDev &P, unsigned epoch, long long i, long long n, float v) {
  const long long off = ((long long)(epoch & 1u) * P.world + P.rank) * P.n_max + i;
  if (i < n) {
    for (int p = 0; p < P.world; ++p) __builtin_nontemporal_store(v, P.slots[p] + off);
  }
  // Publish: every wave drains its own stores (an EXPLICIT s_waitcnt vmcnt(0): __syncthreads() alone does not wait for a wave's global
  // stores — a workgroup-scope release needs no vmcnt wait on this target), then ONE release at system scope by the first wave
  // covers the workgroup's stores, and lanes 0..world-1 of that wave add the arrival to THEIR peer's counter side by side — relaxed,
  // results unused.  (Rounds 1-3 fenced in every thread and then let one thread issue `world` release-ordered adds one after the
  // other: each of those waits for the previous remote add to complete before its own fence retires — `world` xGMI round trips in a
  // row per workgroup, invisible with two ranks on one GPU and an order of magnitude over the latency budget of DESIGN section 6.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if ((int)threadIdx.x < P.world)
      __hip_atomic_fetch_add(P.flags[threadIdx.x] + P.rank * P2P_FLAG_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// consumer: wait until `want` producer blocks (all exchanges so far; every rank runs the same sequence of exchanges) of every
// rank have arrived here.  Returns false on timeout (block-uniform).
__device__ __forceinline__ bool p2p_wait(const P2pDev &P, unsigned want) {
  __shared__ int s_p2p_ok;
  if (threadIdx.x < 64) {
    // Lane r of the first wave polls rank r's arrival counter — all ranks at once, relaxed system-scope loads (they do not come from
    // this CU's caches) — and ONE acquire behind the loop covers the slot reads of every wave behind the barrier.  (Rounds 1-3: one
    // thread, rank after rank, an acquire-ordered load per poll — a cache invalidate each.)
    // A timeout is sticky: once this rank has given up on an exchange every later wait fails at once (the host sees the status
    // word and raises) instead of spending the full bounded wait per launch.
    const int r = (int)threadIdx.x;
    const bool mine = r < P.world;
    bool done = !mine;
    int ok = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u;
    if (ok) {
      const unsigned int *flag = P.flags[P.rank] + (mine ? r : 0) * P2P_FLAG_STRIDE;
      for (int spin = 0; spin < P2P_SPIN_MAX; ++spin) {
        if (!done) {
          const unsigned have = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          done = (int)(have - want) >= 0;
        }
        if (__all(done)) break;
        __builtin_amdgcn_s_sleep(4);
      }
      ok = __all(done);
    }
    if (!ok && threadIdx.x == 0) __hip_atomic_store(P.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (threadIdx.x == 0) s_p2p_ok = ok;
  }
  __syncthreads();
  return s_p2p_ok != 0;
}

// sum of the world slots of element i, in rank order
__device__ __forceinline__ float p2p_sum(const P2pDev &P, unsigned epoch, long long i) {
  const float *base = P.slots[P.rank] + (long long)(epoch & 1u) * P.world * P.n_max + i;
  float s = 0.f;
  for (int r = 0; r < P.world; ++r) s += __builtin_nontemporal_load(base + (long long)r * P.n_max);
  return s;
}

// grid_bar.h — the in-launch grid barrier of the persistent kernels (stream_kc.hip: the RF sampler; semdec_persist.hip: a semantic-
// decoder step).  All workgroups of the launch must be RESIDENT (grid <= what the chip holds at once: the hosts check).
//
// Flag barrier.  (A first build counted arrivals with one agent-scope atomic word: 256 read-modify-writes of ONE address are served
// one after the other at the memory side — 18 us per barrier, profiles/r05_rf_persist_ab.txt.)  Every workgroup owns one word of a
// 1 KiB flag array — word (wg % 8) * 32 + wg / 8: the workgroups of one XCD share a 128-byte line — and stores the barrier's epoch into
// it (plain stores to distinct words pipeline); wave 0 of every workgroup polls the whole array, 4 words per lane, until every word
// has reached the epoch.  Epochs only grow: the host zeroes the array once per call and hands each launch its first epoch.
// Data handed between workgroups must be stored write-through and loaded past the L1 (`sc1`: AUX_SC1 buffer accesses, agent-scope
// atomics) — the barrier contains NO release / acquire fence (an L2 write-back + L1 invalidate per barrier measured 6 us:
// tools/exp/gridbar_bench.hip).  The wait is bounded (wall clock): on expiry word 256 of the array is set, the caller's registered
// status word (mn_persist_set_status_word: sticky across calls, read by the host at its next sync) is raised and `dead` stays up.
// A workgroup that times out ALONE runs on without waits over stale data while the others see a complete barrier: after its last wait
// every workgroup therefore calls poisoned(), which ORs the error word (agent-scope load) into its own flag — the result is NaN in
// every workgroup, never a silently wrong number.
#pragma once
#include "common.h"

constexpr int MN_GRID_BAR_WORDS = 320;               // 256 flags + the error word, padded

struct GridBar {
  unsigned* flags; unsigned G, epoch; uint64_t ticks; int dead; unsigned* status;
  __device__ __forceinline__ void arrive() {
    ++epoch;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {                          // (payload stores were write-through and are drained: no release fence)
      const unsigned wg = blockIdx.x;
      __hip_atomic_store(flags + ((wg & 7u) * 32u + (wg >> 3)), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __device__ __forceinline__ void wait() {
    if (threadIdx.x < 64 && !dead) {
      const unsigned lane = threadIdx.x;
      const uint64_t t0 = wall_clock64();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned i = lane + 64u * j, wg = (i & 31u) * 8u + (i >> 5);
          const unsigned v = __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (wg >= G || (int32_t)(v - epoch) >= 0);
        }
        if (__all(ok)) break;
        if (wall_clock64() - t0 > ticks) {
          if (lane == 0) {
            atomicExch(flags + 256, 0x300u);
            if (status) atomicOr(status, 0x300u);
          }
          dead = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();                                // (the phases read other workgroups' data past the L1: no acquire fence)
  }
  // After the launch's LAST wait: has this or ANY workgroup given up?  (A lone straggler raised the error word before its later
  // arrivals — its stores are drained by arrive()'s vmcnt(0) — so whoever passed the last barrier sees it.)  Workgroup-uniform.
  __device__ __forceinline__ int poisoned() {
    unsigned e = 0;
    if (threadIdx.x == 0) e = __hip_atomic_load(flags + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __syncthreads_or((threadIdx.x < 64 && dead) || e != 0);
  }
};

// The caller's sticky status word (device memory, or nullptr): mn_persist_set_status_word, capi / stream_kc.hip
unsigned* mn_persist_status_word();

// Host side (stream_kc.hip): two persistent launches must never share the device from different streams — each could hold CUs the
// other waits for — so they are ordered by an event.  Call before the launch (may enqueue a wait on `st`) and after it (records).
int mn_persist_order_before(hipStream_t st);
void mn_persist_order_after(hipStream_t st);

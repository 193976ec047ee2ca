// Micro-benchmark of in-kernel grid barriers on one MI355X (256 workgroups x 512 threads, one per CU): us per barrier for
//   mode 0  one atomic counter + generation word (sense reversal)
//   mode 1  one flag per workgroup, every workgroup's wave 0 polls all flags
//   mode 2  one flag per workgroup, workgroup 0 polls them and publishes a generation word per XCD line; the others poll their line
// each with / without the agent-scope release (L2 write-back) and acquire (L2 invalidate) fences, and over the poll sleep.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/gridbar_bench.hip -o tools/exp/bin/gridbar_bench && tools/exp/bin/gridbar_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE, bool REL, bool ACQ, int SLEEP>
__global__ __launch_bounds__(512) void bar_kernel(unsigned* w, int iters, float* sink) {
  const unsigned G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  unsigned gen = 0, epoch = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // a little "work" that writes memory, like a phase's epilogue
    if (tid < 64) sink[wg * 64 + tid] = acc + it;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (MODE == 0) {
      if (tid == 0) {
        if (REL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G - 1) {
          __hip_atomic_store(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(w + 64, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(w + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(SLEEP);
        if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      ++gen;
    } else if (MODE == 1) {
      if (tid == 0) {
        if (REL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(w + ((wg & 7u) * 32u + (wg >> 3)), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid < 64) {
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned i = tid + 64u * j, g2 = (i & 31u) * 8u + (i >> 5);
            const unsigned v = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (g2 >= G || (int32_t)(v - epoch) >= 0);
          }
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(SLEEP);
        }
        if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    } else {
      if (tid == 0) {
        if (REL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(w + ((wg & 7u) * 32u + (wg >> 3)), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (wg == 0 && tid < 64) {
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned i = tid + 64u * j, g2 = (i & 31u) * 8u + (i >> 5);
            const unsigned v = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (g2 >= G || (int32_t)(v - epoch) >= 0);
          }
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(SLEEP);
        }
        if (tid < 8) __hip_atomic_store(w + 512 + tid * 32, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) {
        while ((int32_t)(__hip_atomic_load(w + 512 + (wg & 7u) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) __builtin_amdgcn_s_sleep(SLEEP);
        if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    acc += sink[((wg + 1) % G) * 64 + (tid & 63)];        // read a neighbour's value after the barrier
  }
  if (tid == 0) sink[G * 64 + wg] = acc;
}

template <int MODE, bool REL, bool ACQ, int SLEEP>
void run(const char* name, unsigned* w, float* sink, int G) {
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(w, 0, 4096 * 4);
    hipEventRecord(a);
    hipLaunchKernelGGL((bar_kernel<MODE, REL, ACQ, SLEEP>), dim3(G), dim3(512), 100 * 1024, 0, w, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (rep) printf("%-60s G=%3d  %7.2f us per barrier\n", name, G, ms * 1e3f / iters);
  }
}

int main() {
  unsigned* w; float* sink;
  hipMalloc(&w, 4096 * 4); hipMalloc(&sink, 1 << 20);
  hipMemset(sink, 0, 1 << 20);
#define OPT(K) hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)
#define RUN(M, R, A, S, NAME) OPT((bar_kernel<M, R, A, S>)); run<M, R, A, S>(NAME, w, sink, 256)
  RUN(0, true, true, 2, "counter + generation, release + acquire, sleep 2");
  RUN(0, false, false, 2, "counter + generation, no fences, sleep 2");
  RUN(1, true, true, 1, "flags, all poll, release + acquire, sleep 1");
  RUN(1, false, false, 1, "flags, all poll, no fences, sleep 1");
  RUN(1, true, false, 1, "flags, all poll, release only, sleep 1");
  RUN(1, false, true, 1, "flags, all poll, acquire only, sleep 1");
  RUN(1, false, false, 8, "flags, all poll, no fences, sleep 8");
  RUN(1, false, false, 32, "flags, all poll, no fences, sleep 32");
  RUN(1, true, true, 8, "flags, all poll, release + acquire, sleep 8");
  RUN(2, false, false, 1, "flags, master polls + 8 generation lines, no fences, sleep 1");
  RUN(2, true, true, 1, "flags, master polls + 8 generation lines, release + acquire, sleep 1");
  RUN(2, false, false, 8, "flags, master polls + 8 generation lines, no fences, sleep 8");
  return 0;
}

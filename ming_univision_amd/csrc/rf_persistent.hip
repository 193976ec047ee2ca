// rf_persistent.hip — one persistent launch for all residual blocks of one Euler step of the RF head.
//
// A ResBlock (diff_loss_rf_swiglu.py:268-272) is two dependent weight-streaming GEMVs:
//   A: hid = SwiGLU(W12 · modulate(LN(h)) + b12)        100.7 MB of bf16 weights at w = 3072
//   B: h  += gate * (W3 · hid + b3)                       50.3 MB
// As separate launches each pays ≈ 6 µs of launch / first-load latency / tail on top of ≈ 15 / 7 µs of
// streaming.  Here 256 workgroups (one per CU, 12 waves) stay resident for all `depth`
// blocks; between the phases they meet at a grid barrier, and — the point of the exercise — every wave
// requests the first RING chunks of its NEXT phase's weight rows before it arrives at the barrier, so the
// HBM pipes keep streaming while the barrier and the next phase's activation staging run.
//
// Inter-workgroup visibility follows the CDNA4 recipe: every wave drains its stores (vmcnt(0)), the
// workgroup barriers, lane 0 issues an agent-scope release fence (+ asm vmcnt(0)) and a relaxed
// agent-scope fetch_add on a monotonic counter; it polls with relaxed loads + s_sleep, then one agent-scope
// acquire fence, then the workgroup barriers again and stages the activations with plain loads.  Spins are
// bounded: on timeout an error word is set and the workgroup proceeds (results garbage, no hang).
#include "skinny_device.h"

namespace {

constexpr int PNT = 512;      // 8 waves per CU (2 per SIMD): each wave may use up to 256 VGPRs for the two weight rings
constexpr int PRING = 4;
constexpr int MAX_DEPTH = 16;

struct RfPersistArgs {
  int32_t w, hidden, depth, A;
  const bf16_t* w12[MAX_DEPTH]; const bf16_t* b12[MAX_DEPTH];
  const bf16_t* w3[MAX_DEPTH];  const bf16_t* b3[MAX_DEPTH];
  const bf16_t* ln_g[MAX_DEPTH]; const bf16_t* ln_b[MAX_DEPTH];
  const float* ada;   // [M, A] modulations of this step: per block (shift, scale, gate) x w
  float* h;           // [M, w] in/out
  float* hid;         // [M, hidden] scratch
  unsigned* bar;      // [0] arrival counter (zeroed before the launch), [1] error word
};

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) {   // ~ seconds: something is wrong (workgroup not resident?)
        __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int M>
__global__ __launch_bounds__(PNT) void rf_blocks_persistent_kernel(const RfPersistArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * (PNT / 64);
  const int w = p.w, hidden = p.hidden;
  const int ncA = (w + 511) >> 9, ncB = (hidden + 511) >> 9;       // chunks per weight row
  const int KpA = ncA << 9, KpB = ncB << 9;
  float* xs = smem;
  float* red = smem + (int64_t)M * (KpA > KpB ? KpA : KpB);
  const int wid = blockIdx.x * (PNT / 64) + wave;

  u32x4 ringA[PRING][2], ringB[PRING];
  // phase A: output row g (0..hidden-1) needs rows g and g + hidden of W12; phase B: row g of W3
  auto issueA = [&](int blk, int g, int ct, u32x4 (&dst)[2]) {
    const int k = min(ct * 512 + lane * 8, w - 8);
    const bf16_t* wp = p.w12[blk] + k;
    const int n = min(g, hidden - 1);
    dst[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (int64_t)n * w));
    dst[1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (int64_t)(n + hidden) * w));
  };
  auto issueB = [&](int blk, int g, int ct, u32x4& dst) {
    const int k = min(ct * 512 + lane * 8, hidden - 8);
    const int n = min(g, w - 1);
    dst = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.w3[blk] + (int64_t)n * hidden + k));
  };
  auto headA = [&](int blk, int g) {
#pragma unroll
    for (int d = 0; d < PRING; ++d)
      if (d < ncA) issueA(blk, g, d, ringA[d]);
  };
  auto headB = [&](int blk, int g) {
#pragma unroll
    for (int d = 0; d < PRING; ++d)
      if (d < ncB) issueB(blk, g, d, ringB[d]);
  };

  KArgs ka;   // staging descriptor, rebuilt per phase (wave-uniform)
  ka.nseg = 1; ka.batch = 1; ka.inv_nchunk = 0;
  ka.a.seg_scale = nullptr; ka.a.x_batch_div = 1; ka.a.x_batch_stride = 0; ka.a.eps = 1e-6f;

  if (wid < hidden) headA(0, wid);
  const float* xl = xs + lane * 4;
  unsigned epoch = 0;
  for (int blk = 0; blk < p.depth; ++blk) {
    const float* mod = p.ada + (int64_t)blk * 3 * w;
    // ------------------------------------------------------------------ phase A
    ka.a.x = p.h; ka.a.ldx = w; ka.a.K = w; ka.nchunk = ncA;
    ka.a.prologue = MN_PRO_LN_MOD; ka.a.ln_g = p.ln_g[blk]; ka.a.ln_b = p.ln_b[blk];
    ka.a.pro_a = mod; ka.a.ld_pro_a = p.A; ka.a.pro_b = mod + w; ka.a.ld_pro_b = p.A;
    stage_x<M, PNT>(ka, xs, red, 0);
    __syncthreads();
    for (int g = wid; g < hidden; g += nwaves) {
      float acc[2][M];
#pragma unroll
      for (int m = 0; m < M; ++m) { acc[0][m] = 0.f; acc[1][m] = 0.f; }
      int c0 = 0;
      for (; c0 + 2 * PRING <= ncA; c0 += PRING) {
#pragma unroll
        for (int d = 0; d < PRING; ++d) {
          fma_chunk<M>(ringA[d][0], xl + (c0 + d) * 512, KpA, acc[0]);
          fma_chunk<M>(ringA[d][1], xl + (c0 + d) * 512, KpA, acc[1]);
          issueA(blk, g, c0 + d + PRING, ringA[d]);
        }
      }
#pragma unroll
      for (int d = 0; d < PRING; ++d) {
        if (c0 + d < ncA) {
          fma_chunk<M>(ringA[d][0], xl + (c0 + d) * 512, KpA, acc[0]);
          fma_chunk<M>(ringA[d][1], xl + (c0 + d) * 512, KpA, acc[1]);
          if (c0 + d + PRING < ncA) issueA(blk, g, c0 + d + PRING, ringA[d]);
        }
      }
      c0 += PRING;
#pragma unroll
      for (int d = 0; d < PRING; ++d)
        if (c0 + d < ncA) {
          fma_chunk<M>(ringA[d][0], xl + (c0 + d) * 512, KpA, acc[0]);
          fma_chunk<M>(ringA[d][1], xl + (c0 + d) * 512, KpA, acc[1]);
        }
      if (g + nwaves < hidden) headA(blk, g + nwaves);
      else if (wid < w) headB(blk, wid);           // last group of this phase: start streaming phase B's rows
#pragma unroll
      for (int m = 0; m < M; ++m) { acc[0][m] = wave_sum(acc[0][m]); acc[1][m] = wave_sum(acc[1][m]); }
#pragma unroll
      for (int m = 0; m < M; ++m)
        if (lane == m) {
          const float y1 = acc[0][m] + bf16_to_f32(p.b12[blk][g]);
          const float y2 = acc[1][m] + bf16_to_f32(p.b12[blk][g + hidden]);
          p.hid[(int64_t)m * hidden + g] = silu_f(y1) * y2;
        }
    }
    if (wid >= hidden && wid < w) headB(blk, wid);   // waves without phase-A work still prefetch phase B
    grid_barrier(p.bar, (++epoch) * gridDim.x);
    // ------------------------------------------------------------------ phase B
    ka.a.x = p.hid; ka.a.ldx = hidden; ka.a.K = hidden; ka.nchunk = ncB;
    ka.a.prologue = MN_PRO_NONE; ka.a.ln_g = nullptr; ka.a.ln_b = nullptr; ka.a.pro_a = nullptr; ka.a.pro_b = nullptr;
    stage_x<M, PNT>(ka, xs, red, 0);
    __syncthreads();
    const bool more = blk + 1 < p.depth;
    for (int g = wid; g < w; g += nwaves) {
      float acc[M];
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = 0.f;
      int c0 = 0;
      for (; c0 + 2 * PRING <= ncB; c0 += PRING) {
#pragma unroll
        for (int d = 0; d < PRING; ++d) {
          fma_chunk<M>(ringB[d], xl + (c0 + d) * 512, KpB, acc);
          issueB(blk, g, c0 + d + PRING, ringB[d]);
        }
      }
#pragma unroll
      for (int d = 0; d < PRING; ++d) {
        if (c0 + d < ncB) {
          fma_chunk<M>(ringB[d], xl + (c0 + d) * 512, KpB, acc);
          if (c0 + d + PRING < ncB) issueB(blk, g, c0 + d + PRING, ringB[d]);
        }
      }
      c0 += PRING;
#pragma unroll
      for (int d = 0; d < PRING; ++d)
        if (c0 + d < ncB) fma_chunk<M>(ringB[d], xl + (c0 + d) * 512, KpB, acc);
      if (g + nwaves < w) headB(blk, g + nwaves);
      else if (more && wid < hidden) headA(blk + 1, wid);
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = wave_sum(acc[m]);
#pragma unroll
      for (int m = 0; m < M; ++m)
        if (lane == m) {
          const float y = acc[m] + bf16_to_f32(p.b3[blk][g]);
          float* hp = p.h + (int64_t)m * w + g;
          *hp = *hp + mod[(int64_t)m * p.A + 2 * w + g] * y;
        }
    }
    if (more && wid >= w && wid < hidden) headA(blk + 1, wid);
    if (more) grid_barrier(p.bar, (++epoch) * gridDim.x);
  }
}

template <int M>
int launch_persist(const RfPersistArgs& p, int cus, size_t lds, hipStream_t st) {
  static bool opted = false;
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rf_blocks_persistent_kernel<M>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    opted = true;
  }
  hipLaunchKernelGGL((rf_blocks_persistent_kernel<M>), dim3(cus), dim3(PNT), lds, st, p);
  return 0;
}

}  // namespace

// Runs all `depth` residual blocks of one Euler step. bar: 2 x uint32 device words (zeroed here, stream-ordered).
// Internal entry point used by mn_rf_sample (engine.hip); returns MN_EINVAL when the shape is not supported
// (caller falls back to per-GEMV launches).
extern "C" int mn_rf_blocks_persistent(int rows, int w, int hidden, int depth, int A, const uint16_t* const* w12,
                                       const uint16_t* const* b12, const uint16_t* const* w3,
                                       const uint16_t* const* b3, const uint16_t* const* ln_g,
                                       const uint16_t* const* ln_b, const float* ada, float* h, float* hid,
                                       unsigned* bar, void* stream) {
  MN_CHECK_ARG(rows >= 1 && rows <= 4 && depth >= 1 && depth <= MAX_DEPTH, "mn_rf_blocks_persistent: rows/depth");
  MN_CHECK_ARG((w % 8) == 0 && (hidden % 8) == 0 && w >= 8 && hidden >= 8, "mn_rf_blocks_persistent: w/hidden %% 8");
  RfPersistArgs p;
  p.w = w; p.hidden = hidden; p.depth = depth; p.A = A;
  for (int i = 0; i < depth; ++i) {
    p.w12[i] = w12[i]; p.b12[i] = b12[i]; p.w3[i] = w3[i]; p.b3[i] = b3[i]; p.ln_g[i] = ln_g[i]; p.ln_b[i] = ln_b[i];
    MN_CHECK_ARG(b12[i] && b3[i] && ln_g[i] && ln_b[i], "mn_rf_blocks_persistent: null bias / LN");
  }
  p.ada = ada; p.h = h; p.hid = hid; p.bar = bar;
  const int kmax = w > hidden ? w : hidden;
  const size_t lds = ((size_t)rows * (((kmax + 511) / 512) * 512) + 64) * sizeof(float);
  MN_CHECK_ARG(lds <= 160 * 1024, "mn_rf_blocks_persistent: rows x K too large for LDS");
  hipStream_t st = mn_stream(stream);
  const int cus = mn_num_cus();
  if (hipMemsetAsync(bar, 0, 2 * sizeof(unsigned), st) != hipSuccess) { mn_set_error("memset failed"); return MN_ELAUNCH; }
  switch (rows) {
    case 1: launch_persist<1>(p, cus, lds, st); break;
    case 2: launch_persist<2>(p, cus, lds, st); break;
    case 3: launch_persist<3>(p, cus, lds, st); break;
    default: launch_persist<4>(p, cus, lds, st); break;
  }
  MN_CHECK_LAUNCH("mn_rf_blocks_persistent");
  return MN_OK;
}

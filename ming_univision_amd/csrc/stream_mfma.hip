// stream_mfma.hip — weight-streaming GEMM for 3..16 activation rows on the matrix cores (gfx950).
//
//   partial[z][m][n] = sum_{k in slice z} (x_hi[m,k] + x_lo[m,k]) * W[n,k]
//
// Same roofline as the fp32 skinny kernel (HBM: every weight byte is read once), but the multiply runs
// on v_mfma_f32_16x16x32_bf16 so that up to 16 rows cost the same as one.  The activations arrive
// pre-split into bf16 hi + lo halves (x = hi + lo to 2^-17, so products stay fp32-accurate); a
// workgroup copies its K-slice of both halves into LDS once and every wave then streams weight rows
// STRAIGHT from HBM into MFMA B-fragments — a weight element is used exactly once per workgroup, so
// staging it through LDS would be pure overhead:
//
//   B fragment (W):  lane l holds W[n0 + (l & 15)][k + (l >> 4) * 8 .. +8]   one 16-byte nt load
//   A fragment (x):  lane l holds x[m = l & 15][k + (l >> 4) * 8 .. +8]      ds_read_b128 (hi) + (lo)
//   D:               lane l, reg r holds out[m = (l >> 4) * 4 + r][n0 + (l & 15)]
//
// A wave owns a tile of 16 output rows for the workgroup's K-slice (KS = 1024: 32 MFMA steps, a
// 4-deep ring of 8 loads each keeps 32 KiB per wave in flight at 8 waves per CU), then moves on to its next tile.
// Slices are reduced (with bias / activation / residual epilogues) by medium_epilogue_kernel.
#include <stdlib.h>

#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SKS = 1024;            // K-slice per workgroup
constexpr int SROW = SKS + 8;        // LDS row stride in bf16 elements (+16 B: conflict-free b128 fragment reads)
constexpr int SNT = 512;             // 8 waves

// Y: [2][M][K] bf16 (hi rows then lo rows, row stride K).  P: [nz][M][Ntot] fp32.
template <bool NTLOAD>
__device__ __forceinline__ u32x4 ldw(const bf16_t* p) {
  if (NTLOAD) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return *reinterpret_cast<const u32x4*>(p);
}

template <bool NTLOAD>
__global__ __launch_bounds__(SNT) void stream_mfma_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ W,
                                                          float* __restrict__ P, int M, int Ntot, int K) {
  extern __shared__ __attribute__((aligned(16))) bf16_t xs_raw[];      // [2][16][SROW] = 66 KiB
  bf16_t (*xs)[16][SROW] = reinterpret_cast<bf16_t (*)[16][SROW]>(xs_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.y;
  const int k0 = z * SKS, klen = min(SKS, K - k0);          // K % 8 == 0; klen % 8 == 0
  const int fr = lane & 15, fq = lane >> 4;
  const int ntiles = (Ntot + 15) >> 4;
  const int nsteps = (klen + 31) >> 5;                       // MFMA steps of 32 k
  const int twaves = gridDim.x * (SNT / 64);
  const int kcl = klen - 8 - fq * 8;                         // last valid 8-element offset for this lane group
  u32x4 ring[8];
  auto head = [&](int t) {                                   // request the first 8 weight fragments of tile t
    const int n = min(t * 16 + fr, Ntot - 1);
    const bf16_t* wp = W + (int64_t)n * K + k0 + fq * 8;
#pragma unroll
    for (int d = 0; d < 8; ++d) ring[d] = ldw<NTLOAD>(wp + min(d * 32, kcl));
  };
  int t = blockIdx.x * (SNT / 64) + wave;
  if (t < ntiles) head(t);                                   // weights first: their HBM latency overlaps the x staging
  // ---- stage this slice of x (rows >= M are zero so that unused MFMA rows contribute nothing)
  for (int i = tid; i < 2 * 16 * (SKS / 8); i += SNT) {
    const int slot = i % (SKS / 8), m = (i / (SKS / 8)) % 16, h = i / (16 * (SKS / 8));
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m < M && slot * 8 < klen) v = *reinterpret_cast<const u32x4*>(Y + ((int64_t)h * M + m) * K + k0 + slot * 8);
    *reinterpret_cast<u32x4*>(&xs[h][m][slot * 8]) = v;
  }
  const bf16_t* xh = &xs[0][fr][fq * 8];
  const bf16_t* xl = &xs[1][fr][fq * 8];
  __syncthreads();
  for (; t < ntiles; t += twaves) {
    const int n = min(t * 16 + fr, Ntot - 1);
    const bf16_t* wp = W + (int64_t)n * K + k0 + fq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 16 <= nsteps; s += 8) {
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const bf16x8 w = __builtin_bit_cast(bf16x8, ring[d]);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xh + (s + d) * 32);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(xl + (s + d) * 32);
        ring[d] = ldw<NTLOAD>(wp + min((s + d + 8) * 32, kcl));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w, acc, 0, 0, 0);
      }
    }
    // drain (steps beyond nsteps multiply the zero-padded x tail or re-read clamped in-bounds weights
    // against x columns >= klen, which are zero in LDS)
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      if (s + d < nsteps) {
        const bf16x8 w = __builtin_bit_cast(bf16x8, ring[d]);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xh + (s + d) * 32);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(xl + (s + d) * 32);
        if (s + d + 8 < nsteps) ring[d] = ldw<NTLOAD>(wp + min((s + d + 8) * 32, kcl));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w, acc, 0, 0, 0);
      }
    }
    s += 8;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      if (s + d < nsteps) {
        const bf16x8 w = __builtin_bit_cast(bf16x8, ring[d]);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xh + (s + d) * 32);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(xl + (s + d) * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w, acc, 0, 0, 0);
      }
    }
    if (t + twaves < ntiles) head(t + twaves);               // next tile's head overlaps the stores below
    // D layout: row m = fq*4 + r, col n = t*16 + fr
    const int nn = t * 16 + fr;
    if (nn < Ntot) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = fq * 4 + r;
        if (m < M) P[((int64_t)z * M + m) * Ntot + nn] = acc[r];
      }
    }
  }
}


// ---- variant 2: weights staged through a wave-private LDS tile with fully coalesced global loads ----
// The direct B-fragment load above touches 16 rows x 64 B per instruction (two instructions per 128-B line);
// here a wave loads its 16-row x 256-k chunk as 8 instructions of 8 rows x 128 B (whole lines, like the
// fp32 skinny kernel), parks it in its own 8 KiB of LDS (XOR-swizzled 16-byte slots, wave-private so no
// workgroup barrier is involved) and reads MFMA fragments back with ds_read_b128.  One chunk (8 KiB per
// wave) is in flight in registers while the previous one is multiplied.
constexpr int WCH = 256;                       // k per chunk
__device__ __forceinline__ int wslot(int row, int slot) { return row * (WCH * 2) + (((slot) ^ (row & 15)) << 4); }

__global__ __launch_bounds__(SNT) void stream_mfma_lds_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ W,
                                                              float* __restrict__ P, int M, int Ntot, int K) {
  extern __shared__ __attribute__((aligned(16))) bf16_t xs_raw[];      // [2][16][SROW] x image, then 8 x 8 KiB weight tiles
  bf16_t (*xs)[16][SROW] = reinterpret_cast<bf16_t (*)[16][SROW]>(xs_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* wbuf = reinterpret_cast<char*>(xs_raw) + (size_t)2 * 16 * SROW * sizeof(bf16_t) + (size_t)wave * 16 * WCH * 2;
  const int z = blockIdx.y;
  const int k0 = z * SKS, klen = min(SKS, K - k0);
  const int fr = lane & 15, fq = lane >> 4;
  const int r8 = lane >> 3, c8 = lane & 7;
  const int ntiles = (Ntot + 15) >> 4;
  const int nch = (klen + WCH - 1) / WCH;
  const int twaves = gridDim.x * (SNT / 64);
  u32x4 ring[8];
  // instruction i: rows (i & 1) * 8 + r8, 16-byte slot (i >> 1) * 8 + c8 of the chunk
  auto issue = [&](int t, int ch) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (i & 1) * 8 + r8;
      const int n = min(t * 16 + row, Ntot - 1);
      const int k = min(ch * WCH + ((i >> 1) * 8 + c8) * 8, klen - 8);
      ring[i] = *reinterpret_cast<const u32x4*>(W + (int64_t)n * K + k0 + k);
    }
  };
  int t = blockIdx.x * (SNT / 64) + wave;
  if (t < ntiles) issue(t, 0);
  for (int i = tid; i < 2 * 16 * (SKS / 8); i += SNT) {
    const int slot = i % (SKS / 8), m = (i / (SKS / 8)) % 16, h = i / (16 * (SKS / 8));
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m < M && slot * 8 < klen) v = *reinterpret_cast<const u32x4*>(Y + ((int64_t)h * M + m) * K + k0 + slot * 8);
    *reinterpret_cast<u32x4*>(&xs[h][m][slot * 8]) = v;
  }
  const bf16_t* xh = &xs[0][fr][fq * 8];
  const bf16_t* xl = &xs[1][fr][fq * 8];
  __syncthreads();
  for (; t < ntiles; t += twaves) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nch; ++ch) {
      // park the landed chunk in LDS ...
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<u32x4*>(wbuf + wslot((i & 1) * 8 + r8, (i >> 1) * 8 + c8)) = ring[i];
      // ... request the next one (next chunk of this tile, or the first chunk of the wave's next tile) ...
      if (ch + 1 < nch) issue(t, ch + 1);
      else if (t + twaves < ntiles) issue(t + twaves, 0);
      // ... and multiply this one: 8 MFMA steps of 32 k (columns >= klen of x are zero in LDS)
#pragma unroll
      for (int sstep = 0; sstep < 8; ++sstep) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wbuf + wslot(fr, sstep * 4 + fq));
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xh + ch * WCH + sstep * 32);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(xl + ch * WCH + sstep * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w, acc, 0, 0, 0);
      }
    }
    const int nn = t * 16 + fr;
    if (nn < Ntot) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = fq * 4 + r;
        if (m < M) P[((int64_t)z * M + m) * Ntot + nn] = acc[r];
      }
    }
  }
}

}  // namespace

// Tuning knobs for in-process A/B (not part of the stable ABI): mode 0 = direct fragment loads, 1 = LDS-staged.
static int g_mode = -1, g_nt = -1, g_cap = -1;
extern "C" void mn_stream_tune(int mode, int nt, int cap) { g_mode = mode; g_nt = nt; g_cap = cap; }

// Internal: returns the number of K slices written (partials [nz][M][Ntot]).
extern "C" int mn_stream_mfma(const uint16_t* Y, const uint16_t* W, float* P, int M, int Ntot, int K, void* stream) {
  MN_CHECK_ARG(Y && W && P && M >= 1 && M <= 16 && Ntot >= 1 && K >= 8 && (K % 8) == 0, "mn_stream_mfma: bad args");
  const int nz = (K + SKS - 1) / SKS;
  const int ntiles = (Ntot + 15) / 16;
  const int cus = mn_num_cus();
  // 2 workgroups (66 KiB LDS each) per CU; every wave should get >= 1 tile
  int gx = (int)mn_cdiv(ntiles, SNT / 64);
  const int cap_x = g_cap > 0 ? g_cap : 4;
  const int use_nt = g_nt >= 0 ? g_nt : 0;
  const int cap = (int)mn_cdiv((int64_t)cap_x * cus, nz * 2);   // cap_x / 2 workgroups per CU
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  static bool opted = false;
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    opted = true;
  }
  const int mode = g_mode >= 0 ? g_mode : 1;   // default: LDS-staged coalesced loads (12.9 vs 14.1 us on RF w3, equal on w12)
  if (mode == 1) {   // LDS-staged, fully coalesced weight loads; one workgroup per CU
    static bool opted2 = false;
    if (!opted2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_mfma_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      opted2 = true;
    }
    int gx2 = (int)mn_cdiv(ntiles, SNT / 64);
    const int cap2 = (int)mn_cdiv(cus, nz);
    if (gx2 > cap2) gx2 = cap2;
    if (gx2 < 1) gx2 = 1;
    const size_t lds2 = (size_t)2 * 16 * SROW * sizeof(bf16_t) + (size_t)(SNT / 64) * 16 * WCH * 2;
    hipLaunchKernelGGL(stream_mfma_lds_kernel, dim3(gx2, nz), dim3(SNT), lds2, mn_stream(stream), Y, W, P, M, Ntot, K);
    MN_CHECK_LAUNCH("mn_stream_mfma");
    return nz;
  }
  const size_t lds = (size_t)2 * 16 * SROW * sizeof(bf16_t);
  if (use_nt)
    hipLaunchKernelGGL(stream_mfma_kernel<true>, dim3(gx, nz), dim3(SNT), lds, mn_stream(stream), Y, W, P, M, Ntot, K);
  else
    hipLaunchKernelGGL(stream_mfma_kernel<false>, dim3(gx, nz), dim3(SNT), lds, mn_stream(stream), Y, W, P, M, Ntot, K);
  MN_CHECK_LAUNCH("mn_stream_mfma");
  return nz;
}

extern "C" int mn_stream_mfma_slices(int K) { return (K + SKS - 1) / SKS; }

// stream_mfma.hip — weight-streaming GEMM for 5..32 activation rows on the matrix cores (gfx950); 33..64 rows are
// forwarded to the K-loop form in stream_kloop.hip.
//
//   partial[z][row][n] = sum_{k in slice z} (x_hi[xrow,k] + x_lo[xrow,k]) * W_g[n,k]
//
// Same roofline as the fp32 skinny kernel (HBM: every weight byte is read once), but the multiply runs
// on v_mfma_f32_16x16x32_bf16 so that up to 32 rows cost the same as one.  The activations arrive
// pre-split into bf16 hi + lo halves (x = hi + lo to 2^-17, so products stay fp32-accurate).
//
// A workgroup owns one K-slice (`ks` = 1..4 chunks of 256 k) of one weight matrix: it copies that slice of both
// x halves into LDS once, and each of its waves then streams 16-row weight tiles:
//   * a wave loads its 16-row x 256-k chunk as 8 instructions of 8 rows x 128 B (whole cache lines), parks
//     it in its own 8 KiB of LDS (XOR-swizzled 16-byte slots; wave-private, so no workgroup barrier is
//     involved) and reads MFMA B fragments back with ds_read_b128; the next chunk is in flight in registers
//     while this one is multiplied (deeper rings, parking chunk 0 before the x barrier and nontemporal partial
//     stores all measured no faster;
//     B fragments loaded straight from HBM, 16 rows x 64 B per instruction, also slower: 14.2 vs 12.9 us on RF w3);
//   * A fragments (x): lane l holds x[m = l & 15][k + (l >> 4) * 8 .. +8], ds_read_b128 of hi and of lo;
//   * D: lane l, reg r holds out[m = (l >> 4) * 4 + r][n0 + (l & 15)].
// K-slice partials are reduced (with bias / activation / residual epilogues) by the caller's next kernel.
//
// The grouped form (MoE experts) runs on the K-loop kernel (stream_kloop.hip) at every row count.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "stream_fuse.h"
#include "w8_codec.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// K-loop form for 33..64 rows (stream_kloop.hip)
extern "C" int mn_stream_kloop(const uint16_t* Y, const uint16_t* W, float* P, int M, int Ntot, int K, void* stream);
extern "C" int mn_stream_kloop_slices(int M, int Ntot, int K);
extern "C" int mn_stream_kloop_grouped(const uint16_t* Y, int y_rows, const uint16_t* W, int64_t w_stride, float* P,
                                       int p_rows, const int32_t* off, const int32_t* xrows, int G, int max_rows, int row_lo,
                                       int nz, int Ntot, int K, void* stream);
// ... and on fp8 weights (e4m3 bytes + one fp32 scale per output row)
extern "C" int mn_stream_kloop_wq(const uint16_t* Y, const uint8_t* Wq, const float* wscale, float* P, int M, int Ntot, int K, int wfmt, void* stream);
extern "C" int mn_stream_kloop_wq_slices(int wfmt, int M, int Ntot, int K);
extern "C" int mn_stream_kloop_grouped_wq(const uint16_t* Y, int y_rows, const uint8_t* Wq, int64_t w_stride, const float* wscale,
                                          int64_t s_stride, float* P, int p_rows, const int32_t* off, const int32_t* xrows, int G,
                                          int max_rows, int nz, int Ntot, int K, int wfmt, void* stream);

namespace {

constexpr int WCH = 256;             // k per chunk
constexpr int MAX_KCH = 4;           // K-slice <= 1024
constexpr size_t LDS_CAP = 160 * 1024;

__device__ __forceinline__ int wslot(int row, int slot) { return row * (WCH * 2) + (((slot) ^ (row & 15)) << 4); }
// NF4 chunks are parked by eight lanes per row that own FOUR consecutive slots each (32 k = 16 bytes of codes per lane): with the
// layout above the eight lanes of a write group would hit two bank groups.  The 4 x 8 slot grid of a row is transposed instead —
// slot 4c + j lives at c + 8j — and XOR-ed with row & 7 inside the low three bits: the parked slots of a step (same j, c = 0..7) and
// the fragment reads of a step (same slot, rows 0..7 | 8..15) both cover eight distinct 16-byte bank groups.
__device__ __forceinline__ int wslot4(int row, int slot) { return row * (WCH * 2) + (((((slot >> 2) | ((slot & 3) << 3))) ^ (row & 7)) << 4); }
// x image: element offset of 16-byte slot `slot` of row `row` (row stride = ks elements, a multiple of 256)
__device__ __forceinline__ int xslot(int row, int srow, int slot) { return row * srow + ((slot ^ (row & 15)) << 3); }

// ---- fused form for the RF ResBlock chain at <= 4 rows (the reference's call shape: the CFG rows of ONE image) ----
// At 2-3 rows a ResBlock was four launches — w12, [slab sum + SwiGLU + split], w3, [slab sum + gated residual + LayerNorm-modulate +
// split] — and each glue launch costs 5-6 us of a 40 us block although it moves a few KiB.  FUSE_SWIGLU folds the first one into
// the w3 launch: the activation image a workgroup parks in LDS is COMPUTED from the w12 launch's slabs instead of copied,
//   x[m, k] = silu(b[k] + sum_z Pp[z][m][k]) * (b[K + k] + sum_z Pp[z][m][K + k])
// (every workgroup redoes it for its K-slice: 32 KiB of L2 reads at 2 rows, free next to its 200 KiB of weights — w3 takes 11.2 us
// with or without).  RF sampler 8.35 -> 7.70 ms at 2 rows, 6.32 -> 5.74 in fp8 (profiles/r04_rf_fused_chain_ab.txt).  The same
// file holds the negative result for the second glue launch: LayerNorm needs the whole row, so folding it takes a grid-wide
// hand-off — the last-arriving workgroup of each column group reducing the slabs (write-through stores, one relaxed ticket, sc1
// loads) measured 7.4 us of tail against the 6.0 us launch it replaced; removed again.
constexpr int FUSE_PNZ = 6;         // slabs of the previous launch the prologue sums from registers (one float4 element per thread)

// Y: bf16, hi rows at Y, lo rows at Y + y_lo (row stride K).  P: [nz][p_rows][Ntot] fp32, p_slab = p_rows * Ntot.
// W8: the weights are OCP e4m3 bytes [Ntot][K] with one fp32 scale per output row (W[n,k] = e4m3(Wq[n,k]) * wscale[n]): a chunk
// is 4 KiB of HBM traffic instead of 8, converted to bf16 (exact) in registers on its way into the wave's LDS tile, so the MFMA
// loop is the bf16 one; the row scale multiplies the fp32 accumulators when a tile's partials are stored.
// WQ: 0 = bf16 weights, 1 = one byte per weight (e4m3 | int8: `wf`), 2 = NF4 (two codes per byte, wscale = absmax [Ntot][K / 64]:
// a chunk is 2 KiB of codes, every lane owns 32 k of one row = one 64-element block, decoded through the block's value table —
// w8_codec.h — on its way into the wave's LDS tile; the products need no scale afterwards).
template <int MT, int DEPTH, int MAXT, int WQ, int FUSE = FUSE_NONE>
__global__ __launch_bounds__(MAXT) void stream_mfma_lds_kernel(const bf16_t* __restrict__ Y, int64_t y_lo,
                                                               const void* __restrict__ Wv, const float* __restrict__ wscale,
                                                               float* __restrict__ P, int64_t p_slab, int M, int Ntot, int K, int ks,
                                                               StreamFuse f = StreamFuse{}, int wf = MN_W_FP8_E4M3) {
  constexpr bool W8 = WQ == 1, W4 = WQ == 2;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(Wv);
  const uint8_t* __restrict__ Wq = reinterpret_cast<const uint8_t*>(Wv);
  extern __shared__ __attribute__((aligned(16))) bf16_t xs_raw[];      // [2][16*MT][ks] x image, then nw x 8 KiB weight tiles
  const int row0 = 0, nrows = M;
  constexpr int XR = 16 * MT;
  constexpr int RPW = (2 * XR + 7) / 8;                                  // x rows (hi and lo) per wave at >= 8 waves
  const int srow = ks;                                                  // 16-byte slots XOR-swizzled by row (xslot)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  char* wbuf = reinterpret_cast<char*>(xs_raw) + (size_t)2 * XR * srow * sizeof(bf16_t) + (size_t)wave * 16 * WCH * 2;
  const int z = blockIdx.y;
  const int k0 = z * ks, klen = min(ks, K - k0);
  const int fr = lane & 15, fq = lane >> 4;
  const int r8 = lane >> 3, c8 = lane & 7;
  const int ntiles = (Ntot + 15) >> 4;
  const int nch = (klen + WCH - 1) / WCH;
  const int twaves = gridDim.x * nw;
  const int t0 = blockIdx.x * nw + wave;
  const int mytiles = t0 < ntiles ? (ntiles - t0 + twaves - 1) / twaves : 0;
  const int total = mytiles * nch;                  // chunks this wave streams, tile-major
  // ---- this slice of x goes to registers FIRST: loads retire in order, so the x image can be written to LDS
  // while the (younger) weight loads below are still in flight
  const int slots = ks >> 3;
  const int mtn = (nrows + 15) >> 4;                // 16-row tiles of x actually populated
  const int xr_used = mtn * 16;
  // wave w takes rows w, w + nw, ... of the 2 * xr_used hi/lo rows, a lane the 16-byte slots lane and lane + 64
  // (no integer division: at 768-k slices it cost more than the loads)
  u32x4 xv[RPW][2];
  auto load_x = [&]() {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int rr = wave + j * nw;
      const int h = rr >= xr_used ? 1 : 0, m = rr - h * xr_used;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int slot = lane + u * 64;
        xv[j][u] = u32x4{0u, 0u, 0u, 0u};
        if (rr < 2 * xr_used && m < nrows && slot * 8 < klen) {   // rows >= nrows stay zero: unused MFMA rows contribute nothing
          const int xr = row0 + m;
          xv[j][u] = *reinterpret_cast<const u32x4*>(Y + h * y_lo + (int64_t)xr * K + k0 + slot * 8);
        }
      }
    }
  };
  // fused form: the operands of the x image this workgroup BUILDS go to registers first, like the copied image's
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2v __attribute__((ext_vector_type(2)));
  constexpr int NPZ = FUSE ? FUSE_PNZ : 1;
  f4 pg[NPZ], pu[NPZ];
  u2v bg = {0u, 0u}, bu = {0u, 0u};
  const int kq = ks >> 2;                           // float4 elements per row of the slice; thread tid builds element (fm, fq4)
  const int fm = FUSE ? tid / kq : 0, fq4 = tid - fm * kq;
  const bool fon = FUSE && fm < M && fq4 * 4 < klen;
  f4 hv = {0.f, 0.f, 0.f, 0.f}, rowv[FUSE == FUSE_RMSNORM ? FUSE_MAX_ROWS : 1];     // FUSE_RMSNORM: this thread's element, its 4 columns of every row
  if constexpr (FUSE == FUSE_NONE) {
    load_x();   // before the weights: loads retire in order, so the x image never waits behind a weight chunk
  } else if constexpr (FUSE == FUSE_RMSNORM) {
    if (fon) {
      hv = *reinterpret_cast<const f4*>(f.h + (int64_t)fm * K + k0 + fq4 * 4);
      bg = *reinterpret_cast<const u2v*>(f.norm_w + k0 + fq4 * 4);
    }
#pragma unroll
    for (int r = 0; r < FUSE_MAX_ROWS; ++r)                 // blockDim.x * 4 == K (host check): the glue launch's column split
      rowv[r] = r < M ? *reinterpret_cast<const f4*>(f.h + (int64_t)r * K + tid * 4) : f4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int zz = 0; zz < NPZ; ++zz) pg[zz] = pu[zz] = f4{0.f, 0.f, 0.f, 0.f};
    if (fon) {
      const int64_t pslab = (int64_t)M * 2 * K;
      const float* pp = f.pP + (int64_t)fm * 2 * K + k0 + fq4 * 4;
#pragma unroll
      for (int zz = 0; zz < NPZ; ++zz) {
        if (zz < f.pnz) {
          pg[zz] = *reinterpret_cast<const f4*>(pp + zz * pslab);
          pu[zz] = *reinterpret_cast<const f4*>(pp + zz * pslab + K);
        }
      }
      if (f.pb) {
        bg = *reinterpret_cast<const u2v*>(f.pb + k0 + fq4 * 4);
        bu = *reinterpret_cast<const u2v*>(f.pb + K + k0 + fq4 * 4);
      }
    }
  }
  // ---- weight ring: one chunk (8 KiB per wave) in flight in registers
  constexpr int NI = W4 ? 2 : (W8 ? 4 : 8);         // 16-byte loads per lane and chunk
  u32x4 ring[DEPTH][NI];
  float ringa[DEPTH][W4 ? NI : 1];                  // NF4: the absmax of the block each load lies in
  int it = t0, ich = 0;                             // issue cursor (tile, chunk), tile-major
  // bf16: instruction i of a chunk: rows (i & 1) * 8 + r8, 16-byte slot (i >> 1) * 8 + c8 — whole 128-byte lines
  // fp8:  instruction i: rows i * 4 + fq, bytes fr * 16 .. + 16 of the row's 256 (two whole lines per row)
  // nf4:  instruction i: rows i * 8 + r8, bytes c8 * 16 .. + 16 of the row's 128 (one whole line per row) = k c8 * 32 .. + 32
  auto issue = [&](u32x4 (&dst)[NI], float (&dsta)[W4 ? NI : 1]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if constexpr (W4) {
        const int n = min(it * 16 + i * 8 + r8, Ntot - 1);
        const int k = k0 + min(ich * WCH + c8 * 32, klen - 32);
        dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Wq + (((int64_t)n * K + k) >> 1)));
        dsta[i] = wscale[(int64_t)n * (K >> 6) + (k >> 6)];
      } else if constexpr (W8) {
        const int n = min(it * 16 + i * 4 + fq, Ntot - 1);
        const int k = min(ich * WCH + fr * 16, klen - 16);
        dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Wq + (int64_t)n * K + k0 + k));
      } else {
        const int row = (i & 1) * 8 + r8;
        const int n = min(it * 16 + row, Ntot - 1);
        const int k = min(ich * WCH + ((i >> 1) * 8 + c8) * 8, klen - 8);
        // nontemporal: every weight byte is used once (21.7 vs 23.4 us on RF w12 at 16 rows)
        dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + (int64_t)n * K + k0 + k));
      }
    }
    if (++ich == nch) { ich = 0; it += twaves; }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < total) issue(ring[d], ringa[d]);
  if constexpr (FUSE == FUSE_NONE) {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int rr = wave + j * nw;
      const int h = rr >= xr_used ? 1 : 0, m = rr - h * xr_used;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int slot = lane + u * 64;
        if (rr < 2 * xr_used && slot < slots) *reinterpret_cast<u32x4*>(xs_raw + xslot(h * XR + m, srow, slot)) = xv[j][u];
      }
    }
  } else {
    // rows >= M and columns >= klen of the image are zero: clear it, then write what this launch computes
    for (int i = tid; i < 2 * XR * (srow >> 3); i += blockDim.x) *reinterpret_cast<u32x4*>(xs_raw + i * 8) = u32x4{0u, 0u, 0u, 0u};
    float* scr = reinterpret_cast<float*>(reinterpret_cast<char*>(xs_raw) + (size_t)2 * XR * srow * sizeof(bf16_t));   // wave 0's weight tile: free until the K loop
    if constexpr (FUSE == FUSE_RMSNORM) {                   // row statistics, summed as llm_glue_kernel's block_sum does: wave sums, then the waves in order
#pragma unroll
      for (int r = 0; r < FUSE_MAX_ROWS; ++r) {
        const f4 v = rowv[r];
        const float sw = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
        if (lane == 0) scr[r * 16 + wave] = sw;
      }
    }
    __syncthreads();
    if constexpr (FUSE == FUSE_RMSNORM) {
      if (fon) {
        float ss = 0.f;
        for (int i = 0; i < nw; ++i) ss += scr[fm * 16 + i];
        const float rstd = rsqrtf(ss / (float)K + f.eps);
        const float o0 = hv.x * rstd * bf16lo_to_f32(bg.x), o1 = hv.y * rstd * bf16hi_to_f32(bg.x);
        const float o2 = hv.z * rstd * bf16lo_to_f32(bg.y), o3 = hv.w * rstd * bf16hi_to_f32(bg.y);
        uint32_t h0, l0, h1, l1;
        split_pk_bf16(o0, o1, h0, l0);
        split_pk_bf16(o2, o3, h1, l1);
        const int off = xslot(fm, srow, fq4 >> 1) + (fq4 & 1) * 4;
        *reinterpret_cast<u2v*>(xs_raw + off) = u2v{h0, h1};
        *reinterpret_cast<u2v*>(xs_raw + XR * srow + off) = u2v{l0, l1};
      }
    } else
    if (fon) {
      f4 g = {bf16lo_to_f32(bg.x), bf16hi_to_f32(bg.x), bf16lo_to_f32(bg.y), bf16hi_to_f32(bg.y)};
      f4 u = {bf16lo_to_f32(bu.x), bf16hi_to_f32(bu.x), bf16lo_to_f32(bu.y), bf16hi_to_f32(bu.y)};
#pragma unroll
      for (int zz = 0; zz < NPZ; ++zz) { g += pg[zz]; u += pu[zz]; }
      uint32_t h0, l0, h1, l1;
      split_pk_bf16(silu_f(g[0]) * u[0], silu_f(g[1]) * u[1], h0, l0);
      split_pk_bf16(silu_f(g[2]) * u[2], silu_f(g[3]) * u[3], h1, l1);
      const int off = xslot(fm, srow, fq4 >> 1) + (fq4 & 1) * 4;
      *reinterpret_cast<u2v*>(xs_raw + off) = u2v{h0, h1};
      *reinterpret_cast<u2v*>(xs_raw + XR * srow + off) = u2v{l0, l1};
    }
  }
  __syncthreads();
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  int t = t0, ch = 0;
  for (int q0 = 0; q0 < total; q0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int q = q0 + d;
      if (q < total) {
        // park the landed chunk in the wave's LDS tile ...
        if constexpr (W4) {                          // 32 codes -> 32 bf16 = the row's slots 4 c8 .. 4 c8 + 3
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const Nf4Tab tb = nf4_table(ringa[d][i]);
            const int row = i * 8 + r8;
            *reinterpret_cast<u32x4*>(wbuf + wslot4(row, 4 * c8 + 0)) = nf4x8_to_bf16(tb, ring[d][i].x);
            *reinterpret_cast<u32x4*>(wbuf + wslot4(row, 4 * c8 + 1)) = nf4x8_to_bf16(tb, ring[d][i].y);
            *reinterpret_cast<u32x4*>(wbuf + wslot4(row, 4 * c8 + 2)) = nf4x8_to_bf16(tb, ring[d][i].z);
            *reinterpret_cast<u32x4*>(wbuf + wslot4(row, 4 * c8 + 3)) = nf4x8_to_bf16(tb, ring[d][i].w);
          }
        } else if constexpr (W8) {                   // 16 e4m3 -> 16 bf16 = the row's slots 2 fr and 2 fr + 1
          // every other quad of lanes stores its odd slot first: the 8 lanes of one LDS write group (same row) then cover
          // slots {0, 2, 4, 6, 9, 11, 13, 15} (mod 16) = 8 distinct 16-byte bank groups
          auto park8 = [&](auto i8) {
            constexpr bool I8 = decltype(i8)::value;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              const int row = i * 4 + fq, sw = (fr >> 2) & 1;
              const float sc = I8 ? wscale[min(t * 16 + row, Ntot - 1)] : 1.0f;      // int8: the row scale rides the conversion
              const u32x4 a = w8x8_to_bf16<I8>(ring[d][i].x, ring[d][i].y, sc), b = w8x8_to_bf16<I8>(ring[d][i].z, ring[d][i].w, sc);
              *reinterpret_cast<u32x4*>(wbuf + wslot(row, 2 * fr + sw)) = sw ? b : a;
              *reinterpret_cast<u32x4*>(wbuf + wslot(row, 2 * fr + 1 - sw)) = sw ? a : b;
            }
          };
          if (wf == MN_W_INT8) park8(std::true_type{}); else park8(std::false_type{});      // one scalar branch per chunk
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            *reinterpret_cast<u32x4*>(wbuf + wslot((i & 1) * 8 + r8, (i >> 1) * 8 + c8)) = ring[d][i];
        }
        // ... refill its registers with the chunk DEPTH ahead ...
        if (q + DEPTH < total) issue(ring[d], ringa[d]);
        // ... and multiply: 8 MFMA steps of 32 k (columns >= klen of x are zero in LDS)
#pragma unroll
        for (int sstep = 0; sstep < 8; ++sstep) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(wbuf + (W4 ? wslot4(fr, sstep * 4 + fq) : wslot(fr, sstep * 4 + fq)));
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (mt < mtn) {
              const int xs4 = ch * (WCH / 8) + sstep * 4 + fq;       // XR % 16 == 0: hi and lo rows swizzle alike
              const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xs_raw + xslot(mt * 16 + fr, srow, xs4));
              const bf16x8 al = *reinterpret_cast<const bf16x8*>(xs_raw + xslot(XR + mt * 16 + fr, srow, xs4));
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w, acc[mt], 0, 0, 0);
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w, acc[mt], 0, 0, 0);
            }
          }
        }
        if (++ch == nch) {                          // tile done: D layout row m = fq*4 + r, col n = t*16 + fr
          const int nn = t * 16 + fr;
          float rs = 1.0f;
          if constexpr (W8) rs = wf == MN_W_INT8 ? 1.0f : wscale[min(nn, Ntot - 1)];       // (int8 products are already scaled)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (nn < Ntot) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + fq * 4 + r;
                if (m < nrows) P[(int64_t)z * p_slab + (int64_t)(row0 + m) * Ntot + nn] = W8 ? acc[mt][r] * rs : acc[mt][r];
              }
            }
            acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          ch = 0;
          t += twaves;
        }
      }
    }
  }
}

struct StreamPlan { int ks, nw, gx, nz; size_t lds; };
static inline int wq_of(int wfmt) { return wfmt == MN_W_NF4 ? 2 : (wfmt ? 1 : 0); }

// Tuning knobs for in-process A/B (not part of the stable ABI): force the slice length (in 256-k chunks) and the
// waves per workgroup.
int g_kch = 0, g_nw = 0, g_tune_K = 0, g_w8_depth = 1;      // fp8: one 4 KiB chunk in flight per wave measured faster than two (13.0 vs 13.3 us on RF w12, 8.1 vs 9.3 on w3)

// Launch shape for one [Ntot, K] matrix that has `slots` CUs to itself (dense: the whole chip; grouped: the chip's
// share of one group).  One workgroup per CU (LDS-bound); the cost is the bytes a CU moves: weight chunks of its
// waves + its x image + its share of the partial slabs (written here, read back by the reducing kernel).
// E.g. RF w12 (1024 tiles x 12 chunks) runs as 4 slices of 768 x 64 workgroups x 8 waves = exactly 2 tiles per wave,
// RF w3 (192 tiles x 32 chunks) as 16 slices of 512 x 16 workgroups x 12 waves = exactly 1 tile per wave.
StreamPlan stream_plan(int mt, int Ntot, int K, int slots, int wq = 0) {      // wq: 0 bf16, 1 one byte per weight, 2 NF4
  const int ntiles = (Ntot + 15) / 16;
  const double chunk_bytes = wq == 2 ? 2304.0 : (wq ? 4096.0 : 8192.0);  // HBM bytes of one 16-row x 256-k weight chunk (NF4: codes + absmax)
  StreamPlan best{};
  double best_cost = 1e30;
  for (int kch = 1; kch <= MAX_KCH; ++kch) {
    if (g_kch > 0 && kch != g_kch && (g_tune_K == 0 || g_tune_K == K)) continue;
    const int ks = kch * WCH, nz = (K + ks - 1) / ks;
    for (int nw = 8; nw <= 16; nw += 4) {
      if (g_nw > 0 && nw != g_nw && (g_tune_K == 0 || g_tune_K == K)) continue;
      const size_t lds = (size_t)2 * 16 * mt * ks * sizeof(bf16_t) + (size_t)nw * 16 * WCH * 2;
      if (lds > LDS_CAP) continue;
      int gx = (int)mn_cdiv(ntiles, nw);
      const int gxmax = slots / nz > 0 ? slots / nz : 1;
      if (gx > gxmax) gx = gxmax;
      const int tiles_w = (int)mn_cdiv(ntiles, (int64_t)gx * nw);
      const double cost = (double)nw * tiles_w * kch * chunk_bytes + kch * 16384.0 * mt +
                          (double)nz * 16 * mt * Ntot * 8.0 / slots + (nz > slots ? 1e12 : 0.0);
      if (cost < best_cost) { best_cost = cost; best = StreamPlan{ks, nw, gx, nz, lds}; }
    }
  }
  return best;
}

template <int MT, int DEPTH, int MAXT, int W8>
void stream_launch_d(const StreamPlan& pl, int G, const bf16_t* Y, int64_t y_lo, const void* W, const float* wscale, float* P,
                     int64_t p_slab, int M, int Ntot, int K, hipStream_t st, int wf = MN_W_FP8_E4M3) {
  static bool opted = false;
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_mfma_lds_kernel<MT, DEPTH, MAXT, W8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP);
    opted = true;
  }
  hipLaunchKernelGGL((stream_mfma_lds_kernel<MT, DEPTH, MAXT, W8>), dim3(pl.gx, pl.nz, G), dim3(pl.nw * 64), pl.lds, st, Y, y_lo, W,
                     wscale, P, p_slab, M, Ntot, K, pl.ks, StreamFuse{}, wf);
}

template <int MT, int W8>       // W8 = the kernel's WQ: 0 bf16, 1 byte formats, 2 NF4
void stream_launch(const StreamPlan& pl, int G, const bf16_t* Y, int64_t y_lo, const void* W, const float* wscale, float* P,
                   int64_t p_slab, int M, int Ntot, int K, hipStream_t st, int wf = MN_W_FP8_E4M3) {
  // 8-wave workgroups compile for 512 threads, larger ones for 1024
  // (a 2-deep ring measured 2-4 % slower at every shape: 20.4 vs 20.0 us on RF w12 at 16 rows, 24.6 vs 23.8 at 32)
  if (W8 == 1 && g_w8_depth == 2) {
    if (pl.nw <= 8) stream_launch_d<MT, (W8 == 1 ? 2 : 1), 512, W8>(pl, G, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, st, wf);
    else stream_launch_d<MT, (W8 == 1 ? 2 : 1), 1024, W8>(pl, G, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, st, wf);
    return;
  }
  if (pl.nw <= 8) stream_launch_d<MT, 1, 512, W8>(pl, G, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, st, wf);
  else stream_launch_d<MT, 1, 1024, W8>(pl, G, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, st, wf);
}

template <int W8>
void stream_launch_fused(const StreamPlan& pl, const void* W, const float* wscale, float* P, int M, int Ntot, int K, const StreamFuse& f,
                         hipStream_t st, int wf) {
  static bool opted[2] = {false, false};
  const int big = pl.nw > 8;
  if (!opted[big]) {
    if (big) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_mfma_lds_kernel<1, 1, 1024, W8, FUSE_SWIGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_mfma_lds_kernel<1, 1, 512, W8, FUSE_SWIGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP);
    opted[big] = true;
  }
  if (big)
    hipLaunchKernelGGL((stream_mfma_lds_kernel<1, 1, 1024, W8, FUSE_SWIGLU>), dim3(pl.gx, pl.nz, 1), dim3(pl.nw * 64), pl.lds, st, (const bf16_t*)nullptr,
                       (int64_t)0, W, wscale, P, (int64_t)M * Ntot, M, Ntot, K, pl.ks, f, wf);
  else
    hipLaunchKernelGGL((stream_mfma_lds_kernel<1, 1, 512, W8, FUSE_SWIGLU>), dim3(pl.gx, pl.nz, 1), dim3(pl.nw * 64), pl.lds, st, (const bf16_t*)nullptr,
                       (int64_t)0, W, wscale, P, (int64_t)M * Ntot, M, Ntot, K, pl.ks, f, wf);
}

}  // namespace

// The launch plan of the fused form is the plain one's (same slices, same slabs).
bool stream_fused_ok(int wfmt, int M, int Ntot, int K, int prev_nz) {
  if (M < 1 || M > FUSE_MAX_ROWS || (K % (wfmt == MN_W_NF4 ? 64 : (wfmt ? 16 : 8))) != 0 || prev_nz < 1 || prev_nz > FUSE_PNZ) return false;
  const StreamPlan pl = stream_plan(1, Ntot, K, mn_num_cus(), wq_of(wfmt));
  return (int64_t)M * (pl.ks / 4) <= (int64_t)pl.nw * 64;      // one float4 element of the x image per thread
}

// RMSNorm prologue (the decoder chain's QKV launch): the plain launch's plan; the statistic's column split needs blockDim * 4 == K
bool stream_rmsnorm_ok(int M, int Ntot, int K) {
  if (M < 1 || M > FUSE_MAX_ROWS || (K % 8) != 0) return false;
  const StreamPlan pl = stream_plan(1, Ntot, K, mn_num_cus(), 0);
  return pl.nw == 8 && pl.nw * 64 * 4 == K && (int64_t)M * (pl.ks / 4) <= (int64_t)pl.nw * 64 && pl.nw <= 16;
}
int stream_rmsnorm(const bf16_t* W, float* P, int M, int Ntot, int K, const float* h, const bf16_t* norm_w, float eps, void* stream) {
  MN_CHECK_ARG(W && P && h && norm_w && stream_rmsnorm_ok(M, Ntot, K), "stream_rmsnorm: shape cannot run fused");
  const StreamPlan pl = stream_plan(1, Ntot, K, mn_num_cus(), 0);
  static bool opted = false;
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_mfma_lds_kernel<1, 1, 512, 0, FUSE_RMSNORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP);
    opted = true;
  }
  StreamFuse f{};
  f.h = h; f.norm_w = norm_w; f.eps = eps;
  hipLaunchKernelGGL((stream_mfma_lds_kernel<1, 1, 512, 0, FUSE_RMSNORM>), dim3(pl.gx, pl.nz, 1), dim3(pl.nw * 64), pl.lds, mn_stream(stream), (const bf16_t*)nullptr,
                     (int64_t)0, (const void*)W, (const float*)nullptr, P, (int64_t)M * Ntot, M, Ntot, K, pl.ks, f, 0);
  MN_CHECK_LAUNCH("stream_rmsnorm");
  return pl.nz;
}

int stream_fused(int wfmt, const void* W, const float* wscale, float* P, int M, int Ntot, int K, const StreamFuse& f, void* stream) {
  MN_CHECK_ARG(W && P && f.pP && stream_fused_ok(wfmt, M, Ntot, K, f.pnz) && (!wfmt || wscale), "stream_fused: shape cannot run fused");
  const StreamPlan pl = stream_plan(1, Ntot, K, mn_num_cus(), wq_of(wfmt));
  if (wfmt == MN_W_NF4) stream_launch_fused<2>(pl, W, wscale, P, M, Ntot, K, f, mn_stream(stream), wfmt);
  else if (wfmt) stream_launch_fused<1>(pl, W, wscale, P, M, Ntot, K, f, mn_stream(stream), wfmt);
  else stream_launch_fused<0>(pl, W, wscale, P, M, Ntot, K, f, mn_stream(stream), 0);
  MN_CHECK_LAUNCH("stream_fused");
  return pl.nz;
}

#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_stream_tune_plan(int kch, int nw) { g_kch = kch; g_nw = nw; g_tune_K = 0; }
extern "C" MN_DEV_API void mn_stream_tune_plan_k(int K, int kch, int nw) { g_kch = kch; g_nw = nw; g_tune_K = K; }   // only matrices with this K
extern "C" MN_DEV_API void mn_stream_tune_w8(int depth) { g_w8_depth = depth; }
#endif

extern "C" int mn_stream_mfma_slices(int M, int Ntot, int K) {
  if (M > 32) return mn_stream_kloop_slices(M, Ntot, K);
  return stream_plan(M > 16 ? 2 : 1, Ntot, K, mn_num_cus()).nz;
}
extern "C" int mn_stream_mfma_wq_slices(int wfmt, int M, int Ntot, int K) {
  if (wfmt == MN_W_BF16) return mn_stream_mfma_slices(M, Ntot, K);
  if (M > 32) return mn_stream_kloop_wq_slices(wfmt, M, Ntot, K);
  return stream_plan(M > 16 ? 2 : 1, Ntot, K, mn_num_cus(), wq_of(wfmt)).nz;
}
extern "C" int mn_stream_mfma_w8_slices(int M, int Ntot, int K) { return mn_stream_mfma_wq_slices(MN_W_FP8_E4M3, M, Ntot, K); }

// Dense: Y [2][M][K] bf16 (hi rows then lo rows), W [Ntot][K], P [nz][M][Ntot].  Returns nz (< 0: error).
extern "C" int mn_stream_mfma(const uint16_t* Y, const uint16_t* W, float* P, int M, int Ntot, int K, void* stream) {
  MN_CHECK_ARG(Y && W && P && M >= 1 && M <= 64 && Ntot >= 1 && K >= 8 && (K % 8) == 0, "mn_stream_mfma: bad args");
  if (M > 32) return mn_stream_kloop(Y, W, P, M, Ntot, K, stream);      // 33..64 rows: K-loop form, two tiles per wave
  const int mt = M > 16 ? 2 : 1;
  const StreamPlan pl = stream_plan(mt, Ntot, K, mn_num_cus());
  if (mt == 1) stream_launch<1, 0>(pl, 1, Y, (int64_t)M * K, W, nullptr, P, (int64_t)M * Ntot, M, Ntot, K, mn_stream(stream));
  else stream_launch<2, 0>(pl, 1, Y, (int64_t)M * K, W, nullptr, P, (int64_t)M * Ntot, M, Ntot, K, mn_stream(stream));
  MN_CHECK_LAUNCH("mn_stream_mfma");
  return pl.nz;
}

// The same launch on fp8 weights: Wq e4m3 bytes [Ntot][K] (K % 16 == 0, 16-byte aligned rows), wscale fp32 [Ntot];
// P [nz][M][Ntot] with nz = mn_stream_mfma_w8_slices(M, Ntot, K).
extern "C" int mn_stream_mfma_wq(const uint16_t* Y, const uint8_t* Wq, const float* wscale, float* P, int M, int Ntot, int K, int wfmt,
                                 void* stream) {
  MN_CHECK_ARG(Y && Wq && wscale && P && M >= 1 && M <= 64 && Ntot >= 1 && K >= 16 && (K % 16) == 0 && (((uintptr_t)Wq) & 15) == 0 &&
                   (wfmt == MN_W_FP8_E4M3 || wfmt == MN_W_INT8 || (wfmt == MN_W_NF4 && (K % 64) == 0)),
               "mn_stream_mfma_wq: bad args (K %% 16 == 0 — NF4: %% 64 —, 16-byte aligned weights, wfmt 1 | 2 | 3)");
  if (M > 32) return mn_stream_kloop_wq(Y, Wq, wscale, P, M, Ntot, K, wfmt, stream);
  const int mt = M > 16 ? 2 : 1;
  const StreamPlan pl = stream_plan(mt, Ntot, K, mn_num_cus(), wq_of(wfmt));
  if (wfmt == MN_W_NF4) {
    if (mt == 1) stream_launch<1, 2>(pl, 1, Y, (int64_t)M * K, Wq, wscale, P, (int64_t)M * Ntot, M, Ntot, K, mn_stream(stream), wfmt);
    else stream_launch<2, 2>(pl, 1, Y, (int64_t)M * K, Wq, wscale, P, (int64_t)M * Ntot, M, Ntot, K, mn_stream(stream), wfmt);
  } else if (mt == 1) stream_launch<1, 1>(pl, 1, Y, (int64_t)M * K, Wq, wscale, P, (int64_t)M * Ntot, M, Ntot, K, mn_stream(stream), wfmt);
  else stream_launch<2, 1>(pl, 1, Y, (int64_t)M * K, Wq, wscale, P, (int64_t)M * Ntot, M, Ntot, K, mn_stream(stream), wfmt);
  MN_CHECK_LAUNCH("mn_stream_mfma_wq");
  return pl.nz;
}
extern "C" int mn_stream_mfma_w8(const uint16_t* Y, const uint8_t* Wq, const float* wscale, float* P, int M, int Ntot, int K,
                                 void* stream) {
  return mn_stream_mfma_wq(Y, Wq, wscale, P, M, Ntot, K, MN_W_FP8_E4M3, stream);
}

// ---- grouped form (MoE experts): every row count runs the K-loop kernel with 2 K-ranges — measured against this file's
// K-slice kernel with a blockIdx.z = group dimension (tools/ab_moe_grouped.py, all 66 experts of a 16B-A3B layer):
// 64 rows 207 vs 236 us per layer, 32 rows 201 vs 227, 16 rows 158 vs 162.
extern "C" int mn_stream_mfma_grouped_slices(int G, int max_rows, int Ntot, int K) {
  (void)G; (void)max_rows; (void)Ntot;
  return K > 64 ? 2 : 1;
}

// Group g (of G) multiplies the x rows xrows[off[g] .. off[g+1]) (identity when xrows == NULL) by W + g * w_stride and
// writes partial rows off[g].. of P [nz][p_rows][Ntot].  Y holds y_rows hi rows then y_rows lo rows.  No group may have
// more than max_rows (<= 64) rows.  off / xrows live in device memory.
extern "C" int mn_stream_mfma_grouped(const uint16_t* Y, int y_rows, const uint16_t* W, int64_t w_stride, float* P,
                                      int p_rows, const int32_t* off, const int32_t* xrows, int G, int max_rows,
                                      int Ntot, int K, void* stream) {
  const int nz = mn_stream_mfma_grouped_slices(G, max_rows, Ntot, K);
  return mn_stream_kloop_grouped(Y, y_rows, W, w_stride, P, p_rows, off, xrows, G, max_rows, 0, nz, Ntot, K, stream);
}

// The grouped form on fp8 weights: group g's matrix is Wq + g * w_stride bytes, its row scales wscale + g * s_stride.
extern "C" int mn_stream_mfma_grouped_wq(const uint16_t* Y, int y_rows, const uint8_t* Wq, int64_t w_stride, const float* wscale,
                                         int64_t s_stride, float* P, int p_rows, const int32_t* off, const int32_t* xrows, int G,
                                         int max_rows, int Ntot, int K, int wfmt, void* stream) {
  // (K-ranges are whole weight pieces — 128 k of e4m3 / int8, 256 k of NF4 codes — so a short K may have fewer ranges than the bf16 form's
  // two; callers size their slabs with mn_stream_mfma_grouped_slices, an upper bound, and reduce the count this call returns)
  const int piece_k = wfmt == MN_W_NF4 ? 256 : 128;
  int nz = mn_stream_mfma_grouped_slices(G, max_rows, Ntot, K);
  if (nz > (K + piece_k - 1) / piece_k) nz = (K + piece_k - 1) / piece_k;
  return mn_stream_kloop_grouped_wq(Y, y_rows, Wq, w_stride, wscale, s_stride, P, p_rows, off, xrows, G, max_rows, nz, Ntot, K, wfmt, stream);
}
extern "C" int mn_stream_mfma_grouped_w8(const uint16_t* Y, int y_rows, const uint8_t* Wq, int64_t w_stride, const float* wscale,
                                         int64_t s_stride, float* P, int p_rows, const int32_t* off, const int32_t* xrows, int G,
                                         int max_rows, int Ntot, int K, void* stream) {
  return mn_stream_mfma_grouped_wq(Y, y_rows, Wq, w_stride, wscale, s_stride, P, p_rows, off, xrows, G, max_rows, Ntot, K, MN_W_FP8_E4M3, stream);
}

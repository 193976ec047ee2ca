// stream_kc.hip — "K-complete" weight-streaming launches for the RF ResBlock chain at <= 2 rows (the CFG rows of ONE image: the
// reference's own call shape, modeling_bailing_moe.py:1659-1670 -> diff_loss_rf_swiglu.py:263-272).
//
// The K-slice kernel (stream_mfma.hip) gives every workgroup one K-slice of the matrix and lets its waves own different output
// tiles, so a launch ends in split-K slabs that a LATER launch has to sum: per ResBlock  w12 -> [w3, with the slab sum + SwiGLU in
// its prologue] -> [glue: slab sum + gated residual + LayerNorm-modulate + hi/lo split] = three dependent launches, the third a
// 2-workgroup kernel that took 6 of a block's 35.5 us (profiles/r04_batch1_final_site_stats.txt: 10 % of an image).  LayerNorm needs
// whole rows, so the glue cannot ride w3's tail as long as w3's workgroups hold K-slices (round 4 measured the last-arriver form: slower).
//
// Here the decomposition is turned round: a workgroup owns whole OUTPUT tiles and its waves split K, so the sum over K finishes inside
// the workgroup (one LDS exchange) and the epilogue sees finished values:
//   w12'  256 workgroups x 8 waves = 2 (gate, up) tile pairs x 2 K-halves; prologue: LayerNorm-modulate of the whole row block h
//         [M, w] (every workgroup redoes it for itself: 24 KB of L2 reads and two block reductions, issued AFTER the first weight
//         chunks, so the weight stream never waits for it) -> bf16 hi/lo image in LDS; epilogue: bias + SwiGLU + hi/lo split ->
//         w3's operand [2][M][hidden] bf16 (64 KB in all).
//   w3'   192 workgroups (one per 16-column tile) x 8 waves = 8 K-ranges; prologue: copy the operand into LDS; epilogue: bias,
//         gated residual  h[m, n] += gate[m, n] * y  in place.
// Two launches per ResBlock, no slabs, no glue.  Weight bytes, MFMA work and per-wave streaming (6 / 4 chunks of 8 KiB per wave; one /
// two chunks in flight: deeper rings measured slower, as in the K-slice kernel) are those of the K-slice form; all four weight formats (bf16, e4m3, int8, NF4: w8_codec.h) are template instances.
// Rows >= M of the MFMA's 16-row operand carry copies of the real rows: output rows are independent, the copies' results are never stored.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "grid_bar.h"
#include "stream_fuse.h"
#include "w8_codec.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WCH = 256;             // k per weight chunk
constexpr int KC_WAVES = 8;
constexpr int KC_MAX_M = 4;

__device__ __forceinline__ int wslot(int row, int slot) { return row * (WCH * 2) + (((slot) ^ (row & 15)) << 4); }
__device__ __forceinline__ int wslot4(int row, int slot) { return row * (WCH * 2) + (((((slot >> 2) | ((slot & 3) << 3))) ^ (row & 7)) << 4); }

template <int WQ>
struct Chunk {
  static constexpr int NI = WQ == 2 ? 2 : (WQ == 1 ? 4 : 8);        // 16-byte loads per lane and 16 x 256 chunk
  u32x4 q[NI];
  float a[WQ == 2 ? NI : 1];                                         // NF4: absmax of the block each load lies in
};

// One 16-row x 256-k chunk of W (rows n0 .., k from k0) into registers: whole-line nontemporal loads (layouts: stream_mfma.hip).
template <int WQ>
__device__ __forceinline__ void issue(Chunk<WQ>& c, const void* Wv, const float* wscale, int n0, int Ntot, int K, int k0, int lane) {
  const int fr = lane & 15, fq = lane >> 4, r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
  for (int i = 0; i < Chunk<WQ>::NI; ++i) {
    if constexpr (WQ == 2) {
      const int n = min(n0 + i * 8 + r8, Ntot - 1), k = k0 + c8 * 32;
      c.q[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(Wv) + (((int64_t)n * K + k) >> 1)));
      c.a[i] = wscale[(int64_t)n * (K >> 6) + (k >> 6)];
    } else if constexpr (WQ == 1) {
      const int n = min(n0 + i * 4 + fq, Ntot - 1);
      c.q[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(Wv) + (int64_t)n * K + k0 + fr * 16));
    } else {
      const int n = min(n0 + (i & 1) * 8 + r8, Ntot - 1);
      c.q[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(Wv) + (int64_t)n * K + k0 + ((i >> 1) * 8 + c8) * 8));
    }
  }
}

// ... and from the registers into the wave's 8 KiB LDS tile as bf16 (exact conversions; int8 / NF4 carry their scales here)
template <int WQ>
__device__ __forceinline__ void park(const Chunk<WQ>& c, char* wbuf, int lane, int wf, const float* wscale, int n0, int Ntot) {
  const int fr = lane & 15, fq = lane >> 4, r8 = lane >> 3, c8 = lane & 7;
  if constexpr (WQ == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const Nf4Tab tb = nf4_table(c.a[i]);
      const int row = i * 8 + r8;
      *reinterpret_cast<u32x4*>(wbuf + wslot4(row, 4 * c8 + 0)) = nf4x8_to_bf16(tb, c.q[i].x);
      *reinterpret_cast<u32x4*>(wbuf + wslot4(row, 4 * c8 + 1)) = nf4x8_to_bf16(tb, c.q[i].y);
      *reinterpret_cast<u32x4*>(wbuf + wslot4(row, 4 * c8 + 2)) = nf4x8_to_bf16(tb, c.q[i].z);
      *reinterpret_cast<u32x4*>(wbuf + wslot4(row, 4 * c8 + 3)) = nf4x8_to_bf16(tb, c.q[i].w);
    }
  } else if constexpr (WQ == 1) {
    auto park8 = [&](auto i8) {
      constexpr bool I8 = decltype(i8)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 4 + fq, sw = (fr >> 2) & 1;
        const float sc = I8 ? wscale[min(n0 + row, Ntot - 1)] : 1.0f;
        const u32x4 a = w8x8_to_bf16<I8>(c.q[i].x, c.q[i].y, sc), b = w8x8_to_bf16<I8>(c.q[i].z, c.q[i].w, sc);
        *reinterpret_cast<u32x4*>(wbuf + wslot(row, 2 * fr + sw)) = sw ? b : a;
        *reinterpret_cast<u32x4*>(wbuf + wslot(row, 2 * fr + 1 - sw)) = sw ? a : b;
      }
    };
    if (wf == MN_W_INT8) park8(std::true_type{}); else park8(std::false_type{});
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(wbuf + wslot((i & 1) * 8 + r8, (i >> 1) * 8 + c8)) = c.q[i];
  }
}

// x image in LDS: rows [0, M) hi, [M, 2M) lo, `xstride` bytes per row (K * 2 + 64: two rows' 64-byte fragment reads of one MFMA
// step land on disjoint banks)
__device__ __forceinline__ int xoff(int row, int slot, int xstride) { return row * xstride + (slot << 4); }

// 8 MFMA steps of one parked chunk whose k starts at kc (absolute in the x image)
template <int WQ>
__device__ __forceinline__ void mma_chunk(f32x4& acc, const char* wbuf, const char* xs, int xstride, int M, int kc, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  const int ra = fr < M ? fr : 0;                   // rows >= M: copies (their output rows are never stored; masking their LDS reads off measured 2-4 % slower)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const bf16x8 w = *reinterpret_cast<const bf16x8*>(wbuf + (WQ == 2 ? wslot4(fr, s * 4 + fq) : wslot(fr, s * 4 + fq)));
    const int slot = (kc >> 3) + s * 4 + fq;
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xs + xoff(ra, slot, xstride));
    const bf16x8 al = *reinterpret_cast<const bf16x8*>(xs + xoff(M + ra, slot, xstride));
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w, acc, 0, 0, 0);
  }
}

// ---- half tiles (3 rows: w3's operand image takes 96 of the 160 KiB, the waves' weight tiles shrink from 8 to 4 KiB) -----------------------
// A chunk still travels as 16 rows x 256 k in registers; it is parked and multiplied in two halves of 128 k: half h = slots [16 h, 16 h + 16)
// of the rows (the XOR swizzle of wslot permutes within 16 slots, so a half keeps its own bank pattern), 256 bytes per row.
__device__ __forceinline__ int wslot_half(int row, int slot16) { return row * WCH + ((slot16 ^ (row & 15)) << 4); }

// compiler-only ordering point for data one wave's lanes hand each other through LDS (no instruction is emitted)
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

template <int WQ, int HF>
__device__ __forceinline__ void park_half(const Chunk<WQ>& c, char* wbuf, int lane) {
  // bf16 only.  (An e4m3 / int8 form — the lanes of one half parking while the others idle — was built and measured: the conversions run
  // twice at full cost under the exec mask, 3 rows 6.73 ms against the chain's 5.88; NF4 tiles are transposed over the whole chunk.)
  static_assert(WQ == 0, "half tiles: bf16 weights");
  const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
  for (int i = 4 * HF; i < 4 * HF + 4; ++i) *reinterpret_cast<u32x4*>(wbuf + wslot_half((i & 1) * 8 + r8, ((i >> 1) & 1) * 8 + c8)) = c.q[i];
}

// the 4 MFMA steps of half HF of a parked chunk whose k starts at kc
template <int HF>
__device__ __forceinline__ void mma_half(f32x4& acc, const char* wbuf, const char* xs, int xstride, int M, int kc, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  const int ra = fr < M ? fr : 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    // (read with the type the tile was STORED with: park_half stores u32x4, and a bf16x8-typed load may be scheduled across those stores
    // by type-based alias analysis)
    const bf16x8 w = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wbuf + wslot_half(fr, s * 4 + fq)));
    const int slot = (kc >> 3) + (4 * HF + s) * 4 + fq;
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xs + xoff(ra, slot, xstride));
    const bf16x8 al = *reinterpret_cast<const bf16x8*>(xs + xoff(M + ra, slot, xstride));
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w, acc, 0, 0, 0);
  }
}

// One wave's weight stream of a phase: rows [n0, n0 + 16) of W [Ntot][K], k from kbeg, nch chunks.
struct WStream {
  const void* W; const float* wscale; int n0, Ntot, K, kbeg, nch, wf; bool live;
};
template <int WQ>
__device__ __forceinline__ void issue(Chunk<WQ>& c, const WStream& st, int chunk, int lane) {
  issue<WQ>(c, st.W, st.wscale, st.n0, st.Ntot, st.K, st.kbeg + chunk * WCH, lane);
}

// Data that workgroups hand to each other INSIDE a launch (the persistent form below) is stored write-through and loaded past the L1
// (`sc1`, cdna_hip_programming.md Guideline 16, R1): no release / acquire fence — an L2 write-back + L1 invalidate per barrier measured
// 6 us (tools/exp/gridbar_bench.hip) — and the same instructions otherwise.
constexpr int AUX_SC1 = 16;
// dev-library timeline of the persistent launch (mn_rf_kc_trace): workgroup 0, thread 0 stamps the 100 MHz clock
__device__ __forceinline__ void stamp(uint64_t* tr, int k) { if (tr && threadIdx.x == 0) tr[k] = wall_clock64(); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

struct W12Args {
  const float* h; int M, w, hid;
  const bf16_t* ln_g; const bf16_t* ln_b; const float* shift; const float* scale; int64_t ldmod;
  const void* W; const float* wscale; const bf16_t* bias; int wf;
  bf16_t* Y;                                        // [2][M][hid]: hi rows, lo rows (w3's operand)
};

// ---- w12': LayerNorm-modulate prologue, (gate, up) tile pairs x K-halves, SwiGLU + split epilogue -------------------------------------
// wave -> (tile pair, gate | up, K-range) of workgroup `vb`: the first weight row, the first k and the chunk count of its stream
template <int NW>
struct W12Wave {
  static constexpr int KS = NW / 4;                 // K-splits: NW waves = 2 tile pairs x (gate, up) x KS K-ranges
  int n0, kbeg, nch;
  bool live;
  __device__ __forceinline__ W12Wave(const W12Args& a, int vb, int wave) {
    const int pair = wave / (2 * KS), which = (wave / KS) & 1, kh = wave % KS;
    const int tile = vb * 2 + pair;                 // hidden units [16 tile, + 16)
    live = tile * 16 < a.hid;
    n0 = which * a.hid + tile * 16;
    const int Kh = a.w / KS;
    kbeg = kh * Kh;
    nch = Kh / WCH;
  }
};

// the first RD chunks of a wave's stream (the persistent form requests them BEFORE the grid barrier: weights do not depend on it)
template <int WQ, int RD, int NW>
__device__ __forceinline__ void w12_issue_first(const W12Args& a, int vb, Chunk<WQ> (&ring)[RD], int wave, int lane) {
  const W12Wave<NW> wv(a, vb, wave);
  if (wv.live) {
#pragma unroll
    for (int d = 0; d < RD; ++d)
      if (d < wv.nch) issue<WQ>(ring[d], a.W, a.wscale, wv.n0, 2 * a.hid, a.w, wv.kbeg + d * WCH, lane);
  }
}

template <int NW>
__device__ __forceinline__ WStream w12_stream(const W12Args& a, int vb, int wave, bool on) {
  const W12Wave<NW> wv(a, vb, wave);
  return WStream{a.W, a.wscale, wv.n0, 2 * a.hid, a.w, wv.kbeg, wv.nch, a.wf, on && wv.live};
}

// Persistent form, before the barrier wait: chunk 0 of the wave's NEXT stream into its LDS tile `wbuf` (the workgroup's previous phase
// is over: LDS is free), chunks 1 .. RD requested — what crosses the barrier in flight.
template <int WQ, int RD>
__device__ __forceinline__ void prefetch(const WStream& st, char* wbuf, Chunk<WQ> (&ring)[RD], int lane) {
  if (!st.live) return;
  // chunk 0 and the ring are requested TOGETHER (one memory round trip, not two dependent ones) and chunk 0 is parked when it lands, the
  // ring still in flight: sampler call 6.22 -> 6.16 ms bf16, 4.97 -> 4.84 e4m3, 5.17 -> 5.18 NF4 (profiles/r05_rf_persist_ab.txt)
  Chunk<WQ> c0;
  issue<WQ>(c0, st, 0, lane);
#pragma unroll
  for (int d = 0; d < RD; ++d)
    if (1 + d < st.nch) issue<WQ>(ring[d], st, 1 + d, lane);
  park<WQ>(c0, wbuf, lane, st.wf, st.wscale, st.n0, st.Ntot);
}

// Half-tile form (3 rows): nothing is parked before the wait — the ring takes chunks 0 .. RD - 1 and the body starts like a launch's
template <int WQ, int RD>
__device__ __forceinline__ void prefetch_ring(const WStream& st, Chunk<WQ> (&ring)[RD], int lane) {
  if (!st.live) return;
#pragma unroll
  for (int d = 0; d < RD; ++d)
    if (d < st.nch) issue<WQ>(ring[d], st, d, lane);
}

// The body of workgroup `vb` of w12' (PRE, the persistent form: chunk 0 is parked, the ring holds chunks 1 .. RD, and the data other
// workgroups produced is read / written coherently).
template <int WQ, int MR, int RD, int NW, bool PRE>
__device__ __forceinline__ void w12_body(const W12Args& a, char* lds, int vb, Chunk<WQ> (&ring)[RD], uint64_t* tr = nullptr) {
  constexpr int KS = NW / 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, K = a.w, hid = a.hid, Ntot = 2 * hid;
  const int xstride = K * 2 + 64;
  char* xs = lds;
  char* wbuf = lds + (size_t)2 * M * xstride + (size_t)wave * 16 * WCH * 2;
  float* red = reinterpret_cast<float*>(lds + (size_t)2 * M * xstride + (size_t)NW * 16 * WCH * 2);    // [NW waves][KC_MAX_M][16] + [16 stats]
  float* stat = red + NW * KC_MAX_M * 16;
  const W12Wave<NW> wv(a, vb, wave);
  const bool live = wv.live;
  const int n0 = wv.n0, kbeg = wv.kbeg, nch = wv.nch;
  // ---- prologue: x = LayerNorm(h; g, b) * (1 + scale) + shift  for all M rows, split into bf16 hi / lo  (diff_loss_rf_swiglu.py:270)
  constexpr int PC = 1024 / (NW * 64);               // float4 columns per thread and row: K <= 4096
  const int nq = K >> 2;
  f4 hv[MR][PC];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const int c = tid + j * (NW * 64);
      if constexpr (PRE) {
        const u32x4 t = (m < M && c < nq) ? __builtin_amdgcn_raw_buffer_load_b128(coh_rsrc(a.h, (uint32_t)M * K * 4), (m * K + c * 4) * 4, 0, AUX_SC1) : u32x4{0u, 0u, 0u, 0u};
        hv[m][j] = f4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
      } else {
        hv[m][j] = (m < M && c < nq) ? *reinterpret_cast<const f4*>(a.h + (int64_t)m * K + c * 4) : f4{0.f, 0.f, 0.f, 0.f};
      }
    }
  // modulation / LayerNorm parameters requested before the reductions (one round trip, like the glue kernel it replaces)
  f4 sc[MR][PC], sh[MR][PC];
  u2 lg[PC], lb[PC];
#pragma unroll
  for (int j = 0; j < PC; ++j) {
    const int c = tid + j * (NW * 64);
    lg[j] = (a.ln_g && c < nq) ? *reinterpret_cast<const u2*>(a.ln_g + c * 4) : u2{0x3f803f80u, 0x3f803f80u};
    lb[j] = (a.ln_b && c < nq) ? *reinterpret_cast<const u2*>(a.ln_b + c * 4) : u2{0u, 0u};
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      sc[m][j] = (m < M && c < nq) ? *reinterpret_cast<const f4*>(a.scale + (int64_t)m * a.ldmod + c * 4) : f4{0.f, 0.f, 0.f, 0.f};
      sh[m][j] = (m < M && c < nq) ? *reinterpret_cast<const f4*>(a.shift + (int64_t)m * a.ldmod + c * 4) : f4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- then this wave's first weight chunks: loads retire in order, so the prologue's operands (requested above) arrive first and the
  // LayerNorm runs while the weight chunks are in flight
  if constexpr (!PRE) w12_issue_first<WQ, RD, NW>(a, vb, ring, wave, lane);
  float mean[MR], rstd[MR];
  {
    float s[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      s[m] = 0.f;
#pragma unroll
      for (int j = 0; j < PC; ++j) s[m] += (hv[m][j].x + hv[m][j].y) + (hv[m][j].z + hv[m][j].w);
      s[m] = wave_sum(s[m]);
    }
    if (lane == 0)
#pragma unroll
      for (int m = 0; m < MR; ++m) red[wave * KC_MAX_M + m] = s[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float t = 0.f;
      for (int wv = 0; wv < NW; ++wv) t += red[wv * KC_MAX_M + m];
      mean[m] = t / (float)K;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < PC; ++j) {
        const int c = tid + j * (NW * 64);
        if (c < nq) { const f4 d = hv[m][j] - mean[m]; ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
      }
      s[m] = wave_sum(ss);
    }
    if (lane == 0)
#pragma unroll
      for (int m = 0; m < MR; ++m) red[wave * KC_MAX_M + m] = s[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float t = 0.f;
      for (int wv = 0; wv < NW; ++wv) t += red[wv * KC_MAX_M + m];
      rstd[m] = rsqrtf(t / (float)K + 1e-6f);
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (m < M) {
#pragma unroll
      for (int j = 0; j < PC; ++j) {
        const int c = tid + j * (NW * 64);
        if (c < nq) {
          const float g4[4] = {bf16lo_to_f32(lg[j].x), bf16hi_to_f32(lg[j].x), bf16lo_to_f32(lg[j].y), bf16hi_to_f32(lg[j].y)};
          const float b4[4] = {bf16lo_to_f32(lb[j].x), bf16hi_to_f32(lb[j].x), bf16lo_to_f32(lb[j].y), bf16hi_to_f32(lb[j].y)};
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = (hv[m][j][e] - mean[m]) * rstd[m];
            t = t * g4[e] + b4[e];
            v[e] = t * (1.0f + sc[m][j][e]) + sh[m][j][e];
          }
          uint32_t h0, l0, h1, l1;
          split_pk_bf16(v[0], v[1], h0, l0);
          split_pk_bf16(v[2], v[3], h1, l1);
          *reinterpret_cast<u2*>(xs + (size_t)m * xstride + c * 8) = u2{h0, h1};
          *reinterpret_cast<u2*>(xs + (size_t)(M + m) * xstride + c * 8) = u2{l0, l1};
        }
      }
    }
  }
  __syncthreads();
  stamp(tr, 1);
  // ---- stream this wave's K-half of its tile: RD chunks in flight
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    if constexpr (PRE) mma_chunk<WQ>(acc, wbuf, xs, xstride, M, kbeg, lane);       // chunk 0 was parked before the barrier; the ring holds 1 ..
    for (int c = PRE ? 1 : 0; c < nch; c += RD) {
#pragma unroll
      for (int d = 0; d < RD; ++d) {
        if (c + d < nch) {
          park<WQ>(ring[d], wbuf, lane, a.wf, a.wscale, n0, Ntot);
          if (c + d + RD < nch) issue<WQ>(ring[d], a.W, a.wscale, n0, Ntot, K, kbeg + (c + d + RD) * WCH, lane);
          mma_chunk<WQ>(acc, wbuf, xs, xstride, M, kbeg + (c + d) * WCH, lane);
        }
      }
    }
  }
  // ---- the K-halves meet in LDS: lane (fr, fq = 0) holds rows 0..3 of column fr
  if ((lane >> 4) == 0) {
    float rs = 1.0f;
    if constexpr (WQ == 1) rs = a.wf == MN_W_INT8 ? 1.0f : a.wscale[min(n0 + (lane & 15), Ntot - 1)];      // e4m3: the row scale on the sums
#pragma unroll
    for (int r = 0; r < KC_MAX_M; ++r) red[(wave * KC_MAX_M + r) * 16 + (lane & 15)] = acc[r] * rs;
  }
  __syncthreads();
  stamp(tr, 2);
  // thread t < 2 pairs x M x 16: y = silu(gate + bg) * (up + bu), split, stored as w3's operand  (diff_loss_rf_swiglu.py:30-34)
  if (tid < 2 * KC_MAX_M * 16) {
    const int p = tid / (KC_MAX_M * 16), m = (tid / 16) % KC_MAX_M, col = tid & 15;
    const int t2 = vb * 2 + p, n = t2 * 16 + col;
    if (m < M && n < hid) {
      const int wg = p * 2 * KS, wu = wg + KS;
      float g = 0.f, u = 0.f;
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        g += red[((wg + q) * KC_MAX_M + m) * 16 + col];
        u += red[((wu + q) * KC_MAX_M + m) * 16 + col];
      }
      if (a.bias) { g += bf16_to_f32(a.bias[n]); u += bf16_to_f32(a.bias[hid + n]); }
      const float y = silu_f(g) * u;
      const bf16_t hi = f32_to_bf16(y);
      const bf16_t lo = f32_to_bf16(y - bf16_to_f32(hi));
      if constexpr (PRE) {
        const __amdgpu_buffer_rsrc_t ry = coh_rsrc(a.Y, (uint32_t)2 * M * hid * 2);
        __builtin_amdgcn_raw_buffer_store_b16((short)hi, ry, (m * hid + n) * 2, 0, AUX_SC1);
        __builtin_amdgcn_raw_buffer_store_b16((short)lo, ry, ((M + m) * hid + n) * 2, 0, AUX_SC1);
      } else {
        a.Y[(int64_t)m * hid + n] = hi;
        a.Y[(int64_t)(M + m) * hid + n] = lo;
      }
    }
  }
  (void)stat;
}

template <int WQ, int MR, int RD, int NW>
__global__ __launch_bounds__(NW * 64) void rf_w12_kc_kernel(const W12Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  Chunk<WQ> ring[RD];
  w12_body<WQ, MR, RD, NW, false>(a, lds, (int)blockIdx.x, ring);
}

struct W3Args {
  const bf16_t* Y; int M, w, hid;                   // operand [2][M][hid]
  const void* W; const float* wscale; const bf16_t* bias; int wf;
  const float* gate; int64_t ldmod;
  float* h;                                         // [M][w] fp32, updated in place
};

// ---- w3': one output tile per workgroup, 8 K-ranges, gated-residual epilogue ---------------------------------------------------------------
template <int WQ, int RD>
__device__ __forceinline__ void w3_issue_first(const W3Args& a, int vb, Chunk<WQ> (&ring)[RD], int wave, int lane) {
  const int Kw = a.hid / KC_WAVES, nch = Kw / WCH;
#pragma unroll
  for (int d = 0; d < RD; ++d)
    if (d < nch) issue<WQ>(ring[d], a.W, a.wscale, vb * 16, a.w, a.hid, wave * Kw + d * WCH, lane);
}

__device__ __forceinline__ WStream w3_stream(const W3Args& a, int vb, int wave, bool on) {
  const int Kw = a.hid / KC_WAVES;
  return WStream{a.W, a.wscale, vb * 16, a.w, a.hid, wave * Kw, Kw / WCH, a.wf, on};
}

// The body of workgroup `vb` (one 16-column tile) of w3' (PRE: as in w12_body).
// HALF (3 rows): 4 KiB weight tiles, chunks parked in two halves; with PRE the ring already holds chunks 0 .. RD - 1 (prefetch_ring).
// XN: 16-byte pieces of the operand per thread (2 M hid / 8 <= XN x 512).
template <int WQ, int RD, bool PRE, int XN = 8, bool HALF = false>
__device__ __forceinline__ void w3_body(const W3Args& a, char* lds, int vb, Chunk<WQ> (&ring)[RD], uint64_t* tr = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, K = a.hid, Ntot = a.w;
  const int xstride = K * 2 + 64;
  constexpr int TILE = (HALF ? 8 : 16) * WCH * 2;   // bytes of a wave's weight tile
  char* xs = lds;
  char* wbuf = lds + (size_t)2 * M * xstride + (size_t)wave * TILE;
  float* red = reinterpret_cast<float*>(lds + (size_t)2 * M * xstride + (size_t)KC_WAVES * TILE);
  const int n0 = vb * 16;
  const int Kw = K / KC_WAVES, kbeg = wave * Kw, nch = Kw / WCH;
  // ---- the operand (hi rows, lo rows: 2 M hid bf16, 16 bytes per thread and step) and the epilogue's operands go to registers FIRST:
  // loads retire in order, so they land before the (younger) weight chunks and the x image is in LDS while those are still in flight
  const int spr = K >> 3;                           // 16-byte slots per row
  u32x4 xr[XN];
#pragma unroll
  for (int j = 0; j < XN; ++j) {
    const int i = tid + j * (KC_WAVES * 64);
    xr[j] = u32x4{0u, 0u, 0u, 0u};
    if (i < 2 * M * spr) {                          // rows are contiguous: piece i = (row i / spr, slot i % spr)
      if constexpr (PRE) xr[j] = __builtin_amdgcn_raw_buffer_load_b128(coh_rsrc(a.Y, (uint32_t)2 * M * K * 2), i * 16, 0, AUX_SC1);
      else xr[j] = *reinterpret_cast<const u32x4*>(a.Y + (int64_t)i * 8);
    }
  }
  float h_old = 0.f, g_old = 0.f, b_old = 0.f, s_old = 1.f;
  if (tid < KC_MAX_M * 16) {
    const int m = tid >> 4, n = n0 + (tid & 15);
    if (m < M && n < Ntot) {
      if constexpr (PRE) h_old = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(coh_rsrc(a.h, (uint32_t)M * Ntot * 4), (m * Ntot + n) * 4, 0, AUX_SC1));
      else h_old = a.h[(int64_t)m * Ntot + n];
      g_old = a.gate[(int64_t)m * a.ldmod + n];
      if (a.bias) b_old = bf16_to_f32(a.bias[n]);
      if constexpr (WQ == 1) { if (a.wf != MN_W_INT8) s_old = a.wscale[n]; }
    }
  }
  // RD chunks in flight per wave: with RD = 4 a wave's whole 1024-k range is requested up front — one HBM round trip per launch
  if constexpr (!PRE) w3_issue_first<WQ, RD>(a, vb, ring, wave, lane);
#pragma unroll
  for (int j = 0; j < XN; ++j) {
    const int i = tid + j * (KC_WAVES * 64);
    if (i < 2 * M * spr) { const int r = i / spr, sl = i - r * spr; *reinterpret_cast<u32x4*>(xs + xoff(r, sl, xstride)) = xr[j]; }
  }
  __syncthreads();
  stamp(tr, 1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr bool PARKED = PRE && !HALF;            // chunk 0 sits in the tile, the ring holds chunks 1 ..
  if constexpr (PARKED) mma_chunk<WQ>(acc, wbuf, xs, xstride, M, kbeg, lane);
  for (int c = PARKED ? 1 : 0; c < nch; c += RD) {
#pragma unroll
    for (int d = 0; d < RD; ++d) {
      if (c + d < nch) {
        if constexpr (HALF) {                      // (the ring slot is free once its second half is parked: re-issue in between)
          // The tile is re-used WITHIN the chunk and its lanes exchange data through it: the compiler sees one thread, whose own stores
          // and loads it may prove disjoint — an e4m3 instance of this loop did move park_half<1>'s stores above mma_half<0>'s loads
          // and multiplied the wrong half (tools/exp/kc_w3_probe.hip).  wave_fence(): no instruction, but no LDS access moves across it; the
          // hardware keeps one wave's LDS operations in order.
          park_half<WQ, 0>(ring[d], wbuf, lane);
          wave_fence();
          mma_half<0>(acc, wbuf, xs, xstride, M, kbeg + (c + d) * WCH, lane);
          wave_fence();
          park_half<WQ, 1>(ring[d], wbuf, lane);
          wave_fence();
          if (c + d + RD < nch) issue<WQ>(ring[d], a.W, a.wscale, n0, Ntot, K, kbeg + (c + d + RD) * WCH, lane);
          mma_half<1>(acc, wbuf, xs, xstride, M, kbeg + (c + d) * WCH, lane);
          wave_fence();
        } else {
          park<WQ>(ring[d], wbuf, lane, a.wf, a.wscale, n0, Ntot);
          if (c + d + RD < nch) issue<WQ>(ring[d], a.W, a.wscale, n0, Ntot, K, kbeg + (c + d + RD) * WCH, lane);
          mma_chunk<WQ>(acc, wbuf, xs, xstride, M, kbeg + (c + d) * WCH, lane);
        }
      }
    }
  }
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int r = 0; r < KC_MAX_M; ++r) red[(wave * KC_MAX_M + r) * 16 + (lane & 15)] = acc[r];
  }
  __syncthreads();
  stamp(tr, 2);
  // thread t < M x 16:  h[m, n] += gate[m, n] * (sum over the K-ranges + b3[n])   (ResBlock, diff_loss_rf_swiglu.py:272)
  if (tid < KC_MAX_M * 16) {
    const int m = tid >> 4, col = tid & 15, n = n0 + col;
    if (m < M && n < Ntot) {
      float y = 0.f;
#pragma unroll
      for (int wv = 0; wv < KC_WAVES; ++wv) y += red[(wv * KC_MAX_M + m) * 16 + col];
      y = y * s_old + b_old;                        // (s_old: the e4m3 row scale, 1 otherwise)
      if constexpr (PRE) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h_old + g_old * y), coh_rsrc(a.h, (uint32_t)M * Ntot * 4), (m * Ntot + n) * 4, 0, AUX_SC1);
      else a.h[(int64_t)m * Ntot + n] = h_old + g_old * y;
    }
  }
}

template <int WQ, int RD, int XN = 8, bool HALF = false>
__global__ __launch_bounds__(KC_WAVES * 64) void rf_w3_kc_kernel(const W3Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  Chunk<WQ> ring[RD];
  w3_body<WQ, RD, false, XN, HALF>(a, lds, (int)blockIdx.x, ring);
}

// ---- the ResBlocks of one Euler step as ONE persistent launch ----------------------------------------------------------------------------
// 24 dependent launches per Euler step leave the HBM pipe empty for a launch gap + a ramp each (w12' 20.8 us for 14.6 us of bytes, w3'
// 12.5 for 7.3: profiles/r05_batch1_site_stats.txt).  Weights do not depend on the activations, so a workgroup that stays resident can
// REQUEST the next phase's first weight chunks before it waits for the other workgroups: the grid barrier's round trip is covered by
// loads already in flight.  One workgroup per CU (co-residency: grid <= CU count, checked by the host; LDS > 80 KiB keeps a second
// workgroup of another launch off the CU), phases = the bodies above, between them a grid barrier in device memory (GridBar): every
// wave drains its stores (vmcnt(0)), workgroup barrier, one lane raises the workgroup's flag; wave 0 polls the flags with agent-scope
// loads and releases its workgroup.  The barrier contains NO release / acquire fence (an L2 write-back + L1 invalidate per barrier
// measured 6 us): the eight XCDs' L2s are not coherent with each other, so correctness rests on the PAYLOAD — every datum handed between
// workgroups (h, w3's operand Y, v) MUST be stored write-through (`sc1` buffer stores) and loaded past the L1 (`sc1` loads).  A later
// phase that reads or writes a hand-off buffer with plain loads / stores is a coherence bug, not a style choice.  The wait is bounded
// (wall clock): on expiry the error word and the caller's status word are set, the result is NaN in every workgroup (GridBar::poisoned)
// and the launch runs to its end without further waits.
constexpr int PB_MAX = 16;
struct PersistBlk {
  const void* W12; const float* s12; const bf16_t* b12; const bf16_t* ln_g; const bf16_t* ln_b;
  const void* W3; const float* s3; const bf16_t* b3;
};
struct PersistArgs {
  float* h; bf16_t* Y; int M, w, hid, wf, nblk;
  const float* mod; int64_t ldmod;                  // block b: shift = mod + 3 w b, scale = shift + w, gate = shift + 2 w
  unsigned* bar;                                    // [0, 256) one flag per workgroup, [256] error word
  unsigned epoch0;                                  // the launch's barriers publish epoch0 + 1, + 2, ...
  unsigned* status;                                 // the caller's sticky status word (grid_bar.h) or nullptr
  uint64_t wait_ticks;
  uint64_t* trace;                                  // dev library only: [workgroups][2 nblk phases][8 stamps], or NULL
  // the whole sampler in the launch (steps > 0): Euler-step boundaries and the final layer as two more phases per step
  int steps, T, rpi, n_images;
  int64_t mod_step;                                 // floats between two Euler steps' modulations
  const bf16_t* in_w; const bf16_t* in_b;           // input_proj [w][T], [w]
  const bf16_t* fin_w; const bf16_t* fin_b;         // final_layer.linear [T][w], [T]
  const float* noise; float temperature, text_cfg, image_cfg;
  float* v;                                         // [M][T]: the final layer's output, handed to the next boundary phase
  float* latent;                                    // [n_images][T]
  uint32_t lds_top;                                 // byte offset of the launch-long LDS state: x [M][T] (<= 512 floats) + 64 floats of scratch
  int fault_wg;                                     // dev library only: the workgroup that plays "late-resident" (-1: none)
  PersistBlk blk[PB_MAX];
};

// ---- whole-sampler form: the two light phases around the blocks of an Euler step ---------------------------------------------------------
// Boundary phase (every workgroup): the ODE state x [M][T] lives in EVERY workgroup's LDS (replicated: 64 floats, no hand-off, no race) —
// step 0: x = noise * temperature; later: v = the final layer's output of the previous step (coherent loads), CFG combine over the image's rows,
// x += vg / steps (diff_loss_rf_swiglu.py:144-179) — then the workgroup's 16 columns of h = input_proj(x) (:371), stored write-through.
// The arithmetic and its order are rf_step_boundary_kernel's (engine.hip).
__device__ __forceinline__ float persist_x_update(const PersistArgs& p, const float* xs, int m, int t) {
  const int T = p.T, rpi = p.rpi, r0 = (m / rpi) * rpi;
  const __amdgpu_buffer_rsrc_t rv = coh_rsrc(p.v, (uint32_t)p.M * T * 4);
  float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 3; ++r)
    if (r < rpi) v[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, ((r0 + r) * T + t) * 4, 0, AUX_SC1));
  const float vg = rpi == 3 ? v[1] + p.image_cfg * (v[2] - v[1]) + p.text_cfg * (v[0] - v[2])
                            : (rpi == 2 ? v[1] + p.text_cfg * (v[0] - v[1]) : v[0]);
  return xs[m * T + t] + vg * (1.0f / (float)p.steps);
}

__device__ __forceinline__ void persist_boundary_phase(const PersistArgs& p, char* lds, int s, int vb, int n3) {
  float* xs = reinterpret_cast<float*>(lds + p.lds_top);
  const int tid = threadIdx.x, M = p.M, T = p.T;
  if (tid < M * T) {
    const int m = tid / T, t = tid - m * T;
    xs[tid] = s == 0 ? p.noise[(m / p.rpi) * T + t] * p.temperature : persist_x_update(p, xs, m, t);
  }
  __syncthreads();
  if (vb < n3 && tid < KC_MAX_M * 16) {
    const int m = tid >> 4, n = vb * 16 + (tid & 15);
    if (m < M && n < p.w) {
      const bf16_t* wr = p.in_w + (int64_t)n * T;
      const float* xk = xs + m * T;
      float a = bf16_to_f32(p.in_b[n]);
      for (int k = 0; k < T; k += 8) {                // T % 8 == 0 (host check): 16-byte rows
        const u32x4 q = *reinterpret_cast<const u32x4*>(wr + k);
        a = fmaf(bf16lo_to_f32(q.x), xk[k], a); a = fmaf(bf16hi_to_f32(q.x), xk[k + 1], a);
        a = fmaf(bf16lo_to_f32(q.y), xk[k + 2], a); a = fmaf(bf16hi_to_f32(q.y), xk[k + 3], a);
        a = fmaf(bf16lo_to_f32(q.z), xk[k + 4], a); a = fmaf(bf16hi_to_f32(q.z), xk[k + 5], a);
        a = fmaf(bf16lo_to_f32(q.w), xk[k + 6], a); a = fmaf(bf16hi_to_f32(q.w), xk[k + 7], a);
      }
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), coh_rsrc(p.h, (uint32_t)M * p.w * 4), (m * p.w + n) * 4, 0, AUX_SC1);
    }
  }
}

// Final-layer phase (workgroup t < T owns output column t of every row): v[m][t] = fin_b[t] + sum_k fin_w[t][k] *
// (LayerNorm(h[m])[k] * (1 + scale[m][k]) + shift[m][k])   (no affine; diff_loss_rf_swiglu.py:288-292) in fp32.
template <int MR>
__device__ __forceinline__ void persist_final_phase(const PersistArgs& p, char* lds, const float* mod_s, int vb) {
  if (vb >= p.T) return;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  constexpr int NT = KC_WAVES * 64, PC = 1024 / NT;
  float* red = reinterpret_cast<float*>(lds + p.lds_top) + 512;        // [8 waves][MR rows] (64 floats of scratch: MR <= 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, M = p.M, K = p.w, nq = K >> 2;
  const float* shift = mod_s + (int64_t)p.nblk * 3 * K;
  const float* scale = shift + K;
  const __amdgpu_buffer_rsrc_t rh = coh_rsrc(p.h, (uint32_t)M * K * 4);
  f4 hv[MR][PC], sc[MR][PC], sh[MR][PC];
  u2 fw[PC];
#pragma unroll
  for (int j = 0; j < PC; ++j) {
    const int c = tid + j * NT;
    fw[j] = c < nq ? *reinterpret_cast<const u2*>(p.fin_w + (int64_t)vb * K + c * 4) : u2{0u, 0u};
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const bool on = m < M && c < nq;
      const u32x4 t = on ? __builtin_amdgcn_raw_buffer_load_b128(rh, (m * K + c * 4) * 4, 0, AUX_SC1) : u32x4{0u, 0u, 0u, 0u};
      hv[m][j] = f4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
      sc[m][j] = on ? *reinterpret_cast<const f4*>(scale + (int64_t)m * p.ldmod + c * 4) : f4{0.f, 0.f, 0.f, 0.f};
      sh[m][j] = on ? *reinterpret_cast<const f4*>(shift + (int64_t)m * p.ldmod + c * 4) : f4{0.f, 0.f, 0.f, 0.f};
    }
  }
  auto block_sum2 = [&](float (&x)[MR]) {             // sums over the workgroup of MR values, every thread gets them
#pragma unroll
    for (int m = 0; m < MR; ++m) x[m] = wave_sum(x[m]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
      for (int m = 0; m < MR; ++m) red[wave * MR + m] = x[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float t = 0.f;
      for (int wv = 0; wv < KC_WAVES; ++wv) t += red[wv * MR + m];
      x[m] = t;
    }
  };
  float mean[MR], rstd[MR], dot[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    mean[m] = 0.f;
#pragma unroll
    for (int j = 0; j < PC; ++j) mean[m] += (hv[m][j].x + hv[m][j].y) + (hv[m][j].z + hv[m][j].w);
  }
  block_sum2(mean);
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    mean[m] /= (float)K;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < PC; ++j)
      if (tid + j * NT < nq) { const f4 d = hv[m][j] - mean[m]; ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
    rstd[m] = ss;
  }
  block_sum2(rstd);
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    rstd[m] = rsqrtf(rstd[m] / (float)K + 1e-6f);
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const float w4[4] = {bf16lo_to_f32(fw[j].x), bf16hi_to_f32(fw[j].x), bf16lo_to_f32(fw[j].y), bf16hi_to_f32(fw[j].y)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xn = (hv[m][j][e] - mean[m]) * rstd[m];
        a = fmaf(xn * (1.0f + sc[m][j][e]) + sh[m][j][e], w4[e], a);
      }
    }
    dot[m] = a;
  }
  block_sum2(dot);
  if (tid < M) {
    float y = dot[0];
#pragma unroll
    for (int m = 1; m < MR; ++m) y = tid == m ? dot[m] : y;
    y += bf16_to_f32(p.fin_b[vb]);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), coh_rsrc(p.v, (uint32_t)M * p.T * 4), (tid * p.T + vb) * 4, 0, AUX_SC1);
  }
}

// MR: rows the launch is compiled for — 2 (1 or 2 rows: text -> image) or 3 (3 CFG rows: editing; w3' then runs on half tiles, see w3_body)
template <int WQ, int MR = 2>
__global__ __launch_bounds__(KC_WAVES * 64) void rf_blocks_persist_kernel(const PersistArgs p) {
  constexpr int RD3 = WQ == 1 ? 4 : 2;
  constexpr bool HALF = MR > 2;
  constexpr int XN3 = MR > 2 ? 12 : 8;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vb = blockIdx.x, n12 = (p.hid + 31) / 32, n3 = (p.w + 15) / 16;
  GridBar gb{p.bar, gridDim.x, p.epoch0, p.wait_ticks, 0, p.status};
  // ONE register ring for both phases (w12' keeps one chunk in flight, w3' RD3).  Requesting the next phase's chunk 0 EARLY — while
  // this phase's stream runs out — measured slower (its traffic delays the workgroups still streaming,
  // and 32 more live registers spill; profiles/README.md r05): chunk 0 is requested after the arrival.
  Chunk<WQ> r3[RD3];
  Chunk<WQ> (&r12)[1] = *reinterpret_cast<Chunk<WQ> (*)[1]>(&r3[0]);
  auto args12 = [&](const float* base, int b) {
    const float* mod = base + (int64_t)b * 3 * p.w;
    return W12Args{p.h, p.M, p.w, p.hid, p.blk[b].ln_g, p.blk[b].ln_b, mod, mod + p.w, p.ldmod, p.blk[b].W12, p.blk[b].s12, p.blk[b].b12, p.wf, p.Y};
  };
  auto args3 = [&](const float* base, int b) {
    return W3Args{p.Y, p.M, p.w, p.hid, p.blk[b].W3, p.blk[b].s3, p.blk[b].b3, p.wf, base + (int64_t)b * 3 * p.w + 2 * p.w, p.ldmod, p.h};
  };
  char* wbuf12 = lds + (size_t)2 * p.M * (p.w * 2 + 64) + (size_t)wave * 16 * WCH * 2;
  char* wbuf3 = lds + (size_t)2 * p.M * (p.hid * 2 + 64) + (size_t)wave * (HALF ? 8 : 16) * WCH * 2;
  const bool whole = p.steps > 0;                   // the whole sampler: steps x (boundary phase, blocks, final-layer phase)
  const int nsteps = whole ? p.steps : 1;
  W12Args a12 = args12(p.mod, 0);
  WStream s12 = w12_stream<KC_WAVES>(a12, vb, wave, vb < n12);
  prefetch<WQ, 1>(s12, wbuf12, r12, lane);      // (neither light phase touches the weight tiles or the ring)
  __syncthreads();
  uint64_t* tr = !whole && p.trace ? p.trace + (size_t)vb * 2 * p.nblk * 8 : nullptr;
#ifdef MN_DEV_HOOKS
  if (p.fault_wg == vb) {                           // fault injection (tests): this workgroup is "late-resident" — 3 x the others' patience
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < 3 * p.wait_ticks) __builtin_amdgcn_s_sleep(8);
  }
#endif
  for (int s = 0; s < nsteps; ++s) {
  const float* mod_s = p.mod + (int64_t)s * p.mod_step;
  if (whole) {
    persist_boundary_phase(p, lds, s, vb, n3);
    gb.arrive();
    gb.wait();
  }
  for (int b = 0; b < p.nblk; ++b) {
    const W3Args a3 = args3(mod_s, b);
    const WStream s3 = w3_stream(a3, vb, wave, vb < n3);
    stamp(tr, 0);
    if (vb < n12) w12_body<WQ, MR, 1, KC_WAVES, true>(a12, lds, vb, r12, tr);
    stamp(tr, 3);
    gb.arrive();
    stamp(tr, 4);
    if constexpr (HALF) prefetch_ring<WQ, RD3>(s3, r3, lane);
    else prefetch<WQ, RD3>(s3, wbuf3, r3, lane);
    stamp(tr, 5);
    gb.wait();
    stamp(tr, 6);
    if (tr) tr += 8;
    stamp(tr, 0);
    const bool last = b + 1 == p.nblk;
    const bool more = !last || (whole && s + 1 < nsteps);         // another w12' follows in this launch: the next block's, or the next step's first
    if (more) {
      a12 = args12(last ? mod_s + p.mod_step : mod_s, last ? 0 : b + 1);
      s12 = w12_stream<KC_WAVES>(a12, vb, wave, vb < n12);
    } else s12.live = false;
    if (vb < n3) w3_body<WQ, RD3, true, XN3, HALF>(a3, lds, vb, r3, tr);
    stamp(tr, 3);
    if (more || whole) {
      gb.arrive();
      stamp(tr, 4);
      prefetch<WQ, 1>(s12, wbuf12, r12, lane);
      stamp(tr, 5);
      gb.wait();
      stamp(tr, 6);
    }
    if (tr) tr += 8;
  }
  if (whole) {
    persist_final_phase<MR>(p, lds, mod_s, vb);
    gb.arrive();
    gb.wait();
  }
  }
  // a timed-out barrier poisons the state: the caller's result is NaN, never a silently wrong number
  const int dead = gb.poisoned();
  if (whole && vb == 0 && tid < p.n_images * p.T) {      // the last Euler update; one row per image leaves (the rows of an image hold the same x)
    const float* xs = reinterpret_cast<const float*>(lds + p.lds_top);
    const int img = tid / p.T, t = tid - img * p.T;
    p.latent[img * p.T + t] = dead ? __builtin_nanf("") : persist_x_update(p, xs, img * p.rpi, t);
  }
  if (dead && vb < n3 && tid < p.M * 16 && vb * 16 + (tid & 15) < p.w) p.h[(int64_t)(tid >> 4) * p.w + vb * 16 + (tid & 15)] = __builtin_nanf("");
}

uint64_t* g_kc_trace = nullptr;
int g_kc_persist_all = 0;            // dev library: the persistent launch for int8 as well (mn_rf_kc_persist_all)
int g_kc_fault_wg = -1; unsigned g_kc_fault_ms = 0;   // dev library: fault injection of the grid barrier (mn_rf_kc_fault)
int g_kc_rd12 = 1, g_kc_rd3 = 0;     // weight chunks in flight per wave; rd3 = 0: by format (dev-library A/B knob: mn_rf_kc_tune)

size_t w12_lds(int M, int w, int nw) { return (size_t)2 * M * (w * 2 + 64) + (size_t)nw * 16 * WCH * 2 + (nw * KC_MAX_M * 16 + 16) * sizeof(float); }
// (3 rows: the operand image takes 96 KiB, the waves' weight tiles are half tiles — w3_body<.., HALF>)
size_t w3_lds(int M, int hid) { return (size_t)2 * M * (hid * 2 + 64) + (size_t)KC_WAVES * (M > 2 ? 8 : 16) * WCH * 2 + KC_WAVES * KC_MAX_M * 16 * sizeof(float); }

template <typename Kern>
void opt_in(Kern k) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace

#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_rf_kc_tune(int rd12, int rd3) { g_kc_rd12 = rd12 & 15; g_kc_rd3 = rd3; }
extern "C" MN_DEV_API void mn_rf_kc_persist_all(int on) { g_kc_persist_all = on; }
extern "C" MN_DEV_API void mn_rf_kc_trace(void* buf) { g_kc_trace = static_cast<uint64_t*>(buf); }
extern "C" MN_DEV_API void mn_rf_kc_fault(int wg, unsigned wait_ms) { g_kc_fault_wg = wg; g_kc_fault_ms = wait_ms; }
#endif

// Can the ResBlock chain of this shape run as K-complete launches?  (whole chunks per wave, the x images + weight tiles within the
// 160 KiB of LDS, the LayerNorm prologue's two float4 columns per thread)
bool rf_kc_ok(int wfmt, int M, int w, int hid) {
  if (M < 1 || M > 3 || w > 4096 || (w % (2 * WCH)) != 0 || (hid % (KC_WAVES * WCH)) != 0 || (hid % 32) != 0) return false;
  if (wfmt == MN_W_NF4 && ((w % 64) != 0 || (hid % 64) != 0)) return false;
  if (M > 2 && wfmt != MN_W_BF16) return false;    // 3 rows: bf16 only — the byte formats keep the three-launch chain (faster than a masked half-tile park: park_half)
  if ((int64_t)M * hid > (M > 2 ? 24576 : 16384)) return false;       // w3': the operand image is staged through 8 (3 rows: 12) x 16 bytes per thread
  return w12_lds(M, w, 8) <= 160 * 1024 && w3_lds(M, hid) <= 160 * 1024;
}

// Can the whole block chain of a step run as one persistent launch?  K-complete shapes whose phases fit one workgroup per CU.
bool rf_persist_ok(int wfmt, int M, int w, int hid, void* stream) {
  if (!rf_kc_ok(wfmt, M, w, hid)) return false;
  // MINGNATIVE_RF_PERSIST=0 keeps the launches: a persistent grid needs every CU of the device for itself (two PROCESSES sharing one GPU
  // could hold each other's CUs until the barrier's 2 s timeout poisons the result with NaN; inside one process launches are ordered)
  static const bool env_on = [] { const char* e = getenv("MINGNATIVE_RF_PERSIST"); return !(e && e[0] == '0'); }();
  if (!env_on) return false;
  // bf16, e4m3 and NF4 gain — sampler call at 2 rows, 24 launches per step -> one launch per step -> the whole sampler in one launch:
  // bf16 6.81 -> 6.46 -> 6.22 ms, e4m3 5.26 -> 5.17 -> 4.97, NF4 5.39 -> 5.40 -> 5.17; int8 spends the phases in its decoder, not in
  // launch gaps (5.91 -> 6.35 -> 6.17) and keeps the launches  (tools/exp/rf_persist_ab.py, profiles/r05_rf_persist_ab.txt)
  if (wfmt == MN_W_INT8 && !g_kc_persist_all) return false;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    n_cu = v;
  }
  const int G = mn_cdiv(hid, 32) > mn_cdiv(w, 16) ? mn_cdiv(hid, 32) : mn_cdiv(w, 16);
  if (G > n_cu || G > 256) return false;                // (256: the barrier's flag array)
  const size_t l12 = w12_lds(M, w, KC_WAVES), l3 = w3_lds(M, hid);
  if ((l12 > l3 ? l12 : l3) <= 80 * 1024) return false;   // (two workgroups would fit a CU: the co-residency argument needs one)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(mn_stream(stream), &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;      // (no event calls inside a capture)
  return true;
}

// ... and the whole sampler (every Euler step: boundary phase, blocks, final-layer phase) when all blocks fit one launch, the ODE state fits
// the replicated LDS copy and there is a workgroup per final-layer column.
bool rf_sampler_persist_ok(int M, int w, int hid, int depth, int T, int rpi, int n_images) {
  const int G = mn_cdiv(hid, 32) > mn_cdiv(w, 16) ? mn_cdiv(hid, 32) : mn_cdiv(w, 16);
  const size_t l12 = w12_lds(M, w, KC_WAVES), l3 = w3_lds(M, hid);
  return depth >= 1 && depth <= PB_MAX && T >= 8 && (T % 8) == 0 && M * T <= 512 && T <= G && rpi >= 1 && rpi <= 3 && n_images * rpi == M &&
         (l12 > l3 ? l12 : l3) + 16 + (512 + 64) * sizeof(float) + 256 <= 160 * 1024;
}

// Blocks [b0, b0 + nblk) of one Euler step in one launch.  `bar`: RF_PERSIST_BAR_WORDS words of device memory, zeroed by the caller;
// epoch0: this launch's first barrier epoch — successive launches on one flag array take epoch0 = 0, 64, 128, ... (a launch has
// 2 nblk - 1 <= 31 barriers).  `whole`: the whole sampler in this launch (steps x (2 nblk + 2) barriers: give it a zeroed flag array).  Two persistent launches must never share the device — each would hold CUs the other
// waits for — so launches on different streams are ordered by an event.
int rf_blocks_persist(int wfmt, float* h, bf16_t* Y3, int M, int w, int hid, const float* mod, int64_t ldmod, int nblk,
                      const void* const* W12, const float* const* s12, const bf16_t* const* b12, const bf16_t* const* ln_g,
                      const bf16_t* const* ln_b, const void* const* W3, const float* const* s3, const bf16_t* const* b3,
                      unsigned* bar, unsigned epoch0, const RfSamplerTail* whole, void* stream) {
  MN_CHECK_ARG(h && Y3 && mod && bar && nblk >= 1 && nblk <= PB_MAX && rf_kc_ok(wfmt, M, w, hid), "rf_blocks_persist: bad args");
  MN_CHECK_ARG(!whole || rf_sampler_persist_ok(M, w, hid, nblk, whole->T, whole->rpi, whole->n_images), "rf_blocks_persist: shape cannot run the whole sampler in one launch");
  PersistArgs p{};
  p.h = h; p.Y = Y3; p.M = M; p.w = w; p.hid = hid; p.wf = wfmt; p.nblk = nblk; p.mod = mod; p.ldmod = ldmod; p.bar = bar; p.epoch0 = epoch0; p.trace = g_kc_trace; p.status = mn_persist_status_word();
  p.wait_ticks = 2000ull * 100000ull;               // 2 s of the 100 MHz clock
  p.fault_wg = g_kc_fault_wg;
  if (g_kc_fault_ms) p.wait_ticks = (uint64_t)g_kc_fault_ms * 100000ull;
  for (int b = 0; b < nblk; ++b)
    p.blk[b] = PersistBlk{W12[b], wfmt ? s12[b] : nullptr, b12[b], ln_g[b], ln_b[b], W3[b], wfmt ? s3[b] : nullptr, b3[b]};
  const int G = mn_cdiv(hid, 32) > mn_cdiv(w, 16) ? mn_cdiv(hid, 32) : mn_cdiv(w, 16);
  const size_t l12 = w12_lds(M, w, KC_WAVES), l3 = w3_lds(M, hid), top = ((l12 > l3 ? l12 : l3) + 15) & ~(size_t)15;
  const size_t lds = top + (whole ? (512 + 64) * sizeof(float) : 0);
  if (whole) {
    p.steps = whole->steps; p.T = whole->T; p.rpi = whole->rpi; p.n_images = whole->n_images; p.mod_step = whole->mod_step;
    p.in_w = whole->in_w; p.in_b = whole->in_b; p.fin_w = whole->fin_w; p.fin_b = whole->fin_b; p.noise = whole->noise;
    p.temperature = whole->temperature; p.text_cfg = whole->text_cfg; p.image_cfg = whole->image_cfg; p.v = whole->v; p.latent = whole->latent;
    p.lds_top = (uint32_t)top;
    p.trace = nullptr;
  }
  hipStream_t st = mn_stream(stream);
  static std::mutex mu_attr;
  {
    std::lock_guard<std::mutex> lk(mu_attr);
    static size_t opted = 0;                         // (the kernel holds 256 bytes of static LDS: ask for what the launch needs, not for all 160 KiB)
    if (lds > opted) {
      const void* ks[4] = {reinterpret_cast<const void*>(&rf_blocks_persist_kernel<0, 2>), reinterpret_cast<const void*>(&rf_blocks_persist_kernel<1, 2>),
                           reinterpret_cast<const void*>(&rf_blocks_persist_kernel<2, 2>), reinterpret_cast<const void*>(&rf_blocks_persist_kernel<0, 3>)};
      for (const void* k : ks)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          mn_set_error("rf_blocks_persist: cannot reserve %zu bytes of LDS", lds);
          return MN_ELAUNCH;
        }
      opted = lds;
    }
  }
  { const int rc_o = mn_persist_order_before(st); if (rc_o != MN_OK) return rc_o; }
  const dim3 grid((unsigned)G), block(KC_WAVES * 64);
  if (wfmt == MN_W_NF4) hipLaunchKernelGGL((rf_blocks_persist_kernel<2, 2>), grid, block, lds, st, p);
  else if (M > 2) hipLaunchKernelGGL((rf_blocks_persist_kernel<0, 3>), grid, block, lds, st, p);
  else if (wfmt) hipLaunchKernelGGL((rf_blocks_persist_kernel<1, 2>), grid, block, lds, st, p);
  else hipLaunchKernelGGL((rf_blocks_persist_kernel<0, 2>), grid, block, lds, st, p);
  mn_persist_order_after(st);
  MN_CHECK_LAUNCH("rf_blocks_persist");
  return MN_OK;
}

// (grid_bar.h) the caller's sticky status word: raised (0x300) by a persistent launch whose grid barrier gave up
static std::atomic<unsigned*> g_persist_status{nullptr};
unsigned* mn_persist_status_word() { return g_persist_status.load(std::memory_order_relaxed); }
extern "C" int mn_persist_set_status_word(uint32_t* device_word) {
  g_persist_status.store(device_word, std::memory_order_relaxed);
  return MN_OK;
}

// (grid_bar.h) persistent launches on different streams are ordered by an event
static std::mutex g_persist_mu;
static hipEvent_t g_persist_ev = nullptr;
static hipStream_t g_persist_last = nullptr;
static bool g_persist_any = false;
int mn_persist_order_before(hipStream_t st) {
  g_persist_mu.lock();                               // held until mn_persist_order_after: launch + record are one step
  if (!g_persist_ev && hipEventCreateWithFlags(&g_persist_ev, hipEventDisableTiming) != hipSuccess) {
    g_persist_mu.unlock();
    mn_set_error("persistent launch: hipEventCreate failed");
    return MN_ELAUNCH;
  }
  if (g_persist_any && g_persist_last != st) (void)hipStreamWaitEvent(st, g_persist_ev, 0);
  return MN_OK;
}
void mn_persist_order_after(hipStream_t st) {
  (void)hipEventRecord(g_persist_ev, st);
  g_persist_last = st; g_persist_any = true;
  g_persist_mu.unlock();
}

int rf_w12_kc(int wfmt, const float* h, int M, int w, int hid, const bf16_t* ln_g, const bf16_t* ln_b, const float* shift, const float* scale,
              int64_t ldmod, const void* W12, const float* s12, const bf16_t* b12, bf16_t* Y3, void* stream) {
  MN_CHECK_ARG(h && shift && scale && W12 && Y3 && rf_kc_ok(wfmt, M, w, hid) && (!wfmt || s12), "rf_w12_kc: shape cannot run K-complete");
  const W12Args a{h, M, w, hid, ln_g, ln_b, shift, scale, ldmod, W12, s12, b12, wfmt, Y3};
  // 8 waves per workgroup (2 tile pairs x (gate, up) x 2 K-halves), ONE chunk in flight per wave: deeper rings and a 16-wave form
  // (4 K-ranges) measured slower — bf16 6.87 / 6.87 / 7.23 ms per sampler call at 1 / 2 / 3 chunks in flight, 6.99 with 16 waves
  // (tools/exp/rf_kc_sweep.py, profiles/r05_rf_kc_sweep.txt)
  constexpr int nw = 8;
  const dim3 grid((unsigned)mn_cdiv(hid, 32)), block(nw * 64);
  const size_t lds = w12_lds(M, w, nw);
  static bool opted = false;
  if (!opted) {
#define MN_OPT(WQ_) opt_in(&rf_w12_kc_kernel<WQ_, 2, 1, 8>); opt_in(&rf_w12_kc_kernel<WQ_, 2, 2, 8>); opt_in(&rf_w12_kc_kernel<WQ_, 2, 3, 8>);
    MN_OPT(0) MN_OPT(1) MN_OPT(2)
#undef MN_OPT
    opt_in(&rf_w12_kc_kernel<0, 3, 1, 8>);
    opted = true;
  }
  if (M > 2) {                                      // 3 rows (editing; bf16): the MR = 3 instance, one chunk in flight
    hipLaunchKernelGGL((rf_w12_kc_kernel<0, 3, 1, 8>), grid, block, lds, mn_stream(stream), a);
    MN_CHECK_LAUNCH("rf_w12_kc");
    return MN_OK;
  }
#define MN_KC12(WQ_)                                                                                                       \
  do {                                                                                                                     \
    if (g_kc_rd12 == 1) hipLaunchKernelGGL((rf_w12_kc_kernel<WQ_, 2, 1, 8>), grid, block, lds, mn_stream(stream), a);       \
    else if (g_kc_rd12 == 2) hipLaunchKernelGGL((rf_w12_kc_kernel<WQ_, 2, 2, 8>), grid, block, lds, mn_stream(stream), a);  \
    else hipLaunchKernelGGL((rf_w12_kc_kernel<WQ_, 2, 3, 8>), grid, block, lds, mn_stream(stream), a);                      \
  } while (0)
  if (wfmt == MN_W_NF4) MN_KC12(2); else if (wfmt) MN_KC12(1); else MN_KC12(0);
#undef MN_KC12
  MN_CHECK_LAUNCH("rf_w12_kc");
  return MN_OK;
}

int rf_w3_kc(int wfmt, const bf16_t* Y3, int M, int w, int hid, const void* W3, const float* s3, const bf16_t* b3, const float* gate,
             int64_t ldmod, float* h, void* stream) {
  MN_CHECK_ARG(Y3 && W3 && gate && h && rf_kc_ok(wfmt, M, w, hid) && (!wfmt || s3), "rf_w3_kc: shape cannot run K-complete");
  const W3Args a{Y3, M, w, hid, W3, s3, b3, wfmt, gate, ldmod, h};
  const dim3 grid((unsigned)mn_cdiv(w, 16)), block(KC_WAVES * 64);
  const size_t lds = w3_lds(M, hid);
  static bool opted = false;
  if (!opted) {
    opt_in(&rf_w3_kc_kernel<0, 1>); opt_in(&rf_w3_kc_kernel<0, 2>); opt_in(&rf_w3_kc_kernel<0, 4>);
    opt_in(&rf_w3_kc_kernel<1, 1>); opt_in(&rf_w3_kc_kernel<1, 2>); opt_in(&rf_w3_kc_kernel<1, 4>);
    opt_in(&rf_w3_kc_kernel<2, 1>); opt_in(&rf_w3_kc_kernel<2, 2>); opt_in(&rf_w3_kc_kernel<2, 4>);
    opt_in(&rf_w3_kc_kernel<0, 2, 12, true>);
    opted = true;
  }
  if (M > 2) {                                      // 3 rows (bf16): half tiles (w3_body<.., HALF>), two chunks in flight
    hipLaunchKernelGGL((rf_w3_kc_kernel<0, 2, 12, true>), grid, block, lds, mn_stream(stream), a);
    MN_CHECK_LAUNCH("rf_w3_kc");
    return MN_OK;
  }
  // chunks in flight per wave, measured best per format (tools/exp/rf_kc_sweep.py): bf16 2 (16 KiB), e4m3 / int8 4 (16 KiB), NF4 2
  const int rd3 = g_kc_rd3 ? g_kc_rd3 : ((wfmt == MN_W_FP8_E4M3 || wfmt == MN_W_INT8) ? 4 : 2);
#define MN_KC3(WQ_)                                                                                                  \
  do {                                                                                                               \
    if (rd3 == 1) hipLaunchKernelGGL((rf_w3_kc_kernel<WQ_, 1>), grid, block, lds, mn_stream(stream), a);         \
    else if (rd3 == 2) hipLaunchKernelGGL((rf_w3_kc_kernel<WQ_, 2>), grid, block, lds, mn_stream(stream), a);    \
    else hipLaunchKernelGGL((rf_w3_kc_kernel<WQ_, 4>), grid, block, lds, mn_stream(stream), a);                       \
  } while (0)
  if (wfmt == MN_W_NF4) MN_KC3(2); else if (wfmt) MN_KC3(1); else MN_KC3(0);
#undef MN_KC3
  MN_CHECK_LAUNCH("rf_w3_kc");
  return MN_OK;
}

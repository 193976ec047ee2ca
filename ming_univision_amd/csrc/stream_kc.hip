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
#include <type_traits>

#include "common.h"
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
  const int ra = fr < M ? fr : 0;                   // rows >= M: copies (their output rows are never stored)
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

struct W12Args {
  const float* h; int M, w, hid;
  const bf16_t* ln_g; const bf16_t* ln_b; const float* shift; const float* scale; int64_t ldmod;
  const void* W; const float* wscale; const bf16_t* bias; int wf;
  bf16_t* Y;                                        // [2][M][hid]: hi rows, lo rows (w3's operand)
};

// ---- w12': LayerNorm-modulate prologue, (gate, up) tile pairs x K-halves, SwiGLU + split epilogue -------------------------------------
template <int WQ, int MR, int RD, int NW>
__global__ __launch_bounds__(NW * 64) void rf_w12_kc_kernel(const W12Args a) {
  constexpr int KS = NW / 4;                        // K-splits: NW waves = 2 tile pairs x (gate, up) x KS K-ranges
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, K = a.w, hid = a.hid, Ntot = 2 * hid;
  const int xstride = K * 2 + 64;
  char* xs = lds;
  char* wbuf = lds + (size_t)2 * M * xstride + (size_t)wave * 16 * WCH * 2;
  float* red = reinterpret_cast<float*>(lds + (size_t)2 * M * xstride + (size_t)NW * 16 * WCH * 2);    // [NW waves][KC_MAX_M][16] + [16 stats]
  float* stat = red + NW * KC_MAX_M * 16;
  // wave -> (pair, gate | up, K-half)
  const int pair = wave / (2 * KS), which = (wave / KS) & 1, kh = wave % KS;
  const int tile = blockIdx.x * 2 + pair;           // hidden units [16 tile, + 16)
  const bool live = tile * 16 < hid;
  const int n0 = which * hid + tile * 16;
  const int Kh = K / KS, kbeg = kh * Kh, nch = Kh / WCH;
  // ---- prologue: x = LayerNorm(h; g, b) * (1 + scale) + shift  for all M rows, split into bf16 hi / lo  (diff_loss_rf_swiglu.py:270)
  constexpr int PC = 1024 / (NW * 64);               // float4 columns per thread and row: K <= 4096
  const int nq = K >> 2;
  f4 hv[MR][PC];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const int c = tid + j * (NW * 64);
      hv[m][j] = (m < M && c < nq) ? *reinterpret_cast<const f4*>(a.h + (int64_t)m * K + c * 4) : f4{0.f, 0.f, 0.f, 0.f};
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
  Chunk<WQ> ring[RD];
  if (live) {
#pragma unroll
    for (int d = 0; d < RD; ++d)
      if (d < nch) issue<WQ>(ring[d], a.W, a.wscale, n0, Ntot, K, kbeg + d * WCH, lane);
  }
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
  // ---- stream this wave's K-half of its tile: RD chunks in flight
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    for (int c = 0; c < nch; c += RD) {
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
  // thread t < 2 pairs x M x 16: y = silu(gate + bg) * (up + bu), split, stored as w3's operand  (diff_loss_rf_swiglu.py:30-34)
  if (tid < 2 * KC_MAX_M * 16) {
    const int p = tid / (KC_MAX_M * 16), m = (tid / 16) % KC_MAX_M, col = tid & 15;
    const int t2 = blockIdx.x * 2 + p, n = t2 * 16 + col;
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
      a.Y[(int64_t)m * hid + n] = hi;
      a.Y[(int64_t)(M + m) * hid + n] = f32_to_bf16(y - bf16_to_f32(hi));
    }
  }
  (void)stat;
}

struct W3Args {
  const bf16_t* Y; int M, w, hid;                   // operand [2][M][hid]
  const void* W; const float* wscale; const bf16_t* bias; int wf;
  const float* gate; int64_t ldmod;
  float* h;                                         // [M][w] fp32, updated in place
};

// ---- w3': one output tile per workgroup, 8 K-ranges, gated-residual epilogue ---------------------------------------------------------------
template <int WQ, int RD>
__global__ __launch_bounds__(KC_WAVES * 64) void rf_w3_kc_kernel(const W3Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, K = a.hid, Ntot = a.w;
  const int xstride = K * 2 + 64;
  char* xs = lds;
  char* wbuf = lds + (size_t)2 * M * xstride + (size_t)wave * 16 * WCH * 2;
  float* red = reinterpret_cast<float*>(lds + (size_t)2 * M * xstride + (size_t)KC_WAVES * 16 * WCH * 2);
  const int n0 = blockIdx.x * 16;
  const int Kw = K / KC_WAVES, kbeg = wave * Kw, nch = Kw / WCH;
  // ---- the operand (hi rows, lo rows: 2 M hid bf16, 16 bytes per thread and step) and the epilogue's operands go to registers FIRST:
  // loads retire in order, so they land before the (younger) weight chunks and the x image is in LDS while those are still in flight
  constexpr int XN = 8;                             // 16-byte pieces per thread: 2 M hid / 8 <= 8 x 512 (host check)
  const int spr = K >> 3;                           // 16-byte slots per row
  u32x4 xr[XN];
#pragma unroll
  for (int j = 0; j < XN; ++j) {
    const int i = tid + j * (KC_WAVES * 64);
    xr[j] = u32x4{0u, 0u, 0u, 0u};
    if (i < 2 * M * spr) xr[j] = *reinterpret_cast<const u32x4*>(a.Y + (int64_t)i * 8);      // rows are contiguous: piece i = (row i / spr, slot i % spr)
  }
  float h_old = 0.f, g_old = 0.f, b_old = 0.f, s_old = 1.f;
  if (tid < KC_MAX_M * 16) {
    const int m = tid >> 4, n = n0 + (tid & 15);
    if (m < M && n < Ntot) {
      h_old = a.h[(int64_t)m * Ntot + n];
      g_old = a.gate[(int64_t)m * a.ldmod + n];
      if (a.bias) b_old = bf16_to_f32(a.bias[n]);
      if constexpr (WQ == 1) { if (a.wf != MN_W_INT8) s_old = a.wscale[n]; }
    }
  }
  // RD chunks in flight per wave: with RD = 4 a wave's whole 1024-k range is requested up front — one HBM round trip per launch
  Chunk<WQ> ring[RD];
#pragma unroll
  for (int d = 0; d < RD; ++d)
    if (d < nch) issue<WQ>(ring[d], a.W, a.wscale, n0, Ntot, K, kbeg + d * WCH, lane);
#pragma unroll
  for (int j = 0; j < XN; ++j) {
    const int i = tid + j * (KC_WAVES * 64);
    if (i < 2 * M * spr) { const int r = i / spr, sl = i - r * spr; *reinterpret_cast<u32x4*>(xs + xoff(r, sl, xstride)) = xr[j]; }
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nch; c += RD) {
#pragma unroll
    for (int d = 0; d < RD; ++d) {
      if (c + d < nch) {
        park<WQ>(ring[d], wbuf, lane, a.wf, a.wscale, n0, Ntot);
        if (c + d + RD < nch) issue<WQ>(ring[d], a.W, a.wscale, n0, Ntot, K, kbeg + (c + d + RD) * WCH, lane);
        mma_chunk<WQ>(acc, wbuf, xs, xstride, M, kbeg + (c + d) * WCH, lane);
      }
    }
  }
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int r = 0; r < KC_MAX_M; ++r) red[(wave * KC_MAX_M + r) * 16 + (lane & 15)] = acc[r];
  }
  __syncthreads();
  // thread t < M x 16:  h[m, n] += gate[m, n] * (sum over the K-ranges + b3[n])   (ResBlock, diff_loss_rf_swiglu.py:272)
  if (tid < KC_MAX_M * 16) {
    const int m = tid >> 4, col = tid & 15, n = n0 + col;
    if (m < M && n < Ntot) {
      float y = 0.f;
#pragma unroll
      for (int wv = 0; wv < KC_WAVES; ++wv) y += red[(wv * KC_MAX_M + m) * 16 + col];
      y = y * s_old + b_old;                        // (s_old: the e4m3 row scale, 1 otherwise)
      a.h[(int64_t)m * Ntot + n] = h_old + g_old * y;
    }
  }
}

int g_kc_rd12 = 1, g_kc_rd3 = 0;     // weight chunks in flight per wave; rd3 = 0: by format (dev-library A/B knob: mn_rf_kc_tune)       // weight chunks in flight per wave (dev-library A/B knob: mn_rf_kc_tune)

size_t w12_lds(int M, int w, int nw) { return (size_t)2 * M * (w * 2 + 64) + (size_t)nw * 16 * WCH * 2 + (nw * KC_MAX_M * 16 + 16) * sizeof(float); }
size_t w3_lds(int M, int hid) { return (size_t)2 * M * (hid * 2 + 64) + (size_t)KC_WAVES * 16 * WCH * 2 + KC_WAVES * KC_MAX_M * 16 * sizeof(float); }

template <typename Kern>
void opt_in(Kern k) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace

#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_rf_kc_tune(int rd12, int rd3) { g_kc_rd12 = rd12 & 15; g_kc_rd3 = rd3; }
#endif

// Can the ResBlock chain of this shape run as K-complete launches?  (whole chunks per wave, the x images + weight tiles within the
// 160 KiB of LDS, the LayerNorm prologue's two float4 columns per thread)
bool rf_kc_ok(int wfmt, int M, int w, int hid) {
  if (M < 1 || M > 2 || w > 4096 || (w % (2 * WCH)) != 0 || (hid % (KC_WAVES * WCH)) != 0 || (hid % 32) != 0) return false;
  if (wfmt == MN_W_NF4 && ((w % 64) != 0 || (hid % 64) != 0)) return false;
  if ((int64_t)M * hid > 16384) return false;       // w3': the operand image is staged through 8 x 16 bytes per thread
  return w12_lds(M, w, 8) <= 160 * 1024 && w3_lds(M, hid) <= 160 * 1024;
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
    opted = true;
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
    opted = true;
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

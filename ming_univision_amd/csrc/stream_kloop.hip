// stream_kloop.hip — weight-streaming MFMA GEMM for 5..64 activation rows, K-loop form (gfx950).
//
//   partial[z][row][n] = sum_{k in K-range z} (x_hi[xrow,k] + x_lo[xrow,k]) * W_g[n,k]
//
// Same contract as stream_mfma.hip (bf16 hi+lo activations, fp32 K-range partials, optional expert groups), different
// decomposition: a workgroup of 8 waves owns 8 weight tiles (16 rows each) over a LONG K-range and walks it in chunks
// of 64 k.  The x chunk of all rows is double-buffered in LDS and refilled from a register ring while the previous
// chunk is multiplied, so
//   * x never has to fit in LDS as a whole (64 rows x 4 B x K would not): the x image costs 8 KiB x MT per buffer;
//   * the x fetch is pipelined along K instead of being a start-up bubble in front of the weight stream;
//   * K-ranges are long, so few (often 1-2) partial slabs are written.
// Each wave streams its own 16 NT x 64 weight chunks (whole-line nontemporal loads) through a D-deep register ring into
// a wave-private swizzled LDS tile; one workgroup barrier per chunk orders the x buffers.
//
// Used for 33..64 rows (stream_mfma.hip forwards): there each wave owns TWO adjacent weight tiles (NT = 2) so that an x
// fragment read from LDS feeds two MFMAs per half — with one tile per wave every 4 KiB weight chunk costs 32 KiB of x
// fragment reads and the kernel is LDS-bound (RF w12 at 64 rows: 40 us with NT = 1, 28-29 us with NT = 2, ring depth 2;
// a 4-deep ring spills).  At <= 32 rows it is on par with the K-slice kernel (23-25 vs 24 us), which stays the default there.
#include <type_traits>

#include "common.h"
#include "w8_codec.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KW = 8;                // waves per workgroup

struct KGroups {
  const int32_t* off;      // [G + 1] offsets into the sorted row list; nullptr = one dense group of M rows
  const int32_t* xrows;    // [total] x row of each sorted row; nullptr = identity (off[g] + r)
  int64_t w_stride;        // elements between consecutive groups' weight matrices
  int row_lo, row_hi;      // only groups with row_lo < rows <= row_hi are processed by this launch
  int64_t s_stride;        // 8-bit weights: floats between consecutive groups' row scales
  int wf;                  // 8-bit weights: MN_W_FP8_E4M3 | MN_W_INT8 (w8_codec.h)
};

// byte offset of 16-byte slot `slot` of chunk row `row` (CK k per chunk = CK/8 slots per row): the XOR swizzle makes
// the b128 fragment reads (16 rows x 4 slots per lane group) conflict-free for 256-byte and 128-byte rows alike
template <int CK>
__device__ __forceinline__ int cslot(int row, int slot) { return row * (CK * 2) + ((slot ^ (row & (CK / 8 - 1))) << 4); }

// W8: the weights are OCP e4m3 bytes with one fp32 scale per output row (group g: Wq + g * w_stride bytes, scales
// wscale + g * s_stride).  A wave's weight load then covers a PIECE of 128 k (16 NT rows x 128 B: whole lines, 16 bytes = 16 k per
// lane) that serves TWO 64-k chunks.  The sum over k may run in any order, so inside a piece the chunks are interleaved: chunk
// 2p takes the first 8 k of every lane's 16, chunk 2p + 1 the second 8 — every lane converts 8 bytes to one 16-byte bf16 slot
// (exact) and parks it at each chunk, full-wave stores into the SAME 64-k bf16 tile the bf16 form uses, and the x pieces are
// gathered with the matching stride (slot j of chunk 2p + h = k 16 j + 8 h .. + 8 of the piece).  The MFMA loop, the LDS
// footprint (two workgroups per CU) and the barriers are those of the bf16 form; HBM bytes halve; the row scale multiplies the
// fp32 accumulators at the partial store.  CK must be 64; D = 4 is the loop's unroll (two pieces in flight per wave), the x ring
// stays two chunks deep so that the four-row-tile form keeps the bf16 form's register count (two workgroups per CU).
// WQ = 2 (NF4, w8_codec.h): two codes per byte + wscale = the absmax table [rows][K / 64].  A weight load (16 bytes per lane) is a
// piece of 256 k that serves FOUR 64-k chunks — chunk 4p + h takes dword h of every lane's four, i.e. k 32 j + 8 h .. + 8 of the piece
// for the lane with slot j —; a lane's 32 k lie in one 64-element block, whose 16-entry value table is built when the piece's first
// chunk is parked and kept for the other three.  D = 8 (two pieces per loop iteration, one in flight while the other is consumed).
template <int MT, int NT, int D, int CK, int WQ>
__global__ __launch_bounds__(KW * 64, (WQ == 1 && MT == 4 && NT == 2) ? 4 : 1) void stream_kloop_kernel(const bf16_t* __restrict__ Y, int64_t y_lo,
                                                               const void* __restrict__ Wv, const float* __restrict__ wscale,
                                                               float* __restrict__ P, int64_t p_slab, int M, int Ntot, int K,
                                                               int nz, KGroups g) {
  constexpr bool W8 = WQ == 1, W4 = WQ == 2;
  constexpr int PC = W4 ? 4 : (W8 ? 2 : 1);                           // 64-k chunks one weight load (piece) serves
  static_assert(!W8 || (CK == 64 && D == 4), "fp8 weight pieces span two 64-k chunks, two pieces in flight");
  static_assert(!W4 || (CK == 64 && D == 8), "NF4 weight pieces span four 64-k chunks, two pieces per loop iteration");
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(Wv);
  const uint8_t* __restrict__ Wq = reinterpret_cast<const uint8_t*>(Wv);
  extern __shared__ __attribute__((aligned(16))) char lds[];          // x: [2 bufs][2 x 16MT rows][256 B]; w: [KW][16 NT][256 B]
  constexpr int ROWB = CK * 2;                                        // bytes per LDS row of a chunk
  constexpr int SPR = CK / 8;                                         // 16-byte slots per row (16 or 8)
  constexpr int XR = 16 * MT;                                         // x rows per half (hi / lo)
  constexpr int XB = 2 * XR * ROWB;                                   // bytes per x buffer
  constexpr int XJ = (2 * XR * SPR + KW * 64 - 1) / (KW * 64);        // x pieces per thread and chunk
  int row0 = 0, nrows = M;
  if (g.off) {
    row0 = g.off[blockIdx.z];
    nrows = g.off[blockIdx.z + 1] - row0;
    if (nrows <= g.row_lo || nrows > g.row_hi) return;
    W += (int64_t)blockIdx.z * g.w_stride;
    Wq += W4 ? ((int64_t)blockIdx.z * g.w_stride) >> 1 : (int64_t)blockIdx.z * g.w_stride;
    if constexpr (WQ != 0) wscale += (int64_t)blockIdx.z * g.s_stride;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  char* wt = lds + 2 * XB + wave * (16 * NT) * ROWB;
  // K-range of this workgroup: chunks [c0, c0 + nc), balanced over the nz ranges
  const int nch = (K + CK - 1) / CK;
  const int z = blockIdx.y;
  int c0, nc;
  if constexpr (PC > 1) {                                             // ranges of whole PIECES = PC chunks (a partial last piece
    const int np = (K + PC * 64 - 1) / (PC * 64), base = np / nz, rem = np % nz;   // still has all its chunks: each covers 8 of every lane's k)
    c0 = PC * (z * base + min(z, rem));
    nc = PC * (base + (z < rem ? 1 : 0));
  } else {
    const int base = nch / nz, rem = nch % nz;
    c0 = z * base + min(z, rem);
    nc = base + (z < rem ? 1 : 0);
  }
  const int ntiles = (Ntot + 15) >> 4;
  const int t = (blockIdx.x * KW + wave) * NT;                        // first of this wave's NT adjacent tiles
  const bool active = t < ntiles;                                     // idle waves still help with x and the barriers
  const int mtn = (nrows + 15) >> 4;                                  // 16-row x tiles actually populated

  // ---- x pieces of this thread: piece p = tid + j * 512 -> row p / SPR (hi rows, then lo rows), slot p % SPR
  const bf16_t* xp[XJ];
  int xo[XJ];                                                         // LDS byte offset inside a buffer, -1 = nothing to write
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int p = tid + j * (KW * 64);
    const int rr = p / SPR, slot = p % SPR;
    const int h = rr >= XR ? 1 : 0, m = rr - h * XR;
    xo[j] = (rr < 2 * XR && m < mtn * 16) ? cslot<CK>(h * XR + m, slot) : -1;
    xp[j] = nullptr;
    if (rr < 2 * XR && m < nrows) {
      const int xr = g.xrows ? g.xrows[row0 + m] : row0 + m;
      xp[j] = Y + h * y_lo + (int64_t)xr * K + slot * (8 * PC);   // fp8 / NF4: slot j of a chunk = k 8 PC j (+ 8 h for chunk h) of its piece
    }
  }
  // ---- weight pieces of this wave: instruction i -> row i * RPI + lane / SPR of its 16 NT rows, slot lane % SPR
  // (fp8: 16 bytes per lane = slot lane % 8 of BOTH chunks of the piece)
  constexpr int RPI = 64 / SPR;                                       // rows per instruction (4 x 256 B or 8 x 128 B)
  constexpr int WI = 16 * NT / RPI;
  constexpr int DW = D / PC;                                          // ring depth in weight loads
  constexpr int XD = PC > 1 ? 2 : D;                                  // ring depth of the x chunks
  const bf16_t* wp[WI];
  const uint8_t* wq[WI];
  const float* wa[WI];                                                // NF4: the row's absmax table
  float wsc[WI];
  int wo[WI];
  const int wslot_k = (lane % SPR) * (8 * PC);
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = i * RPI + lane / SPR;
    const int n = min(t * 16 + row, Ntot - 1);
    wp[i] = W + (int64_t)n * K;
    wq[i] = Wq + (W4 ? ((int64_t)n * K) >> 1 : (int64_t)n * K);
    wa[i] = W4 ? wscale + (int64_t)n * (K >> 6) : nullptr;
    wo[i] = cslot<CK>(row, lane % SPR);
    wsc[i] = (W8 && g.wf == MN_W_INT8) ? wscale[n] : 1.0f;             // int8: the row scale rides the conversion (w8_codec.h)
  }
  u32x4 xr_[XD][XJ], wr_[DW][WI];
  float wa_[W4 ? DW : 1][WI];                                         // NF4: absmax of the block every load of the piece lies in
  Nf4Tab tab[W4 ? WI : 1];
  auto load_x = [&](u32x4 (&dst)[XJ], int c) {
    // bf16: chunk c = k [(c0 + c) 64, + 64); fp8: piece (c0 + c) / 2 (c0 is even), the lanes' first / second 8 k for even / odd c
    const int k = PC > 1 ? (c0 + (c & ~(PC - 1))) * CK + (c & (PC - 1)) * 8 : (c0 + c) * CK;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      dst[j] = u32x4{0u, 0u, 0u, 0u};                                  // rows >= nrows and k >= K stay zero
      if (xp[j] && k + ((tid + j * (KW * 64)) % SPR) * (8 * PC) < K) dst[j] = *reinterpret_cast<const u32x4*>(xp[j] + k);
    }
  };
  // bf16: chunk c of the K-range; fp8: piece c = chunks 2c, 2c + 1
  auto load_w = [&](u32x4 (&dst)[WI], float (&dsta)[WI], int c) {
    if constexpr (W4) {
      const int k = min((c0 + 4 * c) * CK + wslot_k, K - 32);         // beyond K the x image is zero: any finite value will do
#pragma unroll
      for (int i = 0; i < WI; ++i) {
        dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wq[i] + (k >> 1)));
        dsta[i] = wa[i][k >> 6];
      }
    } else if constexpr (W8) {
      const int k = min((c0 + 2 * c) * CK + wslot_k, K - 16);         // beyond K the x image is zero: any finite value will do
#pragma unroll
      for (int i = 0; i < WI; ++i) dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wq[i] + k));
    } else {
      const int k = min((c0 + c) * CK + wslot_k, K - 8);              // beyond K the x image is zero: any finite value will do
#pragma unroll
      for (int i = 0; i < WI; ++i) dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[i] + k));
    }
  };
  auto store_x = [&](const u32x4 (&src)[XJ], int buf) {
#pragma unroll
    for (int j = 0; j < XJ; ++j)
      if (xo[j] >= 0) *reinterpret_cast<u32x4*>(lds + buf * XB + xo[j]) = src[j];
  };
  // ---- prologue: D chunks of x and of weights in flight; x chunk 0 becomes visible
#pragma unroll
  for (int d = 0; d < XD; ++d)
    if (d < nc) load_x(xr_[d], d);
  const int nwl = (nc + PC - 1) / PC;                                // weight loads of this K-range
  if (active) {
#pragma unroll
    for (int d = 0; d < DW; ++d)
      if (d < nwl) load_w(wr_[d], wa_[W4 ? d : 0], d);
  }
  store_x(xr_[0], 0);
  if (XD < nc) load_x(xr_[0], XD);
  __syncthreads();

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int cb = 0; cb < nc; cb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int c = cb + d;
      if (c < nc) {
        const int buf = d & 1;                                        // D is even: chunk parity == d parity
        if (active) {
          // park the landed weight chunk, refill its registers with the chunk D ahead
          if constexpr (W4) {
            // piece c / 4 sits in ring slot d / 4 (cb is a multiple of D = 8): dword d & 3 of every load is this chunk's slot
            if ((d & 3) == 0) {
#pragma unroll
              for (int i = 0; i < WI; ++i) tab[i] = nf4_table(wa_[d >> 2][i]);
            }
#pragma unroll
            for (int i = 0; i < WI; ++i) {
              const u32x4 q = wr_[d >> 2][i];
              const uint32_t x = (d & 3) == 0 ? q.x : ((d & 3) == 1 ? q.y : ((d & 3) == 2 ? q.z : q.w));
              *reinterpret_cast<u32x4*>(wt + wo[i]) = nf4x8_to_bf16(tab[i], x);
            }
            if ((d & 3) == 3 && (c >> 2) + DW < nwl) load_w(wr_[d >> 2], wa_[d >> 2], (c >> 2) + DW);
          } else if constexpr (W8) {
            // piece c / 2 sits in ring slot d / 2 (cb is a multiple of D): its first 8 k per lane for the even chunk, the second
            // 8 for the odd one, converted to one bf16 slot; after the odd chunk the registers take the piece DW ahead
            auto park8 = [&](auto i8) {
              constexpr bool I8 = decltype(i8)::value;
#pragma unroll
              for (int i = 0; i < WI; ++i) {
                const u32x4 q = wr_[d >> 1][i];
                *reinterpret_cast<u32x4*>(wt + wo[i]) = (d & 1) ? w8x8_to_bf16<I8>(q.z, q.w, wsc[i]) : w8x8_to_bf16<I8>(q.x, q.y, wsc[i]);
              }
            };
            if (g.wf == MN_W_INT8) park8(std::true_type{}); else park8(std::false_type{});    // one scalar branch per chunk
            if ((d & 1) && (c >> 1) + DW < nwl) load_w(wr_[d >> 1], wa_[0], (c >> 1) + DW);
          } else {
#pragma unroll
            for (int i = 0; i < WI; ++i) *reinterpret_cast<u32x4*>(wt + wo[i]) = wr_[d][i];
            if (c + D < nc) load_w(wr_[d], wa_[0], c + D);
          }
          // CK / 32 MFMA steps of 32 k against every populated x tile (hi and lo)
#pragma unroll
          for (int s = 0; s < CK / 32; ++s) {
            bf16x8 w[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w[nt] = *reinterpret_cast<const bf16x8*>(wt + cslot<CK>(nt * 16 + fr, s * 4 + fq));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              if (mt < mtn) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(lds + buf * XB + cslot<CK>(mt * 16 + fr, s * 4 + fq));
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(lds + buf * XB + cslot<CK>(XR + mt * 16 + fr, s * 4 + fq));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                  acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w[nt], acc[mt][nt], 0, 0, 0);
                  acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w[nt], acc[mt][nt], 0, 0, 0);
                }
              }
            }
          }
        }
        // next x chunk into the other buffer (its last readers passed the previous barrier), ring refilled
        if (c + 1 < nc) {
          const int dn = (d + 1) % XD;
          store_x(xr_[dn], buf ^ 1);
          if (c + 1 + XD < nc) load_x(xr_[dn], c + 1 + XD);
        }
        __syncthreads();
      }
    }
  }
  // D layout: row m = fq*4 + r, col n = tile*16 + fr
  if (active) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nn = (t + nt) * 16 + fr;
      if (nn < Ntot) {
        float rs = 1.0f;
        if constexpr (W8) rs = g.wf == MN_W_INT8 ? 1.0f : wscale[nn];      // (int8 products are already scaled)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + fq * 4 + r;
            if (m < nrows) P[(int64_t)z * p_slab + (int64_t)(row0 + m) * Ntot + nn] = W8 ? acc[mt][nt][r] * rs : acc[mt][nt][r];
          }
        }
      }
    }
  }
}

int g_kl_nz = 0, g_kl_depth = 0, g_kl_nt = 0, g_kl_ck = 0;

// k per chunk: 64 (x double buffer 32 KiB + weight tiles 32 KiB at 64 rows: two workgroups fit a CU, also one of another
// stream's kernels) measured 27.0 vs 28.0 us on RF w12 and 17.2 vs 20.2 us on w3 against 128
int kloop_ck() { return g_kl_ck > 0 ? g_kl_ck : 64; }

// weight tiles per wave: two above 32 rows (halves the x fragment reads per weight byte, which bound the 64-row regime)
int kloop_nt(int max_rows) { return g_kl_nt > 0 ? g_kl_nt : (max_rows > 32 ? 2 : 1); }

// number of K-ranges: enough workgroups to fill `slots` CUs-worth of residency, never more ranges than chunks (fp8: chunk pairs)
static inline int kloop_piece_k(int wfmt) { return wfmt == MN_W_NF4 ? 256 : (wfmt ? 128 : 0); }   // k per weight load (0: one chunk)
int kloop_nz(int Ntot, int K, int slots, int nt, int piece_k = 0) {
  const int tb = (int)mn_cdiv(mn_cdiv(Ntot, 16), KW * nt);
  const int nch = (int)mn_cdiv(K, piece_k ? piece_k : kloop_ck());
  if (g_kl_nz > 0) return g_kl_nz < nch ? g_kl_nz : nch;
  int nz = slots / tb;
  if (nz < 1) nz = 1;
  if (nz > nch) nz = nch;
  return nz;
}

template <int MT, int NT, int D, int CK, int W8>
void kloop_launch(int G, int nz, const bf16_t* Y, int64_t y_lo, const void* W, const float* wscale, float* P, int64_t p_slab, int M,
                  int Ntot, int K, const KGroups& g, hipStream_t st) {
  constexpr int ROWB = CK * 2;
  const size_t lds = (size_t)2 * 2 * 16 * MT * ROWB + (size_t)KW * 16 * NT * ROWB;
  static bool opted = false;
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kloop_kernel<MT, NT, D, CK, W8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    opted = true;
  }
  const int tb = (int)mn_cdiv(mn_cdiv(Ntot, 16), KW * NT);
  hipLaunchKernelGGL((stream_kloop_kernel<MT, NT, D, CK, W8>), dim3(tb, nz, G), dim3(KW * 64), lds, st, Y, y_lo, W, wscale, P, p_slab,
                     M, Ntot, K, nz, g);
}

template <int MT>
void kloop_launch_d(int nt, int G, int nz, const bf16_t* Y, int64_t y_lo, const void* W, const float* wscale, float* P, int64_t p_slab,
                    int M, int Ntot, int K, const KGroups& g, hipStream_t st) {
  if (wscale && g.wf == MN_W_NF4) {     // NF4: 256-k weight pieces = four chunks
    if (nt == 2) kloop_launch<MT, 2, 8, 64, 2>(G, nz, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, g, st);
    else kloop_launch<MT, 1, 8, 64, 2>(G, nz, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, g, st);
    return;
  }
  if (wscale) {     // fp8 weights: 64-k x chunks, 128-k weight pieces, two pieces (= four chunks) in flight per wave
    if (nt == 2) kloop_launch<MT, 2, 4, 64, 1>(G, nz, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, g, st);
    else kloop_launch<MT, 1, 4, 64, 1>(G, nz, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, g, st);
    return;
  }
  const int depth = g_kl_depth > 0 ? g_kl_depth : (nt == 2 ? 2 : 4);
  const int ck = kloop_ck(); (void)ck;
#ifdef MN_DEV_HOOKS      // the 128-k chunk form exists for the A/B hook only (it spills at four row tiles): not in the product library
#define MN_KL(NT_, D_)                                                                                      \
  do {                                                                                                      \
    if (ck == 64) kloop_launch<MT, NT_, D_, 64, 0>(G, nz, Y, y_lo, W, nullptr, P, p_slab, M, Ntot, K, g, st);           \
    else kloop_launch<MT, NT_, D_, 128, 0>(G, nz, Y, y_lo, W, nullptr, P, p_slab, M, Ntot, K, g, st);                   \
  } while (0)
#else
#define MN_KL(NT_, D_) kloop_launch<MT, NT_, D_, 64, 0>(G, nz, Y, y_lo, W, nullptr, P, p_slab, M, Ntot, K, g, st)
#endif
  if (nt == 2) { if (depth == 2) MN_KL(2, 2); else MN_KL(2, 4); }
  else { if (depth == 2) MN_KL(1, 2); else MN_KL(1, 4); }
#undef MN_KL
}

int kloop_dispatch(int G, int max_rows, int nz, const bf16_t* Y, int64_t y_lo, const void* W, const float* wscale, float* P,
                   int64_t p_slab, int M, int Ntot, int K, const KGroups& g, hipStream_t st) {
  const int nt = kloop_nt(max_rows);
  if (max_rows <= 16) kloop_launch_d<1>(nt, G, nz, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, g, st);
  else if (max_rows <= 32) kloop_launch_d<2>(nt, G, nz, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, g, st);
  else kloop_launch_d<4>(nt, G, nz, Y, y_lo, W, wscale, P, p_slab, M, Ntot, K, g, st);
  return nz;
}

}  // namespace

#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_stream_kloop_tune(int nz, int depth, int nt) { g_kl_nz = nz; g_kl_depth = depth & 15; g_kl_nt = nt; g_kl_ck = (depth >> 4) ? 128 : 0; }
#endif

// CUs a dense launch asks for: matrices under 64 MB (QKV, dense, gate, semantic-decoder GEMVs, RF w3) take half the chip — fewer
// K-ranges to reduce, and other streams' kernels run beside them (end to end +3 % with three stream groups; a third or a
// quarter of the chip, or narrowing RF w12 as well, measured worse)
int g_kl_small_div = 2, g_kl_small_mb = 64;
int kloop_dense_slots(int Ntot, int K) {
  const int cus = mn_num_cus();
  return ((int64_t)Ntot * K * 2 < ((int64_t)g_kl_small_mb << 20) && g_kl_small_div > 1) ? cus / g_kl_small_div : cus;
}
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_stream_kloop_tune_small(int div) { g_kl_small_div = div & 15; if (div >> 4) g_kl_small_mb = div >> 4; }
#endif
extern "C" int mn_stream_kloop_slices(int M, int Ntot, int K) { return kloop_nz(Ntot, K, kloop_dense_slots(Ntot, K), kloop_nt(M)); }
// (the half-chip rule counts ELEMENTS: an fp8 matrix of the same shape keeps the bf16 form's share of the chip — with the byte
// count RF w12 fell under the threshold and ran 34.9 instead of 2x us at 48 rows)
extern "C" int mn_stream_kloop_wq_slices(int wfmt, int M, int Ntot, int K) {
  return kloop_nz(Ntot, K, kloop_dense_slots(Ntot, K), kloop_nt(M), kloop_piece_k(wfmt));
}
extern "C" int mn_stream_kloop_w8_slices(int M, int Ntot, int K) { return mn_stream_kloop_wq_slices(MN_W_FP8_E4M3, M, Ntot, K); }

// Dense: Y [2][M][K] bf16 (hi rows then lo rows), W [Ntot][K], P [nz][M][Ntot].  Returns nz (< 0: error).  M <= 64.
extern "C" int mn_stream_kloop(const uint16_t* Y, const uint16_t* W, float* P, int M, int Ntot, int K, void* stream) {
  MN_CHECK_ARG(Y && W && P && M >= 1 && M <= 64 && Ntot >= 1 && K >= 8 && (K % 8) == 0, "mn_stream_kloop: bad args");
  const int nz = kloop_nz(Ntot, K, kloop_dense_slots(Ntot, K), kloop_nt(M));
  const KGroups g{nullptr, nullptr, 0, 0, 1 << 30, 0, 0};
  kloop_dispatch(1, M, nz, Y, (int64_t)M * K, W, nullptr, P, (int64_t)M * Ntot, M, Ntot, K, g, mn_stream(stream));
  MN_CHECK_LAUNCH("mn_stream_kloop");
  return nz;
}

// Dense on 8-bit weights (wfmt = MN_W_FP8_E4M3 | MN_W_INT8): Wq bytes [Ntot][K] (K % 16 == 0), wscale fp32 [Ntot].
// nz = mn_stream_kloop_w8_slices(M, Ntot, K).
extern "C" int mn_stream_kloop_wq(const uint16_t* Y, const uint8_t* Wq, const float* wscale, float* P, int M, int Ntot, int K, int wfmt, void* stream) {
  MN_CHECK_ARG(Y && Wq && wscale && P && M >= 1 && M <= 64 && Ntot >= 1 && K >= 16 && (K % 16) == 0 &&
                   (wfmt == MN_W_FP8_E4M3 || wfmt == MN_W_INT8 || (wfmt == MN_W_NF4 && (K % 64) == 0)),
               "mn_stream_kloop_wq: bad args");
  const int nz = kloop_nz(Ntot, K, kloop_dense_slots(Ntot, K), kloop_nt(M), kloop_piece_k(wfmt));
  const KGroups g{nullptr, nullptr, 0, 0, 1 << 30, 0, wfmt};
  kloop_dispatch(1, M, nz, Y, (int64_t)M * K, Wq, wscale, P, (int64_t)M * Ntot, M, Ntot, K, g, mn_stream(stream));
  MN_CHECK_LAUNCH("mn_stream_kloop_wq");
  return nz;
}

// Grouped form: processes the groups with more than row_lo rows (<= 64), writing `nz` K-range slabs (the caller's
// K-slice kernel covers the smaller groups with the same slab count).
extern "C" int mn_stream_kloop_grouped(const uint16_t* Y, int y_rows, const uint16_t* W, int64_t w_stride, float* P,
                                       int p_rows, const int32_t* off, const int32_t* xrows, int G, int max_rows, int row_lo,
                                       int nz, int Ntot, int K, void* stream) {
  MN_CHECK_ARG(Y && W && P && off && G >= 1 && max_rows >= 1 && max_rows <= 64 && Ntot >= 1 && K >= 8 && (K % 8) == 0 &&
                   nz >= 1 && nz <= (K + kloop_ck() - 1) / kloop_ck(),
               "mn_stream_kloop_grouped: bad args");
  const KGroups g{off, xrows, w_stride, row_lo, max_rows, 0, 0};
  kloop_dispatch(G, max_rows, nz, Y, (int64_t)y_rows * K, W, nullptr, P, (int64_t)p_rows * Ntot, 0, Ntot, K, g, mn_stream(stream));
  MN_CHECK_LAUNCH("mn_stream_kloop_grouped");
  return nz;
}

// Grouped form on 8-bit weights (group g: Wq + g * w_stride bytes, wscale + g * s_stride floats).
extern "C" int mn_stream_kloop_grouped_wq(const uint16_t* Y, int y_rows, const uint8_t* Wq, int64_t w_stride, const float* wscale,
                                          int64_t s_stride, float* P, int p_rows, const int32_t* off, const int32_t* xrows, int G,
                                          int max_rows, int nz, int Ntot, int K, int wfmt, void* stream) {
  MN_CHECK_ARG(Y && Wq && wscale && P && off && G >= 1 && max_rows >= 1 && max_rows <= 64 && Ntot >= 1 && K >= 16 && (K % 16) == 0 &&
                   (w_stride % 32) == 0 && nz >= 1 && nz <= (K + kloop_piece_k(wfmt) - 1) / kloop_piece_k(wfmt) &&
                   (wfmt == MN_W_FP8_E4M3 || wfmt == MN_W_INT8 || (wfmt == MN_W_NF4 && (K % 64) == 0)),
               "mn_stream_kloop_grouped_wq: bad args");
  const KGroups g{off, xrows, w_stride, 0, max_rows, s_stride, wfmt};
  kloop_dispatch(G, max_rows, nz, Y, (int64_t)y_rows * K, Wq, wscale, P, (int64_t)p_rows * Ntot, 0, Ntot, K, g, mn_stream(stream));
  MN_CHECK_LAUNCH("mn_stream_kloop_grouped_wq");
  return nz;
}

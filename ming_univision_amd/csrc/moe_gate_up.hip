// moe_gate_up.hip — router + expert gate/up projections of a 1-row decode step (text decode) in ONE launch:
//
//   xn = RMSNorm(h[b]; ln2)                                             (modeling_bailing_moe.py:1218-1221)
//   (ti, tw)[b] = top-k of softmax(xn · gate^T), shared experts appended  (BailingMoeGate.forward :505-520)
//   hmid[b][s][i] = silu(xn · Wg[ti[b, s]][i]) * (xn · Wu[ti[b, s]][i])   (BailingMoeMLP, :608-639, the selected + shared experts)
//
// Round 4/5 ran this as two launches: the one-workgroup-per-row router (8.6 us: one CU pulls the 256 KB of gate rows and does the
// top-k while 255 CUs idle) and the fp32-FMA pair launch (18.8 us for 92 MB).  Routing is cheap to REPEAT and expensive to WAIT for:
// here every workgroup routes its row itself — each wave holds the normalised row in registers (no LDS image, no block reduction: a
// wave reduces the row's square sum on its own), computes 8 of the 64 gate logits (the 256 KB of gate rows come from
// L2), one workgroup barrier, then every wave does the 64-lane softmax / top-k redundantly and wave s takes slot s — and streams
// `UNITS` = 6 hidden units of that expert (gate row + up row each: 192 registers of loads per lane, ALL in flight at once: the whole
// launch is one HBM round trip; one workgroup per CU, 235 workgroups for I = 1408).  (Letting the shared experts' waves stream before
// the routing would overlap a quarter of the bytes with it, but their 192 registers of loads and the routing waves' gate rows do not
// fit one register file: 660 B of scratch per lane.)
// (A first form — 2 units per workgroup, 704 workgroups, two per CU — was SLOWER than the two launches, 40 us per layer against 27: every
// workgroup re-reads the 256 KB of gate rows, 180 MB of L2 traffic for 92 MB of weights.)
// The logits' per-lane arithmetic and the tie rule (lowest expert id among equal probabilities) are moe_router_row_kernel's
// (decode_ops.hip); the row's square sum is reduced per wave here, per workgroup there — 1-ulp differences of the logits.
#include <type_traits>

#include "common.h"
#include "w8_codec.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GU_WAVES = 8;          // = slots per row (top_k + shared) at most
// hidden units per workgroup and wave (template UNITS): 6 x (gate, up) rows x K in flight per lane = 192 registers of bf16 pieces at H = 2048;
// the byte formats' pieces are half / a quarter as long: 12 units — two rows' launches then fit ONE round of workgroups (118 x 2 <= 256)
constexpr int GB = 8;                // gate rows per wave (64 experts over 8 waves), requested together: one L2 round trip

struct GateUpArgs {
  const float* h; int64_t ldh;                       // [batch][H] fp32
  const bf16_t* norm_w; float eps;
  const bf16_t* gate_w;                              // [E][H]
  const void* W; int64_t w_stride;                   // [E + S][2 I][H] bf16 (or e4m3 bytes): gate rows [0, I), up rows [I, 2 I)
  const float* wscale; int64_t wscale_stride;        // e4m3 / int8: one fp32 scale per weight row, [E + S][2 I]; NF4: one absmax per 64 k, [E + S][2 I][H / 64]
  int H, I, E, top_k, n_shared, norm_topk_prob;
  float* hmid; int64_t ld_hmid;                      // [batch][n_slot * I]
  int32_t* ti; float* tw; float* logits;             // [batch][n_slot], [batch][n_slot], [batch][E]
  const float* P; int nz; int64_t slab;               // the decoder chain's form: the row is h + sum of nz partial slabs P [nz][batch][H] (slab floats apart); h itself is
                                                      // NOT updated here (workgroups of a later round would read the updated row): the down projection adds the slabs to its residual
};

__device__ __forceinline__ float dot8(const u32x2 q, const float* x, float t) {       // eight e4m3 weights
  float w0[4], w1[4];
  fp8x4_to_f32(q.x, w0);
  fp8x4_to_f32(q.y, w1);
#pragma unroll
  for (int e = 0; e < 4; ++e) t = fmaf(w0[e], x[e], t);
#pragma unroll
  for (int e = 0; e < 4; ++e) t = fmaf(w1[e], x[4 + e], t);
  return t;
}
__device__ __forceinline__ float dot8(const u32x4 q, const float* x, float t) {
  t = fmaf(bf16lo_to_f32(q.x), x[0], t); t = fmaf(bf16hi_to_f32(q.x), x[1], t);
  t = fmaf(bf16lo_to_f32(q.y), x[2], t); t = fmaf(bf16hi_to_f32(q.y), x[3], t);
  t = fmaf(bf16lo_to_f32(q.z), x[4], t); t = fmaf(bf16hi_to_f32(q.z), x[5], t);
  t = fmaf(bf16lo_to_f32(q.w), x[6], t); t = fmaf(bf16hi_to_f32(q.w), x[7], t);
  return t;
}

template <int NCK, int WQ, int UNITS>                // H = NCK x 512; WQ 0: bf16 experts, 1: e4m3 bytes + row scales (on the K sums), 2: int8 (quanto), 3: NF4 (w8_codec.h)
__global__ __launch_bounds__(GU_WAVES * 64) void moe_gate_up_routed_kernel(const GateUpArgs a) {
  __shared__ float lg[64];
  const int tid = threadIdx.x, lane = tid & 63, s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, i0 = blockIdx.x * UNITS, H = a.H, I = a.I, E = a.E, n_slot = a.top_k + a.n_shared;
  const bool live = s < n_slot;
  // ---- the row, this lane's K positions (k = c 512 + lane 8 ..): every wave holds the whole row
  const float* xr = a.h + (int64_t)b * a.ldh;
  float x[NCK][8];
  u32x4 nw[NCK];
#pragma unroll
  for (int c = 0; c < NCK; ++c) {
    const int k = c * 512 + lane * 8;
    f4 lo = *reinterpret_cast<const f4*>(xr + k), hi = *reinterpret_cast<const f4*>(xr + k + 4);
    if (a.P) {                                       // (slab order, like llm_glue_kernel: the same bits as the glue launch it replaces)
      const float* pp = a.P + (int64_t)b * H + k;
      for (int z = 0; z < a.nz; ++z) { lo += *reinterpret_cast<const f4*>(pp + z * a.slab); hi += *reinterpret_cast<const f4*>(pp + z * a.slab + 4); }
    }
    x[c][0] = lo.x; x[c][1] = lo.y; x[c][2] = lo.z; x[c][3] = lo.w; x[c][4] = hi.x; x[c][5] = hi.y; x[c][6] = hi.z; x[c][7] = hi.w;
    nw[c] = *reinterpret_cast<const u32x4*>(a.norm_w + k);
  }
  // a piece = this lane's 8 consecutive weights of one row and chunk: 16 bytes of bf16, 8 of e4m3 / int8, 4 of NF4 codes
  typedef typename std::conditional<WQ == 0, u32x4, typename std::conditional<WQ == 3, uint32_t, u32x2>::type>::type wv;
  constexpr int BPP = WQ == 0 ? 16 : (WQ == 3 ? 4 : 8);
  wv wq[UNITS][2][NCK];
  float wsc[UNITS][2][WQ == 3 ? NCK : 1];             // int8: the row's scale; NF4: the absmax of the piece's 64-block; e4m3: the row scale (used on the sums)
  auto request = [&](int e) {                        // ALL of this wave's expert rows in one round trip
    const uint8_t* wb = static_cast<const uint8_t*>(a.W) + ((int64_t)e * a.w_stride >> 3) * BPP;
    const float* sb = WQ ? a.wscale + (int64_t)e * a.wscale_stride : nullptr;
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      const int i = min(i0 + u, I - 1);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int64_t row = (int64_t)g * I + i;
        const uint8_t* wr = wb + ((row * H) >> 3) * BPP;
#pragma unroll
        for (int c = 0; c < NCK; ++c) wq[u][g][c] = __builtin_nontemporal_load(reinterpret_cast<const wv*>(wr + (c * 64 + lane) * BPP));
        if constexpr (WQ == 3) {
#pragma unroll
          for (int c = 0; c < NCK; ++c) wsc[u][g][c] = sb[row * (H >> 6) + c * 8 + (lane >> 3)];
        } else if constexpr (WQ != 0) {
          wsc[u][g][0] = sb[row];
        }
      }
    }
  };
  auto dot_piece = [&](const wv& q, float sc, const float* xx, float t) {
    if constexpr (WQ == 0) return dot8(q, xx, t);
    else if constexpr (WQ == 1) return dot8(q, xx, t);                                  // e4m3 values; the row scale multiplies the sum
    else if constexpr (WQ == 2) return dot8(w8x8_to_bf16<true>(q.x, q.y, sc), xx, t);      // bf16(q * scale) per element (quanto)
    else return dot8(nf4x8_to_bf16(nf4_table(sc), q), xx, t);                            // bf16(NF4[code] * absmax) per element (bitsandbytes)
  };
  // ---- RMSNorm: the wave reduces the square sum of the row on its own (same value in every wave)
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCK; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += x[c][e] * x[c][e];
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)H + a.eps);
#pragma unroll
  for (int c = 0; c < NCK; ++c) {
    const uint32_t w4[4] = {nw[c].x, nw[c].y, nw[c].z, nw[c].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[c][2 * e] = x[c][2 * e] * rstd * bf16lo_to_f32(w4[e]);
      x[c][2 * e + 1] = x[c][2 * e + 1] * rstd * bf16hi_to_f32(w4[e]);
    }
  }
  // ---- gate logits: the 8 waves share the E experts (wave s: experts s, s + 8, ...), all rows of a wave requested together (the 256 KB
  // of gate rows are L2-resident after the first workgroup; the expert rows' registers are not live yet)
  {
    u32x4 gq[GB][NCK];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const int e = min(s + GU_WAVES * j, E - 1);
#pragma unroll
      for (int c = 0; c < NCK; ++c) gq[j][c] = *reinterpret_cast<const u32x4*>(a.gate_w + (int64_t)e * H + c * 512 + lane * 8);
    }
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < NCK; ++c) t = dot8(gq[j][c], x[c], t);
      t = wave_sum(t);
      if (lane == 0 && s + GU_WAVES * j < E) lg[s + GU_WAVES * j] = t;
    }
  }
  __syncthreads();
  // ---- softmax over E (fp32), iterative arg-max, ties -> lowest expert id: every wave, redundantly; wave s keeps slot s
  float sl = lane < E ? lg[lane] : -INFINITY;
  const float mx = wave_max(sl);
  float pr = lane < E ? __expf(sl - mx) : 0.f;
  const float denom = wave_sum(pr);
  pr = pr / denom;
  float cur = lane < E ? pr : -1.f, wsum = 0.f, myw = 0.f;
  int myidx = 0, slot_e = E + (s - a.top_k);
  for (int k = 0; k < a.top_k; ++k) {
    const float best = wave_max(cur);
    const int sel = __ffsll((long long)__ballot(cur == best)) - 1;
    if (lane == k) { myw = best; myidx = sel; }
    if (k == s) slot_e = sel;
    if (lane == sel) cur = -1.f;
    wsum += best;
  }
  if (blockIdx.x == 0 && s == 0) {                   // one workgroup per row publishes the routing (the down projection reads it)
    if (lane < E && a.logits) a.logits[(int64_t)b * E + lane] = sl;
    if (lane < a.top_k) {
      a.ti[(int64_t)b * n_slot + lane] = myidx;
      a.tw[(int64_t)b * n_slot + lane] = (a.norm_topk_prob && a.top_k > 1) ? myw / wsum : myw;
    } else if (lane < n_slot) {
      a.ti[(int64_t)b * n_slot + lane] = E + (lane - a.top_k);
      a.tw[(int64_t)b * n_slot + lane] = 1.0f;
    }
  }
  if (!live) return;
  request(slot_e);
  // ---- UNITS hidden units of expert slot_e: y = silu(xn . Wg[i]) * (xn . Wu[i])
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    float g = 0.f, up = 0.f;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      g = dot_piece(wq[u][0][c], wsc[u][0][WQ == 3 ? c : 0], x[c], g);
      up = dot_piece(wq[u][1][c], wsc[u][1][WQ == 3 ? c : 0], x[c], up);
    }
    g = wave_sum(g);
    up = wave_sum(up);
    if constexpr (WQ == 1) { g *= wsc[u][0][0]; up *= wsc[u][1][0]; }
    if (lane == 0 && i0 + u < I) a.hmid[(int64_t)b * a.ld_hmid + (int64_t)s * I + i0 + u] = silu_f(g) * up;
  }
}

}  // namespace

// Can the router + gate/up of this shape run as the one launch?  (bf16, e4m3, int8 or NF4 experts, <= 64 routed experts, one wave per slot, the row in registers)
bool moe_gate_up_ok(int wfmt, int H, int I, int E, int top_k, int n_shared) {
  return (wfmt == MN_W_BF16 || wfmt == MN_W_FP8_E4M3 || wfmt == MN_W_INT8 || wfmt == MN_W_NF4) && H >= 512 && H <= 2048 && (H % 512) == 0 &&
         I >= 1 && E >= 1 && E <= 64 && top_k >= 1 && top_k <= E && top_k + n_shared <= GU_WAVES;
}

int moe_gate_up_routed(int wfmt, const float* h, int64_t ldh, const bf16_t* norm_w, float eps, const bf16_t* gate_w, const void* W, int64_t w_stride,
                       const float* wscale, int64_t wscale_stride, int batch, int H, int I, int E, int top_k, int n_shared, int norm_topk_prob,
                       float* hmid, int64_t ld_hmid, int32_t* ti, float* tw, float* logits, const float* P, int nz, int64_t slab, void* stream) {
  MN_CHECK_ARG(h && norm_w && gate_w && W && hmid && ti && tw && batch >= 1 && moe_gate_up_ok(wfmt, H, I, E, top_k, n_shared) &&
                   (ldh % 4) == 0 && (((uintptr_t)W) & 15) == 0 && (((uintptr_t)gate_w) & 15) == 0 && (w_stride % 8) == 0 &&
                   (wfmt == MN_W_BF16 || wscale) && (!P || (nz >= 1 && ldh == H)), "moe_gate_up_routed: bad args");
  const GateUpArgs a{h, ldh, norm_w, eps, gate_w, W, w_stride, wscale, wscale_stride, H, I, E, top_k, n_shared, norm_topk_prob, hmid, ld_hmid, ti, tw, logits,
                     P, nz, slab};
  // two rows in a byte format: 12 units per wave — both rows' workgroups are resident at once (one round trip for the launch)
  const bool wide = batch >= 2 && wfmt != MN_W_BF16 && 2 * mn_cdiv(I, 12) <= 256;
  const dim3 grid((unsigned)mn_cdiv(I, wide ? 12 : 6), (unsigned)batch), block(GU_WAVES * 64);
  hipStream_t st = mn_stream(stream);
#define MN_GU2(WQ_, U_)                                                                                         \
  do {                                                                                                          \
    switch (H / 512) {                                                                                          \
      case 1: hipLaunchKernelGGL((moe_gate_up_routed_kernel<1, WQ_, U_>), grid, block, 0, st, a); break;        \
      case 2: hipLaunchKernelGGL((moe_gate_up_routed_kernel<2, WQ_, U_>), grid, block, 0, st, a); break;        \
      case 3: hipLaunchKernelGGL((moe_gate_up_routed_kernel<3, WQ_, U_>), grid, block, 0, st, a); break;        \
      default: hipLaunchKernelGGL((moe_gate_up_routed_kernel<4, WQ_, U_>), grid, block, 0, st, a); break;       \
    }                                                                                                           \
  } while (0)
#define MN_GU(WQ_) do { if (wide) MN_GU2(WQ_, 12); else MN_GU2(WQ_, 6); } while (0)
  if (wfmt == MN_W_FP8_E4M3) MN_GU(1); else if (wfmt == MN_W_INT8) MN_GU(2); else if (wfmt == MN_W_NF4) MN_GU(3); else MN_GU2(0, 6);
#undef MN_GU
#undef MN_GU2
  MN_CHECK_LAUNCH("moe_gate_up_routed");
  return MN_OK;
}

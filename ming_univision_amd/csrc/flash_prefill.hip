// flash_prefill.hip — flash attention of the batch / prefill passes (bf16 MFMA, fp32 online softmax), 64-key tiles.
//
//   SRC 0  flash_attn_func on packed qkv, head dim 64 (mingtok/vision_tower/layers/attention.py:78-108 full, :177-239 causal):
//          q / k / v bf16 inside qkv [B, T, 3, nh, 64]; a workgroup = 128 queries of one (image, head), 32 per wave.
//   SRC 1  BailingMoeFlashAttention2 on a long prompt (modeling_bailing_moe.py:848-1045; bottom-right causal, :844-846), head
//          dim 128, GQA 4:1: q bf16 [rows, n_q, 128] (RoPE'd, pre-scaled), K / V read from the fp32 KV arena the decode
//          kernels use; a workgroup = 32 queries x the 4 query heads of one KV head (one head per wave), so a staged K / V
//          tile serves 128 query rows in both forms; several prompts per launch through a (sequence, first row, length) table.
//
// Per 64-key tile, with q = lane & 15 (a query COLUMN), g = lane >> 4, two 16-query groups per wave:
//   S^T[key, q] = sum_d K[key, d] Q[q, d]      A = K fragment (ds_read_b128 from the row-major tile, rows padded by 16 B),
//                                              B = Q fragment (registers, loaded once)
//     -> accumulator reg r of key fragment f holds key 16 f + 4 g + r for query q
//   online softmax per query: max over 16 registers + xor-16 / xor-32 shuffles, p = 2^(s log2e - m log2e) (one fma + v_exp_f32),
//     running sums kept per lane (reduced once at the end), accumulators rescaled only when some query's max moved, and no mask
//     arithmetic on tiles every query of the wave sees whole (measured: the round-1 form spent 26 VALU + 13 SALU per MFMA)
//   O^T[d, q] += sum_key V[key, d] P[q, key]   B = P^T straight from those registers: MFMA k-slot (g, e) is DEFINED as key
//     32 s + (e < 4 ? 4 g + e : 16 + 4 g + e - 4); A = V^T fragment with the same slot map, delivered by the transposing LDS read
//     (ds_read_b64_tr_b16: lane i of a 16-lane group receives element (i & 3) of the 8-byte chunks addressed by lanes
//     i/4, i/4 + 4, i/4 + 8, i/4 + 12 — measured, tools/exp/tr_probe.hip) from the ROW-major V tile, so V is staged like K with
//     16-byte writes and never transposed by stores.  V rows are padded by 32 B: a 16-lane group reads 4 keys x 32 B and the
//     eight groups of two waves tile the 64 banks.
// The staging of tile t+1 is issued to registers before tile t is computed and written to LDS after it (one tile in flight).
//
// HILO (round 4; SRC 0 only): the fp32-class regime's attention — q / k / v are FP32 inside qkv, every operand is carried as a bf16
// hi + lo pair (x = hi + lo to 2^-17) and every product as three MFMAs (hi hi + hi lo + lo hi; the lo lo term is 2^-16 of the
// product), scores / softmax / accumulators fp32, output fp32 and / or the next GEMM's hi / lo operand.  It replaces the per-row
// decode kernels the regime borrowed before (every query row re-reading all of K and V through L2: 8.5 ms of the 14.8 ms a
// 1024^2 image spends in MingTok) with the tiled form: a staged 64-key tile serves 128 query rows.
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int FKT = 64;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
struct HL { uint32_t h, l; };
__device__ __forceinline__ HL spl(float a, float b) { HL r; split_pk_bf16(a, b, r.h, r.l); return r; }
typedef __attribute__((address_space(3))) v4s lds_v4s;

struct FlashP {
  const bf16_t* q; const void* k; const void* v; bf16_t* out;
  float* out_f32; bf16_t* out_split; int64_t split_lo_off;   // HILO: fp32 result and / or bf16 hi rows + lo rows (lo rows split_lo_off elements on)
  int64_t q_rs, q_hs, kv_rs, kv_hs, o_rs;          // element strides: rows and heads
  int64_t q_bs, kv_bs, o_bs;                       // SRC 0: per image
  int T, past, causal;
  const uint8_t* key_mask; int64_t mask_bs;        // optional [*, keys] (1 = attend); SRC 1: one row per table entry
  const int32_t* seq_tab; int64_t kv_seq_stride;   // SRC 1: [n][tab_w] = (cache sequence, first q / out row, span length[, past]); NULL: (0, 0, T)
  int tab_w;                                       // 3, or 4 = every span brings its own `past`
  int mask_per_row;                                // (unused with key_mask == NULL)
};

template <int HD, int SRC, bool HILO = false>
__global__ __launch_bounds__(256, 2) void flash_prefill_kernel(const FlashP p) {
  // query groups (16 queries each) per wave: the hi/lo form at head dim 128 keeps one (q, o, s, p hi + lo would not fit 256 registers)
  constexpr int QG = (HILO && HD == 128) ? 1 : 2, QW = 16 * QG;
  constexpr int KROW = HD + 8, VROW = HD + 16, NKK = HD / 32, NDT = HD / 16;
  constexpr int ESZ = (SRC == 0 && !HILO) ? 2 : 4, PPR = HD * ESZ / 16, KPP = 256 / PPR, NP = FKT / KPP;
  __shared__ __attribute__((aligned(16))) bf16_t ks[FKT * KROW * (HILO ? 2 : 1)];     // HILO: the hi tile, then the lo tile
  __shared__ __attribute__((aligned(16))) bf16_t vs[FKT * VROW * (HILO ? 2 : 1)];
  constexpr int KLO = FKT * KROW, VLO = FKT * VROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  int T = p.T, q0, q_hi_wg;
  int past = p.past;
  int64_t o_off;                                    // element offset of the (image | span, head)'s first output
  const bf16_t* qb;
  bf16_t* ob;
  const char *kb, *vb;
  const uint8_t* km = p.key_mask;
  if (SRC == 0) {
    const int head = blockIdx.y, b = blockIdx.z;
    q0 = blockIdx.x * 128 + wave * 32;
    q_hi_wg = blockIdx.x * 128 + 127;
    qb = p.q + (b * p.q_bs + head * p.q_hs) * (HILO ? 2 : 1);          // HILO: q is fp32 (pointer arithmetic in bf16 units)
    o_off = b * p.o_bs + head * HD;
    ob = p.out + o_off;
    kb = reinterpret_cast<const char*>(p.k) + (b * p.kv_bs + head * p.kv_hs) * ESZ;
    vb = reinterpret_cast<const char*>(p.v) + (b * p.kv_bs + head * p.kv_hs) * ESZ;
    if (km) km += b * p.mask_bs;
  } else {
    int seq = 0, r0 = 0;
    if (p.seq_tab) {
      const int32_t* te = p.seq_tab + blockIdx.z * p.tab_w;
      seq = te[0]; r0 = te[1]; T = te[2];
      if (p.tab_w > 3) past = te[3];
    }
    const int kvh = blockIdx.y, head = kvh * 4 + wave;
    q0 = blockIdx.x * QW;
    q_hi_wg = q0 + QW - 1;
    if (q0 >= T) return;                                   // uniform per workgroup
    qb = p.q + ((int64_t)r0 * p.q_rs + head * p.q_hs) * (HILO ? 2 : 1);
    o_off = (int64_t)r0 * p.o_rs + head * HD;
    ob = p.out + o_off;
    kb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(p.k) + seq * p.kv_seq_stride + kvh * p.kv_hs);
    vb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(p.v) + seq * p.kv_seq_stride + kvh * p.kv_hs);
    if (km) km += blockIdx.z * p.mask_bs;
  }
  const int k_total = past + T;
  const bool wave_on = q0 < T;
  const int wave_kmax = p.causal ? past + min(T, q0 + QW) : k_total;       // keys this wave's queries can see
  const int ntile = ((p.causal ? past + min(T, q_hi_wg + 1) : k_total) + FKT - 1) / FKT;

  // Q fragments (B operand): Q[q][d = 32 kk + 8 g .. +8]; SRC 0 scales by 64^-0.5 = 0.125 (exact in bf16)
  bf16x8 qf[QG][NKK], qfl[HILO ? QG : 1][HILO ? NKK : 1];
  int q_idx[QG];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    q_idx[qg] = q0 + qg * 16 + i16;
    if constexpr (HILO) {
      const float* qr = reinterpret_cast<const float*>(qb) + (int64_t)min(q_idx[qg], T - 1) * p.q_rs + g * 8;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        f32x4 a = *reinterpret_cast<const f32x4*>(qr + kk * 32), b = *reinterpret_cast<const f32x4*>(qr + kk * 32 + 4);
        if (SRC == 1) { a *= 8.0f; b *= 8.0f; }          // SRC 1: q arrives pre-scaled (the 0.125 below is SRC 0's 64^-0.5)
        u32x4 hi, lo;
        { const HL t_ = spl(a.x * 0.125f, a.y * 0.125f); hi.x = t_.h; lo.x = t_.l; } { const HL t_ = spl(a.z * 0.125f, a.w * 0.125f); hi.y = t_.h; lo.y = t_.l; }
        { const HL t_ = spl(b.x * 0.125f, b.y * 0.125f); hi.z = t_.h; lo.z = t_.l; } { const HL t_ = spl(b.z * 0.125f, b.w * 0.125f); hi.w = t_.h; lo.w = t_.l; }
        qf[qg][kk] = __builtin_bit_cast(bf16x8, hi);
        qfl[qg][kk] = __builtin_bit_cast(bf16x8, lo);
      }
      continue;
    }
    const bf16_t* qr = qb + (int64_t)min(q_idx[qg], T - 1) * p.q_rs + g * 8;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      u32x4 raw = *reinterpret_cast<const u32x4*>(qr + kk * 32);
      if (SRC == 0) {
        raw.x = cvt_pk_bf16(bf16lo_to_f32(raw.x) * 0.125f, bf16hi_to_f32(raw.x) * 0.125f);
        raw.y = cvt_pk_bf16(bf16lo_to_f32(raw.y) * 0.125f, bf16hi_to_f32(raw.y) * 0.125f);
        raw.z = cvt_pk_bf16(bf16lo_to_f32(raw.z) * 0.125f, bf16hi_to_f32(raw.z) * 0.125f);
        raw.w = cvt_pk_bf16(bf16lo_to_f32(raw.w) * 0.125f, bf16hi_to_f32(raw.w) * 0.125f);
      }
      qf[qg][kk] = __builtin_bit_cast(bf16x8, raw);
    }
  }
  f32x4 o[QG][NDT];
  float m_run[QG], l_run[QG];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    m_run[qg] = -INFINITY; l_run[qg] = 0.f;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[qg][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // staging: piece = 16 source bytes; a pass of the workgroup covers KPP whole key rows (coalesced)
  const int piece = tid % PPR, krow = tid / PPR;
  u32x4 rk[NP], rv[NP];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int kr = min(k0 + ps * KPP + krow, k_total - 1);
      const int64_t off = (int64_t)kr * p.kv_rs * ESZ + piece * 16;
      rk[ps] = *reinterpret_cast<const u32x4*>(kb + off);
      rv[ps] = *reinterpret_cast<const u32x4*>(vb + off);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int key = ps * KPP + krow;
      if constexpr (HILO) {
        const f32x4 a = __builtin_bit_cast(f32x4, rk[ps]), b = __builtin_bit_cast(f32x4, rv[ps]);
        u32x2 h, l;
        { const HL t_ = spl(a.x, a.y); h.x = t_.h; l.x = t_.l; } { const HL t_ = spl(a.z, a.w); h.y = t_.h; l.y = t_.l; }
        *reinterpret_cast<u32x2*>(&ks[key * KROW + piece * 4]) = h;
        *reinterpret_cast<u32x2*>(&ks[KLO + key * KROW + piece * 4]) = l;
        { const HL t_ = spl(b.x, b.y); h.x = t_.h; l.x = t_.l; } { const HL t_ = spl(b.z, b.w); h.y = t_.h; l.y = t_.l; }
        *reinterpret_cast<u32x2*>(&vs[key * VROW + piece * 4]) = h;
        *reinterpret_cast<u32x2*>(&vs[VLO + key * VROW + piece * 4]) = l;
      } else if (SRC == 0) {
        *reinterpret_cast<u32x4*>(&ks[key * KROW + piece * 8]) = rk[ps];
        *reinterpret_cast<u32x4*>(&vs[key * VROW + piece * 8]) = rv[ps];
      } else {
        const f32x4 a = __builtin_bit_cast(f32x4, rk[ps]), b = __builtin_bit_cast(f32x4, rv[ps]);
        *reinterpret_cast<u32x2*>(&ks[key * KROW + piece * 4]) = u32x2{cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(a.z, a.w)};
        *reinterpret_cast<u32x2*>(&vs[key * VROW + piece * 4]) = u32x2{cvt_pk_bf16(b.x, b.y), cvt_pk_bf16(b.z, b.w)};
      }
    }
  };
  // per-lane LDS addresses: K fragment row (lane & 15), V^T chunk (key 4 g + i/4, d 4 (i & 3))
  const bf16_t* kfr = &ks[i16 * KROW + g * 8];
  const bf16_t* vfr = &vs[(g * 4 + (i16 >> 2)) * VROW + 4 * (i16 & 3)];

  fetch(0);
  for (int kt = 0; kt < ntile; ++kt) {
    const int k0 = kt * FKT;
    __syncthreads();                       // the previous tile's fragments are read
    park();
    __syncthreads();                       // tile kt visible
    if (kt + 1 < ntile) fetch(k0 + FKT);   // in flight under this tile's MFMAs
    if (!wave_on || k0 >= wave_kmax) continue;

    f32x4 s[QG][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int qg = 0; qg < QG; ++qg) s[qg][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kfr + f * 16 * KROW + kk * 32);
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) s[qg][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qg][kk], s[qg][f], 0, 0, 0);
        if constexpr (HILO) {
          const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kfr + KLO + f * 16 * KROW + kk * 32);
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) {
            s[qg][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qfl[qg][kk], s[qg][f], 0, 0, 0);
            s[qg][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qf[qg][kk], s[qg][f], 0, 0, 0);
          }
        }
      }
    }
    // a tile every query of the wave sees whole needs no mask arithmetic (all but the diagonal / last tile of a span)
    const bool whole = k0 + FKT <= k_total && (!p.causal || k0 + FKT - 1 <= past + q0) && !km;
    if (!whole) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int kb4 = k0 + f * 16 + g * 4;                 // this lane's four keys of fragment f
        uint32_t live = 0xf;                                  // bit r: key kb4 + r exists and is not masked out
        if (km) {
          live = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) live |= (km[min(kb4 + r, k_total - 1)] != 0 ? 1u : 0u) << r;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb4 + r;
          const bool there = key < k_total && ((live >> r) & 1u);
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) {
            const bool ok = there && (!p.causal || key <= past + q_idx[qg]);
            s[qg][f][r] = ok ? s[qg][f][r] : -INFINITY;
          }
        }
      }
    }
    bf16x8 pf[QG][2], pfl[HILO ? QG : 1][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      float mx = fmaxf(fmaxf(s[qg][0][0], s[qg][0][1]), fmaxf(s[qg][0][2], s[qg][0][3]));
#pragma unroll
      for (int f = 1; f < 4; ++f) mx = fmaxf(mx, fmaxf(fmaxf(s[qg][f][0], s[qg][f][1]), fmaxf(s[qg][f][2], s[qg][f][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qg], mx);
      // exp(x - m) = 2^(x log2e - m log2e): one fma + v_exp_f32 per score; a row with no visible key yet keeps m = -inf, p = 0
      constexpr float LOG2E = 1.4426950408889634f;
      const float mb = m_new == -INFINITY ? 0.f : m_new * LOG2E;
      float e[16], psum = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          e[f * 4 + r] = __builtin_amdgcn_exp2f(fmaf(s[qg][f][r], LOG2E, -mb));
          psum += e[f * 4 + r];
        }
      if (__builtin_amdgcn_ballot_w64(m_new > m_run[qg]) != 0) {      // the running max moved for some query of the wave
        const float alpha = m_run[qg] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_run[qg] - m_new) * LOG2E);
        l_run[qg] *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[qg][dt] *= alpha;
      }
      l_run[qg] += psum;                       // this lane's 16 keys only: the four lane groups are summed after the last tile
      m_run[qg] = m_new;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        if constexpr (HILO) {
          u32x4 hi, lo;
          { const HL t_ = spl(e[sl * 8 + 0], e[sl * 8 + 1]); hi.x = t_.h; lo.x = t_.l; } { const HL t_ = spl(e[sl * 8 + 2], e[sl * 8 + 3]); hi.y = t_.h; lo.y = t_.l; }
          { const HL t_ = spl(e[sl * 8 + 4], e[sl * 8 + 5]); hi.z = t_.h; lo.z = t_.l; } { const HL t_ = spl(e[sl * 8 + 6], e[sl * 8 + 7]); hi.w = t_.h; lo.w = t_.l; }
          pf[qg][sl] = __builtin_bit_cast(bf16x8, hi);
          pfl[qg][sl] = __builtin_bit_cast(bf16x8, lo);
          continue;
        }
        const u32x4 pk = {cvt_pk_bf16(e[sl * 8 + 0], e[sl * 8 + 1]), cvt_pk_bf16(e[sl * 8 + 2], e[sl * 8 + 3]),
                          cvt_pk_bf16(e[sl * 8 + 4], e[sl * 8 + 5]), cvt_pk_bf16(e[sl * 8 + 6], e[sl * 8 + 7])};
        pf[qg][sl] = __builtin_bit_cast(bf16x8, pk);
      }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(vfr + (sl * 32) * VROW + dt * 16));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(vfr + (sl * 32 + 16) * VROW + dt * 16));
        const v8s vv = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const bf16x8 vf = __builtin_bit_cast(bf16x8, vv);
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) o[qg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qg][sl], o[qg][dt], 0, 0, 0);
        if constexpr (HILO) {
          const v4s llo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(vfr + VLO + (sl * 32) * VROW + dt * 16));
          const v4s lhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(vfr + VLO + (sl * 32 + 16) * VROW + dt * 16));
          const v8s vl = {llo.x, llo.y, llo.z, llo.w, lhi.x, lhi.y, lhi.z, lhi.w};
          const bf16x8 vfl = __builtin_bit_cast(bf16x8, vl);
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) {
            o[qg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfl[qg][sl], o[qg][dt], 0, 0, 0);
            o[qg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfl, pf[qg][sl], o[qg][dt], 0, 0, 0);
          }
        }
      }
  }
  if (!wave_on) return;
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    float l = l_run[qg];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (q_idx[qg] >= T) continue;
    const float inv = l > 0.f ? 1.0f / l : 0.f;   // no attended key: 0, not NaN
    if constexpr (HILO) {
      const int64_t eo = o_off + (int64_t)q_idx[qg] * p.o_rs + g * 4;      // element offset of this lane's first output
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const f32x4 v = o[qg][dt] * inv;
        if (p.out_f32) *reinterpret_cast<f32x4*>(p.out_f32 + eo + dt * 16) = v;
        if (p.out_split) {
          u32x2 h, lo2;
          { const HL t_ = spl(v.x, v.y); h.x = t_.h; lo2.x = t_.l; } { const HL t_ = spl(v.z, v.w); h.y = t_.h; lo2.y = t_.l; }
          *reinterpret_cast<u32x2*>(p.out_split + eo + dt * 16) = h;
          *reinterpret_cast<u32x2*>(p.out_split + p.split_lo_off + eo + dt * 16) = lo2;
        }
      }
      continue;
    }
    bf16_t* op = ob + (int64_t)q_idx[qg] * p.o_rs + g * 4;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const u32x2 pk = {cvt_pk_bf16(o[qg][dt][0] * inv, o[qg][dt][1] * inv), cvt_pk_bf16(o[qg][dt][2] * inv, o[qg][dt][3] * inv)};
      *reinterpret_cast<u32x2*>(op + dt * 16) = pk;
    }
  }
}
}  // namespace

// flash_attn_func on packed qkv bf16 [B, T, 3, nh, 64] -> out bf16 [B, T, nh * 64]  (mingtok attention.py:78-108, 177-239)
extern "C" int mn_attn_prefill_hd64(const uint16_t* qkv, uint16_t* out, int B, int T, int n_heads, int causal, void* stream) {
  MN_CHECK_ARG(qkv && out && B >= 1 && T >= 1 && n_heads >= 1, "mn_attn_prefill_hd64: bad args");
  FlashP p{};
  const int64_t rs = (int64_t)3 * n_heads * 64;
  p.q = qkv; p.k = qkv + (int64_t)n_heads * 64; p.v = qkv + (int64_t)2 * n_heads * 64; p.out = out;
  p.q_rs = rs; p.q_hs = 64; p.kv_rs = rs; p.kv_hs = 64; p.o_rs = (int64_t)n_heads * 64;
  p.q_bs = (int64_t)T * rs; p.kv_bs = (int64_t)T * rs; p.o_bs = (int64_t)T * n_heads * 64;
  p.T = T; p.past = 0; p.causal = causal;
  hipLaunchKernelGGL((flash_prefill_kernel<64, 0>), dim3(mn_cdiv(T, 128), n_heads, B), dim3(256), 0, mn_stream(stream), p);
  MN_CHECK_LAUNCH("mn_attn_prefill_hd64");
  return MN_OK;
}

// The fp32-class regime's form: qkv FP32 [B, T, 3, nh, 64] -> out fp32 [B * T, nh * 64] and / or split bf16 [2][B * T][nh * 64]
// (hi rows, then lo rows: the projection GEMM's operand).  Operands as bf16 hi + lo pairs, three MFMAs per product.
extern "C" int mn_attn_prefill_hd64_f32(const float* qkv, float* out, uint16_t* split, int B, int T, int n_heads, int causal, void* stream) {
  MN_CHECK_ARG(qkv && (out || split) && B >= 1 && T >= 1 && n_heads >= 1, "mn_attn_prefill_hd64_f32: bad args");
  FlashP p{};
  const int64_t rs = (int64_t)3 * n_heads * 64;
  p.q = reinterpret_cast<const bf16_t*>(qkv); p.k = qkv + (int64_t)n_heads * 64; p.v = qkv + (int64_t)2 * n_heads * 64;
  p.out = nullptr; p.out_f32 = out; p.out_split = split; p.split_lo_off = (int64_t)B * T * n_heads * 64;
  p.q_rs = rs; p.q_hs = 64; p.kv_rs = rs; p.kv_hs = 64; p.o_rs = (int64_t)n_heads * 64;
  p.q_bs = (int64_t)T * rs; p.kv_bs = (int64_t)T * rs; p.o_bs = (int64_t)T * n_heads * 64;
  p.T = T; p.past = 0; p.causal = causal;
  hipLaunchKernelGGL((flash_prefill_kernel<64, 0, true>), dim3(mn_cdiv(T, 128), n_heads, B), dim3(256), 0, mn_stream(stream), p);
  MN_CHECK_LAUNCH("mn_attn_prefill_hd64_f32");
  return MN_OK;
}

// one span against one cache sequence (kv_seq [2, n_kv, t_max, 128]): the form mn_attn_prefill_gqa_hd128 forwards to
extern "C" int mn_flash_prefill_gqa_hd128_one(const uint16_t* q, const float* kv_seq, int64_t t_max, int n_q, int n_kv, int past, int T,
                                              const uint8_t* key_mask, uint16_t* out, void* stream) {
  FlashP p{};
  p.q = q; p.k = kv_seq; p.v = kv_seq + (int64_t)n_kv * t_max * 128; p.out = out;
  p.q_rs = (int64_t)n_q * 128; p.q_hs = 128; p.kv_rs = 128; p.kv_hs = t_max * 128; p.o_rs = (int64_t)n_q * 128;
  p.T = T; p.past = past; p.causal = 1; p.key_mask = key_mask;
  hipLaunchKernelGGL((flash_prefill_kernel<128, 1>), dim3(mn_cdiv(T, 32), n_kv, 1), dim3(256), 0, mn_stream(stream), p);
  MN_CHECK_LAUNCH("mn_attn_prefill_gqa_hd128");
  return MN_OK;
}

// The fp32-class form of mn_flash_prefill_gqa_hd128 (the wide-route prefill of mn_llm_step_spans): q FP32 [rows, n_q, 128] (RoPE'd,
// pre-scaled), K / V from the fp32 arena, all operands as bf16 hi + lo pairs (three MFMAs per product); span table [n_spans][4] =
// (cache sequence, first q / out row, span length, past); out fp32 [rows, n_q * 128] and / or split bf16 hi rows + lo rows
// (lo rows split_lo_off elements after the hi rows).  No key mask.
extern "C" int mn_flash_prefill_gqa_hd128_f32(const float* q, const float* kv_layer, int64_t t_max, int n_q, int n_kv, const int32_t* span_tab,
                                              int n_spans, int max_len, float* out, uint16_t* split, int64_t split_lo_off, void* stream) {
  MN_CHECK_ARG(q && kv_layer && (out || split) && span_tab && n_spans >= 1 && max_len >= 1 && max_len <= t_max && n_kv >= 1 && n_q == 4 * n_kv,
               "mn_flash_prefill_gqa_hd128_f32: bad args (n_q must be 4 n_kv)");
  FlashP p{};
  p.q = reinterpret_cast<const bf16_t*>(q); p.k = kv_layer; p.v = kv_layer + (int64_t)n_kv * t_max * 128;
  p.out_f32 = out; p.out_split = split; p.split_lo_off = split_lo_off;
  p.q_rs = (int64_t)n_q * 128; p.q_hs = 128; p.kv_rs = 128; p.kv_hs = t_max * 128; p.o_rs = (int64_t)n_q * 128;
  p.T = max_len; p.past = 0; p.causal = 1;
  p.seq_tab = span_tab; p.tab_w = 4; p.kv_seq_stride = (int64_t)2 * n_kv * t_max * 128;
  hipLaunchKernelGGL((flash_prefill_kernel<128, 1, true>), dim3(mn_cdiv(max_len, 16), n_kv, n_spans), dim3(256), 0, mn_stream(stream), p);
  MN_CHECK_LAUNCH("mn_flash_prefill_gqa_hd128_f32");
  return MN_OK;
}

// GQA 4:1 flash attention, head dim 128, of one or several prompt spans against the fp32 KV arena of one layer
// (kv_layer [n_seq_total, 2, n_kv, t_max, 128]; span i = rows [r0_i, r0_i + len_i) of q / out, keys [0, past + len_i) of cache
// sequence seq_i, bottom-right causal).  seq_tab: device int32 [n_spans][3] = (seq_i, r0_i, len_i); max_len >= every len_i.
// key_mask: optional uint8 [n_spans, mask_stride] (1 = attend).
extern "C" int mn_flash_prefill_gqa_hd128(const uint16_t* q, const float* kv_layer, int64_t t_max, int n_q, int n_kv, int past,
                                          const int32_t* seq_tab, int n_spans, int max_len, const uint8_t* key_mask,
                                          int64_t mask_stride, uint16_t* out, void* stream) {
  MN_CHECK_ARG(q && kv_layer && out && seq_tab && n_spans >= 1 && max_len >= 1 && past >= 0 && past + max_len <= t_max && n_kv >= 1 &&
                   n_q == 4 * n_kv, "mn_flash_prefill_gqa_hd128: bad args (n_q must be 4 n_kv)");
  FlashP p{};
  p.q = q; p.k = kv_layer; p.v = kv_layer + (int64_t)n_kv * t_max * 128; p.out = out;
  p.q_rs = (int64_t)n_q * 128; p.q_hs = 128; p.kv_rs = 128; p.kv_hs = t_max * 128; p.o_rs = (int64_t)n_q * 128;
  p.T = max_len; p.past = past; p.causal = 1; p.key_mask = key_mask; p.mask_bs = mask_stride;
  p.seq_tab = seq_tab; p.tab_w = 3; p.kv_seq_stride = (int64_t)2 * n_kv * t_max * 128;
  hipLaunchKernelGGL((flash_prefill_kernel<128, 1>), dim3(mn_cdiv(max_len, 32), n_kv, n_spans), dim3(256), 0, mn_stream(stream), p);
  MN_CHECK_LAUNCH("mn_flash_prefill_gqa_hd128");
  return MN_OK;
}

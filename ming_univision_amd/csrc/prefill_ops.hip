// prefill_ops.hip — many-token (prefill) operators of the Bailing-MoE decoder on the bf16 MFMA path (gfx950):
//   mn_rmsnorm_bf16            BailingMoeRMSNorm, fp32 residual in -> bf16 GEMM operand out
//   mn_rope_kv_prefill         rotary on q/k of T new tokens, k/v appended to the fp32 KV arena, q -> bf16
//   mn_attn_prefill_gqa_hd128  flash attention, head_dim 128, GQA, bottom-right causal + arbitrary key mask,
//                              keys/values read from the fp32 KV arena (past context included)
//   mn_moe_topk_logits         softmax + top-k + renormalise over precomputed gate logits (T tokens)
//   mn_moe_sort / mn_gather_rows_bf16 / mn_moe_combine   token <-> expert permutation around the grouped GEMM
// Replaces, for q_len > 16: modeling_bailing_moe.py:131-136 (RMSNorm), :428-461 + :789 (RoPE + cache update),
// :946-1007 (flash-attn varlen prefill), :505-520 (gate), :608-639 (moe_infer: argsort, per-expert loop, un-permute,
// weighted sum).  The grouped GEMMs themselves are mn_gemm_bf16_grouped (batch_ops.hip).
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------- RMSNorm -> bf16
__global__ __launch_bounds__(256) void rmsnorm_bf16_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const bf16_t* __restrict__ g, float eps,
                                                           bf16_t* __restrict__ y, int64_t ldy, int M, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  float ss = 0.f;
  for (int k = lane * 4; k < D; k += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  for (int k = lane * 4; k < D; k += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
    u32x2 pk = {pack_bf16x2(v.x * rstd * bf16_to_f32(g[k]), v.y * rstd * bf16_to_f32(g[k + 1])),
                pack_bf16x2(v.z * rstd * bf16_to_f32(g[k + 2]), v.w * rstd * bf16_to_f32(g[k + 3]))};
    *reinterpret_cast<u32x2*>(y + (int64_t)row * ldy + k) = pk;
  }
}

extern "C" int mn_rmsnorm_bf16(const float* x, int64_t ldx, const uint16_t* g, float eps, uint16_t* y, int64_t ldy,
                               int M, int D, void* stream) {
  MN_CHECK_ARG(x && g && y && M >= 1 && D >= 4 && (D % 4) == 0 && (ldx % 4) == 0 && (ldy % 4) == 0, "mn_rmsnorm_bf16: bad args");
  hipLaunchKernelGGL(rmsnorm_bf16_kernel, dim3(mn_cdiv(M, 4)), dim3(256), 0, mn_stream(stream), x, ldx, g, eps, y, ldy, M, D);
  MN_CHECK_LAUNCH("mn_rmsnorm_bf16");
  return MN_OK;
}

// ------------------------------------------------------------------------------------------- RoPE + KV append (T tokens)
// grid (T, n_q + 2 n_kv), block hd/2.  Token t of the new span goes to cache slot slot0 + t, rotary position pos[t].
__global__ void rope_kv_prefill_kernel(const float* __restrict__ qkv, int64_t ldqkv, int n_q, int n_kv, int hd,
                                       const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                       const int32_t* __restrict__ pos, int slot0, float q_scale,
                                       bf16_t* __restrict__ q_out, float* __restrict__ kv_seq, int64_t t_max) {
  const int t = blockIdx.x, h = blockIdx.y, i = threadIdx.x, half = hd >> 1;
  const float* src = qkv + (int64_t)t * ldqkv + (int64_t)h * hd;
  float x1 = src[i], x2 = src[i + half];
  const bool is_q = h < n_q, is_k = !is_q && h < n_q + n_kv;
  if (is_q || is_k) {
    const int p = pos[t];
    const float c = cos_tab[(int64_t)p * half + i], s = sin_tab[(int64_t)p * half + i];
    const float o1 = x1 * c - x2 * s, o2 = x2 * c + x1 * s;
    x1 = o1; x2 = o2;
  }
  if (is_q) {
    bf16_t* dst = q_out + ((int64_t)t * n_q + h) * hd;
    dst[i] = f32_to_bf16(x1 * q_scale);
    dst[i + half] = f32_to_bf16(x2 * q_scale);
  } else {
    const int kvh = is_k ? h - n_q : h - n_q - n_kv;
    float* dst = kv_seq + ((((int64_t)(is_k ? 0 : 1)) * n_kv + kvh) * t_max + slot0 + t) * hd;
    dst[i] = x1;
    dst[i + half] = x2;
  }
}

extern "C" int mn_rope_kv_prefill(const float* qkv, int64_t ldqkv, int T, int n_q, int n_kv, int hd, const float* cos_tab,
                                  const float* sin_tab, const int32_t* pos, int slot0, float q_scale, uint16_t* q_out,
                                  float* kv_seq, int64_t t_max, void* stream) {
  MN_CHECK_ARG(qkv && cos_tab && sin_tab && pos && q_out && kv_seq && T >= 1 && (hd == 64 || hd == 128) && slot0 >= 0 &&
                   slot0 + T <= t_max, "mn_rope_kv_prefill: bad args");
  hipLaunchKernelGGL(rope_kv_prefill_kernel, dim3(T, n_q + 2 * n_kv), dim3(hd / 2), 0, mn_stream(stream), qkv, ldqkv, n_q,
                     n_kv, hd, cos_tab, sin_tab, pos, slot0, q_scale, q_out, kv_seq, t_max);
  MN_CHECK_LAUNCH("mn_rope_kv_prefill");
  return MN_OK;
}

// The same for several prompt spans in one launch: span s = rows [r0, r0 + len) of qkv / q_out (positions pos[row]) appended to
// cache sequence seq at slots slot0 .. of kv_layer [n_seq_total, 2, n_kv, t_max, hd]; seq_tab [n_spans][3] = (seq, r0, len).
__global__ void rope_kv_prefill_spans_kernel(const float* __restrict__ qkv, int64_t ldqkv, int n_q, int n_kv, int hd,
                                             const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                             const int32_t* __restrict__ pos, int slot0, float q_scale, bf16_t* __restrict__ q_out,
                                             float* __restrict__ kv_layer, int64_t t_max, const int32_t* __restrict__ seq_tab) {
  const int s = blockIdx.z, seq = seq_tab[s * 3], r0 = seq_tab[s * 3 + 1], len = seq_tab[s * 3 + 2];
  const int t = blockIdx.x, h = blockIdx.y, i = threadIdx.x, half = hd >> 1;
  if (t >= len) return;
  const int64_t row = r0 + t;
  const float* src = qkv + row * ldqkv + (int64_t)h * hd;
  float x1 = src[i], x2 = src[i + half];
  const bool is_q = h < n_q, is_k = !is_q && h < n_q + n_kv;
  if (is_q || is_k) {
    const int p = pos[row];
    const float c = cos_tab[(int64_t)p * half + i], sn = sin_tab[(int64_t)p * half + i];
    const float o1 = x1 * c - x2 * sn, o2 = x2 * c + x1 * sn;
    x1 = o1; x2 = o2;
  }
  if (is_q) {
    bf16_t* dst = q_out + (row * n_q + h) * hd;
    dst[i] = f32_to_bf16(x1 * q_scale);
    dst[i + half] = f32_to_bf16(x2 * q_scale);
  } else {
    const int kvh = is_k ? h - n_q : h - n_q - n_kv;
    float* dst = kv_layer + (int64_t)seq * 2 * n_kv * t_max * hd + ((((int64_t)(is_k ? 0 : 1)) * n_kv + kvh) * t_max + slot0 + t) * hd;
    dst[i] = x1;
    dst[i + half] = x2;
  }
}

extern "C" int mn_rope_kv_prefill_spans(const float* qkv, int64_t ldqkv, int n_q, int n_kv, int hd, const float* cos_tab,
                                        const float* sin_tab, const int32_t* pos, int slot0, float q_scale, uint16_t* q_out,
                                        float* kv_layer, int64_t t_max, const int32_t* seq_tab, int n_spans, int max_len, void* stream) {
  MN_CHECK_ARG(qkv && cos_tab && sin_tab && pos && q_out && kv_layer && seq_tab && n_spans >= 1 && max_len >= 1 &&
                   (hd == 64 || hd == 128) && slot0 >= 0 && slot0 + max_len <= t_max, "mn_rope_kv_prefill_spans: bad args");
  hipLaunchKernelGGL(rope_kv_prefill_spans_kernel, dim3(max_len, n_q + 2 * n_kv, n_spans), dim3(hd / 2), 0, mn_stream(stream), qkv,
                     ldqkv, n_q, n_kv, hd, cos_tab, sin_tab, pos, slot0, q_scale, q_out, kv_layer, t_max, seq_tab);
  MN_CHECK_LAUNCH("mn_rope_kv_prefill_spans");
  return MN_OK;
}

// ------------------------------------------------------------------------------------------- GQA flash attention hd=128
// Block = (64-query tile, q head); wave = 16 queries (a query COLUMN per lane & 15, as in attn_prefill_hd64_kernel).
// Per 32-key tile the block converts K [32][128] and V^T [128][32] from the fp32 arena to bf16 in LDS once;
//   S^T[key, q] = K Q^T  (8 MFMAs: 2 key sub-tiles x 4 d-steps),  online softmax per query column,
//   O^T[d, q] += V^T P^T (8 MFMAs: 8 d-tiles, k = the 32 keys in the slot order the S^T accumulators already have).
namespace {
constexpr int AKT = 32;
constexpr int KROW = 128 + 8;   // bf16 elements per K row in LDS (+16 B pad)
constexpr int VROW = AKT + 8;

__global__ __launch_bounds__(256) void attn_prefill_gqa_hd128_kernel(
    const bf16_t* __restrict__ q, const float* __restrict__ kv_seq, int64_t t_max, int n_q, int n_kv, int past, int T,
    const uint8_t* __restrict__ key_mask, bf16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) bf16_t ks[AKT][KROW];
  __shared__ __attribute__((aligned(16))) bf16_t vt[128][VROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y;
  const int kvh = h / (n_q / n_kv);
  const float* Kb = kv_seq + ((int64_t)0 * n_kv + kvh) * t_max * 128;
  const float* Vb = kv_seq + ((int64_t)1 * n_kv + kvh) * t_max * 128;
  const int ql = lane & 15, g = lane >> 4;
  const int q_idx = qt * 64 + wave * 16 + ql;            // index within the new span
  const int q_ld = min(q_idx, T - 1);
  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    qf[kk] = *reinterpret_cast<const bf16x8*>(q + ((int64_t)q_ld * n_q + h) * 128 + kk * 32 + g * 8);
  f32x4 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int q_hi = min(T, qt * 64 + 64) - 1;
  const int k_end = past + q_hi + 1;                      // bottom-right causal: query i sees keys <= past + i
  const int ntile = (k_end + AKT - 1) / AKT;
  for (int kt = 0; kt < ntile; ++kt) {
    const int k0 = kt * AKT;
    __syncthreads();
    {  // stage K (row-major) and V^T: thread -> key = tid / 8, 16 d per thread
      const int key = tid >> 3, dc = (tid & 7) * 16;
      const int kr = min(k0 + key, past + T - 1);
      const float* kp = Kb + (int64_t)kr * 128 + dc;
      const float* vp = Vb + (int64_t)kr * 128 + dc;
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        const f32x4 kv4 = *reinterpret_cast<const f32x4*>(kp + j);
        const f32x4 vv4 = *reinterpret_cast<const f32x4*>(vp + j);
        u32x2 pk = {pack_bf16x2(kv4.x, kv4.y), pack_bf16x2(kv4.z, kv4.w)};
        *reinterpret_cast<u32x2*>(&ks[key][dc + j]) = pk;
        vt[dc + j + 0][key] = f32_to_bf16(vv4.x);
        vt[dc + j + 1][key] = f32_to_bf16(vv4.y);
        vt[dc + j + 2][key] = f32_to_bf16(vv4.z);
        vt[dc + j + 3][key] = f32_to_bf16(vv4.w);
      }
    }
    __syncthreads();
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&ks[t * 16 + ql][kk * 32 + g * 8]);
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + t * 16 + g * 4 + r;
        const bool ok = key <= past + q_idx && key < past + T && (!key_mask || key_mask[key] != 0);
        s[t][r] = ok ? s[t][r] : -INFINITY;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
    float psum = 0.f, p[8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (m_new == -INFINITY) ? 0.f : __expf(s[t][r] - m_new);
        p[t * 4 + r] = e;
        psum += e;
      }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] *= alpha;
    u32x4 pk = {pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const bf16_t* vr = &vt[dt * 16 + ql][0];
      const u32x2 lo = *reinterpret_cast<const u32x2*>(vr + g * 4);
      const u32x2 hi = *reinterpret_cast<const u32x2*>(vr + 16 + g * 4);
      u32x4 vv = {lo.x, lo.y, hi.x, hi.y};
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[dt], 0, 0, 0);
    }
  }
  if (q_idx < T) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;   // no attended key: 0, not NaN
    bf16_t* op = out + ((int64_t)q_idx * n_q + h) * 128;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      u32x2 pk = {pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
      *reinterpret_cast<u32x2*>(op + dt * 16 + g * 4) = pk;
    }
  }
}
}  // namespace

extern "C" int mn_flash_prefill_gqa_hd128_one(const uint16_t* q, const float* kv_seq, int64_t t_max, int n_q, int n_kv, int past, int T,
                                              const uint8_t* key_mask, uint16_t* out, void* stream);

extern "C" int mn_attn_prefill_gqa_hd128(const uint16_t* q, const float* kv_seq, int64_t t_max, int n_q, int n_kv, int past,
                                         int T, const uint8_t* key_mask, uint16_t* out, void* stream) {
  MN_CHECK_ARG(q && kv_seq && out && T >= 1 && past >= 0 && past + T <= t_max && n_q >= 1 && n_kv >= 1 && n_q % n_kv == 0,
               "mn_attn_prefill_gqa_hd128: bad args");
  if (n_q == 4 * n_kv)     // flash_prefill.hip: 64-key tiles, the 4 query heads of a KV head share its tiles (the 16B-A3B ratio)
    return mn_flash_prefill_gqa_hd128_one(q, kv_seq, t_max, n_q, n_kv, past, T, key_mask, out, stream);
  // any other GQA ratio (the tiny test configurations): one query head per workgroup, 32-key tiles
  hipLaunchKernelGGL(attn_prefill_gqa_hd128_kernel, dim3(mn_cdiv(T, 64), n_q), dim3(256), 0, mn_stream(stream), q, kv_seq,
                     t_max, n_q, n_kv, past, T, key_mask, out);
  MN_CHECK_LAUNCH("mn_attn_prefill_gqa_hd128");
  return MN_OK;
}

// ------------------------------------------------------------------------------------------- router top-k from logits
// one wave per token
__global__ __launch_bounds__(256) void moe_topk_logits_kernel(const float* __restrict__ logits_text,
                                                              const float* __restrict__ logits_image,
                                                              const uint8_t* __restrict__ image_mask, int T, int E, int top_k,
                                                              int norm_topk_prob, int n_shared, int32_t* __restrict__ topk_idx,
                                                              float* __restrict__ topk_w) {
  const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float* lg = (image_mask && logits_image && image_mask[t]) ? logits_image : logits_text;
  float s = lane < E ? lg[(int64_t)t * E + lane] : -INFINITY;
  const float mx = wave_max(s);
  float p = lane < E ? __expf(s - mx) : 0.f;
  p = p / wave_sum(p);
  float cur = lane < E ? p : -1.f, wsum = 0.f, myw = 0.f;
  int myidx = 0;
  const int n_slot = top_k + n_shared;
  for (int k = 0; k < top_k; ++k) {
    const float best = wave_max(cur);
    const int sel = __ffsll((long long)__ballot(cur == best)) - 1;
    if (lane == k) { myw = best; myidx = sel; }
    if (lane == sel) cur = -1.f;
    wsum += best;
  }
  if (lane < top_k) {
    topk_idx[(int64_t)t * n_slot + lane] = myidx;
    topk_w[(int64_t)t * n_slot + lane] = (norm_topk_prob && top_k > 1) ? myw / wsum : myw;
  } else if (lane < n_slot) {
    topk_idx[(int64_t)t * n_slot + lane] = E + (lane - top_k);
    topk_w[(int64_t)t * n_slot + lane] = 1.0f;
  }
}

extern "C" int mn_moe_topk_logits(const float* logits_text, const float* logits_image, const uint8_t* image_mask, int T, int E,
                                  int top_k, int norm_topk_prob, int n_shared_slots, int32_t* topk_idx, float* topk_w,
                                  void* stream) {
  MN_CHECK_ARG(logits_text && topk_idx && topk_w && T >= 1 && E >= 1 && E <= 64 && top_k >= 1 && top_k <= E &&
                   top_k + n_shared_slots <= 64, "mn_moe_topk_logits: bad args");
  hipLaunchKernelGGL(moe_topk_logits_kernel, dim3(mn_cdiv(T, 4)), dim3(256), 0, mn_stream(stream), logits_text, logits_image,
                     image_mask, T, E, top_k, norm_topk_prob, n_shared_slots, topk_idx, topk_w);
  MN_CHECK_LAUNCH("mn_moe_topk_logits");
  return MN_OK;
}

// ------------------------------------------------------------------------------------------- token <-> expert permutation
// Single workgroup (T * n_slot <= 65536): counts per expert, exclusive offsets, then a scatter that gives every
// (token, pick) its position in the expert-sorted order.  Order inside an expert's segment is arbitrary (LDS
// atomics) but every consumer goes through slot_of / perm, so results do not depend on it.
__global__ __launch_bounds__(1024) void moe_sort_kernel(const int32_t* __restrict__ topk_idx, int n_pairs, int n_groups,
                                                        int32_t* __restrict__ counts, int32_t* __restrict__ offsets,
                                                        int32_t* __restrict__ perm, int32_t* __restrict__ slot_of,
                                                        int n_slot, int tile_rows, int32_t* __restrict__ tile_g,
                                                        int32_t* __restrict__ tile_m0, int32_t* __restrict__ n_tiles,
                                                        int g_lo, int g_hi) {
  __shared__ int cnt[128], off[129], cur[128], toff[129], wtot[2];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 128) { cnt[tid] = 0; cur[tid] = 0; }
  __syncthreads();
  for (int i = tid; i < n_pairs; i += 1024) atomicAdd(&cnt[topk_idx[i]], 1);
  __syncthreads();
  // exclusive scans over the <= 128 groups — rows per group, and row tiles of the groups in [g_lo, g_hi) — by the first two waves:
  // inclusive scan inside a wave by shuffles, the second wave adds the first one's totals
  int c = 0, t = 0, sc = 0, stl = 0;
  if (tid < 128) {
    c = tid < n_groups ? cnt[tid] : 0;
    t = (tile_g && tid >= g_lo && tid < g_hi) ? (c + tile_rows - 1) / tile_rows : 0;
    sc = c; stl = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int a = __shfl_up(sc, d, 64), b = __shfl_up(stl, d, 64);
      if (lane >= d) { sc += a; stl += b; }
    }
    if (tid == 63) { wtot[0] = sc; wtot[1] = stl; }
  }
  __syncthreads();
  if (tid < 128) {
    if (tid >= 64) { sc += wtot[0]; stl += wtot[1]; }
    off[tid] = sc - c;
    toff[tid] = stl - t;
    if (tid == 127) { off[128] = sc; toff[128] = stl; }      // totals (groups past n_groups count 0, so off[n_groups] is the total too)
  }
  __syncthreads();
  if (tid < n_groups) { counts[tid] = c; offsets[tid] = off[tid]; }
  if (tid == 0) offsets[n_groups] = off[n_groups];
  if (tile_g) {        // live row tiles of the grouped GEMMs, in group order; expert parallelism: only the groups this rank holds get tiles
    if (tid >= g_lo && tid < g_hi) {
      int k = toff[tid];
      for (int m0 = 0; m0 < c; m0 += tile_rows) { tile_g[k] = tid; tile_m0[k] = m0; ++k; }
    }
    if (tid == 0) *n_tiles = toff[128];
  }
  for (int i = tid; i < n_pairs; i += 1024) {
    const int e = topk_idx[i];
    const int pos = off[e] + atomicAdd(&cur[e], 1);
    perm[pos] = i / n_slot;       // sorted position -> token
    slot_of[i] = pos;             // (token, pick) -> sorted position
  }
}

extern "C" int mn_moe_sort(const int32_t* topk_idx, int T, int n_slot, int n_groups, int32_t* counts, int32_t* offsets,
                           int32_t* perm, int32_t* slot_of, void* stream) {
  MN_CHECK_ARG(topk_idx && counts && offsets && perm && slot_of && T >= 1 && n_slot >= 1 && n_groups >= 1 && n_groups <= 128 &&
                   (int64_t)T * n_slot <= 65536, "mn_moe_sort: bad args");
  hipLaunchKernelGGL(moe_sort_kernel, dim3(1), dim3(1024), 0, mn_stream(stream), topk_idx, T * n_slot, n_groups, counts,
                     offsets, perm, slot_of, n_slot, 0, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, 0, 0);
  MN_CHECK_LAUNCH("mn_moe_sort");
  return MN_OK;
}

// Internal (engine.hip): mn_moe_sort that also lists the live row tiles (tile_rows rows each) of the grouped GEMMs:
// tile_g / tile_m0 [<= T * n_slot / tile_rows + n_groups], *n_tiles.
extern "C" int mn_moe_sort_tiles(const int32_t* topk_idx, int T, int n_slot, int n_groups, int32_t* counts, int32_t* offsets,
                                 int32_t* perm, int32_t* slot_of, int tile_rows, int32_t* tile_g, int32_t* tile_m0,
                                 int32_t* n_tiles, void* stream) {
  MN_CHECK_ARG(topk_idx && counts && offsets && perm && slot_of && tile_g && tile_m0 && n_tiles && tile_rows >= 1 && T >= 1 &&
                   n_slot >= 1 && n_groups >= 1 && n_groups <= 128 && (int64_t)T * n_slot <= 65536, "mn_moe_sort_tiles: bad args");
  hipLaunchKernelGGL(moe_sort_kernel, dim3(1), dim3(1024), 0, mn_stream(stream), topk_idx, T * n_slot, n_groups, counts,
                     offsets, perm, slot_of, n_slot, tile_rows, tile_g, tile_m0, n_tiles, 0, n_groups);
  MN_CHECK_LAUNCH("mn_moe_sort_tiles");
  return MN_OK;
}

// Expert-parallel dispatch (replicate-and-reduce EP): every rank holds all rows and the global routing; it sorts the
// (row, pick) pairs by GLOBAL expert id like mn_moe_sort_tiles but lists row tiles only for the experts [expert0, expert0 + n_local)
// it owns — the grouped GEMMs then touch only local weights (W biased by -expert0 groups) and local pairs.
extern "C" int mn_ep_dispatch(const int32_t* topk_idx, int T, int n_slot, int n_experts, int expert0, int n_local, int32_t* counts,
                              int32_t* offsets, int32_t* perm, int32_t* slot_of, int tile_rows, int32_t* tile_g, int32_t* tile_m0,
                              int32_t* n_tiles, void* stream) {
  MN_CHECK_ARG(topk_idx && counts && offsets && perm && slot_of && tile_g && tile_m0 && n_tiles && tile_rows >= 1 && T >= 1 &&
                   n_slot >= 1 && n_experts >= 1 && n_experts <= 128 && (int64_t)T * n_slot <= 65536 && expert0 >= 0 && n_local >= 1 &&
                   expert0 + n_local <= n_experts, "mn_ep_dispatch: bad args");
  hipLaunchKernelGGL(moe_sort_kernel, dim3(1), dim3(1024), 0, mn_stream(stream), topk_idx, T * n_slot, n_experts, counts,
                     offsets, perm, slot_of, n_slot, tile_rows, tile_g, tile_m0, n_tiles, expert0, expert0 + n_local);
  MN_CHECK_LAUNCH("mn_ep_dispatch");
  return MN_OK;
}

__global__ void gather_rows_bf16_kernel(const bf16_t* __restrict__ x, int64_t ldx, const int32_t* __restrict__ perm,
                                        bf16_t* __restrict__ y, int64_t ldy, int n_rows, int D) {
  const int per = D / 8;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * per) return;
  const int r = (int)(i / per), c = (int)(i % per) * 8;
  *reinterpret_cast<u32x4*>(y + (int64_t)r * ldy + c) = *reinterpret_cast<const u32x4*>(x + (int64_t)perm[r] * ldx + c);
}

extern "C" int mn_gather_rows_bf16(const uint16_t* x, int64_t ldx, const int32_t* perm, uint16_t* y, int64_t ldy, int n_rows,
                                   int D, void* stream) {
  MN_CHECK_ARG(x && perm && y && n_rows >= 1 && D >= 8 && (D % 8) == 0 && (ldx % 8) == 0 && (ldy % 8) == 0, "mn_gather_rows_bf16: bad args");
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(mn_cdiv((int64_t)n_rows * (D / 8), 256)), dim3(256), 0, mn_stream(stream),
                     x, ldx, perm, y, ldy, n_rows, D);
  MN_CHECK_LAUNCH("mn_gather_rows_bf16");
  return MN_OK;
}

// h[t] += sum_j w[t, j] * y[slot_of[t, j]]   (fp32; the weighted un-permute of moe_infer, :629-638, + residual)
__global__ void moe_combine_kernel(const float* __restrict__ y, int64_t ldy, const int32_t* __restrict__ slot_of,
                                   const float* __restrict__ w, int n_slot, float* __restrict__ h, int64_t ldh, int T, int D) {
  const int per = D / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * per) return;
  const int t = (int)(i / per), c = (int)(i % per) * 4;
  f32x4 acc = *reinterpret_cast<const f32x4*>(h + (int64_t)t * ldh + c);
  for (int j = 0; j < n_slot; ++j) {
    const float wj = w[(int64_t)t * n_slot + j];
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + (int64_t)slot_of[(int64_t)t * n_slot + j] * ldy + c);
    acc.x = fmaf(wj, v.x, acc.x); acc.y = fmaf(wj, v.y, acc.y); acc.z = fmaf(wj, v.z, acc.z); acc.w = fmaf(wj, v.w, acc.w);
  }
  *reinterpret_cast<f32x4*>(h + (int64_t)t * ldh + c) = acc;
}

extern "C" int mn_moe_combine(const float* y, int64_t ldy, const int32_t* slot_of, const float* w, int n_slot, float* h,
                              int64_t ldh, int T, int D, void* stream) {
  MN_CHECK_ARG(y && slot_of && w && h && T >= 1 && D >= 4 && (D % 4) == 0 && (ldy % 4) == 0 && (ldh % 4) == 0, "mn_moe_combine: bad args");
  hipLaunchKernelGGL(moe_combine_kernel, dim3(mn_cdiv((int64_t)T * (D / 4), 256)), dim3(256), 0, mn_stream(stream), y, ldy,
                     slot_of, w, n_slot, h, ldh, T, D);
  MN_CHECK_LAUNCH("mn_moe_combine");
  return MN_OK;
}

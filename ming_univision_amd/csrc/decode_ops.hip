// decode_ops.hip — small-row (decode) operators around the skinny GEMM (gfx950):
//   mn_moe_router       RMSNorm + gate GEMV + softmax + top-k (+ image-gate override)
//   mn_rope_kv_append   neox-style rotary on q/k, append k/v to the fp32 KV cache
//   mn_attn_decode      masked GQA/MHA attention of <= 8 query rows against the cache
//                       (split over the key range = flash-decoding, then a combine pass)
// All of these are HBM/latency-bound index-and-reduce work: coalesced loads, wave shuffles,
// no MFMA.
#include <string.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------
// MoE router = gate logits (skinny GEMV with a fused RMSNorm prologue, all CUs) + this kernel:
// one 256-thread block per row writes the normalised row (input of the experts) and wave 0 does
// the fp32 softmax + iterative arg-max top-k + renormalisation.
//
// TIE RULE (every router of this library: this kernel, moe_router_row_kernel below, moe_route_group_kernel in engine.hip,
// moe_topk_logits_kernel in prefill_ops.hip and the wide route's top-k in wide_llm.inl): expert e lives in lane e; a round takes the
// wave maximum and `__ffsll(__ballot(cur == best))`, i.e. among EQUAL scores the LOWEST expert id wins, and the picked slots are in
// descending score / ascending id order.  The reference calls `torch.topk` (modeling_bailing_moe.py:512), whose order among equal
// values is unspecified; exact ties need two identical gate rows (fp32 scores of real weights tie with probability zero).
// Pinned by tests/test_gpu_router_ties.py.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moe_topk_kernel(
    const float* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ norm_w, float eps,
    const float* __restrict__ logits_text, const float* __restrict__ logits_image,
    const uint8_t* __restrict__ image_mask, int H, int E, int top_k, int norm_topk_prob, int n_shared,
    float* __restrict__ x_norm, int32_t* __restrict__ topk_idx, float* __restrict__ topk_w) {
  __shared__ float red[8];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (int64_t)m * ldx;
  float ss = 0.f;
  for (int k = tid; k < H; k += 256) { float v = xr[k]; ss += v * v; }
  ss = block_sum(ss, red);
  const float rstd = rsqrtf(ss / (float)H + eps);
  for (int k = tid; k < H; k += 256) x_norm[(int64_t)m * H + k] = xr[k] * rstd * bf16_to_f32(norm_w[k]);
  if (wave == 0) {
    const float* lg = (image_mask && logits_image && image_mask[m]) ? logits_image : logits_text;
    // softmax over E (fp32) then iterative arg-max; ties -> lowest expert index
    float s = lane < E ? lg[(int64_t)m * E + lane] : -INFINITY;
    const float mx = wave_max(s);
    float p = lane < E ? __expf(s - mx) : 0.f;
    const float denom = wave_sum(p);
    p = p / denom;
    float cur = lane < E ? p : -1.f;
    float wsum = 0.f;
    float myw = 0.f;
    int myidx = 0;
    const int n_slot = top_k + n_shared;
    for (int k = 0; k < top_k; ++k) {
      const float best = wave_max(cur);
      const unsigned long long ball = __ballot(cur == best);
      const int sel = __ffsll((long long)ball) - 1;
      if (lane == k) { myw = best; myidx = sel; }
      if (lane == sel) cur = -1.f;
      wsum += best;
    }
    if (lane < top_k) {
      topk_idx[(int64_t)m * n_slot + lane] = myidx;
      topk_w[(int64_t)m * n_slot + lane] = (norm_topk_prob && top_k > 1) ? myw / wsum : myw;
    } else if (lane < n_slot) {
      topk_idx[(int64_t)m * n_slot + lane] = E + (lane - top_k);
      topk_w[(int64_t)m * n_slot + lane] = 1.0f;
    }
  }
}

// Few rows (text decode, one image): the whole router in ONE launch — a 1024-thread workgroup per row does the RMSNorm, the gate
// GEMV of that row's own gate (image rows read image_gate only; 64 x H bf16 = 256 KB from L2, four experts per wave), and wave 0
// the softmax / top-k.  Replaces a 16-workgroup GEMV launch (9 us) + moe_topk_kernel (5 us) per layer.
__global__ __launch_bounds__(1024) void moe_router_row_kernel(
    const float* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ norm_w, float eps, const bf16_t* __restrict__ gate_w,
    const bf16_t* __restrict__ image_gate_w, const uint8_t* __restrict__ image_mask, int M, int H, int E, int top_k,
    int norm_topk_prob, int n_shared, float* __restrict__ x_norm, int32_t* __restrict__ topk_idx, float* __restrict__ topk_w,
    float* __restrict__ logits_out, const float* __restrict__ P, int nz, int64_t slab, float* __restrict__ h_out) {
  __shared__ __attribute__((aligned(16))) float xs[4096];
  __shared__ float red[16];
  __shared__ float lg[64];
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (int64_t)m * ldx;
  const bool img = image_mask && image_gate_w && image_mask[m];
  const bf16_t* G = img ? image_gate_w : gate_w;
  typedef float f4 __attribute__((ext_vector_type(4)));
  float acc[4] = {0.f, 0.f, 0.f, 0.f};          // experts wave, wave + 16, wave + 32, wave + 48: four weight rows in flight per lane
  // H <= 2048 (the 16B-A3B width): this lane's 16 weight slots do not depend on x — requested FIRST, so the launch is one memory round
  // trip (x and the gate rows together) + the reductions, not two dependent ones
  const bool pre = H <= 2048 && (H % 512) == 0;
  u4 gw[4][4];
  if (pre) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = wave + 16 * j, k = lane * 8 + 512 * i;
        gw[i][j] = u4{0u, 0u, 0u, 0u};
        if (e < E && k < H) gw[i][j] = *reinterpret_cast<const u4*>(G + (int64_t)e * H + k);
      }
  }
  // P != NULL (the decoder chain at <= 4 rows): the row is h + the nz K-slice partial slabs of the attention output projection, summed
  // here (in slab order, like llm_glue_kernel) and written back to h_out — the residual glue launch and the gate launch fold into this one
  float xv[4];                                    // H <= 4096: columns tid, tid + 1024, ...
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = tid + 1024 * i;
    xv[i] = 0.f;
    if (k < H) {
      float v = xr[k];
      if (P) {
        const float* pp = P + (int64_t)m * H + k;
        for (int z0 = 0; z0 < nz; z0 += 8) {
          float t[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = z0 + j < nz ? pp[(int64_t)(z0 + j) * slab] : 0.f;
          for (int j = 0; j < 8; ++j) v += t[j];
        }
        h_out[(int64_t)m * H + k] = v;
      }
      xv[i] = v;
      ss += v * v;
    }
  }
  ss = block_sum(ss, red);
  const float rstd = rsqrtf(ss / (float)H + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = tid + 1024 * i;
    if (k < H) {
      const float v = xv[i] * rstd * bf16_to_f32(norm_w[k]);
      xs[k] = v;
      x_norm[(int64_t)m * H + k] = v;
    }
  }
  __syncthreads();
  if (pre) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = lane * 8 + 512 * i;
      if (k < H) {
        const f4 xa = *reinterpret_cast<const f4*>(&xs[k]), xb = *reinterpret_cast<const f4*>(&xs[k + 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u4 wv = gw[i][j];
          float a = acc[j];
          a = fmaf(bf16lo_to_f32(wv.x), xa.x, a); a = fmaf(bf16hi_to_f32(wv.x), xa.y, a);
          a = fmaf(bf16lo_to_f32(wv.y), xa.z, a); a = fmaf(bf16hi_to_f32(wv.y), xa.w, a);
          a = fmaf(bf16lo_to_f32(wv.z), xb.x, a); a = fmaf(bf16hi_to_f32(wv.z), xb.y, a);
          a = fmaf(bf16lo_to_f32(wv.w), xb.z, a); a = fmaf(bf16hi_to_f32(wv.w), xb.w, a);
          acc[j] = a;
        }
      }
    }
  } else
  for (int k = lane * 8; k < H; k += 512) {
    const f4 xa = *reinterpret_cast<const f4*>(&xs[k]), xb = *reinterpret_cast<const f4*>(&xs[k + 4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = wave + 16 * j;
      if (e < E) {
        const u4 wv = *reinterpret_cast<const u4*>(G + (int64_t)e * H + k);
        float a = acc[j];
        a = fmaf(bf16lo_to_f32(wv.x), xa.x, a); a = fmaf(bf16hi_to_f32(wv.x), xa.y, a);
        a = fmaf(bf16lo_to_f32(wv.y), xa.z, a); a = fmaf(bf16hi_to_f32(wv.y), xa.w, a);
        a = fmaf(bf16lo_to_f32(wv.z), xb.x, a); a = fmaf(bf16hi_to_f32(wv.z), xb.y, a);
        a = fmaf(bf16lo_to_f32(wv.w), xb.z, a); a = fmaf(bf16hi_to_f32(wv.w), xb.w, a);
        acc[j] = a;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = wave_sum(acc[j]);
    if (lane == 0 && wave + 16 * j < E) lg[wave + 16 * j] = t;
  }
  __syncthreads();
  if (wave == 0) {
    // softmax over E (fp32) then iterative arg-max; ties -> lowest expert index  (BailingMoeGate.forward :505-520)
    float s = lane < E ? lg[lane] : -INFINITY;
    if (lane < E) logits_out[((int64_t)(img ? 1 : 0) * M + m) * E + lane] = s;
    const float mx = wave_max(s);
    float p = lane < E ? __expf(s - mx) : 0.f;
    const float denom = wave_sum(p);
    p = p / denom;
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
      topk_idx[(int64_t)m * n_slot + lane] = myidx;
      topk_w[(int64_t)m * n_slot + lane] = (norm_topk_prob && top_k > 1) ? myw / wsum : myw;
    } else if (lane < n_slot) {
      topk_idx[(int64_t)m * n_slot + lane] = E + (lane - top_k);
      topk_w[(int64_t)m * n_slot + lane] = 1.0f;
    }
  }
}

static int g_router_rows = 4;      // rows up to which mn_moe_router runs as one launch (A/B hook: mn_moe_router_tune)
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_moe_router_tune(int max_rows) { g_router_rows = max_rows; }
#endif

// The decoder chain's form at <= 4 rows (engine.hip): h[m] += sum of the nz partial slabs P [nz][M][H]; RMSNorm -> x_norm; gate; top-k — one launch
bool moe_router_rows_ok(int M, int H, int E) { return M >= 1 && M <= g_router_rows && H <= 4096 && (H % 8) == 0 && E <= 64; }
int moe_router_rows(float* h, const float* P, int nz, const uint16_t* norm_w, float eps, const uint16_t* gate_w, int M, int H, int E, int top_k,
                    int norm_topk_prob, int n_shared_slots, float* x_norm, int32_t* topk_idx, float* topk_w, float* logits_ws, void* stream) {
  MN_CHECK_ARG(h && P && norm_w && gate_w && x_norm && topk_idx && topk_w && logits_ws && moe_router_rows_ok(M, H, E) && top_k >= 1 && top_k <= E &&
               top_k + n_shared_slots <= 64, "moe_router_rows: bad args");
  hipLaunchKernelGGL(moe_router_row_kernel, dim3(M), dim3(1024), 0, mn_stream(stream), (const float*)h, (int64_t)H, norm_w, eps, gate_w,
                     (const bf16_t*)nullptr, (const uint8_t*)nullptr, M, H, E, top_k, norm_topk_prob, n_shared_slots, x_norm, topk_idx, topk_w,
                     logits_ws, P, nz, (int64_t)M * H, h);
  MN_CHECK_LAUNCH("moe_router_rows");
  return MN_OK;
}

extern "C" int mn_moe_router(const float* x, int64_t ldx, const uint16_t* norm_w, float eps,
                             const uint16_t* gate_w, const uint16_t* image_gate_w, const uint8_t* image_mask,
                             int M, int H, int E, int top_k, int norm_topk_prob, int n_shared_slots,
                             float* x_norm, int32_t* topk_idx, float* topk_w, float* logits_ws, void* ws, size_t ws_bytes,
                             void* stream) {
  MN_CHECK_ARG(M >= 1 && M <= 64 && H >= 8 && (H % 8) == 0, "mn_moe_router: bad M=%d H=%d", M, H);
  MN_CHECK_ARG(E >= 1 && E <= 64 && top_k >= 1 && top_k <= E && top_k + n_shared_slots <= 64,
               "mn_moe_router: bad E=%d top_k=%d", E, top_k);
  MN_CHECK_ARG(x && norm_w && gate_w && x_norm && topk_idx && topk_w && logits_ws, "mn_moe_router: null pointer");
  const bool both = image_mask && image_gate_w;
  if (M <= g_router_rows && H <= 4096) {
    hipLaunchKernelGGL(moe_router_row_kernel, dim3(M), dim3(1024), 0, mn_stream(stream), x, ldx, norm_w, eps, gate_w, image_gate_w,
                       image_mask, M, H, E, top_k, norm_topk_prob, n_shared_slots, x_norm, topk_idx, topk_w, logits_ws,
                       (const float*)nullptr, 0, (int64_t)0, (float*)nullptr);
    MN_CHECK_LAUNCH("mn_moe_router");
    return MN_OK;
  }
  // >= 5 rows with scratch: one launch on the matrix-core route; otherwise the fp32 path, 8 rows at a time
  const bool mfma = M <= 64 && ws != nullptr && ws_bytes >= mn_skinny_workspace_bytes(M, E, H, 0) && mn_skinny_workspace_bytes(M, E, H, 0) > 0;
  const int mstep = mfma ? M : 8;
  for (int gsel = 0; gsel < (both ? 2 : 1); ++gsel)
    for (int m0 = 0; m0 < M; m0 += mstep) {
      mn_skinny_args a;
      memset(&a, 0, sizeof(a));
      a.x = x + (int64_t)m0 * ldx; a.ldx = ldx; a.w = gsel ? image_gate_w : gate_w; a.ldw = H;
      a.out = logits_ws + (int64_t)gsel * M * E + (int64_t)m0 * E; a.ldo = E;
      a.M = (M - m0) < mstep ? (M - m0) : mstep; a.N = E; a.K = H;
      a.prologue = MN_PRO_RMSNORM; a.ln_g = norm_w; a.eps = eps;
      if (mfma) { a.ws = ws; a.ws_bytes = ws_bytes; }
      const int rc = mn_skinny_gemm(&a, stream);
      if (rc != MN_OK) return rc;
    }
  hipLaunchKernelGGL(moe_topk_kernel, dim3(M), dim3(256), 0, mn_stream(stream), x, ldx, norm_w, eps, logits_ws,
                     both ? logits_ws + (int64_t)M * E : nullptr, image_mask, H, E, top_k, norm_topk_prob,
                     n_shared_slots, x_norm, topk_idx, topk_w);
  MN_CHECK_LAUNCH("mn_moe_router");
  return MN_OK;
}

// -------------------------------------------------------------------------------------------
// RoPE + KV append. grid (M, n_q + 2*n_kv), block hd/2 threads.
// -------------------------------------------------------------------------------------------
// nz > 1: qkv is a stack of nz K-slice partial slabs (slab elements apart) of the QKV projection, summed here — the
// reduction of the streaming GEMM folded into the consumer (no bias: use_qkv_bias is false on this path).
__global__ void rope_kv_append_kernel(const float* __restrict__ qkv, int64_t ldqkv, int nz, int64_t slab, int n_q, int n_kv, int hd,
                                      int rope, const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                      const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_slot,
                                      const int32_t* __restrict__ row_pos, int M, int sec_t, int sec_h, float q_scale,
                                      float* __restrict__ q_out, float* __restrict__ kv_cache, int64_t t_max, int round_kv) {
  const int m = blockIdx.x, h = blockIdx.y, i = threadIdx.x, half = hd >> 1;
  const float* src = qkv + (int64_t)m * ldqkv + (int64_t)h * hd;
  float x1 = 0.f, x2 = 0.f;
  for (int z0 = 0; z0 < nz; z0 += 8) {              // eight slabs' loads in flight (one by one: nz dependent round trips)
    float a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a1[j] = z0 + j < nz ? src[(int64_t)(z0 + j) * slab + i] : 0.f;
      a2[j] = z0 + j < nz ? src[(int64_t)(z0 + j) * slab + i + half] : 0.f;
    }
    for (int j = 0; j < 8; ++j) x1 += a1[j];                    // (the one-by-one loop's order: same bits)
    for (int j = 0; j < 8; ++j) x2 += a2[j];
  }
  const bool is_q = h < n_q, is_k = !is_q && h < n_q + n_kv;
  if (rope && (is_q || is_k)) {
    // 3D rotary (sec_t > 0): row_pos is [3][M] = t, h, w positions; frequency i of each half follows the t stream for
    // i < sec_t, the h stream for the next sec_h, the w stream for the rest (apply_multimodal_rotary_pos_emb, :463-469)
    const int stream = sec_t <= 0 ? 0 : (i < sec_t ? 0 : (i < sec_t + sec_h ? 1 : 2));
    const int pos = row_pos[stream * M + m];
    const float c = cos_tab[(int64_t)pos * half + i], s = sin_tab[(int64_t)pos * half + i];
    const float o1 = x1 * c - x2 * s, o2 = x2 * c + x1 * s;
    x1 = o1; x2 = o2;
  }
  if (is_q) {
    float* dst = q_out + (int64_t)m * n_q * hd + (int64_t)h * hd;
    dst[i] = x1 * q_scale;
    dst[i + half] = x2 * q_scale;
  } else {
    const int kvh = is_k ? h - n_q : h - n_q - n_kv;
    const int64_t seq = row_seq[m], slot = row_slot[m];
    if ((uint64_t)slot >= (uint64_t)t_max) return;      // never write outside the sequence's arena rows (the host raises before a cache fills up)
    float* dst = kv_cache + (((seq * 2 + (is_k ? 0 : 1)) * n_kv + kvh) * t_max + slot) * hd;
    if (round_kv) {     // measurement only (dev library, tests/measure/kv_bf16_error.py): what a bf16 KV cache would hold
      x1 = bf16_to_f32(f32_to_bf16(x1));
      x2 = bf16_to_f32(f32_to_bf16(x2));
    }
    dst[i] = x1;
    dst[i + half] = x2;
  }
}

static int g_kv_round_bf16 = 0;
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_kv_round_bf16(int on) { g_kv_round_bf16 = on; }
#endif

// Internal (engine.hip): as mn_rope_kv_append_3d, reading the QKV projection as nz partial slabs.
extern "C" int mn_rope_kv_from_partials(const float* qkv, int64_t ldqkv, int nz, int64_t slab, int M, int n_q, int n_kv, int hd,
                                        int rope, const float* cos_tab, const float* sin_tab, const int32_t* row_seq,
                                        const int32_t* row_slot, const int32_t* row_pos, int sec_t, int sec_h, float q_scale,
                                        float* q_out, float* kv_cache, int64_t t_max, void* stream) {
  MN_CHECK_ARG(M >= 1 && n_q >= 1 && n_kv >= 1 && (hd == 64 || hd == 128) && nz >= 1, "mn_rope_kv_append: bad shape");
  MN_CHECK_ARG(qkv && row_seq && row_slot && q_out && kv_cache, "mn_rope_kv_append: null pointer");
  MN_CHECK_ARG(!rope || (cos_tab && sin_tab && row_pos), "mn_rope_kv_append: rope needs tables and positions");
  MN_CHECK_ARG(sec_t >= 0 && sec_h >= 0 && sec_t + sec_h <= hd / 2, "mn_rope_kv_append: bad rotary sections %d/%d", sec_t, sec_h);
  hipLaunchKernelGGL(rope_kv_append_kernel, dim3(M, n_q + 2 * n_kv), dim3(hd / 2), 0, mn_stream(stream), qkv, ldqkv, nz, slab,
                     n_q, n_kv, hd, rope, cos_tab, sin_tab, row_seq, row_slot, row_pos, M, sec_t, sec_h, q_scale, q_out,
                     kv_cache, t_max, g_kv_round_bf16);
  MN_CHECK_LAUNCH("mn_rope_kv_append");
  return MN_OK;
}

extern "C" int mn_rope_kv_append_3d(const float* qkv, int64_t ldqkv, int M, int n_q, int n_kv, int hd, int rope,
                                    const float* cos_tab, const float* sin_tab, const int32_t* row_seq,
                                    const int32_t* row_slot, const int32_t* row_pos, int sec_t, int sec_h, float q_scale,
                                    float* q_out, float* kv_cache, int64_t t_max, void* stream) {
  return mn_rope_kv_from_partials(qkv, ldqkv, 1, 0, M, n_q, n_kv, hd, rope, cos_tab, sin_tab, row_seq, row_slot, row_pos, sec_t,
                                  sec_h, q_scale, q_out, kv_cache, t_max, stream);
}

extern "C" int mn_rope_kv_append(const float* qkv, int64_t ldqkv, int M, int n_q, int n_kv, int hd, int rope,
                                 const float* cos_tab, const float* sin_tab, const int32_t* row_seq,
                                 const int32_t* row_slot, const int32_t* row_pos, float q_scale, float* q_out,
                                 float* kv_cache, int64_t t_max, void* stream) {
  return mn_rope_kv_append_3d(qkv, ldqkv, M, n_q, n_kv, hd, rope, cos_tab, sin_tab, row_seq, row_slot, row_pos, 0, 0, q_scale,
                              q_out, kv_cache, t_max, stream);
}

// -------------------------------------------------------------------------------------------
// Decode attention, split over keys. One wave per (row, q head, split); 4 waves per block
// cover 4 consecutive q heads (= one GQA group when n_q/n_kv == 4, so K/V lines are shared in L1).
// partial layout: [M][n_q][S][HD + 2]  (acc[HD], m, l)
// -------------------------------------------------------------------------------------------
// FUSED (round 4): the launch takes the RAW QKV projection instead of q — the rotary embedding of the row's q and new k, the
// q scale and the K / V append of the new token (rope_kv_append_kernel, one launch per layer and step before) happen here: every
// wave rotates its own q head into LDS; in the split that holds the new key, the first q head of each KV group writes the group's
// K / V line into the arena (n_q / n_kv in {1, 2, 4}: the 4 heads of a block never straddle a KV group, so the line's readers are
// waves of the SAME workgroup, ordered by vmcnt(0) + the barrier; the vector L1 holds no older copy of a line nobody read yet).
struct AttnFuse {
  const float* qkv; int64_t ldqkv; int nz; int64_t slab;        // raw QKV rows [(n_q + 2 n_kv) * HD], optionally nz K-slice partial slabs
  int rope; const float* cos_tab; const float* sin_tab;          // [n_pos, HD / 2]
  const int32_t* row_slot; const int32_t* row_pos; int sec_t, sec_h; float q_scale; int M;
  float* kv_cache_w;                                              // the arena, writable
};

template <int HD, bool FUSED>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(
    const float* __restrict__ q, int n_q, int n_kv, const float* __restrict__ kv_cache, int64_t t_max,
    const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_len, const uint8_t* __restrict__ key_mask,
    int64_t ld_mask, int S, int chunk_cap, float* __restrict__ partial, const AttnFuse fz) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ __attribute__((aligned(16))) float qs[FUSED ? 4 * HD : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x, s = blockIdx.z;
  const int h_raw = blockIdx.y * 4 + wave;
  const bool active = h_raw < n_q;          // inactive waves shadow the last head, store nothing
  const int h = active ? h_raw : n_q - 1;
  float* sc = sm + wave * chunk_cap;
  const int len = row_len[m];
  int chunk = (len + S - 1) / S;
  chunk = (chunk + 3) & ~3;
  const int j0 = s * chunk, j1 = min(len, j0 + chunk);
  const int kvh = h / (n_q / n_kv);
  const int64_t seq = row_seq[m];
  const float* Kb = kv_cache + ((seq * 2 + 0) * n_kv + kvh) * t_max * HD;
  const float* Vb = kv_cache + ((seq * 2 + 1) * n_kv + kvh) * t_max * HD;
  const float* qr = q + ((int64_t)m * n_q + h) * HD;
  const uint8_t* mk = key_mask ? key_mask + (int64_t)m * ld_mask : nullptr;
  float* out = partial + (((int64_t)m * n_q + h) * S + s) * (HD + 2);

  if constexpr (FUSED) {
    constexpr int half = HD / 2;
    const float* row = fz.qkv + (int64_t)m * fz.ldqkv;
    auto fetch = [&](int head, int i) {                 // element i of projection head `head` (q heads, then k heads, then v heads)
      const float* p_ = row + (int64_t)head * HD + i;
      float v = p_[0];
      for (int z = 1; z < fz.nz; ++z) v += p_[z * fz.slab];
      return v;
    };
    float c = 1.f, sn = 0.f;
    if (lane < half && fz.rope) {
      const int stream = fz.sec_t <= 0 ? 0 : (lane < fz.sec_t ? 0 : (lane < fz.sec_t + fz.sec_h ? 1 : 2));
      const int pos = fz.row_pos[stream * fz.M + m];
      c = fz.cos_tab[(int64_t)pos * half + lane];
      sn = fz.sin_tab[(int64_t)pos * half + lane];
    }
    if (lane < half) {                                  // this wave's q head: rotate, scale, park in LDS
      const float x1 = fetch(h, lane), x2 = fetch(h, lane + half);
      qs[wave * HD + lane] = (x1 * c - x2 * sn) * fz.q_scale;
      qs[wave * HD + lane + half] = (x2 * c + x1 * sn) * fz.q_scale;
    }
    const int slot = fz.row_slot[m];
    const int ratio = n_q / n_kv;
    // the new token's K / V line: written once per KV group, by the group's first q head, in the split that will read it
    if (active && h % ratio == 0 && slot >= j0 && slot < j0 + chunk && (uint64_t)slot < (uint64_t)t_max && lane < half) {
      const float k1 = fetch(n_q + kvh, lane), k2 = fetch(n_q + kvh, lane + half);
      float* kd = fz.kv_cache_w + (((seq * 2 + 0) * n_kv + kvh) * t_max + slot) * HD;
      float* vd = fz.kv_cache_w + (((seq * 2 + 1) * n_kv + kvh) * t_max + slot) * HD;
      kd[lane] = k1 * c - k2 * sn;
      kd[lane + half] = k2 * c + k1 * sn;
      vd[lane] = fetch(n_q + n_kv + kvh, lane);
      vd[lane + half] = fetch(n_q + n_kv + kvh, lane + half);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (callers pass row_len = row_slot + 1: the split that contains `slot` always exists)
  }

  // ---- scores: 16 lanes per key, 4 keys per wave-iteration
  constexpr int PER = HD / 16;  // floats per lane (8 for 128, 4 for 64)
  const int sub = lane & 15, kq = lane >> 4;
  float qv[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) qv[i] = FUSED ? qs[wave * HD + sub * PER + i] : qr[sub * PER + i];
  float mx = -INFINITY;
  for (int j = j0 + kq; j < j0 + chunk; j += 4) {
    float d = 0.f;
    const bool ok = j < j1;
    if (ok) {
      const float* kr = Kb + (int64_t)j * HD + sub * PER;
#pragma unroll
      for (int i = 0; i < PER; i += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + i);
        d = fmaf(kv.x, qv[i], d); d = fmaf(kv.y, qv[i + 1], d); d = fmaf(kv.z, qv[i + 2], d); d = fmaf(kv.w, qv[i + 3], d);
      }
    }
    d = row16_sum(d);           // the 16 lanes of a key are one DPP row
    const bool keep = ok && (!mk || mk[j] != 0);
    const float sv = keep ? d : -INFINITY;
    if (sub == 0 && (j - j0) < chunk_cap) sc[j - j0] = sv;
    mx = fmaxf(mx, sv);
  }
  mx = wave_max(mx);
  // ---- softmax numerators
  __syncthreads();
  float l = 0.f;
  const int n = max(0, j1 - j0);
  for (int j = lane; j < n; j += 64) {
    const float p = (mx == -INFINITY) ? 0.f : __expf(sc[j] - mx);
    sc[j] = p;
    l += p;
  }
  l = wave_sum(l);
  __syncthreads();
  // ---- PV: lane owns HD/64 dims
  constexpr int DPL = HD / 64;  // 2 or 1
  float acc[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) acc[i] = 0.f;
  const float* vr = Vb + (int64_t)j0 * HD + lane * DPL;
  int j = 0;
  for (; j + 4 <= n; j += 4) {
    float v[4][DPL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < DPL; ++i) v[u][i] = vr[(int64_t)(j + u) * HD + i];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float p = sc[j + u];
#pragma unroll
      for (int i = 0; i < DPL; ++i) acc[i] = fmaf(p, v[u][i], acc[i]);
    }
  }
  for (; j < n; ++j) {
    const float p = sc[j];
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = fmaf(p, vr[(int64_t)j * HD + i], acc[i]);
  }
  if (active) {
#pragma unroll
    for (int i = 0; i < DPL; ++i) out[lane * DPL + i] = acc[i];
    if (lane == 0) { out[HD] = mx; out[HD + 1] = l; }
  }
}

// GQA form for many rows: one wave per (row, KV head, split) serves all G = n_q / n_kv query heads of its group, so every
// K / V line is loaded ONCE per group instead of once per query head (the per-head kernel above re-reads them through L1:
// 4x the load instructions at G = 4).  Same partial layout, same combine kernel.  Block = 4 waves = 4 KV heads of one row.
template <int HD, int G>
__global__ __launch_bounds__(256) void attn_decode_gqa_kernel(
    const float* __restrict__ q, int n_q, int n_kv, const float* __restrict__ kv_cache, int64_t t_max,
    const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_len, const uint8_t* __restrict__ key_mask,
    int64_t ld_mask, int S, int chunk_cap, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x, s = blockIdx.z;
  const int kvh_raw = blockIdx.y * 4 + wave;
  const bool active = kvh_raw < n_kv;          // inactive waves shadow the last KV head, store nothing
  const int kvh = active ? kvh_raw : n_kv - 1;
  float* sc = sm + (size_t)wave * G * chunk_cap;            // [G][chunk_cap]
  const int len = row_len[m];
  int chunk = (len + S - 1) / S;
  chunk = (chunk + 3) & ~3;
  const int j0 = s * chunk, j1 = min(len, j0 + chunk);
  const int64_t seq = row_seq[m];
  const float* Kb = kv_cache + ((seq * 2 + 0) * n_kv + kvh) * t_max * HD;
  const float* Vb = kv_cache + ((seq * 2 + 1) * n_kv + kvh) * t_max * HD;
  const uint8_t* mk = key_mask ? key_mask + (int64_t)m * ld_mask : nullptr;
  constexpr int PER = HD / 16;                 // floats per lane in the score pass (16 lanes per key)
  const int sub = lane & 15, kq = lane >> 4;
  float qv[G][PER];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int i = 0; i < PER; ++i) qv[g][i] = q[((int64_t)m * n_q + kvh * G + g) * HD + sub * PER + i];
  float mx[G];
#pragma unroll
  for (int g = 0; g < G; ++g) mx[g] = -INFINITY;
#pragma unroll 2
  for (int j = j0 + kq; j < j0 + chunk; j += 4) {
    float d[G];
#pragma unroll
    for (int g = 0; g < G; ++g) d[g] = 0.f;
    const bool ok = j < j1;
    if (ok) {
      const float* kr = Kb + (int64_t)j * HD + sub * PER;
#pragma unroll
      for (int i = 0; i < PER; i += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + i);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          d[g] = fmaf(kv.x, qv[g][i], d[g]); d[g] = fmaf(kv.y, qv[g][i + 1], d[g]);
          d[g] = fmaf(kv.z, qv[g][i + 2], d[g]); d[g] = fmaf(kv.w, qv[g][i + 3], d[g]);
        }
      }
    }
    const bool keep = ok && (!mk || mk[j] != 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float v = d[g];
      v = row16_sum(v);         // the 16 lanes of a key are one DPP row
      const float sv = keep ? v : -INFINITY;
      if (sub == 0 && (j - j0) < chunk_cap) sc[g * chunk_cap + (j - j0)] = sv;
      mx[g] = fmaxf(mx[g], sv);
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) mx[g] = wave_max(mx[g]);
  __syncthreads();
  const int n = max(0, j1 - j0);
  float l[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float a = 0.f;
    for (int j = lane; j < n; j += 64) {
      const float pj = (mx[g] == -INFINITY) ? 0.f : __expf(sc[g * chunk_cap + j] - mx[g]);
      sc[g * chunk_cap + j] = pj;
      a += pj;
    }
    l[g] = wave_sum(a);
  }
  __syncthreads();
  constexpr int DPL = HD / 64;                 // dims per lane in the PV pass
  float acc[G][DPL];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[g][i] = 0.f;
  const float* vr = Vb + (int64_t)j0 * HD + lane * DPL;
  int j = 0;
  for (; j + 4 <= n; j += 4) {
    float v[4][DPL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < DPL; ++i) v[u][i] = vr[(int64_t)(j + u) * HD + i];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const f32x4 pj = *reinterpret_cast<const f32x4*>(sc + g * chunk_cap + j);      // chunk_cap % 4 == 0, j % 4 == 0
      const float pv[4] = {pj.x, pj.y, pj.z, pj.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[g][i] = fmaf(pv[u], v[u][i], acc[g][i]);
    }
  }
  for (; j < n; ++j) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float pj = sc[g * chunk_cap + j];
#pragma unroll
      for (int i = 0; i < DPL; ++i) acc[g][i] = fmaf(pj, vr[(int64_t)j * HD + i], acc[g][i]);
    }
  }
  if (active) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float* out = partial + (((int64_t)m * n_q + kvh * G + g) * S + s) * (HD + 2);
#pragma unroll
      for (int i = 0; i < DPL; ++i) out[lane * DPL + i] = acc[g][i];
      if (lane == 0) { out[HD] = mx[g]; out[HD + 1] = l[g]; }
    }
  }
}

// out (fp32 [M, n_q*HD]) and / or split (bf16 [2][M][n_q*HD]: hi rows then lo rows, the next GEMV's MFMA operand)
template <int HD>
__global__ void attn_decode_combine_kernel(const float* __restrict__ partial, int n_q, int S, float* __restrict__ out,
                                           bf16_t* __restrict__ split, int M, const float* __restrict__ kv_cache, int n_kv,
                                           int64_t t_max, const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_len) {
  const int m = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
  const float* p = partial + (((int64_t)m * n_q + h) * S) * (HD + 2);
  float l = 0.f, acc = 0.f;
  if (S <= 64) {
    // lane s of every wave holds split s: the statistics are one parallel load + wave reductions, and the weighted sum over
    // the splits runs on independent loads (the serial form cost 14 us at 32 splits: 64 dependent round trips to L2)
    const int lane = d & 63;
    float ms = -INFINITY, ls = 0.f;
    if (lane < S) { ms = p[lane * (HD + 2) + HD]; ls = p[lane * (HD + 2) + HD + 1]; }
    const float mx = wave_max(ms);
    const float f = (ms == -INFINITY) ? 0.f : __expf(ms - mx);
    l = wave_sum(f * ls);
    for (int s0 = 0; s0 < S; s0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s0 + j < S ? p[(s0 + j) * (HD + 2) + d] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(lane_f(f, (s0 + j) & 63), v[j], acc);      // uniform source lane: v_readlane, not ds_bpermute
    }
  } else {
    float mx = -INFINITY;
    for (int s = 0; s < S; ++s) mx = fmaxf(mx, p[s * (HD + 2) + HD]);
    for (int s = 0; s < S; ++s) {
      const float ms = p[s * (HD + 2) + HD];
      const float f = (ms == -INFINITY) ? 0.f : __expf(ms - mx);
      l = fmaf(f, p[s * (HD + 2) + HD + 1], l);
      acc = fmaf(f, p[s * (HD + 2) + d], acc);
    }
  }
  float v;
  if (l > 0.f) {
    v = acc / l;
  } else {
    // No attended key among the row's keys: the reference's additive finfo.min mask (modeling_bailing_moe.py:1466, :802-806) then
    // absorbs every score (s + finfo.min == finfo.min in fp32), the softmax is UNIFORM over the row's keys [0, len) and the output
    // is the mean of their V rows.  Degenerate (no caller builds such a row); computed here, off the fast path, for parity.
    const int len = row_len[m];
    const int kvh = h / (n_q / n_kv);
    const float* Vb = kv_cache + (((int64_t)row_seq[m] * 2 + 1) * n_kv + kvh) * t_max * HD;
    float sum = 0.f;
    for (int j = 0; j < len; ++j) sum += Vb[(int64_t)j * HD + d];
    v = len > 0 ? sum / (float)len : 0.f;
  }
  const int64_t o = ((int64_t)m * n_q + h) * HD + d;
  if (out) out[o] = v;
  if (split) {
    const bf16_t hi = f32_to_bf16(v);
    split[o] = hi;
    split[(int64_t)M * n_q * HD + o] = f32_to_bf16(v - bf16_to_f32(hi));
  }
}

// -------------------------------------------------------------------------------------------
// One-launch form for a few rows with short caches (the reference's call shapes: 1 row of text decode, the 2-3 CFG rows of an image,
// the semantic decoder's row; <= 1024 keys): one 16-wave workgroup per (row, q head).  The key range is split over the WAVES of the
// workgroup, their (acc, max, sum) partials meet in LDS — no partial round trip through memory and no combine launch (split 9.8 us +
// combine 4.7 us per layer at 2 rows before).  Scores and PV run on 16 / 8 keys' loads in flight per wave.
// FUSED: as in the split kernel the launch takes the raw QKV row; here every workgroup rotates its q head AND the new key of its KV
// group into LDS and attends to that key from there (the first q head of the group also writes the K / V line into the arena: no
// workgroup reads the new line from memory in this launch, so the heads of a group need no ordering among themselves).
// (A form with ALL of a wave's K and V loads issued before the first score — <= 24 keys per wave, 123 VGPRs — measured no faster:
// 2.046 vs 2.059 ms per 28-layer step at 1 row / 296 keys, slower at 40 keys; removed.  profiles/r04_attn_fused_ab.txt)
template <int HD, bool FUSED>
__global__ __launch_bounds__(1024) void attn_decode_one_kernel(
    const float* __restrict__ q, int n_q, int n_kv, const float* __restrict__ kv_cache, int64_t t_max,
    const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_len, const uint8_t* __restrict__ key_mask,
    int64_t ld_mask, int chunk_cap, float* __restrict__ out, bf16_t* __restrict__ split, int M, const AttnFuse fz) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // [16][chunk_cap] scores, [16][HD + 2] partials, [3][HD] q / new k / new v
  constexpr int NW = 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x, h = blockIdx.y;
  float* sc = sm + wave * chunk_cap;
  float* part = sm + NW * chunk_cap;
  float* qs = part + NW * (HD + 2);
  float* knew = qs + HD;
  float* vnew = knew + HD;
  const int len = row_len[m];
  int chunk = (len + NW - 1) / NW;
  chunk = (chunk + 3) & ~3;
  const int j0 = wave * chunk, j1 = min(len, j0 + chunk);
  const int kvh = h / (n_q / n_kv);
  const int64_t seq = row_seq[m];
  const float* Kb = kv_cache + ((seq * 2 + 0) * n_kv + kvh) * t_max * HD;
  const float* Vb = kv_cache + ((seq * 2 + 1) * n_kv + kvh) * t_max * HD;
  const uint8_t* mk = key_mask ? key_mask + (int64_t)m * ld_mask : nullptr;
  int slot = -1;                                 // FUSED: the new key lives in LDS, not in the arena
  if constexpr (FUSED) {
    constexpr int half = HD / 2;
    slot = fz.row_slot[m];
    if (wave < 3 && lane < half) {               // wave 0: q head h, wave 1: the group's new k, wave 2: its new v
      const float* row = fz.qkv + (int64_t)m * fz.ldqkv;
      const int head = wave == 0 ? h : (wave == 1 ? n_q + kvh : n_q + n_kv + kvh);
      const float* p_ = row + (int64_t)head * HD + lane;
      float x1 = 0.f, x2 = 0.f;
      for (int z0 = 0; z0 < fz.nz; z0 += 8) {      // eight slabs' loads in flight
        float a1[8], a2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          a1[j] = z0 + j < fz.nz ? p_[(int64_t)(z0 + j) * fz.slab] : 0.f;
          a2[j] = z0 + j < fz.nz ? p_[(int64_t)(z0 + j) * fz.slab + half] : 0.f;
        }
        for (int j = 0; j < 8; ++j) x1 += a1[j];                    // (the one-by-one loop's order: same bits)
        for (int j = 0; j < 8; ++j) x2 += a2[j];
      }
      if (fz.rope && wave < 2) {
        const int stream = fz.sec_t <= 0 ? 0 : (lane < fz.sec_t ? 0 : (lane < fz.sec_t + fz.sec_h ? 1 : 2));
        const int pos = fz.row_pos[stream * fz.M + m];
        const float c = fz.cos_tab[(int64_t)pos * half + lane], sn = fz.sin_tab[(int64_t)pos * half + lane];
        const float o1 = x1 * c - x2 * sn, o2 = x2 * c + x1 * sn;
        x1 = o1; x2 = o2;
      }
      if (wave == 0) { x1 *= fz.q_scale; x2 *= fz.q_scale; }
      float* dst = wave == 0 ? qs : (wave == 1 ? knew : vnew);
      dst[lane] = x1; dst[lane + half] = x2;
      if (wave > 0 && h % (n_q / n_kv) == 0 && (uint64_t)slot < (uint64_t)t_max) {     // the arena line, once per KV group
        float* ad = fz.kv_cache_w + (((seq * 2 + (wave - 1)) * n_kv + kvh) * t_max + slot) * HD;
        ad[lane] = x1; ad[lane + half] = x2;
      }
    }
    __syncthreads();
  }
  // ---- scores: 16 lanes per key, 4 keys per step, 4 steps' loads in flight
  constexpr int PER = HD / 16;
  constexpr int DPL = HD / 64;
  const int sub = lane & 15, kq = lane >> 4;
  const int n = max(0, j1 - j0);
  float qv[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) qv[i] = FUSED ? qs[sub * PER + i] : q[((int64_t)m * n_q + h) * HD + sub * PER + i];
  float mx = -INFINITY;
  for (int jb = j0; jb < j0 + chunk; jb += 16) {
    f32x4 kv[4][PER / 4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jb + u * 4 + kq;
#pragma unroll
      for (int i = 0; i < PER / 4; ++i) {
        kv[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < j1) {
          if (FUSED && j == slot) kv[u][i] = *reinterpret_cast<const f32x4*>(knew + sub * PER + i * 4);
          else kv[u][i] = *reinterpret_cast<const f32x4*>(Kb + (int64_t)j * HD + sub * PER + i * 4);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jb + u * 4 + kq;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < PER / 4; ++i) {
        d = fmaf(kv[u][i].x, qv[i * 4], d); d = fmaf(kv[u][i].y, qv[i * 4 + 1], d);
        d = fmaf(kv[u][i].z, qv[i * 4 + 2], d); d = fmaf(kv[u][i].w, qv[i * 4 + 3], d);
      }
      d = row16_sum(d);
      const bool keep = j < j1 && (!mk || mk[j] != 0);
      const float sv = keep ? d : -INFINITY;
      if (sub == 0 && j < j0 + chunk && (j - j0) < chunk_cap) sc[j - j0] = sv;
      mx = fmaxf(mx, sv);
    }
  }
  mx = wave_max(mx);
  __syncthreads();
  float l = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float p = (mx == -INFINITY) ? 0.f : __expf(sc[j] - mx);
    sc[j] = p;
    l += p;
  }
  l = wave_sum(l);
  __syncthreads();
  // ---- PV: lane owns HD / 64 dims, 8 keys' loads in flight
  float acc[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) acc[i] = 0.f;
  for (int jb = 0; jb < n; jb += 8) {
    float v[8][DPL];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < DPL; ++i) {
        const int j = j0 + jb + u;
        v[u][i] = 0.f;
        if (jb + u < n) v[u][i] = (FUSED && j == slot) ? vnew[lane * DPL + i] : Vb[(int64_t)j * HD + lane * DPL + i];
      }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float p = jb + u < n ? sc[jb + u] : 0.f;
#pragma unroll
      for (int i = 0; i < DPL; ++i) acc[i] = fmaf(p, v[u][i], acc[i]);
    }
  }
  float* pw = part + wave * (HD + 2);
#pragma unroll
  for (int i = 0; i < DPL; ++i) pw[lane * DPL + i] = acc[i];
  if (lane == 0) { pw[HD] = mx; pw[HD + 1] = l; }
  __syncthreads();
  // ---- the 16 waves' partials -> the head's output
  const int d = threadIdx.x;
  if (d >= HD) return;
  float mall = -INFINITY;
#pragma unroll
  for (int w = 0; w < NW; ++w) mall = fmaxf(mall, part[w * (HD + 2) + HD]);
  float lt = 0.f, at = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float ms = part[w * (HD + 2) + HD];
    const float fw = (ms == -INFINITY) ? 0.f : __expf(ms - mall);
    lt = fmaf(fw, part[w * (HD + 2) + HD + 1], lt);
    at = fmaf(fw, part[w * (HD + 2) + d], at);
  }
  float v;
  if (lt > 0.f) {
    v = at / lt;
  } else {      // no attended key: uniform over the row's keys (see attn_decode_combine_kernel)
    float sum = 0.f;
    for (int j = 0; j < len; ++j) sum += (FUSED && j == slot) ? vnew[d] : Vb[(int64_t)j * HD + d];
    v = len > 0 ? sum / (float)len : 0.f;
  }
  const int64_t o = ((int64_t)m * n_q + h) * HD + d;
  if (out) out[o] = v;
  if (split) {
    const bf16_t hi = f32_to_bf16(v);
    split[o] = hi;
    split[(int64_t)M * n_q * HD + o] = f32_to_bf16(v - bf16_to_f32(hi));
  }
}

// Key-range splits (flash-decoding): enough workgroups to fill the chip at few rows, none needed at hundreds of rows;
// one split never holds more than 4096 keys (its scores live in LDS).
static bool attn_use_gqa(int M, int n_q, int n_kv, int hd) { return M > 64 && hd == 128 && n_q == 4 * n_kv; }
// the one-launch form: <= g_one_rows rows whose caches hold <= 1024 keys (dev-library A/B: mn_attn_tune_one)
static int g_one_rows = 16;
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_attn_tune_one(int max_rows) { g_one_rows = max_rows; }
#endif
static bool attn_use_one(int M, int64_t t_max) { return M <= g_one_rows && t_max <= 1024; }
static int attn_splits(int M, int n_q, int64_t t_max) {
  int S = (int)mn_cdiv(t_max, 32);
  if (S > 32) S = 32;
  const int want = (int)mn_cdiv(4096, M);       // a few thousand waves on the chip whatever the row count
  if (S > want) S = want;
  const int need = (int)mn_cdiv(t_max, 1024);   // scores of a split live in LDS (4 heads x 1024 keys per wave at most)
  if (S < need) S = need;
  if (S < 1) S = 1;
  return S;
}

extern "C" size_t mn_attn_decode_workspace_bytes(int M, int n_q, int hd, int64_t t_max) {
  return (size_t)M * n_q * attn_splits(M, n_q, t_max) * (hd + 2) * sizeof(float);
}

static int attn_decode_launch(const float* q, const AttnFuse* fz, int M, int n_q, int n_kv, int hd, const float* kv_cache, int64_t t_max,
                              const int32_t* row_seq, const int32_t* row_len, const uint8_t* key_mask, int64_t ld_mask, float* out,
                              uint16_t* split, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st0 = mn_stream(stream);
  if (attn_use_one(M, t_max)) {                 // a few rows, short caches: one launch, the key split lives inside the workgroup
    int cc = (int)mn_cdiv(t_max, 16);
    cc = (cc + 3) & ~3;
    const size_t lds1 = ((size_t)16 * cc + 16 * (hd + 2) + 3 * hd) * sizeof(float);
    const AttnFuse nf{};
    const AttnFuse& fa = fz ? *fz : nf;
    const dim3 g1(M, n_q), b1(1024);
#define MN_ONE(HD_, F_) hipLaunchKernelGGL((attn_decode_one_kernel<HD_, F_>), g1, b1, lds1, st0, q, n_q, n_kv, kv_cache, t_max, row_seq, row_len, \
                                           key_mask, ld_mask, cc, out, split, M, fa)
    if (hd == 128) { if (fz) MN_ONE(128, true); else MN_ONE(128, false); }
    else           { if (fz) MN_ONE(64, true); else MN_ONE(64, false); }
#undef MN_ONE
    MN_CHECK_LAUNCH("mn_attn_decode(one launch)");
    return MN_OK;
  }
  const int S = attn_splits(M, n_q, t_max);
  const size_t need = (size_t)M * n_q * S * (hd + 2) * sizeof(float);
  if (workspace_bytes < need) { mn_set_error("mn_attn_decode: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  int chunk_cap = (int)mn_cdiv(t_max, S);
  chunk_cap = (chunk_cap + 3) & ~3;
  const size_t lds = (size_t)4 * chunk_cap * sizeof(float);
  float* partial = reinterpret_cast<float*>(workspace);
  dim3 grid(M, (n_q + 3) / 4, S);
  hipStream_t st = mn_stream(stream);
  const AttnFuse none{};
  if (fz) {
    if (hd == 128)
      hipLaunchKernelGGL((attn_decode_split_kernel<128, true>), grid, dim3(256), lds, st, q, n_q, n_kv, kv_cache, t_max, row_seq, row_len,
                         key_mask, ld_mask, S, chunk_cap, partial, *fz);
    else
      hipLaunchKernelGGL((attn_decode_split_kernel<64, true>), grid, dim3(256), lds, st, q, n_q, n_kv, kv_cache, t_max, row_seq, row_len,
                         key_mask, ld_mask, S, chunk_cap, partial, *fz);
  } else if (attn_use_gqa(M, n_q, n_kv, hd)) {
    hipLaunchKernelGGL((attn_decode_gqa_kernel<128, 4>), dim3(M, (n_kv + 3) / 4, S), dim3(256), lds * 4, st, q, n_q, n_kv, kv_cache,
                       t_max, row_seq, row_len, key_mask, ld_mask, S, chunk_cap, partial);
  } else if (hd == 128) {
    hipLaunchKernelGGL((attn_decode_split_kernel<128, false>), grid, dim3(256), lds, st, q, n_q, n_kv, kv_cache, t_max, row_seq,
                       row_len, key_mask, ld_mask, S, chunk_cap, partial, none);
  } else {
    hipLaunchKernelGGL((attn_decode_split_kernel<64, false>), grid, dim3(256), lds, st, q, n_q, n_kv, kv_cache, t_max, row_seq,
                       row_len, key_mask, ld_mask, S, chunk_cap, partial, none);
  }
  if (hd == 128)
    hipLaunchKernelGGL(attn_decode_combine_kernel<128>, dim3(M, n_q), dim3(128), 0, st, partial, n_q, S, out, split, M, kv_cache, n_kv, t_max,
                       row_seq, row_len);
  else
    hipLaunchKernelGGL(attn_decode_combine_kernel<64>, dim3(M, n_q), dim3(64), 0, st, partial, n_q, S, out, split, M, kv_cache, n_kv, t_max,
                       row_seq, row_len);
  MN_CHECK_LAUNCH("mn_attn_decode");
  return MN_OK;
}

// Internal (engine.hip): mn_attn_decode whose output can also (or only) be written as bf16 hi/lo rows.
extern "C" int mn_attn_decode_split(const float* q, int M, int n_q, int n_kv, int hd, const float* kv_cache,
                                    int64_t t_max, const int32_t* row_seq, const int32_t* row_len,
                                    const uint8_t* key_mask, int64_t ld_mask, float* out, uint16_t* split, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  MN_CHECK_ARG(M >= 1 && n_q >= 1 && n_kv >= 1 && n_q % n_kv == 0 && (hd == 64 || hd == 128), "mn_attn_decode: bad shape");
  MN_CHECK_ARG(q && kv_cache && row_seq && row_len && (out || split) && workspace, "mn_attn_decode: null pointer");
  return attn_decode_launch(q, nullptr, M, n_q, n_kv, hd, kv_cache, t_max, row_seq, row_len, key_mask, ld_mask, out, split, workspace,
                            workspace_bytes, stream);
}

// Internal (engine.hip): RoPE + q scale + KV append + masked decode attention in ONE split launch (+ the combine) from the raw QKV
// projection (optionally as nz K-slice partial slabs): what mn_rope_kv_append_3d / mn_rope_kv_from_partials followed by
// mn_attn_decode_split compute, one launch fewer per layer and step and no q round trip.  Needs row_len == row_slot + 1, at most 64
// rows (the many-row GQA kernel keeps the separate append) and n_q / n_kv in {1, 2, 4}; mn_attn_fused_ok says whether a shape qualifies.
// Where the append rides the attention launch (tools/exp/attn_fused_ab.py, profiles/r04_attn_fused_ab.txt): always in the one-launch form
// (every row count it serves gains 1-2 %: each workgroup keeps the new key in LDS, no hazard, no redundant reads worth counting); in
// the split form at ONE row only (2.00 -> 1.92 ms per 28-layer step; from 2 rows on every split re-reduces the QKV slabs: 2.59 -> 2.74).
static int g_attn_fuse = 1;                          // dev-library A/B switch: 0 = never fused
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_attn_tune_fuse(int on) { g_attn_fuse = on; }
#endif
extern "C" int mn_attn_fused_ok(int M, int n_q, int n_kv, int hd, int64_t t_max) {
  if (g_kv_round_bf16 || !g_attn_fuse) return 0;     // the bf16-KV measurement hook lives in the stand-alone append kernel
  if (!attn_use_one(M, t_max) && M > 1) return 0;
  const int ratio = n_kv > 0 && n_q % n_kv == 0 ? n_q / n_kv : 0;
  return M >= 1 && !attn_use_gqa(M, n_q, n_kv, hd) && (hd == 64 || hd == 128) && (ratio == 1 || ratio == 2 || ratio == 4);
}
extern "C" int mn_attn_decode_fused(const float* qkv, int64_t ldqkv, int nz, int64_t slab, int M, int n_q, int n_kv, int hd, int rope,
                                    const float* cos_tab, const float* sin_tab, const int32_t* row_seq, const int32_t* row_slot,
                                    const int32_t* row_pos, int sec_t, int sec_h, float q_scale, float* kv_cache, int64_t t_max,
                                    const int32_t* row_len, const uint8_t* key_mask, int64_t ld_mask, float* out, uint16_t* split,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  MN_CHECK_ARG(mn_attn_fused_ok(M, n_q, n_kv, hd, t_max) && nz >= 1, "mn_attn_decode_fused: unsupported shape");
  MN_CHECK_ARG(qkv && kv_cache && row_seq && row_slot && row_len && (out || split) && workspace, "mn_attn_decode_fused: null pointer");
  MN_CHECK_ARG(!rope || (cos_tab && sin_tab && row_pos), "mn_attn_decode_fused: rope needs tables and positions");
  MN_CHECK_ARG(sec_t >= 0 && sec_h >= 0 && sec_t + sec_h <= hd / 2, "mn_attn_decode_fused: bad rotary sections %d/%d", sec_t, sec_h);
  AttnFuse fz;
  fz.qkv = qkv; fz.ldqkv = ldqkv; fz.nz = nz; fz.slab = slab; fz.rope = rope; fz.cos_tab = cos_tab; fz.sin_tab = sin_tab;
  fz.row_slot = row_slot; fz.row_pos = row_pos; fz.sec_t = sec_t; fz.sec_h = sec_h; fz.q_scale = q_scale; fz.M = M; fz.kv_cache_w = kv_cache;
  return attn_decode_launch(nullptr, &fz, M, n_q, n_kv, hd, kv_cache, t_max, row_seq, row_len, key_mask, ld_mask, out, split, workspace,
                            workspace_bytes, stream);
}

extern "C" int mn_attn_decode(const float* q, int M, int n_q, int n_kv, int hd, const float* kv_cache,
                                 int64_t t_max, const int32_t* row_seq, const int32_t* row_len,
                                 const uint8_t* key_mask, int64_t ld_mask, float* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  MN_CHECK_ARG(out, "mn_attn_decode: null pointer");
  return mn_attn_decode_split(q, M, n_q, n_kv, hd, kv_cache, t_max, row_seq, row_len, key_mask, ld_mask, out, nullptr, workspace,
                              workspace_bytes, stream);
}

// engine.hip — composite device-side sequences of the decode path (gfx950).
//
// One C call enqueues every launch of: the rectified-flow sampler for one visual token
// (mn_rf_sample), one Bailing-MoE decoder-stack step (mn_llm_step) and one cached
// MingTok semantic-decoder step + linear_proj (mn_semdec_step).  Nothing here touches
// the host after enqueueing: positions, lengths, expert indices all live in device memory,
// so a whole visual token can be captured into a hipGraph by the caller.
#include <stdlib.h>
#include <string.h>

#include <array>
#include <initializer_list>

#include "common.h"

extern "C" size_t mn_attn_decode_workspace_bytes(int M, int n_q, int hd, int64_t t_max);
extern "C" int mn_stream_mfma(const uint16_t* Y, const uint16_t* W, float* P, int M, int Ntot, int K, void* stream);
extern "C" int mn_stream_mfma_slices(int M, int Ntot, int K);
extern "C" int mn_stream_mfma_grouped_slices(int G, int max_rows, int Ntot, int K);
extern "C" int mn_rope_kv_from_partials(const float* qkv, int64_t ldqkv, int nz, int64_t slab, int M, int n_q, int n_kv, int hd,
                                        int rope, const float* cos_tab, const float* sin_tab, const int32_t* row_seq,
                                        const int32_t* row_slot, const int32_t* row_pos, int sec_t, int sec_h, float q_scale,
                                        float* q_out, float* kv_cache, int64_t t_max, void* stream);
extern "C" int mn_attn_decode_split(const float* q, int M, int n_q, int n_kv, int hd, const float* kv_cache, int64_t t_max,
                                    const int32_t* row_seq, const int32_t* row_len, const uint8_t* key_mask, int64_t ld_mask,
                                    float* out, uint16_t* split, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int mn_attn_fused_ok(int M, int n_q, int n_kv, int hd, int64_t t_max);
extern "C" int mn_flash_prefill_gqa_hd128_f32(const float* q, const float* kv_layer, int64_t t_max, int n_q, int n_kv, const int32_t* span_tab,
                                              int n_spans, int max_len, float* out, uint16_t* split, int64_t split_lo_off, void* stream);
extern "C" int mn_attn_decode_fused(const float* qkv, int64_t ldqkv, int nz, int64_t slab, int M, int n_q, int n_kv, int hd, int rope,
                                    const float* cos_tab, const float* sin_tab, const int32_t* row_seq, const int32_t* row_slot,
                                    const int32_t* row_pos, int sec_t, int sec_h, float q_scale, float* kv_cache, int64_t t_max,
                                    const int32_t* row_len, const uint8_t* key_mask, int64_t ld_mask, float* out, uint16_t* split,
                                    void* workspace, size_t workspace_bytes, void* stream);
extern "C" int mn_stream_mfma_grouped(const uint16_t* Y, int y_rows, const uint16_t* W, int64_t w_stride, float* P,
                                      int p_rows, const int32_t* off, const int32_t* xrows, int G, int max_rows,
                                      int Ntot, int K, void* stream);
extern "C" size_t mn_skinny_workspace_bytes_w8(int M, int N, int K, int epilogue);

// ---- weight-format dispatch of the streaming launches (mingnative.h section 7): wfmt != 0 -> 8-bit bytes (e4m3 | int8) + fp32 row scales ----
static inline bool mn_w8(int wfmt) { return wfmt == MN_W_FP8_E4M3 || wfmt == MN_W_INT8 || wfmt == MN_W_NF4; }
// widths of a quantised matrix: K must hold whole 16-byte weight loads (NF4: whole 64-element absmax blocks)
static inline int mn_wq_kmult(int wfmt) { return wfmt == MN_W_NF4 ? 64 : 16; }
// scale-table floats per weight row: one row scale (e4m3 / int8) or one absmax per 64 k (NF4)
static inline int64_t mn_wq_scales_per_row(int wfmt, int K) { return wfmt == MN_W_NF4 ? K / 64 : 1; }
extern "C" int mn_stream_mfma_wq_slices(int wfmt, int M, int Ntot, int K);
extern "C" int mn_stream_mfma_wq(const uint16_t* Y, const uint8_t* Wq, const float* wscale, float* P, int M, int Ntot, int K, int wfmt, void* stream);
extern "C" int mn_stream_mfma_grouped_wq(const uint16_t* Y, int y_rows, const uint8_t* Wq, int64_t w_stride, const float* wscale, int64_t s_stride,
                                         float* P, int p_rows, const int32_t* off, const int32_t* xrows, int G, int max_rows, int Ntot, int K,
                                         int wfmt, void* stream);
static inline int stream_slices(int wfmt, int M, int Ntot, int K) { return mn_stream_mfma_wq_slices(wfmt, M, Ntot, K); }
static inline int stream_dense(int wfmt, const uint16_t* Y, const void* W, const float* wscale, float* P, int M, int Ntot, int K, void* stream) {
  return wfmt ? mn_stream_mfma_wq(Y, reinterpret_cast<const uint8_t*>(W), wscale, P, M, Ntot, K, wfmt, stream)
              : mn_stream_mfma(Y, reinterpret_cast<const uint16_t*>(W), P, M, Ntot, K, stream);
}
// grouped: w_stride counts weight ELEMENTS (= bytes for fp8), s_stride the row scales per group
static inline int stream_grouped(int wfmt, const uint16_t* Y, int y_rows, const void* W, int64_t w_stride, const float* wscale, int64_t s_stride,
                                 float* P, int p_rows, const int32_t* off, const int32_t* xrows, int G, int max_rows, int Ntot, int K,
                                 void* stream) {
  return wfmt ? mn_stream_mfma_grouped_wq(Y, y_rows, reinterpret_cast<const uint8_t*>(W), w_stride, wscale,
                                          s_stride * mn_wq_scales_per_row(wfmt, K), P, p_rows, off, xrows, G, max_rows, Ntot, K, wfmt, stream)
              : mn_stream_mfma_grouped(Y, y_rows, reinterpret_cast<const uint16_t*>(W), w_stride, P, p_rows, off, xrows, G, max_rows, Ntot,
                                       K, stream);
}

namespace {

struct Carver {  // carve 256-byte aligned pieces out of a caller-provided workspace
  char* base;
  size_t off, cap;
  bool dry;
  Carver(void* p, size_t n, bool d = false) : base((char*)p), off(0), cap(n), dry(d) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    T* r = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += bytes;
    return r;
  }
  bool ok() const { return dry || off <= cap; }
};

// ---------------- small fp32 row kernels ----------------
__global__ __launch_bounds__(256) void rmsnorm_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const bf16_t* __restrict__ g, float eps,
                                                          float* __restrict__ y, int64_t ldy, int D) {
  __shared__ float red[8];
  const int m = blockIdx.x;
  const float* xr = x + (int64_t)m * ldx;
  float ss = 0.f;
  for (int k = threadIdx.x; k < D; k += 256) { float v = xr[k]; ss += v * v; }
  ss = block_sum(ss, red);
  const float rstd = rsqrtf(ss / (float)D + eps);
  for (int k = threadIdx.x; k < D; k += 256) y[(int64_t)m * ldy + k] = xr[k] * rstd * bf16_to_f32(g[k]);
}

__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const bf16_t* __restrict__ g, const bf16_t* __restrict__ b,
                                                            float eps, float* __restrict__ y, int64_t ldy, int D) {
  __shared__ float red[8];
  const int m = blockIdx.x;
  const float* xr = x + (int64_t)m * ldx;
  float s = 0.f;
  for (int k = threadIdx.x; k < D; k += 256) s += xr[k];
  const float mean = block_sum(s, red) / (float)D;
  float ss = 0.f;
  for (int k = threadIdx.x; k < D; k += 256) { float d = xr[k] - mean; ss += d * d; }
  ss = block_sum(ss, red);
  const float rstd = rsqrtf(ss / (float)D + eps);
  for (int k = threadIdx.x; k < D; k += 256) {
    float v = (xr[k] - mean) * rstd;
    if (g) v *= bf16_to_f32(g[k]);
    if (b) v += bf16_to_f32(b[k]);
    y[(int64_t)m * ldy + k] = v;
  }
}

// x[r] = noise[image(r)] * temperature; rows are image-major: image = r / rows_per_image
__global__ void rf_init_x_kernel(const float* __restrict__ noise, float temperature, float* __restrict__ x, int rows,
                                 int target, int rpi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * target) x[i] = noise[(i / target / rpi) * target + i % target] * temperature;
}

// Y[s*rows + r, k] = silu(temb[s, k] + c[r, k]) split into bf16 hi (rows 0..SR-1) and lo (rows SR..2SR-1):
// the input of every adaLN projection of every Euler step (diff_loss_rf_swiglu.py:263-266, 376).
__global__ void rf_build_y_kernel(const float* __restrict__ temb, const float* __restrict__ c, bf16_t* __restrict__ y,
                                  int steps, int rows, int w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t SR = (int64_t)steps * rows;
  if (i >= SR * w) return;
  const int k = (int)(i % w), sr = (int)(i / w), s = sr / rows, r = sr % rows;
  const float v = silu_f(temb[(int64_t)s * w + k] + c[(int64_t)r * w + k]);
  const bf16_t hi = f32_to_bf16(v);
  y[i] = hi;
  y[SR * w + i] = f32_to_bf16(v - bf16_to_f32(hi));
}

// ---- glue kernels of the matrix-core RF chain (rows >= 5): they sit between two weight-streaming launches and
// fuse "reduce the K-slice partials + bias + epilogue of GEMV i" with "prologue + bf16 hi/lo split of GEMV i+1".
// P: [nz][M][Ntot] fp32 partials (stream_mfma.hip); Y: [2][M][K] bf16 (hi rows, lo rows).
// row_range (optional, device): only rows [row_range[0], row_range[1]) are live (expert parallelism: the sorted positions of the
// rank's own experts; the other rows of P were never written and are not touched).
__global__ __launch_bounds__(256) void rf_glue_swiglu_split_kernel(const float* __restrict__ P, int nz, int M, int hidden,
                                                                   const bf16_t* __restrict__ b12, bf16_t* __restrict__ Y,
                                                                   const int32_t* __restrict__ row_range = nullptr,
                                                                   const int32_t* __restrict__ row_range_end = nullptr) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;    // 4 consecutive columns per thread (hidden % 4 == 0)
  if (i >= (int64_t)M * hidden) return;
  const int m = (int)(i / hidden), n = (int)(i % hidden);
  if (row_range && (m < *row_range || m >= *row_range_end)) return;
  const int64_t slab = (int64_t)M * 2 * hidden;
  f4 y1 = {0.f, 0.f, 0.f, 0.f}, y2 = {0.f, 0.f, 0.f, 0.f};
  if (b12) {
    y1 = f4{bf16_to_f32(b12[n]), bf16_to_f32(b12[n + 1]), bf16_to_f32(b12[n + 2]), bf16_to_f32(b12[n + 3])};
    y2 = f4{bf16_to_f32(b12[n + hidden]), bf16_to_f32(b12[n + hidden + 1]), bf16_to_f32(b12[n + hidden + 2]),
            bf16_to_f32(b12[n + hidden + 3])};
  }
  const float* p1 = P + (int64_t)m * 2 * hidden + n;
  const float* p2 = p1 + hidden;
  int z = 0;
  for (; z + 4 <= nz; z += 4) {                    // 8 independent 16-byte loads in flight
    const f4 a0 = *reinterpret_cast<const f4*>(p1 + (z + 0) * slab), b0 = *reinterpret_cast<const f4*>(p2 + (z + 0) * slab);
    const f4 a1 = *reinterpret_cast<const f4*>(p1 + (z + 1) * slab), b1 = *reinterpret_cast<const f4*>(p2 + (z + 1) * slab);
    const f4 a2 = *reinterpret_cast<const f4*>(p1 + (z + 2) * slab), b2 = *reinterpret_cast<const f4*>(p2 + (z + 2) * slab);
    const f4 a3 = *reinterpret_cast<const f4*>(p1 + (z + 3) * slab), b3 = *reinterpret_cast<const f4*>(p2 + (z + 3) * slab);
    y1 += (a0 + a1) + (a2 + a3);
    y2 += (b0 + b1) + (b2 + b3);
  }
  for (; z < nz; ++z) {
    y1 += *reinterpret_cast<const f4*>(p1 + z * slab);
    y2 += *reinterpret_cast<const f4*>(p2 + z * slab);
  }
  bf16_t hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = silu_f(y1[j]) * y2[j];
    hi[j] = f32_to_bf16(v);
    lo[j] = f32_to_bf16(v - bf16_to_f32(hi[j]));
  }
  *reinterpret_cast<u2*>(Y + i) = u2{(uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16)};
  *reinterpret_cast<u2*>(Y + (int64_t)M * hidden + i) =
      u2{(uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16)};
}

// One block per row.  P != NULL: h[m] += gate[m] * (sum_z P + b3)   (ResBlock residual, diff_loss:272);
// then Y = split( LayerNorm(h[m]; g?, b?) * (1 + scale[m]) + shift[m] )  — the next GEMV's modulated input (:270,290).
__global__ __launch_bounds__(1024) void rf_glue_resid_ln_split_kernel(
    const float* __restrict__ P, int nz, int M, int w, const bf16_t* __restrict__ b3, const float* __restrict__ gate,
    float* __restrict__ h, const bf16_t* __restrict__ ln_g, const bf16_t* __restrict__ ln_b,
    const float* __restrict__ shift, const float* __restrict__ scale, int64_t ldmod, bf16_t* __restrict__ Y) {
  __shared__ float red[32];
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int m = blockIdx.x, tid = threadIdx.x;
  const int col = tid * 4;                       // w <= 4096, w % 4 == 0: one float4 of columns per thread
  const bool act = col < w;
  f4 hv = {0.f, 0.f, 0.f, 0.f};
  float s = 0.f;
  // Every operand of the row is requested up front — the stream value, all nz partial slabs (nz <= 16 in ONE batch of independent
  // loads), the gate, the modulation and the LayerNorm parameters: the kernel is one memory round trip + two block reductions.
  // (Loading scale / shift / ln after the reductions, behind their barriers, cost a second round trip: 7.1 us per launch, 192 per token.)
  f4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
  u2 lg = {0u, 0u}, lb = {0u, 0u}, b3v = {0u, 0u};
  if (act) {
    hv = *reinterpret_cast<const f4*>(h + (int64_t)m * w + col);
    sc = *reinterpret_cast<const f4*>(scale + (int64_t)m * ldmod + col);
    sh = *reinterpret_cast<const f4*>(shift + (int64_t)m * ldmod + col);
    if (ln_g) lg = *reinterpret_cast<const u2*>(ln_g + col);
    if (ln_b) lb = *reinterpret_cast<const u2*>(ln_b + col);
    if (P) {
      gv = *reinterpret_cast<const f4*>(gate + (int64_t)m * ldmod + col);
      b3v = *reinterpret_cast<const u2*>(b3 + col);
      f4 y = {bf16lo_to_f32(b3v.x), bf16hi_to_f32(b3v.x), bf16lo_to_f32(b3v.y), bf16hi_to_f32(b3v.y)};
      const float* pp = P + (int64_t)m * w + col;
      const int64_t slab = (int64_t)M * w;
      int z = 0;
      for (; z + 16 <= nz; z += 16) {            // 16 independent 16-byte loads in flight
        f4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f4*>(pp + (z + j) * slab);
        y += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
             (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
      }
      for (; z + 8 <= nz; z += 8) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f4*>(pp + (z + j) * slab);
        y += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      for (; z + 4 <= nz; z += 4) {
        const f4 a = *reinterpret_cast<const f4*>(pp + (z + 0) * slab), b = *reinterpret_cast<const f4*>(pp + (z + 1) * slab);
        const f4 c = *reinterpret_cast<const f4*>(pp + (z + 2) * slab), d = *reinterpret_cast<const f4*>(pp + (z + 3) * slab);
        y += (a + b) + (c + d);
      }
      for (; z < nz; ++z) y += *reinterpret_cast<const f4*>(pp + z * slab);
      hv += gv * y;
      *reinterpret_cast<f4*>(h + (int64_t)m * w + col) = hv;
    }
    s = (hv.x + hv.y) + (hv.z + hv.w);
  }
  const float mean = block_sum(s, red) / (float)w;
  float ss = 0.f;
  if (act) { const f4 d = hv - mean; ss = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w; }
  const float rstd = rsqrtf(block_sum(ss, red) / (float)w + 1e-6f);
  if (act) {
    float v[4] = {(hv.x - mean) * rstd, (hv.y - mean) * rstd, (hv.z - mean) * rstd, (hv.w - mean) * rstd};
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float lgv[4] = {bf16lo_to_f32(lg.x), bf16hi_to_f32(lg.x), bf16lo_to_f32(lg.y), bf16hi_to_f32(lg.y)};
    const float lbv[4] = {bf16lo_to_f32(lb.x), bf16hi_to_f32(lb.x), bf16lo_to_f32(lb.y), bf16hi_to_f32(lb.y)};
    bf16_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ln_g) v[j] *= lgv[j];
      if (ln_b) v[j] += lbv[j];
      v[j] = v[j] * (1.0f + scv[j]) + shv[j];
      hi[j] = f32_to_bf16(v[j]);
      lo[j] = f32_to_bf16(v[j] - bf16_to_f32(hi[j]));
    }
    u2 ph = {(uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16)};
    u2 pl = {(uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16)};
    *reinterpret_cast<u2*>(Y + (int64_t)m * w + col) = ph;
    *reinterpret_cast<u2*>(Y + (int64_t)(M + m) * w + col) = pl;
  }
}

__global__ void rf_glue_bias_out_kernel(const float* __restrict__ P, int nz, int M, int N, const bf16_t* __restrict__ b,
                                        float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  float y = bf16_to_f32(b[i % N]);
  for (int z0 = 0; z0 < nz; z0 += 4) {              // four slabs' loads in flight
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = z0 + j < nz ? P[(int64_t)(z0 + j) * M * N + i] : 0.f;
    for (int j = 0; j < 4; ++j) y += t[j];                      // (the one-by-one loop's order: same bits)
  }
  out[i] = y;
}

// CFG combine + Euler step (diff_loss_rf_swiglu.py:144-179): v rows = [cond, uncond, text_uncond]
// one block per image; `rows` = CFG rows of ONE image
__global__ void rf_euler_kernel(const float* __restrict__ v0, float* __restrict__ x0, int rows, int target,
                                float text_cfg, float image_cfg, float step) {
  const int i = threadIdx.x;
  if (i >= target) return;
  const float* v = v0 + (int64_t)blockIdx.x * rows * target;
  float* x = x0 + (int64_t)blockIdx.x * rows * target;
  float vg;
  if (rows == 3) {
    const float vc = v[i], vu = v[target + i], vtu = v[2 * target + i];
    vg = vu + image_cfg * (vtu - vu) + text_cfg * (vc - vtu);
  } else if (rows == 2) {
    const float vc = v[i], vu = v[target + i];
    vg = vu + text_cfg * (vc - vu);
  } else {
    vg = v[i];
  }
  for (int r = 0; r < rows; ++r) x[r * target + i] += vg * step;
}

// The boundary between two Euler steps of the matrix-core chain in ONE launch (one workgroup per row; five launches before: slab
// sum + bias, CFG + Euler, the three launches of the input projection, LayerNorm-modulate + split):
//   !first: v = fin_b + sum_z P[z] (the previous step's final-layer slabs), CFG combine over the image's rows, x[m] += vg * step
//   h[m] = in_w x[m] + in_b   (diff_loss_rf_swiglu.py:371; target <= 256 inputs per output: a dot product per thread)
//   Y = split( (LayerNorm(h[m]) * ln_g + ln_b) * (1 + scale[m]) + shift[m] )  — block 0's modulated input (:270, 290)
// Every workgroup recomputes its image's vg (rpi x target sums of nz floats) and updates only its own row of x.
__global__ __launch_bounds__(1024) void rf_step_boundary_kernel(
    const float* __restrict__ P, int nz, int M, int T, const bf16_t* __restrict__ fin_b, float* __restrict__ x, int rpi, float text_cfg,
    float image_cfg, float step, int first, const bf16_t* __restrict__ in_w, const bf16_t* __restrict__ in_b, int w, float* __restrict__ h,
    const bf16_t* __restrict__ ln_g, const bf16_t* __restrict__ ln_b, const float* __restrict__ shift, const float* __restrict__ scale,
    int64_t ldmod, bf16_t* __restrict__ Y) {
  __shared__ float red[32];
  __shared__ float xs[256];
  const int m = blockIdx.x, tid = threadIdx.x;
  if (tid < T) {
    float xv = x[(int64_t)m * T + tid];
    if (!first) {
      const int r0 = (m / rpi) * rpi;
      float v[3] = {0.f, 0.f, 0.f};
      const float fb = bf16_to_f32(fin_b[tid]);
      // all rows' slabs in batches of 8 independent loads (one by one this was rpi x nz = 24 dependent round trips: 16 us of launch)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (r < rpi) {
          float y = fb;
          for (int z0 = 0; z0 < nz; z0 += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = z0 + j < nz ? P[((int64_t)(z0 + j) * M + r0 + r) * T + tid] : 0.f;
            for (int j = 0; j < 8; ++j) y += t[j];                    // (the one-by-one loop's order: same bits)
          }
          v[r] = y;
        }
      }
      const float vg = rpi == 3 ? v[1] + image_cfg * (v[2] - v[1]) + text_cfg * (v[0] - v[2])
                                : (rpi == 2 ? v[1] + text_cfg * (v[0] - v[1]) : v[0]);
      xv += vg * step;
      x[(int64_t)m * T + tid] = xv;
    }
    xs[tid] = xv;
  }
  // (the weights of this thread's columns do not depend on x: with the usual 32 inputs they are requested before the barrier, so the
  //  launch is one memory round trip + the reductions instead of three dependent ones: 16 -> 9 us)
  mn_u4_t wq[4][4];
  float bq[4];
  const bool pre = T == 32;
  if (pre) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = tid + c * 1024;
      bq[c] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) wq[c][j] = mn_u4_t{0u, 0u, 0u, 0u};
      if (n < w) {
        bq[c] = bf16_to_f32(in_b[n]);
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[c][j] = *reinterpret_cast<const mn_u4_t*>(in_w + (int64_t)n * 32 + j * 8);
      }
    }
  }
  __syncthreads();
  float hv[4];                                   // w <= 4096: columns tid, tid + 1024, ...
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = tid + c * 1024;
    hv[c] = 0.f;
    if (n < w) {
      float a;
      if (pre) {
        a = bq[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const mn_u4_t q = wq[c][j];
          const float* xk = xs + j * 8;
          a = fmaf(bf16lo_to_f32(q.x), xk[0], a); a = fmaf(bf16hi_to_f32(q.x), xk[1], a);
          a = fmaf(bf16lo_to_f32(q.y), xk[2], a); a = fmaf(bf16hi_to_f32(q.y), xk[3], a);
          a = fmaf(bf16lo_to_f32(q.z), xk[4], a); a = fmaf(bf16hi_to_f32(q.z), xk[5], a);
          a = fmaf(bf16lo_to_f32(q.w), xk[6], a); a = fmaf(bf16hi_to_f32(q.w), xk[7], a);
        }
      } else {
        const bf16_t* wr = in_w + (int64_t)n * T;
        a = bf16_to_f32(in_b[n]);
        for (int k = 0; k < T; k += 8) {           // T % 8 == 0 (host check): 16-byte rows
          const mn_u4_t q = *reinterpret_cast<const mn_u4_t*>(wr + k);
          a = fmaf(bf16lo_to_f32(q.x), xs[k], a); a = fmaf(bf16hi_to_f32(q.x), xs[k + 1], a);
          a = fmaf(bf16lo_to_f32(q.y), xs[k + 2], a); a = fmaf(bf16hi_to_f32(q.y), xs[k + 3], a);
          a = fmaf(bf16lo_to_f32(q.z), xs[k + 4], a); a = fmaf(bf16hi_to_f32(q.z), xs[k + 5], a);
          a = fmaf(bf16lo_to_f32(q.w), xs[k + 6], a); a = fmaf(bf16hi_to_f32(q.w), xs[k + 7], a);
        }
      }
      hv[c] = a;
      h[(int64_t)m * w + n] = a;
      sum += a;
    }
  }
  float lgq[4], lbq[4], scq[4], shq[4];           // requested before the reductions' barriers
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = tid + c * 1024;
    lgq[c] = 1.f; lbq[c] = 0.f; scq[c] = 0.f; shq[c] = 0.f;
    if (n < w) {
      if (ln_g) lgq[c] = bf16_to_f32(ln_g[n]);
      if (ln_b) lbq[c] = bf16_to_f32(ln_b[n]);
      scq[c] = scale[(int64_t)m * ldmod + n];
      shq[c] = shift[(int64_t)m * ldmod + n];
    }
  }
  const float mean = block_sum(sum, red) / (float)w;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (tid + c * 1024 < w) { const float d = hv[c] - mean; ss += d * d; }
  const float rstd = rsqrtf(block_sum(ss, red) / (float)w + 1e-6f);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = tid + c * 1024;
    if (n < w) {
      float v = (hv[c] - mean) * rstd;
      v = v * lgq[c] + lbq[c];
      v = v * (1.0f + scq[c]) + shq[c];
      const bf16_t hi = f32_to_bf16(v);
      Y[(int64_t)m * w + n] = hi;
      Y[(int64_t)(M + m) * w + n] = f32_to_bf16(v - bf16_to_f32(hi));
    }
  }
}

// semantic decoder input: de-normalise, Linear(in_dim -> D) + channel-repeat shortcut
// (modeling_mingtok.py:168; vision_transformer.py:373-380)
__global__ void semdec_in_kernel(const float* __restrict__ latent, int in_dim, float scale, float mean,
                                 const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, float* __restrict__ out,
                                 int D) {
  const int m = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= D) return;
  const float* lr = latent + (int64_t)m * in_dim;
  float acc = bf16_to_f32(b[n]);
  for (int k = 0; k < in_dim; ++k) acc = fmaf(bf16_to_f32(w[(int64_t)n * in_dim + k]), lr[k] * scale + mean, acc);
  acc += lr[n / (D / in_dim)] * scale + mean;
  out[(int64_t)m * D + n] = acc;
}

__global__ void copy_f32_kernel(const float* __restrict__ a, float* __restrict__ b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}

// latent_out[img] = x[img * rpi] (every CFG row of an image carries the same latent)
__global__ void rf_gather_latent_kernel(const float* __restrict__ x, float* __restrict__ out, int n_images, int rpi, int target) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_images * target) out[i] = x[(int64_t)(i / target) * rpi * target + i % target];
}

// rows copy: b[m] = a[(m / row_div) * lda]  (lda == 0 broadcasts one row; row_div > 1 shares one source row
// between the CFG rows of an image)
__global__ void copy_rows_f32_kernel(const float* __restrict__ a, int64_t lda, int row_div, float* __restrict__ b, int M, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)M * D) b[i] = a[((i / D) / row_div) * lda + (i % D)];
}

__global__ void rows_advance_kernel(int32_t* a, int32_t* b, int32_t* c, int M, int delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) {
    if (a) a[i] += delta;
    if (b) b[i] += delta;
    if (c) c[i] += delta;
  }
}

// scratch for the M > 8 route of mn_skinny_gemm; set by the running composite (single-threaded host code)
thread_local void* t_sk_ws = nullptr;
thread_local size_t t_sk_ws_bytes = 0;

mn_skinny_args sk(const float* x, int64_t ldx, const bf16_t* w, int64_t ldw, const bf16_t* bias, float* out,
                  int64_t ldo, int M, int N, int K) {
  mn_skinny_args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.bias = bias; a.out = out; a.ldo = ldo;
  a.M = M; a.N = N; a.K = K;
  a.ws = t_sk_ws; a.ws_bytes = t_sk_ws_bytes;
  return a;
}

size_t sk_ws_need(int M, std::initializer_list<std::array<int, 3>> shapes) {   // {N, K, epilogue}
  size_t mx = 0;
  for (auto& s : shapes) {
    const size_t b = mn_skinny_workspace_bytes(M, s[0], s[1], s[2]);
    if (b > mx) mx = b;
  }
  return mx;
}

#define MN_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != MN_OK) return rc__; \
  } while (0)

}  // namespace

#include "stream_fuse.h"
#include "wide_glue.h"
#include "wide_rf.inl"

// ===========================================================================================
// Rectified-flow head
// ===========================================================================================
// rows >= 5 run the RF blocks as the matrix-core chain; its glue kernels move 4 columns (16 bytes) per thread
// (1 row in bf16: the fp32-FMA kernels, unless the shape runs K-complete — then the matrix-core launches win: 7.40 -> 6.3 ms per call)
static bool rf_chain_ok(const mn_rf_head* h, int rows) {
  return (h->wfmt || mn_skinny_workspace_bytes(rows, h->w, h->w, 0) > 0 || rf_kc_ok(h->wfmt, rows, h->w, h->hidden)) && h->w <= 4096 &&
         (h->w % 4) == 0 && (h->hidden % 4) == 0;
}
// fp8 weight mode: the RF blocks must be able to run as the matrix-core chain (the fp32-FMA kernels read bf16 rows)
static bool rf_fp8_ok(const mn_rf_head* h) {
  return mn_w8(h->wfmt) && h->w12_scale && h->w3_scale && h->w <= 4096 && (h->w % mn_wq_kmult(h->wfmt)) == 0 &&
         (h->hidden % mn_wq_kmult(h->wfmt)) == 0;
}

// <= 4 rows (the CFG rows of one image — the reference's call shape): the SwiGLU glue launch is folded into w3's prologue
// (stream_fuse.h): three launches per ResBlock instead of four.  g_rf_fuse: dev-library A/B switch.
static int g_rf_fuse = 1, g_rf_boundary = 1, g_rf_kc = 1, g_rf_persist = 1, g_rf_whole = 1;
static void g_rf_ada_stream_set(int v);
#ifdef MN_DEV_HOOKS
// bit 0: SwiGLU glue fused into w3; bit 1: one-launch Euler-step boundary; bit 2: adaLN NOT streamed; bit 3: K-complete chain OFF
extern "C" MN_DEV_API void mn_rf_tune_fuse(int on) { g_rf_fuse = on & 1; g_rf_boundary = (on >> 1) & 1; g_rf_ada_stream_set((on >> 2) & 1 ? 0 : 1); g_rf_kc = (on >> 3) & 1 ? 0 : 1; g_rf_persist = (on >> 4) & 1 ? 0 : 1; g_rf_whole = (on >> 5) & 1 ? 0 : 1; }
#endif
static bool rf_fused_shape_ok(const mn_rf_head* h, int rows) {
  return rows <= FUSE_MAX_ROWS && rf_chain_ok(h, rows) &&
         stream_fused_ok(h->wfmt, rows, h->w, h->hidden, stream_slices(h->wfmt, rows, 2 * h->hidden, h->w));
}

// fp8 adaLN: the modulations of all Euler steps as ONE streaming launch on e4m3 bytes when their rows fit it
static bool rf_ada_w8(const mn_rf_head* h, int rows) { return h->wfmt && h->ada_q && h->ada_scale && (int64_t)h->steps * rows <= 64; }
// bf16 adaLN at <= 64 (step, row) pairs (the CFG rows of one or two images): the streaming launch (K-slice form to 32 pairs, K-loop form
// above) reads the 0.72 GB once at ~4 TB/s where the 128-tile hi/lo GEMM manages 2.7 — RF sampler 7.49 -> 7.30 ms at 2 rows, 7.70 ->
// 7.51 at 3, 7.74 -> 7.54 at 4 (tools/exp/rf_fused_chain.py); g_rf_ada_stream: dev-library A/B switch
static int g_rf_ada_stream = 1;
static void g_rf_ada_stream_set(int v) { g_rf_ada_stream = v; }
static bool rf_ada_stream(const mn_rf_head* h, int rows) {
  return g_rf_ada_stream && !h->wfmt && (int64_t)h->steps * rows <= 64 && (h->w % 8) == 0;
}

static size_t rf_carve(const mn_rf_head* h, int rows, void* ws, size_t cap, float** z, float** c, float** ada,
                       float** hh, float** hid, float** v, float** x, bf16_t** y, unsigned** bar, char** skws,
                       size_t* skws_bytes, bf16_t** ya, bf16_t** yb, float** pbuf, float** pada = nullptr, float** pbuf3 = nullptr) {
  Carver cv(ws, cap, ws == nullptr);
  const int A = h->depth * 3 * h->w + 2 * h->w;
  *z = cv.take<float>((size_t)rows * h->z_dim);
  *c = cv.take<float>((size_t)rows * h->w);
  *ada = cv.take<float>((size_t)h->steps * rows * A);       // modulations of every Euler step
  *y = cv.take<bf16_t>((size_t)2 * h->steps * rows * h->w);
  *bar = cv.take<unsigned>(RF_PERSIST_BAR_WORDS);
  *hh = cv.take<float>((size_t)rows * h->w);
  *hid = cv.take<float>((size_t)rows * h->hidden);
  *v = cv.take<float>((size_t)rows * h->target);
  *x = cv.take<float>((size_t)rows * h->target);
  *skws_bytes = sk_ws_need(rows, {{h->z_dim, h->llm_hidden, 0}, {h->w, h->z_dim, 0}, {h->hidden, h->w, MN_EPI_SWIGLU},
                                   {h->w, h->hidden, 0}, {h->target, h->w, 0}, {h->w, h->target, 0}});
  *skws = cv.take<char>(*skws_bytes);
  // matrix-core chain (rows >= 5): split activations of both GEMVs and the K-slice partial slabs
  const bool chain = rf_chain_ok(h, rows);
  const size_t p12 = (size_t)stream_slices(h->wfmt, rows, 2 * h->hidden, h->w) * 2 * h->hidden;
  const size_t p3 = (size_t)stream_slices(h->wfmt, rows, h->w, h->hidden) * h->w;
  const size_t pf = (size_t)mn_stream_mfma_slices(rows, h->target, h->w) * h->target;
  const size_t pmax = p12 > p3 ? (p12 > pf ? p12 : pf) : (p3 > pf ? p3 : pf);
  *ya = cv.take<bf16_t>(chain ? (size_t)2 * rows * h->w : 0);
  *yb = cv.take<bf16_t>(chain ? (size_t)2 * rows * h->hidden : 0);
  *pbuf = cv.take<float>(chain ? pmax * rows : 0);
  const int SRn = h->steps * rows;
  float* pa = cv.take<float>(rf_ada_w8(h, rows) ? (size_t)stream_slices(h->wfmt, SRn, A, h->w) * SRn * A
                             : ((!h->wfmt && (int64_t)SRn <= 64 && (h->w % 8) == 0) ? (size_t)mn_stream_mfma_slices(SRn, A, h->w) * SRn * A : 0));
  if (pada) *pada = pa;
  // fused w3: its slabs live beside w12's (its prologue reads those while other workgroups already write w3's); the workspace does
  // not depend on the A/B switch
  float* p3b = cv.take<float>(rf_fused_shape_ok(h, rows) ? p3 * rows : 0);
  if (pbuf3) *pbuf3 = p3b;
  return cv.off;
}

extern "C" size_t mn_rf_workspace_bytes(const mn_rf_head* h, int rows) {
  if (rf_wide_ok(h, rows)) { RfWideWs ww; return rf_wide_carve(h, rows, nullptr, 0, &ww); }
  float *a, *b, *c, *d, *e, *f, *g;
  bf16_t* y;
  unsigned* bar;
  char* sw;
  size_t swb;
  bf16_t *ya, *yb;
  float* pb;
  return rf_carve(h, rows, nullptr, 0, &a, &b, &c, &d, &e, &f, &g, &y, &bar, &sw, &swb, &ya, &yb, &pb);
}

extern "C" int mn_rf_sample(const mn_rf_head* h, const float* hidden, int64_t ld_hidden, int rows, int n_images,
                            const float* noise, float temperature, float text_cfg, float image_cfg,
                            float* latent_out, void* workspace, size_t workspace_bytes, void* stream) {
  MN_CHECK_ARG(h && hidden && noise && latent_out && workspace, "mn_rf_sample: null pointer");
  MN_CHECK_ARG(n_images >= 1 && rows >= n_images && rows % n_images == 0 && rows / n_images <= 3 && (rows <= 64 || rf_wide_ok(h, rows)),
               "mn_rf_sample: rows=%d n_images=%d (1..3 CFG rows per image; <= 64 rows, or <= 2048 with 64-aligned widths)", rows, n_images);
  MN_CHECK_ARG(h->target <= 256, "mn_rf_sample: target too large");
  MN_CHECK_ARG(h->wfmt == MN_W_BF16 || rf_fp8_ok(h), "mn_rf_sample: bad fp8 weight description (wfmt %d: row scales, widths %% 16)", h->wfmt);
  if (rf_wide_ok(h, rows))
    return rf_sample_wide(h, hidden, ld_hidden, rows, n_images, noise, temperature, text_cfg, image_cfg, latent_out, workspace,
                          workspace_bytes, stream);
  const int rpi = rows / n_images;
  float *z, *c, *ada, *hh, *hid, *v, *x;
  bf16_t* y;
  unsigned* bar;
  char* skws;
  size_t skws_bytes;
  bf16_t *ya, *yb;
  float *pbuf, *pada, *pbuf3;
  const size_t need = rf_carve(h, rows, workspace, workspace_bytes, &z, &c, &ada, &hh, &hid, &v, &x, &y, &bar, &skws, &skws_bytes,
                               &ya, &yb, &pbuf, &pada, &pbuf3);
  const bool chain = rf_chain_ok(h, rows), fused = g_rf_fuse && rf_fused_shape_ok(h, rows);
  t_sk_ws = skws; t_sk_ws_bytes = skws_bytes;
  if (need > workspace_bytes) { mn_set_error("mn_rf_sample: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  hipStream_t st = mn_stream(stream);
  const int w = h->w, A = h->depth * 3 * w + 2 * w, T = h->target;

  // z = vis_head Linear(hidden); c = cond_embed(LayerNorm(z))  (modeling_bailing_moe.py:1571-1574,1659; diff_loss:374)
  {
    mn_skinny_args a = sk(hidden, ld_hidden, h->vis_w, h->llm_hidden, h->vis_b, z, h->z_dim, rows, h->z_dim, h->llm_hidden);
    MN_TRY(mn_skinny_gemm(&a, stream));
    a = sk(z, h->z_dim, h->cond_w, h->z_dim, h->cond_b, c, w, rows, w, h->z_dim);
    a.prologue = MN_PRO_LN; a.ln_g = h->vis_ln_g; a.ln_b = h->vis_ln_b; a.eps = 1e-6f;
    MN_TRY(mn_skinny_gemm(&a, stream));
  }
  hipLaunchKernelGGL(rf_init_x_kernel, dim3(mn_cdiv(rows * T, 256)), dim3(256), 0, st, noise, temperature, x, rows, T, rpi);
  // The adaLN inputs SiLU(t_emb[s] + c) do not depend on the ODE state, so the modulations of ALL
  // Euler steps are one [2*steps*rows, w] x [w, depth*3w+2w] MFMA GEMM that reads the 0.7 GB of adaLN
  // weights once per token instead of once per step (activations split into bf16 hi+lo, both passes of the K loop
  // accumulate into one output, so that the products stay fp32-accurate).   Linear(SiLU(t_emb[s] + c))  (diff_loss:263-266,283-286,376)
  const int64_t SR = (int64_t)h->steps * rows;
  hipLaunchKernelGGL(rf_build_y_kernel, dim3(mn_cdiv(SR * w, 256)), dim3(256), 0, st, h->temb, c, y, h->steps, rows, w);
  if (rf_ada_w8(h, rows)) {      // <= 64 (step, row) pairs: stream the 0.36 GB of e4m3 adaLN bytes once, then slabs + bias -> ada
    const int nza = mn_stream_mfma_wq(y, h->ada_q, h->ada_scale, pada, (int)SR, A, w, h->wfmt, stream);
    if (nza < 0) return nza;
    hipLaunchKernelGGL(rf_glue_bias_out_kernel, dim3(mn_cdiv(SR * A, 256)), dim3(256), 0, st, pada, nza, (int)SR, A, h->ada_b, ada);
  } else if (rf_ada_stream(h, rows)) {
    const int nza = mn_stream_mfma(y, h->ada_w, pada, (int)SR, A, w, stream);
    if (nza < 0) return nza;
    hipLaunchKernelGGL(rf_glue_bias_out_kernel, dim3(mn_cdiv(SR * A, 256)), dim3(256), 0, st, pada, nza, (int)SR, A, h->ada_b, ada);
  } else {
    MN_TRY(mn_gemm_bf16_hilo(y, w, SR * w, h->ada_w, w, h->ada_b, ada, A, (int)SR, A, w, stream));
  }
  const float step = 1.0f / (float)h->steps;
  const float* ada_all = ada;
  const bool boundary = chain && g_rf_boundary && T <= 256 && (T % 8) == 0 && w <= 4096;
  int nz_fin = 0;
  unsigned n_persist = 0;                           // persistent launches so far (each takes its own range of barrier epochs)
  // <= 2 rows, one workgroup per CU: the WHOLE sampler as one persistent launch — Euler-step boundaries and the final layer are two
  // more phases per step instead of three launches (stream_kc.hip)
  if (chain && g_rf_kc && g_rf_persist && g_rf_whole && fused && rf_persist_ok(h->wfmt, rows, w, h->hidden, stream) &&
      rf_sampler_persist_ok(rows, w, h->hidden, h->depth, T, rpi, n_images)) {
    if (hipMemsetAsync(bar, 0, RF_PERSIST_BAR_WORDS * sizeof(unsigned), st) != hipSuccess) { mn_set_error("mn_rf_sample: hipMemsetAsync failed"); return MN_ELAUNCH; }
    const RfSamplerTail tail{h->steps, T, rpi, n_images, (int64_t)rows * A, h->in_w, h->in_b, h->fin_w, h->fin_b, noise, temperature, text_cfg, image_cfg, v, latent_out};
    MN_TRY(rf_blocks_persist(h->wfmt, hh, yb, rows, w, h->hidden, ada_all, (int64_t)A, h->depth, reinterpret_cast<const void* const*>(h->w12),
                             h->wfmt ? h->w12_scale : nullptr, h->b12, h->ln_g, h->ln_b, reinterpret_cast<const void* const*>(h->w3),
                             h->wfmt ? h->w3_scale : nullptr, h->b3, bar, 0u, &tail, stream));
    MN_CHECK_LAUNCH("mn_rf_sample");
    return MN_OK;
  }
  for (int s = 0; s < h->steps; ++s) {
    const float* ada = ada_all + (int64_t)s * rows * A;
    // h = input_proj(x)  (diff_loss:371)
    mn_skinny_args a;
    if (!boundary) {
      a = sk(x, T, h->in_w, T, h->in_b, hh, w, rows, w, T);
      MN_TRY(mn_skinny_gemm(&a, stream));
    }
    if (chain) {
      // rows >= 5: stream(w12) -> [reduce + SwiGLU + split] -> stream(w3) -> [reduce + gated residual + next LN-modulate + split]
      const int hid_n = h->hidden;
      if (boundary)      // previous step's velocity + CFG + Euler, input projection, block 0's modulated input: one launch
        hipLaunchKernelGGL(rf_step_boundary_kernel, dim3(rows), dim3(1024), 0, st, (const float*)pbuf, nz_fin, rows, T, h->fin_b, x, rpi,
                           text_cfg, image_cfg, step, s == 0 ? 1 : 0, h->in_w, h->in_b, w, hh, h->ln_g[0], h->ln_b[0], ada, ada + w,
                           (int64_t)A, ya);
      else
        hipLaunchKernelGGL(rf_glue_resid_ln_split_kernel, dim3(rows), dim3(1024), 0, st, (const float*)nullptr, 0, rows, w,
                           (const bf16_t*)nullptr, (const float*)nullptr, hh, h->ln_g[0], h->ln_b[0], ada, ada + w, (int64_t)A, ya);
      // <= 2 rows (the CFG rows of one image): K-complete launches (stream_kc.hip) — w12' normalises / modulates h itself and writes
      // w3's operand, w3' applies the gated residual in place: two launches per block, no slabs, no glue between the blocks
      const bool kc = g_rf_kc && fused && rf_kc_ok(h->wfmt, rows, w, hid_n);
      // ... and, one workgroup per CU, all blocks of the step as ONE persistent launch with grid barriers between the phases
      const bool persist = kc && g_rf_persist && rf_persist_ok(h->wfmt, rows, w, hid_n, stream);
      if (persist && s == 0) {
        if (hipMemsetAsync(bar, 0, RF_PERSIST_BAR_WORDS * sizeof(unsigned), st) != hipSuccess) { mn_set_error("mn_rf_sample: hipMemsetAsync failed"); return MN_ELAUNCH; }
      }
      for (int b = 0; persist && b < h->depth; b += RF_PERSIST_MAX_BLOCKS) {
        const int nb = h->depth - b < RF_PERSIST_MAX_BLOCKS ? h->depth - b : RF_PERSIST_MAX_BLOCKS;
        MN_TRY(rf_blocks_persist(h->wfmt, hh, yb, rows, w, hid_n, ada + (int64_t)b * 3 * w, (int64_t)A, nb,
                                 reinterpret_cast<const void* const*>(h->w12 + b), h->wfmt ? h->w12_scale + b : nullptr, h->b12 + b,
                                 h->ln_g + b, h->ln_b + b, reinterpret_cast<const void* const*>(h->w3 + b),
                                 h->wfmt ? h->w3_scale + b : nullptr, h->b3 + b, bar, 64u * n_persist++, nullptr, stream));
      }
      for (int b = 0; kc && !persist && b < h->depth; ++b) {
        const float* mod = ada + (int64_t)b * 3 * w;
        MN_TRY(rf_w12_kc(h->wfmt, hh, rows, w, hid_n, h->ln_g[b], h->ln_b[b], mod, mod + w, (int64_t)A, h->w12[b],
                         h->wfmt ? h->w12_scale[b] : nullptr, h->b12[b], yb, stream));
        MN_TRY(rf_w3_kc(h->wfmt, yb, rows, w, hid_n, h->w3[b], h->wfmt ? h->w3_scale[b] : nullptr, h->b3[b], mod + 2 * w, (int64_t)A, hh, stream));
      }
      if (kc) {     // the final layer's input: LayerNorm (no affine) of h, modulated  (diff_loss_rf_swiglu.py:288-292)
        const float* nmod = ada + (int64_t)h->depth * 3 * w;
        hipLaunchKernelGGL(rf_glue_resid_ln_split_kernel, dim3(rows), dim3(1024), 0, st, (const float*)nullptr, 0, rows, w,
                           (const bf16_t*)nullptr, (const float*)nullptr, hh, (const bf16_t*)nullptr, (const bf16_t*)nullptr, nmod, nmod + w,
                           (int64_t)A, ya);
      }
      for (int b = 0; !kc && b < h->depth; ++b) {
        const float* mod = ada + (int64_t)b * 3 * w;
        int nz = stream_dense(h->wfmt, ya, h->w12[b], h->wfmt ? h->w12_scale[b] : nullptr, pbuf, rows, 2 * hid_n, w, stream);
        if (nz < 0) return nz;
        float* p3 = pbuf;
        if (fused) {               // <= 4 rows: w3 builds SwiGLU(w12's slabs + bias) itself — no glue launch in between
          const StreamFuse f{pbuf, nz, h->b12[b]};
          p3 = pbuf3;
          nz = stream_fused(h->wfmt, h->w3[b], h->wfmt ? h->w3_scale[b] : nullptr, p3, rows, w, hid_n, f, stream);
        } else {
          hipLaunchKernelGGL(rf_glue_swiglu_split_kernel, dim3(mn_cdiv((int64_t)rows * hid_n, 1024)), dim3(256), 0, st, pbuf, nz,
                             rows, hid_n, h->b12[b], yb);
          nz = stream_dense(h->wfmt, yb, h->w3[b], h->wfmt ? h->w3_scale[b] : nullptr, pbuf, rows, w, hid_n, stream);
        }
        if (nz < 0) return nz;
        const bool last = b + 1 == h->depth;
        const float* nmod = last ? ada + (int64_t)h->depth * 3 * w : ada + (int64_t)(b + 1) * 3 * w;
        hipLaunchKernelGGL(rf_glue_resid_ln_split_kernel, dim3(rows), dim3(1024), 0, st, p3, nz, rows, w, h->b3[b],
                           mod + 2 * w, hh, last ? (const bf16_t*)nullptr : h->ln_g[b + 1],
                           last ? (const bf16_t*)nullptr : h->ln_b[b + 1], nmod, nmod + w, (int64_t)A, ya);
      }
      const int nz = mn_stream_mfma(ya, h->fin_w, pbuf, rows, T, w, stream);   // final_layer.linear on the modulated LN(h)
      if (nz < 0) return nz;
      nz_fin = nz;
      if (!boundary || s + 1 == h->steps) {      // (with the boundary launch the next step consumes the slabs itself)
        hipLaunchKernelGGL(rf_glue_bias_out_kernel, dim3(mn_cdiv(rows * T, 256)), dim3(256), 0, st, pbuf, nz, rows, T, h->fin_b, v);
        hipLaunchKernelGGL(rf_euler_kernel, dim3(n_images), dim3(256), 0, st, v, x, rpi, T, text_cfg, image_cfg, step);
      }
      continue;
    }
    for (int b = 0; b < h->depth; ++b) {
      const float* mod = ada + (int64_t)b * 3 * w;
      a = sk(hh, w, h->w12[b], w, h->b12[b], hid, h->hidden, rows, h->hidden, w);
      a.prologue = MN_PRO_LN_MOD; a.ln_g = h->ln_g[b]; a.ln_b = h->ln_b[b]; a.eps = 1e-6f;
      a.pro_a = mod; a.ld_pro_a = A; a.pro_b = mod + w; a.ld_pro_b = A;
      a.epilogue = MN_EPI_SWIGLU;
      MN_TRY(mn_skinny_gemm(&a, stream));
      a = sk(hid, h->hidden, h->w3[b], h->hidden, h->b3[b], hh, w, rows, w, h->hidden);
      a.epilogue = MN_EPI_RESID_GATE; a.res = hh; a.ldres = w; a.gate = mod + 2 * w; a.ldgate = A;
      MN_TRY(mn_skinny_gemm(&a, stream));
    }
    const float* modf = ada + (int64_t)h->depth * 3 * w;
    a = sk(hh, w, h->fin_w, w, h->fin_b, v, T, rows, T, w);
    a.prologue = MN_PRO_LN_MOD; a.eps = 1e-6f; a.pro_a = modf; a.ld_pro_a = A; a.pro_b = modf + w; a.ld_pro_b = A;
    MN_TRY(mn_skinny_gemm(&a, stream));
    hipLaunchKernelGGL(rf_euler_kernel, dim3(n_images), dim3(256), 0, st, v, x, rpi, T, text_cfg, image_cfg, step);
  }
  hipLaunchKernelGGL(rf_gather_latent_kernel, dim3(mn_cdiv(n_images * T, 256)), dim3(256), 0, st, x, latent_out, n_images, rpi, T);
  MN_CHECK_LAUNCH("mn_rf_sample");
  return MN_OK;
}

// ===========================================================================================
// Bailing-MoE decoder stack step
// ===========================================================================================
// ---- grouped expert path (rows >= 5): every distinct expert's weights are streamed once on the matrix cores ----
// Block 0 sorts the (row, slot) pairs by expert — off[g] .. off[g+1] are the sorted positions of group g, a pair's
// rank inside its group is its pair index order, so the layout is deterministic; blocks 1..M split row m-1 of the
// normalised activations into bf16 hi / lo halves for the MFMA kernel.
__global__ __launch_bounds__(256) void moe_group_split_kernel(const int32_t* __restrict__ topk_idx, int M, int n_slot, int G,
                                                              int32_t* __restrict__ off, int32_t* __restrict__ xrows,
                                                              int32_t* __restrict__ pair_pos, const float* __restrict__ xn,
                                                              int H, bf16_t* __restrict__ Y) {
  __shared__ int32_t s_idx[1024];
  __shared__ int32_t s_off[257];
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {
    const int m = blockIdx.x - 1;
    for (int k = tid; k < H; k += 256) {
      const float v = xn[(int64_t)m * H + k];
      const bf16_t hi = f32_to_bf16(v);
      Y[(int64_t)m * H + k] = hi;
      Y[(int64_t)(M + m) * H + k] = f32_to_bf16(v - bf16_to_f32(hi));
    }
    return;
  }
  const int P = M * n_slot;                         // <= 1024 pairs, G <= 256 groups (checked by the caller)
  for (int p = tid; p < P; p += 256) s_idx[p] = topk_idx[p];
  for (int g = tid; g <= G; g += 256) s_off[g] = 0;
  __syncthreads();
  for (int p = tid; p < P; p += 256) atomicAdd(&s_off[s_idx[p] + 1], 1);
  __syncthreads();
  if (tid == 0) for (int g = 0; g < G; ++g) s_off[g + 1] += s_off[g];
  __syncthreads();
  for (int g = tid; g <= G; g += 256) off[g] = s_off[g];
  for (int p = tid; p < P; p += 256) {
    const int e = s_idx[p];
    int rank = 0;
    for (int q = 0; q < p; ++q) rank += (s_idx[q] == e);
    const int pos = s_off[e] + rank;
    xrows[pos] = p / n_slot;
    pair_pos[p] = pos;
  }
}

// h[m, n] += sum_s w[m, s] * sum_z P[z][pos(m, s)][n]   (weighted sum of the routed + shared experts, residual add)
__global__ __launch_bounds__(256) void moe_combine_resid_kernel(const float* __restrict__ P, int nz, int64_t slab, int M, int H,
                                                                int n_slot, const int32_t* __restrict__ pair_pos,
                                                                const float* __restrict__ tw, float* __restrict__ h) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;    // 4 consecutive columns per thread (H % 4 == 0)
  if (i >= (int64_t)M * H) return;
  const int m = (int)(i / H), n = (int)(i % H);
  f4 acc = *reinterpret_cast<const f4*>(h + i);
  for (int s = 0; s < n_slot; s += 2) {           // two slots x nz slabs of independent 16-byte loads in flight
    const bool two = s + 1 < n_slot;
    const float* pa = P + (int64_t)pair_pos[m * n_slot + s] * H + n;
    const float* pb = P + (int64_t)pair_pos[m * n_slot + (two ? s + 1 : s)] * H + n;
    f4 ya = {0.f, 0.f, 0.f, 0.f}, yb = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nz; ++z) {
      ya += *reinterpret_cast<const f4*>(pa + z * slab);
      yb += *reinterpret_cast<const f4*>(pb + z * slab);
    }
    acc += tw[m * n_slot + s] * ya;
    if (two) acc += tw[m * n_slot + s + 1] * yb;
  }
  *reinterpret_cast<f4*>(h + i) = acc;
}

// ---- decoder-stack chain (rows >= 2): between two weight-streaming launches ONE kernel reduces the K-slice partials of
// the producer, applies the residual / expert combine, and builds the RMSNorm'ed bf16 hi/lo operand of the consumer.
// One workgroup per row (the RMS statistic needs the whole row), 4 columns per thread.
//   mode 0: h[m] = x[m / row_div]                                              (stack input)
//   mode 1: h[m] += sum_z P[z][m][:]                                           (attention output projection)
//   mode 2: h[m] += sum_s tw[m,s] * sum_z P[z][pos(m,s)][:]                    (routed + shared experts)
//   mode 3: h[m] as it is                                                      (experts already accumulated into h)
// then, with norm_w: xn = RMSNorm(h[m]) -> xn_out fp32 (optional) and Y bf16 hi rows / lo rows (optional).
__global__ __launch_bounds__(1024) void llm_glue_kernel(int mode, const float* __restrict__ x, int64_t ldx, int row_div,
                                                        const float* __restrict__ P, int nz, int64_t slab,
                                                        const int32_t* __restrict__ pair_pos, const float* __restrict__ tw,
                                                        int n_slot, float* __restrict__ h, int M, int H,
                                                        const bf16_t* __restrict__ norm_w, float eps,
                                                        float* __restrict__ xn_out, bf16_t* __restrict__ Y) {
  __shared__ float red[32];
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int m = blockIdx.x, col = threadIdx.x * 4;
  const bool act = col < H;
  f4 v = {0.f, 0.f, 0.f, 0.f};
  u2 nw2 = {0u, 0u};                               // the norm weights of this thread's columns, requested with the row (not behind the barrier)
  if (act && norm_w) nw2 = *reinterpret_cast<const u2*>(norm_w + col);
  if (act) {
    if (mode == 0) {
      v = *reinterpret_cast<const f4*>(x + (int64_t)(m / row_div) * ldx + col);
    } else {
      v = *reinterpret_cast<const f4*>(h + (int64_t)m * H + col);
      if (mode == 1) {
        const float* pp = P + (int64_t)m * H + col;
        for (int z0 = 0; z0 < nz; z0 += 8) {        // eight slabs' loads in flight (a one-by-one loop is nz dependent round trips)
          f4 t[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = z0 + j < nz ? *reinterpret_cast<const f4*>(pp + (int64_t)(z0 + j) * slab) : f4{0.f, 0.f, 0.f, 0.f};
          for (int j = 0; j < 8; ++j) v += t[j];                      // (the one-by-one loop's order: same bits)
        }
      } else if (mode == 2) {
        for (int s = 0; s < n_slot; s += 4) {       // four slots x nz slabs of independent 16-byte loads in flight
          const float* pp[4];
          float wv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int sj = s + j < n_slot ? s + j : s;
            pp[j] = P + (int64_t)pair_pos[m * n_slot + sj] * H + col;
            wv[j] = s + j < n_slot ? tw[m * n_slot + sj] : 0.f;
          }
          f4 y[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          for (int z = 0; z < nz; ++z) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] += *reinterpret_cast<const f4*>(pp[j] + z * slab);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) v += wv[j] * y[j];
        }
      }
    }
    if (mode != 3) *reinterpret_cast<f4*>(h + (int64_t)m * H + col) = v;
  }
  if (!norm_w) return;
  const float ss = block_sum(act ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : 0.f, red);
  const float rstd = rsqrtf(ss / (float)H + eps);
  if (act) {
    float o[4] = {v.x, v.y, v.z, v.w};
    const float nwv[4] = {bf16lo_to_f32(nw2.x), bf16hi_to_f32(nw2.x), bf16lo_to_f32(nw2.y), bf16hi_to_f32(nw2.y)};
    bf16_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = o[j] * rstd * nwv[j];
      hi[j] = f32_to_bf16(o[j]);
      lo[j] = f32_to_bf16(o[j] - bf16_to_f32(hi[j]));
    }
    if (xn_out) *reinterpret_cast<f4*>(xn_out + (int64_t)m * H + col) = f4{o[0], o[1], o[2], o[3]};
    if (Y) {
      *reinterpret_cast<u2*>(Y + (int64_t)m * H + col) = u2{(uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16)};
      *reinterpret_cast<u2*>(Y + (int64_t)(M + m) * H + col) = u2{(uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16)};
    }
  }
}

// Router tail in ONE workgroup: reduce the gate GEMV's K-slice partials, softmax + top-k + renormalise per row (one wave
// per row, 64 experts = 64 lanes; BailingMoeGate.forward :505-520), append the shared pseudo-experts, and — when off is
// given — sort the (row, slot) pairs by expert for the grouped expert GEMMs (same layout as moe_group_split_kernel).
__global__ __launch_bounds__(1024) void moe_route_group_kernel(const float* __restrict__ P, int nz, int M, int E, int top_k,
                                                               int norm_topk_prob, int n_shared, float* __restrict__ topk_w,
                                                               int32_t* __restrict__ topk_idx, int G, int32_t* __restrict__ off,
                                                               int32_t* __restrict__ xrows, int32_t* __restrict__ pair_pos) {
  __shared__ int32_t s_idx[1024];
  __shared__ int32_t s_off[257];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_slot = top_k + n_shared;
  for (int m = wave; m < M; m += 16) {
    float s = -INFINITY;
    if (lane < E) {
      s = 0.f;
      for (int z0 = 0; z0 < nz; z0 += 8) {          // eight slabs' loads in flight
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = z0 + j < nz ? P[((int64_t)(z0 + j) * M + m) * E + lane] : 0.f;
        for (int j = 0; j < 8; ++j) s += t[j];                    // (the one-by-one loop's order: same bits)
      }
    }
    const float mx = wave_max(s);
    float p = lane < E ? __expf(s - mx) : 0.f;
    p = p / wave_sum(p);
    float cur = lane < E ? p : -1.f, wsum = 0.f, myw = 0.f;
    int myidx = 0;
    for (int k = 0; k < top_k; ++k) {             // iterative arg-max; ties -> lowest expert index
      const float best = wave_max(cur);
      const int sel = __ffsll((long long)__ballot(cur == best)) - 1;
      if (lane == k) { myw = best; myidx = sel; }
      if (lane == sel) cur = -1.f;
      wsum += best;
    }
    if (lane < n_slot) {
      const int idx = lane < top_k ? myidx : E + (lane - top_k);
      const float wv = lane < top_k ? ((norm_topk_prob && top_k > 1) ? myw / wsum : myw) : 1.0f;
      topk_idx[m * n_slot + lane] = idx;
      topk_w[m * n_slot + lane] = wv;
      s_idx[m * n_slot + lane] = idx;
    }
  }
  if (!off) return;
  const int Pn = M * n_slot;
  for (int g = tid; g <= G; g += 1024) s_off[g] = 0;
  __syncthreads();
  for (int p = tid; p < Pn; p += 1024) atomicAdd(&s_off[s_idx[p] + 1], 1);
  __syncthreads();
  if (tid == 0) for (int g = 0; g < G; ++g) s_off[g + 1] += s_off[g];
  __syncthreads();
  for (int g = tid; g <= G; g += 1024) off[g] = s_off[g];
  for (int p = tid; p < Pn; p += 1024) {
    const int e = s_idx[p];
    int rank = 0;
    for (int q = 0; q < p; ++q) rank += (s_idx[q] == e);
    const int pos = s_off[e] + rank;
    xrows[pos] = p / n_slot;
    pair_pos[p] = pos;
  }
}

// Diagnostic (tests: teacher-forced routing parity, mn_llm_route_capture): when set, every layer's routing [M][n_slot] int32 — the top-k
// expert ids, then the shared pseudo-experts — is copied to capture + l * M * n_slot right after the layer's router has run.
static int32_t* g_route_capture = nullptr;
static inline void route_capture(int l, const int32_t* ti, int M, int n_slot, hipStream_t st) {
  if (g_route_capture)
    (void)hipMemcpyAsync(g_route_capture + (size_t)l * M * n_slot, ti, (size_t)M * n_slot * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
}
extern "C" int mn_llm_route_capture(int32_t* capture) { g_route_capture = capture; return MN_OK; }

#include "wide_llm.inl"
#include "tp.inl"

struct MoeWs {
  int32_t *off, *xrows, *pair_pos;
  bf16_t *y1, *y2;
  float *p1, *p2;
};

static size_t moe_carve(Carver& cv, int rows, int H, int I, int G, int n_slot, MoeWs* o) {
  const size_t P = (size_t)rows * n_slot;
  const size_t before = cv.off;
  o->off = cv.take<int32_t>((size_t)G + 1);
  o->xrows = cv.take<int32_t>(P);
  o->pair_pos = cv.take<int32_t>(P);
  o->y1 = cv.take<bf16_t>((size_t)2 * rows * H);
  o->y2 = cv.take<bf16_t>(2 * P * I);
  o->p1 = cv.take<float>((size_t)mn_stream_mfma_grouped_slices(G, rows, 2 * I, H) * P * 2 * I);
  o->p2 = cv.take<float>((size_t)mn_stream_mfma_grouped_slices(G, rows, H, I) * P * H);
  return cv.off - before;
}

// grouped-expert route from 3 rows (bf16) / 6 rows (e4m3).  Round 6 re-measured the switch after the pair route got the one-launch router, the
// wave-segmented down projection, the RMSNorm prologue of the QKV launch and a launch plan that keeps the 24 / 32 / 40 pairs' workgroups in
// ONE round over the CUs (skinny_gemm.hip: the cap was rounded up — 264 workgroups for 24 pairs).  On RANDOM rows the pairs win up to 4
// (bf16) / 5 (e4m3) rows — decoder step, 28 layers, grouped vs pairs: bf16 3 rows 3.05 vs 2.73 ms, 4 rows 3.42 vs 3.19, 5 rows 3.64 vs 3.96;
// e4m3 3 rows 2.58 vs 1.92, 4 rows 2.79 vs 2.13, 5 rows 3.00 vs 2.58, 6 rows 3.17 vs 4.19 (profiles/r06_moe_min_rows_ab.txt).  But the CFG
// rows of ONE image select overlapping experts (10.0 distinct routed experts of 18 pairs per layer at 3 rows, 9.3 of 12 at 2:
// tools/exp/cfg_row_expert_overlap.py), which the grouped route reads once: in REAL generation (tools/exp/edit_shape_ab.py,
// profiles/r06_edit_shape_ab.txt) bf16 keeps the grouped route at 3 rows (98.3 vs 97.3 tokens/s; 4 rows: a tie), e4m3 takes the pairs
// (3 rows 110.8 -> 113.0 and more, 2 images x 2 rows 196.8 -> 205.5).
static int g_moe_down = 1;            // the 1- / 2-row down projection on moe_down.hip (dev-library A/B switch: mn_moe_tune_down)
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_moe_tune_down(int on) { g_moe_down = on; }
#endif
static int g_moe_min_rows_bf16 = 3, g_moe_min_rows_fp8 = 6;
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_moe_tune_min_rows(int rows) {   // A/B hook: 0 = the shipped thresholds
  g_moe_min_rows_bf16 = rows > 0 ? rows : 3;
  g_moe_min_rows_fp8 = rows > 0 ? rows : 6;
}
#endif
// NF4 / int8 experts run the grouped streaming launch from ONE row on: their products are rounded to bf16 per element (bitsandbytes' /
// quanto's de-quantisation), which the fp32-FMA pair kernels of 1- / 2-row steps do not do
static inline int moe_mfma_min_rows(const mn_llm* m) {
  return (m->wfmt == MN_W_NF4 || m->wfmt == MN_W_INT8) ? 1 : (m->wfmt == MN_W_FP8_E4M3 ? g_moe_min_rows_fp8 : g_moe_min_rows_bf16);
}
static bool moe_mfma_ok(const mn_llm* m, int rows) {
  return rows >= moe_mfma_min_rows(m) && rows * (m->top_k + m->n_shared_slots) <= 1024 &&
         m->n_experts + m->n_shared_slots <= 256 && (m->hidden % 8) == 0 && (m->moe_inter % 8) == 0;   // % 4: vector glue
}

// 2..32 rows run the decoder stack as a chain of weight-streaming launches and fused glue kernels (4 columns per thread,
// one wave per router row: 64 experts at most).  Measured end to end against the unfused sequence (same box, tokens/s):
// 2 rows 78.5 vs 76.5, 8 rows 254 vs 247, 16 rows 451 vs 442, 32 rows 699 vs 698; at 64 rows the one-workgroup-per-row
// glue kernels lose to the wider unfused ones (1084 vs 1122), so the chain stops at 32.
static int g_chain_max_rows = 32, g_chain_router = 1, g_chain_fuse_ln1 = 1;
static int g_moe_gate_up = 1, g_moe_gate_up_rows = 1, g_moe_gate_up_chain_rows = 2, g_moe_gate_up_chain_all = 0;      // (chain: int8 / NF4 by default — bf16 / e4m3 keep the pair launches: 2.25 vs 2.64 ms, 1.71 vs 1.84)      // the one-launch router + gate/up of 1-row steps (dev-library A/B: mn_moe_tune_gate_up)
bool moe_router_rows_ok(int M, int H, int E);
int moe_router_rows(float* h, const float* P, int nz, const uint16_t* norm_w, float eps, const uint16_t* gate_w, int M, int H, int E, int top_k,
                    int norm_topk_prob, int n_shared_slots, float* x_norm, int32_t* topk_idx, float* topk_w, float* logits_ws, void* stream);
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_moe_tune_gate_up(int on, int max_rows) { g_moe_gate_up = on; g_moe_gate_up_rows = max_rows & 0xff; g_moe_gate_up_chain_rows = (max_rows >> 8) & 0xff; g_moe_gate_up_chain_all = (max_rows >> 16) & 1; }
extern "C" MN_DEV_API void mn_llm_tune_chain(int max_rows) { g_chain_max_rows = max_rows & 0xffff; g_chain_router = (max_rows >> 16) & 1 ? 0 : 1; g_chain_fuse_ln1 = (max_rows >> 17) & 1 ? 0 : 1; }   // A/B hook (tools/): bit 16 = the one-launch router of the chain OFF, bit 17 = RMSNorm(ln1) as its own glue launch
#endif
static bool llm_chain_ok(const mn_llm* m, int rows) {
  return rows >= 2 && rows <= g_chain_max_rows && (m->hidden % 8) == 0 && m->hidden <= 4096 && ((m->n_q * m->head_dim) % 8) == 0 &&
         m->n_experts <= 64 && rows * (m->top_k + m->n_shared_slots) <= 1024 && m->n_experts + m->n_shared_slots <= 256;
}

struct LlmWs {
  float *h, *qkv, *q, *attn, *xn, *tw, *hmid, *logits;
  int32_t* ti;
  MoeWs moe;
  bf16_t *yh, *ya;      // chain: bf16 hi/lo operands of width H and n_q * head_dim
  float* pp;            // chain: K-slice partials of the QKV / dense / gate launches
  void* attn_ws;
  size_t attn_ws_bytes;
  char* sk_ws;
  size_t sk_ws_bytes;
};

static size_t llm_carve(const mn_llm* m, int rows, int64_t t_max, void* ws, size_t cap, LlmWs* o) {
  Carver cv(ws, cap, ws == nullptr);
  const int n_slot = m->top_k + m->n_shared_slots;
  const int qkv_dim = (m->n_q + 2 * m->n_kv) * m->head_dim;
  o->h = cv.take<float>((size_t)rows * m->hidden);
  o->qkv = cv.take<float>((size_t)rows * qkv_dim);
  o->q = cv.take<float>((size_t)rows * m->n_q * m->head_dim);
  o->attn = cv.take<float>((size_t)rows * m->n_q * m->head_dim);
  o->xn = cv.take<float>((size_t)rows * m->hidden);
  o->tw = cv.take<float>((size_t)rows * n_slot);
  o->ti = cv.take<int32_t>((size_t)rows * n_slot);
  o->hmid = cv.take<float>((size_t)rows * n_slot * m->moe_inter);
  o->logits = cv.take<float>((size_t)2 * rows * m->n_experts);
  o->attn_ws_bytes = mn_attn_decode_workspace_bytes(rows, m->n_q, m->head_dim, t_max);
  o->attn_ws = cv.take<char>(o->attn_ws_bytes);
  o->sk_ws_bytes = sk_ws_need(rows, {{qkv_dim, m->hidden, 0}, {m->hidden, m->n_q * m->head_dim, 0}, {m->n_experts, m->hidden, 0}});
  o->sk_ws = cv.take<char>(o->sk_ws_bytes);
  if (moe_mfma_ok(m, rows)) moe_carve(cv, rows, m->hidden, m->moe_inter, m->n_experts + m->n_shared_slots, n_slot, &o->moe);
  if (llm_chain_ok(m, rows)) {
    const int ad = m->n_q * m->head_dim;
    o->yh = cv.take<bf16_t>((size_t)2 * rows * m->hidden);
    o->ya = cv.take<bf16_t>((size_t)2 * rows * ad);
    size_t pmax = (size_t)mn_stream_mfma_slices(rows, qkv_dim, m->hidden) * qkv_dim;
    const size_t p2 = (size_t)mn_stream_mfma_slices(rows, m->hidden, ad) * m->hidden;
    const size_t p3 = (size_t)mn_stream_mfma_slices(rows, m->n_experts, m->hidden) * m->n_experts;
    if (p2 > pmax) pmax = p2;
    if (p3 > pmax) pmax = p3;
    o->pp = cv.take<float>(pmax * rows);
  }
  return cv.off;
}

// Y (bf16 hi rows, lo rows y_lo_off elements further; y_lo_off = 0: plain bf16) = act(norm(x)): the operand of a hi/lo gemm256
// launch from an fp32 row block.  norm 0: none, 1: RMSNorm(g), 2: LayerNorm(g?, b?); act 1: exact-erf GELU.
extern "C" int mn_norm_act_split(const float* x, int64_t ldx, int norm, const uint16_t* g, const uint16_t* b, float eps, int act,
                                 uint16_t* Y, int64_t ldy, int64_t y_lo_off, float* out, int64_t ldo, int M, int D, void* stream) {
  MN_CHECK_ARG(x && (Y || out) && M >= 1 && wide_glue_ok(D) && (ldx % 4) == 0 && (ldy % 4) == 0 && (y_lo_off % 4) == 0 && (ldo % 4) == 0 &&
                   norm >= 0 && norm <= 2 && (norm != 1 || g) && (act == 0 || act == 1) && (((uintptr_t)x | (uintptr_t)out) & 15) == 0 &&
                   (((uintptr_t)Y & 7) == 0),
               "mn_norm_act_split: bad args (D %% 4 == 0, D <= 4096, 16-byte rows)");
  WideGlue gl;
  memset(&gl, 0, sizeof(gl));
  gl.h = x; gl.ldh = ldx; gl.norm = norm; gl.ng = g; gl.nb = b; gl.eps = eps; gl.act = act;
  gl.Y = Y; gl.ldy = ldy; gl.y_lo_off = y_lo_off; gl.out = out; gl.ldo = ldo; gl.M = M; gl.D = D;
  wide_glue(gl, mn_stream(stream));
  MN_CHECK_LAUNCH("mn_norm_act_split");
  return MN_OK;
}

// Tail of a split-K nn.Linear whose result joins the fp32 residual stream (MingTok Block.forward, layers/block.py:80-105: x = x +
// proj(attn) / x = x + mlp(...)), fused with the LayerNorm of the NEXT consumer: h[m] += sum_z P[z][m][:] (the bias rides slab 0);
// if y: y[m] = bf16(LayerNorm(h[m]) (ln_g, ln_b optional), GELU after it when gelu != 0).  One launch instead of a reduce pass
// and a LayerNorm pass; used when the row count leaves a 256 x 256-tile GEMM without split-K on a fraction of the chip.
extern "C" int mn_slab_resid_norm(const float* P, int nz, int64_t slab, float* h, int64_t ldh, const uint16_t* ln_g, const uint16_t* ln_b,
                                  float eps, int gelu, uint16_t* y, int64_t ldy, int M, int D, void* stream) {
  MN_CHECK_ARG(P && h && nz >= 1 && M >= 1 && wide_glue_ok(D) && (ldh % 4) == 0 && (!y || (ldy % 4) == 0), "mn_slab_resid_norm: bad args");
  WideGlue g;
  memset(&g, 0, sizeof(g));
  g.h = h; g.ldh = ldh; g.P = P; g.nz = nz; g.slab = slab; g.h_out = h; g.ldho = ldh;
  if (y) { g.norm = 2; g.ng = ln_g; g.nb = ln_b; g.eps = eps; g.act = gelu ? 1 : 0; g.Y = y; g.ldy = ldy; g.y_lo_off = 0; }
  g.M = M; g.D = D;
  wide_glue(g, mn_stream(stream));
  MN_CHECK_LAUNCH("mn_slab_resid_norm");
  return MN_OK;
}

// Weighted un-permute of the expert outputs + residual (moe_infer, modeling_bailing_moe.py:630-639) fused with the RMSNorm of the
// next consumer (:1186 of the following layer): h[t] += sum_s tw[t, s] * yg[slot_of[t, s]];  if y: y[t] = bf16(RMSNorm(h[t]) * norm_w).
extern "C" int mn_moe_combine_norm(const float* yg, const int32_t* slot_of, const float* tw, int n_slot, float* h, int64_t ldh,
                                   const uint16_t* norm_w, float eps, uint16_t* y, int64_t ldy, int T, int H, void* stream) {
  MN_CHECK_ARG(yg && slot_of && tw && h && n_slot >= 1 && T >= 1 && wide_glue_ok(H) && (ldh % 4) == 0 && (!y || (norm_w && (ldy % 4) == 0)),
               "mn_moe_combine_norm: bad args");
  WideGlue g;
  memset(&g, 0, sizeof(g));
  g.h = h; g.ldh = ldh; g.cy = yg; g.cpos = slot_of; g.cw = tw; g.n_slot = n_slot; g.h_out = h; g.ldho = ldh;
  if (y) { g.norm = 1; g.ng = norm_w; g.eps = eps; g.Y = y; g.ldy = ldy; g.y_lo_off = 0; }
  g.M = T; g.D = H;
  wide_glue(g, mn_stream(stream));
  MN_CHECK_LAUNCH("mn_moe_combine_norm");
  return MN_OK;
}

extern "C" int mn_llm_max_rows(const mn_llm* m) { return llm_wide_ok(m, 2048) ? 2048 : 64; }
// Largest row count one call of each composite accepts for this configuration: 2048 when the wide route applies
// (64-aligned widths), else 64.
extern "C" int mn_rf_max_rows(const mn_rf_head* h) { return rf_wide_ok(h, 2048) ? 2048 : 64; }
extern "C" int mn_semdec_max_rows(const mn_semdec* s) { return sem_wide_ok(s, 2048) ? 2048 : 64; }

extern "C" size_t mn_llm_workspace_bytes(const mn_llm* m, int rows, int64_t t_max) {
  size_t wide = 0;
  if (llm_wide_ok(m, rows)) {
    LlmWideWs ww;
    wide = llm_wide_carve(m, rows, t_max, nullptr, 0, &ww);
    if (rows > 64) return wide;
  }
  LlmWs w{};                                 // <= 64 rows: a step with the image-gate override stays on the streaming route
  const size_t narrow = llm_carve(m, rows, t_max, nullptr, 0, &w);
  return narrow > wide ? narrow : wide;
}

extern "C" int mn_rows_advance(int32_t* a, int32_t* b, int32_t* c, int M, int delta, void* stream) {
  MN_CHECK_ARG(M >= 1, "mn_rows_advance: M=%d", M);
  hipLaunchKernelGGL(rows_advance_kernel, dim3(mn_cdiv(M, 256)), dim3(256), 0, mn_stream(stream), a, b, c, M, delta);
  MN_CHECK_LAUNCH("mn_rows_advance");
  return MN_OK;
}

static int llm_step_impl(const mn_llm* m, const float* x, int64_t ldx, int x_row_div, int M, const uint8_t* image_mask, const int32_t* row_seq,
                         const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len,
                         const uint8_t* key_mask, int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max,
                         float* hidden_out, void* workspace, size_t workspace_bytes, int flags, void* stream,
                         const int32_t* span_tab = nullptr, int n_spans = 0, int span_max_len = 0);

extern "C" int mn_llm_step(const mn_llm* m, const float* x, int64_t ldx, int x_row_div, int M, const uint8_t* image_mask, const int32_t* row_seq,
                           const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len,
                           const uint8_t* key_mask, int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max,
                           float* hidden_out, void* workspace, size_t workspace_bytes, void* stream) {
  return llm_step_impl(m, x, ldx, x_row_div, M, image_mask, row_seq, row_slot, row_pos, row_len, key_mask, ld_mask, kv_cache, n_seq, t_max,
                       hidden_out, workspace, workspace_bytes, 0, stream);
}

extern "C" int mn_llm_step_ex(const mn_llm* m, const float* x, int64_t ldx, int x_row_div, int M, const uint8_t* image_mask, const int32_t* row_seq,
                              const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len,
                              const uint8_t* key_mask, int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max,
                              float* hidden_out, void* workspace, size_t workspace_bytes, int flags, void* stream) {
  MN_CHECK_ARG((flags & ~MN_STEP_DISTINCT_SEQUENCES) == 0, "mn_llm_step_ex: unknown flags 0x%x", flags);
  return llm_step_impl(m, x, ldx, x_row_div, M, image_mask, row_seq, row_slot, row_pos, row_len, key_mask, ld_mask, kv_cache, n_seq, t_max,
                       hidden_out, workspace, workspace_bytes, flags, stream);
}

// A prefill chunk whose rows are whole spans of cache sequences (mingnative.h): the attention of the wide route runs on the tiled
// hi/lo flash kernel instead of the per-row decode kernels.  Other shapes take mn_llm_step's path (same results).
extern "C" int mn_llm_step_spans(const mn_llm* m, const float* x, int64_t ldx, int M, const uint8_t* image_mask, const int32_t* row_seq,
                                 const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len, float* kv_cache, int n_seq,
                                 int64_t t_max, const int32_t* span_tab, int n_spans, int max_len, float* hidden_out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  MN_CHECK_ARG(span_tab && n_spans >= 1 && max_len >= 1 && max_len <= M, "mn_llm_step_spans: bad span table (n_spans %d, max_len %d, M %d)", n_spans,
               max_len, M);
  return llm_step_impl(m, x, ldx, 1, M, image_mask, row_seq, row_slot, row_pos, row_len, nullptr, 0, kv_cache, n_seq, t_max, hidden_out, workspace,
                       workspace_bytes, 0, stream, span_tab, n_spans, max_len);
}

static int llm_step_impl(const mn_llm* m, const float* x, int64_t ldx, int x_row_div, int M, const uint8_t* image_mask, const int32_t* row_seq,
                         const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len,
                         const uint8_t* key_mask, int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max,
                         float* hidden_out, void* workspace, size_t workspace_bytes, int flags, void* stream,
                         const int32_t* span_tab, int n_spans, int span_max_len) {
  MN_CHECK_ARG(m && x && row_seq && row_slot && row_pos && row_len && kv_cache && hidden_out && workspace,
               "mn_llm_step: null pointer");
  MN_CHECK_ARG(M >= 1 && (M <= 64 || llm_wide_ok(m, M)) && x_row_div >= 1,
               "mn_llm_step: M=%d (1..64, or up to 2048 rows with 64-aligned widths)", M);
  // fp8 experts: 1 or 2 rows run the one-row fp8 kernel on the (row, expert) pairs, more the grouped streaming kernels
  MN_CHECK_ARG(m->wfmt == MN_W_BF16 || (mn_w8(m->wfmt) && m->w_gate_up_scale && m->w_down_scale && (m->hidden % mn_wq_kmult(m->wfmt)) == 0 &&
                                        (m->moe_inter % mn_wq_kmult(m->wfmt)) == 0 && (llm_wide_ok(m, M) || (M <= 64 && (M < moe_mfma_min_rows(m) || moe_mfma_ok(m, M))))),
               "mn_llm_step: quantised experts need scale tables, widths %% 16 == 0 (NF4: %% 64) and <= 64 rows or the wide route's shapes (M = %d)", M);
  if (llm_wide_ok(m, M))
    return llm_step_wide(m, x, ldx, x_row_div, M, image_mask, row_seq, row_slot, row_pos, row_len, key_mask, ld_mask, kv_cache, n_seq, t_max,
                         hidden_out, workspace, workspace_bytes, stream, span_tab, n_spans, span_max_len);
  LlmWs w{};
  const size_t need = llm_carve(m, M, t_max, workspace, workspace_bytes, &w);
  if (need > workspace_bytes) { mn_set_error("mn_llm_step: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  hipStream_t st = mn_stream(stream);
  const int H = m->hidden, hd = m->head_dim, nq = m->n_q, nkv = m->n_kv, I = m->moe_inter;
  const int qkv_dim = (nq + 2 * nkv) * hd, n_slot = m->top_k + m->n_shared_slots;
  const int64_t layer_kv = (int64_t)n_seq * 2 * nkv * t_max * hd;
  const float q_scale = 1.0f / sqrtf((float)hd);
  t_sk_ws = w.sk_ws; t_sk_ws_bytes = w.sk_ws_bytes;
  // rows of DISTINCT cache sequences (decode steps; not a prefill chunk, whose rows read each other's new K / V lines): RoPE + KV
  // append ride the attention launch
  const bool fuse_attn = (flags & MN_STEP_DISTINCT_SEQUENCES) && mn_attn_fused_ok(M, nq, nkv, hd, t_max);
  if (llm_chain_ok(m, M) && !(image_mask && m->image_gate)) {
    // ---- chain path: 12 launches per layer instead of 18 (2 rows: 6 — QKV with the RMSNorm prologue, attention with RoPE + KV append, dense,
    // the one-launch router, the expert pairs' gate/up, the down projection into h).  glue = llm_glue_kernel.
    const int E = m->n_experts, S = m->n_shared_slots, G = E + S, ad = nq * hd, P = M * n_slot;
    // 2 rows (the CFG rows of one image): the experts run as router + gate/up in ONE launch and the wave-segmented down projection,
    // straight into h (moe_gate_up.hip, moe_down.hip) — int8 / NF4 (2.26 -> 2.05 ms, 2.48 -> 2.38 per step against their grouped streaming launches;
    // bf16 / e4m3 keep the pair launches, which are faster there)
    const bool gu_chain = g_moe_gate_up && M <= g_moe_gate_up_chain_rows && g_moe_down && moe_gate_up_ok(m->wfmt, H, I, E, m->top_k, S) &&
                          moe_down_ok(m->wfmt, n_slot, H, I) && (m->wfmt == MN_W_INT8 || m->wfmt == MN_W_NF4 || g_moe_gate_up_chain_all);
    const bool grouped = moe_mfma_ok(m, M) && !gu_chain;
    const int gt = ((H / 4 + 63) / 64) * 64;                           // threads of a glue workgroup
    int nz2 = 0;                                                       // slabs of the previous layer's expert down-projection
    for (int l = 0; l < m->n_layers; ++l) {
      float* kv_l = kv_cache + (int64_t)l * layer_kv;
      bool ln1_fused = false;
      // glue: (previous experts' combine + residual | stack input) -> RMSNorm(ln1) -> yh
      if (l == 0)
        hipLaunchKernelGGL(llm_glue_kernel, dim3(M), dim3(gt), 0, st, 0, x, ldx, x_row_div, (const float*)nullptr, 0, (int64_t)0,
                           (const int32_t*)nullptr, (const float*)nullptr, n_slot, w.h, M, H, m->ln1[l], m->rms_eps,
                           (float*)nullptr, w.yh);
      else if (!(ln1_fused = !grouped && g_chain_fuse_ln1 && stream_rmsnorm_ok(M, qkv_dim, H)))
        hipLaunchKernelGGL(llm_glue_kernel, dim3(M), dim3(gt), 0, st, grouped ? 2 : 3, (const float*)nullptr, (int64_t)0, 1,
                           (const float*)w.moe.p2, nz2, (int64_t)P * H, (const int32_t*)w.moe.pair_pos, (const float*)w.tw,
                           n_slot, w.h, M, H, m->ln1[l], m->rms_eps, (float*)nullptr, w.yh);
      // QKV launch -> partials; RoPE + KV append reduce them  (:743-789).  With the experts already summed into h (2-4 rows) the
      // launch normalises h itself (stream_mfma.hip FUSE_RMSNORM: the glue launch's arithmetic in every workgroup's staging pass)
      int nz = ln1_fused ? stream_rmsnorm(m->wqkv[l], w.pp, M, qkv_dim, H, w.h, m->ln1[l], m->rms_eps, stream)
                         : mn_stream_mfma(w.yh, m->wqkv[l], w.pp, M, qkv_dim, H, stream);
      if (nz < 0) return nz;
      if (fuse_attn) {
        MN_TRY(mn_attn_decode_fused(w.pp, qkv_dim, nz, (int64_t)M * qkv_dim, M, nq, nkv, hd, 1, m->cos_tab, m->sin_tab, row_seq, row_slot,
                                    row_pos, m->mrope_sec_t, m->mrope_sec_h, q_scale, kv_l, t_max, row_len, key_mask, ld_mask, nullptr,
                                    w.ya, w.attn_ws, w.attn_ws_bytes, stream));
      } else {
        MN_TRY(mn_rope_kv_from_partials(w.pp, qkv_dim, nz, (int64_t)M * qkv_dim, M, nq, nkv, hd, 1, m->cos_tab, m->sin_tab,
                                        row_seq, row_slot, row_pos, m->mrope_sec_t, m->mrope_sec_h, q_scale, w.q, kv_l, t_max,
                                        stream));
        // masked GQA; the combine writes the dense projection's bf16 operand directly  (:791-812)
        MN_TRY(mn_attn_decode_split(w.q, M, nq, nkv, hd, kv_l, t_max, row_seq, row_len, key_mask, ld_mask, nullptr, w.ya, w.attn_ws,
                                    w.attn_ws_bytes, stream));
      }
      nz = mn_stream_mfma(w.ya, m->wdense[l], w.pp, M, H, ad, stream);
      if (nz < 0) return nz;
      // residual + RMSNorm(ln2) + router + the selected experts' gate/up in ONE launch (every workgroup sums the slabs and routes its row
      // itself), then the down projection
      if (gu_chain) {
        MN_TRY(moe_gate_up_routed(m->wfmt, w.h, H, m->ln2[l], m->rms_eps, m->gate[l], m->w_gate_up[l], (int64_t)2 * I * H,
                                  m->wfmt ? m->w_gate_up_scale[l] : nullptr, (int64_t)2 * I * mn_wq_scales_per_row(m->wfmt, H), M, H, I, E, m->top_k,
                                  S, m->norm_topk_prob, w.hmid, (int64_t)n_slot * I, w.ti, w.tw, w.logits, w.pp, nz, (int64_t)M * H, stream));
        route_capture(l, w.ti, M, n_slot, st);
        MN_TRY(moe_down_rows(m->wfmt, w.hmid, (int64_t)n_slot * I, m->w_down[l], (int64_t)H * I, m->wfmt ? m->w_down_scale[l] : nullptr,
                             (int64_t)H * mn_wq_scales_per_row(m->wfmt, I), w.ti, w.tw, w.h, H, w.h, H, M, H, I, n_slot, stream, w.pp, nz,
                             (int64_t)M * H));
        nz2 = 0;
        continue;
      }
      // <= 4 rows on the fp32-FMA expert kernels: residual + RMSNorm(ln2) + gate + top-k in ONE launch (decode_ops.hip) instead of glue,
      // gate launch and the one-workgroup top-k
      if (!grouped && g_chain_router && moe_router_rows_ok(M, H, E)) {
        MN_TRY(moe_router_rows(w.h, w.pp, nz, m->ln2[l], m->rms_eps, m->gate[l], M, H, E, m->top_k, m->norm_topk_prob, S, w.xn, w.ti, w.tw,
                               w.logits, stream));
        route_capture(l, w.ti, M, n_slot, st);
        goto experts_2rows;
      }
      // glue: h += dense partials; RMSNorm(ln2) -> xn (fp32 for the 2-row expert kernels) and yh (gate + expert operand)
      hipLaunchKernelGGL(llm_glue_kernel, dim3(M), dim3(gt), 0, st, 1, (const float*)nullptr, (int64_t)0, 1, (const float*)w.pp, nz,
                         (int64_t)M * H, (const int32_t*)nullptr, (const float*)nullptr, n_slot, w.h, M, H, m->ln2[l],
                         m->rms_eps, grouped ? (float*)nullptr : w.xn, w.yh);
      // gate launch -> partials; ONE workgroup does softmax / top-k / expert sort  (:505-520, 565-592)
      nz = mn_stream_mfma(w.yh, m->gate[l], w.pp, M, E, H, stream);
      if (nz < 0) return nz;
      hipLaunchKernelGGL(moe_route_group_kernel, dim3(1), dim3(1024), 0, st, (const float*)w.pp, nz, M, E, m->top_k,
                         m->norm_topk_prob, S, w.tw, w.ti, G, grouped ? w.moe.off : (int32_t*)nullptr, w.moe.xrows,
                         w.moe.pair_pos);
      route_capture(l, w.ti, M, n_slot, st);
      if (grouped) {
        nz = stream_grouped(m->wfmt, w.yh, M, m->w_gate_up[l], (int64_t)2 * I * H, m->wfmt ? m->w_gate_up_scale[l] : nullptr, 2 * I,
                            w.moe.p1, P, w.moe.off, w.moe.xrows, G, M, 2 * I, H, stream);
        if (nz < 0) return nz;
        hipLaunchKernelGGL(rf_glue_swiglu_split_kernel, dim3(mn_cdiv((int64_t)P * I, 1024)), dim3(256), 0, st, w.moe.p1, nz, P,
                           I, (const bf16_t*)nullptr, w.moe.y2);
        nz2 = stream_grouped(m->wfmt, w.moe.y2, P, m->w_down[l], (int64_t)H * I, m->wfmt ? m->w_down_scale[l] : nullptr, H, w.moe.p2, P,
                             w.moe.off, nullptr, G, M, H, I, stream);
        if (nz2 < 0) return nz2;
      } else {   // 2 rows: (row, expert) pairs on the fp32-FMA kernels, accumulated straight into h
      experts_2rows:
        mn_skinny_args a = sk(w.xn, H, m->w_gate_up[l], H, nullptr, w.hmid, I, 1, I, H);
        a.epilogue = MN_EPI_SWIGLU;
        a.batch = M * n_slot; a.w_index = w.ti; a.w_batch_stride = (int64_t)2 * I * H;
        a.x_batch_stride = H; a.x_batch_div = n_slot; a.out_batch_stride = I;
        if (m->wfmt) { a.wfmt = m->wfmt; a.wscale = m->w_gate_up_scale[l]; a.wscale_batch_stride = 2 * I; a.ws = nullptr; }
        MN_TRY(mn_skinny_gemm(&a, stream));
        if (g_moe_down && moe_down_ok(m->wfmt, n_slot, H, I)) {      // segments over the waves: one HBM round trip (moe_down.hip)
          MN_TRY(moe_down_rows(m->wfmt, w.hmid, (int64_t)n_slot * I, m->w_down[l], (int64_t)H * I, m->wfmt ? m->w_down_scale[l] : nullptr, (int64_t)H * mn_wq_scales_per_row(m->wfmt, I), w.ti, w.tw, w.h, H, w.h, H, M, H, I, n_slot, stream));
          continue;
        }
        a = sk(w.hmid, (int64_t)n_slot * I, m->w_down[l], I, nullptr, w.h, H, 1, H, I);
        a.epilogue = MN_EPI_RESID; a.res = w.h; a.ldres = H; a.res_batch_stride = H;
        a.batch = M; a.w_index = nullptr; a.w_batch_stride = 0;
        a.x_batch_stride = (int64_t)n_slot * I; a.x_batch_div = 1; a.out_batch_stride = H;
        a.nseg = n_slot; a.seg_index = w.ti; a.seg_scale = w.tw; a.seg_w_stride = (int64_t)H * I;
        if (m->wfmt) { a.wfmt = m->wfmt; a.wscale = m->w_down_scale[l]; a.wscale_seg_stride = H; a.ws = nullptr; }
        MN_TRY(mn_skinny_gemm(&a, stream));
      }
    }
    // last experts' combine + residual -> final RMSNorm -> hidden_out
    hipLaunchKernelGGL(llm_glue_kernel, dim3(M), dim3(gt), 0, st, grouped ? 2 : 3, (const float*)nullptr, (int64_t)0, 1,
                       (const float*)w.moe.p2, nz2, (int64_t)P * H, (const int32_t*)w.moe.pair_pos, (const float*)w.tw, n_slot,
                       w.h, M, H, m->final_norm, m->rms_eps, hidden_out, (bf16_t*)nullptr);
    MN_CHECK_LAUNCH("mn_llm_step");
    return MN_OK;
  }
  hipLaunchKernelGGL(copy_rows_f32_kernel, dim3(mn_cdiv((int64_t)M * H, 256)), dim3(256), 0, st, x, ldx, x_row_div, w.h, M, H);
  for (int l = 0; l < m->n_layers; ++l) {
    float* kv_l = kv_cache + (int64_t)l * layer_kv;
    // attention: RMSNorm -> QKV -> RoPE/KV append -> masked GQA -> dense + residual  (:1204-1215, :743-829)
    mn_skinny_args a = sk(w.h, H, m->wqkv[l], H, nullptr, w.qkv, qkv_dim, M, qkv_dim, H);
    a.prologue = MN_PRO_RMSNORM; a.ln_g = m->ln1[l]; a.eps = m->rms_eps;
    MN_TRY(mn_skinny_gemm(&a, stream));
    if (fuse_attn) {
      MN_TRY(mn_attn_decode_fused(w.qkv, qkv_dim, 1, 0, M, nq, nkv, hd, 1, m->cos_tab, m->sin_tab, row_seq, row_slot, row_pos,
                                  m->mrope_sec_t, m->mrope_sec_h, q_scale, kv_l, t_max, row_len, key_mask, ld_mask, w.attn, nullptr,
                                  w.attn_ws, w.attn_ws_bytes, stream));
    } else {
      MN_TRY(mn_rope_kv_append_3d(w.qkv, qkv_dim, M, nq, nkv, hd, 1, m->cos_tab, m->sin_tab, row_seq, row_slot, row_pos,
                                  m->mrope_sec_t, m->mrope_sec_h, q_scale, w.q, kv_l, t_max, stream));
      MN_TRY(mn_attn_decode(w.q, M, nq, nkv, hd, kv_l, t_max, row_seq, row_len, key_mask, ld_mask, w.attn, w.attn_ws,
                            w.attn_ws_bytes, stream));
    }
    a = sk(w.attn, nq * hd, m->wdense[l], nq * hd, nullptr, w.h, H, M, H, nq * hd);
    a.epilogue = MN_EPI_RESID; a.res = w.h; a.ldres = H;
    MN_TRY(mn_skinny_gemm(&a, stream));
    // one row: RMSNorm + router + the selected experts' gate/up in ONE launch (moe_gate_up.hip), then the down projection (moe_down.hip) — every
    // weight format (int8 / NF4 decode per element like the grouped MFMA launch they replace here: text decode int8 434 -> 699, int4 426 -> 619 tokens/s)
    if (g_moe_gate_up && M <= g_moe_gate_up_rows && !(image_mask && m->image_gate) && g_moe_down &&
        moe_gate_up_ok(m->wfmt, H, I, m->n_experts, m->top_k, m->n_shared_slots) && moe_down_ok(m->wfmt, n_slot, H, I)) {
      MN_TRY(moe_gate_up_routed(m->wfmt, w.h, H, m->ln2[l], m->rms_eps, m->gate[l], m->w_gate_up[l], (int64_t)2 * I * H,
                                m->wfmt ? m->w_gate_up_scale[l] : nullptr, (int64_t)2 * I * mn_wq_scales_per_row(m->wfmt, H), M, H, I, m->n_experts, m->top_k,
                                m->n_shared_slots, m->norm_topk_prob, w.hmid, (int64_t)n_slot * I, w.ti, w.tw, w.logits, nullptr, 0, 0, stream));
      route_capture(l, w.ti, M, n_slot, st);
      MN_TRY(moe_down_rows(m->wfmt, w.hmid, (int64_t)n_slot * I, m->w_down[l], (int64_t)H * I, m->wfmt ? m->w_down_scale[l] : nullptr, (int64_t)H * mn_wq_scales_per_row(m->wfmt, I), w.ti, w.tw, w.h, H, w.h, H, M, H, I, n_slot, stream));
      continue;
    }
    // MoE: RMSNorm + router -> grouped expert gate/up (SwiGLU) -> grouped down + weighted sum + residual (:1218-1225, :556-639)
    MN_TRY(mn_moe_router(w.h, H, m->ln2[l], m->rms_eps, m->gate[l], m->image_gate ? m->image_gate[l] : nullptr,
                         image_mask, M, H, m->n_experts, m->top_k, m->norm_topk_prob, m->n_shared_slots, w.xn, w.ti,
                         w.tw, w.logits, w.sk_ws, w.sk_ws_bytes, stream));
    route_capture(l, w.ti, M, n_slot, st);
    if (moe_mfma_ok(m, M)) {
      const int G = m->n_experts + m->n_shared_slots, P = M * n_slot;
      hipLaunchKernelGGL(moe_group_split_kernel, dim3(M + 1), dim3(256), 0, st, w.ti, M, n_slot, G, w.moe.off, w.moe.xrows,
                         w.moe.pair_pos, w.xn, H, w.moe.y1);
      int nz = stream_grouped(m->wfmt, w.moe.y1, M, m->w_gate_up[l], (int64_t)2 * I * H, m->wfmt ? m->w_gate_up_scale[l] : nullptr,
                              2 * I, w.moe.p1, P, w.moe.off, w.moe.xrows, G, M, 2 * I, H, stream);
      if (nz < 0) return nz;
      hipLaunchKernelGGL(rf_glue_swiglu_split_kernel, dim3(mn_cdiv((int64_t)P * I, 1024)), dim3(256), 0, st, w.moe.p1, nz, P, I,
                         (const bf16_t*)nullptr, w.moe.y2);
      nz = stream_grouped(m->wfmt, w.moe.y2, P, m->w_down[l], (int64_t)H * I, m->wfmt ? m->w_down_scale[l] : nullptr, H, w.moe.p2, P,
                          w.moe.off, nullptr, G, M, H, I, stream);
      if (nz < 0) return nz;
      hipLaunchKernelGGL(moe_combine_resid_kernel, dim3(mn_cdiv((int64_t)M * H, 1024)), dim3(256), 0, st, w.moe.p2, nz,
                         (int64_t)P * H, M, H, n_slot, w.moe.pair_pos, w.tw, w.h);
      continue;
    }
    a = sk(w.xn, H, m->w_gate_up[l], H, nullptr, w.hmid, I, 1, I, H);
    a.epilogue = MN_EPI_SWIGLU;
    a.batch = M * n_slot; a.w_index = w.ti; a.w_batch_stride = (int64_t)2 * I * H;
    a.x_batch_stride = H; a.x_batch_div = n_slot; a.out_batch_stride = I;
    if (m->wfmt) { a.wfmt = m->wfmt; a.wscale = m->w_gate_up_scale[l]; a.wscale_batch_stride = 2 * I; a.ws = nullptr; }
    MN_TRY(mn_skinny_gemm(&a, stream));
    if (g_moe_down && moe_down_ok(m->wfmt, n_slot, H, I)) {          // segments over the waves: one HBM round trip (moe_down.hip)
      MN_TRY(moe_down_rows(m->wfmt, w.hmid, (int64_t)n_slot * I, m->w_down[l], (int64_t)H * I, m->wfmt ? m->w_down_scale[l] : nullptr, (int64_t)H * mn_wq_scales_per_row(m->wfmt, I), w.ti, w.tw, w.h, H, w.h, H, M, H, I, n_slot, stream));
      continue;
    }
    a = sk(w.hmid, (int64_t)n_slot * I, m->w_down[l], I, nullptr, w.h, H, 1, H, I);
    a.epilogue = MN_EPI_RESID; a.res = w.h; a.ldres = H; a.res_batch_stride = H;
    a.batch = M; a.w_index = nullptr; a.w_batch_stride = 0;
    a.x_batch_stride = (int64_t)n_slot * I; a.x_batch_div = 1; a.out_batch_stride = H;
    a.nseg = n_slot; a.seg_index = w.ti; a.seg_scale = w.tw; a.seg_w_stride = (int64_t)H * I;
    if (m->wfmt) { a.wfmt = m->wfmt; a.wscale = m->w_down_scale[l]; a.wscale_seg_stride = H; a.ws = nullptr; }
    MN_TRY(mn_skinny_gemm(&a, stream));
  }
  hipLaunchKernelGGL(rmsnorm_f32_kernel, dim3(M), dim3(256), 0, st, w.h, (int64_t)H, m->final_norm, m->rms_eps,
                     hidden_out, (int64_t)H, H);
  MN_CHECK_LAUNCH("mn_llm_step");
  return MN_OK;
}

// ===========================================================================================
// MingTok semantic decoder: cached causal decode step + linear_proj
// ===========================================================================================
struct SemWs {
  float *h, *qkv, *q, *attn, *hid, *sem, *p0;
  void* attn_ws;
  size_t attn_ws_bytes;
  char* sk_ws;
  size_t sk_ws_bytes;
};

static size_t sem_carve(const mn_semdec* s, int rows, int64_t t_max, void* ws, size_t cap, SemWs* o) {
  Carver cv(ws, cap, ws == nullptr);
  o->h = cv.take<float>((size_t)rows * s->dim);
  o->qkv = cv.take<float>((size_t)rows * 3 * s->dim);
  o->q = cv.take<float>((size_t)rows * s->dim);
  o->attn = cv.take<float>((size_t)rows * s->dim);
  o->hid = cv.take<float>((size_t)rows * s->hidden);
  o->sem = cv.take<float>((size_t)rows * s->dim);
  o->p0 = cv.take<float>((size_t)rows * s->proj_dim * 2);
  o->attn_ws_bytes = mn_attn_decode_workspace_bytes(rows, s->n_heads, 64, t_max);
  o->attn_ws = cv.take<char>(o->attn_ws_bytes);
  o->sk_ws_bytes = sk_ws_need(rows, {{3 * s->dim, s->dim, 0}, {s->dim, s->dim, 0}, {s->hidden, s->dim, MN_EPI_SWIGLU},
                                      {s->dim, s->hidden, 0}, {s->proj_dim, s->dim, 0}, {s->proj_dim, s->proj_dim, 0}});
  o->sk_ws = cv.take<char>(o->sk_ws_bytes);
  return cv.off;
}

extern "C" size_t mn_semdec_workspace_bytes(const mn_semdec* s, int rows, int64_t t_max) {
  if (sem_wide_ok(s, rows)) { SemWideWs ww; return sem_wide_carve(s, rows, t_max, nullptr, 0, &ww); }
  SemWs w;
  return sem_carve(s, rows, t_max, nullptr, 0, &w);
}

extern "C" int mn_semdec_step(const mn_semdec* s, const float* latent_norm, int M, const int32_t* row_seq,
                              const int32_t* row_slot, const int32_t* row_len, float* kv_cache, int n_seq,
                              int64_t t_max, float* sem_out, float* embed_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  MN_CHECK_ARG(s && latent_norm && row_seq && row_slot && row_len && kv_cache && workspace, "mn_semdec_step: null pointer");
  MN_CHECK_ARG(M >= 1 && (M <= 64 || sem_wide_ok(s, M)) && s->dim == s->n_heads * 64 && s->dim % s->in_dim == 0,
               "mn_semdec_step: bad shape (M = %d: 1..64 rows, or up to 2048 with the padded SwiGLU weights)", M);

  MN_CHECK_ARG(!embed_out || (s->proj_depth >= 1 && s->proj_depth <= 2), "mn_semdec_step: proj_depth must be 1 or 2");
  if (sem_wide_ok(s, M))
    return semdec_step_wide(s, latent_norm, M, row_seq, row_slot, row_len, kv_cache, n_seq, t_max, sem_out, embed_out, workspace,
                            workspace_bytes, stream);
  SemWs w;
  const size_t need = sem_carve(s, M, t_max, workspace, workspace_bytes, &w);
  if (need > workspace_bytes) { mn_set_error("mn_semdec_step: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  hipStream_t st = mn_stream(stream);
  const int D = s->dim, nh = s->n_heads;
  const int64_t layer_kv = (int64_t)n_seq * 2 * nh * t_max * 64;
  t_sk_ws = w.sk_ws; t_sk_ws_bytes = w.sk_ws_bytes;
  hipLaunchKernelGGL(semdec_in_kernel, dim3(mn_cdiv(D, 256), M), dim3(256), 0, st, latent_norm, s->in_dim, s->scale,
                     s->mean, s->in_w, s->in_b, w.h, D);
  for (int l = 0; l < s->depth; ++l) {
    float* kv_l = kv_cache + (int64_t)l * layer_kv;
    mn_skinny_args a = sk(w.h, D, s->wqkv[l], D, s->bqkv[l], w.qkv, 3 * D, M, 3 * D, D);
    a.prologue = MN_PRO_LN; a.ln_g = s->ln1_g[l]; a.ln_b = s->ln1_b[l]; a.eps = 1e-6f;
    MN_TRY(mn_skinny_gemm(&a, stream));
    if (mn_attn_fused_ok(M, nh, nh, 64, t_max)) {     // one row (batch 1): the K / V append rides the attention launch (see mn_llm_step_ex)
      MN_TRY(mn_attn_decode_fused(w.qkv, 3 * D, 1, 0, M, nh, nh, 64, 0, nullptr, nullptr, row_seq, row_slot, nullptr, 0, 0, 0.125f, kv_l,
                                  t_max, row_len, nullptr, 0, w.attn, nullptr, w.attn_ws, w.attn_ws_bytes, stream));
    } else {
      MN_TRY(mn_rope_kv_append(w.qkv, 3 * D, M, nh, nh, 64, 0, nullptr, nullptr, row_seq, row_slot, nullptr, 0.125f,
                               w.q, kv_l, t_max, stream));
      MN_TRY(mn_attn_decode(w.q, M, nh, nh, 64, kv_l, t_max, row_seq, row_len, nullptr, 0, w.attn, w.attn_ws,
                            w.attn_ws_bytes, stream));
    }
    a = sk(w.attn, D, s->wproj[l], D, s->bproj[l], w.h, D, M, D, D);
    a.epilogue = MN_EPI_RESID; a.res = w.h; a.ldres = D;
    MN_TRY(mn_skinny_gemm(&a, stream));
    a = sk(w.h, D, s->w12[l], D, s->b12[l], w.hid, s->hidden, M, s->hidden, D);
    a.prologue = MN_PRO_LN; a.ln_g = s->ln2_g[l]; a.ln_b = s->ln2_b[l]; a.eps = 1e-6f;
    a.epilogue = MN_EPI_SWIGLU;
    MN_TRY(mn_skinny_gemm(&a, stream));
    a = sk(w.hid, s->hidden, s->w3[l], s->hidden, s->b3[l], w.h, D, M, D, s->hidden);
    a.epilogue = MN_EPI_RESID; a.res = w.h; a.ldres = D;
    MN_TRY(mn_skinny_gemm(&a, stream));
  }
  float* sem = sem_out ? sem_out : w.sem;
  hipLaunchKernelGGL(layernorm_f32_kernel, dim3(M), dim3(256), 0, st, w.h, (int64_t)D, s->norm_g, s->norm_b, 1e-6f, sem,
                     (int64_t)D, D);
  if (embed_out) {
    // linear_proj = Linear [GELU Linear]  (modeling_bailingmm.py:111-115)
    if (s->proj_depth == 1) {
      mn_skinny_args a = sk(sem, D, s->proj_w[0], D, s->proj_b[0], embed_out, s->proj_dim, M, s->proj_dim, D);
      MN_TRY(mn_skinny_gemm(&a, stream));
    } else {
      mn_skinny_args a = sk(sem, D, s->proj_w[0], D, s->proj_b[0], w.p0, s->proj_dim, M, s->proj_dim, D);
      a.epilogue = MN_EPI_GELU;
      MN_TRY(mn_skinny_gemm(&a, stream));
      a = sk(w.p0, s->proj_dim, s->proj_w[1], s->proj_dim, s->proj_b[1], embed_out, s->proj_dim, M, s->proj_dim, s->proj_dim);
      MN_TRY(mn_skinny_gemm(&a, stream));
    }
  }
  MN_CHECK_LAUNCH("mn_semdec_step");
  return MN_OK;
}


// ===========================================================================================
// lm_head + greedy pick  (compute_logit, modeling_bailing_moe.py:1604-1620, followed by the argmax of greedy decoding)
// ===========================================================================================
namespace {
// one workgroup per row: arg-max over V fp32 logits, ties -> the lowest index (torch.argmax's rule)
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float* __restrict__ logits, int64_t ld, int V, int64_t vocab_offset,
                                                           int64_t* __restrict__ idx, float* __restrict__ val) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* r = logits + (int64_t)m * ld;
  // torch.argmax's order: NaN is the maximum, equal values (NaN among NaN too) resolve to the lowest index — a row of NaN logits
  // yields a VALID id (the first NaN), never an out-of-range one
  auto better = [](float v, int j, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || j < bi);
    return v > bv || (v == bv && j < bi);
  };
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = threadIdx.x; j < V; j += 1024) {
    const float v = r[j];
    if (better(v, j, bv, bi)) { bv = v; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w2 = 1; w2 < 16; ++w2)
      if (better(sv[w2], si[w2], bv, bi)) { bv = sv[w2]; bi = si[w2]; }
    if (bi >= V) bi = 0;                               // V >= 1: unreachable, kept as the last line of defence for the gather that follows
    idx[m] = (int64_t)bi + vocab_offset;
    if (val) val[m] = bv;
  }
}
// Few rows (text decode): one workgroup scanning 126 k logits was 46 us of a 1.9 ms token — the scan is spread over ARGMAX_PARTS
// workgroups per row (same comparator, so the same winner: value, then lowest index; NaN first), a 64-lane launch picks among them.
constexpr int ARGMAX_PARTS = 64;
__device__ __forceinline__ bool argmax_better(float v, int j, float bv, int bi) {
  const bool vn = v != v, bn = bv != bv;
  if (vn || bn) return vn && (!bn || j < bi);
  return v > bv || (v == bv && j < bi);
}
__global__ __launch_bounds__(256) void argmax_part_kernel(const float* __restrict__ logits, int64_t ld, int V, float* __restrict__ pv,
                                                          int* __restrict__ pi) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int m = blockIdx.x, part = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = (V + ARGMAX_PARTS - 1) / ARGMAX_PARTS;
  const int j0 = part * chunk, j1 = min(V, j0 + chunk);
  const float* r = logits + (int64_t)m * ld;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int jb = j0; jb < j1; jb += 2048) {         // eight independent loads per thread in flight (one batch at V = 126 464)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = jb + threadIdx.x + u * 256;
      v[u] = j < j1 ? r[j] : -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = jb + threadIdx.x + u * 256;
      if (j < j1 && argmax_better(v[u], j, bv, bi)) { bv = v[u]; bi = j; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (argmax_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w2 = 1; w2 < 4; ++w2)
      if (argmax_better(sv[w2], si[w2], bv, bi)) { bv = sv[w2]; bi = si[w2]; }
    pv[m * ARGMAX_PARTS + part] = bv;
    pi[m * ARGMAX_PARTS + part] = bi;
  }
}
__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int V, int64_t vocab_offset,
                                                          int64_t* __restrict__ idx, float* __restrict__ val) {
  const int m = blockIdx.x, lane = threadIdx.x;
  float bv = pv[m * ARGMAX_PARTS + lane];
  int bi = pi[m * ARGMAX_PARTS + lane];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (argmax_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) {
    if (bi >= V) bi = 0;
    idx[m] = (int64_t)bi + vocab_offset;
    if (val) val[m] = bv;
  }
}
}  // namespace

extern "C" size_t mn_lmhead_argmax_workspace_bytes(int M, int V, int H) {
  size_t n = (((size_t)M * V * sizeof(float)) + 255) & ~(size_t)255;        // the logits
  n += (((size_t)M * ARGMAX_PARTS * 8) + 255) & ~(size_t)255;              // (value, index) of every scan part
  if (M > 8) n += (((size_t)2 * M * H * sizeof(bf16_t)) + 255) & ~(size_t)255;   // hi/lo operand of the MFMA route
  return n;
}

// idx[m] = vocab_offset + argmax_v ( hidden[m] . W[v] ), val[m] = that logit (fp32; may be NULL).  W bf16 [V, H] is the whole
// lm_head or, under tensor parallelism, this rank's vocabulary slice starting at vocab_offset (the ranks' (val, idx) pairs are
// then reduced by max-val / lowest-idx).  <= 8 rows: the weight-streaming skinny kernel (HBM-bound: 0.52 GB per token at V =
// 126 464); more rows: one gemm256 launch on the hi/lo operand.  The logits stay in the workspace (fp32 [M, V], its first bytes).
extern "C" int mn_lmhead_argmax(const float* hidden, int64_t ld_hidden, int M, const uint16_t* W, int64_t ldw, int V, int H,
                                int64_t vocab_offset, int64_t* idx, float* val, void* workspace, size_t workspace_bytes, void* stream) {
  MN_CHECK_ARG(hidden && W && idx && workspace && M >= 1 && V >= 1 && H >= 8 && (H % 8) == 0, "mn_lmhead_argmax: bad args");
  const size_t need = mn_lmhead_argmax_workspace_bytes(M, V, H);
  if (workspace_bytes < need) { mn_set_error("mn_lmhead_argmax: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  float* logits = reinterpret_cast<float*>(workspace);
  hipStream_t st = mn_stream(stream);
  const bool mfma = M > 8 && (H % 64) == 0 && (V % 4) == 0 && wide_glue_ok(H) && (ld_hidden % 4) == 0 &&
                    (int64_t)V * ldw * 2 < ((int64_t)1 << 32) && (int64_t)2 * M * H * 2 < ((int64_t)1 << 32);
  if (mfma) {
    bf16_t* y = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(workspace) + ((((size_t)M * V * sizeof(float)) + 255) & ~(size_t)255));
    WideGlue g;
    memset(&g, 0, sizeof(g));
    g.h = hidden; g.ldh = ld_hidden; g.Y = y; g.ldy = H; g.y_lo_off = (int64_t)M * H; g.M = M; g.D = H;
    wide_glue(g, st);
    mn_g256 a = g256_hilo(y, H, (int64_t)M * H, W, ldw, nullptr, logits, V, M, V, H);
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
  } else {
    for (int m0 = 0; m0 < M; m0 += 8) {
      mn_skinny_args a;
      memset(&a, 0, sizeof(a));
      a.x = hidden + (int64_t)m0 * ld_hidden; a.ldx = ld_hidden; a.w = W; a.ldw = ldw; a.out = logits + (int64_t)m0 * V; a.ldo = V;
      a.M = M - m0 < 8 ? M - m0 : 8; a.N = V; a.K = H;
      MN_TRY(mn_skinny_gemm(&a, stream));
    }
  }
  if (M <= 8 && V >= 16 * ARGMAX_PARTS) {
    float* pv = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + need - ((((size_t)M * ARGMAX_PARTS * 8) + 255) & ~(size_t)255));
    int* pi = reinterpret_cast<int*>(pv + (size_t)M * ARGMAX_PARTS);
    hipLaunchKernelGGL(argmax_part_kernel, dim3(M, ARGMAX_PARTS), dim3(256), 0, st, (const float*)logits, (int64_t)V, V, pv, pi);
    hipLaunchKernelGGL(argmax_final_kernel, dim3(M), dim3(64), 0, st, (const float*)pv, (const int*)pi, V, vocab_offset, idx, val);
  } else {
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(M), dim3(1024), 0, st, (const float*)logits, (int64_t)V, V, vocab_offset, idx, val);
  }
  MN_CHECK_LAUNCH("mn_lmhead_argmax");
  return MN_OK;
}

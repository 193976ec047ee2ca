// skinny_device.h — device building blocks shared by the skinny GEMM launches (skinny_gemm.hip) and the
// persistent RF-block kernel (rf_persistent.hip): LDS staging of the activation rows with the fused
// prologues, and the per-chunk FMA.
#pragma once
#include "common.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct KArgs {  // device copy of mn_skinny_args (plain data)
  mn_skinny_args a;
  int32_t nchunk;   // chunks of 512 k per segment
  int32_t nseg;
  int32_t batch;
  int32_t inv_nchunk;  // ceil(65536 / nchunk): ct / nchunk == (ct * inv_nchunk) >> 16 for ct < 4096
};

__device__ __forceinline__ int perm_k(int k) {
  return (k & ~511) | (((k >> 2) & 1) << 8) | (((k >> 3) & 63) << 2) | (k & 3);
}

// Block-wide sums of M per-thread values (one barrier pair for all rows).
template <int M, int NT>
__device__ __forceinline__ void block_sum_multi(float (&v)[M], float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < M; ++m) v[m] = wave_sum(v[m]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < M; ++m) red[wave * M + m] = v[m];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < M; ++m) {
    float t = 0.f;
    for (int i = 0; i < NT / 64; ++i) t += red[i * M + m];
    v[m] = t;
  }
}

// Stage prologue(x) for this block's batch entry into LDS. xs layout: [M][nseg][nchunk*512] permuted.
template <int M, int NT>
__device__ void stage_x(const KArgs& ka, float* xs, float* red, int b) {
  const mn_skinny_args& a = ka.a;
  const int K = a.K, Kp = ka.nchunk << 9, nseg = ka.nseg;
  const int tid = threadIdx.x;
  const float* xb = a.x + (int64_t)(b / (a.x_batch_div > 0 ? a.x_batch_div : 1)) * a.x_batch_stride;
  const int pro = a.prologue;

  if (pro == MN_PRO_NONE || pro == MN_PRO_SILU || pro == MN_PRO_ADD_SILU) {
    for (int s = 0; s < nseg; ++s) {
      const float sc = a.seg_scale ? a.seg_scale[(int64_t)b * nseg + s] : 1.0f;
      for (int k = tid; k < Kp; k += NT) {
        const int p = perm_k(k);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          float v = 0.f;
          if (k < K) {
            v = xb[(int64_t)m * a.ldx + s * K + k];
            if (pro == MN_PRO_ADD_SILU) v += a.pro_a[(int64_t)m * a.ld_pro_a + k];
            if (pro != MN_PRO_NONE) v = silu_f(v);
            v *= sc;
          }
          xs[((int64_t)m * nseg + s) * Kp + p] = v;
        }
      }
    }
    return;
  }
  // Normalising prologues (nseg == 1): statistics of all M rows together, from the LDS-resident raw rows.
  float s1[M];
#pragma unroll
  for (int m = 0; m < M; ++m) s1[m] = 0.f;
  for (int k = tid; k < Kp; k += NT) {
    const int p = perm_k(k);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float v = (k < K) ? xb[(int64_t)m * a.ldx + k] : 0.f;
      xs[(int64_t)m * Kp + p] = v;
      s1[m] += v;
    }
  }
  float mean[M], rstd[M];
  if (pro == MN_PRO_RMSNORM) {
#pragma unroll
    for (int m = 0; m < M; ++m) mean[m] = 0.f;
  } else {
    block_sum_multi<M, NT>(s1, red);
#pragma unroll
    for (int m = 0; m < M; ++m) mean[m] = s1[m] / (float)K;
  }
  float s2[M];
#pragma unroll
  for (int m = 0; m < M; ++m) s2[m] = 0.f;
  for (int k = tid; k < K; k += NT) {
    const int p = perm_k(k);
#pragma unroll
    for (int m = 0; m < M; ++m) { const float d = xs[(int64_t)m * Kp + p] - mean[m]; s2[m] += d * d; }
  }
  block_sum_multi<M, NT>(s2, red);
#pragma unroll
  for (int m = 0; m < M; ++m) rstd[m] = rsqrtf(s2[m] / (float)K + a.eps);
  for (int k = tid; k < K; k += NT) {
    const int p = perm_k(k);
    const float g = a.ln_g ? bf16_to_f32(a.ln_g[k]) : 1.0f;
    const float be = (a.ln_b && pro != MN_PRO_RMSNORM) ? bf16_to_f32(a.ln_b[k]) : 0.0f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float v = (xs[(int64_t)m * Kp + p] - mean[m]) * rstd[m] * g + be;
      if (pro == MN_PRO_LN_MOD)
        v = v * (1.0f + a.pro_b[(int64_t)m * a.ld_pro_b + k]) + a.pro_a[(int64_t)m * a.ld_pro_a + k];
      xs[(int64_t)m * Kp + p] = v;
    }
  }
}

template <int M>
__device__ __forceinline__ void fma_chunk(const u32x4 w, const float* xrow0, int64_t xstride, float (&acc)[M]) {
  const float w0 = bf16lo_to_f32(w.x), w1 = bf16hi_to_f32(w.x), w2 = bf16lo_to_f32(w.y), w3 = bf16hi_to_f32(w.y);
  const float w4 = bf16lo_to_f32(w.z), w5 = bf16hi_to_f32(w.z), w6 = bf16lo_to_f32(w.w), w7 = bf16hi_to_f32(w.w);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const f32x4 xa = *reinterpret_cast<const f32x4*>(xrow0 + m * xstride);
    const f32x4 xb = *reinterpret_cast<const f32x4*>(xrow0 + m * xstride + 256);
    float t = acc[m];
    t = fmaf(w0, xa.x, t); t = fmaf(w1, xa.y, t); t = fmaf(w2, xa.z, t); t = fmaf(w3, xa.w, t);
    t = fmaf(w4, xb.x, t); t = fmaf(w5, xb.y, t); t = fmaf(w6, xb.z, t); t = fmaf(w7, xb.w, t);
    acc[m] = t;
  }
}

// fp8_ops.hip — load-time row-wise quantisation of bf16 weights to OCP e4m3fn + one power-of-two fp32 scale per output row
// (mingnative.h section 7), and the inverse.  HBM-bound elementwise work, run once per tensor when a model is loaded in fp8 mode.
//
// Why power-of-two scales: e4m3 is a floating-point format, so a scale that maps a row's amax into (224, 448] instead of exactly
// onto 448 costs no relative precision on normal values (at most one binade of range at the subnormal end), and it makes
// e4m3(q) * scale exactly representable in bf16 — the dequantised model is a bf16 model, runnable through every bf16 route and
// through the fp32 oracle with bit-identical weights.
#include "common.h"
#include "w8_codec.h"

namespace {

// scale = 2^es with amax / 2^es in (224, 448]:  amax = ma * 2^ea (1 <= ma < 2), 448 = 1.75 * 2^8  ->  es = ea - 8 (+1 if ma > 1.75)
__device__ __forceinline__ float pow2_scale_for(float amax) {
  const uint32_t u = __float_as_uint(amax);
  if ((u & 0x7fffffffu) == 0u) return 1.0f;
  int es = (int)(u >> 23) - 127 - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
  es = es < -126 ? -126 : (es > 127 ? 127 : es);
  return __uint_as_float((uint32_t)(es + 127) << 23);
}

// one workgroup per row; K % 4 == 0
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ W, int64_t ldw, uint8_t* __restrict__ Q, int64_t ldq,
                                                             float* __restrict__ scale, int K) {
  __shared__ float red[4];
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int64_t n = blockIdx.x;
  const bf16_t* wr = W + n * ldw;
  float amax = 0.f;
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    const u2 v = *reinterpret_cast<const u2*>(wr + k);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf16lo_to_f32(v.x)), fabsf(bf16hi_to_f32(v.x))), fmaxf(fabsf(bf16lo_to_f32(v.y)), fabsf(bf16hi_to_f32(v.y)))));
  }
  amax = block_max(amax, red);
  const float s = pow2_scale_for(amax);
  const float inv = 1.0f / s;                          // exact: s is a power of two
  if (threadIdx.x == 0) scale[n] = s;
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    const u2 v = *reinterpret_cast<const u2*>(wr + k);
    int q = 0;
    q = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo_to_f32(v.x) * inv, bf16hi_to_f32(v.x) * inv, q, false);   // v_cvt_pk_fp8_f32: RNE
    q = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo_to_f32(v.y) * inv, bf16hi_to_f32(v.y) * inv, q, true);
    *reinterpret_cast<uint32_t*>(Q + n * ldq + k) = (uint32_t)q;
  }
}

__global__ __launch_bounds__(256) void dequant_fp8_rows_kernel(const uint8_t* __restrict__ Q, int64_t ldq, const float* __restrict__ scale,
                                                               bf16_t* __restrict__ W, int64_t ldw, int K) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int64_t n = blockIdx.x;
  const float s = scale[n];
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    const uint32_t q = *reinterpret_cast<const uint32_t*>(Q + n * ldq + k);
    const mn_f2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(q, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(q, true);
    *reinterpret_cast<u2*>(W + n * ldw + k) = u2{cvt_pk_bf16(a.x * s, a.y * s), cvt_pk_bf16(b.x * s, b.y * s)};
  }
}

// ---- int8 (MN_W_INT8): optimum-quanto's qint8 weights (the reference's dtype="int8", mingunivisioninfer.py:59-68; restated in
// oracle/int8_ref.py): symmetric, one scale per output row, everything in the weight's dtype (bf16):
//     scale[n] = bf16(amax_n / 127)            (absmax_scale: qranges / qmax on a bf16 tensor)
//     q[n, k]  = clamp(round_half_even(bf16(W[n, k] / scale[n])), -128, 127)      (quantize_symmetric: base / scale is a bf16 tensor)
//     W'[n, k] = bf16(q[n, k] * scale[n])      (qbytes_mm multiplies scale * weights in the activation dtype before the matmul)
// The scale is stored as fp32 holding the bf16 value.
__global__ __launch_bounds__(256) void quant_int8_rows_kernel(const bf16_t* __restrict__ W, int64_t ldw, uint8_t* __restrict__ Q, int64_t ldq,
                                                              float* __restrict__ scale, int K) {
  __shared__ float red[4];
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int64_t n = blockIdx.x;
  const bf16_t* wr = W + n * ldw;
  float amax = 0.f;
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    const u2 v = *reinterpret_cast<const u2*>(wr + k);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf16lo_to_f32(v.x)), fabsf(bf16hi_to_f32(v.x))), fmaxf(fabsf(bf16lo_to_f32(v.y)), fabsf(bf16hi_to_f32(v.y)))));
  }
  amax = block_max(amax, red);
  float s = bf16_to_f32(f32_to_bf16(amax / 127.0f));
  if (s == 0.f) s = 1.0f;                               // an all-zero row (or an amax that underflows bf16): q = 0
  if (threadIdx.x == 0) scale[n] = s;
  auto q8 = [&](float w) { return (uint32_t)((int)fminf(fmaxf(rintf(bf16_to_f32(f32_to_bf16(w / s))), -128.f), 127.f) & 0xff); };
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    const u2 v = *reinterpret_cast<const u2*>(wr + k);
    *reinterpret_cast<uint32_t*>(Q + n * ldq + k) =
        q8(bf16lo_to_f32(v.x)) | (q8(bf16hi_to_f32(v.x)) << 8) | (q8(bf16lo_to_f32(v.y)) << 16) | (q8(bf16hi_to_f32(v.y)) << 24);
  }
}

__global__ __launch_bounds__(256) void dequant_int8_rows_kernel(const uint8_t* __restrict__ Q, int64_t ldq, const float* __restrict__ scale,
                                                                bf16_t* __restrict__ W, int64_t ldw, int K) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int64_t n = blockIdx.x;
  const float s = scale[n];
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    const mn_u2_t o = i8x4_to_bf16(*reinterpret_cast<const uint32_t*>(Q + n * ldq + k), s);      // the kernels' own conversion (w8_codec.h)
    *reinterpret_cast<u2*>(W + n * ldw + k) = u2{o.x, o.y};
  }
}

// ---- NF4 (MN_W_NF4): bitsandbytes' blockwise 4-bit NormalFloat (oracle/int4_ref.py; mingunivisioninfer.py:46-58).  One thread owns
// eight consecutive k (one dword of codes), eight threads one 64-element block: absmax by three xor-shuffles, x = w * (1 / absmax) in fp32,
// code = number of the 15 midpoints below x (dQuantizeNF4's comparison tree: a value ON a midpoint takes the lower entry).
__global__ __launch_bounds__(256) void quant_nf4_rows_kernel(const bf16_t* __restrict__ W, int64_t ldw, uint8_t* __restrict__ Q, int64_t ldq,
                                                             float* __restrict__ absmax, int K) {
  constexpr float T[16] = MN_NF4_TABLE;
  const int64_t n = blockIdx.x;
  const bf16_t* wr = W + n * ldw;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {      // K % 64 == 0: the eight lanes of a block are in or out together
    const mn_u4_t v = *reinterpret_cast<const mn_u4_t*>(wr + k);
    const float w[8] = {bf16lo_to_f32(v.x), bf16hi_to_f32(v.x), bf16lo_to_f32(v.y), bf16hi_to_f32(v.y),
                        bf16lo_to_f32(v.z), bf16hi_to_f32(v.z), bf16lo_to_f32(v.w), bf16hi_to_f32(v.w)};
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a = fmaxf(a, fabsf(w[i]));
    a = fmaxf(a, __shfl_xor(a, 1, 64));
    a = fmaxf(a, __shfl_xor(a, 2, 64));
    a = fmaxf(a, __shfl_xor(a, 4, 64));
    const float inv = a == 0.f ? 0.f : 1.0f / a;
    if ((threadIdx.x & 7) == 0) absmax[n * (K >> 6) + (k >> 6)] = a;
    uint32_t q = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = w[i] * inv;
      uint32_t c = 0;
#pragma unroll
      for (int j = 0; j < 15; ++j) c += x > (float)(((double)T[j] + (double)T[j + 1]) * 0.5) ? 1u : 0u;
      q |= c << (i < 4 ? 8 * i : 8 * (i - 4) + 4);        // nibble order e0 e4 e1 e5 e2 e6 e3 e7 (w8_codec.h)
    }
    *reinterpret_cast<uint32_t*>(Q + n * ldq + (k >> 1)) = q;
  }
}

__global__ __launch_bounds__(256) void dequant_nf4_rows_kernel(const uint8_t* __restrict__ Q, int64_t ldq, const float* __restrict__ absmax,
                                                               bf16_t* __restrict__ W, int64_t ldw, int K) {
  const int64_t n = blockIdx.x;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    const Nf4Tab t = nf4_table(absmax[n * (K >> 6) + (k >> 6)]);
    *reinterpret_cast<mn_u4_t*>(W + n * ldw + k) = nf4x8_to_bf16(t, *reinterpret_cast<const uint32_t*>(Q + n * ldq + (k >> 1)));
  }
}

// ---- the byte formats' expansion for CONTIGUOUS row blocks (ldq = ldw = K), 16 weights per thread: the wide route
// of the weight-only modes expands a decoder layer's experts on every step (wide_llm.inl) — 46 GB of traffic per step at the 16B-A3B
// shape — and the row-per-workgroup kernels above (4-byte loads, written for load time) moved it at 3.9 TB/s; these: 5.5 TB/s.  (NF4's
// row kernel already stores 16 bytes per lane, coalesced; a 32-weights-per-thread form measured slower.)
template <bool I8>
__global__ __launch_bounds__(256) void dequant8_flat_kernel(const uint8_t* __restrict__ Q, const float* __restrict__ scale, bf16_t* __restrict__ W,
                                                            int64_t n_pieces, int K) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;                 // piece = 16 consecutive weights of one row (K % 16 == 0)
  if (i >= n_pieces) return;
  const int64_t e = i * 16;
  const float s = scale[e / K];
  const mn_u4_t q = __builtin_nontemporal_load(reinterpret_cast<const mn_u4_t*>(Q + e));
  mn_u4_t lo, hi;
  if constexpr (I8) {
    const mn_u2_t a = i8x4_to_bf16(q.x, s), b = i8x4_to_bf16(q.y, s), c = i8x4_to_bf16(q.z, s), d = i8x4_to_bf16(q.w, s);
    lo = mn_u4_t{a.x, a.y, b.x, b.y}; hi = mn_u4_t{c.x, c.y, d.x, d.y};
  } else {
    const uint32_t qs[4] = {q.x, q.y, q.z, q.w};
    uint32_t o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const mn_f2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(qs[j], false), b = __builtin_amdgcn_cvt_pk_f32_fp8(qs[j], true);
      o[2 * j] = cvt_pk_bf16(a.x * s, a.y * s); o[2 * j + 1] = cvt_pk_bf16(b.x * s, b.y * s);
    }
    lo = mn_u4_t{o[0], o[1], o[2], o[3]}; hi = mn_u4_t{o[4], o[5], o[6], o[7]};
  }
  *reinterpret_cast<mn_u4_t*>(W + e) = lo;
  *reinterpret_cast<mn_u4_t*>(W + e + 8) = hi;
}
}  // namespace

extern "C" int mn_quant_int8_rows(const uint16_t* W, int64_t ldw, uint8_t* Wq, int64_t ldq, float* scale, int64_t n_rows, int K, void* stream) {
  MN_CHECK_ARG(W && Wq && scale && n_rows >= 1 && n_rows < ((int64_t)1 << 31) && K >= 4 && (K % 4) == 0 && (ldw % 4) == 0 && (ldq % 4) == 0 &&
                   (((uintptr_t)W) & 7) == 0 && (((uintptr_t)Wq) & 3) == 0,
               "mn_quant_int8_rows: bad args (K, ldw, ldq multiples of 4)");
  hipLaunchKernelGGL(quant_int8_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, mn_stream(stream), W, ldw, Wq, ldq, scale, K);
  MN_CHECK_LAUNCH("mn_quant_int8_rows");
  return MN_OK;
}

extern "C" int mn_dequant_int8_rows(const uint8_t* Wq, int64_t ldq, const float* scale, uint16_t* W, int64_t ldw, int64_t n_rows, int K,
                                    void* stream) {
  MN_CHECK_ARG(W && Wq && scale && n_rows >= 1 && n_rows < ((int64_t)1 << 31) && K >= 4 && (K % 4) == 0 && (ldw % 4) == 0 && (ldq % 4) == 0 &&
                   (((uintptr_t)W) & 7) == 0 && (((uintptr_t)Wq) & 3) == 0,
               "mn_dequant_int8_rows: bad args (K, ldw, ldq multiples of 4)");
  if (ldq == K && ldw == K && (K % 16) == 0 && ((((uintptr_t)W) | ((uintptr_t)Wq)) & 15) == 0) {      // contiguous rows: 16 weights per thread
    const int64_t np = n_rows * (K / 16);
    hipLaunchKernelGGL(dequant8_flat_kernel<true>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, mn_stream(stream), Wq, scale, W, np, K);
  } else
  hipLaunchKernelGGL(dequant_int8_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, mn_stream(stream), Wq, ldq, scale, W, ldw, K);
  MN_CHECK_LAUNCH("mn_dequant_int8_rows");
  return MN_OK;
}

extern "C" int mn_quant_fp8_rows(const uint16_t* W, int64_t ldw, uint8_t* Wq, int64_t ldq, float* scale, int64_t n_rows, int K, void* stream) {
  MN_CHECK_ARG(W && Wq && scale && n_rows >= 1 && n_rows < ((int64_t)1 << 31) && K >= 4 && (K % 4) == 0 && (ldw % 4) == 0 && (ldq % 4) == 0 &&
                   (((uintptr_t)W) & 7) == 0 && (((uintptr_t)Wq) & 3) == 0,
               "mn_quant_fp8_rows: bad args (K, ldw, ldq multiples of 4)");
  hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, mn_stream(stream), W, ldw, Wq, ldq, scale, K);
  MN_CHECK_LAUNCH("mn_quant_fp8_rows");
  return MN_OK;
}

extern "C" int mn_dequant_fp8_rows(const uint8_t* Wq, int64_t ldq, const float* scale, uint16_t* W, int64_t ldw, int64_t n_rows, int K,
                                   void* stream) {
  MN_CHECK_ARG(W && Wq && scale && n_rows >= 1 && n_rows < ((int64_t)1 << 31) && K >= 4 && (K % 4) == 0 && (ldw % 4) == 0 && (ldq % 4) == 0 &&
                   (((uintptr_t)W) & 7) == 0 && (((uintptr_t)Wq) & 3) == 0,
               "mn_dequant_fp8_rows: bad args (K, ldw, ldq multiples of 4)");
  if (ldq == K && ldw == K && (K % 16) == 0 && ((((uintptr_t)W) | ((uintptr_t)Wq)) & 15) == 0) {      // contiguous rows: 16 weights per thread
    const int64_t np = n_rows * (K / 16);
    hipLaunchKernelGGL(dequant8_flat_kernel<false>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, mn_stream(stream), Wq, scale, W, np, K);
  } else
  hipLaunchKernelGGL(dequant_fp8_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, mn_stream(stream), Wq, ldq, scale, W, ldw, K);
  MN_CHECK_LAUNCH("mn_dequant_fp8_rows");
  return MN_OK;
}

// NF4: W bf16 [n_rows][K] (row stride ldw elements) -> Wq [n_rows][K / 2] bytes (row stride ldq BYTES) + absmax fp32 [n_rows][K / 64].
extern "C" int mn_quant_nf4_rows(const uint16_t* W, int64_t ldw, uint8_t* Wq, int64_t ldq, float* absmax, int64_t n_rows, int K, void* stream) {
  MN_CHECK_ARG(W && Wq && absmax && n_rows >= 1 && n_rows < ((int64_t)1 << 31) && K >= 64 && (K % 64) == 0 && (ldw % 8) == 0 && (ldq % 4) == 0 &&
                   (((uintptr_t)W) & 15) == 0 && (((uintptr_t)Wq) & 3) == 0,
               "mn_quant_nf4_rows: bad args (K a multiple of the 64-element block, 16-byte aligned rows)");
  hipLaunchKernelGGL(quant_nf4_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, mn_stream(stream), W, ldw, Wq, ldq, absmax, K);
  MN_CHECK_LAUNCH("mn_quant_nf4_rows");
  return MN_OK;
}

extern "C" int mn_dequant_nf4_rows(const uint8_t* Wq, int64_t ldq, const float* absmax, uint16_t* W, int64_t ldw, int64_t n_rows, int K,
                                   void* stream) {
  MN_CHECK_ARG(W && Wq && absmax && n_rows >= 1 && n_rows < ((int64_t)1 << 31) && K >= 64 && (K % 64) == 0 && (ldw % 8) == 0 && (ldq % 4) == 0 &&
                   (((uintptr_t)W) & 15) == 0 && (((uintptr_t)Wq) & 3) == 0,
               "mn_dequant_nf4_rows: bad args (K a multiple of the 64-element block, 16-byte aligned rows)");
  hipLaunchKernelGGL(dequant_nf4_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, mn_stream(stream), Wq, ldq, absmax, W, ldw, K);
  MN_CHECK_LAUNCH("mn_dequant_nf4_rows");
  return MN_OK;
}

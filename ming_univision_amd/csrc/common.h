// common.h — shared device helpers for libmingnative (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "mingnative.h"
#ifdef MN_DEV_HOOKS
#include "mingnative_dev.h"
#endif

typedef uint16_t bf16_t;

#define MN_WAVE 64

// ---- error plumbing ---------------------------------------------------------------------
void mn_set_error(const char* fmt, ...);
#define MN_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      mn_set_error(__VA_ARGS__);           \
      return MN_EINVAL;                    \
    }                                      \
  } while (0)
#define MN_CHECK_LAUNCH(what)                                            \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      mn_set_error("%s: %s", what, hipGetErrorString(e__));              \
      return MN_ELAUNCH;                                                 \
    }                                                                    \
  } while (0)

// ---- bf16 <-> f32 -------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
// round-to-nearest-even, NaN preserved (matches torch .to(bfloat16))
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// hardware conversion (v_cvt_pk_bf16_f32, round-to-nearest-even: bit-identical to f32_to_bf16 on finite values)
typedef float mn_f2_t __attribute__((ext_vector_type(2)));
typedef __bf16 mn_b2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(mn_f2_t{lo, hi}, mn_b2_t));
}
// (a, b) -> packed bf16 hi parts and packed bf16 lo parts: a = hi.a + lo.a to 2^-17
#ifdef MN_EMUL_F16F8
// Measurement only (never in the product or the dev library: `make dev EXTRA=-DMN_EMUL_F16F8=3` builds a one-off copy): the operand
// an fp16 hi pass + an fp8 lo pass would multiply — x16 = fp16(x) plus the residual rounded to MN_EMUL_F16F8 mantissa bits (3 = e4m3)
// — carried through the existing bf16 hi/lo pair (16 bits hold it to 2^-18).  DESIGN.md §9-1, tests/measure/f16f8_error.py.
__device__ __forceinline__ float emul_f16f8(float a) {
  const float a16 = (float)(_Float16)a;
  uint32_t u = __builtin_bit_cast(uint32_t, a - a16);
  constexpr int drop = 23 - MN_EMUL_F16F8;
  u = (u + (1u << (drop - 1))) & ~((1u << drop) - 1u);
  return a16 + __builtin_bit_cast(float, u);
}
#endif
__device__ __forceinline__ void split_pk_bf16(float a, float b, uint32_t& hi, uint32_t& lo) {
#ifdef MN_EMUL_F16F8
  a = emul_f16f8(a); b = emul_f16f8(b);
#endif
  hi = cvt_pk_bf16(a, b);
  lo = cvt_pk_bf16(a - bf16lo_to_f32(hi), b - bf16hi_to_f32(hi));
}

// ---- fp8 weights (OCP e4m3fn) -----------------------------------------------------------------------
// Four e4m3 bytes of one dword -> four bf16 (two packed dwords), exact: every e4m3 value is a bf16 value.  v_cvt_scalef32_pk_bf16_fp8
// converts the two bytes of the selected 16-bit half in one instruction; the scale operand stays 1.0 (row scales are applied to the
// fp32 accumulators, so any fp32 scale is allowed).
typedef uint32_t mn_u2_t __attribute__((ext_vector_type(2)));
typedef uint32_t mn_u4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mn_u2_t fp8x4_to_bf16(uint32_t q) {
  return mn_u2_t{__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, 1.0f, false)),
                 __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, 1.0f, true))};
}
// eight e4m3 bytes (k ascending) -> eight bf16 = one MFMA fragment / one 16-byte LDS slot
__device__ __forceinline__ mn_u4_t fp8x8_to_bf16(uint32_t q0, uint32_t q1) {
  const mn_u2_t a = fp8x4_to_bf16(q0), b = fp8x4_to_bf16(q1);
  return mn_u4_t{a.x, a.y, b.x, b.y};
}

// ---- activations ----------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---- wave / block reductions -------------------------------------------------------------
// Cross-lane reductions on DPP operands: `__shfl_xor` compiles to ds_bpermute_b32 here — one trip through the LDS crossbar per step
// (64 of them in the GQA decode-attention kernel) — where a VALU instruction with a DPP-permuted operand does the same for nothing.
// A row of 16 lanes is reduced by four symmetric pairings (row_mirror, row_half_mirror, quad_perm [2,3,0,1], quad_perm [1,0,3,2]):
// every lane of the row ends with the bitwise same value.  The four rows of a wave are combined from one lane each (v_readlane),
// in a fixed order, so the whole wave holds one value.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0x140>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0xB1>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0x140>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0xB1>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
// Block-wide sum; `red` is LDS scratch of >= blockDim.x/64 floats. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}


static inline hipStream_t mn_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t mn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }


// ---- internal argument block of the wide-row GEMM (gemm256.hip), shared with the composites in engine.hip ----
typedef struct mn_g256 {
  const bf16_t* A; int64_t lda; int64_t a_lo_off;       // a_lo_off (elements) != 0: hi/lo-stacked rows
  const bf16_t* W; int64_t ldw; int64_t w_pair_rows;    // w_pair_rows != 0: gate/up pairing (SWIGLU_SPLIT), N counts hidden units
  const bf16_t* bias;                                    // bf16 [N] (paired: gate bias at n, up bias at n + w_pair_rows) or NULL
  void* C; int64_t ldc; int64_t c_zstride;               // split-K: slice z writes C + z * c_zstride (F32 epilogue)
  int64_t c_lo_off;                                      // SWIGLU_SPLIT: lo rows c_lo_off elements after the hi rows
  const float* gate; int64_t ldgate;                     // F32_RESID_GATE: C += gate * (acc + bias)
  int M, N, K, Kc;                                       // Kc is set by the launcher (k per split-K slice)
  // grouped form: group g = blockIdx.z owns rows [g_off[g], g_off[g] + g_cnt[g]) (device arrays), weights W + g * w_gstride;
  // a_rows (optional) maps a row position to its source row in A (gather); M bounds every group's row count
  const int32_t* g_off; const int32_t* g_cnt; int64_t w_gstride; const int32_t* a_rows; int n_groups;
  // optional device-built list of the LIVE row tiles of the grouped form (mn_moe_sort_tiles): tile t = rows
  // [tile_m0[t], +128 or 256) of group tile_g[t], *n_tiles entries, at most max_mtiles; the grid then has no empty groups
  const int32_t* tile_g; const int32_t* tile_m0; const int32_t* n_tiles; int max_mtiles;
  int group_m;                                           // set by the launcher: M-tiles per band of the tile order (0: tm fastest over all M-tiles)
  int thin;                                              // set by the launcher: tile lists skip the MFMAs / fragment reads of M-fragments without a live row
  // fp8-MFMA regime (f8 != 0; labelled reduced arithmetic, never the default): A and W point to OCP e4m3 BYTES (lda / ldw / K count
  // bytes = elements, K % 128 == 0), a_scale [M] / w_scale [N or 2N] are the operands' fp32 row scales (the accumulators are multiplied
  // by a_scale[m] * w_scale[n] before bias / epilogue); no hi/lo rows, epilogues F32 (split-K allowed) and SWIGLU_BF16
  // grouped form: group g's weight scales at w_scale + g * w_sstride; a_scale is indexed by the SOURCE row (through a_rows when given)
  int f8; const float* a_scale; const float* w_scale; int64_t w_sstride;
} mn_g256;
enum { MN_G256_F32 = 0, MN_G256_BF16 = 1, MN_G256_BF16_GELU = 2, MN_G256_F32_RESID = 3, MN_G256_SWIGLU_SPLIT = 4,
       MN_G256_F32_RESID_GATE = 5, MN_G256_SWIGLU_BF16 = 6 };
extern "C" int mn_gemm256_ex(const mn_g256* a, int epi, int ksplit, void* stream);
extern "C" int mn_gemm256_slices(int K, int ksplit);

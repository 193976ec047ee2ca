// w8_codec.h — the 8-bit weight codecs of the weight-only modes (mingnative.h section 7): bytes -> bf16 / fp32, exact.
//   MN_W_FP8_E4M3  OCP e4m3fn bytes: v_cvt_scalef32_pk_bf16_fp8 / v_cvt_pk_f32_fp8 (every e4m3 value is a bf16 value)
//   MN_W_INT8      two's-complement bytes in [-127, 127]: sign-extend + v_cvt_f32_i32 (+ v_cvt_pk_bf16_f32: |q| <= 127 is 7 bits)
// The format is a wave-uniform kernel argument: one scalar branch per converted dword group, the kernels are otherwise the same.
#pragma once
#include "common.h"

__device__ __forceinline__ mn_u2_t i8x4_to_bf16(uint32_t q) {
  const int s = (int)q;
  const float f0 = (float)((s << 24) >> 24), f1 = (float)((s << 16) >> 24), f2 = (float)((s << 8) >> 24), f3 = (float)(s >> 24);
  return mn_u2_t{cvt_pk_bf16(f0, f1), cvt_pk_bf16(f2, f3)};
}
// eight weight bytes (k ascending) -> eight bf16 = one MFMA fragment / one 16-byte LDS slot.  I8 is a compile-time choice: the
// callers branch ONCE per parked chunk on the (wave-uniform) format and run a specialised loop (a per-dword branch cost 3 % of an
// fp8 launch)
template <bool I8>
__device__ __forceinline__ mn_u4_t w8x8_to_bf16(uint32_t q0, uint32_t q1) {
  if constexpr (I8) {
    const mn_u2_t a = i8x4_to_bf16(q0), b = i8x4_to_bf16(q1);
    return mn_u4_t{a.x, a.y, b.x, b.y};
  } else {
    return fp8x8_to_bf16(q0, q1);
  }
}
// four weight bytes -> four fp32
template <bool I8>
__device__ __forceinline__ void w8x4_to_f32(uint32_t q, float (&o)[4]) {
  if constexpr (I8) {
    const int s = (int)q;
    o[0] = (float)((s << 24) >> 24); o[1] = (float)((s << 16) >> 24); o[2] = (float)((s << 8) >> 24); o[3] = (float)(s >> 24);
  } else {
    const mn_f2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(q, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(q, true);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
  }
}

// w8_codec.h — the weight codecs of the weight-only modes (mingnative.h section 7): codes -> bf16 / fp32.
//   MN_W_FP8_E4M3  OCP e4m3fn bytes: v_cvt_scalef32_pk_bf16_fp8 / v_cvt_pk_f32_fp8 (every e4m3 value is a bf16 value); the power-of-two
//                  row scale multiplies the fp32 accumulators afterwards
//   MN_W_INT8      optimum-quanto's qint8 rule: W'[n, k] = bf16_rne(q[n, k] * scale[n]) with a bf16-valued scale = amax / 127 — the product
//                  is rounded PER ELEMENT (quanto multiplies scale * weights in bf16 before the matmul), so the row scale rides the
//                  conversion (v_cvt_f32_ubyteN of the byte + 128, one fma, v_cvt_pk_bf16_f32) and nothing is scaled afterwards
//   MN_W_NF4       below
// The format is a wave-uniform kernel argument: one scalar branch per converted chunk, the kernels are otherwise the same.
#pragma once
#include "common.h"

// four int8 bytes q -> bf16_rne(q * s), two packed dwords.  (u - 128) * s as ONE fma of the biased byte: s * u - 128 s is exact in fp32
// before its single rounding (|q| <= 128 is 8 bits, s has 8 significant bits), so this is fp32(q * s) exactly, then bf16 rounding.
__device__ __forceinline__ mn_u2_t i8x4_to_bf16(uint32_t q, float s) {
  const uint32_t u = q ^ 0x80808080u;
  const float b = -128.0f * s;
  const float f0 = fmaf((float)(u & 0xffu), s, b), f1 = fmaf((float)((u >> 8) & 0xffu), s, b);
  const float f2 = fmaf((float)((u >> 16) & 0xffu), s, b), f3 = fmaf((float)(u >> 24), s, b);
  return mn_u2_t{cvt_pk_bf16(f0, f1), cvt_pk_bf16(f2, f3)};
}
// eight weight bytes (k ascending) -> eight bf16 = one MFMA fragment / one 16-byte LDS slot.  I8 is a compile-time choice: the
// callers branch ONCE per parked chunk on the (wave-uniform) format and run a specialised loop (a per-dword branch cost 3 % of an
// fp8 launch).  s: the row's scale (int8 only)
template <bool I8>
__device__ __forceinline__ mn_u4_t w8x8_to_bf16(uint32_t q0, uint32_t q1, float s) {
  if constexpr (I8) {
    const mn_u2_t a = i8x4_to_bf16(q0, s), b = i8x4_to_bf16(q1, s);
    return mn_u4_t{a.x, a.y, b.x, b.y};
  } else {
    return fp8x8_to_bf16(q0, q1);
  }
}
// four e4m3 bytes -> four fp32 (the one-row fp32-FMA kernel of the e4m3 mode, skinny_w8.hip)
__device__ __forceinline__ void fp8x4_to_f32(uint32_t q, float (&o)[4]) {
  const mn_f2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(q, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(q, true);
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}

// ---- NF4 (MN_W_NF4): bitsandbytes' 4-bit NormalFloat, blockwise absmax (block = 64 consecutive k of a row), two codes per byte ----
// W'[n, k] = bf16_rne(NF4[code] * absmax[n, k / 64])   (kDequantizeBlockwise with a bf16 compute dtype; oracle/int4_ref.py)
// The product must be rounded PER ELEMENT before it meets the MFMA, and absmax changes every 64 k, so the decode is a table lookup:
// the 16 possible values of a block, bf16(NF4[i] * absmax), are built once per (lane, block) as two BYTE PLANES (low bytes, high
// bytes: 4 + 4 dwords) and every group of four codes is looked up with v_perm_b32: one perm per plane and table half, a bit-field
// select on code bit 3, two perms to interleave the planes back into bf16 pairs — 13 VALU per 4 codes, no LDS, no multiply per element.
// Byte layout of a row (K / 2 bytes): inside every dword the nibbles are, from bit 0 up, e0 e4 e1 e5 e2 e6 e3 e7 of its eight
// consecutive k, so `x & 0x0f0f0f0f` / `(x >> 4) & 0x0f0f0f0f` are four codes, one per byte (oracle/int4_ref.pack_kernel).
#define MN_NF4_TABLE                                                                                                                  \
  {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f, -0.28444138169288635f, -0.18477343022823334f,            \
   -0.09105003625154495f, 0.0f, 0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,               \
   0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f}

struct Nf4Tab { uint32_t lo[4], hi[4]; };          // byte planes of the block's 16 bf16 values: lo[g] = low bytes of entries 4g .. 4g + 3

__device__ __forceinline__ Nf4Tab nf4_table(float absmax) {
  constexpr float T[16] = MN_NF4_TABLE;
  Nf4Tab t;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint32_t p01 = cvt_pk_bf16(T[4 * g] * absmax, T[4 * g + 1] * absmax), p23 = cvt_pk_bf16(T[4 * g + 2] * absmax, T[4 * g + 3] * absmax);
    t.lo[g] = __builtin_amdgcn_perm(p23, p01, 0x06040200u);     // v_perm_b32: selector bytes 0-3 pick from the 2nd operand, 4-7 from the 1st
    t.hi[g] = __builtin_amdgcn_perm(p23, p01, 0x07050301u);
  }
  return t;
}
// four codes (one per byte of n, 0..15) -> four bf16 as two packed dwords (v0, v1), (v2, v3)
__device__ __forceinline__ void nf4_lut4(const Nf4Tab& t, uint32_t n, uint32_t& o01, uint32_t& o23) {
  const uint32_t sel = n & 0x07070707u, b = n & 0x08080808u;
  const uint32_t m = (b << 5) - (b >> 3);                        // 0xff in every byte whose code has bit 3 set (mod 2^32: exact per byte)
  const uint32_t la = __builtin_amdgcn_perm(t.lo[1], t.lo[0], sel), lb = __builtin_amdgcn_perm(t.lo[3], t.lo[2], sel);
  const uint32_t ha = __builtin_amdgcn_perm(t.hi[1], t.hi[0], sel), hb = __builtin_amdgcn_perm(t.hi[3], t.hi[2], sel);
  const uint32_t lo = (lb & m) | (la & ~m), hi = (hb & m) | (ha & ~m);      // v_bfi_b32
  o01 = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
  o23 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
}
// one dword = eight codes of eight consecutive k -> eight bf16 = one MFMA fragment / one 16-byte LDS slot
__device__ __forceinline__ mn_u4_t nf4x8_to_bf16(const Nf4Tab& t, uint32_t x) {
  uint32_t a, b, c, d;
  nf4_lut4(t, x & 0x0f0f0f0fu, a, b);
  nf4_lut4(t, (x >> 4) & 0x0f0f0f0fu, c, d);
  return mn_u4_t{a, b, c, d};
}

// layout_ops.hip — the layout shuffles of the MingTok ViT as single HIP passes: each one reads its source once and
// writes the consumer's operand directly, instead of a strided view materialised by a copy and then cast / added / clamped by further
// passes.  HBM-bound index arithmetic; results are bit-identical to the view + copy they replace.
//
//   mn_patchify_operand    PatchEmbed's conv k = s = P as a GEMM (layers/patch_embed.py:69-82): image [B,3,Hi,Wi] -> the GEMM's A operand
//                          [B * N, 3 P^2] (row = patch (gy, gx), column = (c, py, px)), cast to bf16 or split into bf16 hi / lo rows
//   mn_tokens_assemble     prepare_tokens (vision_transformer.py:218-223): patch tokens + cls token appended LAST + position embedding
//   mn_subtoken_rearrange  forward_pixel_decoder's "b (h w) (x y c) -> b (h x w y) c" (modeling_mingtok.py:184-188)
//   mn_unpatchify_clamp    unpatchify 'nhwpqc->nchpwq' (vision_transformer.py:515-527) + clamp_(-1, 1) (modeling_mingtok.py:195)
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// one thread = 4 consecutive px of one (patch, c, py): a 16-byte image load, an 8-byte store per half
__global__ __launch_bounds__(256) void patchify_operand_kernel(const float* __restrict__ img, int B, int Hi, int Wi, int P,
                                                               bf16_t* __restrict__ Y, int64_t y_lo_off) {
  const int gw = Wi / P, gh = Hi / P, q = P / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * gh * gw * 3 * P * q;
  if (i >= total) return;
  const int px4 = (int)(i % q);
  int64_t r = i / q;
  const int py = (int)(r % P); r /= P;
  const int c = (int)(r % 3); r /= 3;
  const int gx = (int)(r % gw); r /= gw;
  const int gy = (int)(r % gh);
  const int b = (int)(r / gh);
  const f4 v = *reinterpret_cast<const f4*>(img + (((int64_t)b * 3 + c) * Hi + (gy * P + py)) * Wi + gx * P + px4 * 4);
  const int64_t row = ((int64_t)b * gh + gy) * gw + gx;
  bf16_t* yr = Y + row * (3 * P * P) + (c * P + py) * P + px4 * 4;
  if (y_lo_off) {
    uint32_t h0, l0, h1, l1;
    split_pk_bf16(v.x, v.y, h0, l0);
    split_pk_bf16(v.z, v.w, h1, l1);
    *reinterpret_cast<u2*>(yr) = u2{h0, h1};
    *reinterpret_cast<u2*>(yr + y_lo_off) = u2{l0, l1};
  } else {
    *reinterpret_cast<u2*>(yr) = u2{cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w)};
  }
}

// out[b, n, :] = (n < N ? tok[b * N + n, :] : cls) + pos[n, :]
__global__ __launch_bounds__(256) void tokens_assemble_kernel(const float* __restrict__ tok, const bf16_t* __restrict__ cls,
                                                              const float* __restrict__ pos, float* __restrict__ out, int B, int N, int D) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (int64_t)B * (N + 1) * D) return;
  const int d = (int)(i % D);
  const int64_t r = i / D;
  const int n = (int)(r % (N + 1)), b = (int)(r / (N + 1));
  f4 v;
  if (n < N) v = *reinterpret_cast<const f4*>(tok + ((int64_t)b * N + n) * D + d);
  else v = f4{bf16_to_f32(cls[d]), bf16_to_f32(cls[d + 1]), bf16_to_f32(cls[d + 2]), bf16_to_f32(cls[d + 3])};
  *reinterpret_cast<f4*>(out + i) = v + *reinterpret_cast<const f4*>(pos + (int64_t)n * D + d);
}

// y [B, h, w, r, r, Dp] -> x [B, h, r, w, r, Dp]
__global__ __launch_bounds__(256) void subtoken_rearrange_kernel(const float* __restrict__ y, float* __restrict__ x, int B, int h, int w, int r,
                                                                 int Dp) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (int64_t)B * h * w * r * r * Dp) return;
  const int c = (int)(i % Dp);
  int64_t t = i / Dp;                                   // output row index over (b, hy, sx, wx, sy)
  const int sy = (int)(t % r); t /= r;
  const int wx = (int)(t % w); t /= w;
  const int sx = (int)(t % r); t /= r;
  const int hy = (int)(t % h);
  const int b = (int)(t / h);
  const int64_t src = (((((int64_t)b * h + hy) * w + wx) * r + sx) * r + sy) * Dp + c;
  *reinterpret_cast<f4*>(x + i) = *reinterpret_cast<const f4*>(y + src);
}

// o [B, hh, ww, p, q, 3] -> img [B, 3, hh * p, ww * q] clamped to [lo, hi]; one thread = one output pixel quad along q
__global__ __launch_bounds__(256) void unpatchify_clamp_kernel(const float* __restrict__ o, float* __restrict__ img, int B, int hh, int ww, int p,
                                                               float lo, float hi) {
  const int Wd = ww * p, Hd = hh * p;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (int64_t)B * 3 * Hd * Wd) return;
  const int X = (int)(i % Wd);
  int64_t t = i / Wd;
  const int Y = (int)(t % Hd); t /= Hd;
  const int c = (int)(t % 3);
  const int b = (int)(t / 3);
  const int gy = Y / p, py = Y % p, gx = X / p, q0 = X % p;      // p % 4 == 0: the quad stays inside one patch row
  const float* s = o + (((((int64_t)b * hh + gy) * ww + gx) * p + py) * p + q0) * 3 + c;
  f4 v = {s[0], s[3], s[6], s[9]};
  v = f4{fminf(fmaxf(v.x, lo), hi), fminf(fmaxf(v.y, lo), hi), fminf(fmaxf(v.z, lo), hi), fminf(fmaxf(v.w, lo), hi)};
  *reinterpret_cast<f4*>(img + i) = v;
}

}  // namespace

extern "C" int mn_patchify_operand(const float* image, int B, int Hi, int Wi, int P, uint16_t* Y, int64_t y_lo_off, void* stream) {
  MN_CHECK_ARG(image && Y && B >= 1 && P >= 4 && (P % 4) == 0 && Hi >= P && Wi >= P && (Hi % P) == 0 && (Wi % P) == 0 && (y_lo_off % 4) == 0 &&
                   (((uintptr_t)image) & 15) == 0 && (((uintptr_t)Y) & 7) == 0,
               "mn_patchify_operand: bad args (P %% 4 == 0, image sides multiples of P)");
  const int64_t total = (int64_t)B * (Hi / P) * (Wi / P) * 3 * P * (P / 4);
  hipLaunchKernelGGL(patchify_operand_kernel, dim3((unsigned)mn_cdiv(total, 256)), dim3(256), 0, mn_stream(stream), image, B, Hi, Wi, P, Y, y_lo_off);
  MN_CHECK_LAUNCH("mn_patchify_operand");
  return MN_OK;
}

extern "C" int mn_tokens_assemble(const float* tok, const uint16_t* cls, const float* pos, float* out, int B, int N, int D, void* stream) {
  MN_CHECK_ARG(tok && cls && pos && out && B >= 1 && N >= 1 && D >= 4 && (D % 4) == 0, "mn_tokens_assemble: bad args (D %% 4 == 0)");
  const int64_t total = (int64_t)B * (N + 1) * D / 4;
  hipLaunchKernelGGL(tokens_assemble_kernel, dim3((unsigned)mn_cdiv(total, 256)), dim3(256), 0, mn_stream(stream), tok, cls, pos, out, B, N, D);
  MN_CHECK_LAUNCH("mn_tokens_assemble");
  return MN_OK;
}

extern "C" int mn_subtoken_rearrange(const float* y, float* x, int B, int h, int w, int r, int Dp, void* stream) {
  MN_CHECK_ARG(y && x && y != x && B >= 1 && h >= 1 && w >= 1 && r >= 1 && Dp >= 4 && (Dp % 4) == 0, "mn_subtoken_rearrange: bad args (Dp %% 4 == 0)");
  const int64_t total = (int64_t)B * h * w * r * r * Dp / 4;
  hipLaunchKernelGGL(subtoken_rearrange_kernel, dim3((unsigned)mn_cdiv(total, 256)), dim3(256), 0, mn_stream(stream), y, x, B, h, w, r, Dp);
  MN_CHECK_LAUNCH("mn_subtoken_rearrange");
  return MN_OK;
}

extern "C" int mn_unpatchify_clamp(const float* o, float* image, int B, int hh, int ww, int p, float lo, float hi, void* stream) {
  MN_CHECK_ARG(o && image && B >= 1 && hh >= 1 && ww >= 1 && p >= 4 && (p % 4) == 0, "mn_unpatchify_clamp: bad args (p %% 4 == 0)");
  const int64_t total = (int64_t)B * 3 * hh * p * ww * p / 4;
  hipLaunchKernelGGL(unpatchify_clamp_kernel, dim3((unsigned)mn_cdiv(total, 256)), dim3(256), 0, mn_stream(stream), o, image, B, hh, ww, p, lo, hi);
  MN_CHECK_LAUNCH("mn_unpatchify_clamp");
  return MN_OK;
}

// skinny_w8.hip — the one-row weight-streaming kernel on fp8 weights (mingnative.h section 7): the expert pair launches of a
// decode step with 1 or 2 rows in fp8 weight mode (text decode; the 2-row CFG step of one image).
//
//   out[b][n] = epilogue( sum_seg rs[seg][n] * sum_k x[b][seg][k] * e4m3(Wq[e(b, seg)][n, k]) )
//
// Same decomposition as skinny_gemm.hip's fp32-FMA kernel with ONE activation row per batch entry: a block stages x into LDS as
// fp32 once, its waves walk row groups of R output rows, every lane owns 16 consecutive k of each 1024-wide chunk (one 16-byte
// nontemporal load = 16 e4m3 weights; a wave covers 1 KiB of one weight row per instruction), converts them with v_cvt_pk_f32_fp8
// (exact) and multiplies in fp32 — the result is that of the dequantised weights to fp32 rounding.  Batch entries differ in their
// weight matrix (w_index: one (row, expert) pair each); K-segments sum the experts of one row (seg_index), each segment with its OWN
// row scales: the per-lane partial sums are folded into the total with the segment's scale at every segment end (the scales of a
// row group sit one per lane and are fetched by v_readlane).  Only what the expert launches use is built: prologue NONE (+ the
// router weight as seg_scale), epilogues NONE / SWIGLU / RESID.
#include <stdlib.h>

#include "skinny_device.h"
#include "w8_codec.h"

namespace {

// LDS image of x: position c*1024 + j*256 + lane*4 + i holds x[c*1024 + lane*16 + j*4 + i]  (four conflict-free ds_read_b128 per chunk)
__device__ __forceinline__ int perm_k16(int k) { return (k & ~1023) | (((k >> 2) & 3) << 8) | (((k >> 4) & 63) << 2) | (k & 3); }

struct W8Args {
  mn_skinny_args a;
  int32_t nchunk;      // chunks of 1024 k per segment
  int32_t nseg, batch;
  int32_t inv_nchunk;  // ceil(65536 / nchunk)
};

__device__ __forceinline__ void fma16(const u32x4 q, const float* xl, float& acc) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
  float t = acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(xl + j * 256);
    float wv[4];
    fp8x4_to_f32(w[j], wv);                        // e4m3 bytes -> fp32, exact (w8_codec.h)
    t = fmaf(wv[0], x.x, t); t = fmaf(wv[1], x.y, t); t = fmaf(wv[2], x.z, t); t = fmaf(wv[3], x.w, t);
  }
  acc = t;
}

template <int R, int SW, int NT>
__global__ __launch_bounds__(NT) void skinny_w8_kernel(const W8Args ka) {
  constexpr int RING = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const mn_skinny_args& a = ka.a;
  const int nchunk = ka.nchunk, Kp = nchunk << 10, nseg = ka.nseg, K = a.K, N = a.N;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * (NT / 64);
  const int ngroups = (N + R - 1) / R;
  const int nct = nseg * nchunk;
  const uint8_t* Wq = reinterpret_cast<const uint8_t*>(a.w);
  const int wsel = a.w_index ? a.w_index[b] : b;
  const int32_t* segi = a.seg_index ? a.seg_index + (int64_t)b * nseg : nullptr;
  // per-segment weight / scale offsets, one per lane (lane sg holds segment sg's), fetched by v_readlane in the issue path
  int64_t my_segoff = 0, my_scoff = 0;
  if (segi && lane < nseg) { my_segoff = (int64_t)segi[lane] * a.seg_w_stride; my_scoff = (int64_t)segi[lane] * a.wscale_seg_stride; }
  const int seg_lo = (int)(my_segoff & 0xffffffff), seg_hi = (int)(my_segoff >> 32);
  const int kmax = K - 16;
  const uint8_t* wbase = Wq + (int64_t)wsel * a.w_batch_stride;
  const float* sbase = a.wscale + (int64_t)wsel * a.wscale_batch_stride;

  u32x4 ring[RING][SW][R];
  auto issue = [&](int g, int ct, u32x4 (&dst)[SW][R]) {
    int c = ct;
    int64_t so = 0;
    if (nseg > 1) {
      const int sg = (ct * ka.inv_nchunk) >> 16;
      c = ct - sg * nchunk;
      so = ((int64_t)__builtin_amdgcn_readlane(seg_hi, sg) << 32) | (uint32_t)__builtin_amdgcn_readlane(seg_lo, sg);
    }
    const uint8_t* wp = wbase + so + min(c * 1024 + lane * 16, kmax);
#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = min(g * R + r, N - 1) + s * N;
        dst[s][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (int64_t)n * a.ldw));
      }
  };
  auto issue_head = [&](int g) {
#pragma unroll
    for (int d = 0; d < RING; ++d)
      if (d < nct) issue(g, d, ring[d]);
  };
  // row scales of a group: lane (sg * SW + s) * R + r holds the scale of row n0 + r (+ s N) of segment sg's matrix
  auto load_scales = [&](int g) -> float {
    const int l = lane, r = l % R, s = (l / R) % SW, sg = l / (R * SW);
    if (sg >= nseg) return 0.f;
    const int64_t so = segi ? (int64_t)segi[sg] * a.wscale_seg_stride : 0;
    return sbase[so + min(g * R + r, N - 1) + s * N];
  };

  int g = blockIdx.x * (NT / 64) + wave;
  float rs = 0.f;
  if (g < ngroups) { issue_head(g); rs = load_scales(g); }
  // ---- stage x (prologue NONE; the router weight of a segment rides seg_scale)
  {
    const float* xb = a.x + (int64_t)(b / (a.x_batch_div > 0 ? a.x_batch_div : 1)) * a.x_batch_stride;
    for (int s = 0; s < nseg; ++s) {
      const float sc = a.seg_scale ? a.seg_scale[(int64_t)b * nseg + s] : 1.0f;
      for (int k = threadIdx.x; k < Kp; k += NT) smem[(int64_t)s * Kp + perm_k16(k)] = k < K ? xb[(int64_t)s * K + k] * sc : 0.f;
    }
  }
  __syncthreads();

  const float* xl = smem + lane * 4;
  for (; g < ngroups; g += nwaves) {
    const int n0 = g * R;
    float acc[SW][R], tot[SW][R];
#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r) { acc[s][r] = 0.f; tot[s][r] = 0.f; }
    int seg_left = nchunk, seg_id = 0;                    // chunks left in the current segment
    auto consume = [&](int ct, const u32x4 (&src)[SW][R]) {
      const float* xp = xl + (int64_t)ct * 1024;          // LDS image is [nseg * nchunk * 1024]
#pragma unroll
      for (int s = 0; s < SW; ++s)
#pragma unroll
        for (int r = 0; r < R; ++r) fma16(src[s][r], xp, acc[s][r]);
      if (--seg_left == 0) {                              // segment done: fold its partial sums with ITS row scales
#pragma unroll
        for (int s = 0; s < SW; ++s)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            tot[s][r] = fmaf(lane_f(rs, (seg_id * SW + s) * R + r), acc[s][r], tot[s][r]);
            acc[s][r] = 0.f;
          }
        seg_left = nchunk; ++seg_id;
      }
    };
    int c0 = 0;
    for (; c0 + 2 * RING <= nct; c0 += RING) {
#pragma unroll
      for (int d = 0; d < RING; ++d) {
        consume(c0 + d, ring[d]);
        issue(g, c0 + d + RING, ring[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      if (c0 + d < nct) {
        consume(c0 + d, ring[d]);
        if (c0 + d + RING < nct) issue(g, c0 + d + RING, ring[d]);
      }
    }
    c0 += RING;
#pragma unroll
    for (int d = 0; d < RING; ++d)
      if (c0 + d < nct) consume(c0 + d, ring[d]);
    float rs_next = 0.f;
    if (g + nwaves < ngroups) { issue_head(g + nwaves); rs_next = load_scales(g + nwaves); }

#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r) tot[s][r] = wave_sum(tot[s][r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (lane == r) {
        const int n = n0 + r;
        if (n < N) {
          float y = tot[0][r];
          if (a.bias) y += bf16_to_f32(a.bias[n]);
          float* o = a.out + (int64_t)b * a.out_batch_stride + n;
          switch (a.epilogue) {
            case MN_EPI_SWIGLU: {
              float y2 = tot[SW - 1][r];
              if (a.bias) y2 += bf16_to_f32(a.bias[n + N]);
              y = silu_f(y) * y2;
            } break;
            case MN_EPI_RESID: y += a.res[(int64_t)b * a.res_batch_stride + n]; break;
            default: break;
          }
          *o = y;
        }
      }
    }
    rs = rs_next;
  }
}

template <int R, int SW, int NT>
void launch_one(const W8Args& ka, dim3 grid, size_t lds, hipStream_t st) {
  static bool opted = false;
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_w8_kernel<R, SW, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    opted = true;
  }
  hipLaunchKernelGGL((skinny_w8_kernel<R, SW, NT>), grid, dim3(NT), lds, st, ka);
}

template <int R, int SW>
void launch_nt(const W8Args& ka, int nt, dim3 grid, size_t lds, hipStream_t st) {
  if (nt == 256) launch_one<R, SW, 256>(ka, grid, lds, st);
  else if (nt == 768) launch_one<R, SW, 768>(ka, grid, lds, st);
  else if (nt == 1024) launch_one<R, SW, 1024>(ka, grid, lds, st);
  else launch_one<R, SW, 512>(ka, grid, lds, st);
}

}  // namespace

// One activation row per batch entry on fp8 weights (args as mn_skinny_gemm with wfmt = MN_W_FP8_E4M3, M == 1).  Called by
// mn_skinny_gemm for the batch / segment forms, which the matrix-core route does not take.
extern "C" int mn_skinny_w8_row(const mn_skinny_args* args, void* stream) {
  MN_CHECK_ARG(args != nullptr, "mn_skinny_gemm(fp8 row): null args");
  W8Args ka;
  ka.a = *args;
  const mn_skinny_args& a = ka.a;
  MN_CHECK_ARG(a.M == 1 && a.wfmt == MN_W_FP8_E4M3 && a.wscale && a.x && a.w && a.out, "mn_skinny_gemm(fp8 row): M == 1, e4m3 weights + row scales");
  MN_CHECK_ARG(a.N >= 1 && a.K >= 16 && (a.K % 16) == 0 && (a.ldw % 16) == 0 && (((uintptr_t)a.w) & 15) == 0 &&
                   ((a.w_batch_stride | a.seg_w_stride) % 16) == 0,
               "mn_skinny_gemm(fp8 row): K, ldw and the matrix strides must be multiples of 16");
  MN_CHECK_ARG(a.prologue == MN_PRO_NONE && (a.epilogue == MN_EPI_NONE || a.epilogue == MN_EPI_SWIGLU || a.epilogue == MN_EPI_RESID),
               "mn_skinny_gemm(fp8 row): prologue NONE, epilogue NONE / SWIGLU / RESID only");
  MN_CHECK_ARG(a.epilogue != MN_EPI_RESID || a.res, "mn_skinny_gemm(fp8 row): RESID needs res");
  ka.nseg = a.nseg > 0 ? a.nseg : 1;
  ka.batch = a.batch > 0 ? a.batch : 1;
  const int sw = a.epilogue == MN_EPI_SWIGLU ? 2 : 1;
  ka.nchunk = (a.K + 1023) / 1024;
  ka.inv_nchunk = (65536 + ka.nchunk - 1) / ka.nchunk;
  MN_CHECK_ARG((int64_t)ka.nseg * ka.nchunk < 4096 && ka.nseg * sw * 4 <= 64, "mn_skinny_gemm(fp8 row): too many K chunks / segments");
  const size_t lds = (size_t)ka.nseg * ka.nchunk * 1024 * sizeof(float);
  MN_CHECK_ARG(lds <= 160 * 1024, "mn_skinny_gemm(fp8 row): K = %d x nseg = %d too large for LDS", a.K, ka.nseg);
  const int cus = mn_num_cus();
  int bpc = 1, nt = 512;
  const int64_t groups_r1 = (int64_t)a.N * ka.batch;
  const int max_bpc = (int)((160 * 1024) / lds);
  if (groups_r1 < (int64_t)cus * 8 && max_bpc >= 2) { nt = 256; bpc = max_bpc > 4 ? 4 : max_bpc; }
  else nt = groups_r1 >= (int64_t)cus * 16 ? 1024 : 768;      // more waves per CU shorten the launch (skinny_gemm.hip, one-row plan)
  const int wpb = nt / 64;
  const int64_t resident = mn_cdiv((int64_t)cus * bpc * wpb, ka.batch);
  int R = 1;
  if (sw == 1) { if (a.N >= 8 * resident) R = 4; else if (a.N >= 4 * resident) R = 2; }
  else if (a.N >= 16 * resident) R = 2;
  const int ngroups = (a.N + R - 1) / R;
  int64_t gx = mn_cdiv(ngroups, wpb);
  const int64_t cap = ((int64_t)cus * bpc) / ka.batch;       // one round over the CUs, rounded DOWN (skinny_gemm.hip)
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, (unsigned)ka.batch);
  hipStream_t st = mn_stream(stream);
  if (sw == 2) { if (R >= 2) launch_nt<2, 2>(ka, nt, grid, lds, st); else launch_nt<1, 2>(ka, nt, grid, lds, st); }
  else if (R >= 4) launch_nt<4, 1>(ka, nt, grid, lds, st);
  else if (R >= 2) launch_nt<2, 1>(ka, nt, grid, lds, st);
  else launch_nt<1, 1>(ka, nt, grid, lds, st);
  MN_CHECK_LAUNCH("mn_skinny_gemm(fp8 row)");
  return MN_OK;
}

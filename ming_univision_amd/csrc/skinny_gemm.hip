// skinny_gemm.hip — weight-streaming GEMM for M <= 8 activation rows (gfx950).
//
//   out[m, n] = epilogue( sum_k prologue(x)[m, k] * W[n, k] + bias[n] )
//
// Roofline: HBM. Every weight byte is read exactly once per launch (non-temporal 16-byte loads,
// one wave covers 1 KiB of one weight row per instruction), activations are tiny and live in LDS
// as fp32 (so the result is fp32-accurate against the fp32 oracle; only the weights are bf16).
//
// Work decomposition: a block stages prologue(x) into LDS once (fused RMSNorm / LayerNorm /
// adaLN-modulate / SiLU), then its waves walk "row groups" of R consecutive output rows in a
// grid-stride loop.  Within a row group each lane owns 8 consecutive k of every 512-wide chunk,
// accumulates R x M partial sums in registers and the wave reduces them with xor shuffles.
// The LDS image of x is permuted so that the two ds_read_b128 per (row, chunk) are
// lane-contiguous (conflict-free): position c*512 + j*256 + lane*4 + i holds x[c*512 + lane*8 + j*4 + i].
#include <stdlib.h>

#include "skinny_device.h"

namespace {

// 16-byte weight loads kept in flight per lane per (row, swiglu half). The 512-thread plan runs one
// block per CU (2 waves per SIMD, so each wave may use up to 256 VGPRs) and keeps a deeper ring.
template <int NT> struct RingDepth { static constexpr int value = 4; };

// SW: number of weight row sets per output row (2 for SWIGLU, else 1).
//
// Per wave the weight stream is software-pipelined through a RING-deep register ring: the first
// RING chunks of the wave's first row group are requested BEFORE the block stages x (so the HBM
// latency of the first weights overlaps the prologue), and the first RING chunks of the next row
// group are requested before the current group's cross-lane reduction and epilogue.
template <int M, int R, int SW, int NT>
__global__ __launch_bounds__(NT) void skinny_kernel(const KArgs ka) {
  constexpr int RING = RingDepth<NT>::value;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const mn_skinny_args& a = ka.a;
  const int nchunk = ka.nchunk, Kp = nchunk << 9, nseg = ka.nseg, K = a.K, N = a.N;
  float* xs = smem;
  float* red = smem + (int64_t)M * nseg * Kp;
  const int b = blockIdx.y;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * (NT / 64);
  const int ngroups = (N + R - 1) / R;
  const int nct = nseg * nchunk;               // chunks per row (all segments)
  const int64_t xstride = (int64_t)nseg * Kp;  // LDS row stride (floats)
  const int wsel = a.w_index ? a.w_index[b] : b;
  const int32_t* segi = a.seg_index ? a.seg_index + (int64_t)b * nseg : nullptr;

  // per-segment weight offsets live one per lane (lane sg holds segment sg's offset); a uniform
  // readlane fetches them without touching memory in the issue path
  int64_t my_segoff = 0;
  if (segi && lane < nseg) my_segoff = (int64_t)segi[lane] * a.seg_w_stride;
  const int seg_lo = (int)(my_segoff & 0xffffffff), seg_hi = (int)(my_segoff >> 32);
  const int kmax = K - 8;

  u32x4 ring[RING][SW][R];
  // Branch-free request of chunk ct of row group g.  Lanes past the end of a row (partial last chunk)
  // and rows past N re-read valid in-bounds data; the matching x entries in LDS are zero / unused.
  auto issue = [&](int g, int ct, u32x4 (&dst)[SW][R]) {
    int c = ct;
    int64_t so = 0;
    if (nseg > 1) {
      const int sg = (ct * ka.inv_nchunk) >> 16;
      c = ct - sg * nchunk;
      so = ((int64_t)__builtin_amdgcn_readlane(seg_hi, sg) << 32) | (uint32_t)__builtin_amdgcn_readlane(seg_lo, sg);
    }
    const bf16_t* wp = a.w + (int64_t)wsel * a.w_batch_stride + so + min(c * 512 + lane * 8, kmax);
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

  int g = blockIdx.x * (NT / 64) + wave;
  if (g < ngroups) issue_head(g);
  stage_x<M, NT>(ka, xs, red, b);
  __syncthreads();

  const float* xl = xs + lane * 4;
  for (; g < ngroups; g += nwaves) {
    const int n0 = g * R;
    float acc[SW][R][M];
#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[s][r][m] = 0.f;

    auto consume = [&](int ct, const u32x4 (&src)[SW][R]) {
      const float* xp = xl + (int64_t)ct * 512;   // LDS image is [M][nseg * nchunk * 512]
#pragma unroll
      for (int s = 0; s < SW; ++s)
#pragma unroll
        for (int r = 0; r < R; ++r) fma_chunk<M>(src[s][r], xp, xstride, acc[s][r]);
    };
    int c0 = 0;
    // steady state: every slot is consumed and immediately refilled RING chunks ahead (no branches)
    for (; c0 + 2 * RING <= nct; c0 += RING) {
#pragma unroll
      for (int d = 0; d < RING; ++d) {
        consume(c0 + d, ring[d]);
        issue(g, c0 + d + RING, ring[d]);
      }
    }
    // drain: at most 2*RING - 1 chunks left, of which the first RING are already in the ring
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
    if (g + nwaves < ngroups) issue_head(g + nwaves);   // next group's head overlaps the reduction below

    // wave reduction; afterwards every lane holds every sum
#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[s][r][m] = wave_sum(acc[s][r][m]);

    // epilogue: lane (r*M + m) stores element (m, n0 + r)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (lane == r * M + m) {
          const int n = n0 + r;
          if (n < N) {
            float y = acc[0][r][m];
            if (a.bias) y += bf16_to_f32(a.bias[n]);
            float* o = a.out + (int64_t)b * a.out_batch_stride + (int64_t)m * a.ldo + n;
            switch (a.epilogue) {
              case MN_EPI_SILU: y = silu_f(y); break;
              case MN_EPI_GELU: y = gelu_erf_f(y); break;
              case MN_EPI_SWIGLU: {
                float y2 = acc[SW - 1][r][m];
                if (a.bias) y2 += bf16_to_f32(a.bias[n + N]);
                y = silu_f(y) * y2;
              } break;
              case MN_EPI_RESID:
                y += a.res[(int64_t)b * a.res_batch_stride + (int64_t)m * a.ldres + n];
                break;
              case MN_EPI_RESID_GATE:
                y = a.res[(int64_t)b * a.res_batch_stride + (int64_t)m * a.ldres + n] +
                    a.gate[(int64_t)m * a.ldgate + n] * y;
                break;
              default: break;
            }
            *o = y;
          }
        }
      }
  }
}

// Dynamic LDS above 64 KiB needs an explicit opt-in, once per kernel.
template <int M, int R, int SW, int NT>
void launch_one(const KArgs& ka, dim3 grid, size_t lds, hipStream_t st) {
  static bool opted = false;
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_kernel<M, R, SW, NT>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    opted = true;
  }
  hipLaunchKernelGGL((skinny_kernel<M, R, SW, NT>), grid, dim3(NT), lds, st, ka);
}

template <int M, int R, int SW>
int launch_nt(const KArgs& ka, int nt, dim3 grid, size_t lds, hipStream_t st) {
  if (nt == 256) launch_one<M, R, SW, 256>(ka, grid, lds, st);
  else if (nt == 768) launch_one<M, R, SW, 768>(ka, grid, lds, st);
  else if (nt == 1024) {
    // the 16-wave form exists for one activation row and <= 2 weight rows per group only; the planner below never asks for another
    if constexpr (M == 1 && R <= 2) launch_one<1, R, SW, 1024>(ka, grid, lds, st);
    else return -1;
  } else launch_one<M, R, SW, 512>(ka, grid, lds, st);
  return 0;
}

template <int M>
int launch_m(const KArgs& ka, int R, int sw, int nt, dim3 grid, size_t lds, hipStream_t st) {
  if (sw == 2) {
    if (R >= 2) return launch_nt<M, 2, 2>(ka, nt, grid, lds, st);
    return launch_nt<M, 1, 2>(ka, nt, grid, lds, st);
  }
  if (R >= 4) return launch_nt<M, 4, 1>(ka, nt, grid, lds, st);
  if (R >= 2) return launch_nt<M, 2, 1>(ka, nt, grid, lds, st);
  return launch_nt<M, 1, 1>(ka, nt, grid, lds, st);
}

}  // namespace

// ===========================================================================================
// 5 <= M <= 64 rows (batched generation): weights are still read once, but the fp32 FMA path would be
// VALU-bound, so the product runs on the matrix cores:
//   prologue kernel : x' = prologue(x) in fp32, split into bf16 hi + lo rows  -> Y[2M, K]
//   streaming kernel : partials[z][M][Ntot] = (Y_hi + Y_lo)[:, kz] W[:, kz]^T   (stream_mfma.hip <= 32 rows, stream_kloop.hip above)
//   epilogue kernel : out = epilogue( sum_z partials + bias )
// hi + lo keeps the products fp32-accurate (x' = hi + lo to 2^-17 relative).
// ===========================================================================================
extern "C" int mn_stream_mfma(const uint16_t* Y, const uint16_t* W, float* P, int M, int Ntot, int K, void* stream);
extern "C" int mn_stream_mfma_slices(int M, int Ntot, int K);
extern "C" int mn_stream_mfma_wq(const uint16_t* Y, const uint8_t* Wq, const float* wscale, float* P, int M, int Ntot, int K, int wfmt, void* stream);
extern "C" int mn_stream_mfma_w8_slices(int M, int Ntot, int K);
extern "C" int mn_stream_mfma_wq_slices(int wfmt, int M, int Ntot, int K);
extern "C" int mn_skinny_w8_row(const mn_skinny_args* args, void* stream);      // skinny_w8.hip

namespace {

__global__ __launch_bounds__(1024) void medium_prologue_kernel(const mn_skinny_args a, bf16_t* __restrict__ Y) {
  __shared__ float red[32];
  const int m = blockIdx.x, tid = threadIdx.x, K = a.K, M = a.M;
  const float* xr = a.x + (int64_t)m * a.ldx;
  const int pro = a.prologue;
  float mean = 0.f, rstd = 1.f;
  if (pro >= MN_PRO_RMSNORM) {
    if (pro != MN_PRO_RMSNORM) {
      float s = 0.f;
      for (int k = tid; k < K; k += 1024) s += xr[k];
      mean = block_sum(s, red) / (float)K;
    }
    float ss = 0.f;
    for (int k = tid; k < K; k += 1024) { const float d = xr[k] - mean; ss += d * d; }
    ss = block_sum(ss, red);
    rstd = rsqrtf(ss / (float)K + a.eps);
  }
  for (int k = tid; k < K; k += 1024) {
    float v = xr[k];
    if (pro == MN_PRO_ADD_SILU) v += a.pro_a[(int64_t)m * a.ld_pro_a + k];
    if (pro == MN_PRO_SILU || pro == MN_PRO_ADD_SILU) v = silu_f(v);
    if (pro >= MN_PRO_RMSNORM) {
      v = (v - mean) * rstd;
      if (a.ln_g) v *= bf16_to_f32(a.ln_g[k]);
      if (a.ln_b && pro != MN_PRO_RMSNORM) v += bf16_to_f32(a.ln_b[k]);
      if (pro == MN_PRO_LN_MOD) v = v * (1.0f + a.pro_b[(int64_t)m * a.ld_pro_b + k]) + a.pro_a[(int64_t)m * a.ld_pro_a + k];
    }
    const bf16_t hi = f32_to_bf16(v);
    Y[(int64_t)m * K + k] = hi;
    Y[(int64_t)(M + m) * K + k] = f32_to_bf16(v - bf16_to_f32(hi));
  }
}

__global__ __launch_bounds__(256) void medium_epilogue_kernel(const mn_skinny_args a, const float* __restrict__ P, int nz,
                                                              int Ntot) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int M = a.M, N = a.N;
  if (i >= (int64_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  const int64_t slab = (int64_t)M * Ntot;
  auto gather = [&](int col) {
    float s = 0.f;
    for (int z0 = 0; z0 < nz; z0 += 8) {            // eight slabs' loads in flight
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = z0 + j < nz ? P[(int64_t)(z0 + j) * slab + (int64_t)m * Ntot + col] : 0.f;
      for (int j = 0; j < 8; ++j) s += t[j];                    // (the one-by-one loop's order: same bits)
    }
    return s;
  };
  float y = gather(n);
  if (a.bias) y += bf16_to_f32(a.bias[n]);
  switch (a.epilogue) {
    case MN_EPI_SILU: y = silu_f(y); break;
    case MN_EPI_GELU: y = gelu_erf_f(y); break;
    case MN_EPI_SWIGLU: {
      float y2 = gather(n + N);
      if (a.bias) y2 += bf16_to_f32(a.bias[n + N]);
      y = silu_f(y) * y2;
    } break;
    case MN_EPI_RESID: y += a.res[(int64_t)m * a.ldres + n]; break;
    case MN_EPI_RESID_GATE: y = a.res[(int64_t)m * a.ldres + n] + a.gate[(int64_t)m * a.ldgate + n] * y; break;
    default: break;
  }
  a.out[(int64_t)m * a.ldo + n] = y;
}

int medium_ksplit(int Ntot, int K) {
  const int tiles = (Ntot + 127) / 128;
  int s = (1024 + tiles - 1) / tiles;          // ~4 workgroups per CU
  const int smax = K / 128 > 0 ? K / 128 : 1;  // at least two 64-wide k-steps per slice
  if (s > smax) s = smax;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

}  // namespace

// Rows at or above this count take the MFMA route when the caller provides scratch.  Measured end to end on the 16B-A3B
// 512^2 workload (same box, tokens/s): 2 rows 76.8 vs 70.6 with the fp32-FMA kernels, 3 rows 74.2 vs 56.1, 4 rows 136 vs 97;
// one row (text decode) stays on the fp32-FMA kernel.
constexpr int MEDIUM_MIN_M = 2;

extern "C" size_t mn_skinny_workspace_bytes(int M, int N, int K, int epilogue) {
  if (M < MEDIUM_MIN_M) return 0;
  const int Ntot = epilogue == MN_EPI_SWIGLU ? 2 * N : N;
  const size_t y = ((size_t)2 * M * K * sizeof(bf16_t) + 255) & ~(size_t)255;
  return y + (size_t)mn_stream_mfma_slices(M, Ntot, K) * M * Ntot * sizeof(float) + 256;
}

// quantised weights (wfmt != 0): every row count takes the matrix-core route (the fp32-FMA kernel reads bf16 rows)
extern "C" size_t mn_skinny_workspace_bytes_wq(int wfmt, int M, int N, int K, int epilogue) {
  if (wfmt == MN_W_BF16) return mn_skinny_workspace_bytes(M, N, K, epilogue);
  const int Ntot = epilogue == MN_EPI_SWIGLU ? 2 * N : N;
  const size_t y = ((size_t)2 * M * K * sizeof(bf16_t) + 255) & ~(size_t)255;
  return y + (size_t)mn_stream_mfma_wq_slices(wfmt, M, Ntot, K) * M * Ntot * sizeof(float) + 256;
}
extern "C" size_t mn_skinny_workspace_bytes_w8(int M, int N, int K, int epilogue) {
  return mn_skinny_workspace_bytes_wq(MN_W_FP8_E4M3, M, N, K, epilogue);
}

static int skinny_medium(const mn_skinny_args& a, void* stream) {
  MN_CHECK_ARG(a.M <= 64, "mn_skinny_gemm: M=%d out of range [1,64]", a.M);
  MN_CHECK_ARG((a.batch <= 1) && (a.nseg <= 1), "mn_skinny_gemm: batch/nseg forms need M <= 8");
  MN_CHECK_ARG(a.ldw == a.K, "mn_skinny_gemm: M > 8 needs densely packed weights (ldw == K)");
  const int Ntot = a.epilogue == MN_EPI_SWIGLU ? 2 * a.N : a.N;
  const bool w8 = a.wfmt != MN_W_BF16;
  MN_CHECK_ARG(!w8 || (a.wscale && (a.K % 16) == 0), "mn_skinny_gemm: fp8 weights need wscale and K %% 16 == 0");
  MN_CHECK_ARG(a.wfmt != MN_W_NF4 || (a.K % 64) == 0, "mn_skinny_gemm: NF4 weights need K %% 64 == 0");
  const size_t need = mn_skinny_workspace_bytes_wq(a.wfmt, a.M, a.N, a.K, a.epilogue);
  if (!a.ws || a.ws_bytes < need) { mn_set_error("mn_skinny_gemm: M=%d needs %zu workspace bytes", a.M, need); return MN_ENOSPACE; }
  bf16_t* Y = reinterpret_cast<bf16_t*>(a.ws);
  float* P = reinterpret_cast<float*>(reinterpret_cast<char*>(a.ws) + (((size_t)2 * a.M * a.K * sizeof(bf16_t) + 255) & ~(size_t)255));
  hipStream_t st = mn_stream(stream);
  hipLaunchKernelGGL(medium_prologue_kernel, dim3(a.M), dim3(1024), 0, st, a, Y);
  const int nz = w8 ? mn_stream_mfma_wq(Y, reinterpret_cast<const uint8_t*>(a.w), a.wscale, P, a.M, Ntot, a.K, a.wfmt, stream)
                    : mn_stream_mfma(Y, a.w, P, a.M, Ntot, a.K, stream);
  if (nz < 0) return nz;
  hipLaunchKernelGGL(medium_epilogue_kernel, dim3((unsigned)mn_cdiv((int64_t)a.M * a.N, 256)), dim3(256), 0, st, a, P, nz, Ntot);
  MN_CHECK_LAUNCH("mn_skinny_gemm(medium)");
  return MN_OK;
}

// Tuning overrides for micro-benchmarks (0 = heuristic). Not part of the stable ABI.
static struct { int R, nt, bpc; } g_tune = {0, 0, 0};
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_skinny_tune(int R, int nt, int bpc) { g_tune.R = R; g_tune.nt = nt; g_tune.bpc = bpc; }
#endif

extern "C" int mn_skinny_gemm(const mn_skinny_args* args, void* stream) {
  MN_CHECK_ARG(args != nullptr, "mn_skinny_gemm: null args");
  KArgs ka;
  ka.a = *args;
  mn_skinny_args& a = ka.a;
  MN_CHECK_ARG(a.M >= 1 && a.M <= 64, "mn_skinny_gemm: M=%d out of range [1,64]", a.M);
  MN_CHECK_ARG(a.N >= 1 && a.K >= 8 && (a.K % 8) == 0, "mn_skinny_gemm: bad N=%d K=%d (K %% 8 must be 0)", a.N, a.K);
  MN_CHECK_ARG((a.ldw % 8) == 0, "mn_skinny_gemm: ldw=%lld must be a multiple of 8", (long long)a.ldw);
  MN_CHECK_ARG(a.x && a.w && a.out, "mn_skinny_gemm: null pointer");
  MN_CHECK_ARG(a.prologue >= 0 && a.prologue <= MN_PRO_LN_MOD, "mn_skinny_gemm: bad prologue %d", a.prologue);
  MN_CHECK_ARG(a.epilogue >= 0 && a.epilogue <= MN_EPI_RESID_GATE, "mn_skinny_gemm: bad epilogue %d", a.epilogue);
  ka.nseg = a.nseg > 0 ? a.nseg : 1;
  ka.batch = a.batch > 0 ? a.batch : 1;
  MN_CHECK_ARG(a.wfmt == MN_W_BF16 || a.wfmt == MN_W_FP8_E4M3 || a.wfmt == MN_W_INT8 || a.wfmt == MN_W_NF4, "mn_skinny_gemm: bad wfmt %d", a.wfmt);
  MN_CHECK_ARG((a.wfmt != MN_W_NF4 && a.wfmt != MN_W_INT8) || (ka.batch == 1 && ka.nseg == 1 && a.ws),
               "mn_skinny_gemm: NF4 / int8 weights run the workspace route only (no batch / segment forms: their products are rounded per element)");
  if (a.wfmt != MN_W_BF16) {
    // one row per batch entry with a plain prologue: the one-row fp8 kernel (expert pair launches, batch / segment forms included);
    // everything else: the matrix-core route (checks dense weights / no batch forms / workspace)
    const bool row_form = a.M == 1 && a.prologue == MN_PRO_NONE &&
                          (a.epilogue == MN_EPI_NONE || a.epilogue == MN_EPI_SWIGLU || a.epilogue == MN_EPI_RESID);
    if (row_form && (ka.batch > 1 || ka.nseg > 1 || a.ws == nullptr)) return mn_skinny_w8_row(&a, stream);
    return skinny_medium(a, stream);
  }
  if (a.M >= MEDIUM_MIN_M && ka.batch == 1 && ka.nseg == 1 && a.ldw == a.K && a.ws != nullptr) return skinny_medium(a, stream);
  MN_CHECK_ARG(a.M <= 8, "mn_skinny_gemm: M=%d > 8 needs the workspace route (ws, dense weights, no batch/nseg)", a.M);
  MN_CHECK_ARG(ka.nseg == 1 || a.prologue <= MN_PRO_ADD_SILU, "mn_skinny_gemm: normalising prologue with segments");
  MN_CHECK_ARG(a.prologue != MN_PRO_ADD_SILU || a.pro_a, "mn_skinny_gemm: ADD_SILU needs pro_a");
  MN_CHECK_ARG(a.prologue != MN_PRO_LN_MOD || (a.pro_a && a.pro_b), "mn_skinny_gemm: LN_MOD needs shift/scale");
  MN_CHECK_ARG(a.prologue != MN_PRO_RMSNORM || a.ln_g, "mn_skinny_gemm: RMSNORM needs ln_g");
  MN_CHECK_ARG((a.epilogue != MN_EPI_RESID && a.epilogue != MN_EPI_RESID_GATE) || a.res, "mn_skinny_gemm: RESID needs res");
  MN_CHECK_ARG(a.epilogue != MN_EPI_RESID_GATE || a.gate, "mn_skinny_gemm: RESID_GATE needs gate");
  MN_CHECK_ARG((((uintptr_t)a.w) & 15) == 0 && ((a.w_batch_stride | a.seg_w_stride) % 8) == 0,
               "mn_skinny_gemm: weights must be 16-byte aligned");
  ka.nchunk = (a.K + 511) / 512;
  ka.inv_nchunk = (65536 + ka.nchunk - 1) / ka.nchunk;
  MN_CHECK_ARG((int64_t)ka.nseg * ka.nchunk < 4096 && ka.nseg <= 64, "mn_skinny_gemm: too many K chunks / segments");
  const int64_t Kp = (int64_t)ka.nchunk * 512;
  const size_t lds = ((size_t)a.M * ka.nseg * Kp + 32) * sizeof(float);
  if (lds > 160 * 1024 || (a.M > 4 && a.M < 8)) {
    // x does not fit the 160 KiB LDS (or M has no dedicated instantiation: 5..7 run as 4 + rest):
    // run the rows in slices (weights are re-streamed per slice)
    int mc = (int)((160 * 1024 / sizeof(float) - 32) / ((size_t)ka.nseg * Kp));
    if (a.M > 4 && a.M < 8 && mc > 4) mc = 4;
    MN_CHECK_ARG(mc >= 1, "mn_skinny_gemm: K=%d x nseg=%d too large for LDS", a.K, ka.nseg);
    for (int m0 = 0; m0 < a.M; m0 += mc) {
      mn_skinny_args s = *args;
      s.M = (a.M - m0) < mc ? (a.M - m0) : mc;
      s.x = a.x + (int64_t)m0 * a.ldx;
      s.out = a.out + (int64_t)m0 * a.ldo;
      if (a.res) s.res = a.res + (int64_t)m0 * a.ldres;
      if (a.gate) s.gate = a.gate + (int64_t)m0 * a.ldgate;
      if (a.pro_a) s.pro_a = a.pro_a + (int64_t)m0 * a.ld_pro_a;
      if (a.pro_b) s.pro_b = a.pro_b + (int64_t)m0 * a.ld_pro_b;
      const int rc = mn_skinny_gemm(&s, stream);
      if (rc != MN_OK) return rc;
    }
    return MN_OK;
  }

  const int sw = a.epilogue == MN_EPI_SWIGLU ? 2 : 1;
  // Occupancy plan: ~16 waves per CU (each keeps RING x SW x R KiB of weights in flight), blocks per CU
  // limited by the LDS image of x; a wave should own >= 2 row groups where N allows so that the
  // per-block prologue is amortised and the next group's loads overlap the current reduction.
  const int cus = mn_num_cus();
  // Plan A (default): one 512-thread block per CU — the per-block prologue (x staging, LayerNorm
  // statistics, modulation) is paid once per CU and hidden behind an 8-deep weight ring.
  // Plan B: 256-thread blocks, up to 4 per CU, when there are too few row groups to feed plan A.
  int bpc = 1, nt = 512;
  {
    const int64_t groups_r1 = (int64_t)a.N * ka.batch;
    const int max_bpc = (int)((160 * 1024) / lds);
    if (groups_r1 < (int64_t)cus * 8 && max_bpc >= 2) { nt = 256; bpc = max_bpc > 4 ? 4 : max_bpc; }
  }
  if (nt == 512 && g_tune.nt == 0) {
    // 12 waves per CU balance better than 8 when the row groups per CU are a multiple of 12 but not of 8
    // (e.g. N = 3072 on 256 CUs: 12 groups per CU)
    const int64_t per_cu = mn_cdiv((int64_t)a.N * ka.batch, cus);
    if (per_cu % 12 == 0 && per_cu % 8 != 0) nt = 768;
    // one activation row (the expert pair launches of a 1- / 2-row step, text decode's projections): a wave's chain of dependent
    // chunk round trips is the launch, so more waves per CU shorten it — 16 where every wave still gets a row group, else 12
    // (tools/exp/moe_pair_tune.py: both expert launches 36.2 -> 33.9 us per layer at 1 row, 53.2 -> 49.4 at 2)
    else if (a.M == 1) nt = (int64_t)a.N * ka.batch >= (int64_t)cus * 16 ? 1024 : 768;
  }
  if (g_tune.nt == 256 || g_tune.nt == 512 || g_tune.nt == 768 || (g_tune.nt == 1024 && a.M == 1)) {
    nt = g_tune.nt;
    bpc = g_tune.bpc > 0 ? g_tune.bpc : 1;
    const int max_bpc = (int)((160 * 1024) / lds);
    if (bpc > max_bpc) bpc = max_bpc;
    if (bpc * nt > 2048) bpc = 2048 / nt;
  }
  const int waves_per_block = nt / 64;
  const int64_t resident_waves = mn_cdiv((int64_t)cus * bpc * waves_per_block, ka.batch);
  int R = 1;
  if (sw == 1) {
    if ((int64_t)a.N >= 8 * resident_waves) R = 4;
    else if ((int64_t)a.N >= 4 * resident_waves) R = 2;
  } else {
    if ((int64_t)a.N >= 16 * resident_waves) R = 2;   // SwiGLU already keeps 2 rows per group in flight
  }
  if (g_tune.R > 0) R = g_tune.R;
  if (R * a.M > 64) R = 64 / a.M;
  if (sw == 2 && R > 2) R = 2;
  if (R == 3) R = 2;
  // 16 waves per workgroup are instantiated for <= 2 weight rows per group only (a plan made for 16 waves used to fall through to the
  // 8-wave kernel when N was large enough for R = 4, e.g. the one-row lm_head logits GEMV): such problems keep 16 waves at R = 2
  if (nt == 1024 && R > 2) R = 2;
  const int ngroups = (a.N + R - 1) / R;
  int64_t gx = mn_cdiv(ngroups, waves_per_block);
  // all batch entries' workgroups in ONE round over the CUs: rounded DOWN (24 expert pairs of a 3-row step: 10 x 24 = 240 workgroups;
  // 11 x 24 = 264 sent eight of them into a second round of the whole launch: 3.20 -> 2.7x ms per decoder step)
  const int64_t cap = ((int64_t)cus * bpc) / ka.batch;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, (unsigned)ka.batch);
  hipStream_t st = mn_stream(stream);
  int planned = 0;
  switch (a.M) {
    case 1: planned = launch_m<1>(ka, R, sw, nt, grid, lds, st); break;
    case 2: planned = launch_m<2>(ka, R, sw, nt, grid, lds, st); break;
    case 3: planned = launch_m<3>(ka, R, sw, nt, grid, lds, st); break;
    case 4: planned = launch_m<4>(ka, R, sw, nt, grid, lds, st); break;
    default: planned = launch_m<8>(ka, R, sw, nt, grid, lds, st); break;
  }
  MN_CHECK_ARG(planned == 0, "mn_skinny_gemm: no kernel for the plan (M=%d R=%d nt=%d)", a.M, R, nt);
  MN_CHECK_LAUNCH("mn_skinny_gemm");
  return MN_OK;
}

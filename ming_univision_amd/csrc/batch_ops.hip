// batch_ops.hip — batched (many-row) operators on the bf16 MFMA path (gfx950):
//   mn_gemm_bf16          C = A W^T (+bias, GELU, fp32 residual accumulate), 128x128x64 tiles,
//                         v_mfma_f32_16x16x32_bf16, double-buffered LDS filled by global_load_lds_dwordx4
//                         (register staging for ragged K), XOR swizzle on the 16-byte slots (conflict-free
//                         ds_read_b128 fragments).  874 TFLOP/s at 4096^3 (672 register-staged).
//   mn_attn_prefill_hd64  flash attention, head_dim 64, S^T = K Q^T / O^T = V^T P^T formulation so
//                         that the softmax probabilities never leave registers
//   mn_layernorm_bf16, mn_swiglu_bf16, fp32<->bf16 converters
#include <string.h>

#include "common.h"
#include "wide_glue.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ===========================================================================================
// GEMM
// ===========================================================================================
namespace {
constexpr int BN = 128, BK = 64;  // BM is a template parameter (64 for skinny-M problems, else 128)

// LDS tile: [128 rows][64 k] bf16 = 128 B per row = 8 slots of 16 B; slot index XOR (row & 7).
__device__ __forceinline__ int tile_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

// Grouped form (MoE experts): blockIdx.y = group g works on rows [g_off[g], g_off[g] + g_cnt[g]) of A / C with
// the weights W + g * w_gstride; M is then the upper bound used to size the grid (tiles beyond g_cnt exit).
struct GemmGroups {
  const int32_t* off;
  const int32_t* cnt;
  int64_t w_gstride;
};

// GLDS: the k-tiles are copied global -> LDS by global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass): a wave
// instruction lands lane-linear (8 rows x 8 slots = 1 KiB), so the XOR swizzle is applied to the per-lane SOURCE address
// and again on the fragment read.  Needs whole k-tiles (K range a multiple of 64); out-of-range rows read a clamped row
// (their products are never stored).
template <int EPI, int BM, bool GROUPED = false, bool GLDS = false>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A_, int64_t lda,
                                                        const bf16_t* __restrict__ W, int64_t ldw,
                                                        const bf16_t* __restrict__ bias, void* __restrict__ Cv0,
                                                        int64_t ldc, int M, int N, int K, int Kc, int64_t c_zstride,
                                                        GemmGroups grp = GemmGroups{nullptr, nullptr, 0},
                                                        int64_t a_lo_off = 0) {
  constexpr int MI = BM / 32;   // 16-row MFMA tiles per wave along M (wave grid is 2 x 2)
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;
  __shared__ __attribute__((aligned(16))) char lds_raw[2][A_BYTES + W_BYTES];  // [buf][A | W]
  // split-K: blockIdx.y owns k in [kbeg, kend) and writes its own fp32 partial slab (F32 epilogue only)
  const int kbeg = GROUPED ? 0 : blockIdx.y * Kc, kend = GROUPED ? K : min(K, kbeg + Kc);
  void* Cv = (!GROUPED && (EPI == MN_GEMM_F32 || EPI == MN_GEMM_F32_RESID))
                 ? (void*)(reinterpret_cast<float*>(Cv0) + (int64_t)blockIdx.y * c_zstride) : Cv0;
  if (GROUPED) {
    const int g = blockIdx.y;
    const int row0 = grp.off[g];
    M = grp.cnt[g];
    A_ += (int64_t)row0 * lda;
    W += (int64_t)g * grp.w_gstride;
    Cv = (EPI == MN_GEMM_F32 || EPI == MN_GEMM_F32_RESID) ? (void*)(reinterpret_cast<float*>(Cv0) + (int64_t)row0 * ldc)
                                                           : (void*)(reinterpret_cast<bf16_t*>(Cv0) + (int64_t)row0 * ldc);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: consecutive tiles of one XCD share the A row panel
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int nt = tiles_n * tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nt / 8, r = nt % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (GROUPED && m0 >= M) return;   // this group has fewer rows than the grid's upper bound (uniform per block)
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * 64;

  // global -> register staging: each thread moves 4 slots of A and 4 slots of W per k-tile
  // thread t: row = t/8 + 32*i, slot = t%8
  const int ld_row = tid >> 3, ld_slot = tid & 7;
  u32x4 ra[4], rw[4];
  // a_lo_off != 0: A is a bf16 hi/lo pair (lo rows a_lo_off elements after the hi rows); the K loop then runs twice over
  // the same W tiles, second pass on the lo rows, so that C = (A_hi + A_lo) W^T comes out of one launch
  const int nk1 = (kend - kbeg + BK - 1) / BK;
  auto gload = [&](int kt) {
    const bf16_t* A = kt >= nk1 ? A_ + a_lo_off : A_;
    if (kt >= nk1) kt -= nk1;
    const int k = kbeg + kt * BK + ld_slot * 8;
    const bool kok = k < kend;  // K % 8 == 0 so a slot is all-in or all-out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = ld_row + 32 * i;
      const u32x4 z = {0u, 0u, 0u, 0u};
      if (i < MI) ra[i] = (kok && (m0 + r) < M) ? *reinterpret_cast<const u32x4*>(A + (int64_t)(m0 + r) * lda + k) : z;
      rw[i] = (kok && (n0 + r) < N) ? *reinterpret_cast<const u32x4*>(W + (int64_t)(n0 + r) * ldw + k) : z;
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = ld_row + 32 * i;
      if (i < MI) *reinterpret_cast<u32x4*>(&lds_raw[buf][tile_off(r, ld_slot)]) = ra[i];
      *reinterpret_cast<u32x4*>(&lds_raw[buf][A_BYTES + tile_off(r, ld_slot)]) = rw[i];
    }
  };

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = a_lo_off ? 2 * nk1 : nk1;
  // GLDS staging: wave w copies rows [w * BM/4, +BM/4) of A and [w * 32, +32) of W, 8 rows per instruction
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int g_row = lane >> 3, g_ls = (lane & 7) ^ (g_row & 7);       // rows r0 + g_row with r0 % 8 == 0: swizzle by g_row
  auto gissue = [&](int kt, int buf) {
    const bf16_t* A = kt >= nk1 ? A_ + a_lo_off : A_;
    if (kt >= nk1) kt -= nk1;
    const int k = kbeg + kt * BK + g_ls * 8;
#pragma unroll
    for (int q = 0; q < BM / 32; ++q) {
      const int r = wave * (BM / 4) + q * 8;
      const int gr = min(m0 + r + g_row, M - 1);
      __builtin_amdgcn_global_load_lds((glb_void*)(A + (int64_t)gr * lda + k), (lds_void*)&lds_raw[buf][r * 128], 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wave * 32 + q * 8;
      const int gr = min(n0 + r + g_row, N - 1);
      __builtin_amdgcn_global_load_lds((glb_void*)(W + (int64_t)gr * ldw + k), (lds_void*)&lds_raw[buf][A_BYTES + r * 128], 16, 0,
                                       0);
    }
  };
  if (GLDS) {
    gissue(0, 0);
  } else {
    gload(0);
    swrite(0);
  }
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      if (GLDS) gissue(kt + 1, cur ^ 1);
      else gload(kt + 1);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[MI], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < MI) af[i] = *reinterpret_cast<const bf16x8*>(&lds_raw[cur][tile_off(wm + i * 16 + fr, kk * 4 + fq)]);
        bf[i] = *reinterpret_cast<const bf16x8*>(&lds_raw[cur][A_BYTES + tile_off(wn + i * 16 + fr, kk * 4 + fq)]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (!GLDS && kt + 1 < nk) swrite(cur ^ 1);
    __syncthreads();
  }

  // epilogue. C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn + j * 16 + fr;
    if (n >= N) continue;
    const float bv = bias ? bf16_to_f32(bias[n]) : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + fq * 4 + r;
        if (m >= M) continue;
        float v = acc[i][j][r] + bv;
        if (EPI == MN_GEMM_BF16) {
          reinterpret_cast<bf16_t*>(Cv)[(int64_t)m * ldc + n] = f32_to_bf16(v);
        } else if (EPI == MN_GEMM_BF16_GELU) {
          reinterpret_cast<bf16_t*>(Cv)[(int64_t)m * ldc + n] = f32_to_bf16(gelu_erf_f(v));
        } else if (EPI == MN_GEMM_F32) {
          reinterpret_cast<float*>(Cv)[(int64_t)m * ldc + n] = v;
        } else {
          reinterpret_cast<float*>(Cv)[(int64_t)m * ldc + n] += v;
        }
      }
    }
  }
}
}  // namespace

static int g_gemm_glds = 1, g_gemm_route256 = 1;
#ifdef MN_DEV_HOOKS   // A/B hooks of tools/ (libmingnative_dev.so only, include/mingnative_dev.h)
extern "C" MN_DEV_API void mn_gemm_tune(int glds) { g_gemm_glds = glds; }
extern "C" MN_DEV_API void mn_gemm_route256(int on) { g_gemm_route256 = on; }   // large problems go to gemm256.hip
#endif

static int gemm_launch(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const uint16_t* bias, void* C,
                       int64_t ldc, int M, int N, int K, int epilogue, int ksplit, int64_t c_zstride, hipStream_t st) {
  const int bm = (M <= 32) ? 32 : (M <= 64) ? 64 : 128;
  const int tiles = (int)(mn_cdiv(M, bm) * mn_cdiv(N, BN));
  int Kc = K;
  if (ksplit > 1) Kc = (int)(mn_cdiv(mn_cdiv(K, ksplit), BK) * BK);
  const int nz = (int)mn_cdiv(K, Kc);
  dim3 grid(tiles, nz);
  // whole k-tiles in every split-K range -> global_load_lds staging
  const bool glds = g_gemm_glds && (K % BK) == 0 && (Kc % BK) == 0;
#define MN_GEMM_LAUNCH_G(E, G)                                                                                         \
  do {                                                                                                                 \
    if (bm == 32)                                                                                                      \
      hipLaunchKernelGGL((gemm_bf16_kernel<E, 32, false, G>), grid, dim3(256), 0, st, A, lda, W, ldw, bias, C, ldc, M, N, K, Kc, c_zstride);  \
    else if (bm == 64)                                                                                                 \
      hipLaunchKernelGGL((gemm_bf16_kernel<E, 64, false, G>), grid, dim3(256), 0, st, A, lda, W, ldw, bias, C, ldc, M, N, K, Kc, c_zstride);  \
    else                                                                                                               \
      hipLaunchKernelGGL((gemm_bf16_kernel<E, 128, false, G>), grid, dim3(256), 0, st, A, lda, W, ldw, bias, C, ldc, M, N, K, Kc, c_zstride); \
  } while (0)
#define MN_GEMM_LAUNCH(E)              \
  do {                                 \
    if (glds) MN_GEMM_LAUNCH_G(E, true); \
    else MN_GEMM_LAUNCH_G(E, false);   \
  } while (0)
  switch (epilogue) {
    case MN_GEMM_BF16: MN_GEMM_LAUNCH(MN_GEMM_BF16); break;
    case MN_GEMM_BF16_GELU: MN_GEMM_LAUNCH(MN_GEMM_BF16_GELU); break;
    case MN_GEMM_F32: MN_GEMM_LAUNCH(MN_GEMM_F32); break;
    case MN_GEMM_F32_RESID: MN_GEMM_LAUNCH(MN_GEMM_F32_RESID); break;
    default:
      mn_set_error("mn_gemm_bf16: bad epilogue %d", epilogue);
      return MN_EINVAL;
  }
#undef MN_GEMM_LAUNCH
#undef MN_GEMM_LAUNCH_G
  return nz;
}

extern "C" int mn_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const uint16_t* bias,
                            void* C, int64_t ldc, int M, int N, int K, int epilogue, void* stream) {
  MN_CHECK_ARG(A && W && C, "mn_gemm_bf16: null pointer");
  MN_CHECK_ARG(M >= 1 && N >= 1 && K >= 8 && (K % 8) == 0, "mn_gemm_bf16: bad M=%d N=%d K=%d (K %% 8 == 0)", M, N, K);
  MN_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0,
               "mn_gemm_bf16: A/W rows must be 16-byte aligned");
  // enough 256 x 256 tiles to fill half the chip: the 8-wave 4-phase kernel (gemm256.hip: 1.2 PFLOP/s at 4096^3 against
  // 0.76 here); smaller problems keep the 128 x 128 tiles (more workgroups, 2-3 per CU)
  // (gemm256's epilogue stores 4 consecutive columns per lane and loads the bias 8 bytes at a time: C / bias alignment and ldc % 4
  //  are part of the routing condition, anything else stays on the 128-tile kernel with its scalar epilogue)
  const bool f32_out = epilogue == MN_GEMM_F32 || epilogue == MN_GEMM_F32_RESID;
  if (g_gemm_route256 && (K % 64) == 0 && (N % 4) == 0 && mn_cdiv(M, 256) * mn_cdiv(N, 256) >= 128 &&
      (int64_t)N * ldw * 2 < ((int64_t)1 << 32) && epilogue >= MN_GEMM_BF16 && epilogue <= MN_GEMM_F32_RESID &&
      (ldc % 4) == 0 && ((uintptr_t)C & (f32_out ? 15 : 7)) == 0 && ((uintptr_t)bias & 7) == 0) {
    static const int map[4] = {MN_G256_BF16, MN_G256_BF16_GELU, MN_G256_F32, MN_G256_F32_RESID};
    // gemm256 addresses its operands with 32-bit byte offsets: an activation matrix beyond 4 GiB goes in row chunks
    int64_t chunk = ((((int64_t)1 << 32) - 1) / (lda * 2)) / 256 * 256;
    if (chunk >= M) chunk = M;
    const int64_t c_elt = (epilogue == MN_GEMM_F32 || epilogue == MN_GEMM_F32_RESID) ? 4 : 2;
    for (int64_t m0 = 0; m0 < M; m0 += chunk) {
      mn_g256 a;
      memset(&a, 0, sizeof(a));
      a.A = A + m0 * lda; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias;
      a.C = reinterpret_cast<char*>(C) + m0 * ldc * c_elt; a.ldc = ldc;
      a.M = (int)((M - m0) < chunk ? (M - m0) : chunk); a.N = N; a.K = K;
      const int rc = mn_gemm256_ex(&a, map[epilogue], 1, stream);
      if (rc < 0) return rc;
    }
    return MN_OK;
  }
  const int rc = gemm_launch(A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, 1, 0, mn_stream(stream));
  if (rc < 0) return rc;
  MN_CHECK_LAUNCH("mn_gemm_bf16");
  return MN_OK;
}

// C fp32 [M,N] = (A_hi + A_lo) W^T + bias for activations split into bf16 hi and lo halves (A_lo = A_hi + a_lo_off
// elements): fp32-class products on the bf16 MFMA in ONE launch — the K loop runs over the hi rows, then over the lo
// rows, against the same W tiles; no second output and no combine pass.
extern "C" int mn_gemm_bf16_hilo(const uint16_t* A_hi, int64_t lda, int64_t a_lo_off, const uint16_t* W, int64_t ldw,
                                 const uint16_t* bias, float* C, int64_t ldc, int M, int N, int K, void* stream) {
  MN_CHECK_ARG(A_hi && W && C && a_lo_off > 0, "mn_gemm_bf16_hilo: null pointer");
  MN_CHECK_ARG(M >= 1 && N >= 1 && K >= 8 && (K % 8) == 0, "mn_gemm_bf16_hilo: bad M=%d N=%d K=%d (K %% 8 == 0)", M, N, K);
  MN_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0 && (a_lo_off % 8) == 0 && (((uintptr_t)A_hi | (uintptr_t)W) & 15) == 0,
               "mn_gemm_bf16_hilo: A/W rows must be 16-byte aligned");
  const int bm = (M <= 32) ? 32 : (M <= 64) ? 64 : 128;
  dim3 grid((unsigned)(mn_cdiv(M, bm) * mn_cdiv(N, BN)), 1);
  const bool glds = g_gemm_glds && (K % BK) == 0;
  hipStream_t st = mn_stream(stream);
  const GemmGroups ng{nullptr, nullptr, 0};
#define MN_HL(B_, G_)                                                                                                  \
  hipLaunchKernelGGL((gemm_bf16_kernel<MN_GEMM_F32, B_, false, G_>), grid, dim3(256), 0, st, A_hi, lda, W, ldw, bias, (void*)C, \
                     ldc, M, N, K, K, (int64_t)0, ng, a_lo_off)
  if (glds) { if (bm == 32) MN_HL(32, true); else if (bm == 64) MN_HL(64, true); else MN_HL(128, true); }
  else { if (bm == 32) MN_HL(32, false); else if (bm == 64) MN_HL(64, false); else MN_HL(128, false); }
#undef MN_HL
  MN_CHECK_LAUNCH("mn_gemm_bf16_hilo");
  return MN_OK;
}

// Grouped GEMM over `n_groups` experts: C[off_g : off_g + cnt_g] = A[off_g : off_g + cnt_g] W_g^T with W_g = W + g * w_gstride.
// off / cnt are DEVICE arrays (written by mn_moe_sort, no host sync); m_max bounds every cnt_g (e.g. the token count).
extern "C" int mn_gemm_bf16_grouped(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int64_t w_gstride,
                                    const int32_t* off, const int32_t* cnt, int n_groups, void* C, int64_t ldc, int m_max,
                                    int N, int K, int epilogue, void* stream) {
  MN_CHECK_ARG(A && W && C && off && cnt && n_groups >= 1 && m_max >= 1 && N >= 1 && K >= 8 && (K % 8) == 0,
               "mn_gemm_bf16_grouped: bad args");
  MN_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0 && (w_gstride % 8) == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0,
               "mn_gemm_bf16_grouped: alignment");
  MN_CHECK_ARG(epilogue == MN_GEMM_BF16 || epilogue == MN_GEMM_F32, "mn_gemm_bf16_grouped: epilogue must be BF16 or F32");
  const int tiles = (int)(mn_cdiv(m_max, 128) * mn_cdiv(N, BN));
  dim3 grid(tiles, n_groups);
  GemmGroups g{off, cnt, w_gstride};
  hipStream_t st = mn_stream(stream);
  const bool glds = g_gemm_glds && (K % BK) == 0;
#define MN_GG(E, G)                                                                                               \
  hipLaunchKernelGGL((gemm_bf16_kernel<E, 128, true, G>), grid, dim3(256), 0, st, A, lda, W, ldw, (const bf16_t*)nullptr, C, \
                     ldc, m_max, N, K, K, (int64_t)0, g)
  if (epilogue == MN_GEMM_BF16) { if (glds) MN_GG(MN_GEMM_BF16, true); else MN_GG(MN_GEMM_BF16, false); }
  else { if (glds) MN_GG(MN_GEMM_F32, true); else MN_GG(MN_GEMM_F32, false); }
#undef MN_GG
  MN_CHECK_LAUNCH("mn_gemm_bf16_grouped");
  return MN_OK;
}

// Split-K form for weight-streaming problems with few rows (M <= 64): the K range is cut into `ksplit`
// slices (rounded to 64), slice z writes the fp32 partial product to partials[z][M][N] (ldc = N).  Returns
// the number of slices actually used (>= 1) or a negative error; the caller reduces the slabs.
extern "C" int mn_gemm_bf16_splitk(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, float* partials,
                                   int M, int N, int K, int ksplit, void* stream) {
  MN_CHECK_ARG(A && W && partials, "mn_gemm_bf16_splitk: null pointer");
  MN_CHECK_ARG(M >= 1 && N >= 1 && K >= 8 && (K % 8) == 0 && ksplit >= 1, "mn_gemm_bf16_splitk: bad shape");
  MN_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0,
               "mn_gemm_bf16_splitk: A/W rows must be 16-byte aligned");
  const int nz = gemm_launch(A, lda, W, ldw, nullptr, partials, N, M, N, K, MN_GEMM_F32, ksplit, (int64_t)M * N,
                             mn_stream(stream));
  if (nz < 0) return nz;
  MN_CHECK_LAUNCH("mn_gemm_bf16_splitk");
  return nz;
}

// ===========================================================================================
// LayerNorm fp32 -> bf16 (one wave per row), SwiGLU, converters
// ===========================================================================================
// one wave per row; the row is read ONCE (D <= 4096: at most 16 float4 per lane stay in registers), statistics by wave
// reductions, two-pass variance on the register copy
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const bf16_t* __restrict__ g, const bf16_t* __restrict__ b,
                                                             float eps, bf16_t* __restrict__ y, int64_t ldy, int M, int D,
                                                             int gelu) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  f32x4 v[16];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int k = c * 256 + lane * 4;
    v[c] = k < D ? *reinterpret_cast<const f32x4*>(xr + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int k = c * 256 + lane * 4;
    if (k < D) { const f32x4 d = v[c] - mean; ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int k = c * 256 + lane * 4;
    if (k >= D) continue;
    float o[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = (o[i] - mean) * rstd;
      if (g) t *= bf16_to_f32(g[k + i]);
      if (b) t += bf16_to_f32(b[k + i]);
      if (gelu) t = gelu_erf_f(t);
      o[i] = t;
    }
    *reinterpret_cast<u32x2*>(y + (int64_t)row * ldy + k) = u32x2{cvt_pk_bf16(o[0], o[1]), cvt_pk_bf16(o[2], o[3])};
  }
}

extern "C" int mn_layernorm_bf16(const float* x, int64_t ldx, const uint16_t* g_, const uint16_t* b, float eps,
                                 uint16_t* y, int64_t ldy, int M, int D, int gelu, void* stream) {
  MN_CHECK_ARG(x && y && M >= 1 && D >= 4 && D <= 4096 && (D % 4) == 0 && (ldx % 4) == 0 && (ldy % 4) == 0,
               "mn_layernorm_bf16: bad args (D = %d: a multiple of 4, at most 4096)", D);
  if (D >= 512) {       // one workgroup per row, 16 bytes per thread (wide_glue.h): 4.3 TB/s where the wave-per-row form gave 2.9
    WideGlue g;
    memset(&g, 0, sizeof(g));
    g.h = x; g.ldh = ldx; g.norm = 2; g.ng = g_; g.nb = b; g.eps = eps; g.act = gelu ? 1 : 0; g.Y = y; g.ldy = ldy; g.M = M; g.D = D;
    wide_glue(g, mn_stream(stream));
    MN_CHECK_LAUNCH("mn_layernorm_bf16");
    return MN_OK;
  }
  hipLaunchKernelGGL(layernorm_bf16_kernel, dim3(mn_cdiv(M, 4)), dim3(256), 0, mn_stream(stream), x, ldx, g_, b, eps, y,
                     ldy, M, D, gelu);
  MN_CHECK_LAUNCH("mn_layernorm_bf16");
  return MN_OK;
}

__global__ void swiglu_bf16_kernel(const bf16_t* __restrict__ x12, int64_t ldx, bf16_t* __restrict__ h, int64_t ldh,
                                   int M, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over M * H/8
  const int per = H / 8;
  if (i >= (int64_t)M * per) return;
  const int m = (int)(i / per), c = (int)(i % per) * 8;
  const u32x4 a = *reinterpret_cast<const u32x4*>(x12 + (int64_t)m * ldx + c);
  const u32x4 b = *reinterpret_cast<const u32x4*>(x12 + (int64_t)m * ldx + H + c);
  u32x4 o;
  o.x = pack_bf16x2(silu_f(bf16lo_to_f32(a.x)) * bf16lo_to_f32(b.x), silu_f(bf16hi_to_f32(a.x)) * bf16hi_to_f32(b.x));
  o.y = pack_bf16x2(silu_f(bf16lo_to_f32(a.y)) * bf16lo_to_f32(b.y), silu_f(bf16hi_to_f32(a.y)) * bf16hi_to_f32(b.y));
  o.z = pack_bf16x2(silu_f(bf16lo_to_f32(a.z)) * bf16lo_to_f32(b.z), silu_f(bf16hi_to_f32(a.z)) * bf16hi_to_f32(b.z));
  o.w = pack_bf16x2(silu_f(bf16lo_to_f32(a.w)) * bf16lo_to_f32(b.w), silu_f(bf16hi_to_f32(a.w)) * bf16hi_to_f32(b.w));
  *reinterpret_cast<u32x4*>(h + (int64_t)m * ldh + c) = o;
}

extern "C" int mn_swiglu_bf16(const uint16_t* x12, int64_t ldx, uint16_t* h, int64_t ldh, int M, int H, void* stream) {
  MN_CHECK_ARG(x12 && h && M >= 1 && H >= 8 && (H % 8) == 0 && (ldx % 8) == 0 && (ldh % 8) == 0, "mn_swiglu_bf16: bad args");
  const int64_t n = (int64_t)M * (H / 8);
  hipLaunchKernelGGL(swiglu_bf16_kernel, dim3(mn_cdiv(n, 256)), dim3(256), 0, mn_stream(stream), x12, ldx, h, ldh, M, H);
  MN_CHECK_LAUNCH("mn_swiglu_bf16");
  return MN_OK;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = f32_to_bf16(x[i]);
}
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = bf16_to_f32(x[i]);
}
__global__ void f32_split_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                      int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    const bf16_t h = f32_to_bf16(v);
    hi[i] = h;
    lo[i] = f32_to_bf16(v - bf16_to_f32(h));
  }
}
__global__ void add_bcast_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                                 int64_t n, int64_t period) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    o[i] = a[i] + b[i % period];
}
__global__ void group_mean_add_kernel(const float* __restrict__ y, const float* __restrict__ x, float* __restrict__ o,
                                      int M, int D, int Cout) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * Cout) return;
  const int m = (int)(i / Cout), c = (int)(i % Cout), G = D / Cout;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += x[(int64_t)m * D + c * G + g];
  o[i] = y[i] + s / (float)G;
}
__global__ void repeat_add_kernel(const float* __restrict__ y, const float* __restrict__ sc, float* __restrict__ o,
                                  int M, int D, int Cin, float scale, float shift) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * D) return;
  const int m = (int)(i / D), n = (int)(i % D);
  o[i] = y[i] + sc[(int64_t)m * Cin + n / (D / Cin)] * scale + shift;
}
__global__ void clamp_kernel(float* __restrict__ x, int64_t n, float lo, float hi) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = fminf(fmaxf(x[i], lo), hi);
}
static inline int ew_blocks(int64_t n) { int64_t b = mn_cdiv(n, 256); return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

extern "C" int mn_add_bcast_f32(const float* a, const float* b, float* out, int64_t n, int64_t period, void* stream) {
  MN_CHECK_ARG(a && b && out && n > 0 && period > 0, "mn_add_bcast_f32: bad args");
  hipLaunchKernelGGL(add_bcast_kernel, dim3(ew_blocks(n)), dim3(256), 0, mn_stream(stream), a, b, out, n, period);
  MN_CHECK_LAUNCH("mn_add_bcast_f32");
  return MN_OK;
}
extern "C" int mn_group_mean_add(const float* y, const float* x, float* out, int M, int D, int Cout, void* stream) {
  MN_CHECK_ARG(y && x && out && M > 0 && Cout > 0 && D % Cout == 0, "mn_group_mean_add: bad args");
  hipLaunchKernelGGL(group_mean_add_kernel, dim3(mn_cdiv((int64_t)M * Cout, 256)), dim3(256), 0, mn_stream(stream), y, x,
                     out, M, D, Cout);
  MN_CHECK_LAUNCH("mn_group_mean_add");
  return MN_OK;
}
extern "C" int mn_repeat_add(const float* y, const float* s, float* out, int M, int D, int Cin, float scale, float shift,
                             void* stream) {
  MN_CHECK_ARG(y && s && out && M > 0 && Cin > 0 && D % Cin == 0, "mn_repeat_add: bad args");
  hipLaunchKernelGGL(repeat_add_kernel, dim3(mn_cdiv((int64_t)M * D, 256)), dim3(256), 0, mn_stream(stream), y, s, out, M,
                     D, Cin, scale, shift);
  MN_CHECK_LAUNCH("mn_repeat_add");
  return MN_OK;
}
extern "C" int mn_clamp_f32(float* x, int64_t n, float lo, float hi, void* stream) {
  MN_CHECK_ARG(x && n > 0, "mn_clamp_f32: bad args");
  hipLaunchKernelGGL(clamp_kernel, dim3(ew_blocks(n)), dim3(256), 0, mn_stream(stream), x, n, lo, hi);
  MN_CHECK_LAUNCH("mn_clamp_f32");
  return MN_OK;
}

extern "C" int mn_f32_to_bf16(const float* x, uint16_t* y, int64_t n, void* stream) {
  MN_CHECK_ARG(x && y && n >= 0, "mn_f32_to_bf16: bad args");
  if (n == 0) return MN_OK;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, mn_stream(stream), x, y, n);
  MN_CHECK_LAUNCH("mn_f32_to_bf16");
  return MN_OK;
}
extern "C" int mn_bf16_to_f32(const uint16_t* x, float* y, int64_t n, void* stream) {
  MN_CHECK_ARG(x && y && n >= 0, "mn_bf16_to_f32: bad args");
  if (n == 0) return MN_OK;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, mn_stream(stream), x, y, n);
  MN_CHECK_LAUNCH("mn_bf16_to_f32");
  return MN_OK;
}
extern "C" int mn_f32_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, void* stream) {
  MN_CHECK_ARG(x && hi && lo && n >= 0, "mn_f32_split_bf16: bad args");
  if (n == 0) return MN_OK;
  hipLaunchKernelGGL(f32_split_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, mn_stream(stream), x, hi, lo, n);
  MN_CHECK_LAUNCH("mn_f32_split_bf16");
  return MN_OK;
}

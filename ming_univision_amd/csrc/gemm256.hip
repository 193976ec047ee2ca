// gemm256.hip — the wide-row bf16 MFMA GEMM of the lock-step generation path (gfx950).
//
//   C[M, N] = epilogue( A[M, K] · W[N, K]^T )        A, W bf16 row-major (nn.Linear weight layout), fp32 accumulate
//
// Replaces the cuBLAS calls behind nn.Linear when hundreds of rows are in flight: the rectified-flow head's w12 / w3 /
// adaLN projections (diff_loss_rf_swiglu.py:54-72, 263-272, 283-292) for 128+ images advancing in lock-step, the
// MingTok encoder / decoder linears on image batches (mingtok layers/attention.py:49-51, mlp.py:34-40,
// swiglu_ffn.py:27-34) and the Bailing-MoE projections on long prompts (modeling_bailing_moe.py:760, 824).
//
// Structure (one 512-thread workgroup per CU, 8 waves as 2 (M) x 4 (N), each wave a 128 x 64 output block):
//   * 256 x 256 output tile, K step 64.  LDS: 2 K-tile buffers x {A, W} x 2 half-tiles of 128 rows x 128 B = 128 KiB.
//   * tiles are copied HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPR round trip); the LDS image of a wave
//     instruction is lane-linear (8 rows x 8 slots of 16 B), so the bank swizzle slot ^= row & 7 is applied to the
//     per-lane SOURCE address and again on the fragment read (conflict-free ds_read_b128).
//   * "half-tiles" follow the wave's fragment halves, not its rows: A half h holds the rows of M-fragments 4h..4h+3 of
//     both wave rows, W half h the columns of N-fragments 2h..2h+1 of all four wave columns.  A K-tile is then four
//     quadrant phases  (A0,W0) (A0,W1) (A1,W1) (A1,W0)  of 16 MFMAs each, and a half-tile is dead — free to be
//     re-filled — two phases after the phase that read it.
//   * Schedule: 2 long phases per K-tile, every phase = { ds_read the fragments of the next two quadrants — retired by
//     lgkmcnt(0) BEFORE the barrier, so their half-tiles may be re-filled one phase later —, issue one (phase 1) or three
//     (phase 2) half-tiles of a later K-tile, s_waitcnt vmcnt(8) — never 0 in the steady state, four half-tiles stay in
//     flight across the barriers —, s_barrier, 32 x v_mfma_f32_16x16x32_bf16 at raised priority, s_barrier }.  The two wave
//     rows run one barrier apart, so on every SIMD one wave is in its MFMA cluster while its partner reads LDS and issues
//     loads.  Hazard distances are by construction (see the schedule comments in the kernel): a half-tile is read >= 3 long
//     phases after it was issued and one phase after the counted wait that retires it.
//     (Round-2 A/B arms, removed from the shipped build: four phases of 16 MFMAs with 8 barriers per K-tile — 4-8 % slower on
//     every shape measured, 4096^3 1148-1243 vs 1325 TFLOP/s —, and one barrier per K-tile with vmcnt(0) — 9-13 % slower on the
//     hi/lo shapes of this path.)
//   * Rejected after measurement (tools/ab_gemm256.py, MI355X): the 32x32x16 MFMA in the same schedule (1048 vs 1243
//     TFLOP/s at 4096^3: a quadrant phase then has two independent accumulators for a 64-cycle MFMA); a one-wave-per-SIMD
//     form (4 waves x 128 x 128, the 64 accumulator fragments addressed literally as a[0:255] by inline-asm MFMAs — 154 VGPRs, no
//     scratch —, two fragment register sets, the reads / LDS-DMA of the next step interleaved 1 : 1 : 4 with the MFMAs by source
//     order, ONE barrier per K-tile): 8-25 % SLOWER than the two-phase 8-wave schedule on every shape (4096^3: 1268 vs 1375,
//     adaLN 1050 vs 1328, w12 at 1024 rows 1189 vs 1353 TFLOP/s) — with one wave per SIMD nothing covers the per-K-tile
//     vmcnt(0) + barrier, the issue slots of the interleaved instructions, or the epilogue.
//     A persistent tile loop whose LDS pipeline runs across output tiles (the next tile's first half-tiles issued during
//     the last phases and the epilogue of the current one) measured within +-1.5 % of the plain grid (adaLN 4589 vs 4646 us,
//     w12 at 1024 rows 177 vs 163 us): the dispatcher already starts the next workgroup's prologue while others compute.
//     MFMA issue order inside a quadrant (round 6, a power A/B on this power-limited kernel): weight-fragment-major — the W operand
//     stays for 4 consecutive MFMAs — measured 218.4 vs 215.8 us on RF w12 at 1536 rows and a LOWER clock at the same socket power
//     (1 720 vs 1 775 MHz at 1.35 kW, profiles/r06_ab_mfma_order.txt): the activation-fragment-major order below stays.
//   * The MFMA takes the W fragment as its A operand and the activation fragment as B, so a lane's 4 accumulator
//     registers are 4 CONSECUTIVE output columns of one row: 16-byte fp32 / 8-byte bf16 stores.
//
// Pairing modes (both are free: they only change which global row an LDS row is filled from):
//   * a_lo_off != 0 — hi/lo-stacked activations: A is a bf16 hi/lo pair (x = hi + lo to 2^-17); A half 0 holds 128 hi
//     rows, half 1 the same rows' lo parts, and the epilogue adds the two accumulator halves: fp32-class products from
//     bf16 MFMAs, a tile covers 128 real rows.
//   * w_pair_rows != 0 — gate/up pairing (SwiGLU): W half 0 holds 128 gate rows, half 1 the matching up rows
//     (w_pair_rows further down), so a lane holds silu-gate and up of the same hidden unit: the SwiGLU, its bias and
//     the bf16 hi/lo split of the next GEMM's operand happen in the epilogue, a tile covers 128 hidden units.
//
// fp8-MFMA regime (template F8; BASELINE configs[4]'s "fp8 MFMA", round 6 — a LABELLED reduced-arithmetic leg with its own tolerance, never
// the default): both operands are OCP e4m3 bytes with one fp32 scale per row, a K-tile is 128 k = the same 128 bytes per row, so staging,
// swizzle, fragment reads and schedule are unchanged; the two bf16 MFMAs a (fragment, fragment) pair issued per K-tile (k 0..31 | 32..63)
// become ONE v_mfma_f32_16x16x128_f8f6f4 on the concatenated 32-byte fragments (the builtin is the scaled one with literal-zero scale
// operands: the compiler then selects the UNSCALED instruction — the same bits as unit E8M0 block scales, measured, with one issue slot and
// one VGPR read fewer per MFMA: RF w12 98 -> 87 us) — the k order
// inside the instruction is then {slot fq, slot 4 + fq} for BOTH operands, a permutation the dot product does not see.  Twice the flops
// per K-tile at the same issue rate: the dense fp8 peak is 2 x the bf16 one (MI355X_MICROARCH.md), 4 x the hi/lo pair's.
#include <type_traits>
#include <utility>

#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 64;                       // k per K-tile
constexpr int HALF_BYTES = 128 * 128;        // one half-tile: 128 rows x 64 k bf16
constexpr int LDS_BYTES = 2 * 2 * 2 * HALF_BYTES;   // [buf][A|W][half] = 128 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

typedef mn_g256 G256;

enum { E_F32 = MN_G256_F32, E_BF16 = MN_G256_BF16, E_BF16_GELU = MN_G256_BF16_GELU, E_F32_RESID = MN_G256_F32_RESID,
       E_SWIGLU_SPLIT = MN_G256_SWIGLU_SPLIT, E_F32_RESID_GATE = MN_G256_F32_RESID_GATE, E_SWIGLU_BF16 = MN_G256_SWIGLU_BF16 };
__host__ __device__ constexpr bool epi_paired(int epi) { return epi == E_SWIGLU_SPLIT || epi == E_SWIGLU_BF16; }

__device__ __forceinline__ int lds_off(int buf, int op, int half) { return ((buf * 2 + op) * 2 + half) * HALF_BYTES; }

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N == 0 || N == 8, "counts used by the schedules");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// per-lane byte offsets (from A / W) of the global row behind LDS row L = 16*wave + 8*q + (lane >> 3) of half h, with
// the k-slot swizzle (lane & 7) ^ f(L), f(row) = (row >> 1) & 7, folded in.  All offsets fit 32 bits (host check).
template <bool HILO, bool PAIRED, bool F8 = false>
__device__ __forceinline__ void g256_src_offsets(const G256& p, int wave, int lane, int m0, int n0, int row0, int Mg, int kbeg,
                                                 uint32_t (&srcA)[2][2], uint32_t (&srcW)[2][2]) {
  const int rin = lane >> 3;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int L = wave * 16 + q * 8 + rin;
      const int ks = ((lane & 7) ^ ((L >> 1) & 7)) * 16;
      int gm, gn;
      if (HILO) gm = m0 + L;                                      // half = hi / lo part of the same rows
      else gm = m0 + (L >> 6) * 128 + h * 64 + (L & 63);          // wave row L / 64, fragment half h
      gm = row0 + min(gm, Mg - 1);
      if (p.a_rows) gm = p.a_rows[gm];
      if (PAIRED) gn = min(n0 + L, p.N - 1) + (h ? (int)p.w_pair_rows : 0);
      else gn = min(n0 + (L >> 5) * 64 + h * 32 + (L & 31), p.N - 1);
      constexpr int ES = F8 ? 1 : 2;                              // bytes per element
      srcA[h][q] = (uint32_t)(((int64_t)gm * p.lda + (HILO && h ? p.a_lo_off : 0) + kbeg) * ES + ks);
      srcW[h][q] = (uint32_t)(((int64_t)gn * p.ldw + kbeg) * ES + ks);
    }
}

// Exact-erf GELU (nn.GELU(), mingtok mlp.py:34-40) for a bf16 result: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, five
// fma + rcp + exp2 where erff costs ~40 instructions; 128 values per lane of a 256 x 256 tile made erff 8 us of a 50 us tile).
// The absolute error is 2^-14 of a bf16 ulp at |y| ~ 1; only the bf16 epilogue uses it.
__device__ __forceinline__ float gelu_erf_bf16out(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float e = q * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);      // 1 - erf(|x| / sqrt 2)
  const float one_plus_erf = x >= 0.f ? 2.0f - e : e;
  return 0.5f * x * one_plus_erf;
}

// Epilogue of one output tile.  A lane holds, per (i, j), output row 16 i + fr and the 4 CONSECUTIVE columns 16 j + 4 fq .. +3.
// One lane's 4 CONSECUTIVE output columns n..n+3 of row m (u: the paired "up" values): bias, epilogue, store.
template <int EPI>
__device__ __forceinline__ void g256_emit(const G256& p, char* Cz, int m, int n, f32x4 v, f32x4 u, int zslice) {
  constexpr bool paired = epi_paired(EPI);
  if (p.bias && zslice == 0) {
    const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n);
    v += f32x4{bf16lo_to_f32(b.x), bf16hi_to_f32(b.x), bf16lo_to_f32(b.y), bf16hi_to_f32(b.y)};
    if (paired) {
      const u32x2 b2 = *reinterpret_cast<const u32x2*>(p.bias + p.w_pair_rows + n);
      u += f32x4{bf16lo_to_f32(b2.x), bf16hi_to_f32(b2.x), bf16lo_to_f32(b2.y), bf16hi_to_f32(b2.y)};
    }
  }
  if (EPI == E_F32) {
    *reinterpret_cast<f32x4*>(Cz + ((int64_t)m * p.ldc + n) * 4) = v;
  } else if (EPI == E_F32_RESID) {
    f32x4* c = reinterpret_cast<f32x4*>(Cz + ((int64_t)m * p.ldc + n) * 4);
    *c += v;
  } else if (EPI == E_F32_RESID_GATE) {
    f32x4* c = reinterpret_cast<f32x4*>(Cz + ((int64_t)m * p.ldc + n) * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gate + (int64_t)m * p.ldgate + n);
    *c += g * v;
  } else if (EPI == E_BF16 || EPI == E_BF16_GELU) {
    if (EPI == E_BF16_GELU) v = f32x4{gelu_erf_bf16out(v.x), gelu_erf_bf16out(v.y), gelu_erf_bf16out(v.z), gelu_erf_bf16out(v.w)};
    *reinterpret_cast<u32x2*>(Cz + ((int64_t)m * p.ldc + n) * 2) = u32x2{cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w)};
  } else if (EPI == E_SWIGLU_BF16) {   // y = silu(gate) * up as plain bf16 (the batched bf16 path)
    *reinterpret_cast<u32x2*>(Cz + ((int64_t)m * p.ldc + n) * 2) =
        u32x2{cvt_pk_bf16(silu_f(v.x) * u.x, silu_f(v.y) * u.y), cvt_pk_bf16(silu_f(v.z) * u.z, silu_f(v.w) * u.w)};
  } else {  // E_SWIGLU_SPLIT: y = silu(gate) * up, stored as bf16 hi rows and lo rows
    uint32_t h0, l0, h1, l1;
    split_pk_bf16(silu_f(v.x) * u.x, silu_f(v.y) * u.y, h0, l0);
    split_pk_bf16(silu_f(v.z) * u.z, silu_f(v.w) * u.w, h1, l1);
    bf16_t* c = reinterpret_cast<bf16_t*>(Cz) + (int64_t)m * p.ldc + n;
    *reinterpret_cast<u32x2*>(c) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(c + p.c_lo_off) = u32x2{l0, l1};
  }
}

// Epilogue of one output tile of the 8-wave kernel.  A lane holds, per (i, j), output row 16 i + fr and the 4 CONSECUTIVE
// columns 16 j + 4 fq .. +3.
template <int EPI, bool HILO, bool F8 = false>
__device__ __forceinline__ void g256_epilogue(const G256& p, const f32x4 (&acc)[8][4], int wr, int wc, int fr, int fq, int m0,
                                              int n0, int row0, int Mg, int zslice, int grp = 0) {
  constexpr bool hilo = HILO, paired = epi_paired(EPI);
  char* Cz = reinterpret_cast<char*>(p.C);
  if (EPI == E_F32) Cz += (int64_t)zslice * p.c_zstride * 4;
  constexpr int mi_n = hilo ? 4 : 8, nj_n = paired ? 2 : 4;
#pragma unroll
  for (int i = 0; i < mi_n; ++i) {
    const int ml = hilo ? m0 + wr * 64 + i * 16 + fr : m0 + wr * 128 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < nj_n; ++j) {
      const int n = paired ? n0 + wc * 32 + j * 16 + fq * 4 : n0 + wc * 64 + j * 16 + fq * 4;
      if (ml >= Mg || n >= p.N) continue;           // N % 4 == 0 (host check): the 4 columns are all in or all out
      f32x4 v = acc[i][j];
      if (hilo) v += acc[(i + 4) & 7][j];
      f32x4 u = {0.f, 0.f, 0.f, 0.f};
      if (paired) { u = acc[i][(j + 2) & 3]; if (hilo) u += acc[(i + 4) & 7][(j + 2) & 3]; }
      if constexpr (F8) {                              // e4m3 operands: the row scales of both sides (before bias and epilogue)
        const float sa = p.a_scale[p.a_rows ? p.a_rows[row0 + ml] : row0 + ml];      // (the operand row this output row was multiplied from)
        const float* wsc = p.w_scale + (int64_t)grp * p.w_sstride;
        v *= *reinterpret_cast<const f32x4*>(wsc + n) * sa;
        if (paired) u *= *reinterpret_cast<const f32x4*>(wsc + p.w_pair_rows + n) * sa;
      }
      g256_emit<EPI>(p, Cz, row0 + ml, n, v, u, zslice);
    }
  }
}

template <int EPI, bool HILO, bool F8 = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const G256 p) {
  static_assert(!F8 || !HILO, "the fp8 regime is single-pass");
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  constexpr bool hilo = HILO, paired = epi_paired(EPI);
  const int rows_per_tile = hilo ? 128 : 256, cols_per_tile = paired ? 128 : 256;
  const int tiles_m = (p.M + rows_per_tile - 1) / rows_per_tile, tiles_n = (p.N + cols_per_tile - 1) / cols_per_tile;
  // XCD-aware tile order (bijective for any tile count): the tiles of one XCD are consecutive; inside that range tiles run in
  // bands of group_m M-tiles (below), so the ~32 tiles an XCD has in flight share few activation AND few weight panels in its L2.
  // With a device-built tile list the order covers the LIVE workgroups only: the grid is sized for the worst split of the rows over
  // the groups, and ranging over it left the last XCD(s) with the tail of the list — few tiles, or only the shared experts' full
  // ones (expert launches at 1536 rows: 360 -> 303 us gate/up, 214 -> 173 us down).
  int bid = blockIdx.x;
  {
    int nt = gridDim.x;
    if (p.g_off && p.tile_g) {
      const int gb = p.group_m > 0 ? p.group_m : 1;
      nt = min(nt, (*p.n_tiles + gb - 1) / gb * gb * tiles_n);
      if (bid >= nt) return;                     // uniform per workgroup
    }
    const int q = nt / 8, r = nt % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm = bid % tiles_m, tn = bid / tiles_m;
  if (p.group_m > 0 && !p.g_off) {
    // banded order: bands of group_m M-tiles, N-tiles fastest across a band's columns, so the ~32 tiles an XCD runs at a time
    // span group_m activation panels x 32 / group_m weight panels instead of all M-tiles x 2-3 weight panels
    const int width = p.group_m * tiles_n, gid = bid / width, first = gid * p.group_m;
    const int gsz = min(tiles_m - first, p.group_m), r = bid % width;
    tm = first + r % gsz;
    tn = r / gsz;
  }
  int m0 = tm * rows_per_tile;
  // grouped form (MoE experts): group g owns rows [g_off[g], g_off[g] + g_cnt[g]) of A (through a_rows when given: a
  // gather by index while staging) and of C, and the weights W + g * w_gstride; p.M bounds every group.  The group is
  // blockIdx.z, or — with a device-built tile list — row tile t = bid / tiles_n of the list (no empty workgroups; the
  // N-tiles of one row tile are consecutive, so its gathered rows stay in one L2).
  int Mg = p.M, row0 = 0, grp = 0;
  if (p.g_off) {
    if (p.tile_g) {
      // list order: bands of group_m consecutive row tiles (mostly the tiles of one expert: same weights), row tile fastest,
      // so the workgroups an XCD runs together share weight panels as well as gathered rows; the tail band is narrower
      const int nl = *p.n_tiles, gb = p.group_m > 0 ? p.group_m : 1;
      const int band = bid / (gb * tiles_n), first = band * gb, r = bid % (gb * tiles_n);
      const int gsz = min(nl - first, gb);
      if (first >= nl || r >= gsz * tiles_n) return;   // uniform per workgroup
      const int t = first + r % gsz;
      tn = r / gsz;
      grp = p.tile_g[t];
      m0 = p.tile_m0[t];
    } else {
      grp = blockIdx.z;
    }
    row0 = p.g_off[grp];
    Mg = p.g_cnt[grp];
    if (m0 >= Mg) return;                        // uniform per workgroup
  }
  const int n0 = tn * cols_per_tile;
  const int kbeg = blockIdx.y * p.Kc, kend = min(p.K, kbeg + p.Kc);
  const int nk = (kend - kbeg) / (F8 ? 2 * BK : BK);        // (an fp8 K-tile is 128 k: the same 128 bytes per row)

  // ---- staging: wave w fills LDS rows [16w, 16w+16) of a half-tile with two instructions of 8 rows x 8 slots ----
  uint32_t srcA[2][2], srcW[2][2];
  g256_src_offsets<hilo, paired, F8>(p, wave, lane, m0, n0, row0, Mg, kbeg, srcA, srcW);
  // Row tiles of a tile list (the expert GEMMs) are often mostly padding — 144 rows per expert = a full tile + a 16-row one —, and
  // a power-limited chip pays for the MFMAs and fragment reads of clamped rows in clock: M-fragments of this wave row without a live
  // row are skipped (wave-uniform count, one copy of the K loop per count so that no branch sits inside an MFMA cluster).  The last
  // row tile of a dense problem whose row count is not a multiple of 128 takes the same path.
  constexpr bool thin = HILO && (EPI == E_F32 || EPI == E_SWIGLU_SPLIT);
  const int live_f = (thin && p.thin) ? __builtin_amdgcn_readfirstlane(min(4, max(0, (Mg - m0 - wr * 64 + 15) >> 4))) : 4;
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W) + (int64_t)grp * p.w_gstride * (F8 ? 1 : 2);
  auto stage = [&](int op, int h, int kt, int buf) {   // op 0 = A, 1 = W; all arguments compile-time or wave-uniform
    char* dst = &lds[lds_off(buf, op, h) + wave * 2048];
    const uint32_t koff = (uint32_t)kt * (BK * 2);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const char* src = op == 0 ? Ab + (srcA[h][q] + koff) : Wb + (srcW[h][q] + koff);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + q * 1024), 16, 0, 0);
    }
  };

  // ---- fragment reads (v_mfma_f32_16x16x32_bf16): row = 16 * frag + (lane & 15), k-slot = (4 kk + (lane >> 4)) ^ f(row) ----
  // (A 32x32x16 form of the same schedule was built and measured: 1048 vs 1243 TFLOP/s at 4096^3 — a quadrant phase then
  //  has only two independent accumulators for an MFMA of 64-cycle latency.)
  int lane_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) lane_off[kk] = fr * 128 + (((4 * kk + fq) ^ ((fr >> 1) & 7)) << 4);
  const int a_wave = wr * 64 * 128, w_wave = wc * 32 * 128;
  bf16x8 af[8], wf0[4], wf1[4];
  auto read_a = [&](int buf, int h, auto LVc) {   // LV = live M-fragments of this wave row
    constexpr int LV = decltype(LVc)::value;
    const char* base = &lds[lds_off(buf, 0, h) + a_wave];
#pragma unroll
    for (int x = 0; x < 2 * LV; ++x) af[x] = *reinterpret_cast<const bf16x8*>(base + (x >> 1) * 2048 + lane_off[x & 1]);   // x = 2 i + kk
  };
  auto read_w = [&](int buf, int h, bf16x8 (&wf)[4]) {
    const char* base = &lds[lds_off(buf, 1, h) + w_wave];
#pragma unroll
    for (int x = 0; x < 4; ++x) wf[x] = *reinterpret_cast<const bf16x8*>(base + (x >> 1) * 2048 + lane_off[x & 1]);   // x = 2 j + kk
  };
  f32x4 acc[8][4];                    // [M fragment][N fragment]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // quadrant (mh, nh): D[n][m] += W-frag (as A operand) x activation frag (as B operand)
  auto quad = [&](int mh, int nh, bf16x8 (&wf)[4], auto LVc) {
    constexpr int LV = decltype(LVc)::value;
    if constexpr (F8) {                                 // one K = 128 e4m3 MFMA per fragment pair (zero scale operands -> the unscaled instruction)
#pragma unroll
      for (int i = 0; i < LV; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const i32x8 wa = __builtin_shufflevector(__builtin_bit_cast(i32x4, wf[2 * j]), __builtin_bit_cast(i32x4, wf[2 * j + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
          const i32x8 xa = __builtin_shufflevector(__builtin_bit_cast(i32x4, af[2 * i]), __builtin_bit_cast(i32x4, af[2 * i + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
          acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, acc[mh * 4 + i][nh * 2 + j], 0, 0, 0, 0, 0, 0);
        }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < LV; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[mh * 4 + i][nh * 2 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * j + kk], af[2 * i + kk], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
  };

  // ---------------- 2 long phases per K-tile (32 MFMAs each), counted vmcnt, wave rows one barrier apart ----------------
  // K-tile T in buffer X (Y = X ^ 1):
  //   phase 1: read X.A0, X.W0, X.W1 (retired by lgkmcnt(0) BEFORE the barrier: free for re-issue one phase later);
  //            issue Y.A1 <- T+1;                  MFMAs (A0,W0) (A0,W1)
  //   phase 2: read X.A1 (same rule);  issue X.A0, X.W0, X.W1 <- T+2;      MFMAs (A1,W1) (A1,W0)
  //   RAW: X.A1(T) issued (T-1).1, retired by the vmcnt(8) of T.1 (8 = Y.A1 + the three half-tiles of (T-1).2), read T.2;
  //        Y.{A0,W0,W1}(T+1) issued (T-1).2, retired by the vmcnt(8) of T.2, read (T+1).1 — always one phase after the wait.
  //   WAR: one phase, safe because every wave's reads completed before the barrier that precedes the re-issue.
  auto phase2 = [&](auto Xc, auto PHc, int T, auto LVc) {
    constexpr int X = decltype(Xc)::value, PH = decltype(PHc)::value, Y = X ^ 1;
    const int tgt = PH == 0 ? T + 1 : T + 2;
    constexpr bool any = decltype(LVc)::value > 0;        // a wave row without live rows reads no fragments at all
    if (PH == 0) { if (any) { read_w(X, 0, wf0); read_w(X, 1, wf1); } __builtin_amdgcn_sched_barrier(0); read_a(X, 0, LVc); }
    else read_a(X, 1, LVc);
    if (tgt < nk) {   // (issuing the LDS-DMA before the fragment reads instead measured within +-1 %)
      if (PH == 0) stage(0, 1, tgt, Y);
      else { stage(0, 0, tgt, X); stage(1, 0, tgt, X); stage(1, 1, tgt, X); }
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (PH == 0) { quad(0, 0, wf0, LVc); quad(0, 1, wf1, LVc); }
    else { quad(1, 1, wf1, LVc); quad(1, 0, wf0, LVc); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);        // the next phase's LDS reads stay behind this barrier
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  stage(0, 0, 0, 0); stage(1, 0, 0, 0); stage(1, 1, 0, 0); stage(0, 1, 0, 0);
  if (nk > 1) { stage(0, 0, 1, 1); stage(1, 0, 1, 1); stage(1, 1, 1, 1); wait_vm<8>(); }
  else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();
  auto kloop = [&](auto LVc) {
    for (int t = 0; t < nk; t += 2) {
      phase2(I0{}, I0{}, t, LVc); phase2(I0{}, I1{}, t, LVc);
      if (t + 1 < nk) { phase2(I1{}, I0{}, t + 1, LVc); phase2(I1{}, I1{}, t + 1, LVc); }
    }
  };
  if constexpr (thin) {
    switch (live_f) {
      case 0: kloop(std::integral_constant<int, 0>{}); break;
      case 1: kloop(std::integral_constant<int, 1>{}); break;
      case 2: kloop(std::integral_constant<int, 2>{}); break;
      case 3: kloop(std::integral_constant<int, 3>{}); break;
      default: kloop(std::integral_constant<int, 4>{}); break;
    }
  } else {
    kloop(std::integral_constant<int, 4>{});
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();

  g256_epilogue<EPI, HILO, F8>(p, acc, wr, wc, fr, fq, m0, n0, row0, Mg, blockIdx.y, grp);
}



}  // namespace

// Tile order: dense problems run in bands of 4 M-tiles inside an XCD's tile range; tile lists keep the N-tiles of a row tile together
// (bands of 2-8 row tiles measured 0.7 % slower end to end).
static int g_g256_groupm = 4, g_g256_groupb = 1, g_g256_thin = 1;
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_gemm256_tune_thin(int on) { g_g256_thin = on; }   // A/B hook: skip the MFMAs of dead M-fragments in tile lists
extern "C" MN_DEV_API void mn_gemm256_tune_order(int group_m, int group_list) {   // A/B hook (tools/, libmingnative_dev.so only)
  g_g256_groupm = group_m;
  g_g256_groupb = group_list;
}
#endif

// Generic launcher.  Returns the number of split-K slices used (>= 1) or a negative error.
static int g256_launch(const G256& a, int epi, int ksplit, hipStream_t st) {
  const bool hilo = a.a_lo_off != 0, paired = a.w_pair_rows != 0;
  if (paired != epi_paired(epi)) { mn_set_error("gemm256: epilogue %d and w_pair_rows disagree", epi); return MN_EINVAL; }
  const int tiles = (int)((a.tile_g ? a.max_mtiles : mn_cdiv(a.M, hilo ? 128 : 256)) * mn_cdiv(a.N, paired ? 128 : 256));
  G256 p = a;
  p.group_m = a.tile_g ? g_g256_groupb : g_g256_groupm;
  p.thin = g_g256_thin;
  p.Kc = p.K;
  const int kq = a.f8 ? 4 * BK : 2 * BK;             // an even number of K-tiles per slice (fp8: 128 k per K-tile)
  if (ksplit > 1) p.Kc = (int)(mn_cdiv(mn_cdiv(p.K, ksplit), kq) * kq);
  const int nz = (int)mn_cdiv(p.K, p.Kc);
  dim3 grid(tiles, nz, (a.g_off && !a.tile_g) ? a.n_groups : 1);
  if (a.f8) {                                          // the fp8-MFMA regime: plain rows, fp32 (split-K) or SwiGLU -> bf16 results
    if (hilo || !a.a_scale || !a.w_scale || (a.K % (2 * BK)) != 0 || (epi != E_F32 && epi != E_SWIGLU_BF16)) {
      mn_set_error("gemm256 (fp8): needs row scales, K %% 128 == 0, no hi/lo rows, epilogue F32 or SWIGLU_BF16");
      return MN_EINVAL;
    }
    if (epi == E_F32) hipLaunchKernelGGL((gemm256_kernel<E_F32, false, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm256_kernel<E_SWIGLU_BF16, false, true>), grid, dim3(512), 0, st, p);
    return nz;
  }
#define G256_GO(E)                                                                          \
  do {                                                                                      \
    if (hilo) hipLaunchKernelGGL((gemm256_kernel<E, true>), grid, dim3(512), 0, st, p);     \
    else hipLaunchKernelGGL((gemm256_kernel<E, false>), grid, dim3(512), 0, st, p);         \
  } while (0)
  switch (epi) {
    case E_F32: G256_GO(E_F32); break;
    case E_BF16: G256_GO(E_BF16); break;
    case E_BF16_GELU: G256_GO(E_BF16_GELU); break;
    case E_F32_RESID: G256_GO(E_F32_RESID); break;
    case E_SWIGLU_SPLIT: G256_GO(E_SWIGLU_SPLIT); break;
    case E_F32_RESID_GATE: G256_GO(E_F32_RESID_GATE); break;
    case E_SWIGLU_BF16: G256_GO(E_SWIGLU_BF16); break;
    default: mn_set_error("gemm256: bad epilogue %d", epi); return MN_EINVAL;
  }
#undef G256_GO
  return nz;
}

// Internal generic entry (engine.hip): any combination of hi/lo rows, gate/up pairing, groups, split-K.
extern "C" int mn_gemm256_ex(const mn_g256* a, int epi, int ksplit, void* stream) {
  const int nz = g256_launch(*a, epi, ksplit, mn_stream(stream));
  if (nz < 0) return nz;
  MN_CHECK_LAUNCH("mn_gemm256_ex");
  return nz;
}

// slices mn_gemm256_ex will use for a ksplit request (whole pairs of K-tiles per slice)
extern "C" int mn_gemm256_slices(int K, int ksplit) {
  if (ksplit <= 1) return 1;
  const int Kc = (int)(mn_cdiv(mn_cdiv(K, ksplit), 2 * BK) * 2 * BK);
  return (int)mn_cdiv(K, Kc);
}
extern "C" int mn_gemm256_f8_slices(int K, int ksplit) {      // the fp8 regime: 128 k per K-tile
  if (ksplit <= 1) return 1;
  const int Kc = (int)(mn_cdiv(mn_cdiv(K, ksplit), 4 * BK) * 4 * BK);
  return (int)mn_cdiv(K, Kc);
}

static bool g256_shape_ok(const void* A, int64_t lda, int64_t a_lo_off, const void* W, int64_t ldw, int64_t w_rows, int M, int N,
                          int K) {
  // whole K-tiles; 16-byte aligned rows; every per-lane source offset fits 32 bits; 4 output columns per lane
  return M >= 1 && N >= 4 && (N % 4) == 0 && K >= BK && (K % BK) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (a_lo_off % 8) == 0 &&
         (((uintptr_t)A | (uintptr_t)W) & 15) == 0 && ((int64_t)M * lda + a_lo_off) * 2 < ((int64_t)1 << 32) &&
         w_rows * ldw * 2 < ((int64_t)1 << 32);
}

// the epilogue's vector accesses: 16-byte fp32 / 8-byte bf16 stores of 4 consecutive columns, 8-byte bias loads
static bool g256_out_ok(const void* C, int64_t ldc, int64_t c_lo_off, const void* bias, bool f32_out) {
  return (ldc % 4) == 0 && (c_lo_off % 4) == 0 && ((uintptr_t)C & (f32_out ? 15 : 7)) == 0 && ((uintptr_t)bias & 7) == 0;
}

extern "C" int mn_gemm256_supported(int64_t lda, int64_t a_lo_off, int64_t ldw, int64_t w_rows, int M, int N, int K) {
  return g256_shape_ok(nullptr, lda, a_lo_off, nullptr, ldw, w_rows, M, N, K) ? 1 : 0;
}

// C = epilogue(A W^T + bias).  epilogue: MN_GEMM_BF16 / BF16_GELU / F32 / F32_RESID (same enum as mn_gemm_bf16).
// a_lo_off != 0: A is a bf16 hi/lo pair (lo rows a_lo_off elements after the hi rows), C = (A_hi + A_lo) W^T.
extern "C" int mn_gemm256(const uint16_t* A, int64_t lda, int64_t a_lo_off, const uint16_t* W, int64_t ldw,
                          const uint16_t* bias, void* C, int64_t ldc, int M, int N, int K, int epilogue, void* stream) {
  MN_CHECK_ARG(A && W && C, "mn_gemm256: null pointer");
  MN_CHECK_ARG(g256_shape_ok(A, lda, a_lo_off, W, ldw, N, M, N, K),
               "mn_gemm256: unsupported shape M=%d N=%d K=%d (K %% 64 == 0, N %% 4 == 0, 16-byte rows, < 4 GiB operands)", M, N, K);
  int e;
  switch (epilogue) {
    case MN_GEMM_BF16: e = E_BF16; break;
    case MN_GEMM_BF16_GELU: e = E_BF16_GELU; break;
    case MN_GEMM_F32: e = E_F32; break;
    case MN_GEMM_F32_RESID: e = E_F32_RESID; break;
    default: mn_set_error("mn_gemm256: bad epilogue %d", epilogue); return MN_EINVAL;
  }
  MN_CHECK_ARG(g256_out_ok(C, ldc, 0, bias, e == E_F32 || e == E_F32_RESID),
               "mn_gemm256: C must be 16-byte (fp32) / 8-byte (bf16) aligned with ldc %% 4 == 0, bias 8-byte aligned (ldc=%lld)", (long long)ldc);
  G256 p{};
  p.A = A; p.lda = lda; p.a_lo_off = a_lo_off; p.W = W; p.ldw = ldw; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  const int rc = g256_launch(p, e, 1, mn_stream(stream));
  if (rc < 0) return rc;
  MN_CHECK_LAUNCH("mn_gemm256");
  return MN_OK;
}

// Split-K form: slice z writes the fp32 partial product (bias in slice 0) to partials + z * M * N (ldc = N).
// Returns the number of slices used.
extern "C" int mn_gemm256_splitk(const uint16_t* A, int64_t lda, int64_t a_lo_off, const uint16_t* W, int64_t ldw,
                                 const uint16_t* bias, float* partials, int M, int N, int K, int ksplit, void* stream) {
  MN_CHECK_ARG(A && W && partials && ksplit >= 1, "mn_gemm256_splitk: bad args");
  MN_CHECK_ARG(g256_shape_ok(A, lda, a_lo_off, W, ldw, N, M, N, K), "mn_gemm256_splitk: unsupported shape M=%d N=%d K=%d", M, N, K);
  MN_CHECK_ARG(g256_out_ok(partials, N, 0, bias, true), "mn_gemm256_splitk: partials must be 16-byte aligned, bias 8-byte aligned");
  G256 p{};
  p.A = A; p.lda = lda; p.a_lo_off = a_lo_off; p.W = W; p.ldw = ldw; p.bias = bias; p.C = partials; p.ldc = N;
  p.c_zstride = (int64_t)M * N; p.M = M; p.N = N; p.K = K;
  const int nz = g256_launch(p, E_F32, ksplit, mn_stream(stream));
  if (nz < 0) return nz;
  MN_CHECK_LAUNCH("mn_gemm256_splitk");
  return nz;
}

// SwiGLU-fused form (swiglu_ffn.py:30-34; diff_loss_rf_swiglu.py:54-72): W12 bf16 [2 * hidden, K] (gate rows, then up rows),
// b12 bf16 [2 * hidden] or NULL; Y bf16: hi rows [M, hidden] at Y, lo rows y_lo_off elements further:
//   Y_hi + Y_lo = silu(A W_gate^T + b_gate) * (A W_up^T + b_up)   to 2^-17.
extern "C" int mn_gemm256_swiglu_split(const uint16_t* A, int64_t lda, int64_t a_lo_off, const uint16_t* W12, int64_t ldw,
                                       const uint16_t* b12, uint16_t* Y, int64_t ldy, int64_t y_lo_off, int M, int hidden,
                                       int K, void* stream) {
  MN_CHECK_ARG(A && W12 && Y && y_lo_off > 0, "mn_gemm256_swiglu_split: bad args");
  MN_CHECK_ARG(g256_shape_ok(A, lda, a_lo_off, W12, ldw, 2 * (int64_t)hidden, M, hidden, K) && g256_out_ok(Y, ldy, y_lo_off, b12, false),
               "mn_gemm256_swiglu_split: unsupported shape M=%d hidden=%d K=%d (or Y / b12 not 8-byte aligned, ldy / y_lo_off %% 4)", M, hidden, K);
  G256 p{};
  p.A = A; p.lda = lda; p.a_lo_off = a_lo_off; p.W = W12; p.ldw = ldw; p.w_pair_rows = hidden; p.bias = b12;
  p.C = Y; p.ldc = ldy; p.c_lo_off = y_lo_off; p.M = M; p.N = hidden; p.K = K;
  const int rc = g256_launch(p, E_SWIGLU_SPLIT, 1, mn_stream(stream));
  if (rc < 0) return rc;
  MN_CHECK_LAUNCH("mn_gemm256_swiglu_split");
  return MN_OK;
}

// SwiGLU-fused form with a plain bf16 result (the batched bf16 path: MingTok encoder / semantic-decoder blocks,
// swiglu_ffn.py:30-34): Y bf16 [M, hidden] = silu(A Wg^T + bg) * (A Wu^T + bu).  a_lo_off = 0 for plain bf16 activations.
extern "C" int mn_gemm256_swiglu(const uint16_t* A, int64_t lda, int64_t a_lo_off, const uint16_t* W12, int64_t ldw,
                                 const uint16_t* b12, uint16_t* Y, int64_t ldy, int M, int hidden, int K, void* stream) {
  MN_CHECK_ARG(A && W12 && Y, "mn_gemm256_swiglu: bad args");
  MN_CHECK_ARG(g256_shape_ok(A, lda, a_lo_off, W12, ldw, 2 * (int64_t)hidden, M, hidden, K) && g256_out_ok(Y, ldy, 0, b12, false),
               "mn_gemm256_swiglu: unsupported shape M=%d hidden=%d K=%d (or Y / b12 not 8-byte aligned, ldy %% 4)", M, hidden, K);
  G256 p{};
  p.A = A; p.lda = lda; p.a_lo_off = a_lo_off; p.W = W12; p.ldw = ldw; p.w_pair_rows = hidden; p.bias = b12;
  p.C = Y; p.ldc = ldy; p.M = M; p.N = hidden; p.K = K;
  const int rc = g256_launch(p, E_SWIGLU_BF16, 1, mn_stream(stream));
  if (rc < 0) return rc;
  MN_CHECK_LAUNCH("mn_gemm256_swiglu");
  return MN_OK;
}

// Grouped form (MoE experts; replaces the per-expert loop of modeling_bailing_moe.py:605-639 when hundreds of rows are in
// flight): group g multiplies rows [off[g], off[g] + cnt[g]) — row position r reads A row a_rows[r] when a_rows is given (the
// gather of the expert-sorted order, done while staging) — by W + g * w_gstride and writes rows off[g].. of C.
//   swiglu == 0: C fp32 [*, N] = (A_hi + A_lo) W_g^T
//   swiglu == 1: W_g holds 2N rows (gate, up); C bf16 hi rows [*, N] and lo rows c_lo_off elements further = silu(gate) * up
// off / cnt are device arrays (mn_moe_sort); no group may exceed m_max rows.
extern "C" int mn_gemm256_grouped(const uint16_t* A, int64_t lda, int64_t a_lo_off, int64_t a_rows_total, const int32_t* a_rows,
                                  const uint16_t* W, int64_t ldw, int64_t w_gstride, const int32_t* off, const int32_t* cnt,
                                  int n_groups, void* C, int64_t ldc, int64_t c_lo_off, int m_max, int N, int K, int swiglu,
                                  void* stream) {
  MN_CHECK_ARG(A && W && C && off && cnt && n_groups >= 1 && m_max >= 1 && a_lo_off > 0 && a_rows_total >= 1, "mn_gemm256_grouped: bad args");
  // every per-lane source offset (gathered row * lda + a_lo_off, in bytes) must fit 32 bits: bounded by the row count of A
  MN_CHECK_ARG(N >= 4 && (N % 4) == 0 && K >= BK && (K % BK) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (a_lo_off % 8) == 0 &&
                   (w_gstride % 8) == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0 && (!swiglu || c_lo_off > 0) &&
                   (a_rows_total * lda + a_lo_off) * 2 < ((int64_t)1 << 32) && (int64_t)(swiglu ? 2 : 1) * N * ldw * 2 < ((int64_t)1 << 32) &&
                   g256_out_ok(C, ldc, c_lo_off, nullptr, !swiglu),
               "mn_gemm256_grouped: unsupported shape N=%d K=%d (or C misaligned / ldc %% 4)", N, K);
  G256 p{};
  p.A = A; p.lda = lda; p.a_lo_off = a_lo_off; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.c_lo_off = c_lo_off;
  p.w_pair_rows = swiglu ? N : 0; p.M = m_max; p.N = N; p.K = K;
  p.g_off = off; p.g_cnt = cnt; p.w_gstride = w_gstride; p.a_rows = a_rows; p.n_groups = n_groups;
  const int rc = g256_launch(p, swiglu ? E_SWIGLU_SPLIT : E_F32, 1, mn_stream(stream));
  if (rc < 0) return rc;
  MN_CHECK_LAUNCH("mn_gemm256_grouped");
  return MN_OK;
}

// Grouped form over a device-built row-tile list (mn_moe_sort_tiles: tile t = rows [tile_m0[t], ..) of group tile_g[t]; tile_rows =
// 128 for hi/lo activations, 256 for plain bf16 ones): no workgroup is launched for an empty tile slot beyond *n_tiles, whatever
// the split of the rows over the groups.  a_lo_off = 0: plain bf16 activations (the bf16 prefill path).
//   epi 0: C fp32 [*, N]           1: C bf16 [*, N]
//   epi 4: W_g holds 2N rows (gate, up); C bf16 hi rows and lo rows c_lo_off further = silu(gate) * up       6: the same, plain bf16
// max_mtiles >= sum_g ceil(cnt[g] / tile_rows) (e.g. total_rows / tile_rows + n_groups).
extern "C" int mn_gemm256_grouped_tiles(const uint16_t* A, int64_t lda, int64_t a_lo_off, int64_t a_rows_total, const int32_t* a_rows,
                                        const uint16_t* W, int64_t ldw, int64_t w_gstride, const int32_t* off, const int32_t* cnt,
                                        int n_groups, const int32_t* tile_g, const int32_t* tile_m0, const int32_t* n_tiles,
                                        int max_mtiles, void* C, int64_t ldc, int64_t c_lo_off, int N, int K, int epi, void* stream) {
  const bool paired = epi == E_SWIGLU_SPLIT || epi == E_SWIGLU_BF16;
  MN_CHECK_ARG(A && W && C && off && cnt && tile_g && tile_m0 && n_tiles && n_groups >= 1 && max_mtiles >= 1 && a_lo_off >= 0 &&
                   a_rows_total >= 1, "mn_gemm256_grouped_tiles: bad args");
  MN_CHECK_ARG(epi == E_F32 || epi == E_BF16 || epi == E_SWIGLU_SPLIT || epi == E_SWIGLU_BF16, "mn_gemm256_grouped_tiles: epi %d", epi);
  MN_CHECK_ARG(N >= 4 && (N % 4) == 0 && K >= BK && (K % BK) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (a_lo_off % 8) == 0 &&
                   (w_gstride % 8) == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0 && (epi != E_SWIGLU_SPLIT || c_lo_off > 0) &&
                   (a_rows_total * lda + a_lo_off) * 2 < ((int64_t)1 << 32) && (int64_t)(paired ? 2 : 1) * N * ldw * 2 < ((int64_t)1 << 32) &&
                   g256_out_ok(C, ldc, c_lo_off, nullptr, epi == E_F32),
               "mn_gemm256_grouped_tiles: unsupported shape N=%d K=%d (or C misaligned / ldc %% 4)", N, K);
  G256 p{};
  p.A = A; p.lda = lda; p.a_lo_off = a_lo_off; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.c_lo_off = c_lo_off;
  p.w_pair_rows = paired ? N : 0; p.M = (int)a_rows_total; p.N = N; p.K = K;
  p.g_off = off; p.g_cnt = cnt; p.w_gstride = w_gstride; p.a_rows = a_rows; p.n_groups = n_groups;
  p.tile_g = tile_g; p.tile_m0 = tile_m0; p.n_tiles = n_tiles; p.max_mtiles = max_mtiles;
  const int rc = g256_launch(p, epi, 1, mn_stream(stream));
  if (rc < 0) return rc;
  MN_CHECK_LAUNCH("mn_gemm256_grouped_tiles");
  return MN_OK;
}

// ---- fp8-MFMA regime: the stand-alone entry (mingnative.h section 8) -------------------------------------------------------------
// C = epilogue( (A8 . a_scale) (W8 . w_scale)^T + bias ): A8 e4m3 [M, K] with one fp32 scale per row, W8 e4m3 [N or 2N, K] likewise.
//   swiglu == 0: C fp32 [M, N] (ldc); ksplit > 1: slice z writes its partial product to C + z * M * N (ldc must be N), bias in slice 0
//   swiglu == 1: W8 holds 2N rows (gate rows, then up rows), C bf16 [M, N] = silu(gate) * up
// Returns the number of split-K slices used (>= 1) or a negative error.
extern "C" int mn_gemm256_f8(const uint8_t* A, int64_t lda, const float* a_scale, const uint8_t* W, int64_t ldw, const float* w_scale,
                             const uint16_t* bias, void* C, int64_t ldc, int M, int N, int K, int swiglu, int ksplit, void* stream) {
  MN_CHECK_ARG(A && W && C && a_scale && w_scale && ksplit >= 1 && (!swiglu || ksplit == 1), "mn_gemm256_f8: bad args");
  MN_CHECK_ARG(M >= 1 && N >= 4 && (N % 4) == 0 && K >= 2 * BK && (K % (2 * BK)) == 0 && (lda % 16) == 0 && (ldw % 16) == 0 &&
                   (((uintptr_t)A | (uintptr_t)W) & 15) == 0 && (int64_t)M * lda < ((int64_t)1 << 32) &&
                   (int64_t)(swiglu ? 2 : 1) * N * ldw < ((int64_t)1 << 32) && (((uintptr_t)w_scale) & 15) == 0 &&
                   g256_out_ok(C, ldc, 0, bias, !swiglu) && (ksplit == 1 || ldc == N),
               "mn_gemm256_f8: unsupported shape M=%d N=%d K=%d (K %% 128 == 0, N %% 4 == 0, 16-byte rows and scales, < 4 GiB operands)", M, N, K);
  G256 p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.ldw = ldw; p.bias = bias;
  p.C = C; p.ldc = ldc; p.c_zstride = (int64_t)M * N; p.M = M; p.N = N; p.K = K; p.w_pair_rows = swiglu ? N : 0;
  p.f8 = 1; p.a_scale = a_scale; p.w_scale = w_scale;
  const int nz = g256_launch(p, swiglu ? E_SWIGLU_BF16 : E_F32, ksplit, mn_stream(stream));
  if (nz < 0) return nz;
  MN_CHECK_LAUNCH("mn_gemm256_f8");
  return nz;
}

// wide_glue.h — the row kernel between two wide-row GEMMs (gemm256.hip) of the lock-step generation path.
//
// With hundreds of rows in flight every nn.Linear of the decode path is a gemm256 launch whose operand is a bf16 hi/lo
// pair; everything the reference does between two Linears (residual adds, gated residuals, RMSNorm / LayerNorm, adaLN
// modulate, the MoE weighted combine, the input projection of the RF net) is ONE launch of this kernel: one workgroup
// per row, 4 columns per thread, row statistics by a block reduction, output as fp32 and / or as the next GEMM's
// hi/lo operand.  Call sites cite the reference lines they replace.
#pragma once
#include "common.h"

#define MN_TP_WAIT_MS_DEFAULT 30000u   // wall-time bound of a tensor-parallel arrival wait (mn_tp_comm.wait_ms = 0)

struct WideGlue {
  // ---- source value v[m, :] ----
  const float* h; int64_t ldh;                         // residual stream row (used when neither x nor xin is given)
  const float* x; int64_t ldx; int x_row_div;          // v = x[(m / x_row_div) * ldx]        (stack input; ldx 0 broadcasts)
  const float* xin; int kin; const bf16_t* win; const bf16_t* bin;   // v = bin + xin[m, :kin] · win[n, :kin]   (kin <= 64)
  // ---- accumulate ----
  const float* P; int nz; int64_t slab; const bf16_t* pbias;         // y = sum_z P[z * slab + m * D + :] (+ pbias)
  const float* gate; int64_t ldgate;                                  // v += gate[m] * y   (else v += y)
  const float* cy; const int32_t* cpos; const float* cw; int n_slot;  // v += sum_s cw[m, s] * cy[cpos[m, s] * D + :]
  float* h_out; int64_t ldho;                          // store the updated stream (may alias h), or NULL
  // ---- normalise + modulate ----
  int norm;                                            // 0 none, 1 RMSNorm(ng), 2 LayerNorm(ng?, nb?)
  const bf16_t* ng; const bf16_t* nb; float eps;
  const float* scale; const float* shift; int64_t ldmod;              // v = v * (1 + scale[m]) + shift[m]   (both or neither)
  int act;                                             // 1: v = gelu(v) (exact erf form) before the outputs
  // ---- outputs ----
  float* out; int64_t ldo;                             // fp32 result, or NULL
  bf16_t* Y; int64_t ldy; int64_t y_lo_off;            // bf16 hi rows at Y, lo rows y_lo_off elements further (0: plain bf16, no lo rows), or NULL
  // fp8-MFMA regime (gemm256.hip F8): the row as OCP e4m3 bytes [M, D] at Y8 + one power-of-two scale per row — exactly what
  // mn_quant_fp8_rows makes of the bf16 hi row (amax over the hi values, RNE), without the extra pass; or NULL
  uint8_t* Y8; float* y8_scale;
  int M, D;
  // ---- tensor-parallel all-reduce, consumer side (tp.inl): before reading P (= this rank's inbox: nz = world slabs, one per
  // sender), row m waits until sender s's arrival flag wait_flags[s * wait_stride + m] reached wait_epoch.  The wait is bounded in
  // WALL time (wait_ticks of the 100 MHz constant clock, from mn_tp_comm.wait_ms: host-side skew between ranks — a lazy code-object load, an
  // allocator stall — is legitimate and can be long; a dead peer is not); on expiry the row sets wait_err AND poisons its outputs
  // with NaN instead of consuming stale slabs, so a missed all-reduce can never pass for a result.
  const uint32_t* wait_flags; int wait_n; int64_t wait_stride; uint32_t wait_epoch; uint32_t* wait_err; uint64_t wait_ticks;
  // two-shot all-reduce (tp.inl): the inbox holds the REDUCED row in `nz` column pieces of gather_cols columns, piece z in slab z at
  // [m * gather_cols, + gather_cols) — the pieces are concatenated, not summed.  0: the slabs are summed (one-shot, split-K).
  int gather_cols;
};

namespace {

__global__ __launch_bounds__(1024) void wide_glue_kernel(const WideGlue p) {
  __shared__ float red[32];
  __shared__ float xs[64];
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int m = blockIdx.x, col = threadIdx.x * 4, D = p.D;
  const bool act = col < D;
  int dead = 0;                                          // a sender's flag never came: every output of this row becomes NaN
  if (p.wait_flags) {                                    // uniform per launch
    int expired = 0;
    if ((int)threadIdx.x < p.wait_n) {
      const uint32_t* f = p.wait_flags + (int64_t)threadIdx.x * p.wait_stride + m;
      const uint64_t t0 = wall_clock64();                // constant 100 MHz, independent of the shader clock
      // relaxed polls, ONE acquire once the flag is there (an acquire load per poll would drop this CU's L1 on every iteration)
      while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - p.wait_epoch) < 0) {
        if (wall_clock64() - t0 > p.wait_ticks) {         // a peer died or the launch orders diverged
          if (p.wait_err) atomicExch(p.wait_err, 0x100u | (unsigned)threadIdx.x);
          expired = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      // system scope (the senders may be other GPUs): drops this CU's L1 — the L1 is per CU, so the waves of this workgroup that
      // did not poll read fresh lines after the barrier below (MI355X_MICROARCH.md: one poll, one acquire, barrier, plain loads)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    dead = __syncthreads_or(expired);
  }
  if (p.xin) {                                           // stage the row of the tiny-K projection
    if (threadIdx.x < p.kin) xs[threadIdx.x] = p.xin[(int64_t)m * p.kin + threadIdx.x];
    __syncthreads();
  }
  f4 v = {0.f, 0.f, 0.f, 0.f};
  // modulation and norm parameters of this thread's columns are requested with the row, not behind the reductions' barriers
  f4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
  u2 ng2 = {0u, 0u}, nb2 = {0u, 0u};
  if (act) {
    if (p.scale) {
      sc = *reinterpret_cast<const f4*>(p.scale + (int64_t)m * p.ldmod + col);
      sh = *reinterpret_cast<const f4*>(p.shift + (int64_t)m * p.ldmod + col);
    }
    if (p.norm && p.ng) ng2 = *reinterpret_cast<const u2*>(p.ng + col);
    if (p.norm == 2 && p.nb) nb2 = *reinterpret_cast<const u2*>(p.nb + col);
  }
  if (act) {
    if (p.xin) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = p.bin ? bf16_to_f32(p.bin[col + j]) : 0.f;
        const bf16_t* wr = p.win + (int64_t)(col + j) * p.kin;
        for (int k = 0; k < p.kin; ++k) a = fmaf(bf16_to_f32(wr[k]), xs[k], a);
        o[j] = a;
      }
      v = f4{o[0], o[1], o[2], o[3]};
    } else if (p.x) {
      v = *reinterpret_cast<const f4*>(p.x + (int64_t)(m / p.x_row_div) * p.ldx + col);
    } else {
      v = *reinterpret_cast<const f4*>(p.h + (int64_t)m * p.ldh + col);
    }
    if (p.P) {
      f4 y = {0.f, 0.f, 0.f, 0.f};
      if (p.pbias) y = f4{bf16_to_f32(p.pbias[col]), bf16_to_f32(p.pbias[col + 1]), bf16_to_f32(p.pbias[col + 2]), bf16_to_f32(p.pbias[col + 3])};
      const float* pp = p.P + (int64_t)m * D + col;
      int z = 0;
      if (p.gather_cols) {                               // reduced row, owner by owner: this thread's four columns sit in one piece
        const int zc = col / p.gather_cols;
        y += *reinterpret_cast<const f4*>(p.P + (int64_t)zc * p.slab + (int64_t)m * p.gather_cols + (col - zc * p.gather_cols));
        z = p.nz;
      }
      for (; z + 4 <= p.nz; z += 4) {                    // independent 16-byte loads in flight
        const f4 a = *reinterpret_cast<const f4*>(pp + (z + 0) * p.slab), b = *reinterpret_cast<const f4*>(pp + (z + 1) * p.slab);
        const f4 c = *reinterpret_cast<const f4*>(pp + (z + 2) * p.slab), d = *reinterpret_cast<const f4*>(pp + (z + 3) * p.slab);
        y += (a + b) + (c + d);
      }
      for (; z < p.nz; ++z) y += *reinterpret_cast<const f4*>(pp + z * p.slab);
      if (p.gate) v += *reinterpret_cast<const f4*>(p.gate + (int64_t)m * p.ldgate + col) * y;
      else v += y;
    }
    if (p.cy) {
      for (int s = 0; s < p.n_slot; s += 4) {            // four slots of independent loads in flight
        f4 y[4];
        float wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int sj = s + j < p.n_slot ? s + j : s;
          wv[j] = s + j < p.n_slot ? p.cw[(int64_t)m * p.n_slot + sj] : 0.f;
          y[j] = f4{0.f, 0.f, 0.f, 0.f};
          if (wv[j] != 0.f) y[j] = *reinterpret_cast<const f4*>(p.cy + (int64_t)p.cpos[(int64_t)m * p.n_slot + sj] * D + col);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v += wv[j] * y[j];
      }
    }
    if (dead) { const float q = __builtin_nanf(""); v = f4{q, q, q, q}; }
    if (p.h_out) *reinterpret_cast<f4*>(p.h_out + (int64_t)m * p.ldho + col) = v;
  }
  if (p.norm == 1) {
    const float ss = block_sum(act ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : 0.f, red);
    const float rstd = rsqrtf(ss / (float)D + p.eps);
    if (act) v = f4{v.x * rstd * bf16lo_to_f32(ng2.x), v.y * rstd * bf16hi_to_f32(ng2.x),
                    v.z * rstd * bf16lo_to_f32(ng2.y), v.w * rstd * bf16hi_to_f32(ng2.y)};
  } else if (p.norm == 2) {
    const float mean = block_sum(act ? (v.x + v.y) + (v.z + v.w) : 0.f, red) / (float)D;
    float ss = 0.f;
    if (act) { const f4 d = v - mean; ss = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w; }
    const float rstd = rsqrtf(block_sum(ss, red) / (float)D + p.eps);
    if (act) {
      float o[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
      const float gq[4] = {bf16lo_to_f32(ng2.x), bf16hi_to_f32(ng2.x), bf16lo_to_f32(ng2.y), bf16hi_to_f32(ng2.y)};
      const float bq[4] = {bf16lo_to_f32(nb2.x), bf16hi_to_f32(nb2.x), bf16lo_to_f32(nb2.y), bf16hi_to_f32(nb2.y)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (p.ng) o[j] *= gq[j];
        if (p.nb) o[j] += bq[j];
      }
      v = f4{o[0], o[1], o[2], o[3]};
    }
  }
  if (!act && !p.Y8) return;                             // (Y8: every thread of the workgroup takes part in the row-amax reduction)
  if (p.scale) v = v * (1.0f + sc) + sh;
  if (p.act == 1) v = f4{gelu_erf_f(v.x), gelu_erf_f(v.y), gelu_erf_f(v.z), gelu_erf_f(v.w)};
  if (act && p.out) *reinterpret_cast<f4*>(p.out + (int64_t)m * p.ldo + col) = v;
  uint32_t h0 = 0, l0 = 0, h1 = 0, l1 = 0;
  if (act) { split_pk_bf16(v.x, v.y, h0, l0); split_pk_bf16(v.z, v.w, h1, l1); }
  if (act && p.Y) {
    bf16_t* yr = p.Y + (int64_t)m * p.ldy + col;
    *reinterpret_cast<u2*>(yr) = u2{h0, h1};
    if (p.y_lo_off) *reinterpret_cast<u2*>(yr + p.y_lo_off) = u2{l0, l1};
  }
  if (p.Y8) {                                            // uniform per launch
    const float a0 = bf16lo_to_f32(h0), a1 = bf16hi_to_f32(h0), a2 = bf16lo_to_f32(h1), a3 = bf16hi_to_f32(h1);
    const float amax = block_max(act ? fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fmaxf(fabsf(a2), fabsf(a3))) : 0.f, red);
    // scale = 2^es with amax / 2^es in (224, 448]  (fp8_ops.hip pow2_scale_for: the weight format's rule)
    const uint32_t ua = __float_as_uint(amax);
    float sc8 = 1.0f;
    if ((ua & 0x7fffffffu) != 0u) {
      int es = (int)(ua >> 23) - 127 - 8 + ((ua & 0x7fffffu) > 0x600000u ? 1 : 0);
      es = es < -126 ? -126 : (es > 127 ? 127 : es);
      sc8 = __uint_as_float((uint32_t)(es + 127) << 23);
    }
    const float inv = 1.0f / sc8;                        // exact: a power of two
    if (threadIdx.x == 0) p.y8_scale[m] = sc8;
    if (act) {
      int q = 0;
      q = __builtin_amdgcn_cvt_pk_fp8_f32(a0 * inv, a1 * inv, q, false);
      q = __builtin_amdgcn_cvt_pk_fp8_f32(a2 * inv, a3 * inv, q, true);
      *reinterpret_cast<uint32_t*>(p.Y8 + (int64_t)m * D + col) = (uint32_t)q;
    }
  }
}

inline bool wide_glue_ok(int D) { return D >= 4 && (D % 4) == 0 && D <= 4096; }

inline void wide_glue(const WideGlue& g, hipStream_t st) {
  const int threads = ((g.D / 4 + 63) / 64) * 64;
  hipLaunchKernelGGL(wide_glue_kernel, dim3(g.M), dim3(threads), 0, st, g);
}

}  // namespace

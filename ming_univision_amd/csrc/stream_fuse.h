// stream_fuse.h — internal interface of the fused form of the K-slice streaming launch (stream_mfma.hip) used by the RF ResBlock
// chain at <= 4 rows (engine.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

enum { FUSE_NONE = 0, FUSE_SWIGLU = 2, FUSE_RMSNORM = 3 };

// FUSE_SWIGLU: the launch builds its activation image itself, x[m, k] = silu(pb[k] + sum_z pP[z][m][k]) * (pb[K + k] + sum_z pP[z][m][K + k]),
// from the slabs pP [pnz][M][2 * K] of the previous launch and its bias pb [2 * K] (may be NULL).
// FUSE_RMSNORM: x[m, :] = RMSNorm(h[m, :]; norm_w, eps) (modeling_bailing_moe.py:131-136) from the fp32 residual stream h [M][K]: every
// workgroup reads the M whole rows for the statistic (a few KiB of L2) and normalises its K-slice — the decoder chain's QKV launch
// without the one-workgroup-per-row glue launch in front of it.
struct StreamFuse {
  const float* pP; int pnz; const bf16_t* pb;
  const float* h; const bf16_t* norm_w; float eps;
};

constexpr int FUSE_MAX_ROWS = 4;

// Can this shape run fused (row count, slice length vs threads, slab count of the previous launch)?
bool stream_fused_ok(int wfmt, int M, int Ntot, int K, int prev_nz);
// Launch: W = bf16 [Ntot][K], or e4m3 bytes + wscale (wfmt != 0).  P [nz][M][Ntot], nz = the plain launch's slice count.  Returns nz (< 0: error).
int stream_fused(int wfmt, const void* W, const float* wscale, float* P, int M, int Ntot, int K, const StreamFuse& f, void* stream);
// P [nz][M][Ntot] = RMSNorm(h; norm_w, eps) W^T in K slices (bf16 W): the glue launch's arithmetic (llm_glue_kernel; the statistic is summed in the same order), equal to 1.6e-6 of the hidden state over 28 layers.
bool stream_rmsnorm_ok(int M, int Ntot, int K);
int stream_rmsnorm(const bf16_t* W, float* P, int M, int Ntot, int K, const float* h, const bf16_t* norm_w, float eps, void* stream);

// ---- K-complete launches of the RF ResBlock chain at <= 2 rows (stream_kc.hip): whole output tiles per workgroup, K split over its
// waves — no split-K slabs, so the residual + LayerNorm glue launch between two blocks disappears (two launches per block).
bool rf_kc_ok(int wfmt, int M, int w, int hid);
// Y3 [2][M][hid] bf16 (hi rows, lo rows) = split( silu(g) * u ),  (g, u) = LayerNorm(h; ln_g, ln_b)(1 + scale) + shift  @ W12^T + b12
int rf_w12_kc(int wfmt, const float* h, int M, int w, int hid, const bf16_t* ln_g, const bf16_t* ln_b, const float* shift, const float* scale,
              int64_t ldmod, const void* W12, const float* s12, const bf16_t* b12, bf16_t* Y3, void* stream);
// h[m, n] += gate[m, n] * (Y3 @ W3^T + b3)[m, n]   in place
int rf_w3_kc(int wfmt, const bf16_t* Y3, int M, int w, int hid, const void* W3, const float* s3, const bf16_t* b3, const float* gate,
             int64_t ldmod, float* h, void* stream);

// ... and every block of one Euler step as ONE persistent launch (grid barrier between the phases, the next phase's first weight chunks
// requested before the wait).  rf_persist_ok: a K-complete shape, one workgroup per CU, the stream not being captured.  `bar`:
// RF_PERSIST_BAR_WORDS words of device memory zeroed by the caller; epoch0 = 0, 64, 128, ... for successive launches on it.
// Arithmetic and its order are those of the two launches: same bits.
constexpr int RF_PERSIST_MAX_BLOCKS = 16;
constexpr int RF_PERSIST_BAR_WORDS = 320;
bool rf_persist_ok(int wfmt, int M, int w, int hid, void* stream);
// The whole sampler in ONE launch: per Euler step a boundary phase (CFG + Euler update of the ODE state — replicated in every workgroup's
// LDS — and the input projection), the blocks, and a final-layer phase (LayerNorm-modulate + final linear, one output column per
// workgroup); the last update writes `latent`.  Replaces per step: rf_step_boundary_kernel, the LN glue and the final streaming launch.
struct RfSamplerTail {
  int steps, T, rpi, n_images;
  int64_t mod_step;                                 // floats between two Euler steps' modulations
  const bf16_t* in_w; const bf16_t* in_b; const bf16_t* fin_w; const bf16_t* fin_b;
  const float* noise; float temperature, text_cfg, image_cfg;
  float* v;                                         // [M][T] scratch
  float* latent;                                    // [n_images][T]
};
bool rf_sampler_persist_ok(int M, int w, int hid, int depth, int T, int rpi, int n_images);
int rf_blocks_persist(int wfmt, float* h, bf16_t* Y3, int M, int w, int hid, const float* mod, int64_t ldmod, int nblk,
                      const void* const* W12, const float* const* s12, const bf16_t* const* b12, const bf16_t* const* ln_g,
                      const bf16_t* const* ln_b, const void* const* W3, const float* const* s3, const bf16_t* const* b3,
                      unsigned* bar, unsigned epoch0, const RfSamplerTail* whole, void* stream);

// ---- the MoE down projection of a 1- / 2-row step with the segments spread over the waves of a workgroup (moe_down.hip):
//   out[b][n] = res[b][n] + sum_s tw[b, s] * hmid[b][s * I ..] . W[ti[b, s]][n][..]      (bf16 weights, or e4m3 bytes + row scales)
bool moe_down_ok(int wfmt, int n_slot, int H, int I);
int moe_down_rows(int wfmt, const float* hmid, int64_t ld_hmid, const void* W, int64_t w_stride, const float* wscale, int64_t wscale_stride,
                  const int32_t* ti, const float* tw, const float* res, int64_t ld_res, float* out, int64_t ld_out, int batch, int H, int I,
                  int n_slot, void* stream, const float* P = nullptr, int nz = 0, int64_t slab = 0);

// ---- router + expert gate/up of a 1-row decode step in ONE launch (moe_gate_up.hip): every workgroup routes its row itself, wave s
// streams hidden units of the slot-s expert.  Writes hmid [batch][n_slot * I], and (one workgroup per row) ti / tw / logits.  P != NULL
// (the decoder chain at 2 rows): the row is h + the nz partial slabs P [nz][batch][H] of the attention output projection
// (h is not updated: moe_down_rows adds the same slabs to its residual).
bool moe_gate_up_ok(int wfmt, int H, int I, int E, int top_k, int n_shared);
int moe_gate_up_routed(int wfmt, const float* h, int64_t ldh, const bf16_t* norm_w, float eps, const bf16_t* gate_w, const void* W, int64_t w_stride,
                       const float* wscale, int64_t wscale_stride, int batch, int H, int I, int E, int top_k, int n_shared, int norm_topk_prob,
                       float* hmid, int64_t ld_hmid, int32_t* ti, float* tw, float* logits, const float* P, int nz, int64_t slab, void* stream);

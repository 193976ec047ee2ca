/*
 * mingnative.h — C ABI of libmingnative.so: the MI355X (gfx950) native hot path of
 * Ming-UniVision (MingTok-Vision tokenizer -> Bailing-MoE next-token forward ->
 * rectified-flow SwiGLU head -> MingTok decode).
 *
 * The reference has no FFI: the path is PyTorch module calls that dispatch into
 * third-party GPU kernels (cuBLAS/cuDNN/flash-attn).  Every entry
 * point below replaces one of those call sites; the reference file:line is cited
 * per function (paths relative to the reference repo root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - All pointers are DEVICE pointers unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - Launchers never allocate, never synchronise, never throw: they enqueue work
 *     on `stream` and return 0, or a negative MN_E* code (see mn_last_error()).
 *   - Weights are bf16 (uint16_t bit patterns) in the reference's nn.Linear layout
 *     [out_features, in_features] row-major unless stated.  Small-row ("decode")
 *     activations are fp32; batched (MFMA) activations are bf16 with fp32 accumulate.
 */
#ifndef MINGNATIVE_H
#define MINGNATIVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MN_VERSION 124 /* 0.1.24: mn_persist_set_status_word; mn_tp_allreduce PUSH|REDUCE above the one-shot row limit implies GATHER.  0.1.23: ABI guards mn_sizeof_* / mn_struct_layout.  0.1.20: fp8 weight mode (section 7): wfmt / row-scale fields at the END of mn_skinny_args, mn_rf_head,
                          mn_llm and mn_llm_tp (zero = bf16: callers of 0.1.10 that zero-fill the structs are unchanged),
                          mn_quant_fp8_rows / mn_dequant_fp8_rows / mn_stream_mfma_w8 / mn_stream_mfma_grouped_w8 */

/* Only the entry points declared here are exported (the library is built with -fvisibility=hidden; tests/test_host_logic.py
 * holds `nm -D` to exactly this list).  The launchers keep no mutable process state: launch plans are pure functions of the
 * arguments, so calls are re-entrant per stream.  The A/B hooks of tools/ live in libmingnative_dev.so (mingnative_dev.h). */
#if defined(__GNUC__)
#define MN_API __attribute__((visibility("default")))
#else
#define MN_API
#endif

enum {
  MN_OK = 0,
  MN_EINVAL = -1,   /* bad argument (shape / alignment / enum) */
  MN_ELAUNCH = -2,  /* hip launch failure, see mn_last_error() */
  MN_ENOSPACE = -3  /* workspace too small */
};

MN_API int mn_version(void);
MN_API const char* mn_last_error(void);
/* Number of compute units of the current device (host query, cached). */
MN_API int mn_num_cus(void);

/* ABI guards (round 5): the binding side (ctypes Structures, cgo / JNI struct mirrors) can check that its layout of every struct
 * below is the library's — a field added on one side only would otherwise corrupt calls silently.  mn_sizeof_* = sizeof(the struct);
 * mn_struct_layout(which, offsets, cap) writes offsetof(every field, in declaration order) into offsets[0 .. min(n, cap)) and
 * returns the field count n (MN_EINVAL for an unknown `which`). */
enum mn_struct_id { MN_STRUCT_SKINNY_ARGS = 0, MN_STRUCT_RF_HEAD = 1, MN_STRUCT_LLM = 2, MN_STRUCT_SEMDEC = 3, MN_STRUCT_TP_COMM = 4,
                    MN_STRUCT_LLM_TP = 5 };
MN_API size_t mn_sizeof_skinny_args(void);
MN_API size_t mn_sizeof_rf_head(void);
MN_API size_t mn_sizeof_llm(void);
MN_API size_t mn_sizeof_semdec(void);
MN_API size_t mn_sizeof_tp_comm(void);
MN_API size_t mn_sizeof_llm_tp(void);
MN_API int mn_struct_layout(int which, size_t* offsets, int cap);

/* ------------------------------------------------------------------------------------------
 * 1. Skinny GEMM (M <= 8 rows): out = epilogue( prologue(x) @ W^T + bias )
 *    Weight-streaming, HBM-bound.  Replaces every nn.Linear call with <= 8 rows on the
 *    decode path: modeling_bailing_moe.py:760,824 (query_key_value / dense), :483-484
 *    (expert / shared-expert MLPs), :1551 (lm_head), :1571-1574 (vis_head);
 *    diff_loss_rf_swiglu.py:27-34,194-198,263-266,282-286,324-326 (RF head);
 *    mingtok layers/attention.py:49-51, layers/swiglu_ffn.py:27-28 (semantic-decoder decode step);
 *    modeling_bailingmm.py:111-115 (linear_proj).
 * ------------------------------------------------------------------------------------------ */
enum mn_prologue {
  MN_PRO_NONE = 0,
  MN_PRO_SILU = 1,      /* x' = silu(x) */
  MN_PRO_ADD_SILU = 2,  /* x' = silu(x + pro_a)          (adaLN input SiLU(t_emb + c), diff_loss_rf_swiglu.py:263-266,376) */
  MN_PRO_RMSNORM = 3,   /* x' = x * rsqrt(mean(x^2)+eps) * ln_g   (BailingMoeRMSNorm, modeling_bailing_moe.py:131-136) */
  MN_PRO_LN = 4,        /* x' = LayerNorm(x; ln_g, ln_b, eps)     (nn.LayerNorm eps=1e-6) */
  MN_PRO_LN_MOD = 5     /* x' = LayerNorm(x; ln_g?, ln_b?, eps) * (1 + pro_b) + pro_a   (modulate, diff_loss_rf_swiglu.py:184-185,270,290) */
};
enum mn_epilogue {
  MN_EPI_NONE = 0,
  MN_EPI_SILU = 1,
  MN_EPI_GELU = 2,        /* exact erf GELU (nn.GELU default) */
  MN_EPI_SWIGLU = 3,      /* out[n] = silu(y[n]) * y[n + N]; W and bias hold 2N rows (chunk(2), swiglu_ffn.py:30-34) */
  MN_EPI_RESID = 4,       /* out = res + y */
  MN_EPI_RESID_GATE = 5   /* out = res + gate * y   (ResBlock, diff_loss_rf_swiglu.py:272) */
};

typedef struct mn_skinny_args {
  const float* x;      int64_t ldx;    /* [M, K] fp32 (per batch entry) */
  const uint16_t* w;   int64_t ldw;    /* bf16 [N or 2N, K]; row stride ldw elements */
  const uint16_t* bias;                /* bf16 [N or 2N] or NULL */
  float* out;          int64_t ldo;    /* [M, N] fp32 */
  int32_t M, N, K;                     /* 1 <= M <= 8; K % 8 == 0 (K is per segment when nseg > 1) */
  int32_t prologue, epilogue;
  const float* pro_a;  int64_t ld_pro_a; /* ADD_SILU: addend [K] (ld 0) or [M,K]; LN_MOD: shift [M,K] */
  const float* pro_b;  int64_t ld_pro_b; /* LN_MOD: scale [M,K] */
  const uint16_t* ln_g; const uint16_t* ln_b; /* bf16 [K] norm gain / bias (NULL = none) */
  float eps;
  const float* res;    int64_t ldres;  /* RESID / RESID_GATE */
  const float* gate;   int64_t ldgate; /* RESID_GATE */
  /* Batching over independent problems that differ in weights (MoE experts):
   * entry b uses W + w_index[b]*w_batch_stride (w_index NULL -> b), x + (b / x_batch_div)*x_batch_stride,
   * out + b*out_batch_stride, res + b*res_batch_stride.  batch <= 0 means 1. */
  int32_t batch; const int32_t* w_index; int64_t w_batch_stride;
  int64_t x_batch_stride; int32_t x_batch_div; int64_t out_batch_stride; int64_t res_batch_stride;
  /* K-segments (MoE down-projection summed over the experts of one token):
   * y = sum_s seg_scale[b*nseg+s] * x[:, s*K:(s+1)*K] @ W[seg_index[b*nseg+s]]^T ; nseg <= 0 means 1. */
  int32_t nseg; const int32_t* seg_index; const float* seg_scale; int64_t seg_w_stride;
  /* Scratch for the matrix-core route (M >= 5, batched generation): such launches run as prologue (activations
   * split into bf16 hi + lo) -> weight-streaming MFMA kernel over K slices -> reduce + epilogue, and need
   * mn_skinny_workspace_bytes(M, N, K, epilogue) bytes.  May be NULL for M <= 8 (fp32-FMA kernel is used). */
  void* ws; size_t ws_bytes;
  /* fp8 weights (section 7): wfmt = MN_W_FP8_E4M3 -> `w` points to e4m3 bytes [N or 2N, K] (K % 16 == 0; strides count bytes) and
   * wscale to one fp32 scale per weight row.  M == 1 with prologue NONE and epilogue NONE / SWIGLU / RESID — the expert launches of a
   * 1- or 2-row decode step, batch and K-segment forms included — runs the one-row fp8 kernel (skinny_w8.hip): batch entry b reads its
   * scales at wscale + w_index[b] * wscale_batch_stride, segment s at + seg_index[s] * wscale_seg_stride.  Everything else (1..64
   * rows, dense weights with ldw == K, no batch / segments) takes the matrix-core route and needs ws (mn_skinny_workspace_bytes_w8). */
  int32_t wfmt; const float* wscale; int64_t wscale_batch_stride; int64_t wscale_seg_stride;
} mn_skinny_args;

/* 1 <= M <= 64 (batch / nseg forms: M <= 8). */
MN_API int mn_skinny_gemm(const mn_skinny_args* args, void* stream);
MN_API size_t mn_skinny_workspace_bytes(int M, int N, int K, int epilogue);
MN_API size_t mn_skinny_workspace_bytes_w8(int M, int N, int K, int epilogue);
MN_API size_t mn_skinny_workspace_bytes_wq(int wfmt, int M, int N, int K, int epilogue);   /* any weight format (section 7) */

/* ------------------------------------------------------------------------------------------
 * 2. MoE router: RMSNorm + gate GEMV + fp32 softmax + top-k + renormalise, with the
 *    image-gate override on rows flagged by image_mask.
 *    Replaces BailingMoeGate.forward (modeling_bailing_moe.py:505-520) and the multi-gate blend
 *    of BailingMoeSparseMoeBlock.forward (:565-592).  M <= 64, num_experts <= 64.
 *      x [M,H] fp32 (pre-norm residual stream), norm_w bf16 [H]
 *      gate_w / image_gate_w bf16 [E,H]; image_mask uint8 [M] or NULL
 *      -> x_norm [M,H] fp32 (the normalised rows, input of the experts)
 *         topk_idx int32 [M, n_slot], topk_w fp32 [M, n_slot] where
 *         n_slot = top_k + n_shared_slots; the trailing shared slots are filled with
 *         (E + j, 1.0) so that shared experts ride the same grouped GEMV.
 * ------------------------------------------------------------------------------------------ */
MN_API int mn_moe_router(const float* x, int64_t ldx, const uint16_t* norm_w, float eps,
                  const uint16_t* gate_w, const uint16_t* image_gate_w, const uint8_t* image_mask,
                  int M, int H, int E, int top_k, int norm_topk_prob, int n_shared_slots,
                  float* x_norm, int32_t* topk_idx, float* topk_w, float* logits_ws /* [2*M*E] scratch */,
                  void* ws, size_t ws_bytes /* optional: mn_skinny_workspace_bytes(M, E, H, 0) enables the MFMA route */,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * 3. RoPE + KV-cache append, and masked GQA / MHA decode attention.
 *    Replaces apply_rotary_pos_emb + DynamicCache.update + the eager/flash attention of
 *    BailingMoeAttention.forward (modeling_bailing_moe.py:428-461, 768-812) and, with
 *    rope = 0, CausalAttention.forward + cache (mingtok layers/attention.py:138-163).
 *
 *    KV cache layout: fp32 [n_seq][2 (k,v)][n_kv_heads][t_max][head_dim].
 *    qkv [M, (n_q + 2 n_kv) * hd] fp32, heads ordered q.., k.., v.. (qkv.split, :762-764); the ViT
 *    layout [3][n_heads][hd] (attention.py:140) is the same ordering with n_q == n_kv.
 *    row_seq[m]: cache sequence of row m; row_slot[m]: cache slot to write; row_pos[m]: rotary
 *    position (cumsum(mask)-1, :1905).  cos/sin tables fp32 [n_pos, hd/2].
 *    q_out [M, n_q*hd] fp32 receives rotated (and scaled by q_scale) queries.
 *    A row whose slot is outside [0, t_max) writes no K / V (the Python surface raises before a cache fills up; the kernel
 *    never stores outside the sequence's rows of the arena).
 * ------------------------------------------------------------------------------------------ */
MN_API int mn_rope_kv_append(const float* qkv, int64_t ldqkv, int M, int n_q, int n_kv, int hd,
                      int rope, const float* cos_tab, const float* sin_tab,
                      const int32_t* row_seq, const int32_t* row_slot, const int32_t* row_pos,
                      float q_scale, float* q_out, float* kv_cache, int64_t t_max, void* stream);
/*    3D rotary variant (BailingMoe3DRotaryEmbedding :413-425 + apply_multimodal_rotary_pos_emb :463-469, the
 *    `rope_scaling.type == "3D"` branch at :780-782): row_pos is [3][M] = temporal, height, width positions; rotary
 *    frequency i (of each half of the head) follows the t stream for i < sec_t, the h stream for the next sec_h, the
 *    w stream for the rest (mrope_section [16, 24, 24] at hd = 128).  sec_t == 0 is the Legacy rotary above.  With
 *    equal t/h/w positions the result is bit-identical to Legacy. */
MN_API int mn_rope_kv_append_3d(const float* qkv, int64_t ldqkv, int M, int n_q, int n_kv, int hd,
                         int rope, const float* cos_tab, const float* sin_tab,
                         const int32_t* row_seq, const int32_t* row_slot, const int32_t* row_pos, int sec_t, int sec_h,
                         float q_scale, float* q_out, float* kv_cache, int64_t t_max, void* stream);

/*    out[m] = softmax(q[m] . K[seq]^T + mask) V[seq] over keys j < row_len[m] with
 *    key_mask[m*ld_mask + j] != 0 (key_mask NULL = all ones).  A row with every key masked gets the mean of V over its keys
 *    [0, row_len): what the reference's additive finfo.min mask degenerates to (uniform softmax, modeling_bailing_moe.py:1466).
 *    q is pre-scaled.  out [M, n_q*hd] fp32.  The key range is split over workgroups
 *    (flash-decoding); workspace holds the per-split partials. */
MN_API size_t mn_attn_decode_workspace_bytes(int M, int n_q, int hd, int64_t t_max);
MN_API int mn_attn_decode(const float* q, int M, int n_q, int n_kv, int hd, const float* kv_cache, int64_t t_max,
                   const int32_t* row_seq, const int32_t* row_len, const uint8_t* key_mask, int64_t ld_mask,
                   float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * 4. Batched (MFMA) path: bf16 GEMM with fused epilogues, LayerNorm, flash attention hd=64.
 *    Replaces nn.Linear / nn.LayerNorm / flash_attn_func in the MingTok ViT blocks
 *    (layers/block.py:80-105, layers/attention.py:78-108,213-239, layers/mlp.py:34-40,
 *    layers/swiglu_ffn.py:30-34) and the step-invariant adaLN GEMM of the RF head.
 * ------------------------------------------------------------------------------------------ */
enum mn_gemm_epilogue {
  MN_GEMM_BF16 = 0,        /* C bf16 = A W^T + bias */
  MN_GEMM_BF16_GELU = 1,   /* C bf16 = gelu(A W^T + bias) */
  MN_GEMM_F32 = 2,         /* C fp32 = A W^T + bias */
  MN_GEMM_F32_RESID = 3    /* C fp32 += A W^T + bias   (residual stream accumulate) */
};
/* A bf16 [M,K] (lda), W bf16 [N,K] (ldw), bias bf16 [N] or NULL, C [M,N] (ldc). K % 32 == 0. */
MN_API int mn_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const uint16_t* bias,
                 void* C, int64_t ldc, int M, int N, int K, int epilogue, void* stream);

/* Split-K form for few-row weight-streaming problems: slice z of the K range writes partials[z][M][N] (fp32);
 * returns the number of slices used (>= 1) or a negative error. */
MN_API int mn_gemm_bf16_splitk(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, float* partials,
                        int M, int N, int K, int ksplit, void* stream);

/* ---- many-token (prefill) operators of the Bailing-MoE decoder, bf16 MFMA path -------------------------------
 * Replace, for q_len > 16: BailingMoeRMSNorm (modeling_bailing_moe.py:131-136), apply_rotary_pos_emb +
 * DynamicCache.update (:428-461, :789), the flash-attn varlen prefill (:946-1007), BailingMoeGate (:505-520) and
 * moe_infer's argsort / per-expert loop / un-permute / weighted sum (:608-639). */
MN_API int mn_rmsnorm_bf16(const float* x, int64_t ldx, const uint16_t* g, float eps, uint16_t* y, int64_t ldy, int M, int D,
                    void* stream);
/* qkv fp32 [T, (n_q + 2 n_kv) hd]; token t gets rotary position pos[t] and cache slot slot0 + t of ONE sequence:
 * kv_seq fp32 [2][n_kv][t_max][hd].  q_out bf16 [T, n_q, hd] (rotated, scaled). */
MN_API int mn_rope_kv_prefill(const float* qkv, int64_t ldqkv, int T, int n_q, int n_kv, int hd, const float* cos_tab,
                       const float* sin_tab, const int32_t* pos, int slot0, float q_scale, uint16_t* q_out,
                       float* kv_seq, int64_t t_max, void* stream);
/* Flash attention, head_dim 128, GQA: query i of the T new tokens attends keys j <= past + i with
 * key_mask[j] != 0 (NULL = all).  Keys/values come from the fp32 arena of the sequence.  out bf16 [T, n_q*128]. */
MN_API int mn_attn_prefill_gqa_hd128(const uint16_t* q, const float* kv_seq, int64_t t_max, int n_q, int n_kv, int past, int T,
                              const uint8_t* key_mask, uint16_t* out, void* stream);
/* top-k routing from precomputed gate logits [T, E] (image-gate logits chosen where image_mask is set). */
MN_API int mn_moe_topk_logits(const float* logits_text, const float* logits_image, const uint8_t* image_mask, int T, int E,
                       int top_k, int norm_topk_prob, int n_shared_slots, int32_t* topk_idx, float* topk_w, void* stream);
/* Expert-sort of the (token, pick) pairs: counts[g], offsets[g+1], perm[sorted pos] = token,
 * slot_of[token * n_slot + pick] = sorted pos.  T * n_slot <= 65536, n_groups <= 128.  All device arrays. */
MN_API int mn_moe_sort(const int32_t* topk_idx, int T, int n_slot, int n_groups, int32_t* counts, int32_t* offsets,
                int32_t* perm, int32_t* slot_of, void* stream);
MN_API int mn_gather_rows_bf16(const uint16_t* x, int64_t ldx, const int32_t* perm, uint16_t* y, int64_t ldy, int n_rows, int D,
                        void* stream);
/* h[t] += sum_j w[t, j] * y[slot_of[t, j]]  (fp32) */
MN_API int mn_moe_combine(const float* y, int64_t ldy, const int32_t* slot_of, const float* w, int n_slot, float* h, int64_t ldh,
                   int T, int D, void* stream);
/* C fp32 [M,N] = (A_hi + A_lo) W^T + bias: activations split into bf16 hi and lo halves (A_lo starts a_lo_off elements
 * after A_hi, same row stride), both multiplied against the same W tiles in one launch — fp32-class products on the bf16
 * MFMA (used for the RF head's adaLN projections of all Euler steps, diff_loss_rf_swiglu.py:263-266, 283-286). */
MN_API int mn_gemm_bf16_hilo(const uint16_t* A_hi, int64_t lda, int64_t a_lo_off, const uint16_t* W, int64_t ldw,
                      const uint16_t* bias, float* C, int64_t ldc, int M, int N, int K, void* stream);
/* Grouped GEMM over experts: rows [off[g], off[g] + cnt[g]) of A / C use W + g * w_gstride.  off / cnt are device
 * arrays (from mn_moe_sort); m_max >= every cnt[g].  epilogue: MN_GEMM_BF16 or MN_GEMM_F32, no bias. */
MN_API int mn_gemm_bf16_grouped(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int64_t w_gstride,
                         const int32_t* off, const int32_t* cnt, int n_groups, void* C, int64_t ldc, int m_max, int N,
                         int K, int epilogue, void* stream);

/* ---- wide-row GEMM (gemm256.hip): 256 x 256 x 64 tiles, 8 waves, two long phases per K-tile with counted vmcnt ----------
 * The nn.Linear call sites above when hundreds of rows are in flight (lock-step generation of 128+ images: RF head
 * w12 / w3 / adaLN, diff_loss_rf_swiglu.py:54-72, 263-272, 283-292; MingTok batches; long-prompt prefill).
 * Needs K % 64 == 0, N % 4 == 0, 16-byte aligned rows and operands below 4 GiB (mn_gemm256_supported); C 16-byte (fp32) /
 * 8-byte (bf16) aligned with ldc % 4 == 0 and an 8-byte aligned bias (the epilogue stores 4 consecutive columns per lane).
 *   a_lo_off == 0: A bf16 [M,K].   a_lo_off != 0: A is a bf16 hi/lo pair (lo rows a_lo_off elements after the hi rows),
 *   the product is (A_hi + A_lo) W^T with both halves riding the same W tiles (a tile then covers 128 rows).
 * epilogue: enum mn_gemm_epilogue. */
MN_API int mn_gemm256_supported(int64_t lda, int64_t a_lo_off, int64_t ldw, int64_t w_rows, int M, int N, int K);
MN_API int mn_gemm256(const uint16_t* A, int64_t lda, int64_t a_lo_off, const uint16_t* W, int64_t ldw, const uint16_t* bias,
               void* C, int64_t ldc, int M, int N, int K, int epilogue, void* stream);
/* Split-K: slice z of the K range writes fp32 partials[z][M][N] (bias folded into slice 0); returns the slice count. */
MN_API int mn_gemm256_splitk(const uint16_t* A, int64_t lda, int64_t a_lo_off, const uint16_t* W, int64_t ldw,
                      const uint16_t* bias, float* partials, int M, int N, int K, int ksplit, void* stream);
/* SwiGLU-fused (swiglu_ffn.py:30-34, diff_loss_rf_swiglu.py:54-72): W12 bf16 [2*hidden, K] (gate rows then up rows),
 * b12 bf16 [2*hidden] or NULL.  Y receives silu(A Wg^T + bg) * (A Wu^T + bu) split into bf16 hi rows [M, hidden] (ldy)
 * and lo rows y_lo_off elements further — the operand layout of the next hi/lo GEMM. */
MN_API int mn_gemm256_swiglu_split(const uint16_t* A, int64_t lda, int64_t a_lo_off, const uint16_t* W12, int64_t ldw,
                            const uint16_t* b12, uint16_t* Y, int64_t ldy, int64_t y_lo_off, int M, int hidden, int K,
                            void* stream);

/* The same with a plain bf16 result Y [M, hidden] (batched bf16 path: MingTok SwiGLU blocks); a_lo_off = 0 for bf16 activations. */
MN_API int mn_gemm256_swiglu(const uint16_t* A, int64_t lda, int64_t a_lo_off, const uint16_t* W12, int64_t ldw, const uint16_t* b12,
                      uint16_t* Y, int64_t ldy, int M, int hidden, int K, void* stream);
/* Grouped form (MoE experts, modeling_bailing_moe.py:605-639, hundreds of rows in flight): group g multiplies row positions
 * [off[g], off[g] + cnt[g]) — position r reads A row a_rows[r] when a_rows != NULL (the gather of the expert-sorted order,
 * done while staging) — by W + g * w_gstride and writes rows off[g].. of C.  A is a bf16 hi/lo pair (a_lo_off > 0).
 *   swiglu == 0: C fp32 [*, N];   swiglu == 1: W_g holds 2N rows (gate rows, up rows), C = bf16 hi rows [*, N] (ldc) and
 *   lo rows c_lo_off elements further of silu(gate) * up.   off / cnt: device arrays (mn_moe_sort); cnt[g] <= m_max.
 *   A has a_rows_total rows (every a_rows entry is below it: bounds the 32-bit source offsets). */
MN_API int mn_gemm256_grouped(const uint16_t* A, int64_t lda, int64_t a_lo_off, int64_t a_rows_total, const int32_t* a_rows,
                              const uint16_t* W, int64_t ldw, int64_t w_gstride, const int32_t* off, const int32_t* cnt, int n_groups,
                              void* C, int64_t ldc, int64_t c_lo_off, int m_max, int N, int K, int swiglu, void* stream);

/* Operand of a hi/lo gemm256 launch from an fp32 row block: Y = act(norm(x)) as bf16 hi rows [M, D] (ldy) and lo rows y_lo_off
 * elements further (y_lo_off == 0: plain bf16, no lo rows) and / or as fp32 rows `out` (ldo); either may be NULL.  norm 0: none; 1: RMSNorm(g) (modeling_bailing_moe.py:131-136);
 * 2: LayerNorm(g, b optional) (mingtok layers/block.py:80-105).  act 1: exact-erf GELU (mlp.py:34-40).  D % 4 == 0, D <= 4096.
 * The fp32-class ("precise") form of the MingTok blocks and of linear_proj is built from this + mn_gemm256 on hi/lo operands. */
MN_API int mn_norm_act_split(const float* x, int64_t ldx, int norm, const uint16_t* g, const uint16_t* b, float eps, int act,
                             uint16_t* Y, int64_t ldy, int64_t y_lo_off, float* out, int64_t ldo, int M, int D, void* stream);

/* Tail of a split-K Linear that joins the fp32 residual stream, fused with the next LayerNorm (MingTok layers/block.py:80-105):
 * h[m] += sum_z P[z * slab + m * D + :] (bias already in slab 0, as mn_gemm256_splitk leaves it); if y != NULL:
 * y[m] = bf16(LayerNorm(h[m]; ln_g, ln_b optional, eps)), followed by exact-erf GELU when gelu != 0.  D % 4 == 0, D <= 4096. */
MN_API int mn_slab_resid_norm(const float* P, int nz, int64_t slab, float* h, int64_t ldh, const uint16_t* ln_g, const uint16_t* ln_b,
                       float eps, int gelu, uint16_t* y, int64_t ldy, int M, int D, void* stream);

/* mn_rope_kv_prefill for several prompt spans in one launch: span i = rows [r0_i, r0_i + len_i) of qkv / q_out (rotary position
 * pos[row]), appended to cache sequence seq_i from slot slot0 of kv_layer [n_seq_total, 2, n_kv, t_max, hd];
 * seq_tab: device int32 [n_spans][3] = (seq_i, r0_i, len_i). */
MN_API int mn_rope_kv_prefill_spans(const float* qkv, int64_t ldqkv, int n_q, int n_kv, int hd, const float* cos_tab, const float* sin_tab,
                             const int32_t* pos, int slot0, float q_scale, uint16_t* q_out, float* kv_layer, int64_t t_max,
                             const int32_t* seq_tab, int n_spans, int max_len, void* stream);

/* mn_moe_combine fused with the RMSNorm of the next consumer: h[t] += sum_s tw[t, s] * yg[slot_of[t, s]] (moe_infer,
 * modeling_bailing_moe.py:630-639); if y != NULL: y[t] = bf16(RMSNorm(h[t]; eps) * norm_w).  H % 4 == 0, H <= 4096. */
MN_API int mn_moe_combine_norm(const float* yg, const int32_t* slot_of, const float* tw, int n_slot, float* h, int64_t ldh,
                        const uint16_t* norm_w, float eps, uint16_t* y, int64_t ldy, int T, int H, void* stream);

/* GQA 4:1 flash attention (head dim 128, bottom-right causal; modeling_bailing_moe.py:848-1045) of several prompt spans in one
 * launch, K / V read from the fp32 KV arena of one layer (kv_layer [n_seq_total, 2, n_kv, t_max, 128]): span i = rows
 * [r0_i, r0_i + len_i) of q / out (bf16 [rows, n_q, 128], q RoPE'd and pre-scaled) against keys [0, past + len_i) of cache
 * sequence seq_i.  seq_tab: device int32 [n_spans][3] = (seq_i, r0_i, len_i); max_len >= every len_i; key_mask optional
 * uint8 [n_spans, mask_stride] (1 = attend). */
MN_API int mn_flash_prefill_gqa_hd128(const uint16_t* q, const float* kv_layer, int64_t t_max, int n_q, int n_kv, int past,
                               const int32_t* seq_tab, int n_spans, int max_len, const uint8_t* key_mask, int64_t mask_stride,
                               uint16_t* out, void* stream);
/* The fp32-class form (the attention of mn_llm_step_spans): q FP32 [rows, n_q, 128] (RoPE'd, pre-scaled), all operands as bf16 hi + lo
 * pairs (three MFMAs per product), fp32 softmax / accumulators.  span_tab: device int32 [n_spans][4] = (seq_i, r0_i, len_i, past_i)
 * — every span brings its own `past`.  out fp32 [rows, n_q * 128] and / or split bf16 hi rows, then lo rows split_lo_off elements
 * on (either may be NULL).  No key mask. */
MN_API int mn_flash_prefill_gqa_hd128_f32(const float* q, const float* kv_layer, int64_t t_max, int n_q, int n_kv, const int32_t* span_tab,
                                          int n_spans, int max_len, float* out, uint16_t* split, int64_t split_lo_off, void* stream);

/* mn_moe_sort plus the list of LIVE row tiles of the grouped GEMMs: tile t (t < *n_tiles) = rows [tile_m0[t], tile_m0[t] +
 * tile_rows) of group tile_g[t]; tile_g / tile_m0 hold up to T * n_slot / tile_rows + n_groups entries.  All device arrays.
 * (modeling_bailing_moe.py:608-616: the expert-count / argsort bookkeeping of moe_infer, without the host sync.) */
MN_API int mn_moe_sort_tiles(const int32_t* topk_idx, int T, int n_slot, int n_groups, int32_t* counts, int32_t* offsets, int32_t* perm,
                      int32_t* slot_of, int tile_rows, int32_t* tile_g, int32_t* tile_m0, int32_t* n_tiles, void* stream);

/* Grouped gemm256 over that tile list (tile_rows = 128 for a hi/lo A, 256 for a plain bf16 A with a_lo_off = 0): no workgroup
 * runs for an empty tile, whatever the split of the rows over the experts.  A has a_rows_total rows.
 *   epi 0: C fp32 [*, N];  1: C bf16 [*, N];  4: W_g holds 2N rows (gate, up), C = bf16 hi rows + lo rows c_lo_off further of
 *   silu(gate) * up;  6: the same as plain bf16.   max_mtiles >= sum_g ceil(cnt[g] / tile_rows). */
MN_API int mn_gemm256_grouped_tiles(const uint16_t* A, int64_t lda, int64_t a_lo_off, int64_t a_rows_total, const int32_t* a_rows,
                             const uint16_t* W, int64_t ldw, int64_t w_gstride, const int32_t* off, const int32_t* cnt, int n_groups,
                             const int32_t* tile_g, const int32_t* tile_m0, const int32_t* n_tiles, int max_mtiles, void* C,
                             int64_t ldc, int64_t c_lo_off, int N, int K, int epi, void* stream);

/* Weight-streaming MFMA kernel behind the M >= 5 route of mn_skinny_gemm: Y bf16 [2][M][K] (activations split
 * into hi rows then lo rows), W bf16 [Ntot, K] dense, P fp32 [nz][M][Ntot] K-slice partials with
 * nz = mn_stream_mfma_slices(M, Ntot, K) (the launch plan picks slices of 256..1024 k so that every wave of the
 * chip gets the same number of 16-row weight tiles); returns nz.  M <= 64 (33..64 rows run the K-loop form: 8 x 2
 * tiles per workgroup over a long K-range, x chunks double-buffered in LDS).  HBM-bound: every weight byte is read once. */
MN_API int mn_stream_mfma(const uint16_t* Y, const uint16_t* W, float* P, int M, int Ntot, int K, void* stream);
MN_API int mn_stream_mfma_slices(int M, int Ntot, int K);
/* Grouped form (MoE experts; replaces the per-token expert loop of modeling_bailing_moe.py:605-639 for 5..32
 * rows): group g of G multiplies the x rows xrows[off[g] .. off[g+1]) (identity rows when xrows == NULL) by
 * W + g * w_stride and writes rows off[g].. of P [nz][p_rows][Ntot]; Y holds y_rows hi rows then y_rows lo rows;
 * no group may exceed max_rows (<= 64) rows.  off / xrows are device arrays.  Every distinct expert is streamed
 * once for all the rows routed to it.  Returns nz = mn_stream_mfma_grouped_slices(G, max_rows, Ntot, K). */
MN_API int mn_stream_mfma_grouped(const uint16_t* Y, int y_rows, const uint16_t* W, int64_t w_stride, float* P, int p_rows,
                           const int32_t* off, const int32_t* xrows, int G, int max_rows, int Ntot, int K,
                           void* stream);
MN_API int mn_stream_mfma_grouped_slices(int G, int max_rows, int Ntot, int K);

/* y bf16 [M,D] = LayerNorm(x fp32 [M,D]; g,b bf16, eps) ; optional GELU afterwards (encoder out layer,
 * vision_transformer.py:173-178). */
MN_API int mn_layernorm_bf16(const float* x, int64_t ldx, const uint16_t* g, const uint16_t* b, float eps,
                      uint16_t* y, int64_t ldy, int M, int D, int gelu, void* stream);

/* h bf16 [M,H] = silu(x12[:, :H]) * x12[:, H:]   (x12 bf16 [M,2H]) */
MN_API int mn_swiglu_bf16(const uint16_t* x12, int64_t ldx, uint16_t* h, int64_t ldh, int M, int H, void* stream);

/* Flash attention, head_dim 64, bf16 in/out, fp32 softmax.
 * qkv bf16 [B, T, 3, n_heads, 64] (the reshape of attention.py:83,98); out bf16 [B, T, n_heads*64].
 * causal: 0 = bidirectional (Attention / MemEffAttention), 1 = causal (MemEffCausalAttention). */
MN_API int mn_attn_prefill_hd64(const uint16_t* qkv, uint16_t* out, int B, int T, int n_heads, int causal, void* stream);
/* The same attention in the fp32-class regime (within 1e-3 of the fp32 reference path): qkv FP32
 * [B, T, 3, n_heads, 64]; operands carried as bf16 hi + lo pairs, three MFMAs per product, fp32 softmax and accumulators.
 * out fp32 [B * T, n_heads * 64] and / or split bf16 [2][B * T][n_heads * 64] (hi rows, then lo rows: the projection GEMM's
 * operand); either may be NULL. */
MN_API int mn_attn_prefill_hd64_f32(const float* qkv, float* out, uint16_t* split, int B, int T, int n_heads, int causal, void* stream);

/* Small fp32 elementwise helpers of the ViT glue (all [M, D] row-major contiguous):
 *   mn_add_bcast_f32    out[i] = a[i] + b[i % period]     (+pos-embed, vision_transformer.py:222)
 *   mn_group_mean_add   out[m,c] = y[m,c] + mean_g x[m, c*G + g], G = D/C   (encoder out shortcut, :174)
 *   mn_repeat_add       out[m,n] = y[m,n] + s[m, n / (D/C)] * scale + shift   (decoder in shortcut, :375-379)
 *   mn_clamp_f32        x = clamp(x, lo, hi) in place            (modeling_mingtok.py:194) */
MN_API int mn_add_bcast_f32(const float* a, const float* b, float* out, int64_t n, int64_t period, void* stream);
MN_API int mn_group_mean_add(const float* y, const float* x, float* out, int M, int D, int Cout, void* stream);
MN_API int mn_repeat_add(const float* y, const float* s, float* out, int M, int D, int Cin, float scale, float shift, void* stream);
MN_API int mn_clamp_f32(float* x, int64_t n, float lo, float hi, void* stream);
/* The MingTok layout shuffles as single passes that write the consumer's operand directly:
 * mn_patchify_operand: PatchEmbed's conv k = s = P as a GEMM (layers/patch_embed.py:69-82): image fp32 [B,3,Hi,Wi] -> A operand bf16
 *   [B * (Hi/P) * (Wi/P), 3 P^2] (row = patch, column = (c, py, px)); y_lo_off != 0: bf16 hi rows at Y, lo rows y_lo_off elements further.
 * mn_tokens_assemble: prepare_tokens (vision_transformer.py:218-223): out[b, n] = (n < N ? tok[b N + n] : cls) + pos[n], cls LAST.
 * mn_subtoken_rearrange: "b (h w) (x y c) -> b (h x w y) c" of forward_pixel_decoder (modeling_mingtok.py:184-188), fp32, Dp = c.
 * mn_unpatchify_clamp: unpatchify 'nhwpqc->nchpwq' (vision_transformer.py:515-527) + clamp_ (modeling_mingtok.py:195): o fp32
 *   [B * hh * ww, p * p * 3] -> image [B, 3, hh p, ww p]. */
MN_API int mn_patchify_operand(const float* image, int B, int Hi, int Wi, int P, uint16_t* Y, int64_t y_lo_off, void* stream);
MN_API int mn_tokens_assemble(const float* tok, const uint16_t* cls, const float* pos, float* out, int B, int N, int D, void* stream);
MN_API int mn_subtoken_rearrange(const float* y, float* x, int B, int h, int w, int r, int Dp, void* stream);
MN_API int mn_unpatchify_clamp(const float* o, float* image, int B, int hh, int ww, int p, float lo, float hi, void* stream);

/* fp32 <-> bf16 conversion and hi/lo split helpers (elementwise, n elements). */
MN_API int mn_f32_to_bf16(const float* x, uint16_t* y, int64_t n, void* stream);
MN_API int mn_bf16_to_f32(const uint16_t* x, float* y, int64_t n, void* stream);
/* hi = bf16(x), lo = bf16(x - hi): lets an fp32 activation go through the bf16 MFMA twice at fp32-class accuracy */
MN_API int mn_f32_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * 5. Composite device-side sequences (one C call = many launches, no host sync).
 *    See the struct comments; Python builds these tables once at load time.
 * ------------------------------------------------------------------------------------------ */

/* Rectified-flow head (RectifiedFlowLoss.sample, diff_loss_rf_swiglu.py:103-181, called from
 * forward_for_image_generation_inner, modeling_bailing_moe.py:1659-1670). */
typedef struct mn_rf_head {
  int32_t w, depth, hidden, z_dim, target, steps, llm_hidden;
  /* vis_head = Linear(llm_hidden->z) + LayerNorm(z) */
  const uint16_t *vis_w, *vis_b, *vis_ln_g, *vis_ln_b;
  const uint16_t *cond_w, *cond_b;        /* cond_embed [w, z] */
  const uint16_t *in_w, *in_b;            /* input_proj [w, target] */
  const float* temb;                      /* [steps, w] fp32: time_embed(t_s * 1000), precomputed at load */
  /* all adaLN projections stacked: rows [depth*3w + 2w, w] = blocks' (shift,scale,gate) then final (shift,scale) */
  const uint16_t *ada_w, *ada_b;
  const uint16_t* const* ln_g;  const uint16_t* const* ln_b;   /* [depth] in_ln */
  const uint16_t* const* w12;   const uint16_t* const* b12;    /* [depth] [2*hidden, w] */
  const uint16_t* const* w3;    const uint16_t* const* b3;     /* [depth] [w, hidden] */
  const uint16_t *fin_w, *fin_b;          /* final_layer.linear [target, w] */
  /* fp8 weight mode (section 7; 0 = bf16).  wfmt = MN_W_FP8_E4M3: w12[b] / w3[b] point to e4m3 BYTES in the same [out, in] layout and
   * w12_scale[b] [2*hidden] / w3_scale[b] [w] hold one fp32 scale per output row; every other tensor stays bf16.  Calls then take
   * the weight-streaming route only (rows <= 64: that route is HBM-bound, fp8 halves its bytes; mn_rf_max_rows returns 64). */
  int32_t wfmt;
  const float* const* w12_scale; const float* const* w3_scale;
  /* optional (NULL = the adaLN GEMM always reads the bf16 ada_w): the stacked adaLN matrix as e4m3 bytes [depth*3w + 2w, w] + row
   * scales.  Used when all Euler steps' rows fit one streaming launch (steps * rows <= 64, i.e. <= 4 CFG rows at 16 steps: the
   * reference's call shape); ada_w must then hold the SAME values (the exact bf16 expansion) for the larger row counts. */
  const uint8_t* ada_q; const float* ada_scale;
  /* (0.1.24) arithmetic regime of the WIDE route (> 64 rows in lock-step); 0 = fp32-class (bf16 hi/lo MFMA pairs, the parity regime).
   * MN_ARITH_FP8_MFMA (section 8; needs wfmt == MN_W_FP8_E4M3 with ada_q): the ResBlock GEMMs w12 / w3 and the adaLN GEMM multiply
   * e4m3 activations (quantised per row from the bf16 operand) by the e4m3 weight bytes on the scaled fp8 MFMA — no bf16 expansion of
   * the weights, 2.2-2.6 x the hi/lo pair's rate per GEMM — a LABELLED reduced-arithmetic regime with its own stated tolerance
   * (tests/test_gpu_fp8_mfma.py); every other Linear of the head, and every route up to 64 rows, is unchanged. */
  int32_t arith;
} mn_rf_head;
enum { MN_ARITH_FP32_CLASS = 0, MN_ARITH_FP8_MFMA = 1 };

/* hidden [rows, llm_hidden] fp32 (last hidden states of the LLM step), rows = n_images x R image-major with
 * R = 1 (no CFG), 2 ([cond, uncond]) or 3 ([cond, uncond, text_uncond]) rows per image; rows <= 64.
 * noise [n_images, target] fp32; latent_out [n_images, target] fp32 (all CFG rows of an image carry the same
 * latent).  n_images = 1 is the reference's batch-size-1 call.  Workspace: mn_rf_workspace_bytes(h, rows). */
MN_API size_t mn_rf_workspace_bytes(const mn_rf_head* h, int rows);
/* (0.1.24) At <= 2 rows the sampler runs as ONE persistent launch whose phases are separated by an in-launch grid barrier (every
 * workgroup must be resident).  A barrier wait is bounded (2 s): on expiry the result is NaN in every workgroup — and, because a
 * launcher never synchronises, the library raises the CALLER's status word so the host can tell "NaN because a co-tenant process /
 * a CU mask kept the grid from being resident" from a numerical failure: register one zero-initialised uint32 of DEVICE memory per
 * process (NULL = off; the word is the only process state the library keeps besides the launch-ordering event), read it at any host
 * sync, non-zero = a wait expired since it was last cleared (0x300; clear it yourself).  Remedy: MINGNATIVE_RF_PERSIST=0 (launch chain). */
MN_API int mn_persist_set_status_word(uint32_t* device_word);
/* Rows one call accepts: 64 (weight-streaming kernels), or 2048 when every width is a multiple of 64 — then calls with more
 * than 64 rows take the wide route (each Linear a 256 x 256-tile MFMA GEMM on bf16 hi/lo operands, gemm256.hip). */
MN_API int mn_rf_max_rows(const mn_rf_head* h);
MN_API int mn_rf_sample(const mn_rf_head* h, const float* hidden, int64_t ld_hidden, int rows, int n_images,
                 const float* noise, float temperature, float text_cfg, float image_cfg, float* latent_out,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Bailing-MoE decoder stack, decode-style step for M <= 64 rows
 * (BailingMoeModel.forward, modeling_bailing_moe.py:1391-1540, with q_len rows per sequence). */
typedef struct mn_llm {
  int32_t hidden, n_layers, n_q, n_kv, head_dim, n_experts, top_k, n_shared_slots, moe_inter;
  int32_t norm_topk_prob;
  float rms_eps;
  const uint16_t* const* ln1;        /* [L] input_layernorm [H] */
  const uint16_t* const* wqkv;       /* [L] [(nq+2nkv)*hd, H] */
  const uint16_t* const* wdense;     /* [L] [H, nq*hd] */
  const uint16_t* const* ln2;        /* [L] post_attention_layernorm */
  const uint16_t* const* gate;       /* [L] [E, H] */
  const uint16_t* const* image_gate; /* [L] [E, H] or NULL entries */
  const uint16_t* const* w_gate_up;  /* [L] [E + n_shared_slots, 2*I, H]: rows 0..I-1 gate_proj, I..2I-1 up_proj */
  const uint16_t* const* w_down;     /* [L] [E + n_shared_slots, H, I] */
  const uint16_t* final_norm;        /* [H] */
  const float *cos_tab, *sin_tab;    /* [n_pos, hd/2] */
  int32_t n_pos;
  int32_t mrope_sec_t, mrope_sec_h;  /* 0, 0: Legacy rotary, row_pos [M]; else 3D rotary sections, row_pos [3][M] (t, h, w) */
  /* fp8 weight mode (section 7; 0 = bf16).  wfmt = MN_W_FP8_E4M3: w_gate_up[l] / w_down[l] point to e4m3 BYTES in the same packed
   * layout, w_gate_up_scale[l] [E + S, 2I] / w_down_scale[l] [E + S, H] hold one fp32 scale per output row of every expert;
   * attention, router and norm weights stay bf16.  Steps then take the weight-streaming route only (rows <= 64; every row count
   * runs the grouped expert kernels; mn_llm_max_rows returns 64). */
  int32_t wfmt;
  const float* const* w_gate_up_scale; const float* const* w_down_scale;
  /* (0.1.24) arithmetic regime of the WIDE route, as mn_rf_head.arith: MN_ARITH_FP8_MFMA (needs wfmt == MN_W_FP8_E4M3) runs the grouped
   * expert GEMMs (gate/up with the SwiGLU epilogue, down) on e4m3 activations x the e4m3 expert bytes (section 8) — no per-layer bf16
   * expansion of the experts; attention, router and every route up to 64 rows unchanged.  Labelled reduced arithmetic, own tolerance. */
  int32_t arith;
} mn_llm;

MN_API size_t mn_llm_workspace_bytes(const mn_llm* m, int rows, int64_t t_max);
MN_API int mn_llm_max_rows(const mn_llm* m);      /* 64, or 2048 (wide route: see mn_rf_max_rows) */
/* x fp32 in (embeddings): row m is read from x + (m / x_row_div) * ldx (ldx == 0 broadcasts one row to all M
 * rows; x_row_div = R shares one embedding between the R CFG rows of an image)
 * -> hidden_out [M,H] fp32 (after the final RMSNorm).
 * kv_cache: fp32 [n_layers][n_seq][2][n_kv][t_max][hd]; per-row int32 device arrays as in
 * mn_rope_kv_append / mn_attn_decode (row_len = row_slot + 1 is computed by the caller). */
MN_API int mn_llm_step(const mn_llm* m, const float* x, int64_t ldx, int x_row_div, int M, const uint8_t* image_mask,
                const int32_t* row_seq, const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len,
                const uint8_t* key_mask, int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max,
                float* hidden_out, void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostic (0.1.24; parity tests with teacher-forced routing): while `capture` is non-NULL every mn_llm_step* call copies each layer's
 * routing — int32 [n_layers][M][top_k + n_shared_slots]: the chosen expert ids in the path's own slot order, then the shared
 * pseudo-experts n_experts + j — to it, on the call's stream.  Process-global, not thread-safe; NULL (default) = off. */
MN_API int mn_llm_route_capture(int32_t* capture);

/* mn_llm_step with flags.  MN_STEP_DISTINCT_SEQUENCES: every row belongs to a different cache sequence and row_len == row_slot + 1
 * (a decode step of independent conversations / CFG rows — NOT a prefill chunk, whose rows attend each other's new K / V lines): the
 * rotary embedding and the K / V append then ride the attention launch (one launch fewer per layer; <= 64 rows, n_q / n_kv in
 * {1, 2, 4}).  Results are identical to mn_llm_step. */
#define MN_STEP_DISTINCT_SEQUENCES 1
MN_API int mn_llm_step_ex(const mn_llm* m, const float* x, int64_t ldx, int x_row_div, int M, const uint8_t* image_mask,
                   const int32_t* row_seq, const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len,
                   const uint8_t* key_mask, int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max,
                   float* hidden_out, void* workspace, size_t workspace_bytes, int flags, void* stream);

/* mn_llm_step for a PREFILL chunk whose rows are whole spans of cache sequences (BailingMoeModel.forward on a prompt, :1391-1540;
 * eager / flash causal attention, :791-812, 848-1045): span i = rows [r0_i, r0_i + len_i) of x, in order the slots
 * [past_i, past_i + len_i) of cache sequence seq_i (row_seq / row_slot / row_pos / row_len describe the same rows, row_len =
 * row_slot + 1; no key mask).  span_tab: device int32 [n_spans][4] = (seq_i, r0_i, len_i, past_i); max_len >= every len_i.
 * On the wide route (> 64 rows, head dim 128, GQA 4:1) the attention then runs as tiled flash attention on bf16 hi/lo operands
 * (fp32-class like the rest of the route) instead of row by row; every other shape takes mn_llm_step's path.  Same results. */
MN_API int mn_llm_step_spans(const mn_llm* m, const float* x, int64_t ldx, int M, const uint8_t* image_mask, const int32_t* row_seq,
                             const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len, float* kv_cache, int n_seq,
                             int64_t t_max, const int32_t* span_tab, int n_spans, int max_len, float* hidden_out, void* workspace,
                             size_t workspace_bytes, void* stream);

/* a[i] += delta, b[i] += delta, c[i] += delta for i < M (any pointer may be NULL): advances the
 * device-resident row_slot / row_pos / row_len arrays between autoregressive steps without a host round trip. */
MN_API int mn_rows_advance(int32_t* a, int32_t* b, int32_t* c, int M, int delta, void* stream);

/* MingTok semantic decoder, cached causal decode step for M <= 64 rows of ONE sequence each
 * (MingTok.forward_feature_decoder, modeling_mingtok.py:165-174 -> TransformerDecoder.forward_features,
 * vision_transformer.py:382-451) followed by linear_proj (modeling_bailingmm.py:111-115). */
typedef struct mn_semdec {
  int32_t dim, depth, n_heads, hidden, in_dim, proj_dim, proj_depth;
  float mean, scale;
  const uint16_t *in_w, *in_b;
  const uint16_t* const* ln1_g; const uint16_t* const* ln1_b;
  const uint16_t* const* wqkv;  const uint16_t* const* bqkv;
  const uint16_t* const* wproj; const uint16_t* const* bproj;
  const uint16_t* const* ln2_g; const uint16_t* const* ln2_b;
  const uint16_t* const* w12;   const uint16_t* const* b12;
  const uint16_t* const* w3;    const uint16_t* const* b3;
  const uint16_t *norm_g, *norm_b;
  const uint16_t* const* proj_w; const uint16_t* const* proj_b;  /* [proj_depth] linear_proj layers */
  /* Wide-row route (M > 64, optional; NULL arrays = route unavailable): the SwiGLU hidden width zero-padded to a
   * multiple of 64 (hidden_pad): w12p [2*hidden_pad, dim] (gate rows, zero rows, up rows, zero rows), b12p [2*hidden_pad],
   * w3p [dim, hidden_pad] (zero columns) — load-time copies, results unchanged. */
  int32_t hidden_pad;
  const uint16_t* const* w12p; const uint16_t* const* b12p; const uint16_t* const* w3p;
} mn_semdec;

MN_API size_t mn_semdec_workspace_bytes(const mn_semdec* s, int rows, int64_t t_max);
MN_API int mn_semdec_max_rows(const mn_semdec* s);   /* 64, or 2048 when the padded SwiGLU weights are given */
/* latent_norm [M, in_dim] fp32 (normalised latent from the RF head) -> sem_out [M, dim] fp32 (x_norm),
 * embed_out [M, proj_dim] fp32 (linear_proj(sem)), either may be NULL.
 * kv_cache fp32 [depth][n_seq][2][n_heads][t_max][64]. */
MN_API int mn_semdec_step(const mn_semdec* s, const float* latent_norm, int M,
                   const int32_t* row_seq, const int32_t* row_slot, const int32_t* row_len,
                   float* kv_cache, int n_seq, int64_t t_max, float* sem_out, float* embed_out,
                   void* workspace, size_t workspace_bytes, void* stream);


/* ------------------------------------------------------------------------------------------
 * 6. Tensor / expert parallelism of the decode path over one xGMI node (BASELINE configs[4]; SURVEY.md §8e — no reference
 *    counterpart: the reference runs the model on one device).  One process per GPU; every rank calls the same composites on its
 *    SHARD with the same arguments.  Partitioning, the one-shot all-reduce and the segment mechanism: csrc/tp.inl, DESIGN.md §7.
 * ------------------------------------------------------------------------------------------ */
#define MN_TP_MAX_WORLD 16

/* Communicator: every rank owns an inbox fp32 [2 (epoch parity)][world (sender)][cap] and arrival flags uint32
 * [world (sender)][rows_cap] in fine-grained device memory (mn_tp_alloc), mapped on every other rank (mn_tp_ipc_*).  inbox[p] /
 * flags[p] are the DEVICE addresses of rank p's arrays as seen from THIS rank (host arrays of `world` pointers).  epoch counts the
 * all-reduces completed on the communicator; it is advanced by the calls below, identically on every rank.  err: a local device
 * word that a bounded flag wait sets non-zero when it gives up (a dead peer / diverged launch order never hangs the GPU).  The
 * wait is bounded in WALL time — wait_ms milliseconds on the GPU's constant 100 MHz clock, 0 = the default of 30 s: host-side skew
 * between ranks (first-launch code loading, an allocator stall) is legitimate and can be long — and a row whose wait expired
 * POISONS its outputs with NaN instead of summing stale slabs: callers check err at their next host sync (TpRank does), and a
 * result can never silently miss an all-reduce. */
typedef struct mn_tp_comm {
  int32_t rank, world;
  float* const* inbox;
  uint32_t* const* flags;
  int64_t cap;          /* floats per (parity, sender) slab: >= rows * widest reduced row (3072 for the 16B-A3B RF head) */
  int32_t rows_cap;
  uint32_t epoch;
  uint32_t* err;
  uint32_t wait_ms;     /* 0 = 30 000 */
  /* Rows above which an all-reduce runs TWO-SHOT (round 5): reduce-scatter by pushing column piece j to rank j, the owner sums the
   * `world` pieces and pushes the reduced piece to every rank (all-gather) — 2 (world - 1) / world of the payload per rank instead of
   * (world - 1) x: what a many-row TP step wants; the one-shot form (one xGMI hop) stays for decode sizes.  0 = the default (16 rows);
   * < 0 = never (the relayed transport).  A two-shot all-reduce takes TWO epochs and adds one segment to the composites. */
  int32_t two_shot_rows;
} mn_tp_comm;

/* Host-side setup (these DO allocate / map; they are not launchers): fine-grained (uncached, system-coherent) device memory, its
 * IPC handle (64 bytes) for the other ranks, and the mapping of a peer's handle.  bytes for one rank's inbox = 2 * world * cap * 4,
 * flags = world * rows_cap * 4 (zero-filled by the caller). */
MN_API int mn_tp_alloc(size_t bytes, void** dptr);
MN_API int mn_tp_free(void* dptr);
MN_API int mn_tp_ipc_handle(void* dptr, void* handle_out_64);
MN_API int mn_tp_ipc_open(const void* handle_64, void** dptr);
MN_API int mn_tp_ipc_close(void* dptr);

/* out[m] = sum over the ranks of x[m]: x fp32 [M, D] (ldx == D) is this rank's partial, out fp32 [M, D] (ldo).  One-shot: the rank
 * pushes its rows into every rank's inbox over xGMI and sets their arrival flags; the reduce kernel waits on its LOCAL flags only.
 * 64 <= D <= 4096, D % 4 == 0, M <= rows_cap, M * D <= cap.  (latency-bound 4-18 KB payloads.) */
enum { MN_TP_PUSH = 1, MN_TP_REDUCE = 2, MN_TP_GATHER = 4 };   /* phase bits: all = the whole all-reduce in one call; split = work between them.
                                                                  MN_TP_GATHER (two-shot only, between the other two): the owner's reduce + all-gather push */
/* 1 or 2: how many segments (= epochs) an all-reduce of `rows` x D takes on this communicator (2 = two-shot). */
MN_API int mn_tp_allreduce_segments(const mn_tp_comm* comm, int rows, int D);
MN_API int mn_allreduce_oneshot(mn_tp_comm* comm, const float* x, int64_t ldx, float* out, int64_t ldo, int M, int D, int phase,
                                void* stream);

/* Expert parallelism, replicate-and-reduce: every rank holds all rows and the global routing (topk_idx [T, n_slot]
 * GLOBAL expert ids).  dispatch = moe_infer's count / argsort bookkeeping (modeling_bailing_moe.py:608-616) restricted to the experts
 * [expert0, expert0 + n_local) this rank owns: mn_moe_sort_tiles' outputs, the row-tile list only for those experts.
 * combine = the weighted un-permute (:630-639) across ranks: part[m] = sum over the LOCAL picks of topk_w * yg[slot_of] (+ optional
 * slabs P: the rank's slice of the shared expert), all-reduced; out[m] = h[m] + sum over ranks. */
MN_API int mn_ep_dispatch(const int32_t* topk_idx, int T, int n_slot, int n_experts, int expert0, int n_local, int32_t* counts,
                          int32_t* offsets, int32_t* perm, int32_t* slot_of, int tile_rows, int32_t* tile_g, int32_t* tile_m0,
                          int32_t* n_tiles, void* stream);
MN_API int mn_ep_combine(mn_tp_comm* comm, const float* yg, const int32_t* slot_of, const int32_t* topk_idx, const float* topk_w,
                         int n_slot, int expert0, int n_local, const float* P, int nz, int64_t slab, const float* h, int64_t ldh,
                         float* out, int64_t ldo, int M, int D, void* stream);

/* Decoder-stack step on a shard.  m describes the rank's shard as an mn_llm: n_q / n_kv = its local head counts, wqkv[l] =
 * [(n_q + 2 n_kv) * hd, H] (its q heads, then its K, then its V rows), wdense[l] = [H, n_q * hd] (its columns), n_experts = the
 * GLOBAL expert count with gate / image_gate replicated, n_shared_slots = 0, w_gate_up[l] / w_down[l] = its n_local_experts routed
 * experts; the KV arena holds its KV heads only.  tp adds the expert window and the rank's slice of the shared expert
 * (gate rows then up rows of shared_inter units, zero-padded to a multiple of 64; 0 = none).
 * Segments (mn_llm_tp_segments = 2 * n_layers + 1 with one-shot all-reduces; (that - 1) * mn_tp_allreduce_segments(comm, rows, hidden) + 1
 * in general — a two-shot all-reduce adds the owners' reduce + all-gather segment): [seg_begin, seg_end) selects the launch ranges between all-reduces; a rank
 * in production passes (0, n_segments).  Rows 1..2048; workspace mn_llm_tp_workspace_bytes.  Other arguments as mn_llm_step. */
typedef struct mn_llm_tp {
  int32_t expert0, n_local_experts;
  int32_t shared_inter;
  const uint16_t* const* ws_gate_up;   /* [L] [2 * shared_inter, H] */
  const uint16_t* const* ws_down;      /* [L] [H, shared_inter] */
  /* m->wfmt = MN_W_FP8_E4M3: ws_gate_up / ws_down are e4m3 bytes with these row scales ([2 * shared_inter] / [H] per layer) */
  const float* const* ws_gate_up_scale; const float* const* ws_down_scale;
} mn_llm_tp;
MN_API size_t mn_llm_tp_workspace_bytes(const mn_llm* m, const mn_llm_tp* tp, int rows, int64_t t_max);
MN_API int mn_llm_tp_segments(const mn_llm* m);
MN_API int mn_llm_step_tp(const mn_llm* m, const mn_llm_tp* tp, mn_tp_comm* comm, const float* x, int64_t ldx, int x_row_div, int M,
                          const uint8_t* image_mask, const int32_t* row_seq, const int32_t* row_slot, const int32_t* row_pos,
                          const int32_t* row_len, const uint8_t* key_mask, int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max,
                          float* hidden_out, void* workspace, size_t workspace_bytes, int seg_begin, int seg_end, void* stream);

/* RectifiedFlowLoss.sample on a shard: h->hidden = the rank's share of the SwiGLU width, w12[b] = [2 * hidden, w] (its gate rows,
 * then its up rows), b12[b] likewise, w3[b] = [w, hidden] (its columns); b3 and all other tensors are the full replicated ones.
 * One all-reduce per ResBlock per Euler step; segments = steps * depth + 1.  Other arguments as mn_rf_sample; rows 1..2048. */
MN_API size_t mn_rf_tp_workspace_bytes(const mn_rf_head* h, int rows);
MN_API int mn_rf_tp_segments(const mn_rf_head* h);
MN_API int mn_rf_sample_tp(const mn_rf_head* h, mn_tp_comm* comm, const float* hidden, int64_t ld_hidden, int rows, int n_images,
                           const float* noise, float temperature, float text_cfg, float image_cfg, float* latent_out,
                           void* workspace, size_t workspace_bytes, int seg_begin, int seg_end, void* stream);

/* lm_head + greedy pick (compute_logit, modeling_bailing_moe.py:1604-1620, + the argmax of greedy decoding):
 * idx[m] = vocab_offset + argmax_v hidden[m] . W[v] (ties -> lowest index, torch.argmax's rule), val[m] = that logit (optional).
 * W bf16 [V, H]: the whole lm_head, or a rank's vocabulary slice starting at vocab_offset.  The fp32 logits [M, V] are left in the
 * first bytes of the workspace (mn_lmhead_argmax_workspace_bytes). */
MN_API size_t mn_lmhead_argmax_workspace_bytes(int M, int V, int H);
MN_API int mn_lmhead_argmax(const float* hidden, int64_t ld_hidden, int M, const uint16_t* W, int64_t ldw, int V, int H,
                            int64_t vocab_offset, int64_t* idx, float* val, void* workspace, size_t workspace_bytes, void* stream);

/* Sampled pick (the `do_sample` branch of HF GenerationMixin.generate, which the reference forwards its generate kwargs to:
 * modeling_bailingmm.py:249-262, modeling_bailing_moe.py:1769-1796; warpers in transformers/generation/logits_process.py —
 * TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> one multinomial draw):
 * idx[m] = vocab_offset + the token the inverse CDF of the warped distribution of logits[m] gives at u[m] in [0, 1), tokens ranked by
 * descending score, ties by ascending id.  top_k = 0 and top_p >= 1 switch the respective warper off.  The kernel ranks at most 2048
 * candidates: top_k > 2048 is refused (MN_EINVAL); with top_k = 0 and top_p < 1 the nucleus limit is top_p of the FULL vocabulary's
 * softmax mass and the nucleus is searched among the 2048 largest scores — HF's result exactly whenever its nucleus holds <= 2048 tokens.
 * status (optional, int32 [M], may be NULL) reports the rows where that did not hold: MN_SAMPLE_NUCLEUS_TRUNCATED (the nucleus wanted more
 * than 2048 tokens and was cut to the 2048 best) and MN_SAMPLE_TIES_TRUNCATED (more ties at the k-th score than candidates left: the lowest
 * ids were kept).  The caller owns the uniform stream (u), so a decode is reproducible from a seed.  logits fp32 [M, V], row stride ld
 * (e.g. the first bytes of mn_lmhead_argmax's workspace). */
enum { MN_SAMPLE_NUCLEUS_TRUNCATED = 1, MN_SAMPLE_TIES_TRUNCATED = 2 };
MN_API int mn_sample_logits(const float* logits, int64_t ld, int M, int V, float temperature, int top_k, float top_p, const float* u,
                            int64_t vocab_offset, int64_t* idx, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------
 * 7. fp8 weight mode (BASELINE configs[4] "fp8").
 *    No arithmetic counterpart in the reference: its reduced-byte surface is the `dtype` switch of MingUniVisionInfer
 *    (mingunivision/mingunivisioninfer.py:46-70, int8 / int4 weight-only through quanto / bitsandbytes); the only fp8 trace is
 *    vllm/ming_lite.patch:171-190.  This mode is weight-only too: OCP e4m3fn bytes + one fp32 scale per OUTPUT ROW,
 *        W[n, k] = e4m3(Wq[n, k]) * scale[n],
 *    for the tensors that carry the bytes of the HBM-bound decode route — RF w12 / w3 and the routed + shared experts.
 *    Activations stay fp32 / bf16 hi+lo pairs and the MFMAs stay bf16: the kernels convert e4m3 -> bf16 in registers (exact)
 *    and apply the row scale to the fp32 accumulators, so the result is that of the DEQUANTISED weights to fp32 rounding —
 *    parity is defined against the oracle fed those (tests/test_gpu_fp8.py); the distance between the quantised and the bf16
 *    model is reported separately.  mn_quant_fp8_rows emits power-of-two scales, for which the dequantised weights are exactly
 *    representable in bf16 (mn_dequant_fp8_rows): the same model can be run through every bf16 route.
 * ------------------------------------------------------------------------------------------ */
enum { MN_W_BF16 = 0, MN_W_FP8_E4M3 = 1, MN_W_INT8 = 2, MN_W_NF4 = 3 };
/* MN_W_NF4 (round 5): the reference's `dtype="int4"` surface (mingunivisioninfer.py:46-58: BitsAndBytesConfig(load_in_4bit, nf4,
 * bf16 compute), i.e. bitsandbytes Linear4bit; third-party, absent here; restated in oracle/int4_ref.py) —
 *     W[n, k] = bf16_rne(NF4[code[n, k]] * absmax[n, k / 64]),   blocks of 64 consecutive k, fp32 absmax, no double quantisation.
 * Layout: `w` / Wq = rows of K / 2 bytes (two codes per byte; inside every dword the nibbles hold k 0 4 1 5 2 6 3 7 of its eight
 * consecutive k, from bit 0 up), `wscale` = the absmax table [rows][K / 64].  K % 64 == 0.  Strides of the grouped forms count
 * ELEMENTS for the codes (w_stride weights = w_stride / 2 bytes) and floats for the absmax tables.  The kernels look every block's 16
 * possible bf16 values up with v_perm_b32 (w8_codec.h) on the way into the MFMA tile; nothing is scaled afterwards.  Routes: the
 * weight-streaming launches (<= 64 rows; the experts run the grouped launch from one row on). */
MN_API int mn_quant_nf4_rows(const uint16_t* W, int64_t ldw, uint8_t* Wq, int64_t ldq, float* absmax, int64_t n_rows, int K, void* stream);
MN_API int mn_dequant_nf4_rows(const uint8_t* Wq, int64_t ldq, const float* absmax, uint16_t* W, int64_t ldw, int64_t n_rows, int K,
                               void* stream);
/* K-slice count of a dense streaming launch in format wfmt (the slab count the caller's reducing kernel sees) */
MN_API int mn_stream_mfma_wq_slices(int wfmt, int M, int Ntot, int K);
/* MN_W_INT8: the reference's `dtype="int8"` surface (mingunivisioninfer.py:59-68: HF QuantoConfig(weights="int8") = optimum-quanto qint8
 * weights; third-party, absent here; restated in oracle/int8_ref.py) — symmetric, one scale per output row, in the weight's dtype:
 *     scale[n] = bf16(amax_n / 127),  q = clamp(round(bf16(W / scale)), -128, 127),  W'[n, k] = bf16_rne(q[n, k] * scale[n]).
 * (Round 4 used power-of-two scales applied to the accumulators; round 5 follows quanto: the product is rounded to bf16 PER ELEMENT,
 * as quanto's qbytes_mm multiplies scale * weights in the activation dtype before the matmul — so the row scale rides the byte
 * conversion inside the kernels and nothing is scaled afterwards.)  Routes: the weight-streaming launches (<= 64 rows; experts on the
 * grouped launch from one row on). */
/* Row-wise quantisation at load: scale[n] = 2^ceil(log2(amax_n / 448)) (1 for an all-zero row), Wq[n, k] = e4m3_rne(W[n, k] / scale[n]).
 * W bf16 [n_rows, K] (row stride ldw), Wq bytes (row stride ldq), K % 4 == 0. */
MN_API int mn_quant_fp8_rows(const uint16_t* W, int64_t ldw, uint8_t* Wq, int64_t ldq, float* scale, int64_t n_rows, int K, void* stream);
/* W[n, k] = bf16_rne(e4m3(Wq[n, k]) * scale[n]) — exact for power-of-two scales. */
MN_API int mn_dequant_fp8_rows(const uint8_t* Wq, int64_t ldq, const float* scale, uint16_t* W, int64_t ldw, int64_t n_rows, int K,
                               void* stream);
/* mn_stream_mfma / mn_stream_mfma_grouped on fp8 weights: Wq e4m3 [Ntot, K] dense (K % 16 == 0, 16-byte aligned), wscale fp32
 * [Ntot]; grouped: group g reads Wq + g * w_stride bytes and wscale + g * s_stride floats.  Half the HBM bytes per launch. */
MN_API int mn_stream_mfma_w8(const uint16_t* Y, const uint8_t* Wq, const float* wscale, float* P, int M, int Ntot, int K, void* stream);
/* ... with the byte format as an argument (wfmt = MN_W_FP8_E4M3 | MN_W_INT8; the _w8 forms are wfmt = MN_W_FP8_E4M3) */
MN_API int mn_stream_mfma_wq(const uint16_t* Y, const uint8_t* Wq, const float* wscale, float* P, int M, int Ntot, int K, int wfmt, void* stream);
MN_API int mn_stream_mfma_grouped_wq(const uint16_t* Y, int y_rows, const uint8_t* Wq, int64_t w_stride, const float* wscale,
                                     int64_t s_stride, float* P, int p_rows, const int32_t* off, const int32_t* xrows, int G,
                                     int max_rows, int Ntot, int K, int wfmt, void* stream);
/* int8 rows by quanto's rule (above; scale 1 for an all-zero row), and back.  Arguments as the fp8 pair. */
MN_API int mn_quant_int8_rows(const uint16_t* W, int64_t ldw, uint8_t* Wq, int64_t ldq, float* scale, int64_t n_rows, int K, void* stream);
MN_API int mn_dequant_int8_rows(const uint8_t* Wq, int64_t ldq, const float* scale, uint16_t* W, int64_t ldw, int64_t n_rows, int K,
                                void* stream);
MN_API int mn_stream_mfma_w8_slices(int M, int Ntot, int K);
MN_API int mn_stream_mfma_grouped_w8(const uint16_t* Y, int y_rows, const uint8_t* Wq, int64_t w_stride, const float* wscale,
                                     int64_t s_stride, float* P, int p_rows, const int32_t* off, const int32_t* xrows, int G,
                                     int max_rows, int Ntot, int K, void* stream);

/* ------------------------------------------------------------------------------------------
 * 8. fp8-MFMA regime (0.1.24) — BASELINE.json configs[4]'s "fp8 MFMA".  A LABELLED reduced-arithmetic regime with its own tolerance,
 *    never a default: the reference has no fp8 path, the parity target is the fp32 oracle at a stated looser bar.
 *    Both GEMM operands are OCP e4m3 bytes with one fp32 scale per row (weights: the "fp8" weight format of section 7; activations:
 *    mn_quant_fp8_rows of the bf16 operand the hi/lo route would multiply), products on v_mfma_f32_16x16x128_f8f6f4 (no block
 *    scales), fp32 accumulation — the 256 x 256-tile kernel of the wide route at twice the bf16 MFMA rate per pass, one pass
 *    instead of the hi/lo pair's two.
 *    C = epilogue((A8 . a_scale)(W8 . w_scale)^T + bias): swiglu == 0 -> C fp32 [M, N] (ksplit > 1: slice z at C + z * M * N, ldc == N,
 *    returns the slice count); swiglu == 1 -> W8 holds 2N rows (gate, up), C bf16 [M, N] = silu(gate) * up (swiglu_ffn.py:30-34,
 *    diff_loss_rf_swiglu.py:54-72).  K % 128 == 0, 16-byte aligned rows and scale arrays.
 * ------------------------------------------------------------------------------------------ */
MN_API int mn_gemm256_f8(const uint8_t* A, int64_t lda, const float* a_scale, const uint8_t* W, int64_t ldw, const float* w_scale,
                         const uint16_t* bias, void* C, int64_t ldc, int M, int N, int K, int swiglu, int ksplit, void* stream);
MN_API int mn_gemm256_f8_slices(int K, int ksplit);

#ifdef __cplusplus
}
#endif
#endif /* MINGNATIVE_H */

/*
 * mingnative_dev.h — A/B hooks of tools/ (NOT part of the product ABI).
 *
 * Present only in libmingnative_dev.so (`make -C ming_univision_amd/csrc dev`, compiled with -DMN_DEV_HOOKS); the shipped
 * libmingnative.so does not contain these symbols.  They overwrite process-global launch-plan parameters, are not thread-safe
 * and must be called before the first launch / before any workspace size is queried (workspace carving depends on them).
 */
#ifndef MINGNATIVE_DEV_H
#define MINGNATIVE_DEV_H
#include "mingnative.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MN_DEV_API MN_API

MN_DEV_API void mn_skinny_tune(int R, int nt, int bpc);                 /* skinny_gemm.hip launch plan */
MN_DEV_API void mn_stream_tune_plan(int kch, int nw);                   /* stream_mfma.hip K-slice length / waves */
MN_DEV_API void mn_stream_tune_plan_k(int K, int kch, int nw);          /* stream_mfma.hip: mn_stream_tune_plan for the matrices with this K only */
MN_DEV_API void mn_stream_tune_w8(int depth);                           /* stream_mfma.hip fp8 form: weight chunks in flight per wave (1 or 2) */
MN_DEV_API void mn_moe_tune_down(int on);                               /* engine.hip: 1- / 2-row MoE down projection on moe_down.hip (1, default) or the K-segment skinny kernel (0) */
MN_DEV_API void mn_rf_kc_trace(void* buf);                               /* stream_kc.hip: device buffer of [workgroups][2 x blocks][8] uint64 clock stamps (100 MHz) of the persistent launch (0: phase start, 1: operand image ready, 2: stream done, 3: epilogue done, 4: arrived, 5: next phase's weights requested, 6: barrier passed), or NULL */
MN_DEV_API void mn_rf_kc_fault(int wg, unsigned wait_ms);                /* stream_kc.hip, tests: workgroup `wg` of every persistent launch arrives 3 x wait_ms late and the barrier waits give up after wait_ms (wg < 0, 0: off, 2 s) */
MN_DEV_API void mn_rf_kc_persist_all(int on);                            /* stream_kc.hip: the persistent launches for int8 heads too (default: bf16, e4m3 and NF4) */
MN_DEV_API void mn_rf_kc_tune(int rd12, int rd3);                         /* stream_kc.hip: weight chunks in flight per wave of w12' (1..3) / w3' (1, 2, 4) */
MN_DEV_API void mn_rf_tune_fuse(int on);                                /* engine.hip, RF chain: bit 0 = SwiGLU glue folded into w3's prologue at <= 4 rows, bit 1 = the Euler-step boundary as one launch, bit 2 = bf16 adaLN through the GEMM instead of the streaming launch, bit 3 = K-complete launches OFF, bit 4 = the persistent per-step launch OFF, bit 5 = the whole-sampler launch OFF (default 3) */
MN_DEV_API void mn_attn_tune_fuse(int on);                              /* decode_ops.hip: RoPE + KV append riding the decode-attention launch where it pays (1, default) or never (0) */
MN_DEV_API void mn_attn_tune_one(int max_rows);                         /* decode_ops.hip: rows up to which decode attention over <= 1024-key caches is ONE launch (key split inside the workgroup); 0 = always split + combine */
MN_DEV_API void mn_stream_kloop_tune(int nz, int depth, int nt);        /* stream_kloop.hip */
MN_DEV_API void mn_stream_kloop_tune_small(int div);
MN_DEV_API void mn_moe_router_tune(int max_rows);                       /* one-launch router up to this many rows */
MN_DEV_API void mn_moe_tune_gate_up(int on, int max_rows);               /* engine.hip: router + expert gate/up of <= max_rows-row steps as one launch (moe_gate_up.hip) */
MN_DEV_API void mn_moe_tune_min_rows(int rows);                         /* first row count of a decode step on the grouped expert route (bf16 / e4m3) */
MN_DEV_API void mn_llm_tune_chain(int max_rows);                        /* fused decoder chain up to this many rows */
MN_DEV_API void mn_gemm_tune(int glds);                                 /* batch_ops.hip: global_load_lds staging */
MN_DEV_API void mn_gemm_route256(int on);                               /* mn_gemm_bf16 -> gemm256 for large problems */
MN_DEV_API void mn_gemm256_tune_order(int group_m, int group_list);     /* gemm256 tile order */
MN_DEV_API void mn_wide_tune(int llm_min_rows, int rf_min_rows, int sem_min_rows);   /* first row count on the wide route */
/* Measurement only ("lo-pass map"): bit s = Linear site s of the wide route multiplies plain bf16 activations (the hi
 * rows only).  Sites: 0 vis_head, 1 cond_embed, 2 adaLN, 3 RF w12, 4 RF w3, 5 RF final, 6 QKV, 7 dense, 8 gate, 9 experts (gate/up +
 * down), 10 semdec qkv, 11 semdec proj, 12 semdec w12, 13 semdec w3, 14 linear_proj. */
MN_DEV_API void mn_lo_drop_mask(unsigned mask);
/* Tile lists (expert GEMMs): skip the MFMAs and fragment reads of M-fragments without a live row (1, shipped) or run them all (0). */
MN_DEV_API void mn_gemm256_tune_thin(int on);
/* Measurement only: K / V rows are rounded to bf16 when appended to the (fp32) cache — the values a bf16 KV cache would hold. */
MN_DEV_API void mn_kv_round_bf16(int on);

#ifdef __cplusplus
}
#endif
#endif

// wide_llm.inl — the Bailing-MoE decoder-stack step (BailingMoeModel.forward, modeling_bailing_moe.py:1391-1540, q_len = 1
// per row) and the cached MingTok semantic-decoder step (mingtok/modeling_mingtok.py:165-174) for 65..2048 rows advancing
// in lock-step (textually part of engine.hip).  Same arithmetic as the <= 64-row routes; every Linear is a gemm256 launch on
// bf16 hi/lo operands, every stretch between two Linears one wide_glue launch, and the MoE expert loop (:605-639) is two
// GROUPED gemm256 launches over the expert-sorted (row, pick) pairs — the gate/up launch gathers its rows by index while
// staging and applies SwiGLU + hi/lo split in its epilogue, the weighted un-permute + residual rides the next glue.
extern "C" int mn_moe_sort_tiles(const int32_t* topk_idx, int T, int n_slot, int n_groups, int32_t* counts, int32_t* offsets,
                                 int32_t* perm, int32_t* slot_of, int tile_rows, int32_t* tile_g, int32_t* tile_m0,
                                 int32_t* n_tiles, void* stream);

// Router tail for many rows: one wave per row sums the gate GEMM's split-K slabs, fp32 softmax, iterative arg-max top-k
// (ties -> lowest expert), renormalise, append the shared pseudo-experts  (BailingMoeGate.forward :505-520).
// Rows flagged by image_mask take their logits from the image gate's slabs P_img (multi-gate blend, :565-592).
__global__ __launch_bounds__(256) void moe_topk_partials_kernel(const float* __restrict__ P, const float* __restrict__ P_img,
                                                                const uint8_t* __restrict__ image_mask, int nz, int64_t slab, int M, int E,
                                                                int top_k, int norm_topk_prob, int n_shared,
                                                                int32_t* __restrict__ topk_idx, float* __restrict__ topk_w) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  if (P_img && image_mask && image_mask[m]) P = P_img;      // wave-uniform
  float s = -INFINITY;
  if (lane < E) {
    s = 0.f;
    for (int z = 0; z < nz; ++z) s += P[z * slab + (int64_t)m * E + lane];
  }
  const float mx = wave_max(s);
  float p = lane < E ? __expf(s - mx) : 0.f;
  p = p / wave_sum(p);
  float cur = lane < E ? p : -1.f, wsum = 0.f, myw = 0.f;
  int myidx = 0;
  const int n_slot = top_k + n_shared;
  for (int k = 0; k < top_k; ++k) {
    const float best = wave_max(cur);
    const int sel = __ffsll((long long)__ballot(cur == best)) - 1;
    if (lane == k) { myw = best; myidx = sel; }
    if (lane == sel) cur = -1.f;
    wsum += best;
  }
  if (lane < top_k) {
    topk_idx[(int64_t)m * n_slot + lane] = myidx;
    topk_w[(int64_t)m * n_slot + lane] = (norm_topk_prob && top_k > 1) ? myw / wsum : myw;
  } else if (lane < n_slot) {
    topk_idx[(int64_t)m * n_slot + lane] = E + (lane - top_k);
    topk_w[(int64_t)m * n_slot + lane] = 1.0f;
  }
}

struct LlmWideWs {
  float *h, *pp, *q, *yg, *tw;
  bf16_t *yh, *ya, *y2;
  bf16_t *wq_gu, *wq_dn;           // weight-only modes: ONE layer's expert weights de-quantised, [G][2 I][H] / [G][H][I] (wide_rf.inl)
  uint8_t* yh8; float *s_h, *s_y2; // fp8-MFMA regime: the experts' e4m3 operand [rows][H] + its row scales, the down projection's row scales [P]
  int32_t *ti, *cnt, *off, *perm, *slot_of, *tile_g, *tile_m0, *n_tiles;
  int max_mtiles;
  void* attn_ws;
  size_t attn_ws_bytes;
  int ks_qkv, ks_dense, ks_gate;
};

static bool llm_wide_ok(const mn_llm* m, int rows) {
  const int ad = m->n_q * m->head_dim, n_slot = m->top_k + m->n_shared_slots;
  const bool fmt_ok = m->wfmt == MN_W_BF16 || ((m->wfmt == MN_W_FP8_E4M3 || m->wfmt == MN_W_INT8 || m->wfmt == MN_W_NF4) && m->w_gate_up_scale && m->w_down_scale);
  return fmt_ok && rows >= g_wide_min_llm && rows <= 2048 && wide_glue_ok(m->hidden) && (m->hidden % 64) == 0 && (ad % 64) == 0 && (m->moe_inter % 64) == 0 &&
         m->n_experts <= 64 && (m->n_experts % 4) == 0 && m->n_experts + m->n_shared_slots <= 128 && (int64_t)rows * n_slot <= 65536 &&
         (m->head_dim == 64 || m->head_dim == 128);
}

// the fp8-MFMA regime of the wide route (mn_llm.arith; mingnative.h section 8): the grouped expert GEMMs on e4m3 x e4m3
static bool llm_f8_mfma(const mn_llm* m) {
  return m->arith == MN_ARITH_FP8_MFMA && m->wfmt == MN_W_FP8_E4M3 && m->w_gate_up_scale && m->w_down_scale && (m->hidden % 128) == 0 &&
         (m->moe_inter % 128) == 0;
}

static size_t llm_wide_carve(const mn_llm* m, int rows, int64_t t_max, void* ws, size_t cap, LlmWideWs* o) {
  Carver cv(ws, cap, ws == nullptr);
  const int H = m->hidden, ad = m->n_q * m->head_dim, qkv_dim = (m->n_q + 2 * m->n_kv) * m->head_dim;
  const int n_slot = m->top_k + m->n_shared_slots, G = m->n_experts + m->n_shared_slots;
  const size_t P = (size_t)rows * n_slot;
  o->ks_qkv = rf_wide_ksplit(rows, qkv_dim, H);
  o->ks_dense = rf_wide_ksplit(rows, H, ad);
  o->ks_gate = rf_wide_ksplit(rows, m->n_experts, H);
  size_t pmax = (size_t)mn_gemm256_slices(H, o->ks_qkv) * rows * qkv_dim;
  const size_t p2 = (size_t)mn_gemm256_slices(ad, o->ks_dense) * rows * H;
  const size_t p3 = (size_t)2 * mn_gemm256_slices(H, o->ks_gate) * rows * m->n_experts;    // text gate slabs + image gate slabs
  if (p2 > pmax) pmax = p2;
  if (p3 > pmax) pmax = p3;
  o->h = cv.take<float>((size_t)rows * H);
  o->pp = cv.take<float>(pmax);
  o->q = cv.take<float>((size_t)rows * ad);
  o->yg = cv.take<float>(P * H);
  o->tw = cv.take<float>(P);
  o->yh = cv.take<bf16_t>((size_t)2 * rows * H);
  o->ya = cv.take<bf16_t>((size_t)2 * rows * ad);
  o->y2 = cv.take<bf16_t>(2 * P * m->moe_inter);
  o->ti = cv.take<int32_t>(P);
  o->cnt = cv.take<int32_t>(G);
  o->off = cv.take<int32_t>((size_t)G + 1);
  o->perm = cv.take<int32_t>(P);
  o->slot_of = cv.take<int32_t>(P);
  o->max_mtiles = (int)(P / 128) + G;                   // sum_g ceil(cnt_g / 128) <= P / 128 + G
  o->tile_g = cv.take<int32_t>((size_t)o->max_mtiles);
  o->tile_m0 = cv.take<int32_t>((size_t)o->max_mtiles);
  o->n_tiles = cv.take<int32_t>(4);
  o->attn_ws_bytes = mn_attn_decode_workspace_bytes(rows, m->n_q, m->head_dim, t_max);
  o->attn_ws = cv.take<char>(o->attn_ws_bytes);
  const bool f8 = llm_f8_mfma(m);                       // (the e4m3 bytes ARE the operands: no per-layer bf16 expansion)
  o->wq_gu = cv.take<bf16_t>(m->wfmt && !f8 ? (size_t)G * 2 * m->moe_inter * H : 0);
  o->wq_dn = cv.take<bf16_t>(m->wfmt && !f8 ? (size_t)G * H * m->moe_inter : 0);
  o->yh8 = cv.take<uint8_t>(f8 ? (size_t)rows * H : 0);
  o->s_h = cv.take<float>(f8 ? (size_t)rows + 64 : 0);
  o->s_y2 = cv.take<float>(f8 ? P + 64 : 0);
  return cv.off;
}

static int llm_step_wide(const mn_llm* m, const float* x, int64_t ldx, int x_row_div, int M, const uint8_t* image_mask, const int32_t* row_seq,
                         const int32_t* row_slot, const int32_t* row_pos, const int32_t* row_len, const uint8_t* key_mask,
                         int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max, float* hidden_out, void* workspace,
                         size_t workspace_bytes, void* stream, const int32_t* span_tab = nullptr, int n_spans = 0, int span_max_len = 0) {
  LlmWideWs w;
  const size_t need = llm_wide_carve(m, M, t_max, workspace, workspace_bytes, &w);
  if (need > workspace_bytes) { mn_set_error("mn_llm_step: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  hipStream_t st = mn_stream(stream);
  const int H = m->hidden, hd = m->head_dim, nq = m->n_q, nkv = m->n_kv, I = m->moe_inter, E = m->n_experts;
  const int ad = nq * hd, qkv_dim = (nq + 2 * nkv) * hd, n_slot = m->top_k + m->n_shared_slots, G = E + m->n_shared_slots;
  const int64_t P = (int64_t)M * n_slot;
  const int64_t layer_kv = (int64_t)n_seq * 2 * nkv * t_max * hd;
  const float q_scale = 1.0f / sqrtf((float)hd);
  WideGlue g;
  const bool f8 = llm_f8_mfma(m);
  for (int l = 0; l <= m->n_layers; ++l) {
    // glue: (stack input | previous layer's expert combine + residual) -> RMSNorm(ln1 | final norm)
    const bool fin = l == m->n_layers;
    memset(&g, 0, sizeof(g));
    if (l == 0) { g.x = x; g.ldx = ldx; g.x_row_div = x_row_div; }
    else { g.h = w.h; g.ldh = H; g.cy = w.yg; g.cpos = w.slot_of; g.cw = w.tw; g.n_slot = n_slot; }
    g.h_out = fin ? nullptr : w.h; g.ldho = H;
    g.norm = 1; g.ng = fin ? m->final_norm : m->ln1[l]; g.eps = m->rms_eps;
    if (fin) { g.out = hidden_out; g.ldo = H; }
    else { g.Y = w.yh; g.ldy = H; g.y_lo_off = (int64_t)M * H; }
    g.M = M; g.D = H;
    wide_glue(g, st);
    if (fin) break;
    float* kv_l = kv_cache + (int64_t)l * layer_kv;
    // QKV (split-K slabs) -> RoPE + KV append reduce them  (:743-789)
    mn_g256 a = g256_hilo(w.yh, H, lo_at(LO_LLM_QKV, (int64_t)M * H), m->wqkv[l], H, nullptr, w.pp, qkv_dim, M, qkv_dim, H);
    a.c_zstride = (int64_t)M * qkv_dim;
    int nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_qkv, stream);
    if (nz < 0) return nz;
    MN_TRY(mn_rope_kv_from_partials(w.pp, qkv_dim, nz, (int64_t)M * qkv_dim, M, nq, nkv, hd, 1, m->cos_tab, m->sin_tab, row_seq,
                                    row_slot, row_pos, m->mrope_sec_t, m->mrope_sec_h, q_scale, w.q, kv_l, t_max, stream));
    // masked GQA against the cache; the combine writes the dense projection's hi/lo operand  (:791-812)
    // (a prefill chunk described as spans: the tiled hi/lo flash kernel — a staged K / V tile serves 16 query rows x 4 heads instead of
    //  every row re-reading its whole prefix)
    if (span_tab && hd == 128 && nq == 4 * nkv && !key_mask)
      MN_TRY(mn_flash_prefill_gqa_hd128_f32(w.q, kv_l, t_max, nq, nkv, span_tab, n_spans, span_max_len, nullptr, w.ya, (int64_t)M * ad, stream));
    else
      MN_TRY(mn_attn_decode_split(w.q, M, nq, nkv, hd, kv_l, t_max, row_seq, row_len, key_mask, ld_mask, nullptr, w.ya, w.attn_ws,
                                  w.attn_ws_bytes, stream));
    a = g256_hilo(w.ya, ad, lo_at(LO_LLM_DENSE, (int64_t)M * ad), m->wdense[l], ad, nullptr, w.pp, H, M, H, ad);
    a.c_zstride = (int64_t)M * H;
    nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_dense, stream);
    if (nz < 0) return nz;
    // glue: h += dense slabs; RMSNorm(ln2) -> gate / expert operand  (:1214-1218)
    memset(&g, 0, sizeof(g));
    g.h = w.h; g.ldh = H; g.P = w.pp; g.nz = nz; g.slab = (int64_t)M * H; g.h_out = w.h; g.ldho = H;
    g.norm = 1; g.ng = m->ln2[l]; g.eps = m->rms_eps; g.Y = w.yh; g.ldy = H; g.y_lo_off = (int64_t)M * H; g.M = M; g.D = H;
    if (f8) { g.Y8 = w.yh8; g.y8_scale = w.s_h; }      // the experts' e4m3 operand beside the router's hi/lo one (routing stays fp32-class)
    wide_glue(g, st);
    // router: gate logits (split-K slabs) -> softmax / top-k -> expert sort  (:505-520, 608-616)
    a = g256_hilo(w.yh, H, lo_at(LO_LLM_GATE, (int64_t)M * H), m->gate[l], H, nullptr, w.pp, E, M, E, H);
    a.c_zstride = (int64_t)M * E;
    nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_gate, stream);
    if (nz < 0) return nz;
    const float* p_img = nullptr;
    if (image_mask && m->image_gate && m->image_gate[l]) {      // image-gate logits of all rows, chosen per row by the mask (:565-592)
      float* pi = w.pp + (int64_t)nz * M * E;
      a = g256_hilo(w.yh, H, (int64_t)M * H, m->image_gate[l], H, nullptr, pi, E, M, E, H);
      a.c_zstride = (int64_t)M * E;
      const int nzi = mn_gemm256_ex(&a, MN_G256_F32, w.ks_gate, stream);
      if (nzi < 0) return nzi;
      p_img = pi;
    }
    hipLaunchKernelGGL(moe_topk_partials_kernel, dim3(mn_cdiv(M, 4)), dim3(256), 0, st, (const float*)w.pp, p_img, image_mask, nz,
                       (int64_t)M * E, M, E, m->top_k, m->norm_topk_prob, m->n_shared_slots, w.ti, w.tw);
    route_capture(l, w.ti, M, n_slot, st);
    MN_TRY(mn_moe_sort_tiles(w.ti, M, n_slot, G, w.cnt, w.off, w.perm, w.slot_of, (lo_at(LO_LLM_EXPERTS, 1) && !f8) ? 128 : 256, w.tile_g, w.tile_m0,
                             w.n_tiles, stream));
    if (f8) {
      // fp8-MFMA regime: gate/up = e4m3(RMSNorm'd row, gathered by perm) x e4m3 expert bytes, SwiGLU in the epilogue -> bf16 [P, I];
      // down = e4m3 of that (one quantise pass over the sorted rows) x e4m3 bytes -> yg [P, H] fp32
      mn_g256 a8;
      memset(&a8, 0, sizeof(a8));
      a8.A = reinterpret_cast<const bf16_t*>(w.yh8); a8.lda = H; a8.W = reinterpret_cast<const bf16_t*>(m->w_gate_up[l]); a8.ldw = H;
      a8.C = w.y2; a8.ldc = I; a8.M = M; a8.N = I; a8.K = H; a8.w_pair_rows = I;
      a8.f8 = 1; a8.a_scale = w.s_h; a8.w_scale = m->w_gate_up_scale[l]; a8.w_sstride = (int64_t)2 * I;
      a8.g_off = w.off; a8.g_cnt = w.cnt; a8.w_gstride = (int64_t)2 * I * H; a8.a_rows = w.perm; a8.n_groups = G;
      a8.tile_g = w.tile_g; a8.tile_m0 = w.tile_m0; a8.n_tiles = w.n_tiles; a8.max_mtiles = w.max_mtiles;
      MN_TRYZ(mn_gemm256_ex(&a8, MN_G256_SWIGLU_BF16, 1, stream));
      uint8_t* y28 = reinterpret_cast<uint8_t*>(w.y2 + P * I);
      MN_TRYZ(mn_quant_fp8_rows(w.y2, I, y28, I, w.s_y2, P, I, stream));
      memset(&a8, 0, sizeof(a8));
      a8.A = reinterpret_cast<const bf16_t*>(y28); a8.lda = I; a8.W = reinterpret_cast<const bf16_t*>(m->w_down[l]); a8.ldw = I;
      a8.C = w.yg; a8.ldc = H; a8.M = M; a8.N = H; a8.K = I;
      a8.f8 = 1; a8.a_scale = w.s_y2; a8.w_scale = m->w_down_scale[l]; a8.w_sstride = H;
      a8.g_off = w.off; a8.g_cnt = w.cnt; a8.w_gstride = (int64_t)H * I; a8.n_groups = G;
      a8.tile_g = w.tile_g; a8.tile_m0 = w.tile_m0; a8.n_tiles = w.n_tiles; a8.max_mtiles = w.max_mtiles;
      MN_TRYZ(mn_gemm256_ex(&a8, MN_G256_F32, 1, stream));
      continue;
    }
    // experts: grouped gate/up (rows gathered by perm, SwiGLU + split epilogue), grouped down -> yg [P, H]  (:617-628, 483-484)
    const bf16_t *wgu = m->w_gate_up[l], *wdn = m->w_down[l];
    if (m->wfmt) {          // weight-only mode: this layer's W' into the scratch (1.14 GB at the 16B-A3B shape; ~0.35 ms per layer and step)
      MN_TRYZ(wide_dequant_rows(m->wfmt, m->w_gate_up[l], m->w_gate_up_scale[l], w.wq_gu, (int64_t)G * 2 * I, H, stream));
      MN_TRYZ(wide_dequant_rows(m->wfmt, m->w_down[l], m->w_down_scale[l], w.wq_dn, (int64_t)G * H, I, stream));
      wgu = w.wq_gu; wdn = w.wq_dn;
    }
    a = g256_hilo(w.yh, H, lo_at(LO_LLM_EXPERTS, (int64_t)M * H), wgu, H, nullptr, w.y2, I, M, I, H);
    a.w_pair_rows = I; a.c_lo_off = P * I;
    a.g_off = w.off; a.g_cnt = w.cnt; a.w_gstride = (int64_t)2 * I * H; a.a_rows = w.perm; a.n_groups = G;
    a.tile_g = w.tile_g; a.tile_m0 = w.tile_m0; a.n_tiles = w.n_tiles; a.max_mtiles = w.max_mtiles;
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_SWIGLU_SPLIT, 1, stream));
    a = g256_hilo(w.y2, I, lo_at(LO_LLM_EXPERTS, P * I), wdn, I, nullptr, w.yg, H, M, H, I);
    a.g_off = w.off; a.g_cnt = w.cnt; a.w_gstride = (int64_t)H * I; a.n_groups = G;
    a.tile_g = w.tile_g; a.tile_m0 = w.tile_m0; a.n_tiles = w.n_tiles; a.max_mtiles = w.max_mtiles;
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
  }
  MN_CHECK_LAUNCH("mn_llm_step(wide)");
  return MN_OK;
}

// ===========================================================================================
// MingTok semantic decoder, wide rows
// ===========================================================================================
struct SemWideWs {
  float *h, *pp, *q, *sem, *p0;
  bf16_t *yd, *ya, *yb, *ys, *yp;
  void* attn_ws;
  size_t attn_ws_bytes;
  int ks_qkv, ks_proj, ks_w3;
};

static bool sem_wide_ok(const mn_semdec* s, int rows) {
  return rows >= g_wide_min_sem && rows <= 2048 && s->w12p && s->b12p && s->w3p && s->hidden_pad >= s->hidden && (s->hidden_pad % 64) == 0 &&
         wide_glue_ok(s->dim) && (s->dim % 64) == 0 && s->dim == s->n_heads * 64 &&
         (s->proj_depth == 0 || (wide_glue_ok(s->proj_dim) && (s->proj_dim % 64) == 0));
}

static size_t sem_wide_carve(const mn_semdec* s, int rows, int64_t t_max, void* ws, size_t cap, SemWideWs* o) {
  Carver cv(ws, cap, ws == nullptr);
  const int D = s->dim, HP = s->hidden_pad;
  o->ks_qkv = rf_wide_ksplit(rows, 3 * D, D);
  o->ks_proj = rf_wide_ksplit(rows, D, D);
  o->ks_w3 = rf_wide_ksplit(rows, D, HP);
  size_t pmax = (size_t)mn_gemm256_slices(D, o->ks_qkv) * rows * 3 * D;
  const size_t p2 = (size_t)mn_gemm256_slices(D, o->ks_proj) * rows * D;
  const size_t p3 = (size_t)mn_gemm256_slices(HP, o->ks_w3) * rows * D;
  if (p2 > pmax) pmax = p2;
  if (p3 > pmax) pmax = p3;
  o->h = cv.take<float>((size_t)rows * D);
  o->pp = cv.take<float>(pmax);
  o->q = cv.take<float>((size_t)rows * D);
  o->sem = cv.take<float>((size_t)rows * D);
  o->p0 = cv.take<float>((size_t)rows * s->proj_dim);
  o->yd = cv.take<bf16_t>((size_t)2 * rows * D);
  o->ya = cv.take<bf16_t>((size_t)2 * rows * D);
  o->yb = cv.take<bf16_t>((size_t)2 * rows * HP);
  o->ys = cv.take<bf16_t>((size_t)2 * rows * D);
  o->yp = cv.take<bf16_t>((size_t)2 * rows * s->proj_dim);
  o->attn_ws_bytes = mn_attn_decode_workspace_bytes(rows, s->n_heads, 64, t_max);
  o->attn_ws = cv.take<char>(o->attn_ws_bytes);
  return cv.off;
}

static int semdec_step_wide(const mn_semdec* s, const float* latent_norm, int M, const int32_t* row_seq, const int32_t* row_slot,
                            const int32_t* row_len, float* kv_cache, int n_seq, int64_t t_max, float* sem_out, float* embed_out,
                            void* workspace, size_t workspace_bytes, void* stream) {
  SemWideWs w;
  const size_t need = sem_wide_carve(s, M, t_max, workspace, workspace_bytes, &w);
  if (need > workspace_bytes) { mn_set_error("mn_semdec_step: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  hipStream_t st = mn_stream(stream);
  const int D = s->dim, nh = s->n_heads, HP = s->hidden_pad;
  const int64_t layer_kv = (int64_t)n_seq * 2 * nh * t_max * 64, loD = (int64_t)M * D;
  // in_proj + channel-repeat shortcut on the de-normalised latent  (modeling_mingtok.py:168; vision_transformer.py:373-380)
  hipLaunchKernelGGL(semdec_in_kernel, dim3(mn_cdiv(D, 256), M), dim3(256), 0, st, latent_norm, s->in_dim, s->scale, s->mean,
                     s->in_w, s->in_b, w.h, D);
  WideGlue g;
  memset(&g, 0, sizeof(g));
  g.h = w.h; g.ldh = D; g.norm = 2; g.ng = s->ln1_g[0]; g.nb = s->ln1_b[0]; g.eps = 1e-6f; g.Y = w.yd; g.ldy = D; g.y_lo_off = loD;
  g.M = M; g.D = D;
  wide_glue(g, st);
  for (int l = 0; l < s->depth; ++l) {
    float* kv_l = kv_cache + (int64_t)l * layer_kv;
    // CausalBlock (layers/block.py:301-327): attention
    mn_g256 a = g256_hilo(w.yd, D, lo_at(LO_SEM_QKV, loD), s->wqkv[l], D, s->bqkv[l], w.pp, 3 * D, M, 3 * D, D);
    a.c_zstride = (int64_t)M * 3 * D;
    int nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_qkv, stream);            // bias rides slice 0
    if (nz < 0) return nz;
    MN_TRY(mn_rope_kv_from_partials(w.pp, 3 * D, nz, (int64_t)M * 3 * D, M, nh, nh, 64, 0, nullptr, nullptr, row_seq, row_slot, nullptr,
                                    0, 0, 0.125f, w.q, kv_l, t_max, stream));
    MN_TRY(mn_attn_decode_split(w.q, M, nh, nh, 64, kv_l, t_max, row_seq, row_len, nullptr, 0, nullptr, w.ya, w.attn_ws,
                                w.attn_ws_bytes, stream));
    a = g256_hilo(w.ya, D, lo_at(LO_SEM_PROJ, loD), s->wproj[l], D, s->bproj[l], w.pp, D, M, D, D);
    a.c_zstride = loD;
    nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_proj, stream);
    if (nz < 0) return nz;
    memset(&g, 0, sizeof(g));
    g.h = w.h; g.ldh = D; g.P = w.pp; g.nz = nz; g.slab = loD; g.h_out = w.h; g.ldho = D;
    g.norm = 2; g.ng = s->ln2_g[l]; g.nb = s->ln2_b[l]; g.eps = 1e-6f; g.Y = w.yd; g.ldy = D; g.y_lo_off = loD; g.M = M; g.D = D;
    wide_glue(g, st);
    // SwiGLU FFN on the zero-padded hidden width
    a = g256_hilo(w.yd, D, lo_at(LO_SEM_W12, loD), s->w12p[l], D, s->b12p[l], w.yb, HP, M, HP, D);
    a.w_pair_rows = HP; a.c_lo_off = (int64_t)M * HP;
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_SWIGLU_SPLIT, 1, stream));
    a = g256_hilo(w.yb, HP, lo_at(LO_SEM_W3, (int64_t)M * HP), s->w3p[l], HP, s->b3[l], w.pp, D, M, D, HP);
    a.c_zstride = loD;
    nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_w3, stream);
    if (nz < 0) return nz;
    const bool last = l + 1 == s->depth;
    memset(&g, 0, sizeof(g));
    g.h = w.h; g.ldh = D; g.P = w.pp; g.nz = nz; g.slab = loD; g.h_out = w.h; g.ldho = D;
    g.norm = 2; g.ng = last ? s->norm_g : s->ln1_g[l + 1]; g.nb = last ? s->norm_b : s->ln1_b[l + 1]; g.eps = 1e-6f;
    if (last) { g.out = sem_out ? sem_out : w.sem; g.ldo = D; g.Y = embed_out ? w.ys : nullptr; g.ldy = D; g.y_lo_off = loD; }
    else { g.Y = w.yd; g.ldy = D; g.y_lo_off = loD; }
    g.M = M; g.D = D;
    wide_glue(g, st);
  }
  if (embed_out) {
    // linear_proj = Linear [GELU Linear]  (modeling_bailingmm.py:111-115)
    const int PD = s->proj_dim;
    mn_g256 a = g256_hilo(w.ys, D, lo_at(LO_SEM_LP, loD), s->proj_w[0], D, s->proj_b[0], s->proj_depth == 1 ? embed_out : w.p0, PD, M, PD, D);
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
    if (s->proj_depth == 2) {
      memset(&g, 0, sizeof(g));
      g.h = w.p0; g.ldh = PD; g.act = 1; g.Y = w.yp; g.ldy = PD; g.y_lo_off = (int64_t)M * PD; g.M = M; g.D = PD;
      wide_glue(g, st);
      a = g256_hilo(w.yp, PD, lo_at(LO_SEM_LP, (int64_t)M * PD), s->proj_w[1], PD, s->proj_b[1], embed_out, PD, M, PD, PD);
      MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
    }
  }
  MN_CHECK_LAUNCH("mn_semdec_step(wide)");
  return MN_OK;
}

// wide_rf.inl — RectifiedFlowLoss.sample (diff_loss_rf_swiglu.py:103-181) for 65..2048 CFG rows advancing in lock-step
// (textually part of engine.hip).  Same arithmetic as the <= 64-row route of mn_rf_sample; at this width the launches are
// MFMA-bound, so every Linear is a gemm256 launch on bf16 hi/lo operands (fp32-class products) and the work between two
// Linears is one wide_glue launch:
//
//   per visual token   vis_head Linear -> LayerNorm -> cond_embed           (modeling_bailing_moe.py:1571-1574, diff_loss:374)
//                      adaLN projections of all Euler steps as ONE GEMM     (diff_loss:263-266, 283-286)
//   per Euler step     [input_proj + in_ln + modulate]                      (diff_loss:371, 270)
//     per ResBlock     w12 GEMM with SwiGLU + hi/lo split in the epilogue   (diff_loss:54-72)
//                      w3 GEMM, split-K partial slabs
//                      [slab reduce + b3 + gated residual + next in_ln / final LN + modulate]   (diff_loss:270-272, 290)
//                      final Linear (split-K) -> bias -> CFG combine + Euler step              (diff_loss:144-179, 291)
// First row count that takes the wide route, per stage (measured crossovers); mn_wide_tune is the A/B hook.
static int g_wide_min_llm = 65, g_wide_min_rf = 41, g_wide_min_sem = 65;
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_wide_tune(int llm_min_rows, int rf_min_rows, int sem_min_rows) {
  if (llm_min_rows > 0) g_wide_min_llm = llm_min_rows;
  if (rf_min_rows > 0) g_wide_min_rf = rf_min_rows;
  if (sem_min_rows > 0) g_wide_min_sem = sem_min_rows;
}
#endif

// Per-site single-pass switch (measurement only, DESIGN.md §2 "lo-pass map"): bit s set = the activation operand of site s enters
// its GEMM as plain bf16 (hi rows only).  The product build has no way to set it; tests/measure/lo_map.py does through the dev library.
enum { LO_RF_VIS = 0, LO_RF_COND, LO_RF_ADA, LO_RF_W12, LO_RF_W3, LO_RF_FIN, LO_LLM_QKV, LO_LLM_DENSE, LO_LLM_GATE, LO_LLM_EXPERTS,
       LO_SEM_QKV, LO_SEM_PROJ, LO_SEM_W12, LO_SEM_W3, LO_SEM_LP };
static unsigned g_lo_drop = 0;
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_lo_drop_mask(unsigned mask) { g_lo_drop = mask; }
#endif
static inline int64_t lo_at(int site, int64_t off) { return ((g_lo_drop >> site) & 1u) ? 0 : off; }

struct RfWideWs {
  float *z, *c, *ada, *hh, *v, *x, *pbuf;
  float* pbuf_stream;    // tp.inl, <= 64 rows: K-slice slabs of the weight-streaming kernels
  float* pbuf_stream3;   // tp.inl, <= 4 rows: w3's slabs when its prologue reads w12's (stream_fuse.h)
  bf16_t *hs, *zs, *y, *ya, *yb;
  bf16_t *wq12, *wq3;    // weight-only modes: the blocks' weights de-quantised once per call, [depth][2 hidden][w] / [depth][w][hidden]
  float* f8s;            // fp8-MFMA regime: the activations' row scales — [steps * rows] adaLN operand, [rows] w12 operand, [rows] w3 operand
  int ks12, ks3, ksf;    // split-K requests of the w12 (1 = SwiGLU in the GEMM epilogue), w3 and final GEMMs
};

// Weight-only modes on the wide route: the byte codes are expanded ONCE per call / layer into a bf16 scratch (exactly the format's W':
// mn_dequant_*_rows, bit-identical to what the streaming kernels' decoders produce) and the bf16 GEMMs run on that — the model keeps
// its reduced footprint, a call pays one pass over the codes (the RF head: 1.8 GB of bf16 per sampler call against ~29 GB of GEMM reads).
extern "C" int mn_dequant_fp8_rows(const uint8_t*, int64_t, const float*, uint16_t*, int64_t, int64_t, int, void*);
extern "C" int mn_dequant_int8_rows(const uint8_t*, int64_t, const float*, uint16_t*, int64_t, int64_t, int, void*);
extern "C" int mn_dequant_nf4_rows(const uint8_t*, int64_t, const float*, uint16_t*, int64_t, int64_t, int, void*);
static int wide_dequant_rows(int wfmt, const void* q, const float* scale, bf16_t* out, int64_t n_rows, int K, void* stream) {
  const uint8_t* qb = reinterpret_cast<const uint8_t*>(q);
  if (wfmt == MN_W_NF4) return mn_dequant_nf4_rows(qb, K / 2, scale, out, K, n_rows, K, stream);
  if (wfmt == MN_W_INT8) return mn_dequant_int8_rows(qb, K, scale, out, K, n_rows, K, stream);
  return mn_dequant_fp8_rows(qb, K, scale, out, K, n_rows, K, stream);
}

// y = silu(gate) * up from the split-K slabs of a w12 GEMM run without its SwiGLU epilogue (columns [0, HID) gate, [HID, 2 HID)
// up; b12 in the same order), stored as the w3 GEMM's bf16 hi / lo operand.  Used when the row count leaves the SwiGLU form of
// the GEMM (one workgroup per 128 rows x 128 hidden units, no split) on a fraction of the chip (diff_loss:54-72).
__global__ __launch_bounds__(256) void rf_swiglu_slabs_kernel(const float* __restrict__ P, int nz, int64_t slab, const bf16_t* __restrict__ b12,
                                                              bf16_t* __restrict__ Y, int64_t lo_off, int rows, int HID) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int q = HID / 4;
  if (idx >= (int64_t)rows * q) return;
  const int m = (int)(idx / q), j = (int)(idx % q) * 4;
  f4 g = {bf16_to_f32(b12[j]), bf16_to_f32(b12[j + 1]), bf16_to_f32(b12[j + 2]), bf16_to_f32(b12[j + 3])};
  f4 u = {bf16_to_f32(b12[HID + j]), bf16_to_f32(b12[HID + j + 1]), bf16_to_f32(b12[HID + j + 2]), bf16_to_f32(b12[HID + j + 3])};
  const float* pp = P + (int64_t)m * 2 * HID + j;
  for (int z = 0; z < nz; ++z) {
    g += *reinterpret_cast<const f4*>(pp + z * slab);
    u += *reinterpret_cast<const f4*>(pp + z * slab + HID);
  }
  uint32_t h0, l0, h1, l1;
  split_pk_bf16(silu_f(g.x) * u.x, silu_f(g.y) * u.y, h0, l0);
  split_pk_bf16(silu_f(g.z) * u.z, silu_f(g.w) * u.w, h1, l1);
  bf16_t* yr = Y + (int64_t)m * HID + j;
  *reinterpret_cast<u2*>(yr) = u2{h0, h1};
  *reinterpret_cast<u2*>(yr + lo_off) = u2{l0, l1};
}

// Split-K request of a hi/lo GEMM with `rows` rows: minimise a small cost model of the launch — rounds of workgroups over the
// 256 CUs x (K-tiles per slice at ~1.9 us each + ~10 us of prologue / epilogue) + the fp32 slab round trip through HBM
// (written by the GEMM, read by the consumer) — over the slice counts that keep >= 4 K-tiles per slice.  E.g. RF w3
// (N = 3072, K = 8192): 1024 rows -> 96 tiles -> 2 slices (192 workgroups, one round); 1536 rows -> 144 tiles -> 3 slices
// (432 workgroups, two rounds of 43 K-tiles instead of one round of 128 on 56 % of the chip).
static int rf_wide_ksplit(int rows, int N, int K) {
  const double tiles = (double)(mn_cdiv(rows, 128) * mn_cdiv(N, 256));
  const int kt = K / 64;
  int best = 1;
  double best_cost = 1e30;
  for (int ks = 1; ks <= 16 && kt / ks >= 4; ++ks) {
    const double rounds = (double)mn_cdiv((int64_t)(tiles * ks), 256);
    const double slabs = ks > 1 ? (double)ks * rows * N * 8.0 / 4.0e6 : 0.0;      // us at ~4 TB/s, write + read
    const double cost = rounds * ((double)mn_cdiv(kt, ks) * 1.9 + 10.0) + slabs;
    if (cost < best_cost) { best_cost = cost; best = ks; }
  }
  return best;
}

// the fp8-MFMA regime of the wide route (mn_rf_head.arith; mingnative.h section 8): e4m3 weights present, widths in whole 128-k tiles
static bool rf_f8_mfma(const mn_rf_head* h) {
  return h->arith == MN_ARITH_FP8_MFMA && h->wfmt == MN_W_FP8_E4M3 && h->ada_q && h->ada_scale && h->w12_scale && h->w3_scale &&
         (h->w % 128) == 0 && (h->hidden % 128) == 0;
}
extern "C" int mn_quant_fp8_rows(const uint16_t*, int64_t, uint8_t*, int64_t, float*, int64_t, int, void*);
static mn_g256 g256_f8(const uint8_t* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw, const float* w_scale, const bf16_t* bias,
                       void* C, int64_t ldc, int M, int N, int K) {
  mn_g256 a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const bf16_t*>(A); a.lda = lda; a.W = reinterpret_cast<const bf16_t*>(W); a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc;
  a.M = M; a.N = N; a.K = K; a.f8 = 1; a.a_scale = a_scale; a.w_scale = w_scale;
  return a;
}

static size_t rf_wide_carve(const mn_rf_head* h, int rows, void* ws, size_t cap, RfWideWs* o) {
  Carver cv(ws, cap, ws == nullptr);
  const int64_t SR = (int64_t)h->steps * rows;
  const int A = h->depth * 3 * h->w + 2 * h->w;
  // w12: the SwiGLU epilogue needs the whole K sum, i.e. cdiv(rows, 128) * hidden / 128 workgroups; below a chip's worth of them
  // the plain split-K form + rf_swiglu_slabs_kernel is the faster pair (128 rows: 69 -> 3x us per block)
  o->ks12 = mn_cdiv(rows, 128) * (h->hidden / 128) >= 256 ? 1 : rf_wide_ksplit(rows, 2 * h->hidden, h->w);
  o->ks3 = rf_wide_ksplit(rows, h->w, h->hidden);
  o->ksf = rf_wide_ksplit(rows, h->target, h->w);
  const size_t p3 = (size_t)mn_gemm256_slices(h->hidden, o->ks3) * rows * h->w;
  const size_t pf = (size_t)mn_gemm256_slices(h->w, o->ksf) * rows * h->target;
  const size_t p12 = o->ks12 > 1 ? (size_t)mn_gemm256_slices(h->w, o->ks12) * rows * 2 * h->hidden : 0;
  o->z = cv.take<float>((size_t)rows * h->z_dim);
  o->c = cv.take<float>((size_t)rows * h->w);
  o->ada = cv.take<float>((size_t)SR * A);
  o->hh = cv.take<float>((size_t)rows * h->w);
  o->v = cv.take<float>((size_t)rows * h->target);
  o->x = cv.take<float>((size_t)rows * h->target);
  o->pbuf = cv.take<float>(p12 > p3 && p12 > pf ? p12 : p3 > pf ? p3 : pf);
  o->hs = cv.take<bf16_t>((size_t)2 * rows * h->llm_hidden);
  o->zs = cv.take<bf16_t>((size_t)2 * rows * h->z_dim);
  o->y = cv.take<bf16_t>((size_t)2 * SR * h->w);
  o->ya = cv.take<bf16_t>((size_t)2 * rows * h->w);
  o->yb = cv.take<bf16_t>((size_t)2 * rows * h->hidden);
  const bool f8 = rf_f8_mfma(h);                          // (the e4m3 bytes ARE the operands: no bf16 expansion of the blocks)
  o->wq12 = cv.take<bf16_t>(h->wfmt && !f8 ? (size_t)h->depth * 2 * h->hidden * h->w : 0);
  o->wq3 = cv.take<bf16_t>(h->wfmt && !f8 ? (size_t)h->depth * h->w * h->hidden : 0);
  o->f8s = cv.take<float>(f8 ? (size_t)SR + 2 * (size_t)rows + 64 : 0);
  return cv.off;
}

static bool rf_wide_ok(const mn_rf_head* h, int rows) {
  const bool fmt_ok = h->wfmt == MN_W_BF16 || ((h->wfmt == MN_W_FP8_E4M3 || h->wfmt == MN_W_INT8 || h->wfmt == MN_W_NF4) && h->w12_scale && h->w3_scale && h->ada_w);
  return fmt_ok && rows >= g_wide_min_rf && rows <= 2048 && wide_glue_ok(h->w) && wide_glue_ok(h->z_dim) && wide_glue_ok(h->llm_hidden) &&
         (h->w % 64) == 0 && (h->hidden % 64) == 0 && (h->z_dim % 64) == 0 && (h->llm_hidden % 64) == 0 && h->target <= 64 &&
         (h->target % 4) == 0;
}

static mn_g256 g256_hilo(const bf16_t* A, int64_t lda, int64_t lo_off, const bf16_t* W, int64_t ldw, const bf16_t* bias, void* C,
                         int64_t ldc, int M, int N, int K) {
  mn_g256 a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.a_lo_off = lo_off; a.W = W; a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  return a;
}

#define MN_TRYZ(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ < 0) return rc__;   \
  } while (0)

static int rf_sample_wide(const mn_rf_head* h, const float* hidden, int64_t ld_hidden, int rows, int n_images, const float* noise,
                          float temperature, float text_cfg, float image_cfg, float* latent_out, void* workspace,
                          size_t workspace_bytes, void* stream) {
  RfWideWs w;
  const size_t need = rf_wide_carve(h, rows, workspace, workspace_bytes, &w);
  if (need > workspace_bytes) { mn_set_error("mn_rf_sample: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  hipStream_t st = mn_stream(stream);
  const int W = h->w, HID = h->hidden, T = h->target, A = h->depth * 3 * W + 2 * W, rpi = rows / n_images;
  const int64_t SR = (int64_t)h->steps * rows;

  // z = vis_head Linear(hidden);  c = cond_embed(LayerNorm(z))
  WideGlue g;
  memset(&g, 0, sizeof(g));
  g.h = hidden; g.ldh = ld_hidden; g.Y = w.hs; g.ldy = h->llm_hidden; g.y_lo_off = (int64_t)rows * h->llm_hidden;
  g.M = rows; g.D = h->llm_hidden;
  wide_glue(g, st);
  mn_g256 a = g256_hilo(w.hs, h->llm_hidden, lo_at(LO_RF_VIS, (int64_t)rows * h->llm_hidden), h->vis_w, h->llm_hidden, h->vis_b, w.z, h->z_dim, rows,
                        h->z_dim, h->llm_hidden);
  MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
  memset(&g, 0, sizeof(g));
  g.h = w.z; g.ldh = h->z_dim; g.norm = 2; g.ng = h->vis_ln_g; g.nb = h->vis_ln_b; g.eps = 1e-6f;
  g.Y = w.zs; g.ldy = h->z_dim; g.y_lo_off = (int64_t)rows * h->z_dim; g.M = rows; g.D = h->z_dim;
  wide_glue(g, st);
  a = g256_hilo(w.zs, h->z_dim, lo_at(LO_RF_COND, (int64_t)rows * h->z_dim), h->cond_w, h->z_dim, h->cond_b, w.c, W, rows, W, h->z_dim);
  MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));

  hipLaunchKernelGGL(rf_init_x_kernel, dim3(mn_cdiv(rows * T, 256)), dim3(256), 0, st, noise, temperature, w.x, rows, T, rpi);
  // modulations of every Euler step: [steps * rows, w] x [w, depth*3w + 2w], adaLN weights read once per token
  hipLaunchKernelGGL(rf_build_y_kernel, dim3(mn_cdiv(SR * W, 256)), dim3(256), 0, st, h->temb, w.c, w.y, h->steps, rows, W);
  const bool f8 = rf_f8_mfma(h);
  // fp8-MFMA regime: every GEMM operand below is the bf16 hi part of what the hi/lo route multiplies, quantised per row to e4m3 (the lo
  // rows of the pair — dead in this regime — hold the bytes); scales in w.f8s
  float* s_ada = w.f8s;
  float* s_a = w.f8s + SR;
  float* s_b = s_a + rows;
  if (f8) {
    uint8_t* y8 = reinterpret_cast<uint8_t*>(w.y + SR * W);
    MN_TRYZ(mn_quant_fp8_rows(w.y, W, y8, W, s_ada, SR, W, stream));
    a = g256_f8(y8, W, s_ada, h->ada_q, W, h->ada_scale, h->ada_b, w.ada, A, (int)SR, A, W);
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
  } else {
    a = g256_hilo(w.y, W, lo_at(LO_RF_ADA, SR * W), h->ada_w, W, h->ada_b, w.ada, A, (int)SR, A, W);
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
  }

  if (h->wfmt && !f8) {          // weight-only mode: W' of every block, once per call
    for (int b = 0; b < h->depth; ++b) {
      MN_TRYZ(wide_dequant_rows(h->wfmt, h->w12[b], h->w12_scale[b], w.wq12 + (int64_t)b * 2 * HID * W, (int64_t)2 * HID, W, stream));
      MN_TRYZ(wide_dequant_rows(h->wfmt, h->w3[b], h->w3_scale[b], w.wq3 + (int64_t)b * W * HID, (int64_t)W, HID, stream));
    }
  }
  const float step = 1.0f / (float)h->steps;
  const int64_t lo_a = (int64_t)rows * W, lo_b = (int64_t)rows * HID;
  for (int s = 0; s < h->steps; ++s) {
    const float* ada = w.ada + (int64_t)s * rows * A;
    // h = input_proj(x); ya = split(in_ln_0(h) * (1 + scale_0) + shift_0)
    memset(&g, 0, sizeof(g));
    g.xin = w.x; g.kin = T; g.win = h->in_w; g.bin = h->in_b; g.h_out = w.hh; g.ldho = W;
    g.norm = 2; g.ng = h->ln_g[0]; g.nb = h->ln_b[0]; g.eps = 1e-6f; g.shift = ada; g.scale = ada + W; g.ldmod = A;
    g.Y = w.ya; g.ldy = W; g.y_lo_off = lo_a; g.M = rows; g.D = W;
    if (f8) { g.y_lo_off = 0; g.Y8 = reinterpret_cast<uint8_t*>(w.ya + lo_a); g.y8_scale = s_a; }
    wide_glue(g, st);
    for (int b = 0; b < h->depth; ++b) {
      const float* mod = ada + (int64_t)b * 3 * W;
      const bf16_t* w12b = h->wfmt ? w.wq12 + (int64_t)b * 2 * HID * W : h->w12[b];
      const bf16_t* w3b = h->wfmt ? w.wq3 + (int64_t)b * W * HID : h->w3[b];
      if (f8) {
        // w12: e4m3(ya) x e4m3(W12) with the SwiGLU in the epilogue -> bf16 [rows, HID]; w3: e4m3 of that x e4m3(W3), split-K slabs
        uint8_t* ya8 = reinterpret_cast<uint8_t*>(w.ya + lo_a);
        bf16_t* yb16 = w.yb;
        uint8_t* yb8 = reinterpret_cast<uint8_t*>(w.yb + lo_b);
        // (ya8 / s_a: written by the glue launch that produced ya — wide_glue's Y8 output, the same bytes mn_quant_fp8_rows makes of ya's hi rows)
        a = g256_f8(ya8, W, s_a, h->w12[b], W, h->w12_scale[b], h->b12[b], yb16, HID, rows, HID, W);
        a.w_pair_rows = HID;
        MN_TRYZ(mn_gemm256_ex(&a, MN_G256_SWIGLU_BF16, 1, stream));
        MN_TRYZ(mn_quant_fp8_rows(yb16, HID, yb8, HID, s_b, rows, HID, stream));
        a = g256_f8(yb8, HID, s_b, h->w3[b], HID, h->w3_scale[b], nullptr, w.pbuf, W, rows, W, HID);
        a.c_zstride = (int64_t)rows * W;
        const int nz8 = mn_gemm256_ex(&a, MN_G256_F32, w.ks3, stream);
        if (nz8 < 0) return nz8;
        const bool last8 = b + 1 == h->depth;
        const float* nmod8 = last8 ? ada + (int64_t)h->depth * 3 * W : ada + (int64_t)(b + 1) * 3 * W;
        memset(&g, 0, sizeof(g));
        g.h = w.hh; g.ldh = W; g.P = w.pbuf; g.nz = nz8; g.slab = (int64_t)rows * W; g.pbias = h->b3[b];
        g.gate = mod + 2 * W; g.ldgate = A; g.h_out = w.hh; g.ldho = W;
        g.norm = 2; g.ng = last8 ? nullptr : h->ln_g[b + 1]; g.nb = last8 ? nullptr : h->ln_b[b + 1]; g.eps = 1e-6f;
        g.shift = nmod8; g.scale = nmod8 + W; g.ldmod = A;
        g.Y = w.ya; g.ldy = W; g.y_lo_off = lo_a; g.M = rows; g.D = W;
        if (!last8) { g.y_lo_off = 0; g.Y8 = ya8; g.y8_scale = s_a; }      // the next block multiplies e4m3: hi rows (unused, kept for inspection) + bytes; the last one feeds the hi/lo final layer
        wide_glue(g, st);
        continue;
      }
      if (w.ks12 > 1) {
        a = g256_hilo(w.ya, W, lo_at(LO_RF_W12, lo_a), w12b, W, nullptr, w.pbuf, 2 * HID, rows, 2 * HID, W);
        a.c_zstride = (int64_t)rows * 2 * HID;
        const int nz12 = mn_gemm256_ex(&a, MN_G256_F32, w.ks12, stream);
        if (nz12 < 0) return nz12;
        hipLaunchKernelGGL(rf_swiglu_slabs_kernel, dim3(mn_cdiv((int64_t)rows * (HID / 4), 256)), dim3(256), 0, st, w.pbuf, nz12,
                           (int64_t)rows * 2 * HID, h->b12[b], w.yb, lo_b, rows, HID);
      } else {
        a = g256_hilo(w.ya, W, lo_at(LO_RF_W12, lo_a), w12b, W, h->b12[b], w.yb, HID, rows, HID, W);
        a.w_pair_rows = HID; a.c_lo_off = lo_b;
        MN_TRYZ(mn_gemm256_ex(&a, MN_G256_SWIGLU_SPLIT, 1, stream));
      }
      a = g256_hilo(w.yb, HID, lo_at(LO_RF_W3, lo_b), w3b, HID, nullptr, w.pbuf, W, rows, W, HID);
      a.c_zstride = (int64_t)rows * W;
      const int nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks3, stream);
      if (nz < 0) return nz;
      const bool last = b + 1 == h->depth;
      const float* nmod = last ? ada + (int64_t)h->depth * 3 * W : ada + (int64_t)(b + 1) * 3 * W;
      memset(&g, 0, sizeof(g));
      g.h = w.hh; g.ldh = W; g.P = w.pbuf; g.nz = nz; g.slab = (int64_t)rows * W; g.pbias = h->b3[b];
      g.gate = mod + 2 * W; g.ldgate = A; g.h_out = w.hh; g.ldho = W;
      g.norm = 2; g.ng = last ? nullptr : h->ln_g[b + 1]; g.nb = last ? nullptr : h->ln_b[b + 1]; g.eps = 1e-6f;
      g.shift = nmod; g.scale = nmod + W; g.ldmod = A;
      g.Y = w.ya; g.ldy = W; g.y_lo_off = lo_a; g.M = rows; g.D = W;
      wide_glue(g, st);
    }
    a = g256_hilo(w.ya, W, lo_at(LO_RF_FIN, lo_a), h->fin_w, W, nullptr, w.pbuf, T, rows, T, W);
    a.c_zstride = (int64_t)rows * T;
    const int nz = mn_gemm256_ex(&a, MN_G256_F32, w.ksf, stream);
    if (nz < 0) return nz;
    hipLaunchKernelGGL(rf_glue_bias_out_kernel, dim3(mn_cdiv(rows * T, 256)), dim3(256), 0, st, w.pbuf, nz, rows, T, h->fin_b, w.v);
    hipLaunchKernelGGL(rf_euler_kernel, dim3(n_images), dim3(256), 0, st, w.v, w.x, rpi, T, text_cfg, image_cfg, step);
  }
  hipLaunchKernelGGL(rf_gather_latent_kernel, dim3(mn_cdiv(n_images * T, 256)), dim3(256), 0, st, w.x, latent_out, n_images, rpi, T);
  MN_CHECK_LAUNCH("mn_rf_sample(wide)");
  return MN_OK;
}

// tp.inl — tensor / expert parallelism of the decode path over one xGMI node (textually part of engine.hip).
//
// Partitioning (no reference counterpart — the reference runs the 16B-A3B model on one device):
//   attention   q heads split over the ranks, each KV head on world / n_kv ranks (16 q / 4 kv heads at TP = 8: 2 q heads + 1 KV
//               head per rank); query_key_value split by rows (column-parallel), dense by columns (row-parallel)
//               (modeling_bailing_moe.py:656-829)                                                  -> all-reduce 1 of a layer
//   experts     E / world routed experts per rank (expert parallelism); every rank holds all rows and the global routing
//               (replicate-and-reduce: no token all-to-all for a few rows), computes its experts' share of sum_k w_k y_k, plus
//               its slice of the shared expert's intermediate width (:556-639)                     -> all-reduce 2 of a layer
//   RF head     w12 split by hidden units (column-parallel), w3 by columns (row-parallel): one all-reduce per ResBlock per
//               Euler step; vis_head / cond / adaLN / input / final layers replicated (diff_loss_rf_swiglu.py:54-72, 263-292)
//
// All-reduce = push + arrival flags + the slab sum the consumer already does.  Every rank owns an INBOX (fine-grained device
// memory, peer-mapped on the other ranks over xGMI): fp32 [2 (epoch parity)][world (sender)][cap], and FLAGS uint32
// [world (sender)][rows_cap].  The producer's tail kernel (tp_push) reduces the local split-K slabs / expert outputs of a row and
// stores the row into slab (parity, rank) of EVERY rank's inbox (posted xGMI writes), fences at system scope and then stores the
// all-reduce's epoch into flag (rank, row) on every rank.  The consumer is the wide_glue launch that follows anyway (residual +
// norm + next operand): row m spins (bounded) on its `world` LOCAL flags, then sums the `world` slabs like split-K slabs.
// One-shot: one xGMI hop, no ring, 4-18 KB per rank at decode sizes.  Two parities suffice: a rank can push all-reduce k + 2 only
// after it consumed k + 1, which every peer pushed after consuming k.  Epochs are monotonic over the communicator's lifetime
// (compared as signed differences), so flags are never reset.
//
// Segments: a composite is a straight-line launch sequence cut at its all-reduces into segments; [seg_begin, seg_end) selects
// which run.  Production runs all of them in one call (waits are real).  The single-GPU parity harness runs segment k of every
// rank before segment k + 1 of any rank on one stream — same kernels, same flags, every wait already satisfied.

struct TpPush {
  const float* P; int nz; int64_t slab;                                    // local split-K slabs [nz][M][D] (or NULL)
  const float* cy; const int32_t* cpos; const float* cw; const int32_t* ci; int n_slot; int e0, e1;   // local experts' outputs (or NULL)
  int cy_nz; int64_t cy_slab;                                              // ... as cy_nz K-slice slabs (0 / 1: one array)
  float* inbox[MN_TP_MAX_WORLD];
  uint32_t* flags[MN_TP_MAX_WORLD];
  int world, rank, rows_cap, M, D;
  int64_t cap;
  uint32_t epoch;
  int scatter_cols;                                                        // two-shot: columns [j gc, (j + 1) gc) go to rank j only (0: the row to every rank)
};

// Row m of this rank's partial: v = sum_z P[z][m] + sum_{s : expert(m, s) local} cw[m, s] * cy[cpos[m, s]]  ->  every rank's inbox.
__global__ __launch_bounds__(1024) void tp_push_kernel(const TpPush p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int m = blockIdx.x, col = threadIdx.x * 4, D = p.D;
  if (col < D) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (p.P) {
      const float* pp = p.P + (int64_t)m * D + col;
      for (int z0 = 0; z0 < p.nz; z0 += 8) {        // eight slabs' loads in flight, added in the one-by-one loop's order (same bits)
        f4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = z0 + j < p.nz ? *reinterpret_cast<const f4*>(pp + (int64_t)(z0 + j) * p.slab) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
      }
    }
    if (p.cy) {
      for (int s = 0; s < p.n_slot; ++s) {
        const int e = p.ci[(int64_t)m * p.n_slot + s];
        if (e >= p.e0 && e < p.e1) {
          const float* yr = p.cy + (int64_t)p.cpos[(int64_t)m * p.n_slot + s] * D + col;
          f4 y = *reinterpret_cast<const f4*>(yr);
          for (int z = 1; z < p.cy_nz; ++z) y += *reinterpret_cast<const f4*>(yr + z * p.cy_slab);
          v += p.cw[(int64_t)m * p.n_slot + s] * y;
        }
      }
    }
    if (p.scatter_cols) {      // two-shot, phase 1 (reduce-scatter): this thread's columns belong to ONE owner
      const int j = col / p.scatter_cols;
      *reinterpret_cast<f4*>(p.inbox[j] + ((int64_t)(p.epoch & 1u) * p.world + p.rank) * p.cap + (int64_t)m * p.scatter_cols + (col - j * p.scatter_cols)) = v;
    } else {
      const int64_t off = ((int64_t)(p.epoch & 1u) * p.world + p.rank) * p.cap + (int64_t)m * D + col;
      for (int r = 0; r < p.world; ++r) *reinterpret_cast<f4*>(p.inbox[r] + off) = v;
    }
  }
  // The row is visible on every rank before any of its flags (cdna_hip_programming.md §6 Guideline 16, flag form): every wave drains
  // its stores, barrier, then the flag lanes alone release at system scope — the wait restated after the fence, where the compiler
  // cannot drop it — and raise the flags with relaxed stores.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < p.world) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(p.flags[threadIdx.x] + (int64_t)p.rank * p.rows_cap + m, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Two-shot all-reduce, phase 2 (owner reduce + all-gather): rank j waits for the `world` pieces of its column slice [j gc, (j + 1) gc) of
// row m (epoch `e`), sums them in sender order and pushes the reduced piece into slab (parity of e + 1, sender j) of EVERY rank's
// inbox, then raises flag (j, m) = e + 1 everywhere.  Bytes per rank and all-reduce: 2 (world - 1) / world x M D 4 against (world - 1) x
// M D 4 for the one-shot form — 4 x fewer at world = 8.  Same bounded wait, same poisoning on expiry as the consumer glue.
struct TpGather {
  float* inbox[MN_TP_MAX_WORLD];
  uint32_t* flags[MN_TP_MAX_WORLD];
  int world, rank, rows_cap, M, gc;
  int64_t cap;
  uint32_t epoch;                                                          // of phase 1; phase 2 publishes epoch + 1
  uint32_t* err; uint64_t wait_ticks;
};
__global__ __launch_bounds__(1024) void tp_reduce_gather_kernel(const TpGather p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int m = blockIdx.x, col = threadIdx.x * 4;
  int expired = 0;
  if ((int)threadIdx.x < p.world) {
    const uint32_t* f = p.flags[p.rank] + (int64_t)threadIdx.x * p.rows_cap + m;
    const uint64_t t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - p.epoch) < 0) {
      if (wall_clock64() - t0 > p.wait_ticks) { if (p.err) atomicExch(p.err, 0x200u | (unsigned)threadIdx.x); expired = 1; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  const int dead = __syncthreads_or(expired);
  if (col < p.gc) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    const float* mine = p.inbox[p.rank] + (int64_t)(p.epoch & 1u) * p.world * p.cap + (int64_t)m * p.gc + col;
    for (int s0 = 0; s0 < p.world; s0 += 8) {        // eight senders' loads in flight, added in sender order (every rank: the same bits)
      f4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = s0 + j < p.world ? *reinterpret_cast<const f4*>(mine + (int64_t)(s0 + j) * p.cap) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) v += t[j];
    }
    if (dead) { const float q = __builtin_nanf(""); v = f4{q, q, q, q}; }
    const int64_t off = ((int64_t)((p.epoch + 1u) & 1u) * p.world + p.rank) * p.cap + (int64_t)m * p.gc + col;
    for (int r = 0; r < p.world; ++r) *reinterpret_cast<f4*>(p.inbox[r] + off) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < p.world) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(p.flags[threadIdx.x] + (int64_t)p.rank * p.rows_cap + m, p.epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Which form does an all-reduce of M rows x D take on this communicator?  One-shot up to two_shot_rows rows (default 16: one xGMI hop
// is the point at decode sizes), two-shot above when the row splits into `world` pieces of whole float4s.  -> columns per owner, or 0.
static int g_tp_two_shot_rows = 16;
#ifdef MN_DEV_HOOKS
extern "C" MN_DEV_API void mn_tp_tune_two_shot(int rows) { g_tp_two_shot_rows = rows; }
#endif
static int tp_two_shot_cols(const mn_tp_comm* c, int M, int D) {
  const int lim = c->two_shot_rows ? c->two_shot_rows : g_tp_two_shot_rows;
  if (lim < 0 || M <= lim || c->world < 2 || (D % (4 * c->world)) != 0) return 0;
  return D / c->world;
}

static bool tp_comm_ok(const mn_tp_comm* c, int rows, int D) {
  return c && c->world >= 1 && c->world <= MN_TP_MAX_WORLD && c->rank >= 0 && c->rank < c->world && c->inbox && c->flags &&
         rows <= c->rows_cap && (int64_t)rows * D <= c->cap && (D % 4) == 0 && D >= 64 && D <= 4096;
}

// gc: 0 = one-shot (the row to every rank); > 0 = two-shot phase 1 (column piece j to rank j)
static void tp_push(const mn_tp_comm* c, uint32_t epoch, TpPush p, int M, int D, hipStream_t st, int gc = 0) {
  for (int r = 0; r < c->world; ++r) { p.inbox[r] = c->inbox[r]; p.flags[r] = c->flags[r]; }
  p.world = c->world; p.rank = c->rank; p.rows_cap = c->rows_cap; p.cap = c->cap; p.epoch = epoch; p.M = M; p.D = D; p.scatter_cols = gc;
  hipLaunchKernelGGL(tp_push_kernel, dim3(M), dim3(((D / 4 + 63) / 64) * 64), 0, st, p);
}

// two-shot phase 2: wait for phase 1 (epoch), reduce this rank's piece, publish it everywhere (epoch + 1)
static void tp_reduce_gather(const mn_tp_comm* c, uint32_t epoch, int M, int gc, hipStream_t st) {
  TpGather p;
  for (int r = 0; r < c->world; ++r) { p.inbox[r] = c->inbox[r]; p.flags[r] = c->flags[r]; }
  p.world = c->world; p.rank = c->rank; p.rows_cap = c->rows_cap; p.cap = c->cap; p.epoch = epoch; p.M = M; p.gc = gc; p.err = c->err;
  p.wait_ticks = (uint64_t)(c->wait_ms ? c->wait_ms : MN_TP_WAIT_MS_DEFAULT) * 100000ull;
  const int threads = ((gc / 4 + 63) / 64) * 64;
  hipLaunchKernelGGL(tp_reduce_gather_kernel, dim3(M), dim3(threads < 64 ? 64 : threads), 0, st, p);
}

// Consumer side: the glue's slab sum reads this rank's inbox (one slab per sender) after the row's arrival flags.
// gc > 0: the all-reduce was two-shot — `epoch` is its SECOND epoch and the slabs hold the reduced row in pieces of gc columns.
static void tp_consume(WideGlue& g, const mn_tp_comm* c, uint32_t epoch, int gc = 0) {
  g.P = c->inbox[c->rank] + (int64_t)(epoch & 1u) * c->world * c->cap;
  g.nz = c->world; g.slab = c->cap; g.gather_cols = gc;
  g.wait_flags = c->flags[c->rank]; g.wait_n = c->world; g.wait_stride = c->rows_cap; g.wait_epoch = epoch; g.wait_err = c->err;
  g.wait_ticks = (uint64_t)(c->wait_ms ? c->wait_ms : MN_TP_WAIT_MS_DEFAULT) * 100000ull;      // ms -> ticks of the 100 MHz clock
}

// out[m] = sum over ranks of x[m] (fp32 [M, D] partial per rank).  Stand-alone form of the mechanism above.  phase: MN_TP_PUSH
// (publish x to every rank), MN_TP_REDUCE (wait for the row's arrivals, sum into out; completes the all-reduce and advances the
// epoch) or both; split phases let the caller put other work between them (or run several ranks from one process).
extern "C" int mn_tp_allreduce_segments(const mn_tp_comm* comm, int rows, int D) {
  MN_CHECK_ARG(comm != nullptr, "mn_tp_allreduce_segments: null communicator");
  return tp_two_shot_cols(comm, rows, D) ? 2 : 1;
}

// (name kept from round 3; above two_shot_rows rows the call runs the two-shot form — phases MN_TP_PUSH | MN_TP_GATHER | MN_TP_REDUCE)
extern "C" int mn_allreduce_oneshot(mn_tp_comm* comm, const float* x, int64_t ldx, float* out, int64_t ldo, int M, int D, int phase,
                                    void* stream) {
  MN_CHECK_ARG(M >= 1 && tp_comm_ok(comm, M, D) && (phase & ~7) == 0 && phase != 0 && (!(phase & MN_TP_PUSH) || (x && ldx == D)) &&
                   (!(phase & MN_TP_REDUCE) || (out && (ldo % 4) == 0)),
               "mn_allreduce_oneshot: bad args (rows <= rows_cap, rows * D <= cap, D %% 4 == 0, 64 <= D <= 4096, ldx == D)");
  hipStream_t st = mn_stream(stream);
  const int gc = tp_two_shot_cols(comm, M, D);
  const uint32_t ep = comm->epoch + 1;                 // phase 1's epoch; a two-shot all-reduce completes at ep + 1
  if (phase & MN_TP_PUSH) {
    TpPush p;
    memset(&p, 0, sizeof(p));
    p.P = x; p.nz = 1; p.slab = 0;
    tp_push(comm, ep, p, M, D, st, gc);
  }
  // PUSH | REDUCE in ONE call is the documented "whole all-reduce" (0.1.2x callers pass 3): above the one-shot row limit the owners'
  // reduce-and-publish step belongs to it — without it REDUCE would wait for an epoch nobody publishes.  Split-phase callers
  // (PUSH alone, then GATHER, then REDUCE) are unchanged.
  const bool whole = (phase & (MN_TP_PUSH | MN_TP_REDUCE)) == (MN_TP_PUSH | MN_TP_REDUCE);
  if (((phase & MN_TP_GATHER) || whole) && gc) tp_reduce_gather(comm, ep, M, gc, st);
  if (phase & MN_TP_REDUCE) {
    WideGlue g;
    memset(&g, 0, sizeof(g));
    tp_consume(g, comm, gc ? ep + 1 : ep, gc);
    if (gc) {
      // the pieces ARE the result: add them to a zero source row — reuse piece 0's own slab as a source is not possible (pieces are
      // concatenated), so the glue reads h = out after clearing it
      // (only the payload: `out` may be a strided view — columns [D, ldo) belong to the caller, and row M - 1 ends at column D)
      if (hipMemset2DAsync(out, (size_t)ldo * sizeof(float), 0, (size_t)D * sizeof(float), (size_t)M, st) != hipSuccess) {
        mn_set_error("mn_allreduce_oneshot: hipMemset2DAsync failed");
        return MN_ELAUNCH;
      }
      g.h = out; g.ldh = ldo;
    } else {
      // the glue adds the slabs P[0 .. nz) to a source row: take sender 0's slab as the source and sum the other world - 1
      g.h = g.P; g.ldh = D; g.P += g.slab; g.nz -= 1;
      if (g.nz == 0) g.P = nullptr;
    }
    g.out = out; g.ldo = ldo; g.M = M; g.D = D;
    wide_glue(g, st);
    comm->epoch = gc ? ep + 1 : ep;
  }
  MN_CHECK_LAUNCH("mn_allreduce_oneshot");
  return MN_OK;
}

// Expert-parallel combine (moe_infer's weighted un-permute, modeling_bailing_moe.py:630-639, across ranks): rank-local
//   part[m] = sum_{s : topk_idx[m, s] in [expert0, expert0 + n_local)} topk_w[m, s] * yg[slot_of[m, s]]  (+ sum_z P[z][m])
// pushed to every rank; out[m] = h[m] + sum over ranks of part[m].  P: optional extra fp32 slabs of the rank (its slice of the
// shared expert's down projection).  Counterpart of mn_ep_dispatch.
extern "C" int mn_ep_combine(mn_tp_comm* comm, const float* yg, const int32_t* slot_of, const int32_t* topk_idx, const float* topk_w,
                             int n_slot, int expert0, int n_local, const float* P, int nz, int64_t slab, const float* h, int64_t ldh,
                             float* out, int64_t ldo, int M, int D, void* stream) {
  MN_CHECK_ARG(yg && slot_of && topk_idx && topk_w && h && out && n_slot >= 1 && n_local >= 1 && M >= 1 && tp_comm_ok(comm, M, D) &&
                   (ldh % 4) == 0 && (ldo % 4) == 0 && (!P || nz >= 1), "mn_ep_combine: bad args");
  hipStream_t st = mn_stream(stream);
  const uint32_t ep = comm->epoch + 1;
  TpPush p;
  memset(&p, 0, sizeof(p));
  p.P = P; p.nz = nz; p.slab = slab;
  p.cy = yg; p.cy_nz = 1; p.cpos = slot_of; p.cw = topk_w; p.ci = topk_idx; p.n_slot = n_slot; p.e0 = expert0; p.e1 = expert0 + n_local;
  tp_push(comm, ep, p, M, D, st);                    // (stand-alone form: one call per rank, so always one-shot)
  WideGlue g;
  memset(&g, 0, sizeof(g));
  g.h = h; g.ldh = ldh;
  tp_consume(g, comm, ep);
  g.out = out; g.ldo = ldo; g.M = M; g.D = D;
  wide_glue(g, st);
  comm->epoch = ep;
  MN_CHECK_LAUNCH("mn_ep_combine");
  return MN_OK;
}

// ===========================================================================================
// Bailing-MoE decoder stack, one step, tensor + expert parallel
// ===========================================================================================
struct LlmTpWs {
  LlmWideWs w;
  bf16_t* ysh;          // hi/lo operand of the shared slice's down projection [2][rows][shared_inter]
  float* psh;           // its split-K slabs
  int ks_sh3;
  // <= 64 rows: K-slice slabs of the weight-streaming kernels (QKV / dense / gate x 2 / shared gate-up), grouped expert slabs
  float *pps, *p1, *p2;
};

static bool llm_tp_ok(const mn_llm* m, const mn_llm_tp* tp, const mn_tp_comm* c, int rows) {
  if (!m || !tp || rows < 1 || rows > 2048 || !tp_comm_ok(c, rows, m->hidden)) return false;
  const int ad = m->n_q * m->head_dim;
  return wide_glue_ok(m->hidden) && (m->hidden % 64) == 0 && (ad % 64) == 0 && (m->moe_inter % 64) == 0 && m->n_experts <= 64 &&
         (m->n_experts % 4) == 0 && (int64_t)rows * m->top_k <= 65536 && (m->head_dim == 64 || m->head_dim == 128) &&
         m->n_q % m->n_kv == 0 && m->n_shared_slots == 0 && tp->n_local_experts >= 1 && tp->expert0 >= 0 &&
         tp->expert0 + tp->n_local_experts <= m->n_experts && tp->shared_inter >= 0 && (tp->shared_inter % 64) == 0 &&
         (tp->shared_inter == 0 || (tp->ws_gate_up && tp->ws_down));
}

static size_t llm_tp_carve(const mn_llm* m, const mn_llm_tp* tp, int rows, int64_t t_max, void* ws, size_t cap, LlmTpWs* o) {
  // the wide route's carve on the SHARD's dimensions (local heads; n_experts = the global count: the router is replicated) ...
  size_t off = llm_wide_carve(m, rows, t_max, ws, cap, &o->w);
  Carver cv(ws ? (char*)ws + off : nullptr, cap > off ? cap - off : 0, ws == nullptr);
  // ... + the shared slice: its down projection's slabs share pp (sized below), its operand is ysh
  o->ks_sh3 = tp->shared_inter ? rf_wide_ksplit(rows, m->hidden, tp->shared_inter) : 1;
  o->ysh = cv.take<bf16_t>((size_t)2 * rows * (tp->shared_inter ? tp->shared_inter : 64));
  const int SIc = tp->shared_inter ? tp->shared_inter : 64;
  size_t psh = (size_t)mn_gemm256_slices(SIc, o->ks_sh3) * rows * m->hidden;
  o->pps = o->p1 = o->p2 = nullptr;
  if (rows <= 64) {
    const int H = m->hidden, ad = m->n_q * m->head_dim, qkv_dim = (m->n_q + 2 * m->n_kv) * m->head_dim, I = m->moe_inter;
    const size_t Pn = (size_t)rows * m->top_k;
    size_t pmax = (size_t)mn_stream_mfma_slices(rows, qkv_dim, H) * qkv_dim;
    const size_t c2 = (size_t)mn_stream_mfma_slices(rows, H, ad) * H, c3 = (size_t)2 * mn_stream_mfma_slices(rows, m->n_experts, H) * m->n_experts;
    const size_t c4 = (size_t)stream_slices(m->wfmt, rows, 2 * SIc, H) * 2 * SIc;
    if (c2 > pmax) pmax = c2;
    if (c3 > pmax) pmax = c3;
    if (c4 > pmax) pmax = c4;
    const size_t s3 = (size_t)stream_slices(m->wfmt, rows, H, SIc) * rows * H;
    if (s3 > psh) psh = s3;
    o->pps = cv.take<float>(pmax * rows);
    o->p1 = cv.take<float>((size_t)mn_stream_mfma_grouped_slices(m->n_experts, rows, 2 * I, H) * Pn * 2 * I);
    o->p2 = cv.take<float>((size_t)mn_stream_mfma_grouped_slices(m->n_experts, rows, H, I) * Pn * H);
  }
  o->psh = cv.take<float>(psh);
  return off + cv.off;
}

extern "C" size_t mn_llm_tp_workspace_bytes(const mn_llm* m, const mn_llm_tp* tp, int rows, int64_t t_max) {
  LlmTpWs w;
  return llm_tp_carve(m, tp, rows, t_max, nullptr, 0, &w);
}

extern "C" int mn_llm_tp_segments(const mn_llm* m) { return 2 * m->n_layers + 1; }      // one-shot all-reduces (<= two_shot_rows rows)

extern "C" int mn_llm_step_tp(const mn_llm* m, const mn_llm_tp* tp, mn_tp_comm* comm, const float* x, int64_t ldx, int x_row_div, int M,
                              const uint8_t* image_mask, const int32_t* row_seq, const int32_t* row_slot, const int32_t* row_pos,
                              const int32_t* row_len, const uint8_t* key_mask, int64_t ld_mask, float* kv_cache, int n_seq, int64_t t_max,
                              float* hidden_out, void* workspace, size_t workspace_bytes, int seg_begin, int seg_end, void* stream) {
  MN_CHECK_ARG(m && tp && comm && x && row_seq && row_slot && row_pos && row_len && kv_cache && hidden_out && workspace,
               "mn_llm_step_tp: null pointer");
  MN_CHECK_ARG(llm_tp_ok(m, tp, comm, M) && x_row_div >= 1, "mn_llm_step_tp: unsupported shard / communicator for M=%d rows", M);
  MN_CHECK_ARG(m->wfmt == MN_W_BF16 || (mn_w8(m->wfmt) && M <= 64 && m->w_gate_up_scale && m->w_down_scale && (m->hidden % mn_wq_kmult(m->wfmt)) == 0 &&
                                        (m->moe_inter % mn_wq_kmult(m->wfmt)) == 0 && (tp->shared_inter % mn_wq_kmult(m->wfmt)) == 0 &&
                                        (tp->shared_inter == 0 || (tp->ws_gate_up_scale && tp->ws_down_scale))),
               "mn_llm_step_tp: quantised experts need scale tables, widths %% 16 == 0 (NF4: %% 64) and <= 64 rows (M = %d)", M);
  // above two_shot_rows rows every all-reduce is two-shot: one more segment (the owners' reduce + all-gather) and one more epoch each
  const int gc = tp_two_shot_cols(comm, M, m->hidden), ars = gc ? 2 : 1;
  const int n_seg = 2 * ars * m->n_layers + 1;
  MN_CHECK_ARG(seg_begin >= 0 && seg_begin < seg_end && seg_end <= n_seg, "mn_llm_step_tp: segments [%d, %d) of %d", seg_begin, seg_end, n_seg);
  LlmTpWs tw;
  const size_t need = llm_tp_carve(m, tp, M, t_max, workspace, workspace_bytes, &tw);
  if (need > workspace_bytes) { mn_set_error("mn_llm_step_tp: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  LlmWideWs& w = tw.w;
  float* psh = tw.psh;
  hipStream_t st = mn_stream(stream);
  const int H = m->hidden, hd = m->head_dim, nq = m->n_q, nkv = m->n_kv, I = m->moe_inter, E = m->n_experts, SI = tp->shared_inter;
  const int ad = nq * hd, qkv_dim = (nq + 2 * nkv) * hd, n_slot = m->top_k, e0 = tp->expert0, e1 = tp->expert0 + tp->n_local_experts;
  const int64_t P = (int64_t)M * n_slot;
  const int64_t layer_kv = (int64_t)n_seq * 2 * nkv * t_max * hd;
  const float q_scale = 1.0f / sqrtf((float)hd);
  int seg = 0;
  uint32_t ep = comm->epoch;                       // epoch of the all-reduce the NEXT consumer waits for
  auto on = [&]() { return seg >= seg_begin && seg < seg_end; };
  const bool streaming = M <= 64 && (I % 8) == 0 && (SI % 8) == 0;
  WideGlue g;
  for (int l = 0; l <= m->n_layers; ++l) {
    const bool fin = l == m->n_layers;
    // ---- segment 2l: glue (stack input | h += all-reduced expert partials of layer l-1) -> RMSNorm(ln1 | final norm)
    if (on()) {
      memset(&g, 0, sizeof(g));
      if (l == 0) { g.x = x; g.ldx = ldx; g.x_row_div = x_row_div; }
      else { g.h = w.h; g.ldh = H; tp_consume(g, comm, ep, gc); }
      g.h_out = fin ? nullptr : w.h; g.ldho = H;
      g.norm = 1; g.ng = fin ? m->final_norm : m->ln1[l]; g.eps = m->rms_eps;
      if (fin) { g.out = hidden_out; g.ldo = H; }
      else { g.Y = w.yh; g.ldy = H; g.y_lo_off = (int64_t)M * H; }
      g.M = M; g.D = H;
      wide_glue(g, st);
    }
    if (fin) break;
    float* kv_l = kv_cache + (int64_t)l * layer_kv;
    if (on()) {
      // local heads: QKV rows of this rank -> RoPE + KV append (its KV head) -> masked GQA -> dense over its columns  (:743-829)
      mn_g256 a;
      int nz;
      float* pp = streaming ? tw.pps : w.pp;        // <= 64 rows: the weight-streaming MFMA kernels on the shard's slices
      if (streaming) {
        nz = mn_stream_mfma(w.yh, m->wqkv[l], pp, M, qkv_dim, H, stream);
      } else {
        a = g256_hilo(w.yh, H, (int64_t)M * H, m->wqkv[l], H, nullptr, pp, qkv_dim, M, qkv_dim, H);
        a.c_zstride = (int64_t)M * qkv_dim;
        nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_qkv, stream);
      }
      if (nz < 0) return nz;
      MN_TRY(mn_rope_kv_from_partials(pp, qkv_dim, nz, (int64_t)M * qkv_dim, M, nq, nkv, hd, 1, m->cos_tab, m->sin_tab, row_seq,
                                      row_slot, row_pos, m->mrope_sec_t, m->mrope_sec_h, q_scale, w.q, kv_l, t_max, stream));
      MN_TRY(mn_attn_decode_split(w.q, M, nq, nkv, hd, kv_l, t_max, row_seq, row_len, key_mask, ld_mask, nullptr, w.ya, w.attn_ws,
                                  w.attn_ws_bytes, stream));
      if (streaming) {
        nz = mn_stream_mfma(w.ya, m->wdense[l], pp, M, H, ad, stream);
      } else {
        a = g256_hilo(w.ya, ad, (int64_t)M * ad, m->wdense[l], ad, nullptr, pp, H, M, H, ad);
        a.c_zstride = (int64_t)M * H;
        nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_dense, stream);
      }
      if (nz < 0) return nz;
      TpPush p;                                     // all-reduce 1: the dense partial of this rank's heads
      memset(&p, 0, sizeof(p));
      p.P = pp; p.nz = nz; p.slab = (int64_t)M * H;
      tp_push(comm, ep + 1, p, M, H, st, gc);
    }
    ++seg; ++ep;
    if (gc) { if (on()) tp_reduce_gather(comm, ep, M, gc, st); ++seg; ++ep; }      // two-shot: the owners reduce and publish
    // ---- next segment: h += all-reduced attention partials; RMSNorm(ln2); router (replicated); local experts + shared slice
    if (on()) {
      memset(&g, 0, sizeof(g));
      g.h = w.h; g.ldh = H; tp_consume(g, comm, ep, gc); g.h_out = w.h; g.ldho = H;
      g.norm = 1; g.ng = m->ln2[l]; g.eps = m->rms_eps; g.Y = w.yh; g.ldy = H; g.y_lo_off = (int64_t)M * H; g.M = M; g.D = H;
      wide_glue(g, st);
      mn_g256 a;
      int nz;
      float* pp = streaming ? tw.pps : w.pp;
      if (streaming) {
        nz = mn_stream_mfma(w.yh, m->gate[l], pp, M, E, H, stream);
      } else {
        a = g256_hilo(w.yh, H, (int64_t)M * H, m->gate[l], H, nullptr, pp, E, M, E, H);
        a.c_zstride = (int64_t)M * E;
        nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks_gate, stream);
      }
      if (nz < 0) return nz;
      const float* p_img = nullptr;
      if (image_mask && m->image_gate && m->image_gate[l]) {
        float* pi = pp + (int64_t)nz * M * E;
        int nzi;
        if (streaming) {
          nzi = mn_stream_mfma(w.yh, m->image_gate[l], pi, M, E, H, stream);
        } else {
          a = g256_hilo(w.yh, H, (int64_t)M * H, m->image_gate[l], H, nullptr, pi, E, M, E, H);
          a.c_zstride = (int64_t)M * E;
          nzi = mn_gemm256_ex(&a, MN_G256_F32, w.ks_gate, stream);
        }
        if (nzi < 0) return nzi;
        if (nzi != nz) { mn_set_error("mn_llm_step_tp: gate / image gate slab counts differ"); return MN_EINVAL; }
        p_img = pi;
      }
      hipLaunchKernelGGL(moe_topk_partials_kernel, dim3(mn_cdiv(M, 4)), dim3(256), 0, st, (const float*)pp, p_img, image_mask, nz,
                         (int64_t)M * E, M, E, m->top_k, m->norm_topk_prob, 0, w.ti, w.tw);
      MN_TRY(mn_ep_dispatch(w.ti, M, n_slot, E, e0, e1 - e0, w.cnt, w.off, w.perm, w.slot_of, 128, w.tile_g, w.tile_m0, w.n_tiles, stream));
      TpPush p;                                     // all-reduce 2: local experts' weighted sum + shared slice
      memset(&p, 0, sizeof(p));
      int nzs = 0;
      if (streaming) {
        // local experts on the grouped K-loop kernel: group g of the window = expert e0 + g, its rows are the sorted positions
        // [off[e0 + g], off[e0 + g + 1]) of the GLOBAL sort (gathered through perm); every distinct local expert is streamed once
        int nz1 = stream_grouped(m->wfmt, w.yh, M, m->w_gate_up[l], (int64_t)2 * I * H, m->wfmt ? m->w_gate_up_scale[l] : nullptr, 2 * I,
                                 tw.p1, (int)P, w.off + e0, w.perm, e1 - e0, M, 2 * I, H, stream);
        if (nz1 < 0) return nz1;
        hipLaunchKernelGGL(rf_glue_swiglu_split_kernel, dim3(mn_cdiv(P * I, 1024)), dim3(256), 0, st, (const float*)tw.p1, nz1, (int)P, I,
                           (const bf16_t*)nullptr, w.y2, (const int32_t*)(w.off + e0), (const int32_t*)(w.off + e1));
        const int nz2 = stream_grouped(m->wfmt, w.y2, (int)P, m->w_down[l], (int64_t)H * I, m->wfmt ? m->w_down_scale[l] : nullptr, H,
                                       tw.p2, (int)P, w.off + e0, nullptr, e1 - e0, M, H, I, stream);
        if (nz2 < 0) return nz2;
        p.cy = tw.p2; p.cy_nz = nz2; p.cy_slab = P * H;
        if (SI) {
          int nzg = stream_dense(m->wfmt, w.yh, tp->ws_gate_up[l], m->wfmt ? tp->ws_gate_up_scale[l] : nullptr, pp, M, 2 * SI, H, stream);
          if (nzg < 0) return nzg;
          hipLaunchKernelGGL(rf_glue_swiglu_split_kernel, dim3(mn_cdiv((int64_t)M * SI, 1024)), dim3(256), 0, st, (const float*)pp, nzg, M,
                             SI, (const bf16_t*)nullptr, tw.ysh);
          nzs = stream_dense(m->wfmt, tw.ysh, tp->ws_down[l], m->wfmt ? tp->ws_down_scale[l] : nullptr, psh, M, H, SI, stream);
          if (nzs < 0) return nzs;
        }
      } else {
        // local experts (weights biased by -e0 groups: only local group ids appear in the tile list)  (:617-628, 483-484)
        a = g256_hilo(w.yh, H, (int64_t)M * H, m->w_gate_up[l] - (int64_t)e0 * 2 * I * H, H, nullptr, w.y2, I, M, I, H);
        a.w_pair_rows = I; a.c_lo_off = P * I;
        a.g_off = w.off; a.g_cnt = w.cnt; a.w_gstride = (int64_t)2 * I * H; a.a_rows = w.perm; a.n_groups = E;
        a.tile_g = w.tile_g; a.tile_m0 = w.tile_m0; a.n_tiles = w.n_tiles; a.max_mtiles = w.max_mtiles;
        MN_TRYZ(mn_gemm256_ex(&a, MN_G256_SWIGLU_SPLIT, 1, stream));
        a = g256_hilo(w.y2, I, P * I, m->w_down[l] - (int64_t)e0 * H * I, I, nullptr, w.yg, H, M, H, I);
        a.g_off = w.off; a.g_cnt = w.cnt; a.w_gstride = (int64_t)H * I; a.n_groups = E;
        a.tile_g = w.tile_g; a.tile_m0 = w.tile_m0; a.n_tiles = w.n_tiles; a.max_mtiles = w.max_mtiles;
        MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
        p.cy = w.yg; p.cy_nz = 1;
        // this rank's slice of the shared expert (column-parallel gate/up, row-parallel down)  (:599-606)
        if (SI) {
          a = g256_hilo(w.yh, H, (int64_t)M * H, tp->ws_gate_up[l], H, nullptr, tw.ysh, SI, M, SI, H);
          a.w_pair_rows = SI; a.c_lo_off = (int64_t)M * SI;
          MN_TRYZ(mn_gemm256_ex(&a, MN_G256_SWIGLU_SPLIT, 1, stream));
          a = g256_hilo(tw.ysh, SI, (int64_t)M * SI, tp->ws_down[l], SI, nullptr, psh, H, M, H, SI);
          a.c_zstride = (int64_t)M * H;
          nzs = mn_gemm256_ex(&a, MN_G256_F32, tw.ks_sh3, stream);
          if (nzs < 0) return nzs;
        }
      }
      if (SI) { p.P = psh; p.nz = nzs; p.slab = (int64_t)M * H; }
      p.cpos = w.slot_of; p.cw = w.tw; p.ci = w.ti; p.n_slot = n_slot; p.e0 = e0; p.e1 = e1;
      tp_push(comm, ep + 1, p, M, H, st, gc);
    }
    ++seg; ++ep;
    if (gc) { if (on()) tp_reduce_gather(comm, ep, M, gc, st); ++seg; ++ep; }
  }
  if (seg_end == n_seg) comm->epoch += 2u * (uint32_t)ars * (uint32_t)m->n_layers;
  MN_CHECK_LAUNCH("mn_llm_step_tp");
  return MN_OK;
}

// ===========================================================================================
// Rectified-flow head, tensor parallel over the SwiGLU hidden width
// ===========================================================================================
extern "C" int mn_rf_tp_segments(const mn_rf_head* h) { return h->steps * h->depth + 1; }

// the wide route's carve on the shard + room in pbuf for the streaming kernels' K-slice slabs (<= 64 rows)
static size_t rf_tp_carve(const mn_rf_head* h, int rows, void* ws, size_t cap, RfWideWs* o) {
  const size_t off = rf_wide_carve(h, rows, ws, cap, o);
  if (rows > 64) return off;
  const size_t p12 = (size_t)stream_slices(h->wfmt, rows, 2 * h->hidden, h->w) * 2 * h->hidden * rows;
  const size_t p3 = (size_t)stream_slices(h->wfmt, rows, h->w, h->hidden) * h->w * rows;
  Carver cv(ws ? (char*)ws + off : nullptr, cap > off ? cap - off : 0, ws == nullptr);
  float* pb = cv.take<float>(p12 > p3 ? p12 : p3);
  // <= 4 rows: w3 builds SwiGLU(w12's slabs) in its prologue (stream_fuse.h) and needs its own slab area
  float* pb3 = cv.take<float>(rf_fused_shape_ok(h, rows) ? p3 : 0);
  if (ws) { o->pbuf_stream = pb; o->pbuf_stream3 = pb3; }
  return off + cv.off;
}

extern "C" size_t mn_rf_tp_workspace_bytes(const mn_rf_head* h, int rows) {
  RfWideWs w;
  return rf_tp_carve(h, rows, nullptr, 0, &w);
}

// h: the rank's shard — hidden = its share of the SwiGLU width (w12 [2 * hidden, w] gate rows then up rows, b12 [2 * hidden],
// w3 [w, hidden]); b3 and every other tensor are the full (replicated) ones.  Same arguments as mn_rf_sample otherwise.
extern "C" int mn_rf_sample_tp(const mn_rf_head* h, mn_tp_comm* comm, const float* hidden, int64_t ld_hidden, int rows, int n_images,
                               const float* noise, float temperature, float text_cfg, float image_cfg, float* latent_out,
                               void* workspace, size_t workspace_bytes, int seg_begin, int seg_end, void* stream) {
  MN_CHECK_ARG(h && comm && hidden && noise && latent_out && workspace && rows >= 1 && n_images >= 1 && rows % n_images == 0,
               "mn_rf_sample_tp: bad args");
  MN_CHECK_ARG(rows <= 2048 && wide_glue_ok(h->w) && wide_glue_ok(h->z_dim) && wide_glue_ok(h->llm_hidden) && (h->w % 64) == 0 &&
                   (h->hidden % 64) == 0 && (h->z_dim % 64) == 0 && (h->llm_hidden % 64) == 0 && h->target <= 64 && (h->target % 4) == 0 &&
                   tp_comm_ok(comm, rows, h->w), "mn_rf_sample_tp: unsupported widths / communicator");
  MN_CHECK_ARG(h->wfmt == MN_W_BF16 || (rf_fp8_ok(h) && rows <= 64), "mn_rf_sample_tp: fp8 weights need row scales, widths %% 16 == 0 and <= 64 rows");
  const int gc = tp_two_shot_cols(comm, rows, h->w), ars = gc ? 2 : 1;      // two-shot all-reduces above two_shot_rows rows (one more segment each)
  const int n_seg = ars * h->steps * h->depth + 1;
  MN_CHECK_ARG(seg_begin >= 0 && seg_begin < seg_end && seg_end <= n_seg, "mn_rf_sample_tp: segments [%d, %d) of %d", seg_begin, seg_end, n_seg);
  RfWideWs w;
  const size_t need = rf_tp_carve(h, rows, workspace, workspace_bytes, &w);
  if (need > workspace_bytes) { mn_set_error("mn_rf_sample_tp: workspace %zu < %zu", workspace_bytes, need); return MN_ENOSPACE; }
  float* pbuf_wide = w.pbuf;                        // the replicated final Linear keeps the wide route's slab area
  if (rows <= 64) w.pbuf = w.pbuf_stream;
  hipStream_t st = mn_stream(stream);
  const int W = h->w, HID = h->hidden, T = h->target, A = h->depth * 3 * W + 2 * W, rpi = rows / n_images;
  const int64_t SR = (int64_t)h->steps * rows;
  const int64_t lo_a = (int64_t)rows * W, lo_b = (int64_t)rows * HID;
  const float step = 1.0f / (float)h->steps;
  int seg = 0;
  uint32_t ep = comm->epoch;
  auto on = [&]() { return seg >= seg_begin && seg < seg_end; };
  WideGlue g;
  mn_g256 a;
  // block b of a step on this rank's hidden units: w12 -> SwiGLU -> w3 partial slabs -> push.  Up to 64 rows the two GEMMs are the
  // weight-streaming MFMA kernels of the <= 64-row route (the shard's bytes are what a launch costs there: 12.6 + 6.3 MB per block at
  // TP = 8 against 100.7 + 50.3 unsharded); above, gemm256 like the wide route.
  const bool streaming = rows <= 64, fused = streaming && g_rf_fuse && rf_fused_shape_ok(h, rows);
  auto block_gemms = [&](int b) -> int {
    int nz;
    float* p3 = w.pbuf;
    if (streaming) {
      nz = stream_dense(h->wfmt, w.ya, h->w12[b], h->wfmt ? h->w12_scale[b] : nullptr, w.pbuf, rows, 2 * HID, W, stream);
      if (nz < 0) return nz;
      if (fused) {                 // the CFG rows of one image: no SwiGLU glue launch (engine.hip, mn_rf_sample)
        const StreamFuse f{w.pbuf, nz, h->b12[b]};
        p3 = w.pbuf_stream3;
        nz = stream_fused(h->wfmt, h->w3[b], h->wfmt ? h->w3_scale[b] : nullptr, p3, rows, W, HID, f, stream);
      } else {
        hipLaunchKernelGGL(rf_glue_swiglu_split_kernel, dim3(mn_cdiv((int64_t)rows * HID, 1024)), dim3(256), 0, st, (const float*)w.pbuf, nz,
                           rows, HID, h->b12[b], w.yb);
        nz = stream_dense(h->wfmt, w.yb, h->w3[b], h->wfmt ? h->w3_scale[b] : nullptr, w.pbuf, rows, W, HID, stream);
      }
      if (nz < 0) return nz;
    } else {
      if (w.ks12 > 1) {
        a = g256_hilo(w.ya, W, lo_a, h->w12[b], W, nullptr, w.pbuf, 2 * HID, rows, 2 * HID, W);
        a.c_zstride = (int64_t)rows * 2 * HID;
        const int nz12 = mn_gemm256_ex(&a, MN_G256_F32, w.ks12, stream);
        if (nz12 < 0) return nz12;
        hipLaunchKernelGGL(rf_swiglu_slabs_kernel, dim3(mn_cdiv((int64_t)rows * (HID / 4), 256)), dim3(256), 0, st, w.pbuf, nz12,
                           (int64_t)rows * 2 * HID, h->b12[b], w.yb, lo_b, rows, HID);
      } else {
        a = g256_hilo(w.ya, W, lo_a, h->w12[b], W, h->b12[b], w.yb, HID, rows, HID, W);
        a.w_pair_rows = HID; a.c_lo_off = lo_b;
        MN_TRYZ(mn_gemm256_ex(&a, MN_G256_SWIGLU_SPLIT, 1, stream));
      }
      a = g256_hilo(w.yb, HID, lo_b, h->w3[b], HID, nullptr, w.pbuf, W, rows, W, HID);
      a.c_zstride = (int64_t)rows * W;
      nz = mn_gemm256_ex(&a, MN_G256_F32, w.ks3, stream);
      if (nz < 0) return nz;
    }
    TpPush p;
    memset(&p, 0, sizeof(p));
    p.P = p3; p.nz = nz; p.slab = (int64_t)rows * W;
    tp_push(comm, ep + 1, p, rows, W, st, gc);
    return 0;
  };
  if (on()) {
    // replicated prologue: z = vis_head(hidden); c = cond_embed(LN(z)); adaLN of all Euler steps; x0 = noise * temperature
    memset(&g, 0, sizeof(g));
    g.h = hidden; g.ldh = ld_hidden; g.Y = w.hs; g.ldy = h->llm_hidden; g.y_lo_off = (int64_t)rows * h->llm_hidden;
    g.M = rows; g.D = h->llm_hidden;
    wide_glue(g, st);
    a = g256_hilo(w.hs, h->llm_hidden, (int64_t)rows * h->llm_hidden, h->vis_w, h->llm_hidden, h->vis_b, w.z, h->z_dim, rows, h->z_dim,
                  h->llm_hidden);
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
    memset(&g, 0, sizeof(g));
    g.h = w.z; g.ldh = h->z_dim; g.norm = 2; g.ng = h->vis_ln_g; g.nb = h->vis_ln_b; g.eps = 1e-6f;
    g.Y = w.zs; g.ldy = h->z_dim; g.y_lo_off = (int64_t)rows * h->z_dim; g.M = rows; g.D = h->z_dim;
    wide_glue(g, st);
    a = g256_hilo(w.zs, h->z_dim, (int64_t)rows * h->z_dim, h->cond_w, h->z_dim, h->cond_b, w.c, W, rows, W, h->z_dim);
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
    hipLaunchKernelGGL(rf_init_x_kernel, dim3(mn_cdiv(rows * T, 256)), dim3(256), 0, st, noise, temperature, w.x, rows, T, rpi);
    hipLaunchKernelGGL(rf_build_y_kernel, dim3(mn_cdiv(SR * W, 256)), dim3(256), 0, st, h->temb, w.c, w.y, h->steps, rows, W);
    a = g256_hilo(w.y, W, SR * W, h->ada_w, W, h->ada_b, w.ada, A, (int)SR, A, W);
    MN_TRYZ(mn_gemm256_ex(&a, MN_G256_F32, 1, stream));
  }
  for (int s = 0; s < h->steps; ++s) {
    const float* ada = w.ada + (int64_t)s * rows * A;
    if (on()) {
      // h = input_proj(x); ya = split(in_ln_0(h) * (1 + scale_0) + shift_0); then block 0 on this rank's hidden units
      memset(&g, 0, sizeof(g));
      g.xin = w.x; g.kin = T; g.win = h->in_w; g.bin = h->in_b; g.h_out = w.hh; g.ldho = W;
      g.norm = 2; g.ng = h->ln_g[0]; g.nb = h->ln_b[0]; g.eps = 1e-6f; g.shift = ada; g.scale = ada + W; g.ldmod = A;
      g.Y = w.ya; g.ldy = W; g.y_lo_off = lo_a; g.M = rows; g.D = W;
      wide_glue(g, st);
      MN_TRYZ(block_gemms(0));
    }
    ++seg; ++ep;
    if (gc) { if (on()) tp_reduce_gather(comm, ep, rows, gc, st); ++seg; ++ep; }
    for (int b = 0; b < h->depth; ++b) {
      const float* mod = ada + (int64_t)b * 3 * W;
      const bool last = b + 1 == h->depth;
      if (on()) {
        // hh += gate * (all-reduced w3 partials + b3); next in_ln / final LN + modulate  (diff_loss:270-272, 290)
        const float* nmod = last ? ada + (int64_t)h->depth * 3 * W : ada + (int64_t)(b + 1) * 3 * W;
        memset(&g, 0, sizeof(g));
        g.h = w.hh; g.ldh = W; tp_consume(g, comm, ep, gc); g.pbias = h->b3[b];
        g.gate = mod + 2 * W; g.ldgate = A; g.h_out = w.hh; g.ldho = W;
        g.norm = 2; g.ng = last ? nullptr : h->ln_g[b + 1]; g.nb = last ? nullptr : h->ln_b[b + 1]; g.eps = 1e-6f;
        g.shift = nmod; g.scale = nmod + W; g.ldmod = A;
        g.Y = w.ya; g.ldy = W; g.y_lo_off = lo_a; g.M = rows; g.D = W;
        wide_glue(g, st);
        if (!last) {
          MN_TRYZ(block_gemms(b + 1));
        } else {
          // replicated tail of the step: final Linear -> bias -> CFG combine + Euler step  (diff_loss:144-179, 291)
          a = g256_hilo(w.ya, W, lo_a, h->fin_w, W, nullptr, pbuf_wide, T, rows, T, W);
          a.c_zstride = (int64_t)rows * T;
          const int nz = mn_gemm256_ex(&a, MN_G256_F32, w.ksf, stream);
          if (nz < 0) return nz;
          hipLaunchKernelGGL(rf_glue_bias_out_kernel, dim3(mn_cdiv(rows * T, 256)), dim3(256), 0, st, pbuf_wide, nz, rows, T, h->fin_b, w.v);
          hipLaunchKernelGGL(rf_euler_kernel, dim3(n_images), dim3(256), 0, st, w.v, w.x, rpi, T, text_cfg, image_cfg, step);
          if (s + 1 == h->steps)
            hipLaunchKernelGGL(rf_gather_latent_kernel, dim3(mn_cdiv(n_images * T, 256)), dim3(256), 0, st, w.x, latent_out, n_images, rpi, T);
        }
      }
      if (!last) {
        ++seg; ++ep;
        if (gc) { if (on()) tp_reduce_gather(comm, ep, rows, gc, st); ++seg; ++ep; }
      }
    }
  }
  if (seg_end == n_seg) comm->epoch += (uint32_t)(ars * h->steps * h->depth);
  MN_CHECK_LAUNCH("mn_rf_sample_tp");
  return MN_OK;
}
